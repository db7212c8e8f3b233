#!/bin/bash
# Dev tool (round 2): instruction-mix counters of nn_quad_kernel's COLD sweep (max_iterations = 1), per env setting.
# usage: r2_pmc.sh <tag> <iters>   (env ICPGPU_* passes through)
TAG=${1:-pmc}; ITERS=${2:-1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python $R/scripts/cold_sweeps.py 200000x200000 $ITERS 12 > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python $R/scripts/cold_sweeps.py 200000x200000 $ITERS 12 > $O/b.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(list)
    for path in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if "nn_quad_kernel<false, true, false" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, c), v in per.items():
            acc[c].append(v)
    for c, v in sorted(acc.items()):
        print(f"$TAG {c:24s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
