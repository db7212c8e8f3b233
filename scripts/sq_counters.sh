#!/bin/bash
# Dev tool: instruction-mix counters of the grid NN kernel (one rocprofv3 --pmc pass, --kernel-trace only).
TAG=${1:-sq}
SIZE=${2:-200000x200000}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG/sq
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python $R/scripts/one_align.py $SIZE grid > $O/a.log 2>&1
echo rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python $R/scripts/one_align.py $SIZE grid > $O/b.log 2>&1
echo rc=$?
python - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(list)
    for path in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if ("nn_quad_kernel<false, true, false" in r["Kernel_Name"] or "nn_wave_kernel<false, true, false" in r["Kernel_Name"]):
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, c), v in per.items():
            acc[c].append(v)
    for c, v in sorted(acc.items()):
        print(f"{c:24s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
