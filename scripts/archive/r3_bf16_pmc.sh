#!/bin/bash
# Dev tool (round 3): counters of the bf16-bound brute-force kernel (three --pmc passes, --kernel-trace only) + kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/a -- python $R/scripts/one_align.py 200000x200000 brute > $O/a.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python $R/scripts/one_align.py 200000x200000 brute > $O/b.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/c -- python $R/scripts/one_align.py 200000x200000 brute > $O/c.log 2>&1; echo rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/scripts/one_align.py 200000x200000 brute > $O/s.log 2>&1; echo rc=$?
python - > $O/bf16_pmc.txt <<PY
import csv, glob, collections
for sub in ("a", "b", "c"):
    acc = collections.defaultdict(list)
    for path in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if "nn_brute_bf16_kernel" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, c), v in per.items():
            acc[c].append(v)
    for c, v in sorted(acc.items()):
        print(f"bf16 {c:28s} mean {sum(v)/len(v):18.1f}  (n={len(v)})")
for path in glob.glob("$O/s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        print("stats", r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat $O/bf16_pmc.txt
tail -2 $O/a.log $O/c.log
