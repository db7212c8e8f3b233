#!/bin/bash
# Dev tool: memory-hierarchy counters of the grid NN kernel (rocprofv3 --pmc passes, --kernel-trace only).
TAG=${1:-mem}
SIZE=${2:-200000x200000}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG/mem
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCC_BUSY_avr TCC_TAG_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/scripts/one_align.py $SIZE grid > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if ("nn_quad_kernel<false, true, false" in r["Kernel_Name"] or "nn_wave_kernel<false, true, false" in r["Kernel_Name"]):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c, v in sorted(acc.items()):
    print(f"{c:36s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
