#!/bin/bash
# Round 3, first GPU job: (a) counter list of the box, (b) points-per-wave sweep of nn_quad_kernel (tail / occupancy question),
# (c) the discriminating PMC passes VERDICT r2 item 1 asks for, on the whole kernel and on the octant stage alone
# (ICPGPU_SKIP_UNCERT: uncertified points dropped -- timing only).   usage: r3_diag.sh [tag]
TAG=${1:-r3_diag}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
for q in 16 12 8; do
  for rep in 1 2; do echo -n "QPW=$q: "; ICPGPU_QPW=$q python $R/scripts/iter_profile.py 200000x200000; done
done 2>&1 | tee $O/qpw.txt
pmc() {  # pmc <name> <env> <counters...>
  local name=$1 envs=$2; shift 2
  env $envs rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python $R/scripts/one_align.py 200000x200000 grid > $O/$name.log 2>&1
}
for V in all oct; do
  E="X=1"; [ $V = oct ] && E="ICPGPU_SKIP_UNCERT=1"
  pmc ${V}_a "$E" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc ${V}_b "$E" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
  pmc ${V}_c "$E" SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
  pmc ${V}_d "$E" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pmc ${V}_e "$E" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  pmc ${V}_f "$E" TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum
done
python - <<PY | tee $O/summary.txt
import csv, glob, collections
for V in ("all", "oct"):
    for sub in "abcdef":
        acc = collections.defaultdict(list)
        for path in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (V, sub), recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(path)):
                if "nn_quad_kernel<false, true, false" in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (d, c), v in per.items():
                acc[c].append(v)
        for c, v in sorted(acc.items()):
            print(f"{V} {c:32s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
tail -3 $O/*.log | head -80
