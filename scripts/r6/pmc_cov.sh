#!/bin/bash
# round 6: instruction-issue counters of the covariance kernels (one rocprofv3 --pmc pass with --kernel-trace only) over the
# reference pipeline's drive (scripts/pipeline_breakdown.py): mean per launch, per kernel
TAG=${1:-r6pmccov}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python $R/scripts/pipeline_breakdown.py 13 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python $R/scripts/pipeline_breakdown.py 13 > $O/t.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$O/a/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "gicp_cov" in r["Kernel_Name"]:
            m = re.search(r"(gicp_cov\w+)", r["Kernel_Name"]).group(1)
            per[(m, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
dur = {}
for path in glob.glob("$O/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.reader(open(path)):
        if r and "gicp_cov" in r[0]:
            dur[re.search(r"(gicp_cov\w+)", r[0]).group(1)] = float(r[3]) / 1e3
lines = ["rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace (own pass) over scripts/pipeline_breakdown.py 13:",
         "mean per launch of the covariance kernels on a voxel-filtered ~23k-point cloud (durations: a separate --kernel-trace --stats run)"]
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    us = dur.get(k)
    valu = m.get("SQ_INSTS_VALU", 0.0)
    lines.append(f"{k}: {len(next(iter(acc[k].values())))} launches, {us:.1f} us; wave-instructions per launch: VALU {valu:.3g}, SALU {m.get('SQ_INSTS_SALU', 0):.3g}, "
                 f"VMEM reads {m.get('SQ_INSTS_VMEM_RD', 0):.3g}, LDS {m.get('SQ_INSTS_LDS', 0):.3g}; waves {m.get('SQ_WAVES', 0):.0f}; "
                 f"VALU per wave {valu / max(m.get('SQ_WAVES', 1), 1):.0f}; VALU issue rate {valu / (us * 1e-6) / 1e9 if us else 0:.0f} G wave-instr/s "
                 f"(full-rate peak 1228.8)")
open("$O/pmc_cov.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $O/a $O/t
