#!/bin/bash
# round 5: kernel timeline of the reference pipeline with the quadratic inner solver (one scan, kernel by kernel)
TAG=${1:-r5quadtrace}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
( cd /tmp && ICPGPU_GICP_INNER=quadratic timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python $GRAFT_REPO_ROOT/scripts/pipeline_breakdown.py 23 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1 )
f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/$TAG/timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(n):
    m = re.search(r'(\w+)(?:<[^(]*>)?\(', n.replace('(anonymous namespace)', ''))
    return m.group(1) if m else n[:40]
# the last full scan: from the last voxel_hist_kernel but one to the last one
idx = [i for i, r in enumerate(rows) if "voxel_hist_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print(f"# one scan of the pipeline (quadratic inner solver), {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from its first kernel to the next scan's first")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  {nm(r['Kernel_Name']):32s} {(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size', '')} wg {r.get('Workgroup_Size', '')}")
    prev_end = max(prev_end, e)
PY
cat gpurun_out/$TAG/timeline.txt
rm -rf gpurun_out/$TAG/prof
