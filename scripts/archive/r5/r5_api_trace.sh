#!/bin/bash
# round 5: HIP API + kernel timeline of one pipeline scan (quadratic inner solver): which host calls sit in the gaps
TAG=${1:-r5api}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
( cd /tmp && ICPGPU_GICP_INNER=quadratic timeout 900 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python $GRAFT_REPO_ROOT/scripts/pipeline_breakdown.py 13 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1 )
ls gpurun_out/$TAG/prof/*/ | head
python - gpurun_out/$TAG/prof > gpurun_out/$TAG/api_timeline.txt <<'PY'
import csv, sys, glob, re
d = sys.argv[1]
kf = glob.glob(d + "/*/*kernel_trace.csv")[0]
af = glob.glob(d + "/*/*hip_api_trace.csv")
af = af[0] if af else None
K = list(csv.DictReader(open(kf)))
K.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(K) if "voxel_hist_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0, t1 = int(K[a]["Start_Timestamp"]), int(K[b]["Start_Timestamp"])
ev = []
def nm(n):
    m = re.search(r'(\w+)(?:<[^(]*>)?\(', n.replace('(anonymous namespace)', ''))
    return m.group(1) if m else n[:40]
for r in K[a:b]:
    ev.append((int(r["Start_Timestamp"]), "GPU  " + nm(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
if af:
    A = list(csv.DictReader(open(af)))
    print("# api columns:", list(A[0].keys()))
    for r in A:
        s = int(r["Start_Timestamp"])
        if t0 - 200000 <= s < t1:
            ev.append((s, "host " + r.get("Function", r.get("Name", "?")), (int(r["End_Timestamp"]) - s) / 1e3))
ev.sort()
for s, n, dur in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  {n:48s} {dur:8.1f} us")
PY
head -150 gpurun_out/$TAG/api_timeline.txt
rm -rf gpurun_out/$TAG/prof
