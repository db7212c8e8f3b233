#!/bin/bash
# Dev tool (round 2): voxel filter -- parity tests, device time per cloud, per-kernel timeline under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_parity_golden.py -x -q 2>&1 | tail -5
python scripts/voxel_probe.py 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/vox3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vox3 -- python $R/scripts/voxel_probe.py > /dev/null 2>&1
t=$(find $R/gpurun_out/vox3 -name "*kernel_trace.csv" | head -1)
python - "$t" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "voxel_hist" in r["Kernel_Name"]]
for i0 in (idx[3], idx[9], idx[15], idx[33]):
    t0=int(rows[i0]["Start_Timestamp"])
    for r in rows[i0-2:i0+5]:
        print(r["Kernel_Name"].split("::")[-1][:40], round((int(r["Start_Timestamp"])-t0)/1e3,1), round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,1))
    print()
PY
