#!/bin/bash
# round 2: the evidence behind EXPERIMENTS.md section 9-f2 (voxel filter without the library sort) -> gpurun_out/r2v/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2v; mkdir -p $O; cd $R
{
echo "## device time (HIP events around the filter's launches, after the bounding box), direct path"
python scripts/voxel_probe.py 2>&1 | grep -v amdgpu.ids
echo "## the same through the library-sort path (ICPGPU_VOXEL_SORT=1: round 1's implementation)"
ICPGPU_VOXEL_SORT=1 python scripts/voxel_probe.py 2>&1 | grep -v amdgpu.ids
echo "## phases of the slowest groups of voxel_group_kernel (ICPGPU_VOXEL_DEBUG=1, device clock)"
python scripts/voxel_debug.py 2>&1 | grep -v amdgpu.ids
echo "## campaign against the oracle"
timeout 900 python scripts/voxel_campaign.py 4000 2>&1 | grep -v amdgpu.ids | tail -3
echo "## HBM-bound kernels table"
python scripts/hbm_kernels.py 2>&1 | grep -v amdgpu.ids
} > $O/voxel.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/voxel_probe.py > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/voxel_kernel_stats.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" >> $O/voxel.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "voxel_hist" in r["Kernel_Name"]]
print("## rocprofv3 kernel trace: one filter call per cloud (start us after the first launch, duration us)")
for name,i0 in (("scan200000 leaf 0.2", idx[3]), ("scan50000 leaf 0.2", idx[15]), ("uniform200000 leaf 0.2", idx[27])):
    t0=int(rows[i0]["Start_Timestamp"]); print(name)
    for r in rows[i0-2:i0+4]:
        print("   %-44s %7.1f %6.1f" % (r["Kernel_Name"].split("::")[-1][:44], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
rm -rf $O/prof
tail -60 $O/voxel.txt
