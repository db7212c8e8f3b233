#!/bin/bash
# round 3: the grid search on the matrix cores (icp_tile.hip, ICPGPU_TILE_SEARCH=1) against the shipped grid search
O=gpurun_out/r3t; mkdir -p $O
{
echo "## shipped"; timeout 600 python scripts/tile_check.py $O/a.npz 2>&1 | grep -v amdgpu.ids
for S in 1 2 4 8; do
echo "## tile search, splits $S"; ICPGPU_TILE_SPLITS=$S ICPGPU_TILE_SEARCH=1 timeout 600 python scripts/tile_check.py $O/b$S.npz 2>&1 | grep -v amdgpu.ids
python scripts/tile_compare.py $O/a.npz $O/b$S.npz
done
echo "## pairs offered (splits 4)"; ICPGPU_TILE_SEARCH=2 timeout 600 python scripts/tile_check.py $O/c.npz 2>&1 | grep "tile search" | head -14
echo "## sweeps by rocprofv3 (splits 4)"
export TMPDIR=/tmp
(cd /tmp && ICPGPU_TILE_SEARCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/scripts/one_align.py 200000x200000 > /dev/null 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
} > $O/tile.txt 2>&1
cat $O/tile.txt
