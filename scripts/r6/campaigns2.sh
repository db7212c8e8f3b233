#!/bin/bash
# round 6: the campaigns of earlier rounds on the round's LAST build (what changed under them: staged result clouds, the voxel filter's plan, the covariances)
TAG=${1:-r6camp3}
O=gpurun_out/$TAG; mkdir -p $O
{
echo "== scripts/voxel_campaign.py 1500"; timeout 600 python scripts/voxel_campaign.py 1500 2>&1 | grep -v amdgpu.ids | tail -2
echo "== scripts/pipeline_campaign.py 0 60"; timeout 900 python scripts/pipeline_campaign.py 0 60 2>&1 | grep -v amdgpu.ids | tail -3
echo "== scripts/align_campaign.py 0 60"; timeout 900 python scripts/align_campaign.py 0 60 2>&1 | grep -v amdgpu.ids | tail -3
echo "== scripts/gicp_campaign.py 150 400"; timeout 600 python scripts/gicp_campaign.py 150 400 2>&1 | grep -v amdgpu.ids | tail -2
echo "== scripts/fuzz_campaign.py 0 150"; timeout 900 python scripts/fuzz_campaign.py 0 150 2>&1 | grep -v amdgpu.ids | tail -3
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
