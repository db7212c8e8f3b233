#!/bin/bash
# round 6: campaigns on the round's new kernels (covariances: selecting / far-field / streaming; voxel filter queued behind the box; GICP end to end)
TAG=${1:-r6camp}
O=gpurun_out/$TAG; mkdir -p $O
{
timeout 1200 python scripts/cov_campaign.py 0 300 2>&1 | grep -v amdgpu.ids
[ -n "$QUICK" ] || timeout 600 python scripts/voxel_campaign.py 1200 2>&1 | grep -v amdgpu.ids | tail -3
[ -n "$QUICK" ] || timeout 600 python scripts/gicp_campaign.py 0 150 2>&1 | grep -v amdgpu.ids | tail -3
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
