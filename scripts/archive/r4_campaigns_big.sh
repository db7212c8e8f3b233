#!/bin/bash
# round 4, last build: larger parity campaigns (seed ranges disjoint from scripts/r4_campaigns.sh) -> gpurun_out/r4c/campaigns_big.txt
O=gpurun_out/r4c; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 80000 84000 (grid keys == brute force; 4000 seeds)"
timeout 2400 python scripts/fuzz_campaign.py 80000 84000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## FUZZ_ALL=1 scripts/fuzz_campaign.py 85000 85600 (+ map and voxel filter against the oracle; 600 seeds)"
FUZZ_ALL=1 timeout 2400 python scripts/fuzz_campaign.py 85000 85600 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/align_campaign.py 5000 5400 (400 whole point-to-point alignments against the oracle)"
timeout 2400 python scripts/align_campaign.py 5000 5400 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/gicp_campaign.py 9000 10500 (1500 whole GICP registrations against the oracle; host solver over the evaluation server)"
timeout 2400 python scripts/gicp_campaign.py 9000 10500 2>&1 | grep -v amdgpu.ids | tail -1
echo "## ICPGPU_GICP_DEVICE=1 scripts/gicp_campaign.py 9000 9800 (800 of them through the device solver)"
ICPGPU_GICP_DEVICE=1 timeout 2400 python scripts/gicp_campaign.py 9000 9800 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/voxel_campaign.py 3000 (3000 clouds through the voxel filter)"
timeout 2400 python scripts/voxel_campaign.py 3000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/pipeline_campaign.py 1000 1800 (800 pairs through the reference's per-scan pipeline: filter's box, cell-size hint, adopted grid -- bit for bit)"
timeout 2400 python scripts/pipeline_campaign.py 1000 1800 2>&1 | grep -v amdgpu.ids | tail -1
} > $O/campaigns_big.txt 2>&1
cat $O/campaigns_big.txt
