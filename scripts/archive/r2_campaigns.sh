#!/bin/bash
# round 2: the parity campaigns on the round's final build -> gpurun_out/r2c/campaigns.txt
O=gpurun_out/r2c; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 20000 23000 (grid keys == brute force; quad kernel + previous-neighbour bound; random clouds)"
timeout 2400 python scripts/fuzz_campaign.py 20000 23000 2>&1 | grep -v amdgpu.ids | tail -2
echo "## FUZZ_ALL=1 scripts/fuzz_campaign.py 30000 30400 (+ map and voxel filter against the oracle)"
FUZZ_ALL=1 timeout 2400 python scripts/fuzz_campaign.py 30000 30400 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/align_campaign.py 2000 2200 (whole point-to-point alignments of 33k-60k points against the oracle)"
timeout 2400 python scripts/align_campaign.py 2000 2200 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/gicp_campaign.py 5000 5400 (whole GICP registrations against the oracle, exact-sum definition)"
timeout 2400 python scripts/gicp_campaign.py 5000 5400 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/voxel_campaign.py 6000"
timeout 2400 python scripts/voxel_campaign.py 6000 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
