#!/bin/bash
# round 5: the parity campaigns on the round's build -> gpurun_out/r5c/campaigns.txt
O=gpurun_out/r5c; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 90000 93000 (grid keys == brute force; quad kernel with the previous neighbour kept as a POSITION; 3000 seeds)"
timeout 2400 python scripts/fuzz_campaign.py 90000 93000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## FUZZ_ALL=1 scripts/fuzz_campaign.py 94000 94400 (+ map and voxel filter against the oracle; 400 seeds)"
FUZZ_ALL=1 timeout 2400 python scripts/fuzz_campaign.py 94000 94400 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/align_campaign.py 6000 6300 (300 whole point-to-point alignments against the oracle)"
timeout 2400 python scripts/align_campaign.py 6000 6300 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/gicp_campaign.py 11000 11600 (600 whole GICP registrations against the oracle; measured solver choice)"
timeout 2400 python scripts/gicp_campaign.py 11000 11600 2>&1 | grep -v amdgpu.ids | tail -1
echo "## ICPGPU_GICP_DEVICE=1 scripts/gicp_campaign.py 11000 11400 (400 of them through the device solver)"
ICPGPU_GICP_DEVICE=1 timeout 2400 python scripts/gicp_campaign.py 11000 11400 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/batch_campaign.py 500 620 (120 random batches through icpgpu_align_batch -- P2P lock-step groups, GICP resumable runs -- against single aligns)"
timeout 3000 python scripts/batch_campaign.py 500 620 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/voxel_campaign.py 1000"
timeout 2400 python scripts/voxel_campaign.py 1000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/pipeline_campaign.py 2000 2400 (400 pairs through the reference's per-scan pipeline, bit for bit)"
timeout 2400 python scripts/pipeline_campaign.py 2000 2400 2>&1 | grep -v amdgpu.ids | tail -1
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
