#!/bin/bash
# round 5: the parity campaigns on the round's LAST build (after the quadratic GICP solver) -> gpurun_out/r5cf/campaigns.txt
O=gpurun_out/r5cf; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 95000 96500 (grid keys == brute force; 1500 seeds)"
timeout 2400 python scripts/fuzz_campaign.py 95000 96500 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/align_campaign.py 6300 6500 (200 whole point-to-point alignments against the oracle)"
timeout 2400 python scripts/align_campaign.py 6300 6500 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/gicp_campaign.py 11600 12000 (400 whole GICP registrations against the oracle; default = EXACT inner solver, measured host / device choice)"
timeout 2400 python scripts/gicp_campaign.py 11600 12000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## ICPGPU_GICP_DEVICE=1 scripts/gicp_campaign.py 11600 11900 (300 of them through the device solver)"
ICPGPU_GICP_DEVICE=1 timeout 2400 python scripts/gicp_campaign.py 11600 11900 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/batch_campaign.py 620 680 (60 random batches through icpgpu_align_batch against single aligns)"
timeout 3000 python scripts/batch_campaign.py 620 680 2>&1 | grep -v amdgpu.ids | tail -1
echo "## ICPGPU_GICP_INNER=quadratic scripts/batch_campaign.py 680 740 (60 random batches, GICP pairs with the QUADRATIC inner solver in batches and in single aligns alike)"
ICPGPU_GICP_INNER=quadratic timeout 3000 python scripts/batch_campaign.py 680 740 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/voxel_campaign.py 500"
timeout 2400 python scripts/voxel_campaign.py 500 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/pipeline_campaign.py 2400 2600 (200 pairs through the reference's per-scan pipeline, bit for bit; default inner solver)"
timeout 2400 python scripts/pipeline_campaign.py 2400 2600 2>&1 | grep -v amdgpu.ids | tail -1
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
