#!/bin/bash
# round 3: the parity campaigns on the round's final build -> gpurun_out/r3c/campaigns.txt
O=gpurun_out/r3c; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 40000 41500 (grid keys == brute force; quad kernel + previous-neighbour bound; random clouds)"
timeout 2400 python scripts/fuzz_campaign.py 40000 41500 2>&1 | grep -v amdgpu.ids | tail -2
echo "## FUZZ_ALL=1 scripts/fuzz_campaign.py 50000 50300 (+ map and voxel filter against the oracle)"
FUZZ_ALL=1 timeout 2400 python scripts/fuzz_campaign.py 50000 50300 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/align_campaign.py 3000 3150 (whole point-to-point alignments of 33k-60k points against the oracle)"
timeout 2400 python scripts/align_campaign.py 3000 3150 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/gicp_campaign.py 7000 7400 (whole GICP registrations against the oracle, exact-sum definition; Eigen JacobiSVD restated on both sides)"
timeout 2400 python scripts/gicp_campaign.py 7000 7400 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/voxel_campaign.py 2000  (direct path)"
timeout 2400 python scripts/voxel_campaign.py 2000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## ICPGPU_VOXEL_SORT=1 scripts/voxel_campaign.py 1000  (sort path: the hand-written radix sort + scans of icp_scan.hip)"
ICPGPU_VOXEL_SORT=1 timeout 2400 python scripts/voxel_campaign.py 1000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/pipeline_campaign.py 0 60 (the reference's per-scan pipeline on random raw scans, bit for bit)"
timeout 1200 python scripts/pipeline_campaign.py 0 60 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
