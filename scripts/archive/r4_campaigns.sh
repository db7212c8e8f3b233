#!/bin/bash
# round 4: the parity campaigns on the round's final build -> gpurun_out/r4c/campaigns.txt
O=gpurun_out/r4c; mkdir -p $O
{
echo "## scripts/fuzz_campaign.py 70000 70800 (grid keys == brute force; quad kernel + previous-neighbour bound; random clouds)"
timeout 2400 python scripts/fuzz_campaign.py 70000 70800 2>&1 | grep -v amdgpu.ids | tail -2
echo "## FUZZ_ALL=1 scripts/fuzz_campaign.py 71000 71200 (+ map -- PCL's first box per getKeyBitSize -- and voxel filter against the oracle)"
FUZZ_ALL=1 timeout 2400 python scripts/fuzz_campaign.py 71000 71200 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/align_campaign.py 4000 4150 (whole point-to-point alignments of 33k-60k points against the oracle; self-validating mailbox pairs)"
timeout 2400 python scripts/align_campaign.py 4000 4150 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/gicp_campaign.py 8000 8400 (whole GICP registrations against the oracle: exact sums + correctly rounded trig; host solver over the evaluation server)"
timeout 2400 python scripts/gicp_campaign.py 8000 8400 2>&1 | grep -v amdgpu.ids | tail -2
echo "## ICPGPU_GICP_DEVICE=1 scripts/gicp_campaign.py 8000 8400 (the same 400 registrations through the device solver, gicp_solve_kernel)"
ICPGPU_GICP_DEVICE=1 timeout 2400 python scripts/gicp_campaign.py 8000 8400 2>&1 | grep -v amdgpu.ids | tail -2
echo "## ICPGPU_MAILBOX=release scripts/align_campaign.py 4000 4040 (the release form of the result pairs)"
ICPGPU_MAILBOX=release timeout 1200 python scripts/align_campaign.py 4000 4040 2>&1 | grep -v amdgpu.ids | tail -2
echo "## scripts/voxel_campaign.py 1000  (direct path)"
timeout 2400 python scripts/voxel_campaign.py 1000 2>&1 | grep -v amdgpu.ids | tail -1
echo "## scripts/pipeline_campaign.py 100 300 (the reference's per-scan pipeline on random raw scans, bit for bit)"
timeout 1200 python scripts/pipeline_campaign.py 100 300 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
