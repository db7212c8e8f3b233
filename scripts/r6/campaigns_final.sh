#!/bin/bash
# round 6: the parity campaigns on the round's LAST build (fresh seeds), as profiles/r06_campaigns.txt
TAG=${1:-r6campfinal}
O=gpurun_out/$TAG; mkdir -p $O
run() { echo "## $1"; shift; ( timeout 900 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
{
echo "Round 6 -- parity campaigns on the round's last build (scripts/r6/campaigns_final.sh; one MI355X gpurun box)"
echo
run "scripts/fuzz_campaign.py 97000 99000 (grid keys == brute force; 2000 seeds)" python scripts/fuzz_campaign.py 97000 99000
FUZZ_ALL=1 run "FUZZ_ALL=1 scripts/fuzz_campaign.py 99000 99300 (+ map and voxel filter against the oracle; 300 seeds)" env FUZZ_ALL=1 python scripts/fuzz_campaign.py 99000 99300
run "scripts/align_campaign.py 6500 6700 (200 whole point-to-point alignments against the oracle)" python scripts/align_campaign.py 6500 6700
run "scripts/gicp_campaign.py 12000 12600 (600 whole GICP registrations against the oracle; exact inner solver, host loop)" python scripts/gicp_campaign.py 12000 12600
run "ICPGPU_GICP_DEVICE=1 scripts/gicp_campaign.py 12000 12300 (300 of them through the device solver)" env ICPGPU_GICP_DEVICE=1 python scripts/gicp_campaign.py 12000 12300
run "scripts/batch_campaign.py 740 800 (60 random batches through icpgpu_align_batch -- P2P lock-step groups, GICP resumable runs -- against single aligns)" python scripts/batch_campaign.py 740 800
run "ICPGPU_GICP_INNER=quadratic scripts/batch_campaign.py 800 840 (40 random batches, the QUADRATIC inner solver in batches and in single aligns alike)" env ICPGPU_GICP_INNER=quadratic python scripts/batch_campaign.py 800 840
run "scripts/voxel_campaign.py 2000 (the filter queued behind the box pass, parameters derived on the device)" python scripts/voxel_campaign.py 2000
run "scripts/cov_campaign.py 300 700 (GICP covariances: selecting + far-field kernels; 400 clouds of six kinds)" python scripts/cov_campaign.py 300 700
run "scripts/pipeline_campaign.py 2600 2900 (300 pairs through the reference's per-scan pipeline, bit for bit)" python scripts/pipeline_campaign.py 2600 2900
} > $O/campaigns.txt 2>&1
cat $O/campaigns.txt
