#!/bin/bash
# round 3: the MFMA / vector co-execution probe + the pipeline campaign leg that r3_campaigns.sh called without arguments
O=gpurun_out/r3p; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_coissue.cpp -o /tmp/mfma_coissue 2>/dev/null && timeout 300 /tmp/mfma_coissue > $O/mfma_coissue.txt 2>&1
cat $O/mfma_coissue.txt
{ echo "## scripts/pipeline_campaign.py 0 60 (the reference's per-scan pipeline on random raw scans, bit for bit)"
  timeout 1200 python scripts/pipeline_campaign.py 0 60 2>&1 | grep -v amdgpu.ids | tail -2; } > $O/pipeline_campaign.txt 2>&1
cat $O/pipeline_campaign.txt
