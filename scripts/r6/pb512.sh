#!/bin/bash
# round 6: 512 correspondences per workgroup of the evaluation server -- tests and campaigns (the sums must not depend on the split), rates
TAG=${1:-r6pb512}
O=gpurun_out/$TAG; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py tests/test_cpp_shim.py tests/test_gpu_mailbox.py -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/gicp_campaign.py 12600 13000 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python scripts/pipeline_campaign.py 2900 3050 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python scripts/batch_campaign.py 840 870 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2 3; do timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of\|host wall"; done
timeout 120 python scripts/r5_pipeline_on_bench_pair.py 2>&1 | tail -1
timeout 200 python scripts/gicp_timing.py 50000x50000 200000x200000 2>&1 | grep -v amdgpu.ids | cut -c1-220
} > $O/out.txt 2>&1
cat $O/out.txt
