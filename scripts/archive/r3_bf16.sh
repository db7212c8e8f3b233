#!/bin/bash
# round 3: brute force with the lower bound on the bf16 matrix path -> gpurun_out/r3b/
O=gpurun_out/r3b; mkdir -p $O
{
echo "## parity first"
timeout 1500 python -m pytest tests/test_gpu_brute_bf16.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -12
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
echo "## timing, defaults (G=2 TILE=1024 BLOCK=512)"
timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids
for cfg in "2 512 256" "2 512 512" "4 512 256" "4 1024 512"; do
  set -- $cfg
  echo "## G=$1 TILE=$2 BLOCK=$3"
  BRUTE_VARIANTS=0 ICPGPU_BF16_G=$1 ICPGPU_BF16_TILE=$2 ICPGPU_BF16_BLOCK=$3 timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids | grep matrix
done
echo "## no exact path (timing experiment: results wrong), defaults"
BRUTE_VARIANTS=0 ICPGPU_MFMA_NO_EXACT=1 timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids | grep matrix
echo "## other sizes"
BRUTE_VARIANTS=0,2 timeout 900 python scripts/brute_timing.py 50000x50000 200000x1000000 2>&1 | grep -v amdgpu.ids
} > $O/bf16_v3.txt 2>&1
cat $O/bf16_v3.txt
