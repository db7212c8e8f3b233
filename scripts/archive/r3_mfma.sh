#!/bin/bash
# round 3: matrix-core brute force with software-pipelined folds (ICPGPU_MFMA_PIPE 0 = round 2's order, 1 = pipelined,
# 2 = pipelined and held to 128 registers) -> gpurun_out/r3m/
O=gpurun_out/r3m; mkdir -p $O
{
echo "## parity first (default = PIPE 1)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid.py tests/test_gpu_recognition.py -m gpu -x -q 2>&1 | tail -3
ICPGPU_MFMA_PIPE=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for P in 0 1 2; do
  for W in 16 32; do
    echo "## PIPE=$P WAVES=$W"
    ICPGPU_MFMA_PIPE=$P ICPGPU_MFMA_WAVES=$W timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids | grep "matrix"
  done
done
echo "## PIPE=1 G=4"
ICPGPU_MFMA_PIPE=1 ICPGPU_MFMA_G=4 timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids | grep "matrix"
echo "## PIPE=1, no exact path (timing experiment: results wrong)"
ICPGPU_MFMA_PIPE=1 ICPGPU_MFMA_NO_EXACT=1 timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep -v amdgpu.ids | grep "matrix"
echo "## other sizes, PIPE=1 / 0"
ICPGPU_MFMA_PIPE=1 timeout 900 python scripts/brute_timing.py 50000x50000 200000x1000000 2>&1 | grep -v amdgpu.ids
ICPGPU_MFMA_PIPE=0 timeout 900 python scripts/brute_timing.py 50000x50000 200000x1000000 2>&1 | grep -v amdgpu.ids | grep matrix
} > $O/mfma.txt 2>&1
cat $O/mfma.txt
