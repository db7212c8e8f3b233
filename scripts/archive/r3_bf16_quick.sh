timeout 600 python -m pytest tests/test_gpu_brute_bf16.py -m gpu -x -q 2>&1 | tail -1
BRUTE_VARIANTS=0 timeout 600 python scripts/brute_timing.py 200000x200000 200000x1000000 2>&1 | grep matrix
for cfg in "2 512 256" "2 512 512" "4 512 256"; do set -- $cfg; echo "## G=$1 TILE=$2 BLOCK=$3"
BRUTE_VARIANTS=0 ICPGPU_BF16_G=$1 ICPGPU_BF16_TILE=$2 ICPGPU_BF16_BLOCK=$3 timeout 600 python scripts/brute_timing.py 200000x200000 2>&1 | grep matrix; done
