mkdir -p gpurun_out/r3c
{ echo "## scripts/bf16_campaign.py 0 4000 (bf16-bound brute-force kernel vs the vector kernel, every pair's bound checked exactly)"
timeout 2400 python scripts/bf16_campaign.py 0 4000 2>&1 | grep -v amdgpu.ids | tail -4
echo "## scripts/fuzz_campaign.py 60000 60400 (grid keys == brute force, now the bf16-bound kernel from 8192 points on)"
timeout 1500 python scripts/fuzz_campaign.py 60000 60400 2>&1 | grep -v amdgpu.ids | tail -1; } > gpurun_out/r3c/bf16_campaign.txt 2>&1
cat gpurun_out/r3c/bf16_campaign.txt
