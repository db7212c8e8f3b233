#!/bin/bash
# Round 3: lock-step batch path -- parity tests, then pairs/s with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3_batch}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py tests/test_gpu_multi.py -m gpu -x -q -k "batch or sequence or multi" > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -5
for cfg in "1 0 4 2" "1 1 2 8" "2 1 1 8" "3 1 1 16" "4 1 2 16" "5 1 2 4" "6 1 4 8"; do
  set -- $cfg
  echo -n "LOCKSTEP=$2 threads=$3 depth=$4: "
  ICPGPU_BATCH_LOCKSTEP=$2 ICPGPU_BATCH_THREADS=$3 ICPGPU_BATCH_DEPTH=$4 timeout 300 python bench.py --workload batch50k --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['scan_pairs_per_sec']), 'pairs/s', round(d['ms_per_step'],2), 'ms/step')"
done | tee $O/rates.txt
