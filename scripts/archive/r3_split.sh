#!/bin/bash
# NOTE: the kernel variant this script switches (ICPGPU_FLAT / ICPGPU_TWO_LEVEL / ICPGPU_SPLIT) was measured and REMOVED (EXPERIMENTS.md section 5 (table of variants),
# profiles/r03_*): with the shipped library both settings run the same kernel.  Kept as the record of how the numbers were taken.
# Round 3: split fused sweeps (octant stage + nn_cube16_kernel) -- parity, then per-sweep times with and without
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3_split}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_grid.py tests/test_gpu_recognition.py tests/test_gpu_sequence.py -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -5
for rep in 1 2; do
  for f in 0 1; do
    for s in 200000x200000 50000x50000 200000x1000000; do echo -n "SPLIT=$f "; ICPGPU_SPLIT=$f python scripts/iter_profile.py $s 2>/dev/null; done
  done
done | tee $O/iter.txt
for f in 0 1; do echo -n "SPLIT=$f bench: "; ICPGPU_SPLIT=$f python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s', round(d['ms_per_step'],4), 'ms/step')"; done | tee -a $O/iter.txt
