#!/bin/bash
# NOTE: the kernel variant this script switches (ICPGPU_FLAT / ICPGPU_TWO_LEVEL / ICPGPU_SPLIT) was measured and REMOVED (EXPERIMENTS.md section 5 (table of variants),
# profiles/r03_*): with the shipped library both settings run the same kernel.  Kept as the record of how the numbers were taken.
# Round 3: A/B of the cube search's row walk in nn_quad_kernel (ICPGPU_FLAT=0: two rows per step, 1: one flattened list).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3_flat}; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_errors.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  for f in 0 1; do
    for s in 200000x200000 50000x50000 200000x1000000; do echo -n "FLAT=$f "; ICPGPU_FLAT=$f python scripts/iter_profile.py $s 2>/dev/null; done
  done
done | tee $O/iter.txt
