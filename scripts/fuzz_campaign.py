"""Dev tool: the GPU fuzz tests of tests/test_gpu_grid.py, tests/test_gpu_map.py and tests/test_gpu_voxel.py over many more
seeds than the suite runs.  Usage: python scripts/fuzz_campaign.py FIRST LAST"""
import os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from icpslam_amd import Context
import test_gpu_grid as G
import test_gpu_map as M
import test_gpu_voxel as V

first, last = int(sys.argv[1]), int(sys.argv[2])
fails = 0
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        fns = (G.test_fuzz_grid_keys_equal_brute_force, G.test_fuzz_quad_kernel_and_previous_neighbour_bound)
        if os.environ.get("FUZZ_ALL"):  # the map and the voxel filter too (CPU oracle inside: slower)
            fns += (M.test_fuzz_map_against_oracle, V.test_fuzz_voxel_filter_against_oracle)
        for fn in fns:
            try:
                fn(ctx, seed)
            except Exception:
                fails += 1
                print(f"FAIL {fn.__name__} seed {seed}\n{traceback.format_exc()[-1500:]}", flush=True)
        if seed % 20 == 0:
            print(f"seed {seed} done, {fails} failures, {time.time()-t0:.0f} s", flush=True)
print(f"campaign {first}..{last}: {fails} failures in {time.time()-t0:.0f} s")
