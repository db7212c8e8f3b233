"""Dev tool: does the oracle scale over host threads (ctypes releases the GIL)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from icpslam_amd import synth
from concurrent.futures import ThreadPoolExecutor
src, tgt, _ = synth.make_pair(50000, 50000, seed=4)
p = oracle.default_params(max_iterations=10, force_iterations=1)
oracle.icp_align(src, tgt, p)
for n in (1, 2, 4, 8, 16, 32, 64):
    t = time.perf_counter()
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(lambda _: oracle.icp_align(src, tgt, p)["iterations"], range(n)))
    dt = time.perf_counter() - t
    print(f"{n:3d} threads: {dt:.3f} s -> {10*n/dt:.0f} it/s", flush=True)
