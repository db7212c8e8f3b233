import sys, time
sys.path.insert(0, "/root/repo")
from icpslam_amd import Context
c0 = Context(0)
t0 = time.perf_counter(); cs = [Context(0) for _ in range(16)]; dt = time.perf_counter() - t0
print(f"icpgpu_create: {dt / 16 * 1e3:.2f} ms per context")
