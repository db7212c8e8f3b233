"""Dev tool: does initialising torch's HIP context change the per-iteration host overhead of align()?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode != "plain":
    import torch
    torch.cuda.set_device(0)
    if mode == "tensor":
        x = torch.zeros(16, device="cuda"); torch.cuda.synchronize()
from icpslam_amd import Context, synth, NN_AUTO
src, tgt, _ = synth.make_pair(200000, 200000, seed=4)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=NN_AUTO)
    ctx.set_source(src); ctx.set_target(tgt)
    for _ in range(3): ctx.align()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(20): ctx.align()
    wall = (time.perf_counter() - t0) / 20
    p = ctx.profile()
    print(f"{mode:8s}: align(10) {wall*1e3:.3f} ms, NN kernel {p.grid_ms/max(1,p.grid_timed)*1e3:.1f} us, reduce {p.reduce_ms/max(1,p.reduce_timed)*1e3:.1f} us")
