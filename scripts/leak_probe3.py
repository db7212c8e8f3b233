"""Dev tool: does device (or host-mapped) memory drift with the number of contexts created and destroyed?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icpslam_amd import Context, GICP, synth
src, tgt, _ = synth.make_pair(20000, 20000, seed=1)
free0 = torch.cuda.mem_get_info()[0]
done = 0
for target in (1, 50, 100, 200, 400):
    while done < target:
        with Context(0) as ctx:
            ctx.set_params(ctx.default_params(), method=GICP if done % 2 else 0, max_iterations=3)
            ctx.set_source(src); ctx.set_target(tgt); ctx.align(want_fitness=True)
        done += 1
    print(f"{done} contexts: drift {(free0 - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB", flush=True)
