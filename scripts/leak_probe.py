import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from icpslam_amd import Context, synth
src, tgt, _ = synth.make_pair(60000, 60000, seed=1)
torch.cuda.synchronize()
def free(): return torch.cuda.mem_get_info()[0] / 2**20
f0 = free()
n = 0
for stage in (50, 100, 150, 300):
    while n < stage:
        with Context(0) as ctx:
            if os.environ.get("LEAK_MODE") != "create_only":
                ctx.set_params(ctx.default_params()); ctx.set_source(src[:5000 + n * 10]); ctx.set_target(tgt); ctx.align(want_fitness=True)
        n += 1
    print(n, "contexts: drift %.1f MiB" % (f0 - free()), flush=True)
