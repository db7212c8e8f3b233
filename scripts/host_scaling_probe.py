"""Round 4: how does the C multi-GPU entry's HOST side scale?  icpgpu_align_batch_multi gives every device
entry its own host thread, and each of those drives batch threads that spin on mailboxes -- on an 8-GPU node whose cgroup grants
16 CPUs that is 2 CPUs per GPU.  This box has ONE GPU, so the entries share it; to see the host and not the GPU, the probe runs
small pairs too (5k points: the GPU is mostly idle, the per-pair host work -- two H2D copies, ~35 launches, mailbox polls -- is
the same).  Every case: `entries` device entries (all device 0, host-staged gather) under `taskset` with `cpus` CPUs.
Usage: python scripts/host_scaling_probe.py            (prints a table; ~2 min)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, ctypes as C, numpy as np
from icpslam_amd import _lib, sharding, synth
n_pts, entries, per_entry, steps, repeats = (int(x) for x in sys.argv[1:6])
n = entries * per_entry
base = [synth.make_pair(n_pts, n_pts, seed=1000 + k)[:2] for k in range(min(n, 16))]
pairs = [base[k % len(base)] for k in range(n)]
srcs, tgts = [p[0] for p in pairs], [p[1] for p in pairs]
P = _lib.Params(); _lib.load().icpgpu_default_params(C.byref(P)); P.max_iterations = 10
step = lambda: sharding.align_batch_multi([0] * entries, srcs, tgts, params=P, want_fitness=True, communicator=sharding.COMM_HOST)
for _ in range(8): step()   # (the workers' buffers settle over the first calls: profiles/r05_batch_groups.txt, section 4)
rates = []
for _ in range(repeats):
    t0 = time.perf_counter()
    for _ in range(steps): step()
    rates.append(steps * n / (time.perf_counter() - t0))
rates.sort()
print(json.dumps({"pairs_per_sec": rates[len(rates) // 2], "lo": rates[0], "hi": rates[-1]}))
'''.replace("import sys, time", "import json, sys, time")
rows = []
# round 5: 64 pairs per entry and step, 10 steps per measurement, 3 measurements per cell: median and spread
for n_pts, per_entry, steps in ((50000, 64, 10), (5000, 64, 10)):
    for entries, cpus in ((1, 16), (1, 4), (1, 2), (1, 1), (2, 4), (4, 8), (8, 16), (8, 8)):
        env = dict(os.environ, PYTHONPATH=ROOT)
        cmd = ["taskset", "-c", "0-%d" % (cpus - 1), sys.executable, "-c", CODE, str(n_pts), str(entries), str(per_entry), str(steps), "3"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        if r.returncode != 0:
            rows.append((n_pts, entries, cpus, None, r.stderr.strip()[-200:]))
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append((n_pts, entries, cpus, d["pairs_per_sec"], d["lo"], d["hi"]))
        print(f"{n_pts:6d} points  entries {entries}  cpus {cpus:2d} ({cpus / entries:.1f} per entry): {d['pairs_per_sec']:8.0f} pairs/s total "
              f"(min {d['lo']:.0f}, max {d['hi']:.0f}: spread {100 * (d['hi'] - d['lo']) / d['pairs_per_sec']:.0f} %), {d['pairs_per_sec'] / entries:8.0f} per entry", flush=True)
