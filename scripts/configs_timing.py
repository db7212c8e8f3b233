"""Dev tool: the five BASELINE.json configs on ONE MI355X (configs 4 and 5: one GPU's share of the 8-GPU job), with the
CPU oracle beside them where it finishes in seconds.  Output is committed as profiles/r01_configs.txt."""
import gc, os, sys, time, tempfile
if not os.environ.get('KEEP_GC'): gc.disable()   # a gen-2 collection of CPython (~40 ms) is not the library's jitter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from icpslam_amd import Context, sequence, synth

def t_ms(f, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) * 1e3 / reps, r

with Context(0) as ctx:
    # C1: 5k pair, the reference's CPU-runnable case
    src, tgt, _ = synth.make_pair(5000, 5000, seed=1)
    ctx.set_params(ctx.default_params()); ctx.set_source(src); ctx.set_target(tgt); ctx.align()
    g, r = t_ms(lambda: ctx.align(want_fitness=True), 20)
    c, o = t_ms(lambda: oracle.icp_align(src, tgt, oracle.default_params(), want_fitness=True), 3)
    print(f"C1  5k x 5k, <=10 it: GPU {g:.3f} ms ({r['iterations']} it) | CPU oracle 1 core {c:.1f} ms ({o['iterations']} it)", flush=True)
    # C2: 50k pair, exactly 30 iterations
    src, tgt, _ = synth.make_pair(50000, 50000, seed=2)
    ctx.set_params(ctx.default_params(), max_iterations=30, force_iterations=1, transformation_epsilon=0.0)
    ctx.set_source(src); ctx.set_target(tgt); ctx.align()
    g, r = t_ms(lambda: ctx.align(), 20)
    p = oracle.default_params(max_iterations=30, force_iterations=1, transformation_epsilon=0.0)
    c, o = t_ms(lambda: oracle.icp_align(src, tgt, p), 1)
    print(f"C2  50k x 50k, 30 forced it: GPU {g:.3f} ms = {30e3/g:.0f} it/s | CPU oracle {c:.0f} ms = {30e3/c:.1f} it/s", flush=True)
    # C3: 200k scan vs 1M submap, <= 30 iterations (mapper constants)
    src, tgt, _ = synth.make_scan_vs_submap(200000, 1000000, seed=3)
    ctx.set_params(ctx.default_params(), max_iterations=30)
    ctx.set_source(src)
    tb, _ = t_ms(lambda: ctx.set_target(tgt))
    t1, r = t_ms(lambda: ctx.align())                       # includes the grid builds
    g, r = t_ms(lambda: ctx.align(), 10)
    c, o = t_ms(lambda: oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30)), 1)
    print(f"C3  200k x 1M, <=30 it: GPU {g:.3f} ms ({r['iterations']} it; first align incl. index builds {t1:.2f} ms, H2D of the map {tb:.2f} ms) | "
          f"CPU oracle {c:.0f} ms ({o['iterations']} it, incl. kd-tree build)", flush=True)
    # C4: one GPU's share (64 pairs) of the 512 x 50k batch
    pairs = [synth.make_pair(50000, 50000, seed=1000 + k)[:2] for k in range(64)]
    ctx.set_params(ctx.default_params(), max_iterations=10)
    ctx.align_batch([p[0] for p in pairs[:8]], [p[1] for p in pairs[:8]])
    reps = []
    for _ in range(7):   # a noisy figure (8 host threads): median of 7, and the slowest (usually the first batch) beside it
        t, res = t_ms(lambda: ctx.align_batch([p[0] for p in pairs], [p[1] for p in pairs], want_fitness=True))
        reps.append(t)
    g, g_worst = sorted(reps)[3], max(reps)
    its = sum(r["iterations"] for r in res)
    c, _ = t_ms(lambda: [oracle.icp_align(p[0], p[1], oracle.default_params(), want_fitness=True) for p in pairs[:4]])
    later = reps[1:]  # (the first 64-pair batch sizes every worker's buffers: the warm-up above ran 8 pairs only)
    print(f"C4  64 pairs of 50k (1/8 of 512), <=10 it + fitness, host buffers in: GPU median of 7 batches {g:.1f} ms = {64e3/g:.0f} pairs/s, "
          f"{its*1e3/g:.0f} it/s (slowest batch {g_worst:.1f} ms = {64e3/g_worst:.0f} pairs/s; the batches in order: "
          f"{' '.join('%.1f' % t for t in reps)} ms -- after the first: slowest / median = {max(later)/sorted(later)[len(later)//2]:.2f}) | "
          f"CPU oracle {c/4:.0f} ms per pair = {4e3/c:.2f} pairs/s on 1 core", flush=True)
    # C5: one GPU's share (250 consecutive pairs) of the 2000-scan sequence, 50k points per scan
    n_scans = int(os.environ.get("C5_SCANS", "251"))
    rng = np.random.default_rng(5)
    scene = synth.make_scene(5, extent=120.0)
    poses = [np.eye(4)]
    for _ in range(n_scans - 1):
        poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
    tg0 = time.perf_counter()
    scans = [synth.scan(scene, P, 50000, seed=5000 + k) for k, P in enumerate(poses)]
    tgen = time.perf_counter() - tg0
    ctx.set_params(ctx.default_params())
    g, (graph, recs) = t_ms(lambda: sequence.run_odometry_batched(ctx, scans))
    acc = sum(r["accepted"] for r in recs)
    with tempfile.TemporaryDirectory() as d:
        tw, _ = t_ms(lambda: graph.write_g2o(os.path.join(d, "graph.g2o")))
    g2, (graph2, recs2) = t_ms(lambda: sequence.run_odometry(ctx, scans))
    end_err = np.linalg.norm(np.array(graph.pose(graph.num_poses - 1)[0]) - poses[-1][:3, 3])
    print(f"C5  {n_scans} scans of 50k (1/8 of 2000): batched {g:.1f} ms = {(n_scans-1)*1e3/g:.0f} pairs/s, online loop {g2:.1f} ms = "
          f"{(n_scans-1)*1e3/g2:.0f} pairs/s; accepted {acc}/{n_scans-1}, keyframes {graph.num_keyframes}, g2o export {tw:.2f} ms, "
          f"end point {end_err:.1f} m from ground truth after {0.25*(n_scans-1):.0f} m (point-to-point ICP under-estimates along-track motion on "
          f"this synthetic street, on the CPU oracle just the same: tests/test_gpu_sequence.py); synthetic scan generation {tgen:.0f} s", flush=True)
