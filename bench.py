#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ICP hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, through the C-ABI:
  * pair workloads (default 200kx200k = BASELINE.json's metric configuration, SURVEY.md 8(d) headline pair, seed 4 + rank):
    one scan-pair registration of `--iters` forced point-to-point ICP iterations (NN correspondence search + rejection +
    covariance reduction + host SVD each) on clouds already resident in HBM; 10 iterations per step = ICP_MAX_ITERS of the
    reference's odometer (/root/reference/include/icpslam/icp_odometer.h:65);
  * batch50k (BASELINE config 4): this rank's share of 512 independent 50k-point scan pairs (64 per rank at 8 ranks)
    through icpgpu_align_batch (<= 10 iterations + getFitnessScore each, host buffers in), then the result gather
    (icpslam_amd.sharding.gather_records: one all_gather of 184-byte records, RCCL on GPUs).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); independent scan pairs shard across ranks with no
data-path collective (weak scaling: fixed work per GPU); the only communication is the result gather at the end of the
timed region.  value = iterations executed by all ranks / max-over-ranks time.  `python bench.py --gpus N` without a
launcher re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (32x32x16: 32768 flop per 32 cycles and SIMD)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector peak (packed FMA) == FP32 (f32-in) MFMA dense peak
N_SIMDS = 256 * 4          # 256 CUs x 4 SIMD-32 (MI355X_MICROARCH.md "Wave scheduling")
CLOCK_GHZ = 2.4            # MI355X_MICROARCH.md: peak engine clock
# A wave64 VALU instruction issues over 2 cycles on its SIMD-32 (`v_fma_f32` (wave64): 2 cyc, the guide's per-instruction
# table): the chip issues at most N_SIMDS * clock / 2 = 1228.8 G wave-instructions/s.  (Rounds 1-2 modelled SIMD16s at 4
# cycles = 614.4 G/s -- 2x too low: the plain-VALU brute-force kernel alone issues ~890 G/s.  VERDICT round 2.)
VALU_ISSUE_PEAK_GINST = N_SIMDS * CLOCK_GHZ / 2.0
BRUTE_VALU_INSTS_PER_PAIR = 7.0   # nn_brute_kernel<0,4>: 3 sub + 1 mul + 2 fma + 1/8 of (v_min3 fold + compare + selects) ~ 7 per pair per lane

WORKLOADS = {
    # name: (n_src, n_tgt, kind)
    "5kx5k": (5000, 5000, "pair"),
    "50kx50k": (50000, 50000, "pair"),
    "200kx200k": (200000, 200000, "pair"),
    "200kx1M": (200000, 1000000, "submap"),
    "batch50k": (50000, 50000, "batch"),
}
BATCH_TOTAL_PAIRS_AT_8 = 512      # BASELINE config 4: 512 pairs over 8 GPUs = 64 per GPU (weak scaling: 64 per rank)
BATCH_PAIRS_PER_RANK = BATCH_TOTAL_PAIRS_AT_8 // 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 50 (pair workloads) / 10 (batch50k)")
    ap.add_argument("--warmup", type=int, default=None, help="default 10 (pair workloads) / 4 (batch50k)")
    ap.add_argument("--workload", default="200kx200k", choices=sorted(WORKLOADS))
    ap.add_argument("--iters", type=int, default=10, help="ICP iterations per scan pair (forced for the pair workloads)")
    ap.add_argument("--nn", default="auto", choices=["auto", "brute", "grid"],
                    help="correspondence search: auto (grid-accelerated exact NN where it helps), brute (LDS-tiled)")
    ap.add_argument("--pairs-per-rank", type=int, default=BATCH_PAIRS_PER_RANK, help="batch50k: scan pairs per rank and step")
    ap.add_argument("--setup-aligns", type=int, default=20,
                    help="untimed alignments before the W warm-up steps (bring the device to its running state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (e2e rate, GICP, brute force)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    ap.add_argument("--multi-entry", action="store_true",
                    help="batch50k through the C multi-GPU entry (icpgpu_align_batch_multi): ONE process, one host thread per GPU, "
                         "ncclAllGather of the records -- what INTEGRATION.md section 3 recommends to a C++ host; --gpus N = N device entries")
    ap.add_argument("--steady-seconds", type=float, default=1.5, help="length of the steady-state loop behind the timed region")
    ap.add_argument("--secondary-scans", type=int, default=50, help="scans in each secondary (e2e / GICP / pipeline) loop")
    a = ap.parse_args()
    batch = WORKLOADS[a.workload][2] == "batch"
    # (batch50k: the workers' buffers settle over the first few calls -- a call with reallocations takes 18-32 ms instead of 12.5 --
    #  and five timed steps behind two warm-up calls caught one or two of those on some boxes: 3.9-4.2k pairs/s where twenty steps
    #  behind three say 5.1k, round 6)
    if a.steps is None:
        a.steps = 10 if batch else 50
    if a.warmup is None:
        a.warmup = 4 if batch else 10
    return a


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_self_launch(a) -> None:
    """`python bench.py --gpus N` with no launcher around it: become N ranks (one process per GPU) by re-executing under
    torch.distributed.run on 127.0.0.1.  Under a launcher (WORLD_SIZE set) the world must be the one --gpus names."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return
    if a.gpus <= 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def make_workload(name: str, seed: int):
    from icpslam_amd import synth
    n_s, n_t, kind = WORKLOADS[name]
    if kind == "submap":
        src, tgt, _ = synth.make_scan_vs_submap(n_s, n_t, seed)
    else:
        src, tgt, _ = synth.make_pair(n_s, n_t, seed)
    return src, tgt


def _effective_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU box: 256 CPUs, quota 16)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def shim_pipeline(scan_a, scan_b, leaf: float, iters: int, n_scans: int = 24, threads: int = 4) -> dict:
    """The boundary as INTEGRATION.md integrates it, timed: builds tests/cpp/odometer_pipeline_demo.cpp (g++, header-only shim +
    libicpgpu.so) and runs it on the two raw scans.  Also run with ICPGPU_RECOGNISE=0 (every target uploaded and rebuilt: what
    the shim did until round 3) for the difference."""
    import tempfile
    from icpslam_amd import _lib
    out = {}
    try:
        with tempfile.TemporaryDirectory() as td:
            exe = os.path.join(td, "odometer_pipeline_demo")
            libdir = os.path.dirname(_lib.LIB_PATH)
            subprocess.check_call(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"),
                                   os.path.join(ROOT, "tests", "cpp", "odometer_pipeline_demo.cpp"), "-o", exe, "-L", libdir,
                                   "-licpgpu", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
            pa, pb = os.path.join(td, "a.bin"), os.path.join(td, "b.bin")
            scan_a.tofile(pa)
            scan_b.tofile(pb)
            for key, env in (("scans_per_sec", {}), ("scans_per_sec_without_recognition", {"ICPGPU_RECOGNISE": "0"}),
                             ("scans_per_sec_one_thread", {}), ("scans_per_sec_quadratic_inner", {"ICPGPU_GICP_INNER": "quadratic"})):
                th = 1 if key.endswith("one_thread") else threads
                r = subprocess.run([exe, pa, str(scan_a.shape[0]), pb, str(scan_b.shape[0]), str(n_scans), repr(leaf), str(iters),
                                    str(th), "4"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
                if r.returncode != 0:
                    out[key] = None
                    out["error"] = r.stderr.strip()[-300:]
                    continue
                timing = [l for l in r.stdout.splitlines() if l.startswith("TIMING ")][-1]
                out[key] = float(timing.split()[6])
            out["scans"] = n_scans - 4
            out["threads"] = threads
    except Exception as e:  # the headline must not die with a secondary figure
        out["error"] = repr(e)[:300]
    return out


def gicp_cpu_baseline(scan_a, scan_b, leaf: float, iters: int, budget_s: float) -> dict:
    """The reference's per-scan pipeline on the host: VoxelGrid(leaf) of the new scan + GeneralizedIterativeClosestPoint
    (<= iters outer iterations) + getFitnessScore, as the C restatement in PCL's OWN evaluation order (oracle gicp_sums =
    SEQUENTIAL; kd-tree NN, single thread like PCL 1.8), on a bounded sample; then the same on the raw scans (no filter)."""
    import oracle
    oracle.build()
    from concurrent.futures import ThreadPoolExecutor
    p = oracle.default_params(method=oracle.GICP, max_iterations=iters, gicp_sums=oracle.GICP_SUMS_SEQUENTIAL)
    scans = (scan_a, scan_b)
    filt = [oracle.voxel_grid(s, leaf) for s in scans]

    def one_scan(k):
        new = oracle.voxel_grid(scans[k % 2], leaf)              # the filter is part of every scan's cost
        return oracle.icp_align(new, filt[(k + 1) % 2], p, want_fitness=True)["iterations"]
    n, t0 = 0, time.perf_counter()
    while True:
        one_scan(n)
        n += 1
        t_used = time.perf_counter() - t0
        if t_used >= 0.35 * budget_s or n >= 12:
            break
    n_thr = max(1, min(64, _effective_cpus()))
    t1 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:
        list(ex.map(one_scan, range(n_thr)))
    t_all = time.perf_counter() - t1
    t2 = time.perf_counter()
    raw = oracle.icp_align(scans[0], scans[1], p, want_fitness=True)
    t_raw = time.perf_counter() - t2
    return {"value": n / t_used, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": f"{n} scans: VoxelGrid({leaf} m) of a raw {scan_a.shape[0]}-point scan (-> {filt[0].shape[0]} points) + GICP (<= {iters} outer "
                      f"iterations, PCL's sequential sums) + getFitnessScore, {t_used:.1f} s; oracle/gicp_oracle.c + icp_oracle.c "
                      "(restatement, not PCL binaries)",
            "many_cores": {"value": n_thr / t_all, "unit": "scans/s", "cores": n_thr,
                           "sample": f"{n_thr} independent scans at once, one per thread, {t_all:.1f} s"},
            "raw_scans": {"value": 1.0 / t_raw, "unit": "scans/s", "cores": 1, "iterations": raw["iterations"],
                          "sample": f"one GICP registration of the raw {scan_a.shape[0]} x {scan_b.shape[0]} pair (no filter), {t_raw:.1f} s"},
            "usable_cpus": _effective_cpus(), "cpu_model": _cpu_model()}


def cpu_baseline(pairs, iters: int, force: bool, budget_s: float):
    """The oracle (C restatement of PCL's ICP, kd-tree NN, single thread like PCL 1.8) timed on this box's host, on a bounded
    sample of the same workload.  Every align builds its kd-tree, as PCL does for every scan (`icp` is a stack object at
    icp_odometer.cpp:188): the build is part of the reference's per-pair cost."""
    import oracle
    oracle.build()
    p = oracle.default_params(max_iterations=iters, force_iterations=1 if force else 0)
    done, aligns, t_used = 0, 0, 0.0
    t0 = time.perf_counter()
    while True:
        src, tgt = pairs[aligns % len(pairs)]
        r = oracle.icp_align(src, tgt, p)
        done += r["iterations"]
        aligns += 1
        t_used = time.perf_counter() - t0
        if t_used >= budget_s or aligns >= 10:
            break
    # the same work on many cores at once (independent aligns; ctypes releases the GIL): what a CPU-only host could do for
    # the BATCH configs.  A single alignment cannot use them: PCL's ICP is single-threaded.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(64, _effective_cpus()))
    t1 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:
        its = list(ex.map(lambda k: oracle.icp_align(*pairs[k % len(pairs)], p)["iterations"], range(n_thr)))
    t_all = time.perf_counter() - t1
    many = {"value": sum(its) / t_all, "unit": "iterations/s", "cores": n_thr, "pairs_per_sec": n_thr / t_all,
            "sample": f"{n_thr} concurrent aligns, one per thread, {t_all:.1f} s"}
    n_s, n_t = pairs[0][0].shape[0], pairs[0][1].shape[0]
    return {"value": done / t_used, "unit": "iterations/s", "cores": 1, "kind": "port", "pairs_per_sec": aligns / t_used,
            "many_cores": many,
            "sample": f"{aligns} full align(s) of {n_s}x{n_t} pair(s), {done} iterations, {t_used:.1f} s incl. the kd-tree "
                      f"build of every align (PCL rebuilds it per scan); oracle/icp_oracle.c (restatement, not PCL binaries)",
            "host_cpus": os.cpu_count(), "usable_cpus": _effective_cpus(), "cpu_model": _cpu_model()}


def _load_json(name: str) -> dict:
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def search_kernel_source_sha256() -> str:
    """sha256 over the grid search kernel's sources (the files libicpgpu.so's nn_quad_kernel is built from): the committed PMC
    summaries under profiles/ carry the hash of the sources they were collected on (scripts/kernel_hash.py is the same function
    for the collection scripts), and the bench line says `pmc_stale` when they no longer match."""
    import hashlib
    h = hashlib.sha256()
    for f in ("icp_grid.hip", "icp_grid_device.h"):
        with open(os.path.join(ROOT, "icpslam_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def run_multi_entry(a):
    """BASELINE config 4 through icpgpu_align_batch_multi (include/icpgpu.h; icp_multi.cpp): one process, `--gpus` device
    entries, contiguous shards of a.pairs_per_rank pairs each, one all-gather of the 184-byte records (RCCL; host-staged when
    the box has fewer GPUs than entries, i.e. several entries share a device -- stated in the line).  Same JSON contract."""
    import ctypes as C
    import gc
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import torch

    from icpslam_amd import _lib, sharding, synth
    if "WORLD_SIZE" in os.environ:
        raise SystemExit("bench.py --multi-entry is ONE process (no launcher): it drives all GPUs itself")
    n_s, n_t, _ = WORKLOADS["batch50k"]
    n_dev = a.gpus
    have = torch.cuda.device_count()
    real = have >= n_dev
    devices = list(range(n_dev)) if real else [0] * n_dev
    comm = sharding.COMM_RCCL if real else sharding.COMM_HOST
    n_total = a.pairs_per_rank * n_dev
    with ThreadPoolExecutor(max(2, min(16, _effective_cpus()))) as ex:
        pairs = list(ex.map(lambda k: synth.make_pair(n_s, n_t, seed=1000 + k)[:2], range(n_total)))
    srcs, tgts = [p[0] for p in pairs], [p[1] for p in pairs]
    P = _lib.Params()
    _lib.load().icpgpu_default_params(C.byref(P))
    P.max_iterations = a.iters

    def step():
        return sharding.align_batch_multi(devices, srcs, tgts, params=P, want_fitness=True, communicator=comm)
    for _ in range(1 + a.warmup):
        res, recs = step()
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res, recs = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters_all = a.steps * sum(r["iterations"] for r in res)
    assert np.array_equal(recs[:, 0], np.arange(n_total))
    out = {"metric": "icp_iterations_per_sec", "value": iters_all / elapsed, "unit": "iterations/s", "n_gpus": n_dev, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"batch50k: {a.pairs_per_rank} independent 50k x 50k synthetic scan pairs per GPU entry and step (seeds 1000+k; "
                                  f"BASELINE config 4 = 512 pairs over 8 GPUs), <= {a.iters} point-to-point ICP iterations + getFitnessScore each",
                      "n_src": n_s, "n_tgt": n_t,
                      "parallelism": f"ONE process, icpgpu_align_batch_multi over {n_dev} device entr{'y' if n_dev == 1 else 'ies'} "
                                     f"{devices} (one host thread per entry), records gathered by "
                                     + ("ncclAllGather (RCCL)" if real else "the host-staged communicator: this box shows only "
                                        f"{have} GPU(s), the entries SHARE device 0 -- a logic run, not a scaling measurement")},
           "entry": "icpgpu_align_batch_multi", "devices": devices, "scan_pairs_per_sec": a.steps * n_total / elapsed,
           "usable_cpus": _effective_cpus(),
           "roofline": {"kernel": "(batch workload: see the pair workloads for the kernel's roofline)", "bound": "hbm", "achieved": 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}}
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.multi_entry:
        return run_multi_entry(a)
    maybe_self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # one rank per GPU over RCCL.  ICPGPU_BENCH_BACKEND=gloo (tests only) exercises the same multi-process logic with all
    # ranks sharing the visible GPUs, e.g. two ranks on a 1-GPU box.
    backend = os.environ.get("ICPGPU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world == a.gpus

    from icpslam_amd import GICP, GICP_INNER_EXACT, GICP_INNER_QUADRATIC, NN_AUTO, NN_BRUTE, NN_GRID, Context, sharding, synth

    nn_mode = {"auto": NN_AUTO, "brute": NN_BRUTE, "grid": NN_GRID}[a.nn]
    n_s, n_t, kind = WORKLOADS[a.workload]
    batch = kind == "batch"
    ctx = Context(local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if batch:
        # config 4: independent pairs, seeds 1000..; pair k of the whole job lives on rank owner(k)
        n_total = a.pairs_per_rank * world
        mine = sharding.shard_range(n_total, rank, world)
        pairs = [synth.make_pair(n_s, n_t, seed=1000 + k)[:2] for k in mine]
        srcs, tgts = [p[0] for p in pairs], [p[1] for p in pairs]
        ctx.set_params(ctx.default_params(), max_iterations=a.iters, nn_mode=nn_mode)

        def step():
            res = ctx.align_batch(srcs, tgts, want_fitness=True)
            local = np.stack([sharding.make_record(k, r) for k, r in zip(mine, res)])
            sharding.gather_records(local, n_total, rank, world, coll_dev if world > 1 else None)   # the RCCL gather
            return res
        setup_steps = 1
    else:
        src, tgt = make_workload(a.workload, seed=4 + rank)
        pairs = [(src, tgt)]
        ctx.set_params(ctx.default_params(), max_iterations=a.iters, force_iterations=1, nn_mode=nn_mode)
        ctx.set_source(src)      # inputs resident in HBM before the timed region
        ctx.set_target(tgt)

        def gather(res):
            """Result gather, the only inter-GPU traffic of the path: one 184-byte record per pair."""
            local = sharding.make_record(rank, res)[None]
            sharding.gather_records(local, world, rank, world, coll_dev if world > 1 else None)

        def step():
            return ctx.align()
        setup_steps = a.setup_aligns

    # Set-up, before the W warm-up steps: the result gather once (torch's allocator and RCCL's communicator are created
    # lazily: milliseconds to seconds during which the GPU idles) and --setup-aligns alignments that bring the device back
    # to its running state -- measured: with nothing but 3 warm-up steps the first timed steps ran 3 % below the steady rate.
    if not batch:
        gather({"T": np.eye(4, dtype=np.float32), "iterations": 0, "converged": False, "state": 0, "n_corr": 0, "mse": 0.0,
                "fitness": 0.0})
    # the harness's own interpreter must not stop the clock's world: a generation-2 garbage collection of CPython takes ~40 ms
    # here (the synthetic clouds and records are large containers) and used to land in the SIXTH batch of every batch50k run --
    # one 52 ms step among 15 ms ones (scripts/batch_jitter.py with and without gc.disable(): profiles/r02_batch_scheduler.txt).
    # Collected BEFORE the warm-up since round 6: between the warm-up and the clock it left the GPU idle for those 40 ms, and the
    # K timed steps that followed ran 3-8 % below the steady-state loop's rate (a warm-up followed by a pause is not a warm-up).
    import gc
    gc.collect()
    gc.disable()
    for _ in range(setup_steps):
        step()
    for _ in range(a.warmup):
        last = step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    if not batch:
        gather(last)
    barrier()
    elapsed = time.perf_counter() - t0
    # (the collector stays off through the secondary measurements below: they are timed loops too)
    prof = ctx.profile()
    # Right behind the timed region, on rank 0: a DENSE pass for the roofline -- the same alignments with EVERY sweep's kernel
    # bracketed by HIP events (the timed region samples one sweep in 13, because the event records are barrier packets that cost
    # 6-7 us apiece and would slow `value` down: 15 timed launches at the driver's --steps 20).  Reported beside the sampled figure.
    dense = steady = None
    if rank == 0 and not batch and not a.no_extras:
        ctx.profile_sampling(1)
        ctx.profile_reset()
        for _ in range(20):
            ctx.align()
        pd = ctx.profile()
        if pd.grid_timed:
            dense = {"avg_launch_ms": pd.grid_ms / pd.grid_timed, "timed_launches": int(pd.grid_timed), "alignments": 20}
        ctx.profile_sampling(13)
        # ... and a STEADY-STATE loop of the same step for >= 1 s (the driver's --steps 20 is 15 ms of timed region: a busy
        # figure of a few per cent says nothing about the rate a sequence runs at) -- a secondary figure, `value` stays the K steps
        ctx.profile_reset()
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        n_steady = 0
        while time.perf_counter() - t_s < a.steady_seconds:
            for _ in range(50):
                step()
            n_steady += 50
        torch.cuda.synchronize()
        steady_s = time.perf_counter() - t_s
        ps = ctx.profile()
        steady = {"seconds": steady_s, "steps": n_steady, "iterations_per_sec": int(ps.iterations) / steady_s,
                  "ms_per_step": 1e3 * steady_s / max(1, n_steady),
                  "sampled_avg_launch_ms": ps.grid_ms / max(1, ps.grid_timed), "sampled_launches": int(ps.grid_timed),
                  "kernel_busy_frac": (ps.grid_ms / max(1, ps.grid_timed)) * int(ps.grid_launches) / (1e3 * steady_s),
                  "how": "the timed region's step repeated for --steady-seconds behind it, same context, one sweep in 13 timed; "
                         "kernel_busy_frac = the search kernel's sampled mean x its launches / wall"} if n_steady else None
    iters_done = int(prof.iterations)
    pairs_done = int(prof.aligns)
    if world > 1:
        agg = torch.tensor([elapsed, float(iters_done), float(pairs_done)], dtype=torch.float64, device=coll_dev)
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed, iters_all, pairs_all = float(tmax.item()), int(agg[1].item()), int(agg[2].item())
    else:
        iters_all, pairs_all = iters_done, pairs_done

    extras = {}
    if rank == 0 and not a.no_extras and kind == "pair":
        # (1) scan pairs/s as the reference's odometer produces them, SURVEY.md 8(d) protocol: everything
        # laserCloudCallback does per scan inside the timed region (icp_odometer.cpp:188-210) -- setInputSource from a HOST
        # buffer (H2D copy), the per-scan index build, align, getFitnessScore, *prev_cloud_ = *curr_cloud_ -- over
        # alternating scans A, B, A, ... so that every pair has a new source and a new target index.
        def odometry_loop(n_pairs, voxel_leaf=None, **params):
            ctx.set_params(**params)
            clouds = (src, tgt)
            # voxel_leaf: the scan goes through the voxel filter on its way in (icp_odometer.cpp:96-101,177), on the device
            put = (lambda cl: ctx.set_source_voxel_filtered(cl, voxel_leaf)) if voxel_leaf else ctx.set_source
            put(clouds[1])
            ctx.promote_source_to_target()
            for k in range(2):                      # warm-up (buffers of both roles allocated)
                put(clouds[k % 2]); ctx.align(want_fitness=True); ctx.promote_source_to_target()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for k in range(n_pairs):
                put(clouds[k % 2])
                ctx.align(want_fitness=True)
                ctx.promote_source_to_target()
            torch.cuda.synchronize()
            return n_pairs / (time.perf_counter() - t)
        n_e2e = max(4, a.secondary_scans)
        e2e = odometry_loop(n_e2e, max_iterations=a.iters, force_iterations=1)
        # ... and the resident-pair figure of round 1 (align + getFitnessScore, target index reused)
        ctx.set_source(src); ctx.set_target(tgt)
        ctx.align(want_fitness=True)
        t1 = time.perf_counter()
        for _ in range(n_e2e):
            ctx.align(want_fitness=True)
        torch.cuda.synchronize()
        resident = n_e2e / (time.perf_counter() - t1)
        extras["scan_pairs_per_sec_e2e"] = e2e
        extras["scan_pairs_per_sec_resident"] = resident
        extras["secondary_loop_scans"] = n_e2e
        extras["scan_pair_def"] = (
            f"e2e: per scan set_source from a host buffer (H2D) + index build + align({a.iters} forced iterations) + "
            "getFitnessScore + promote_source_to_target, as icp_odometer.cpp:188-210 runs per scan; resident: align + "
            "getFitnessScore on a resident pair, target index reused")
        # (2) the solver the reference literally instantiates (GICP, icp_odometer.cpp:188) on the same raw pair, same loop
        gicp = odometry_loop(n_e2e, method=GICP, max_iterations=a.iters, force_iterations=0)
        # (3) ... and the reference's whole per-scan pipeline: VoxelGrid at icpslam.yaml's 0.2 m in front of it
        leaf = 0.2
        pg0 = ctx.profile()
        pipeline = odometry_loop(n_e2e, voxel_leaf=leaf, method=GICP, max_iterations=a.iters, force_iterations=0)
        # GICP's own measured line: the evaluation server and the covariance kernel against their algorithmic bytes
        class _Diff:  # the pipeline loop's share of the profile
            def __init__(self, a_, b_):
                for f, _t in a_._fields_:
                    setattr(self, f, getattr(a_, f) - getattr(b_, f))
        pg = _Diff(ctx.profile(), pg0)
        # (3b) the same pipeline with GICP's inner minimisation on the QUADRATIC FORM of each outer iteration (opt-in:
        #      icpgpu_params.gicp_inner; one pass over the correspondences per outer iteration instead of ~35, BFGS on the host;
        #      results within tolerance of, not bit-identical to, the default's -- profiles/r05_gicp_quadratic.txt)
        pq0 = ctx.profile()
        pipeline_quadratic = odometry_loop(n_e2e, voxel_leaf=leaf, method=GICP, max_iterations=a.iters, force_iterations=0,
                                           gicp_inner=GICP_INNER_QUADRATIC)
        pq = _Diff(ctx.profile(), pq0)
        ctx.set_params(gicp_inner=GICP_INNER_EXACT)
        gicp_roofline = None
        if pg.gicp_cost_launches and pg.gicp_eval_ms > 0:
            ev_ms = pg.gicp_eval_ms / pg.gicp_cost_launches
            ev_bytes = 88.0 * pg.gicp_eval_corr / pg.gicp_cost_launches
            cov_ms = pg.gicp_cov_ms / max(1, pg.gicp_cov_launches)
            cov_bytes = 64.0 * pg.gicp_cov_points / max(1, pg.gicp_cov_launches)
            gicp_roofline = {
                "gicp_server_kernel": {
                    "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": ev_bytes / (ev_ms * 1e-3) / 1e9,
                    "frac": ev_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_evaluation_ms": ev_ms,
                    "evaluations": int(pg.gicp_cost_launches), "algorithmic_bytes_per_evaluation": ev_bytes,
                    "outer_iterations_solved_on_the_device": int(pg.gicp_device_solves),
                    "note": "one BFGS function/gradient evaluation = 88 B per correspondence (16 B source point, 16 B target point, 48 B "
                            "Mahalanobis matrix, 8 B key); time = host wall, command written -> 13 sums merged (outer iterations the context's measured choice gives to the device solver, gicp_solve_kernel, count with their wall time / their evaluations: profile.gicp_device_solves).  LATENCY-bound: the "
                            "evaluations are dependent host <-> device round trips of ~6.5 us around ~1.4 us of device work (scripts/"
                            "pipeline_breakdown.py with ICPGPU_GICP_TIMING=1); the server keeps its correspondences in registers, so "
                            "the algorithmic bytes are not even re-read"},
                "gicp_cov_kernel": {
                    "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": cov_bytes / (cov_ms * 1e-3) / 1e9 if cov_ms else None,
                    "frac": cov_bytes / (cov_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if cov_ms else None, "avg_cloud_ms": cov_ms,
                    "clouds": int(pg.gicp_cov_launches), "algorithmic_bytes_per_cloud": cov_bytes,
                    "note": "gicp_cov_select_kernel + gicp_cov_far_kernel + gicp_cov_kernel (what is left) + gicp_cov_finish_kernel per cloud "
                            "(HIP events): 16 B read + 48 B written per point; the time is the 20-NN selection over the cloud's own grid (a latency "
                            "chain per point: ~23k waves for 7k slots) and the per-point 3x3 Jacobi SVD, not the bytes"}}
        # (4) the same solver through icpgpu_align_batch: 32 voxel-filtered pairs as resumable runs (GicpRun: one or two host
        #     threads keep eight registrations in flight, every outer iteration's BFGS inside the device solver)
        gicp_batch = None
        try:
            fa, fb = ctx.voxel_grid(src, leaf), ctx.voxel_grid(tgt, leaf)
            bs, bt = [fa, fb] * 16, [fb, fa] * 16
            ctx.set_params(ctx.default_params(), method=GICP, max_iterations=a.iters, force_iterations=0)
            ctx.align_batch(bs, bt, want_fitness=True)
            tb = time.perf_counter()
            for _ in range(3):
                ctx.align_batch(bs, bt, want_fitness=True)
            gicp_batch = 3 * len(bs) / (time.perf_counter() - tb)
        except Exception as e:  # a secondary figure must not take the headline down
            gicp_batch = repr(e)[:200]
        gicp_batch_quadratic = None
        try:
            ctx.set_params(ctx.default_params(), method=GICP, max_iterations=a.iters, force_iterations=0, gicp_inner=GICP_INNER_QUADRATIC)
            ctx.align_batch(bs, bt, want_fitness=True)
            tb = time.perf_counter()
            for _ in range(3):
                ctx.align_batch(bs, bt, want_fitness=True)
            gicp_batch_quadratic = 3 * len(bs) / (time.perf_counter() - tb)
        except Exception as e:
            gicp_batch_quadratic = repr(e)[:200]
        shim = shim_pipeline(src, tgt, leaf, a.iters, n_scans=n_e2e + 4)
        gicp_cpu = None if a.no_cpu_baseline else gicp_cpu_baseline(src, tgt, leaf, a.iters, a.cpu_seconds)
        extras["gicp"] = {"scan_pairs_per_sec_e2e": gicp,
                          "shim_pipeline_scans_per_sec": shim.get("scans_per_sec"),
                          "shim_pipeline_def": "the SAME pipeline as integrated (INTEGRATION.md): tests/cpp/odometer_pipeline_demo.cpp = "
                                               "laserCloudCallback (icp_odometer.cpp:96-101,147-220) with the two swapped type names, a fresh "
                                               "icpgpu::VoxelGrid + icpgpu::GeneralizedIterativeClosestPoint object per scan, host clouds in and "
                                               "out, `*prev_cloud_ = *curr_cloud_`, callbacks rotating over 4 threads (AsyncSpinner(4)); a C++ "
                                               "process of its own, timed inside",
                          "shim_pipeline": shim,
                          "roofline": gicp_roofline,
                          "cpu_baseline": gicp_cpu,
                          "def": f"the same odometry loop with method = GICP (<= {a.iters} outer iterations, BFGS inner "
                                 "solver): what pcl::GeneralizedIterativeClosestPoint at icp_odometer.cpp:188 runs per scan",
                          "batch_pairs_per_sec": gicp_batch,
                          "batch_def": f"32 voxel-filtered ({leaf} m) pairs of the bench scans through icpgpu_align_batch in GICP mode, "
                                       "<= 10 outer iterations + fitness each (resumable runs, icpgpu_gicp.cpp: GicpRun)",
                          "quadratic_inner": {
                              "reference_pipeline_scans_per_sec": pipeline_quadratic,
                              "batch_pairs_per_sec": gicp_batch_quadratic,
                              "outer_iterations_solved": int(pq.gicp_quadratic_solves),
                              "cost_evaluations_on_the_host": int(pq.gicp_cost_launches),
                              "def": "OPT-IN mode icpgpu_params.gicp_inner = QUADRATIC (ICPGPU_GICP_INNER=quadratic): the same pipeline / "
                                     "batch with ONE device pass per outer iteration (gicp_quadratic_kernel: the 73 coefficient sums of "
                                     "the cost as a quadratic form in T's entries, Mahalanobis matrices on the way) and BFGS on the host "
                                     "without a device round trip; within the BASELINE tolerance of the default's result on most pairs, "
                                     "not bit-identical (it leaves out PCL's float32 rounding of the transformed points): "
                                     "profiles/r05_gicp_quadratic.txt, tests/test_gpu_gicp_quadratic.py"},
                          "reference_pipeline_scans_per_sec": pipeline,
                          "reference_pipeline_def": f"per scan: VoxelGrid({leaf} m) of the raw {n_s}-point scan on the device "
                                                    f"(-> {ctx.n_target} points), then the GICP loop above on the filtered clouds: "
                                                    "laserCloudCallback's registration work (icp_odometer.cpp:96-101,177-210)"}
        ctx.set_params(ctx.default_params(), max_iterations=a.iters, force_iterations=1, nn_mode=nn_mode)
        ctx.set_source(src); ctx.set_target(tgt)

    # the brute-force search (north_star's design) measured in the same process for its own roofline line: the matrix-core
    # kernel (default for large clouds) and the plain-VALU kernel
    brute = None
    if rank == 0 and not a.no_extras and not batch:
        ctx.profile_sampling(1)                    # a handful of launches: time every one of them
        brute = {}
        for name, variant in (("mfma", 0), ("mfma_f32", 2), ("valu", 1)):
            ctx.set_params(nn_mode=NN_BRUTE, max_iterations=a.iters, brute_variant=variant)   # (forced: a whole alignment's sweeps)
            ctx.align()
            ctx.profile_reset()
            ctx.align()
            pb = ctx.profile()
            brute[name] = {"avg_launch_ms": pb.nn_ms / max(1, pb.nn_timed), "launches": int(pb.nn_launches)}
        ctx.profile_sampling(13)
        ctx.set_params(nn_mode=nn_mode, max_iterations=a.iters, brute_variant=0)

    if rank == 0:
        flops = 8.0 * n_s * n_t                      # F_iter = 8 * N_s * N_t (brute force)
        alg_bytes_keys = 16.0 * (n_s + n_t) + 8.0 * n_s   # both clouds once + 8 B key per source point
        alg_bytes_fused = 16.0 * (n_s + n_t) + 64.0       # fused design lower bound
        used_grid = prof.grid_launches > 0
        traffic_all, issue_all = _load_json("pmc_traffic.json"), _load_json("pmc_issue.json")
        traffic = traffic_all.get(a.workload, {})
        issue_pmc = issue_all.get(a.workload, {})
        try:
            src_hash = search_kernel_source_sha256()
        except OSError:
            src_hash = None
        traffic_stale = traffic_all.get("kernel_source_sha256") != src_hash
        issue_stale = issue_all.get("kernel_source_sha256") != src_hash
        brute_roofline = None
        if brute:
            def brute_entry(kernel, ms, note):
                tf = flops / (ms * 1e-3) / 1e12
                return {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / FP32_PEAK_TFLOPS, "avg_launch_ms": ms, "launches_timed": "every sweep of one whole alignment",
                        "hbm": {"achieved": alg_bytes_keys / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg_bytes_keys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": alg_bytes_keys}, "note": note}
            small = min(n_s, n_t) < 8192
            brute_roofline = brute_entry(
                "nn_brute_kernel (plain VALU; below 8192 points the matrix-core kernels are not used)" if small else
                "nn_brute_bf16_kernel (LDS-tiled brute force: certified lower bound of every distance from ONE bf16 MFMA "
                "(K = 16, split operands) per 32 x 32 pairs + exact re-check of the few pairs it cannot settle)",
                brute["mfma"]["avg_launch_ms"],
                "`achieved` / `frac` = the bf16 flops the kernel ISSUES on the matrix path (32 per pair: one 32x32x16 MFMA per 1024 pairs) "
                "against the 2.5 PFLOP/s dense bf16 peak; `useful_flops_equiv` keeps SURVEY.md 8(d)'s 8-flop-per-pair convention for "
                "comparison with the f32 kernels; the launches timed are the sweeps of one whole alignment: the first unseeded, the "
                f"other {a.iters - 1} seeded with the previous sweep's neighbours")
            if not small:
                ms_b = brute["mfma"]["avg_launch_ms"]
                issued = 32.0 * n_s * n_t / (ms_b * 1e-3) / 1e12       # one 32x32x16 MFMA (32768 flop) per 1024 pairs
                # the roofline fraction of THIS kernel is the issued-flop fraction of the bf16 peak (pricing the
                # 8-flop convention against the f32 peak gave 1.5 -- not a fraction); the convention's figure moves aside
                brute_roofline["useful_flops_equiv"] = {"tflops": brute_roofline["achieved"], "of_fp32_peak": brute_roofline["frac"],
                                                        "note": "8 * Ns * Nt per launch (SURVEY.md 8(d)) / time: what a 6-instruction-per-pair "
                                                                "f32 kernel would have to sustain; NOT a roofline fraction of this kernel"}
                brute_roofline.update(achieved=issued, peak=BF16_PEAK_TFLOPS, unit="TFLOP/s (bf16, issued)", frac=issued / BF16_PEAK_TFLOPS)
                brute_roofline["matrix_path"] = {
                    "bound": "mfma", "achieved": issued, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s (bf16, issued)",
                    "frac": issued / BF16_PEAK_TFLOPS,
                    "note": "one v_mfma_f32_32x32x16_bf16 per 1024 pairs = 32 issued flop per pair (14 of the 16 K slots used); "
                            "the kernel is bound by the vector folds of the MFMA's outputs (16 v_min3_f32 + 2 compares per 2048 "
                            "pairs and wave: profiles/r03_brute_force_bf16.txt, r03_mfma_coissue.txt), not by the matrix pipe"}
                brute_roofline["f32_mfma_kernel"] = brute_entry(
                    "nn_brute_mfma_kernel (the same search with the bound in two f32 MFMAs per 32 x 32 pairs: round 2's kernel, "
                    "brute_variant 2)", brute["mfma_f32"]["avg_launch_ms"],
                    "f32 MFMAs and vector instructions of one SIMD do not overlap on gfx950 (profiles/r03_mfma_coissue.txt): "
                    "128 MFMA cycles + ~52 vector cycles per 1024 pairs cap this kernel at ~71 % of the f32 peak")
            brute_roofline["traffic"] = traffic.get("nn_brute_hbm_bytes_per_launch")
            brute_roofline["valu_kernel"] = brute_entry(
                "nn_brute_kernel<0,4> (plain VALU, 7 instructions per pair)", brute["valu"]["avg_launch_ms"],
                "the round-1 kernel: unpacked f32 VALU work (6 flop-carrying + ~1 fold/compare instruction per pair); its "
                "binding roofline is instruction issue -- see `issue` beside it")
            v_ginst = BRUTE_VALU_INSTS_PER_PAIR * n_s * n_t / 64.0 / (brute["valu"]["avg_launch_ms"] * 1e-3) / 1e9
            brute_roofline["valu_kernel"]["issue"] = {
                "bound": "valu_issue", "achieved": v_ginst, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instr/s",
                "frac": v_ginst / VALU_ISSUE_PEAK_GINST,
                "source": f"{BRUTE_VALU_INSTS_PER_PAIR:g} VALU instructions per pair and lane (ISA of the inner loop) x Ns x Nt / 64 "
                          "lanes / the live launch time; same peak as the grid kernel's `issue`"}
        if used_grid and prof.grid_timed and not batch:
            s_ms = prof.grid_ms / max(1, prof.grid_timed)   # the timed region's sampled launches
            # `avg_launch_ms` / `achieved` / `frac` come from the DENSE pass (every launch of 20 alignments between HIP events: 200
            # launches) when it ran; the timed region's own sample (15 launches at the driver's --steps 20) stays beside it
            g_ms = dense["avg_launch_ms"] if dense else s_ms
            gbs = alg_bytes_fused / (g_ms * 1e-3) / 1e9
            roofline = {
                "kernel": "nn_quad_kernel<fused> (uniform-grid exact NN, four points per wave pass, + rejection + 17-term reduction)",
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": traffic.get("nn_grid_hbm_bytes_per_launch"),
                "traffic_source": "STATIC: profiles/pmc_traffic.json (builder-run rocprofv3 --pmc passes, committed), not measured in this run; "
                                  "pmc_stale = the file was collected on other sources of the search kernel than the ones in this tree "
                                  "(sha256 over icp_grid.hip + icp_grid_device.h, embedded by the collection scripts)",
                "pmc_stale": bool(traffic_stale),
                "kernel_source_sha256": src_hash,
                "avg_launch_ms": g_ms, "launches": int(prof.grid_launches),
                "timed_launches": int(dense["timed_launches"]) if dense else int(prof.grid_timed),
                "timed_launches_how": ("DENSE pass: 20 more alignments right behind the timed region with EVERY sweep's launch between HIP "
                                       "events on the context's stream; avg_launch_ms = their mean" if dense else
                                       "HIP-event triples on the context's stream around one sweep in 13 of the timed region"),
                "algorithmic_bytes_per_launch": alg_bytes_fused,
                "sampled_in_timed_region": {
                    "avg_launch_ms": s_ms, "timed_launches": int(prof.grid_timed),
                    "achieved": alg_bytes_fused / (s_ms * 1e-3) / 1e9, "frac": alg_bytes_fused / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "how": "one sweep in 13 of the K timed steps (13 is coprime with the 10 sweeps of an alignment, so every sweep position "
                           "-- the cold first one to the converged tenth -- is sampled equally often; event records are barrier packets "
                           "of 6-7 us, so the timed region cannot time every launch)"},
                "dense_pass": dense,
                "steady_state": steady,
                "note": "dominant kernel of the default (AUTO) path; algorithmic bytes = both clouds once + 64 B of sums "
                        "(SURVEY.md 8(d) fused lower bound).  An exact NN search does not stream; `issue` gives its VALU "
                        "issue rate against the SIMD-32 peak (it sits well below it: the kernel is latency / dependency "
                        "bound, DESIGN.md section 5)"}
            if issue_pmc:
                # Issue roofline: wave-level VALU instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU, mean per launch
                # over the sweeps of an alignment, profiles/pmc_issue.json) / the live launch time, against what 1024
                # SIMD-32s can issue (one wave64 VALU instruction per 2 cycles each).
                valu = float(issue_pmc["valu_insts_per_launch"])
                ginst = valu / (g_ms * 1e-3) / 1e9
                cand = float(issue_pmc.get("candidates_per_launch", 0.0))
                # ... and what the kernel's own instruction MIX can issue: 41 % of its vector instructions are full rate (2.35
                # cycles measured), 59 % half rate (4.2: every compare, select, DPP move, lane read, min/max, shift, convert) --
                # scripts/probes/valu_rate.cpp, profiles/r03_valu_issue_and_pmc.txt
                mix_cycles = 0.41 * 2.35 + 0.59 * 4.2
                mix_peak = N_SIMDS * CLOCK_GHZ / mix_cycles
                act, gui = issue_pmc.get("valu_active_cycles_x4"), issue_pmc.get("grbm_gui_active_sum_over_xcds")
                roofline["issue"] = {
                    "bound": "valu_issue", "achieved": ginst, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instr/s",
                    "frac": ginst / VALU_ISSUE_PEAK_GINST,
                    "peak_for_this_instruction_mix": mix_peak, "frac_of_mix_peak": ginst / mix_peak,
                    "valu_pipe_busy_frac_pmc": (4.0 * act / (N_SIMDS * gui / 8.0)) if act and gui else None,
                    "note": "peak = every SIMD-32 issuing a full-rate wave64 instruction every 2 cycles (v_fma_f32); this kernel's "
                            "static mix averages 3.4 cycles per instruction (measured class rates), and SQ_ACTIVE_INST_VALU x 4 / "
                            "(SIMDs x GRBM_GUI_ACTIVE per XCD) of the same PMC pass says how busy the vector pipes were",
                    "valu_insts_per_launch": valu,
                    "valu_insts_per_source_point": valu / n_s,
                    "salu_insts_per_launch": issue_pmc.get("salu_insts_per_launch"),
                    "candidates_per_launch": cand or None,
                    "lane_slots_per_candidate": (valu * 64.0 / cand) if cand else None,
                    "lane_slots_def": "VALU wave-instructions x 64 lanes / target points evaluated: ~10 would be the distance + compare alone",
                    "pmc_source": "STATIC: profiles/pmc_issue.json (builder-run rocprofv3 --pmc, committed); only the launch time is live",
                    "pmc_stale": bool(issue_stale),
                    "useful_tflops": (8.0 * cand / (g_ms * 1e-3) / 1e12) if cand else None,
                    "useful_flop_frac_of_fp32_peak": (8.0 * cand / (g_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS) if cand else None,
                    "source": issue_pmc.get("_how", "profiles/pmc_issue.json")}
                # the roofline that BINDS this kernel, beside `bound: hbm` (the one the metric names)
                roofline["binding"] = {"bound": "valu_issue", "achieved": ginst, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instr/s",
                                       "frac": ginst / VALU_ISSUE_PEAK_GINST, "frac_of_mix_peak": ginst / mix_peak,
                                       "why": "an exact NN search walks an index: 152 vector instructions per source point against 32 "
                                              "algorithmic bytes; details under `issue`"}
            if brute_roofline:
                roofline["brute_force_kernel"] = brute_roofline
        elif batch:
            roofline = {"kernel": "(batch workload: eight concurrent contexts, kernels overlap; see the pair workloads)",
                        "bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
        else:
            nn_ms = prof.nn_ms / max(1, prof.nn_timed)
            tf = flops / (nn_ms * 1e-3) / 1e12
            roofline = dict(brute_roofline or {}, kernel=(brute_roofline or {}).get("kernel", "nn_brute_kernel (LDS-tiled brute force)"), bound="mfma", achieved=tf,
                            peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_PEAK_TFLOPS, avg_launch_ms=nn_ms,
                            launches=int(prof.nn_launches), traffic=traffic.get("nn_brute_hbm_bytes_per_launch"))
        if batch and used_grid and prof.grid_timed:
            g_ms = prof.grid_ms / max(1, prof.grid_timed)
            gbs = alg_bytes_fused / (g_ms * 1e-3) / 1e9
            roofline = {"kernel": "nn_quad_batch_kernel (nn_quad_kernel's body, one pair per blockIdx.y): a lock-step sweep's HIP-event "
                                  "time / its pairs, at 50k x 50k", "bound": "hbm",
                        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_ms": g_ms, "launches": int(prof.grid_launches), "timed_launches": int(prof.grid_timed),
                        "algorithmic_bytes_per_launch": alg_bytes_fused}
        if batch:
            workload = (f"batch50k: {a.pairs_per_rank} independent 50k x 50k synthetic scan pairs per GPU and step (seeds 1000+k; "
                        f"BASELINE config 4 = 512 pairs over 8 GPUs), <= {a.iters} point-to-point ICP iterations + getFitnessScore "
                        "each through icpgpu_align_batch (host buffers in), then the result gather")
        else:
            workload = (f"{a.workload} synthetic Velodyne-shaped scan pair per GPU (seed 4+rank), "
                        f"{a.iters} forced point-to-point ICP iterations per step, clouds resident in HBM")
        out = {
            "metric": "icp_iterations_per_sec",
            "value": iters_all / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "n_src": n_s, "n_tgt": n_t,
                       "iters_per_step": iters_all / max(1, world * a.steps),
                       "nn": ("uniform-grid exact NN (fused reduce)" if used_grid else "brute-force LDS-tiled") + f" [--nn {a.nn}]",
                       "parallelism": f"{world} rank(s), one per GPU, independent scan pairs; result all_gather only",
                       "setup_aligns": setup_steps},
            "world_size": dist.get_world_size() if world > 1 else 1,
            "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None,
            "scan_pairs_per_sec": pairs_all / elapsed,
            "scan_pairs_def": "alignments finished by all ranks / the timed region"
                              + ("" if batch else " (resident pair, no fitness sweep: see scan_pairs_per_sec_e2e for the odometer's per-scan protocol)"),
            "iterations_timed": iters_all,
            "roofline": roofline,
            "reduce_kernel": {"avg_ms": prof.reduce_ms / max(1, prof.reduce_timed)},
        }
        out.update(extras)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pairs[: 4], a.iters, not batch, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
