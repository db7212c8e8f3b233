"""bench.py's output contract (one JSON line with the driver's keys plus `roofline` and `cpu_baseline`), on a small
workload so that the test stays short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--workload", "50kx50k", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 1e3 * d["config"]["iters_per_step"] / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["scan_pairs_per_sec_e2e"] > 0 and d["gicp"]["scan_pairs_per_sec_e2e"] > 0   # the odometer's per-scan protocol
    assert d["gicp"]["reference_pipeline_scans_per_sec"] > 0          # ... with VoxelGrid 0.2 m in front of it
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # round 6: the figures come from the dense pass (every launch timed), the timed region's own sample and a steady-state loop
    # stand beside them, and the roofline that binds the kernel (vector issue) is at the top level
    assert rf["dense_pass"]["timed_launches"] == 200 and rf["avg_launch_ms"] == rf["dense_pass"]["avg_launch_ms"]
    assert rf["sampled_in_timed_region"]["avg_launch_ms"] > 0 and rf["steady_state"]["iterations_per_sec"] > 0
    assert rf["steady_state"]["seconds"] >= 1.0 and 0 < rf["steady_state"]["kernel_busy_frac"] <= 1.0
    assert rf["binding"]["bound"] == "valu_issue" and 0 < rf["binding"]["frac"] < 1
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0


def test_bench_two_ranks_through_torchrun():
    """The N > 1 path as the driver launches it (torch.distributed.run, one process per rank), with the gloo backend so
    that both ranks can share this box's single GPU: barrier, max-over-ranks timing, aggregate value, one JSON line."""
    env = dict(os.environ, ICPGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--workload", "50kx50k"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    # whole-job aggregate: both ranks' iterations over the slowest rank's time
    assert abs(d["value"] - 2 * 1e3 * d["config"]["iters_per_step"] / d["ms_per_step"]) <= 1e-6 * d["value"]


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it must become two ranks by itself (it re-executes under
    torch.distributed.run on 127.0.0.1) and say so in the JSON line; gloo so that both ranks can share this box's GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ICPGPU_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "50kx50k"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["scaling"] == "weak"
    assert d["iterations_timed"] == 2 * 3 * 10


def test_bench_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_batch50k_workload_small():
    """BASELINE config 4's shape (independent 50k pairs through icpgpu_align_batch + the record gather), 6 pairs per rank."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--workload", "batch50k", "--pairs-per-rank", "6", "--cpu-seconds", "1"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["metric"] == "icp_iterations_per_sec" and "batch50k" in d["config"]["workload"]
    assert d["scan_pairs_per_sec"] > 0 and abs(d["scan_pairs_per_sec"] - 6 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["scan_pairs_per_sec"]
    assert d["cpu_baseline"]["pairs_per_sec"] > 0


def test_bench_multi_entry_times_the_c_multi_gpu_entry():
    """`bench.py --multi-entry --gpus N`: ONE process drives N device entries through icpgpu_align_batch_multi (icp_multi.cpp) --
    the entry INTEGRATION.md recommends to a C++ host -- so that a scaling run can exercise it.  This box has one GPU: the four
    entries share it and the line says so (host-staged communicator)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multi-entry", "--gpus", "4", "--workload", "batch50k",
                        "--pairs-per-rank", "3", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["entry"] == "icpgpu_align_batch_multi" and d["scaling"] == "weak" and d["value"] > 0
    assert d["scan_pairs_per_sec"] > 0 and len(d["devices"]) == 4
    assert "SHARE device 0" in d["config"]["parallelism"] or d["devices"] == [0, 1, 2, 3]
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d, k


def test_brute_force_roofline_is_a_fraction():
    """the bf16-bound brute-force kernel's `frac` is its ISSUED bf16 flops against the bf16 peak (< 1), the 8-flop
    convention sits beside it as `useful_flops_equiv`."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "50kx50k",
                        "--no-cpu-baseline", "--secondary-scans", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    b = d["roofline"]["brute_force_kernel"]
    assert 0.0 < b["frac"] < 1.0 and b["peak"] == 2500.0 and abs(b["frac"] - b["achieved"] / b["peak"]) < 1e-12
    assert b["useful_flops_equiv"]["tflops"] > 0 and "lane_slots_per_candidate" in d["roofline"].get("issue", {"lane_slots_per_candidate": 0})
    assert d["secondary_loop_scans"] == 4 and "STATIC" in d["roofline"]["traffic_source"]
