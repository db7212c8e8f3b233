#!/bin/bash
# round 4: the bench contract tests + one default bench line
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > gpurun_out/r4/bench_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4/bench_tests.log
tail -5 gpurun_out/r4/bench_tests.log
( time python bench.py ) > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
tail -3 gpurun_out/r4/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_default.json").read().strip().splitlines()[-1])
g = d.get("gicp", {})
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "| kernel ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
      "| gicp e2e", g.get("scan_pairs_per_sec_e2e"), "shim", g.get("shim_pipeline_scans_per_sec"), "resident", g.get("reference_pipeline_scans_per_sec"),
      "| p2p e2e", d.get("scan_pairs_per_sec_e2e"), "| brute frac", d["roofline"]["brute_force_kernel"]["frac"], "| lane slots", d["roofline"]["issue"]["lane_slots_per_candidate"])
PY
