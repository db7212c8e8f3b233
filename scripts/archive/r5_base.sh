#!/bin/bash
# round 5: baseline of the round-4 build on today's box: default bench line + kernel stats of the same command
mkdir -p gpurun_out/r5base
( time python bench.py ) > gpurun_out/r5base/bench_default.json 2> gpurun_out/r5base/bench_default.err
tail -3 gpurun_out/r5base/bench_default.err
python bench.py --workload batch50k > gpurun_out/r5base/bench_batch50k.json 2> gpurun_out/r5base/bench_batch50k.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5base/bench_default.json").read().strip().splitlines()[-1])
g = d.get("gicp", {})
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "| kernel ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
      "| gicp e2e", g.get("scan_pairs_per_sec_e2e"), "shim", g.get("shim_pipeline_scans_per_sec"), "resident", g.get("reference_pipeline_scans_per_sec"),
      "| p2p e2e", d.get("scan_pairs_per_sec_e2e"))
b = json.loads(open("gpurun_out/r5base/bench_batch50k.json").read().strip().splitlines()[-1])
print("batch50k", round(b["value"]), b["unit"], b["ms_per_step"])
PY
