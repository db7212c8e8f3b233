#!/bin/bash
# round 5: the search kernel's parity tests + the default bench line (A/B of a kernel change)
TAG=${1:-r5quick}
mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
python bench.py --workload batch50k > gpurun_out/$TAG/bench_batch50k.json 2> gpurun_out/$TAG/bench_batch50k.err
python scripts/iter_profile.py > gpurun_out/$TAG/iter_profile.txt 2>&1
tail -15 gpurun_out/$TAG/iter_profile.txt
python - $TAG <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/{t}/bench_default.json").read().strip().splitlines()[-1])
g = d.get("gicp", {})
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "| kernel ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
      "| gicp e2e", g.get("scan_pairs_per_sec_e2e"), "shim", g.get("shim_pipeline_scans_per_sec"), "resident", g.get("reference_pipeline_scans_per_sec"),
      "| p2p e2e", d.get("scan_pairs_per_sec_e2e"))
b = json.loads(open(f"gpurun_out/{t}/bench_batch50k.json").read().strip().splitlines()[-1])
print("batch50k", round(b["value"]), b["unit"], b["ms_per_step"])
PY
