#!/bin/bash
# round 4: bench.py with the GICP device solver on (default) and off
mkdir -p gpurun_out/r4
python bench.py > gpurun_out/r4/bench_device.json 2> gpurun_out/r4/bench_device.err
ICPGPU_GICP_DEVICE=0 python bench.py > gpurun_out/r4/bench_host.json 2> gpurun_out/r4/bench_host.err
python - <<'PY'
import json
for name in ("device", "host"):
    try:
        d = json.loads(open(f"gpurun_out/r4/bench_{name}.json").read().strip().splitlines()[-1])
        g = d.get("gicp", {})
        print(name, "value", d["value"], "ms/step", d["ms_per_step"], "| gicp e2e", g.get("scan_pairs_per_sec_e2e"), "shim", g.get("shim_pipeline_scans_per_sec"), "resident", g.get("reference_pipeline_scans_per_sec"), "| p2p e2e", d.get("scan_pairs_per_sec_e2e"))
    except Exception as e:
        print(name, "failed", e)
PY
