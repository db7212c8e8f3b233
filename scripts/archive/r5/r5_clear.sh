#!/bin/bash
# round 5: grid_clear_kernel instead of two memsets per count pass: search tests, e2e figures, PMC files on the new source hash
TAG=${1:-r5clear}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_recognition.py tests/test_gpu_map.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of"; ICPGPU_GICP_INNER=quadratic python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of"; done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench.json
python - $TAG <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}/bench.json").read())
g = d["gicp"]
print("value", round(d["value"]), "e2e", round(d["scan_pairs_per_sec_e2e"]), "gicp e2e", round(g["scan_pairs_per_sec_e2e"]), "pipeline", round(g["reference_pipeline_scans_per_sec"]), "quadratic", round(g["quadratic_inner"]["reference_pipeline_scans_per_sec"]), "stale", d["roofline"].get("pmc_stale"))
PY
bash scripts/r5/r5_pmc.sh
