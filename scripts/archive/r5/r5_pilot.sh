#!/bin/bash
# round 5: pilot sweep A/B (ICPGPU_PILOT = shift; 0 = off), dev flavour, per-sweep kernel time + parity of the search.
# (the kernel variant is NOT kept -- EXPERIMENTS.md, round 5 table: it made the cold sweep slower at every shift; this script is the record of how it was measured)
TAG=${1:-r5pilot}
mkdir -p gpurun_out/$TAG
export ICPGPU_FLAVOUR=dev
for p in 0 3 2 4 5; do
  echo "## ICPGPU_PILOT=$p" >> gpurun_out/$TAG/iter.txt
  ICPGPU_PILOT=$p python scripts/iter_profile.py 200000x200000 >> gpurun_out/$TAG/iter.txt 2>&1
  ICPGPU_PILOT=$p python scripts/iter_profile.py 50000x50000 >> gpurun_out/$TAG/iter.txt 2>&1
  ICPGPU_PILOT=$p python scripts/iter_profile.py 200000x1000000 >> gpurun_out/$TAG/iter.txt 2>&1
done
echo "## ICPGPU_PILOT=3 ICPGPU_CUBE_START=0" >> gpurun_out/$TAG/iter.txt
ICPGPU_PILOT=3 ICPGPU_CUBE_START=0 python scripts/iter_profile.py 200000x200000 >> gpurun_out/$TAG/iter.txt 2>&1
cat gpurun_out/$TAG/iter.txt
unset ICPGPU_FLAVOUR
timeout 1500 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
python - $TAG <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/{t}/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "| kernel ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "| p2p e2e", d.get("scan_pairs_per_sec_e2e"))
PY
