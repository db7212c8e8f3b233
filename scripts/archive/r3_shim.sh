#!/bin/bash
# Round 3: the boundary as integrated -- tests, then bench.py with the shim pipeline figure
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3_shim}; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_recognition.py tests/test_cpp_shim.py tests/test_gpu_errors.py tests/test_abi.py -x -q -s 2>&1 | tail -15
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], "e2e", d.get("scan_pairs_per_sec_e2e"))
print(json.dumps(d.get("gicp"), indent=1)[:1500])
PY
