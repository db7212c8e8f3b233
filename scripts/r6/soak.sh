#!/bin/bash
# round 6: soak of the staged result clouds -- the callback harness for 20 000 scans (four rotating threads, then one): every scan of the
# alternating pair must give one of two results (after the first), no wait may time out
TAG=${1:-r6soak}
O=gpurun_out/$TAG; mkdir -p $O /tmp/shim
g++ -std=c++14 -O2 -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for th in 4 1; do
  for inner in exact quadratic; do
    ICPGPU_GICP_INNER=$inner timeout 600 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 10004 0.2 10 $th 4 > /tmp/shim/out_$th$inner.txt 2> /tmp/shim/err_$th$inner.txt; rc=$?
    python - <<PY
lines=[l.split() for l in open('/tmp/shim/out_$th$inner.txt') if l and l[0].isdigit()]
res={}
for l in lines[2:]:
    res.setdefault(int(l[0])%2, set()).add(' '.join(l[1:]))
t=[l for l in open('/tmp/shim/out_$th$inner.txt') if l.startswith('TIMING')]
print('threads $th inner $inner rc $rc:', len(lines), 'scans registered; distinct results for even / odd scans:', len(res.get(0,())), '/', len(res.get(1,())), '|', t[-1].strip() if t else 'no TIMING line', '| stderr bytes', len(open('/tmp/shim/err_$th$inner.txt').read()))
PY
  done
done > $O/soak.txt 2>&1
cat $O/soak.txt
