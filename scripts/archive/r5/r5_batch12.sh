#!/bin/bash
TAG=${1:-r5batch12}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
export TMPDIR=/tmp; cd /tmp
for c in 4x8x8 4x7x8; do
  PAIRS=512 REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$c -- python $R/scripts/r5/r5_batch_probe.py $c > $R/gpurun_out/$TAG/$c.log 2>&1
  f=$(find $R/gpurun_out/$TAG/$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; head -8 $f | cut -c1-200
  grep -v amdgpu.ids $R/gpurun_out/$TAG/$c.log | tail -1
  cp $f $R/gpurun_out/$TAG/kernel_stats_$c.csv
  find $R/gpurun_out/$TAG/$c -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_trace_$c.csv \;
  rm -rf $R/gpurun_out/$TAG/$c
done
ls -la $R/gpurun_out/$TAG
