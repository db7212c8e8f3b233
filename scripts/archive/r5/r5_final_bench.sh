#!/bin/bash
# round 5: the default bench line + its kernel stats (the part of r5_final.sh that carries the headline), into gpurun_out/$1
TAG=${1:-r5final5}
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 200 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
for f in $O/prof/*/*kernel_stats.csv; do
  grep -q "nn_quad_kernel<false, true" $f && grep -q nn_brute_bf16_kernel $f && cp $f $O/kernel_stats.csv
  grep -q "nn_quad_kernel<false, true" $f || cp $f $O/kernel_stats_shim_pipeline.csv
done
rm -rf $O/prof
python bench.py --workload batch50k --steps 20 --warmup 3 > $O/bench_batch50k.json 2> $O/bench_batch50k.err; head -c 200 $O/bench_batch50k.json; echo
