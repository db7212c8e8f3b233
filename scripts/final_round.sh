#!/bin/bash
# Dev tool: everything behind profiles/ and DESIGN.md's tables in one GPU call.  Outputs under gpurun_out/<tag>/.
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bash scripts/gpu_round.sh $TAG > $O/gpu_round.log 2>&1
bash scripts/pmc_round.sh $TAG > $O/pmc_round.log 2>&1
bash scripts/sq_counters.sh $TAG > $O/sq.txt 2>&1
bash scripts/mem_counters.sh $TAG > $O/mem.txt 2>&1
python scripts/configs_timing.py > $O/configs.txt 2>&1
python scripts/reference_pipeline_probe.py 41 > $O/refpipe.txt 2>&1
python scripts/map_timing.py > $O/map.txt 2>&1
python scripts/gicp_timing.py > $O/gicp.txt 2>&1
python scripts/seq_gicp_probe.py 61 > $O/seq.txt 2>&1
python scripts/steady_timing.py 200000x200000 200000x1000000 50000x50000 5000x5000 > $O/steady.txt 2>&1
python scripts/iter_profile.py 200000x200000 > $O/iter.txt 2>&1
python scripts/fitness_probe.py > $O/fitness.txt 2>&1
export TMPDIR=/tmp
cd /tmp
for job in "gicp gicp_timing.py 50000x50000 200000x200000" "mapper map_timing.py 0.5 6" "reference_pipeline reference_pipeline_probe.py 21"; do
  set -- $job
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $R/scripts/$@ > $O/prof_$name.log 2>&1
  find $O/prof_$name -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$name.csv \;
done
echo done
