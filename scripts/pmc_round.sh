#!/bin/bash
# Dev tool: HBM traffic of the two NN kernels from PMC counters, one counter per pass, --kernel-trace only
# (MI355X_MICROARCH.md HBM section). Outputs under gpurun_out/<tag>/pmc/, summary gpurun_out/<tag>/pmc_traffic.json.
TAG=${1:-pmc}
SIZE=${2:-200000x200000}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG/pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for mode in grid brute; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${mode}_$ctr -- python $R/scripts/one_align.py $SIZE $mode > $O/${mode}_$ctr.log 2>&1
    echo "$mode $ctr rc=$?"
  done
done
python $R/scripts/pmc_summarize.py $O $SIZE > $R/gpurun_out/$TAG/pmc_traffic.json
cat $R/gpurun_out/$TAG/pmc_traffic.json
