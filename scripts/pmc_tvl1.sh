#!/bin/bash
# HBM traffic of the TV-L1 stage per 1920x1080 pair -> gpurun_out/r02/tvl1_traffic.json (run on the GPU box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PAIRS=4
pmct() { # tag, counters...
  local tag=$1; shift
  rm -rf $O/pmc_$tag
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --workload tvl1 --pairs-in-flight 1 --pairs-per-step 1 --steps $PAIRS --warmup 0 --no-cpu-baseline > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pmct fetcht FETCH_SIZE || exit 1
pmct writet WRITE_SIZE || exit 1
python3 $R/scripts/pmc_tvl1.py $O/pmc_fetcht $O/pmc_writet $PAIRS $O/tvl1_traffic.json
rm -rf $O/pmc_fetcht $O/pmc_writet
