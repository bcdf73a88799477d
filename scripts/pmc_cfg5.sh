#!/bin/bash
# HBM traffic of the solver kernels at the scale-2 configuration -> gpurun_out/r02/cfg5_traffic.json (run on the GPU box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc5() { # tag, counters...
  local tag=$1; shift
  rm -rf $O/pmc_$tag
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --workload cfg5 --flow synthetic --steps 1 --warmup 0 --iterations 6 --no-cpu-baseline --no-kernel-timing > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pmc5 fetch5 FETCH_SIZE || exit 1
pmc5 write5 WRITE_SIZE || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/cfg5_traffic.json "1920x1080 RGB -> 3840x2160, scale 2, window 17 (the solver launches of the command the passes were run on)" > /dev/null
rm -rf $O/pmc_fetch5 $O/pmc_write5
cat $O/cfg5_traffic.json
