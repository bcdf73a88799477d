#!/bin/bash
# usage: scripts/pmc.sh <tag> <counter list...>   (run on the GPU box; one rocprofv3 --pmc pass)
set -u
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scripts/quick_time.py ${QT_ARGS:---iters 3 --reps 1} > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
