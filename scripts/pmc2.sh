#!/bin/bash
# usage: scripts/pmc2.sh <tag> <library or default> <counter list...>   (one rocprofv3 --pmc pass over quick_time.py)
set -u
tag=$1; lib=$2; shift 2
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
if [ "$lib" != default ]; then export SRX_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scripts/quick_time.py ${QT_ARGS:---iters 3 --reps 1} > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag backproject_update forward_sign
