#!/bin/bash
# usage: scripts/prof_bench.sh <tag> [bench.py args...]  -> kernel-trace stats of the bench command under gpurun_out/prof_<tag>
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
echo "prof $tag rc=$?"
