#!/bin/bash
# usage: scripts/prof_flow.sh <tag> [flow_time args...] -> kernel trace of the TV-L1 flow under gpurun_out/prof_<tag>
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/scripts/flow_time.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
echo "prof $tag rc=$?"
