#!/bin/bash
# usage: scripts/prof_any.sh <tag> <script.py> [args...]  -> kernel-trace stats under gpurun_out/prof_<tag>
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
echo "prof $tag rc=$?"
