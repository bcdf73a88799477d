#!/bin/bash
# kernel trace, benchmark line and HBM traffic of the scale-2 configuration -> gpurun_out/r02/ (run on the GPU box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_cfg5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg5 -- python3 $R/bench.py --workload cfg5 --steps 2 --no-cpu-baseline > $O/trace_cfg5.log 2>&1
echo "trace cfg5 rc=$?"
cp $(ls $O/trace_cfg5/*/*kernel_stats.csv | head -1) $O/cfg5_kernel_stats.csv
grep '^{' $O/trace_cfg5.log > $O/cfg5_line_under_profiler.json
rm -rf $O/trace_cfg5
timeout -k 10 300 python3 $R/bench.py --workload cfg5 --steps 2 --no-cpu-baseline > $O/cfg5_line.json 2>/dev/null; echo "cfg5 rc=$?"
$R/scripts/pmc_cfg5.sh > /dev/null; echo "pmc rc=$?"
