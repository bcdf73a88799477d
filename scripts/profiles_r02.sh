#!/bin/bash
# Everything the round's profiles/ entries are made from, written under gpurun_out/r02/ (run on the GPU box).
# Kernel traces and counter passes are separate runs (counters never together with any trace domain but the kernel trace).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() { # tag, bench args...
  local tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py "$@" > $O/trace_$tag.log 2>&1
  echo "trace $tag rc=$?"
  cp $(ls $O/trace_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  grep '^{' $O/trace_$tag.log > $O/${tag}_line_under_profiler.json
}
pmc() { # tag, counters...
  local tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --flow synthetic --steps 1 --warmup 0 --iterations 12 --no-cpu-baseline --no-kernel-timing > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
# the default command, unprofiled, then under the kernel trace
timeout -k 10 300 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
trace bench --no-cpu-baseline || exit 1
trace cfg5 --workload cfg5 --steps 2 --no-cpu-baseline || exit 1
trace tvl1 --workload tvl1 --steps 5 --warmup 1 --no-cpu-baseline || exit 1
timeout -k 10 300 python3 $R/bench.py --workload cfg5 --steps 2 --no-cpu-baseline > $O/cfg5_line.json 2>/dev/null; echo "cfg5 rc=$?"
timeout -k 10 300 python3 $R/bench.py --workload tvl1 --steps 10 --warmup 2 > $O/tvl1_line.json 2>/dev/null; echo "tvl1 rc=$?"
pmc sq1 SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES || exit 1
pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY || exit 1
pmc fetch FETCH_SIZE || exit 1
pmc write WRITE_SIZE || exit 1
# HBM traffic of the scale-2 configuration's solver kernels
pmc5() { # tag, counters...
  local tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --workload cfg5 --flow synthetic --steps 1 --warmup 0 --iterations 6 --no-cpu-baseline --no-kernel-timing > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
pmc5 fetch5 FETCH_SIZE || exit 1
pmc5 write5 WRITE_SIZE || exit 1
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/cfg5_traffic.json "1920x1080 RGB -> 3840x2160, scale 2, window 17 (the solver launches of the command the passes were run on)" > /dev/null
python3 $R/scripts/pmc_summary.py $O/pmc_sq1 backproject_update forward_sign > $O/pmc_counters.txt
python3 $R/scripts/pmc_summary.py $O/pmc_sq2 backproject_update forward_sign >> $O/pmc_counters.txt
python3 $R/scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json > /dev/null
cat $O/pmc_counters.txt
rm -rf $O/trace_*/ $O/pmc_*/
ls -la $O
