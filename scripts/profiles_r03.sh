#!/bin/bash
# Everything the round's profiles/ entries are made from, written under gpurun_out/r03/ (run on the GPU box).
# Kernel traces and counter passes are separate runs (counters never together with any trace domain but the kernel
# trace); kernel statistics are reduced to the library's own kernels (scripts/filter_kernel_stats.py: the traced
# process also synthesises its clip with torch).
# usage: scripts/profiles_r03.sh [part ...]   parts: lines traces pmc cfg5 tvl1 (default: all)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
parts=${*:-lines traces pmc cfg5 tvl1}
has() { [[ " $parts " == *" $1 "* ]]; }
line() { # tag, bench args...
  local tag=$1; shift
  timeout -k 10 300 python3 $R/bench.py "$@" > $O/${tag}_line.json 2> $O/${tag}_line.err; echo "line $tag rc=$?"
}
trace() { # tag, bench args...
  local tag=$1; shift
  rm -rf $O/trace_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py "$@" > $O/trace_$tag.log 2>&1
  echo "trace $tag rc=$?"
  python3 $R/scripts/filter_kernel_stats.py "$(ls $O/trace_$tag/*/*kernel_stats.csv | head -1)" $O/${tag}_kernel_stats.csv
  grep '^{' $O/trace_$tag.log > $O/${tag}_line_under_profiler.json
  rm -rf $O/trace_$tag
}
pmc() { # tag, workload args, counters...
  local tag=$1 wl=$2; shift 2
  rm -rf $O/pmc_$tag
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py $wl --flow synthetic --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
}
if has lines; then
  line bench || exit 1                                   # the default command: headline, configs[3]
  line cfg1 --workload cfg1 --steps 10 --warmup 2 || exit 1
  line cfg2 --workload cfg2 --steps 10 --warmup 2 || exit 1
  line cfg5 --workload cfg5 --steps 3 --warmup 1 || exit 1
  line tvl1 --workload tvl1 --steps 10 --warmup 2 || exit 1
  line bench_synthetic_flow --flow synthetic --no-cpu-baseline --steps 6 --warmup 2 || exit 1
fi
if has traces; then
  trace bench --no-cpu-baseline || exit 1
  trace cfg1 --workload cfg1 --steps 6 --warmup 1 --no-cpu-baseline || exit 1
  trace cfg2 --workload cfg2 --steps 6 --warmup 1 --no-cpu-baseline || exit 1
  trace cfg5 --workload cfg5 --steps 2 --no-cpu-baseline || exit 1
  trace tvl1 --workload tvl1 --steps 5 --warmup 1 --no-cpu-baseline || exit 1
fi
if has pmc; then
  W="--iterations 12"
  pmc sq1 "$W" SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES || exit 1
  pmc sq2 "$W" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY || exit 1
  pmc fetch "$W" FETCH_SIZE || exit 1
  pmc write "$W" WRITE_SIZE || exit 1
  python3 $R/scripts/pmc_summary.py $O/pmc_sq1 backproject_update forward_sign build_maps > $O/pmc_counters.txt
  python3 $R/scripts/pmc_summary.py $O/pmc_sq2 backproject_update forward_sign build_maps >> $O/pmc_counters.txt
  python3 $R/scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json > /dev/null
  cat $O/pmc_counters.txt
  rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_fetch $O/pmc_write
fi
if has cfg5; then
  W5="--workload cfg5 --iterations 6"
  pmc fetch5 "$W5" FETCH_SIZE || exit 1
  pmc write5 "$W5" WRITE_SIZE || exit 1
  python3 $R/scripts/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/cfg5_traffic.json "1920x1080 RGB -> 3840x2160, scale 2, window 17 (the solver launches of the command the passes were run on)" > /dev/null
  rm -rf $O/pmc_fetch5 $O/pmc_write5
fi
if has tvl1; then
  PAIRS=4
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_t_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_t_$c -- python3 $R/bench.py --workload tvl1 --pairs-in-flight 1 --pairs-per-step 1 --steps $PAIRS --warmup 0 --no-cpu-baseline > $O/pmc_t_$c.log 2>&1
    echo "pmc tvl1 $c rc=$?"
  done
  python3 $R/scripts/pmc_tvl1.py $O/pmc_t_FETCH_SIZE $O/pmc_t_WRITE_SIZE $PAIRS $O/tvl1_traffic.json
  rm -rf $O/pmc_t_FETCH_SIZE $O/pmc_t_WRITE_SIZE
fi
rm -f $O/*.log
ls -la $O
