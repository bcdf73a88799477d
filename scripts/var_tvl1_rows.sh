#!/bin/bash
# TV-L1 stage rate on 1920x1080 and 960x540 pairs for settings of the whole-row iteration kernel (development aid):
# scripts/var_tvl1_rows.sh "kt_big,nt_big,kt_mid,nt_mid" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  for size in "1080 1920" "540 960"; do
    set -- $size
    echo "$v ${2}x${1}: $(SRX_TVL1_ROWS=$v timeout -k 10 120 python3 bench.py --workload tvl1 --rows $1 --cols $2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "pairs/s", round(d["ms_per_step"],3), "ms; frac", round(d["roofline"]["frac"],3), "iters", d["config"]["executed_inner_iterations_per_pair"])')"
  done
done
