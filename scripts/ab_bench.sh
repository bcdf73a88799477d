#!/bin/bash
# headline benchmark, variant library against the built one, alternating (development aid): scripts/ab_bench.sh <variant> [bench args...]
cd "$(dirname "$0")/.."
v=$1; shift
for rep in 1 2; do
  for lib in variants/libsrx_$v.so superresolution_amd/libsrx.so; do
    echo "$lib: $(SRX_LIBRARY=$lib timeout -k 10 150 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],2), "ms; update", round(r["avg_launch_us"],1), "forward", round(r["other_kernel"]["avg_launch_us"],1))')"
  done
done
