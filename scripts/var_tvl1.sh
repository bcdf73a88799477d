#!/bin/bash
# TV-L1 stage rate on 1920x1080 pairs for tuning variants (development aid): scripts/var_tvl1.sh <variant name>...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=variants/libsrx_$v.so; [ "$v" = default ] && lib=superresolution_amd/libsrx.so
  echo "$v: $(SRX_LIBRARY=$lib timeout -k 10 120 python3 bench.py --workload tvl1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "pairs/s", round(d["ms_per_step"],3), "ms")')"
done
