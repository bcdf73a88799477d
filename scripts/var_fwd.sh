#!/bin/bash
# solver kernel times of tuning variants (development aid): scripts/var_fwd.sh <variant name>...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=variants/libsrx_$v.so; [ "$v" = default ] && lib=superresolution_amd/libsrx.so
  for rep in 1 2; do
    echo "$v: $(SRX_LIBRARY=$lib timeout -k 10 120 python3 scripts/quick_time.py --iters 64 --reps 1 2>&1 | grep kernels: | cut -c1-80)"
  done
done
