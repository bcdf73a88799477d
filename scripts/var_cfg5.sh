#!/bin/bash
# solver kernel times at the scale-2 configuration for tuning variants (development aid): scripts/var_cfg5.sh <variant name>...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=variants/libsrx_$v.so; [ "$v" = default ] && lib=superresolution_amd/libsrx.so
  echo "$v: $(SRX_LIBRARY=$lib timeout -k 10 120 python3 scripts/quick_time.py --rows 1080 --cols 1920 --scale 2 --K 17 --iters 24 --reps 1 2>&1 | grep kernels: | cut -c1-80)"
done
