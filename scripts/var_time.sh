#!/bin/bash
# time the default library and every variants/libsrx_*.so with quick_time.py
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in default variants/libsrx_*.so; do
  [ -e "$lib" ] || [ "$lib" = default ] || continue
  if [ "$lib" = default ]; then unset SRX_LIBRARY; else export SRX_LIBRARY=$PWD/$lib; fi
  timeout -k 10 120 python3 scripts/quick_time.py "$@" 2>&1 | grep "kernels:\|per-iteration\|rror" || exit 1
done
