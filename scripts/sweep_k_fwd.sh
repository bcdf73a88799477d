#!/bin/bash
# both solver kernels against the window size (development aid): a step from K = 8 to K = 9 in the forward
# kernel is the ninth wave doubling up on one SIMD
cd "$(dirname "$0")/.."
for K in ${KS:-1 2 3 5 7 8 9}; do
  echo "K=$K: $(timeout -k 10 120 python3 scripts/quick_time.py --K $K --iters 64 --reps 1 2>&1 | grep kernels:)"
done
