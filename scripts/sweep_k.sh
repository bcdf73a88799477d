#!/bin/bash
# update-kernel time against the window size and with / without the BTV term (development aid)
cd "$(dirname "$0")/.."
for lam in 0.03 0; do for K in 1 3 5 9 13; do
  echo "lam=$lam K=$K: $(timeout -k 10 120 python3 scripts/quick_time.py --K $K --lam $lam --iters 12 --reps 2 2>&1 | grep kernels:)"
done; done
