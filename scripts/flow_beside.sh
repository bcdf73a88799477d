#!/bin/bash
# headline bench with the built-in TV-L1 flow beside the solve: default library and every variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b
for lib in default variants/libsrx_*.so; do
  [ -e "$lib" ] || [ "$lib" = default ] || continue
  if [ "$lib" = default ]; then unset SRX_LIBRARY; else export SRX_LIBRARY=$PWD/$lib; fi
  timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/b/fb.json || exit 1
  python3 - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/b/fb.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "Mpix/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 2), d["config"]["step_ms"][:5], "upd/fwd us",
      round(r["avg_launch_us"], 1), round(r["other_kernel"]["avg_launch_us"], 1))
PY
done
