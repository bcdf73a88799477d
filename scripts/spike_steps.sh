#!/bin/bash
# step times of a 40-step run of the headline bench (exact synthetic flow) with the default library and every variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b
for lib in default variants/libsrx_*.so; do
  [ -e "$lib" ] || [ "$lib" = default ] || continue
  if [ "$lib" = default ]; then unset SRX_LIBRARY; else export SRX_LIBRARY=$PWD/$lib; fi
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --flow synthetic "$@" 2>/dev/null > gpurun_out/b/spike.json || exit 1
  python3 - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/b/spike.json").read().strip().splitlines()[-1])
s = d["config"]["step_ms"]
r = d["roofline"]
print(sys.argv[1], "mean", round(d["ms_per_step"], 2), "min", min(s), "max", max(s), "fwd/upd us",
      round(r["other_kernel"]["avg_launch_us"], 1), round(r["avg_launch_us"], 1), r["kernel"][:12])
print("   ", s[18:36])
PY
done
