#!/bin/bash
# occupancy experiment: rebuild the sweep kernel for several waves-per-SIMD targets on the GPU box and bench each
cd "$GRAFT_REPO_ROOT"
for W in "$@"; do
  make -s -C lightspinner_amd/csrc clean >/dev/null 2>&1
  make -s -j8 -C lightspinner_amd/csrc WPE=$W >/dev/null 2>&1 || { echo "build failed WPE=$W"; continue; }
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/wpe_$W.log 2>&1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/wpe_$W.log").read().strip().splitlines()[-1])
r=d["roofline"]; print("WPE=$W", "ms_sweep=%.3f"%r["avg_launch_ms"], "GB/s=%.0f"%r["achieved"], "lds=%d"%r["lds_bytes_per_workgroup"], "single=%.4f"%d["falc_single_column"]["seconds"])
PY
done
