#!/bin/bash
# A/B of build variants on ONE box, interleaved rounds (cdna_hip_programming.md rule 24):
#   bash profiles/ab.sh ROUNDS "XFLAGS of variant A" "XFLAGS of variant B" ...
cd "$GRAFT_REPO_ROOT"
R=$1; shift
i=0
for V in "$@"; do
  make -s -C lightspinner_amd/csrc clean >/dev/null 2>&1
  make -s -j8 -C lightspinner_amd/csrc XFLAGS="$V" >/dev/null 2>&1 || { echo "build failed: $V"; exit 1; }
  cp lightspinner_amd/csrc/liblsx_hip.so /tmp/ab_$i.so
  i=$((i+1))
done
N=$i
for r in $(seq 1 $R); do
  for i in $(seq 0 $((N-1))); do
    cp /tmp/ab_$i.so lightspinner_amd/csrc/liblsx_hip.so
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-single-column > gpurun_out/ab_${i}_$r.log 2>&1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${i}_$r.log").read().strip().splitlines()[-1])
print("variant $i round $r: ms_sweep=%.3f ms_step=%.3f" % (d["roofline"]["avg_launch_ms"], d["ms_per_step"]))
PY
  done
done
cp /tmp/ab_0.so lightspinner_amd/csrc/liblsx_hip.so
