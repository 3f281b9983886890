#!/bin/bash
# A/B of prebuilt libraries on ONE box, interleaved rounds (cdna_hip_programming.md rule 24):
#   bash profiles/ab_so.sh ROUNDS "bench args" ab_so/a.so ab_so/b.so ...
# Build the variants in the container first (hipcc cross-compiles), e.g.
#   make -C lightspinner_amd/csrc XFLAGS=... && cp lightspinner_amd/csrc/liblsx_hip.so ab_so/x.so
cd "$GRAFT_REPO_ROOT"
R=$1; shift
ARGS=$1; shift
cp lightspinner_amd/csrc/liblsx_hip.so /tmp/ab_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp "$V" lightspinner_amd/csrc/liblsx_hip.so
    n=$(basename "$V" .so)
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-single-column $ARGS > gpurun_out/ab_${n}_$r.log 2>&1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${n}_$r.log").read().strip().splitlines()[-1])
f=d["roofline"]["fs_call"]
print("%-28s round $r: ms_sweep=%.3f ms_fs=%.3f ms_step=%.3f" % ("$n", d["roofline"]["avg_launch_ms"], f["ms"], d["ms_per_step"]))
PY
  done
done
cp /tmp/ab_keep.so lightspinner_amd/csrc/liblsx_hip.so
