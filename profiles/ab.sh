#!/bin/bash
# A/B of prebuilt library variants on ONE box, interleaved rounds (cdna_hip_programming.md rule 24):
#   bash profiles/ab.sh ROUNDS "c3 c4" ab_so/a.so ab_so/b.so ...
# Build the variants in the container first (hipcc cross-compiles), e.g.
#   make -C lightspinner_amd/csrc XFLAGS=-DLSX_X && cp lightspinner_amd/csrc/liblsx_hip.so ab_so/x.so
# The in-tree library is never touched: every variant is loaded from its own path (tests/ab_run.py: it checks every variant against the oracle, so it lives with the tests).
cd "$GRAFT_REPO_ROOT"
R=$1; shift
WL=$1; shift
for r in $(seq 1 $R); do
  for w in $WL; do
    for V in "$@"; do
      timeout -k 10 200 python3 tests/ab_run.py "$V" $w 2>&1 | tail -1 || exit 1
    done
  done
done
