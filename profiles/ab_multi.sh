#!/bin/bash
# A/B with time-only variants:  bash profiles/ab_multi.sh OUT ROUNDS "c3 c4" "checked variants" "time-only (ablation) variants"
cd "$GRAFT_REPO_ROOT"
OUT=$1; R=$2; WL=$3; CHK=$4; ABL=$5
: > $OUT
for r in $(seq 1 $R); do
  for w in $WL; do
    for V in $CHK; do
      timeout -k 10 300 python3 tests/ab_run.py ab_so/$V.so $w 2>&1 | tail -1 >> $OUT || exit 1
    done
    for V in $ABL; do
      LSX_AB_TIME_ONLY=1 timeout -k 10 300 python3 tests/ab_run.py ab_so/$V.so $w 2>&1 | tail -1 >> $OUT || exit 1
    done
  done
done
