#!/bin/bash
# A/B of ONE library under different environments / option sets on one box, interleaved:
#   bash profiles/ab_env.sh OUT ROUNDS "c3 c4" LIB "VAR=a" "VAR=b" ...      (an empty string "" = the plain environment)
cd "$GRAFT_REPO_ROOT"
OUT=$1; R=$2; WL=$3; LIB=$4; shift 4
: > $OUT
for r in $(seq 1 $R); do
  for w in $WL; do
    for E in "$@"; do
      echo -n "[$E] " >> $OUT
      env $E timeout -k 10 300 python3 tests/ab_run.py $LIB $w 2>&1 | tail -1 >> $OUT || exit 1
    done
  done
done
