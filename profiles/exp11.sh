cd $GRAFT_REPO_ROOT
V=${1:-ab_so/rs3.so}
for r in 1 2; do for w in c3 c4; do
  LSX_NO_RS=1 timeout -k 10 200 python3 tests/ab_run.py $V $w 2>&1 | tail -1 | sed 's/^/NO_RS /'
  timeout -k 10 200 python3 tests/ab_run.py $V $w 2>&1 | tail -1 | sed 's/^/RS    /'
done; done
WL=c3 bash profiles/kt_variant.sh $V | grep -E "sweep|sum of"
WL=c4 bash profiles/kt_variant.sh $V | grep -E "sweep|sum of"
