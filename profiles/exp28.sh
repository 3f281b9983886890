cd $GRAFT_REPO_ROOT
for r in 1 2; do for w in c3 c4; do
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs9.so $w 2>&1 | tail -1 | sed 's/^/A_npt1 /'
  LSX_RS_MAX_NPT=2 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs9.so $w 2>&1 | tail -1 | sed 's/^/B_npt2 /'
done; done
