cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for w in c3 c4; do
  LSX_NO_RS=1 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs5.so $w 2>&1 | tail -1 | sed 's/^/NO_RS   /'
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs5.so $w 2>&1 | tail -1 | sed 's/^/RS_ALL  /'
  LSX_RS_MAX_NPT=1 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs5.so $w 2>&1 | tail -1 | sed 's/^/RS_NPT1 /'
  LSX_RS_MAX_NPT=0 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs5.so $w 2>&1 | tail -1 | sed 's/^/RS_NPT0 /'
done; done
