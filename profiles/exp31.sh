cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for w in c3 c4; do
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs10.so $w 2>&1 | tail -1 | sed 's/^/Q4 /'
  GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs10.so $w 2>&1 | tail -1 | sed 's/^/Q8 /'
  LSX_PRIO=1 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs10.so $w 2>&1 | tail -1 | sed 's/^/PRIO /'
done; done
