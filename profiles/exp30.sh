cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for w in c3 c4; do for V in rs9 rs9t; do
  timeout -k 10 200 python3 tests/ab_run.py ab_so/$V.so $w 2>&1 | tail -1
done; done; done
