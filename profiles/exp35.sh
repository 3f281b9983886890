cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_production_classes.py tests/test_toy_topologies.py tests/test_response_and_columns.py -m gpu -x -q > gpurun_out/gputest.txt 2>&1; tail -2 gpurun_out/gputest.txt
for r in 1 2 3; do for w in c3 c4; do for V in rs11 rs12; do
  timeout -k 10 200 python3 tests/ab_run.py ab_so/$V.so $w 2>&1 | tail -1
done; done; done
