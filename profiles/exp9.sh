cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_production_classes.py tests/test_response_and_columns.py -m gpu -x -q > gpurun_out/gputest.txt 2>&1; tail -5 gpurun_out/gputest.txt
for r in 1 2; do for w in c3 c4; do
  LSX_NO_RS=1 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs1.so $w 2>&1 | tail -1 | sed 's/^/NO_RS /'
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs1.so $w 2>&1 | tail -1 | sed 's/^/RS    /'
done; done
