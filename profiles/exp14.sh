cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_production_classes.py tests/test_response_and_columns.py -m gpu -x -q > gpurun_out/gputest.txt 2>&1; tail -3 gpurun_out/gputest.txt
for r in 1 2; do for w in c3 c4; do
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs4.so $w 2>&1 | tail -1 | sed 's/^/RS4   /'
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs5.so $w 2>&1 | tail -1 | sed 's/^/RS5   /'
done; done
bash profiles/exp12.sh ab_so/rs5.so | grep -E "==|rs_kernel|sum of"
