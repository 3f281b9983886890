cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/gputest.txt 2>&1; rc=$?; tail -5 gpurun_out/gputest.txt
[ $rc = 0 ] && bash profiles/ab.sh 3 "c3 c4" ab_so/base.so ab_so/diet1.so > gpurun_out/ab_diet1.txt 2>&1
cat gpurun_out/ab_diet1.txt
