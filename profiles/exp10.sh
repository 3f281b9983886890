cd $GRAFT_REPO_ROOT
for r in 1 2; do for w in c3 c4; do
  LSX_NO_RS=1 timeout -k 10 200 python3 tests/ab_run.py ab_so/rs1.so $w 2>&1 | tail -1 | sed 's/^/NO_RS /'
  timeout -k 10 200 python3 tests/ab_run.py ab_so/rs_w2.so $w 2>&1 | tail -1 | sed 's/^/RS    /'
done; done
WL=c3 bash profiles/kt_variant.sh ab_so/rs_w2.so
WL=c4 bash profiles/kt_variant.sh ab_so/rs_w2.so
cat gpurun_out/ktv/rs_w2_sum.txt
