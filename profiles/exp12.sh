cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ktv
V=$1
for WL in c3 c4; do
NC=$([ "$WL" = c4 ] && echo 1250 || echo 1000)
for MODE in rs nors; do
  N=$(basename $V .so)_${WL}_$MODE
  rm -rf gpurun_out/ktv/$N
  if [ $MODE = nors ]; then export LSX_NO_RS=1; else unset LSX_NO_RS; fi
  LSX_HIP_LIBRARY=$PWD/$V LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktv/$N -o kt -- python3 profiles/steptime.py $WL $NC 10 > gpurun_out/ktv/$N.log 2>&1 || exit 1
  python3 profiles/kt_sum.py $(find gpurun_out/ktv/$N -name '*kernel_trace.csv') 13 > gpurun_out/ktv/${N}_sum.txt
  find gpurun_out/ktv/$N -name '*kernel_trace.csv' -delete
  echo "== $N"; grep -E "sweep|sum of" gpurun_out/ktv/${N}_sum.txt
done; done
