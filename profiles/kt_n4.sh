# per-kernel times of one parabolic formal solution (N4), classes serialized: bash profiles/kt_n4.sh TAG [lib.so]   (C3, 1000 columns)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/kt
TAG=$1; LIB=$2; [ -n "$LIB" ] && export LSX_HIP_LIBRARY=$PWD/$LIB
for POL in auto ray-per-lane; do
  N=${TAG}_$POL
  LSX_STEP_SOLVER=parabolic LSX_STEP_POLICY=$POL LSX_FS_ONLY=1 LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt/$N -o kt -- python3 profiles/steptime.py ${WL:-c3} ${NC:-1000} 10 > gpurun_out/kt/$N.log 2>&1 || exit 1
  python3 profiles/kt_sum.py $(find gpurun_out/kt/$N -name '*kernel_trace.csv') 13 > gpurun_out/kt/${N}_sum.txt
  find gpurun_out/kt/$N -name '*kernel_trace.csv' -delete
  echo "== $N"; grep -E "sweep|sum of" gpurun_out/kt/${N}_sum.txt
done
