#!/bin/bash
# per-kernel alone-times (LSX_SERIAL=1) of a formal solution for one library variant:  bash profiles/kt_lib.sh TAG LIB "c3 c4" [ENV=..]
TAG=$1; LIB=$2; WL=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/kt_$TAG
for w in $WL; do
  n=1000; [ $w = c4 ] && n=1250
  env LSX_SERIAL=1 LSX_HIP_LIBRARY=$LIB "$@" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$TAG/$w -o kt -- python3 profiles/steptime.py $w $n 10 > gpurun_out/kt_$TAG/$w.log 2>&1 || exit 1
  python3 profiles/kt_sum.py $(find gpurun_out/kt_$TAG/$w -name '*kernel_trace.csv') 13 > gpurun_out/kt_$TAG/${w}_sum.txt
  find gpurun_out/kt_$TAG/$w -name '*.csv' -delete
done
