#!/bin/bash
# How a sweep class's time depends on the waves resident per SIMD: the same serialized run (LSX_SERIAL=1, one class after
# the other) with the workgroups per CU capped through the LDS request (LSX_OCC_WG, lsx_create), per-kernel times from a
# rocprofv3 kernel trace:
#   WL=c3 bash profiles/occupancy.sh TAG "4 6 8 10 0" [lib.so]
# (workgroups per CU: 4 = 2 waves per SIMD ... 10 = 5; 0 = no cap).  Time ~ 1/occupancy: latency bound; flat: a pipe is full.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/occ
TAG=$1; OCCS=${2:-"4 6 8 10 0"}; LIB=$3
WL=${WL:-c3}; NC=${NC:-$([ "$WL" = c4 ] && echo 1250 || echo 1000)}
[ -n "$LIB" ] && export LSX_HIP_LIBRARY=$PWD/$LIB
for O in $OCCS; do
  N=${TAG}_${WL}_occ$O
  rm -rf gpurun_out/occ/$N
  LSX_OCC_WG=$O LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ/$N -o kt -- python3 profiles/steptime.py $WL $NC 10 > gpurun_out/occ/$N.log 2>&1 || exit 1
  python3 profiles/kt_sum.py $(find gpurun_out/occ/$N -name '*kernel_trace.csv') 13 > gpurun_out/occ/${N}_sum.txt
  rm -rf gpurun_out/occ/$N
  echo "== $N"; grep -E "sweep|fast|finish|sum of" gpurun_out/occ/${N}_sum.txt
done
