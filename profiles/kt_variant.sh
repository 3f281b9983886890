#!/bin/bash
# per-kernel times of library variants (ab_so/NAME.so ...) in one serialized C4 (or $WL) run each:
#   WL=c4 bash profiles/kt_variant.sh ab_so/a.so ab_so/b.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ktv
WL=${WL:-c4}; NC=${NC:-1250}
for V in "$@"; do
  N=$(basename $V .so)
  rm -rf gpurun_out/ktv/$N
  LSX_HIP_LIBRARY=$PWD/$V LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktv/$N -o kt -- python3 profiles/steptime.py $WL $NC 10 > gpurun_out/ktv/$N.log 2>&1 || exit 1
  python3 profiles/kt_sum.py $(find gpurun_out/ktv/$N -name '*kernel_trace.csv') 13 > gpurun_out/ktv/${N}_sum.txt
  find gpurun_out/ktv/$N -name '*kernel_trace.csv' -delete
  echo "== $N"; grep -E "fast|sum of" gpurun_out/ktv/${N}_sum.txt
done
