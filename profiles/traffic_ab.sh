#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, no trace domains) of a formal solution per kernel for library variants:
#   WL=c4 NC=1250 bash profiles/traffic_ab.sh OUT ab_so/a.so ab_so/b.so ...
# bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE in KiB, counts half the bytes on gfx950; calibrated)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/trf
OUT=$1; shift
WL=${WL:-c4}; NC=${NC:-1250}
: > $OUT
for V in "$@"; do
  N=$(basename $V .so)
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/trf/${N}_$C
    LSX_HIP_LIBRARY=$PWD/$V LSX_FS_ONLY=1 timeout -k 10 300 rocprofv3 --pmc $C -d gpurun_out/trf/${N}_$C -o pmc --output-format csv -- python3 profiles/steptime.py $WL $NC 2 > gpurun_out/trf/${N}_$C.log 2>&1 || exit 1
  done
  python3 profiles/traffic_sum.py $N $NC gpurun_out/trf/${N}_FETCH_SIZE gpurun_out/trf/${N}_WRITE_SIZE >> $OUT
  find gpurun_out/trf -name '*counter_collection.csv' -delete
done
cat $OUT
