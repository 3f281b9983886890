#!/bin/bash
# per-kernel alone-times (LSX_SERIAL=1) of the all-atoms workload for one library variant:  bash profiles/kt_all.sh TAG LIB
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/kta_$TAG
LSX_SERIAL=1 LSX_HIP_LIBRARY=$LIB timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kta_$TAG -o kt -- python3 bench.py --workload all --steps 10 --warmup 2 --no-cpu-baseline --no-single-column > gpurun_out/kta_$TAG/run.log 2>&1 || exit 1
cp $(find gpurun_out/kta_$TAG -name "*kernel_stats.csv") gpurun_out/kta_${TAG}_stats.csv
find gpurun_out/kta_$TAG -name "*trace.csv" -delete
grep "fast_gamma\|fast_prepass\|gamma_finish" gpurun_out/kta_${TAG}_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
tail -c 400 gpurun_out/kta_$TAG/run.log | grep -o '"summary": {[^}]*}'
