#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <tag> [bench args...]
# writes gpurun_out/<tag>_{kt,pmc1,pmc3,pmc4,calib}/...csv ; `python profiles/summarize.py gpurun_out/<tag> <ncol>`
# turns them into the JSON kept under profiles/.  Counter passes follow /opt/skills/guides/MI355X_MICROARCH.md
# (separate --pmc passes, no trace domains mixed in).  Every pass writes its own log, so a stuck pass is visible.
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-single-column --kernel-reps 5 $*"
echo "kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_kt.log 2>&1 &&
echo "pmc1"; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d gpurun_out/${TAG}_pmc1 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc1.log 2>&1 &&
echo "pmc3"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmc3 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc3.log 2>&1 &&
echo "pmc4"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/${TAG}_pmc4 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc4.log 2>&1
rc=$?
echo "calib"; timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_calib -o pmc --output-format csv -- python3 profiles/calibrate.py > gpurun_out/${TAG}_calib.log 2>&1
echo "collect exit=$rc"
