#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <tag> [bench args...]
# writes gpurun_out/<tag>_{kt,pmc1..4}/...csv ; copy the summaries you want judged into profiles/.
# Counter passes follow /opt/skills/guides/MI355X_MICROARCH.md (separate --pmc passes, no trace domains mixed in).
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-single-column --kernel-reps 5 $*"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_kt.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d gpurun_out/${TAG}_pmc1 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/${TAG}_pmc2 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmc3 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc3.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/${TAG}_pmc4 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc4.log 2>&1
rc=$?
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_calib -o pmc --output-format csv -- python3 profiles/calibrate.py > gpurun_out/${TAG}_calib.log 2>&1
echo "collect exit=$rc"
ls gpurun_out/${TAG}_pmc1 2>/dev/null | head -3
