#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <tag> [bench args...]
# writes gpurun_out/<tag>_{kt,pmc1,pmc2,pmc3,pmc4,pmc5,calib}/...csv ; `python profiles/summarize.py gpurun_out/<tag> <ncol> <workload>`
# turns them into the JSON kept under profiles/.  Counter passes follow /opt/skills/guides/MI355X_MICROARCH.md
# (separate --pmc passes, no trace domains mixed in).  Every pass is its own statement with its own log and status: a
# pass that fails or times out is reported and STOPS the collection (no further GPU step after a killed one).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-single-column --no-c4-share --kernel-reps 5 $*"
rc=0
run_pass() {   # name, rocprofv3 options...
  local name=$1; shift
  echo "pass $name"
  timeout -k 10 300 rocprofv3 "$@" -d gpurun_out/${TAG}_${name} -o ${name%%[0-9]*} --output-format csv -- python3 bench.py $ARGS > gpurun_out/${TAG}_${name}.log 2>&1
  local s=$?
  echo "pass $name exit=$s"
  return $s
}
run_pass kt --kernel-trace --stats || rc=1
[ $rc = 0 ] && { run_pass pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES || rc=2; }
[ $rc = 0 ] && { run_pass pmc2 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || rc=3; }
[ $rc = 0 ] && { run_pass pmc3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE || rc=4; }
[ $rc = 0 ] && { run_pass pmc4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || rc=5; }
[ $rc = 0 ] && { run_pass pmc5 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU || rc=6; }
if [ $rc = 0 ]; then
  echo "pass calib"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_calib -o pmc --output-format csv -- python3 profiles/calibrate.py > gpurun_out/${TAG}_calib.log 2>&1 || rc=7
  echo "pass calib exit=$?"
fi
echo "collect exit=$rc"
exit $rc
