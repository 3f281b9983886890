#!/bin/bash
# instruction / cycle counters of prebuilt library variants, per sweep class:
#   bash profiles/pmc_ab.sh "bench args" ab_so/a.so ab_so/b.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-single-column --kernel-reps 3 $1"; shift
cp lightspinner_amd/csrc/liblsx_hip.so /tmp/ab_keep.so
for V in "$@"; do
  n=$(basename "$V" .so)
  cp "$V" lightspinner_amd/csrc/liblsx_hip.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d gpurun_out/pab_${n}_1 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pab_${n}_1.log 2>&1 &&
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pab_${n}_2 -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pab_${n}_2.log 2>&1 || echo "pmc failed for $n"
done
cp /tmp/ab_keep.so lightspinner_amd/csrc/liblsx_hip.so
python3 profiles/pmc_ab_report.py "$@"
