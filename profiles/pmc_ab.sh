#!/bin/bash
# counters of library variants on a short serialized run, two --pmc passes each (no trace domains):
#   WL=c3 NC=400 bash profiles/pmc_ab.sh ab_so/a.so ab_so/b.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcv
WL=${WL:-c3}; NC=${NC:-400}
for V in "$@"; do
  N=$(basename $V .so)
  for P in a b; do
    rm -rf gpurun_out/pmcv/${N}_$P
    if [ $P = a ]; then C="SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; else C="SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"; fi
    LSX_HIP_LIBRARY=$PWD/$V LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc $C -d gpurun_out/pmcv/${N}_$P -o pmc --output-format csv -- python3 profiles/steptime.py $WL $NC 2 > gpurun_out/pmcv/${N}_$P.log 2>&1 || exit 1
    echo "== $N pass $P"; python3 profiles/pmc_sum.py gpurun_out/pmcv/${N}_$P | grep -E "sweep|kernel" | tee gpurun_out/pmcv/${N}_${P}_sum.txt
    find gpurun_out/pmcv/${N}_$P -name '*counter_collection.csv' -delete
  done
done
