#!/bin/bash
# issue / memory-path counters of one library variant's kernels on a short serialized run:
#   WL=c4 NC=400 bash profiles/pmc_variant.sh ab_so/x.so
# one --pmc pass (no trace domains), per-kernel sums printed by profiles/pmc_sum.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcv
WL=${WL:-c4}; NC=${NC:-400}; V=$1; N=$(basename $V .so)
rm -rf gpurun_out/pmcv/${N}_a
LSX_HIP_LIBRARY=$PWD/$V LSX_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d gpurun_out/pmcv/${N}_a -o pmc --output-format csv -- python3 profiles/steptime.py $WL $NC 2 > gpurun_out/pmcv/${N}_a.log 2>&1 || exit 1
python3 profiles/pmc_sum.py gpurun_out/pmcv/${N}_a | tee gpurun_out/pmcv/${N}_sum.txt
find gpurun_out/pmcv -name '*counter_collection.csv' -delete
