#!/bin/bash
# the round's rocprofv3 evidence for both workloads: bash profiles/collect_all.sh <tag>  ->  gpurun_out/<tag>_c3_*, <tag>_c4_*
TAG=$1
cd "$GRAFT_REPO_ROOT"
bash profiles/collect.sh ${TAG}_c3 --workload c3 && bash profiles/collect.sh ${TAG}_c4 --workload c4
