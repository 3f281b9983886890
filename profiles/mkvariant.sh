#!/bin/bash
# Build one variant of the HIP library for profiles/ab.sh without touching the in-tree build:
#   bash profiles/mkvariant.sh NAME "-DLSX_SOMETHING ..." [full]
# compiles lsx_sweep.hip with the extra flags (5-ray instances only unless `full`) and links it with the in-tree
# lsx_hip.o into ab_so/NAME.so.
set -e
cd "$(dirname "$0")/../lightspinner_amd/csrc"
NAME=$1; XF=$2; FULL=$3
make -s build/lsx_hip.o build/lsx_setup.o
ONLY="-DLSX_ONLY_NR5"; [ "$FULL" = full ] && ONLY=""
mkdir -p ../../ab_so /tmp/lsxvar
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLSX_WAVES_PER_EU=4 $ONLY $XF -c lsx_sweep.hip -o /tmp/lsxvar/$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so build/lsx_hip.o build/lsx_setup.o /tmp/lsxvar/$NAME.o
echo "built ab_so/$NAME.so"
