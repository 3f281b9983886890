#!/bin/bash
# Build one variant of the HIP library for profiles/ab.sh without touching the in-tree build:
#   bash profiles/mkvariant.sh NAME "-DLSX_SOMETHING ..." [sweep|full|host|rs|rsp]
# all: lsx_hip.hip, both ray-serial units and the plan with the flags.
# sweep (default): compiles lsx_sweep.hip (5-ray instances only) and the host-side plan lsx_plan.cpp (they share the LDS
# layout of lsx_plan.h) with the extra flags, `full`: all instances, `host`: compiles lsx_hip.hip (runtime +
# fast-continuum kernels) with the flags instead; links with the other in-tree objects into ab_so/NAME.so.
set -e
cd "$(dirname "$0")/../lightspinner_amd/csrc"
NAME=$1; XF=$2; MODE=${3:-sweep}
XD=$(for f in $XF; do case $f in -D*) echo -n "$f ";; esac; done)      # what the host-side plan needs of the flags
make -s build/lsx_hip.o build/lsx_setup.o build/lsx_grid.o build/lsx_plan.o build/lsx_sweep.o build/lsx_sweep_rs.o build/lsx_sweep_rs_par.o build/lsx_build_id.o
printf 'extern "C" const char* lsx_build_id(void) { return "variant-%s"; }\n' "$1" > /tmp/lsxvar_id_$1.cpp && mkdir -p /tmp/lsxvar && g++ -O1 -fPIC -c /tmp/lsxvar_id_$1.cpp -o /tmp/lsxvar/id_$1.o
mkdir -p ../../ab_so /tmp/lsxvar
CF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLSX_WAVES_PER_EU=4"
if [ "$MODE" = host ]; then
  /opt/rocm/bin/hipcc $CF $XF -c lsx_hip.hip -o /tmp/lsxvar/$NAME.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so /tmp/lsxvar/$NAME.o build/lsx_setup.o /tmp/lsxvar/id_$NAME.o build/lsx_grid.o build/lsx_plan.o build/lsx_sweep.o build/lsx_sweep_rs.o build/lsx_sweep_rs_par.o
elif [ "$MODE" = rs ]; then     # the ray-serial sweep (lsx_sweep_rs.hip) with the flags
  /opt/rocm/bin/hipcc $CF $XF -c lsx_sweep_rs.hip -o /tmp/lsxvar/$NAME.o
  g++ -O2 -std=c++17 -fPIC $XD -c lsx_plan.cpp -o /tmp/lsxvar/${NAME}_plan.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so build/lsx_hip.o build/lsx_setup.o /tmp/lsxvar/id_$NAME.o build/lsx_grid.o /tmp/lsxvar/${NAME}_plan.o build/lsx_sweep.o /tmp/lsxvar/$NAME.o build/lsx_sweep_rs_par.o
elif [ "$MODE" = all ]; then    # every translation unit that sees lsx_plan.h's switches: runtime, both ray-serial units, the plan (e.g. -DLSX_ELANE=0: round 5's operand layout)
  /opt/rocm/bin/hipcc $CF $XF -c lsx_hip.hip -o /tmp/lsxvar/${NAME}_hip.o &
  /opt/rocm/bin/hipcc $CF $XF -c lsx_sweep_rs.hip -o /tmp/lsxvar/$NAME.o &
  /opt/rocm/bin/hipcc $CF -DLSX_RS_PARABOLIC_TU $XF -c lsx_sweep_rs.hip -o /tmp/lsxvar/${NAME}_par.o &
  g++ -O2 -std=c++17 -fPIC $XD -c lsx_plan.cpp -o /tmp/lsxvar/${NAME}_plan.o
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so /tmp/lsxvar/${NAME}_hip.o build/lsx_setup.o /tmp/lsxvar/id_$NAME.o build/lsx_grid.o /tmp/lsxvar/${NAME}_plan.o build/lsx_sweep.o /tmp/lsxvar/$NAME.o /tmp/lsxvar/${NAME}_par.o
elif [ "$MODE" = rsp ]; then    # the ray-serial instances of the parabolic rule (the same source with LSX_RS_PARABOLIC_TU) with the flags
  /opt/rocm/bin/hipcc $CF -DLSX_RS_PARABOLIC_TU $XF -c lsx_sweep_rs.hip -o /tmp/lsxvar/$NAME.o
  g++ -O2 -std=c++17 -fPIC $XD -c lsx_plan.cpp -o /tmp/lsxvar/${NAME}_plan.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so build/lsx_hip.o build/lsx_setup.o /tmp/lsxvar/id_$NAME.o build/lsx_grid.o /tmp/lsxvar/${NAME}_plan.o build/lsx_sweep.o build/lsx_sweep_rs.o /tmp/lsxvar/$NAME.o
else
  ONLY="-DLSX_ONLY_NR5"; [ "$MODE" = full ] && ONLY=""
  /opt/rocm/bin/hipcc $CF $ONLY $XF -c lsx_sweep.hip -o /tmp/lsxvar/$NAME.o
  g++ -O2 -std=c++17 -fPIC $XD -c lsx_plan.cpp -o /tmp/lsxvar/${NAME}_plan.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_so/$NAME.so build/lsx_hip.o build/lsx_setup.o /tmp/lsxvar/id_$NAME.o build/lsx_grid.o /tmp/lsxvar/${NAME}_plan.o /tmp/lsxvar/$NAME.o build/lsx_sweep_rs.o build/lsx_sweep_rs_par.o
fi
echo "built ab_so/$NAME.so"
