#!/bin/bash
# developer tool: build a tuning variant of the product library under mpc_benchmark_amd/csrc/variants/ (for tools/exp_variants.sh):
#   tools/build_variant.sh NAME UNIT "-DFLAG ..."      UNIT = eval | ric | both: which translation unit gets the flags (the others are the default objects)
set -e
cd "$(dirname "$0")/../mpc_benchmark_amd/csrc"
NAME=$1; UNIT=$2; FLAGS=$3
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-prealloc-sgpr-spill-vgprs -Wall -Wno-unused-variable -Wno-unused-but-set-variable"
mkdir -p variants
make -s -j3 mpc_hip.o eval_multibody.o qp.o
EV=eval_multibody.o; RIC=mpc_hip.o
if [ "$UNIT" = eval ] || [ "$UNIT" = both ]; then EV=variants/eval_$NAME.o; hipcc $HIPFLAGS $FLAGS -c -o $EV eval_multibody.hip & fi
if [ "$UNIT" = ric ] || [ "$UNIT" = both ]; then RIC=variants/ric_$NAME.o; hipcc $HIPFLAGS $FLAGS -c -o $RIC mpc_hip.hip & fi
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o variants/libmpc_hip_$NAME.so $RIC $EV qp.o
echo built variants/libmpc_hip_$NAME.so
