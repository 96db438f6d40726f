#!/bin/bash
# Builds a variant of libmodsgpu.so with extra compiler flags on ONE translation unit (A/B measurements on the GPU box:
# MODS_LIB=<path> selects it).  usage: tools/variant.sh <name> <unit, e.g. sift> "<extra flags>"   ->  mods-light-zmq_amd/_variants/libmodsgpu_<name>.so
set -e
NAME=$1; UNIT=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/mods-light-zmq_amd
mkdir -p $P/_variants
EXTRA=""
[ "$UNIT" = "match" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA $FLAGS \
  -c $P/csrc/$UNIT.hip -o $P/_variants/${UNIT}_$NAME.o
OBJS=$(ls $P/csrc/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_variants/libmodsgpu_$NAME.so $OBJS $P/_variants/${UNIT}_$NAME.o -L/opt/rocm/lib -lrccl
echo $P/_variants/libmodsgpu_$NAME.so
