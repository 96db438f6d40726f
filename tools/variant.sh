#!/bin/bash
# Builds a variant of libmodsgpu.so with extra compiler flags on some translation units (A/B measurements on the GPU box:
# MODS_LIB=<path> selects it).  usage: tools/variant.sh <name> "<units, e.g. sift describe>" "<extra flags>"
#   ->  mods-light-zmq_amd/_variants/libmodsgpu_<name>.so
set -e
NAME=$1; UNITS=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/mods-light-zmq_amd
mkdir -p $P/_variants
OBJS=$(ls $P/csrc/*.o)
for UNIT in $UNITS; do
  EXTRA=""
  [ "$UNIT" = "match" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $EXTRA $FLAGS \
    -c $P/csrc/$UNIT.hip -o $P/_variants/${UNIT}_$NAME.o &
  OBJS=$(echo "$OBJS" | grep -v "/$UNIT.o")
  OBJS="$OBJS $P/_variants/${UNIT}_$NAME.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_variants/libmodsgpu_$NAME.so $OBJS -L/opt/rocm/lib -lrccl
echo $P/_variants/libmodsgpu_$NAME.so
