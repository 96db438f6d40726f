#!/bin/bash
# usage: mkvar.sh name "<flags>" [novgprform]
set -e
NAME=$1; FLAGS=$2; FORM="-mllvm -amdgpu-mfma-vgpr-form=1"; [ "$3" = novgprform ] && FORM=""
P=/root/repo/mods-light-zmq_amd
mkdir -p $P/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $FORM $FLAGS -c $P/csrc/match.hip -o $P/_variants/match_$NAME.o
OBJS=$(ls $P/csrc/*.o | grep -v "/match.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_variants/libmodsgpu_$NAME.so $OBJS $P/_variants/match_$NAME.o -L/opt/rocm/lib -lrccl
python3 /root/repo/tools/kernel_resources.py $P/_variants/libmodsgpu_$NAME.so | grep "match_nn1" | sed "s/^/$NAME: /"
