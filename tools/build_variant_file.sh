#!/bin/bash
# libofd_hip_<tag>.so = current objects with ONE source rebuilt under extra flags: tools/build_variant_file.sh tag file.hip "-DX=1 ..."
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; SRC=$2; FLAGS=$3
OBJ=$ROOT/opticalflowdiffusion_amd/lib/obj
EXTRA=""; case $SRC in warp.hip|diffusion.hip) EXTRA="-ffp-contract=off";; esac
hipcc -x hip -c $ROOT/opticalflowdiffusion_amd/csrc/${SRCFILE:-$SRC} -o /tmp/${SRC}_$TAG.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $EXTRA $FLAGS
objs=$(ls $OBJ/*.o | grep -v "/$SRC.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/opticalflowdiffusion_amd/lib/libofd_hip_$TAG.so $objs /tmp/${SRC}_$TAG.o
echo built libofd_hip_$TAG.so
