#!/bin/bash
# link libofd_hip_<tag>.so = current objects with one source (default conv_wp.hip) rebuilt under extra -D flags:
#   tools/build_wp_variants.sh tag "-DX=1 ..." [source.hip]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; FLAGS=$2; SRC=${3:-conv_wp.hip}
OBJ=$ROOT/opticalflowdiffusion_amd/lib/obj
hipcc -x hip -c $ROOT/opticalflowdiffusion_amd/csrc/$SRC -o /tmp/conv_wp_$TAG.o -I$ROOT/include -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $FLAGS
objs=$(ls $OBJ/*.o | grep -v "/$SRC.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/opticalflowdiffusion_amd/lib/libofd_hip_$TAG.so $objs /tmp/conv_wp_$TAG.o
echo built libofd_hip_$TAG.so
