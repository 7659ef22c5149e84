#!/bin/bash
# Builds libmcfhip with extra compiler flags into build/variants/libmcfhip_<name>.so (travels to the GPU box with the
# snapshot; select it with MCF_LIB=...).  usage: tools/build_variant.sh <name> [-DFLAG=1 ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
cd $root/microclimf_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-function -Wno-unused-value -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o $out/mcf_kernels.o mcf_kernels.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/variants/libmcfhip_$name.so $out/mcf_kernels.o mcf_api.o mcf_terrain.o mcf_snow.o mcf_pointmodel.o mcf_hydro.o
echo build/variants/libmcfhip_$name.so
