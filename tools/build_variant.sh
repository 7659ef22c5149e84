#!/bin/bash
# Builds libmcfhip with extra compiler flags into build/variants/libmcfhip_<name>.so (travels to the GPU box with the
# snapshot; select it with MCF_LIB=...).  usage: [UNIT=mcf_snow] tools/build_variant.sh <name> [-DFLAG=1 ...]
# UNIT: the translation unit rebuilt with the flags (default mcf_kernels); the others are linked as built in-tree.
set -e
name=$1; shift
unit=${UNIT:-mcf_kernels}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
cd $root/microclimf_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-function -Wno-unused-value -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o $out/$unit.o $unit.hip
objs=""
for u in mcf_kernels mcf_api mcf_terrain mcf_snow; do
  if [ $u = $unit ]; then objs="$objs $out/$u.o"; else objs="$objs $u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/variants/libmcfhip_$name.so $objs mcf_pointmodel.o mcf_hydro.o
echo build/variants/libmcfhip_$name.so
