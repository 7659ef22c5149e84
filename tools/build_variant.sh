#!/bin/bash
# Builds a variant of libmcfhip into build/variants/libmcfhip_<name>.so (travels to the GPU box with the snapshot; select it
# with MCF_LIB=...).
#   usage: [UNIT=mcf_snow] [PATCH=tools/variants/x.patch[,y.patch]] tools/build_variant.sh <name> [-DFLAG=1 ...]
# UNIT: the translation unit rebuilt (default mcf_kernels); the others are linked as built in-tree.
# PATCH: timing / ablation variants of the kernels live as patches under tools/variants/ (results wrong on purpose, only the
# launch time is read); they are applied to a scratch copy of csrc/, never to the tree.
set -e
name=$1; shift
unit=${UNIT:-mcf_kernels}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
src=$root/microclimf_amd/csrc
if [ -n "$PATCH" ]; then
  rm -rf $out/src && mkdir -p $out/src/microclimf_amd/csrc $out/src/include
  cp $src/*.hip $src/*.h $src/*.hpp $src/*.cpp $out/src/microclimf_amd/csrc/
  cp $root/include/mcf.h $out/src/include/
  for p in ${PATCH//,/ }; do (cd $out/src && patch -s -p1 < $root/$p); done
  src=$out/src/microclimf_amd/csrc
fi
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-function -Wno-unused-value -Wno-pass-failed"
(cd $src && /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $out/$unit.o $unit.hip)
objs=""
cd $root/microclimf_amd/csrc
for u in mcf_kernels mcf_api mcf_terrain mcf_snow; do
  if [ $u = $unit ]; then objs="$objs $out/$u.o"; else objs="$objs $u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/variants/libmcfhip_$name.so $objs mcf_snowrun.o mcf_pointmodel.o mcf_hydro.o -lz -ldl
echo build/variants/libmcfhip_$name.so
