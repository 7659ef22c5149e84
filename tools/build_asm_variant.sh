#!/bin/bash
# Builds a variant of libmcfhip whose DEVICE code of one translation unit went through a filter on the compiler's assembly:
#   hipcc --cuda-device-only -S  ->  <filter> (a sed / python script: stdin -> stdout)  ->  assemble, link, bundle  ->  host
#   compile with that device binary embedded  ->  build/variants/libmcfhip_<name>.so  (select with MCF_LIB=...)
# usage: [UNIT=mcf_snow] tools/build_asm_variant.sh <name> '<filter command>'
# e.g.   tools/build_asm_variant.sh cnd64 "sed -E 's/v_cndmask_b32_e32 (.*), vcc$/v_cndmask_b32_e64 \1, vcc/'"
set -e
name=$1; filter=$2
unit=${UNIT:-mcf_kernels}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
src=$root/microclimf_amd/csrc
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-function -Wno-unused-value -Wno-pass-failed"
(cd $src && /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $out/$unit.dev.s $unit.hip)
bash -c "$filter" < $out/$unit.dev.s > $out/$unit.dev.filtered.s
echo "filter changed $(diff $out/$unit.dev.s $out/$unit.dev.filtered.s | grep -c '^>') lines"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/$unit.dev.filtered.s -o $out/$unit.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $out/$unit.dev.out $out/$unit.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$out/$unit.dev.out -output=$out/$unit.hipfb
(cd $src && /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $out/$unit.hipfb -c -o $out/$unit.o $unit.hip)
objs=""
cd $root/microclimf_amd/csrc
for u in mcf_kernels mcf_api mcf_terrain mcf_snow; do
  if [ $u = $unit ]; then objs="$objs $out/$u.o"; else objs="$objs $u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/build/variants/libmcfhip_$name.so $objs mcf_pointmodel.o mcf_hydro.o -lz -ldl
echo build/variants/libmcfhip_$name.so
