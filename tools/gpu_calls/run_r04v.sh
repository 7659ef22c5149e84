#!/bin/bash
# round 4, call V: what a scalar instruction costs k_solve (timing variant extra_salu: ten s_mov_b32 behind each of the 17 operand batches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04v; mkdir -p $o
tools/ab_bench.sh $o/ab tree=- xsalu=build/variants/libmcfhip_xsalu.so tree2=- xsalu2=build/variants/libmcfhip_xsalu.so 2>&1 | tee $o/ab.txt
