#!/bin/bash
# round 4, call AD: coarse forcing with the bounded exp (two more VGPR residents, 12 B of scratch) against the shipped kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04ad; mkdir -p $o
EXTRA="--coarse 8x8 --ring-days 10" tools/ab_bench.sh $o/ab tree=- cbexp=build/variants/libmcfhip_cbexp.so tree2=- cbexp2=build/variants/libmcfhip_cbexp.so 2>&1 | tee $o/ab.txt
