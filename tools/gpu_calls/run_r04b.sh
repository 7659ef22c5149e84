#!/bin/bash
# round 4, call B: store flavour and selector-cost variants, counters list
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04b; mkdir -p $o
rocprofv3 -L > $o/counters.txt 2>&1
tools/ab_bench.sh $o/ab tree=- store_nt=build/variants/libmcfhip_store_nt.so allout=build/variants/libmcfhip_allout.so tree2=- store_nt2=build/variants/libmcfhip_store_nt.so allout2=build/variants/libmcfhip_allout.so 2>&1 | tee $o/ab.txt
