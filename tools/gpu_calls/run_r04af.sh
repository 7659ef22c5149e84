#!/bin/bash
# round 4, call AF: compiler-flag variants of k_solve (no post-RA scheduler; -O2 behind -O3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04af; mkdir -p $o
tools/ab_bench.sh $o/ab tree=- nopostsched=build/variants/libmcfhip_nopostsched.so o2=build/variants/libmcfhip_o2.so tree2=- nopostsched2=build/variants/libmcfhip_nopostsched.so o2b=build/variants/libmcfhip_o2.so 2>&1 | tee $o/ab.txt
