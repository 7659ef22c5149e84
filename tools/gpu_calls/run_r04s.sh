#!/bin/bash
# round 4, call S: how much of k_solve's time the bank conflicts of the exp / log table gathers cost (timing variant: every lane reads
# one of two neighbouring entries — wrong results, conflict-free gathers)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04s; mkdir -p $o
tools/ab_bench.sh $o/ab tree=- tabnc=build/variants/libmcfhip_tabnc.so tree2=- tabnc2=build/variants/libmcfhip_tabnc.so 2>&1 | tee $o/ab.txt
