#!/bin/bash
# round 4, call U: what one exposed LDS latency per operand batch costs k_solve (timing variant extra_lds_wait: a second, dependent
# LDS round trip behind each of the 17 batches of a cell-step)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04u; mkdir -p $o
tools/ab_bench.sh $o/ab tree=- xlds=build/variants/libmcfhip_xlds.so tree2=- xlds2=build/variants/libmcfhip_xlds.so 2>&1 | tee $o/ab.txt
