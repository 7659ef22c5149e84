#!/bin/bash
# round 4, call AA: five waves per SIMD, MEASURED — the vector-forcing kernels built for <= 96 VGPRs (-DMCF_WAVES_PER_EU=5: 96-104 B of scratch
# per lane) with 16-cell tiles (6-wave workgroups: three per CU = 18 waves), against the shipped 21-cell tiles at four waves
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04aa; mkdir -p $o
{
tools/ab_bench.sh $o/ab tree=- tree2=-
EXTRA="--cells-per-block 16" tools/ab_bench.sh $o/ab tree_cpb16=- w5_cpb16=build/variants/libmcfhip_w5.so
tools/ab_bench.sh $o/ab w5_cpb21=build/variants/libmcfhip_w5.so
EXTRA="--cells-per-block 16" tools/ab_bench.sh $o/ab w5_cpb16b=build/variants/libmcfhip_w5.so
} 2>&1 | tee $o/ab.txt
