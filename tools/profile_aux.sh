#!/bin/bash
# rocprofv3 kernel stats of the kernels outside k_solve (run on the GPU box via gpurun):
#   tools/profile_aux.sh <tag>  ->  gpurun_out/prof_<tag>_aux/...   (copy the *_kernel_stats.csv into profiles/)
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_${tag}_aux
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/aux_kernels_workload.py \
    > $out/aux.log 2> $out/trace.err
echo done
