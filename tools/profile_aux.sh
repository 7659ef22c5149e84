#!/bin/bash
# rocprofv3 of the kernels outside k_solve (run on the GPU box via gpurun): kernel stats, then two counter passes
# (counters only with --kernel-trace, separate runs) for the snow kernels' VALU work and activity.
#   tools/profile_aux.sh <tag> [python3 args ...]  ->  gpurun_out/prof_<tag>_aux/...   (tools/summarize_aux.py condenses it into profiles/)
# default workload: tools/aux_kernels_workload.py; e.g. `tools/profile_aux.sh r04_c4 bench.py --config 4 --share 8 --steps 1 --warmup 0
# --no-cpu-baseline --no-verify` profiles the kernels of the configs[4] pipeline as that pipeline runs them (k_microsnow_ring,
# k_snowmodel on the year's chunks)
set -e
tag=${1:-r02}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_${tag}_aux
mkdir -p $out
if [ $# -gt 0 ]; then W="$*"; else W="tools/aux_kernels_workload.py"; fi
echo "$W" > $out/workload.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $W \
    > $out/aux.log 2> $out/trace.err
echo "trace done"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 $W > /dev/null 2> $out/pmc_sq.err
echo "sq done"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_misc --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -- python3 $W > /dev/null 2> $out/pmc_misc.err
echo "misc done"
# HBM bytes of the same kernels (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes)
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_fetch --pmc FETCH_SIZE -- python3 $W > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_write --pmc WRITE_SIZE -- python3 $W > /dev/null 2> $out/pmc_write.err
echo done
