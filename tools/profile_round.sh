#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the bench command, then PMC passes (separate runs,
# counters only with --kernel-trace) for HBM traffic and SQ activity.
# Usage: tools/profile_round.sh <tag> [bench args, e.g. --config 1]     -> writes gpurun_out/prof_<tag>/...
set -e
tag=${1:-r02}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
echo "$@" > $out/bench_args.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py "$@" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-secondary > $out/bench_under_rocprof.json 2> $out/trace.err
echo "trace done"
S="python3 bench.py $* --tsteps 1920 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_fetch --pmc FETCH_SIZE -- $S > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_write --pmc WRITE_SIZE -- $S > /dev/null 2> $out/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $S > /dev/null 2> $out/pmc_sq.err
echo "sq done"
rocprofv3 --kernel-trace --output-format csv -d $out/pmc_misc --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $S > /dev/null 2> $out/pmc_misc.err
echo done
