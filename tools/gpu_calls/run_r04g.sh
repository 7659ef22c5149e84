#!/bin/bash
# round 4, call G: k_microsnow_ring as lane-per-(cell, hour): snow tests, then the configs[4] share old vs new
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04g; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snow_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun_gpu.py tests/test_random_snow_gpu.py tests/test_frontend_gpu.py -x -q > $o/tests.log 2>&1; rc=$?
tail -4 $o/tests.log
[ $rc -ne 0 ] && exit $rc
for v in new new2; do
  if [ "${v#old}" != "$v" ]; then export MCF_LIB=$PWD/build/variants/libmcfhip_oldmicro.so; else unset MCF_LIB; fi
  MCF_BENCH_STAGES=1 timeout -k 10 500 python3 bench.py --config 4 --share 8 --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $o/$v.json 2> $o/$v.err
  python3 -c "
import json; d=json.load(open('$o/$v.json')); print('$v', '%.4e'%d['value'], '%.1f ms/year'%d['ms_per_step'], {k: round(x,3) for k,x in (d.get('stage_seconds') or {}).items()})"
done
