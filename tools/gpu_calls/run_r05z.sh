#!/bin/bash
# round 5, call Z: the days' mean ground-snow temperatures written by the snow model's kernel: snow tests, configs[4] share with stage
# times against the library before
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && o=gpurun_out/r05z && mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snow_gpu.py tests/test_random_snow_gpu.py tests/test_snowrun_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun2_gpu.py tests/test_snowmodel2_gpu.py tests/test_cells_run_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -3 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u tools/fuzz_snowrun.py --n 60 --seed 31 2>&1 | grep -v amdgpu.ids | tail -2
export MCF_BENCH_STAGES=1
for m in pre new pre2 new2; do
  case $m in pre*) export MCF_LIB=$GRAFT_REPO_ROOT/build/variants/libmcfhip_pretzd2.so;; *) unset MCF_LIB;; esac
  timeout -k 10 300 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-verify --no-cpu-baseline > $o/ab_$m.json 2> $o/ab_$m.err || exit 1
  python3 -c "
import json; d=json.load(open('$o/ab_$m.json')); print('$m', '%.4e' % d['value'], round(d['ms_per_step'],1), {k: round(v, 3) for k, v in (d.get('stage_seconds') or {}).items() if k in ('solver','microsnow','snowmodel+redistribute','meanD')})"
done
