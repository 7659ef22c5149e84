#!/bin/bash
# round 5, call P: the solver for a subset of the cells (mcf_plan_run_days_cells) — its tests, the snow-run tests and fuzz over it,
# and the configs[4] share with cells / with tiles as the unit of what the snow days leave out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05p; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_cells_run_gpu.py tests/test_snowrun_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun2_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -15 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/fuzz_snowrun.py --n 30 --seed 7 2>&1 | grep -v amdgpu.ids > $o/fuzz_snowrun.txt
rc=$?; tail -3 $o/fuzz_snowrun.txt
[ $rc -eq 0 ] || exit $rc
for m in cells tiles; do
  if [ $m = tiles ]; then export MCF_SNOW_NO_CELL_GATHER=1; fi
  timeout -k 10 600 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_config4_$m.json 2> $o/bench_config4_$m.err
  python3 -c "
import json; d=json.load(open('$o/bench_config4_$m.json')); print('$m', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'), (d.get('verified') or {}).get('max_scaled_err'), d['config'].get('solver_cell_days_gathered'), d['config']['solver_tile_days_left_out'][:8], d.get('stage_seconds'))"
done
