#!/bin/bash
# round 5, call Q: the cell-subset tests, then configs[4] share with cells / with tiles as the unit of what the both-class days leave
# out, with stage times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05q
timeout -k 10 600 python -m pytest tests/test_cells_run_gpu.py tests/test_snowrun_gpu.py -x -q > gpurun_out/r05q/pytest.txt 2>&1 || { tail -20 gpurun_out/r05q/pytest.txt; exit 1; }
tail -2 gpurun_out/r05q/pytest.txt
export MCF_BENCH_STAGES=1
for m in tiles cells tiles2 cells2; do
  case $m in tiles*) export MCF_SNOW_NO_CELL_GATHER=1;; *) unset MCF_SNOW_NO_CELL_GATHER;; esac
  timeout -k 10 300 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-verify --no-cpu-baseline > gpurun_out/r05q/ab_$m.json 2> gpurun_out/r05q/ab_$m.err || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/r05q/ab_$m.json')); print('$m', '%.4e' % d['value'], round(d['ms_per_step'],1), d['config'].get('solver_cell_days_gathered'), {k: round(v, 3) for k, v in (d.get('stage_seconds') or {}).items()})"
done
