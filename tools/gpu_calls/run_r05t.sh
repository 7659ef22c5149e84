#!/bin/bash
# round 5, call T: configs[4] share, stage times, the library before / after the snow-day microclimate's wind block was made common
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && o=gpurun_out/r05t && mkdir -p $o
export MCF_BENCH_STAGES=1
for m in pre new pre2 new2; do
  case $m in pre*) export MCF_LIB=$GRAFT_REPO_ROOT/build/variants/libmcfhip_prewind.so;; *) unset MCF_LIB;; esac
  timeout -k 10 300 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-verify --no-cpu-baseline > $o/ab_$m.json 2> $o/ab_$m.err || exit 1
  python3 -c "
import json; d=json.load(open('$o/ab_$m.json')); print('$m', '%.4e' % d['value'], round(d['ms_per_step'],1), {k: round(v, 3) for k, v in (d.get('stage_seconds') or {}).items()})"
done
