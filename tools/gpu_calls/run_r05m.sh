#!/bin/bash
# round 5, call M: the one-call snow run's rate with chunks kept across a handle's years; one-rank shares of configs[3] / configs[4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05m; mkdir -p $o
echo "== python tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200   (1024 x 1024 x 365 days, Tz only into a host array)" | tee $o/snowrun_rate.txt
timeout -k 10 1000 python -u tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $o/snowrun_rate.txt
timeout -k 10 600 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_config4_share.json 2> $o/bench_config4_share.err
python3 -c "
import json; d=json.load(open('$o/bench_config4_share.json')); print('config4 share', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'), d['roofline'].get('traffic'), d['roofline'].get('counters'))"
timeout -k 10 600 python3 bench.py --config 3 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $o/bench_config3_share.json 2> $o/bench_config3_share.err
python3 -c "
import json; d=json.load(open('$o/bench_config3_share.json')); print('config3 share', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))"
