#!/bin/bash
# round 5, call J: fuzz of the snow run (vector / layered / array weather), the one-call snow run's rate with kept chunks pooled
# across a handle's years, the one-rank shares of configs[3] and configs[4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05j; mkdir -p $o
timeout -k 10 900 python tools/fuzz_snowrun.py --n 60 --seed 5 2>&1 | grep -v amdgpu.ids > $o/fuzz_snowrun.txt
rc=$?; tail -4 $o/fuzz_snowrun.txt
[ $rc -eq 0 ] || exit $rc
{ echo "== python tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200"; timeout -k 10 600 python tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200 2>&1 | grep -v amdgpu.ids; } > $o/snowrun_rate.txt
cat $o/snowrun_rate.txt
timeout -k 10 600 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_config4_share.json 2> $o/bench_config4_share.err
python3 -c "
import json; d=json.load(open('$o/bench_config4_share.json')); print('config4 share', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))"
timeout -k 10 600 python3 bench.py --config 3 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $o/bench_config3_share.json 2> $o/bench_config3_share.err
python3 -c "
import json; d=json.load(open('$o/bench_config3_share.json')); print('config3 share', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))"
