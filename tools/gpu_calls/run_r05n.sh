#!/bin/bash
# round 5, call N: the driver's command on the final tree (default bench line), the configs[4] share with its counters, whole-year parity
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05n; mkdir -p $o
timeout -k 10 900 python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 -c "
import json; d=json.load(open('$o/bench_default.json')); r=d['roofline']; print('default', '%.4e' % d['value'], d['ms_per_step'], 'frac %.4f' % r['frac'], 'traffic', r.get('traffic'), (r.get('traffic_basis') or {}).get('ratio_to_algorithmic'), 'valu', (r.get('valu') or {}).get('per_cell_step'), (d.get('verified') or {}).get('ok')); print({k: (v.get('value'), v.get('traffic')) for k, v in (d.get('secondary') or {}).items()})"
timeout -k 10 600 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 > $o/bench_config4_share.json 2> $o/bench_config4_share.err
python3 -c "
import json; d=json.load(open('$o/bench_config4_share.json')); print('config4 share', '%.4e' % d['value'], d['ms_per_step'], (d.get('verified') or {}).get('ok'), d['roofline'].get('traffic'), d['roofline'].get('counters'))"
{
echo "== python tools/year_parity.py (96 x 96 x 8760 h, vector forcing, all ten outputs)"
timeout -k 10 900 python tools/year_parity.py 2>&1 | grep -v amdgpu.ids
echo "== python tools/year_parity.py --coarse 8x8 --rows 64 --cols 64 (coarse array forcing through the LDS-staged taps)"
timeout -k 10 900 python tools/year_parity.py --coarse 8x8 --rows 64 --cols 64 2>&1 | grep -v amdgpu.ids
} > $o/year_parity.txt 2>&1
cat $o/year_parity.txt
