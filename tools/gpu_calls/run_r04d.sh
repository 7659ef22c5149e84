#!/bin/bash
# round 4, call D: bench lines with the new verification blocks (small sizes first, then the config-4 share)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04d; mkdir -p $o
timeout -k 10 600 python3 bench.py --config 4 --rows 512 --cols 256 --tsteps 1440 --share 8 --steps 1 --warmup 1 --no-cpu-baseline > $o/c4_small.json 2> $o/c4_small.err; echo "c4 small rc=$?"; tail -3 $o/c4_small.err
python3 -c "
import json; d=json.load(open('$o/c4_small.json')); v=d['verified']; print({k:v[k] for k in v if k not in ('what','snowmodel')}); print(v['snowmodel']['ok'], v['snowmodel']['max_scaled_err'])"
timeout -k 10 900 python3 bench.py --config 1 --steps 2 --warmup 1 --no-cpu-baseline > $o/c1.json 2> $o/c1.err; echo "c1 rc=$?"; tail -3 $o/c1.err
python3 -c "
import json; d=json.load(open('$o/c1.json')); print(d['value'], d['verified']['ok']);
for k,v in d['secondary'].items(): print(k, v.get('value'), v.get('verified',{}).get('ok'), v.get('verified',{}).get('max_scaled_err'), v.get('counters'), v.get('error'))"
