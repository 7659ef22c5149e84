#!/bin/bash
# array-forcing A/B on one box: usage tools/ab_af.sh <outdir> name=cpb[,ENV=VAL] ...   (EXTRA: more bench.py flags, e.g. --coarse 8x8)
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  cpb=${rest%%,*}; envs=""
  if [ "$rest" != "$cpb" ]; then envs=${rest#*,}; fi
  ( [ -n "$envs" ] && export ${envs//,/ }; \
    timeout -k 10 400 python3 bench.py --config 1 --array-forcing --cells-per-block $cpb --tsteps ${TSTEPS:-1920} --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-verify ${EXTRA} \
      > $out/$name.json 2> $out/$name.err )
  python3 -c "
import json
try:
    d=json.load(open('$out/$name.json'))
    print('%-14s %.4e cell-steps/s  launch %.3f ms  dispatch %s' % ('$name', d['value'], d['roofline']['avg_launch_ms'], d['config'].get('dispatch')))
except Exception as e:
    print('$name FAILED', e)
"
done
