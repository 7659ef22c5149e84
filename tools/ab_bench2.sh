#!/bin/bash
# like ab_bench.sh, for the headline config (4096^2, device terrain): usage tools/ab_bench2.sh <outdir> name=lib[,ENV=VAL] ...
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=""
  if [ "$rest" != "$lib" ]; then envs=${rest#*,}; fi
  ( [ "$lib" != "-" ] && export MCF_LIB=$PWD/$lib; [ -n "$envs" ] && export ${envs//,/ }; \
    timeout -k 10 500 python3 bench.py --config ${CONFIG:-2} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-secondary --no-verify ${EXTRA} \
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
