#!/bin/bash
# A/B on the GPU box: the config-1 bench (1024^2 x 8760 h) once per library / environment variant.
# usage: tools/ab_bench.sh <outdir> name=lib[,ENV=VAL] ...     (lib "-" = the in-tree library)
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=""
  if [ "$rest" != "$lib" ]; then envs=${rest#*,}; fi
  ( [ "$lib" != "-" ] && export MCF_LIB=$PWD/$lib; [ -n "$envs" ] && export ${envs//,/ }; \
    timeout -k 10 300 python3 bench.py --config 1 --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-secondary ${EXTRA} \
      > $out/$name.json 2> $out/$name.err )
  python3 -c "
import json
try:
    d=json.load(open('$out/$name.json')); v=d.get('verified') or {}
    print('%-14s %.4e cell-steps/s  launch %.3f ms  verified err %.2e ok=%s' % ('$name', d['value'], d['roofline']['avg_launch_ms'], v.get('max_scaled_err', float('nan')), v.get('ok')))
except Exception as e:
    print('$name FAILED', e)
"
done
