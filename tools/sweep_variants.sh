#!/bin/bash
# usage: tools_sweep.sh "<bench args>" lib1 lib2 ...   (prints value + avg launch ms per variant)
args="$1"; shift
for lib in "$@"; do
  for cpb in ${CPBS:-16 32}; do
    MCF_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --cells-per-block $cpb $args 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('$lib cpb=$cpb', '%.3e'%j['value'], 'avg_ms=%.3f'%j['roofline']['avg_launch_ms'])"
  done
done
