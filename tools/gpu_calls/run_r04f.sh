#!/bin/bash
# round 4, call F: coarse array forcing with LDS-staged taps: parity + A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04f; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_coarse_forcing_gpu.py tests/test_multi_device_gpu.py tests/test_dispatch_gpu.py -x -q > $o/tests.log 2>&1; rc=$?
tail -5 $o/tests.log
[ $rc -ne 0 ] && exit $rc
for v in lds nolds lds2 nolds2; do
  if [ ${v#no} != $v ]; then export MCF_NO_COARSE_LDS=1; else unset MCF_NO_COARSE_LDS; fi
  timeout -k 10 300 python3 bench.py --config 1 --coarse 8x8 --ring-days 10 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $o/$v.json 2> $o/$v.err
  python3 -c "
import json; d=json.load(open('$o/$v.json')); print('$v', '%.4e'%d['value'], d['roofline']['avg_launch_ms'], d['verified']['ok'], d['verified']['max_scaled_err'])"
done
