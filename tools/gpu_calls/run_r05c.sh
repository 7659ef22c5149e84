#!/bin/bash
# round 5, call C: the tile-shaped snow-day microclimate kernel (k_microsnow_tiles) — snow pipeline tests, then configs[4]'s
# one-rank share with the new and the old shape on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05c; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -5 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
for v in new old new2 old2; do
  case $v in old*) export MCF_MICRORING_OLD=1;; *) unset MCF_MICRORING_OLD;; esac
  timeout -k 10 600 python3 bench.py --config 4 --share 8 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $o/c4_$v.json 2> $o/c4_$v.err
  python3 -c "
import json; d=json.load(open('$o/c4_$v.json')); print('$v', '%.4e' % d['value'], d.get('ms_per_step'), json.dumps(d.get('stages') or d.get('config',{}).get('stages') or {}))"
done 2>&1 | tee $o/ab.txt
