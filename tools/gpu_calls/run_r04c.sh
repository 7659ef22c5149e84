#!/bin/bash
# round 4, call C: the whole-snow-run entry; snow multi comparison; icache counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04c; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snowrun_gpu.py tests/test_snow_gpu.py tests/test_snow_micro_pipeline_gpu.py -x -q > $o/tests.log 2>&1; rc=$?
tail -25 $o/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/snow_multi_cmp.py > $o/snow_multi_cmp.txt 2>&1; tail -8 $o/snow_multi_cmp.txt
