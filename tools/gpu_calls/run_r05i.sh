#!/bin/bash
# round 5, call I: the array-weather snow run (mcf_runmicrosnow2) and the array-weather chunk loop against the oracle; the snow
# suite again (the snow plan's micro set-up and the solver's day runs were touched)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05i; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_snowrun2_gpu.py tests/test_snowmodel2_gpu.py tests/test_snowrun_gpu.py tests/test_snow_micro_pipeline_gpu.py \
    tests/test_snowfast_gpu.py tests/test_multi_device_gpu.py tests/test_parity_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -12 $o/pytest.txt
exit $rc
