#!/bin/bash
# round 5, call S: the snow-day microclimate with roughness and wind evaluated once per lane (not once per canopy class): snow tests,
# kernel times of the aux workload against the library before
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05s; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snow_gpu.py tests/test_random_snow_gpu.py tests/test_snowrun_gpu.py tests/test_snow_micro_pipeline_gpu.py tests/test_snowrun2_gpu.py tests/test_snowmodel2_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -4 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
bash tools/ab_snow.sh $o pre=build/variants/libmcfhip_prewind.so new=- pre2=build/variants/libmcfhip_prewind.so new2=- 2>&1 | tee $o/ab.txt
