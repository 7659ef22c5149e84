#!/bin/bash
# round 5, call O: the direct-beam factors hoisted per cell + two exact select removals: solver parity tests, same-box A/B against the
# round's earlier kernel and round 4's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05o; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_math_gpu.py tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_edge_cases_gpu.py \
   tests/test_random_configs_gpu.py tests/test_random_more_gpu.py tests/test_coarse_forcing_gpu.py tests/test_layers_gpu.py tests/test_multi_device_gpu.py tests/test_golden_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -4 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
STEPS=3 bash tools/ab_bench.sh $o pre=build/variants/libmcfhip_r05pre.so hoist1=build/variants/libmcfhip_hoist1.so hoist2=- pre2=build/variants/libmcfhip_r05pre.so hoist1b=build/variants/libmcfhip_hoist1.so hoist2b=- 2>&1 | tee $o/ab.txt
