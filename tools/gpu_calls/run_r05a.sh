#!/bin/bash
# round 5, call A: the k_solve instruction diet (one-fma exp reduction, folded satvap factor, 46-bit quotients, two more paired
# reciprocals, single dT cap, non-binding clamps dropped) — solver parity tests, then same-box A/B against round 4's library
# (first run of this call: math, parity, dispatch, edge-case and 99 random configurations green, 242 passed)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05a; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_math_gpu.py tests/test_coarse_forcing_gpu.py -x -q > $o/pytest2.txt 2>&1
rc=$?; tail -5 $o/pytest2.txt
[ $rc -eq 0 ] || exit $rc
STEPS=3 bash tools/ab_bench.sh $o base=build/variants/libmcfhip_r04base.so new=- base2=build/variants/libmcfhip_r04base.so new2=- 2>&1 | tee $o/ab.txt
