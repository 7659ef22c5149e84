#!/bin/bash
# round 5, call X: whole-year parity value by value at the other heights and with array forcing (final kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05x; mkdir -p $o
{
for rq in 0.0 2.0 -0.1; do
  echo "== python tools/year_parity.py --rows 64 --cols 64 --reqhgt $rq"
  timeout -k 10 600 python -u tools/year_parity.py --rows 64 --cols 64 --reqhgt $rq 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "== python tools/year_parity.py --rows 48 --cols 48 --array"
timeout -k 10 600 python -u tools/year_parity.py --rows 48 --cols 48 --array 2>&1 | grep -v amdgpu.ids || exit 1
} 2>&1 | tee $o/year_parity_more.txt | grep --line-buffered -E "^==|worst|Error|assert"
