#!/bin/bash
# round 4, call Z: whole-year parity of the final tree, value by value — vector forcing, coarse array forcing (the LDS-staged taps), and the
# snow kernels against the oracle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04z; mkdir -p $o
{
echo "== python tools/year_parity.py (96 x 96 x 8760 h, vector forcing, all ten outputs)"
timeout -k 10 900 python tools/year_parity.py 2>&1 | grep -v amdgpu.ids
echo "== python tools/year_parity.py --coarse 8x8 --rows 64 --cols 64 (coarse array forcing through the LDS-staged taps)"
timeout -k 10 900 python tools/year_parity.py --coarse 8x8 --rows 64 --cols 64 2>&1 | grep -v amdgpu.ids
echo "== python tools/snow_rate.py --check"
timeout -k 10 900 python tools/snow_rate.py --check 2>&1 | grep -v amdgpu.ids
} > $o/year_parity.txt 2>&1
cat $o/year_parity.txt
