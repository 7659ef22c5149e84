#!/bin/bash
# round 5, call B: streamed bioclim sink (bitwise against the whole-series sink), the closed-stomata shortcut in k_solve
# (parity + A/B), bioclim rate at 4096^2 with three ring budgets
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05b; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_bioclim_gpu.py tests/test_parity_gpu.py tests/test_dispatch_gpu.py tests/test_random_configs_gpu.py \
   tests/test_multi_device_gpu.py -x -q > $o/pytest.txt 2>&1
rc=$?; tail -5 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
STEPS=3 bash tools/ab_bench.sh $o base=build/variants/libmcfhip_r04base.so new=- base2=build/variants/libmcfhip_r04base.so new2=- 2>&1 | tee $o/ab.txt
for gb in 8 14 27; do
  echo "== bioclim 4096^2, MCF_BIOCLIM_RING_GB=$gb" | tee -a $o/bioclim_rate.txt
  MCF_BIOCLIM_RING_GB=$gb timeout -k 10 600 python tools/multi_rate.py --rows 4096 --cols 4096 --devices 0 --what bioclim 2>&1 | grep -v amdgpu.ids | tee -a $o/bioclim_rate.txt
done
echo "== whole-series form (round 4)" | tee -a $o/bioclim_rate.txt
MCF_BIOCLIM_WHOLE=1 timeout -k 10 600 python tools/multi_rate.py --rows 4096 --cols 4096 --devices 0 --what bioclim 2>&1 | grep -v amdgpu.ids | tee -a $o/bioclim_rate.txt
