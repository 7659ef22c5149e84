#!/bin/bash
# round 5, call L: the whole GPU suite on the final kernels, the snow-run fuzz (vector / layered / array weather), the one-call snow
# run's rate with chunks kept across a handle's years, the one-rank shares of configs[3] and configs[4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05l; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1
rc=$?; tail -4 $o/pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/fuzz_snowrun.py --n 60 --seed 5 2>&1 | grep -v amdgpu.ids > $o/fuzz_snowrun.txt
rc=$?; tail -3 $o/fuzz_snowrun.txt
[ $rc -eq 0 ] || exit $rc
{ echo "== python tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200   (1024 x 1024 x 365 days, Tz only into a host array)"; timeout -k 10 900 python tools/snowrun_rate.py --rows 1024 --cols 1024 --keep-gb 200 2>&1 | grep -v amdgpu.ids; } > $o/snowrun_rate.txt
cat $o/snowrun_rate.txt
