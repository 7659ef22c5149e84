#!/bin/bash
# round 5, call H: mcf_snowmodel2 (array-weather chunk loop) against the oracle and the host-orchestrated loop; then profile set b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05h; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_snowmodel2_gpu.py tests/test_snowfast_gpu.py -x -q > $o/pytest.txt 2>&1
tail -6 $o/pytest.txt
bash tools/gpu_calls/run_r05_profiles_b.sh 2>&1 | tail -12
