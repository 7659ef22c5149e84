#!/bin/bash
# round 4, call AB: ONE call of mcf_runmicrosnow1 for a whole year into host arrays (what an R session gets from runmicro(snow = TRUE))
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04ab; mkdir -p $o
{
timeout -k 10 500 python tools/snowrun_rate.py --rows 512 --cols 512 --out Tz
timeout -k 10 500 python tools/snowrun_rate.py --rows 512 --cols 512 --out Tz,relhum,soilm
timeout -k 10 800 python tools/snowrun_rate.py --rows 1024 --cols 1024 --out Tz
} 2>&1 | grep -v amdgpu.ids | tee $o/snowrun_rate.txt
