#!/bin/bash
# round 4, call I: what an R session sees at size through the node route (one device is all there is: unmeasured on N > 1 hardware)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04i; mkdir -p $o
{
echo "== mcf_runmicro1_multi (Tz only into a numpy array) and mcf_runbioclim1_multi, 4096 x 4096, 4 row blocks time-sliced on ONE device"
timeout -k 10 900 python tools/multi_rate.py --rows 4096 --cols 4096 --tsteps 240 --devices 0 --blocks 4 --what solver,bioclim
echo "== the same with one block (the single-device call through the same driver)"
timeout -k 10 900 python tools/multi_rate.py --rows 4096 --cols 4096 --tsteps 240 --devices 0 --blocks 1 --what solver,bioclim
echo "== packed netCDF pipeline (classic container), 4096 x 4096 x 30 days, Tz, into /dev/shm"
timeout -k 10 900 python tools/pipeline_rate.py --rows 4096 --cols 4096 --days 30 --vars Tz --dir /dev/shm
} > $o/sink_rates.txt 2>&1
cat $o/sink_rates.txt | grep -v amdgpu.ids
