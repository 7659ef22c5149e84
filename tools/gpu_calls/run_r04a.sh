#!/bin/bash
# round 4, call A: multi-wave issue costs, the unit of SQ_ACTIVE_INST_VALU, exp-table read issued early, setprio A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04a; mkdir -p $o
timeout -k 10 300 tools/microbench_waves.bin > $o/microbench_waves.txt 2>&1 || exit 1
echo "microbench done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o/mbw_pmc --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- tools/microbench_waves.bin --pmc > $o/mbw_pmc.log 2>&1 || exit 1
echo "microbench pmc done"
timeout -k 10 600 python -m pytest tests/test_math_gpu.py tests/test_parity_gpu.py tests/test_dispatch_gpu.py -x -q > $o/tests.log 2>&1 || { tail -20 $o/tests.log; exit 1; }
tail -2 $o/tests.log
tools/ab_bench.sh $o/ab tree=- r03base=build/variants/libmcfhip_r03base.so setprio_hi=build/variants/libmcfhip_setprio_hi.so setprio_lo=build/variants/libmcfhip_setprio_lo.so sections=build/variants/libmcfhip_sections.so tree2=- r03base2=build/variants/libmcfhip_r03base.so 2>&1 | tee $o/ab.txt
grep -h "mcf sections" $o/ab/sections.err > $o/sections.txt
