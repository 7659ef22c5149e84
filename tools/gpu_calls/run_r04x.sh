#!/bin/bash
# round 4, call X: the lean terrain stencils (32-bit indices, exact three-instruction quotient) against the previous library:
# results bit for bit, kernel times under rocprofv3 --stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04x; mkdir -p $o
MCF_LIB_B=build/variants/libmcfhip_prevterrain.so python tools/terrain_ab.py 3000 2000 2>&1 | grep -v amdgpu.ids > $o/bits.txt || { cat $o/bits.txt; exit 1; }
cat $o/bits.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/new -- python3 tools/terrain_ab.py --child 4096 4096 /tmp/n.npz > /dev/null 2> $o/new.err
MCF_LIB=$GRAFT_REPO_ROOT/build/variants/libmcfhip_prevterrain.so rocprofv3 --kernel-trace --stats --output-format csv -d $o/old -- python3 tools/terrain_ab.py --child 4096 4096 /tmp/o.npz > /dev/null 2> $o/old.err
for v in new old; do echo "== $v"; f=$(ls $o/$v/*/*_kernel_stats.csv | head -1); grep -E "k_horizon|k_windcoef|k_block_mean|k_resample|k_slope" $f | cut -d, -f1-4 | cut -c1-150; done
python -m pytest tests/test_terrain_gpu.py tests/test_multi_device_gpu.py tests/test_snow_gpu.py -x -q -m gpu 2>&1 | tail -2
