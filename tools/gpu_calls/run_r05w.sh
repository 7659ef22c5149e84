#!/bin/bash
# round 5, call W: a longer soak of the one-call snow run (other seeds than the committed runs': vector / layered / array weather, row
# blocks, deep packs — the cell-subset launch and the tile mask both occur)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05w; mkdir -p $o
for seed in 11 23; do
  timeout -k 10 1000 python -u tools/fuzz_snowrun.py --n 120 --seed $seed 2>&1 | grep --line-buffered -v amdgpu.ids | tee $o/fuzz_$seed.txt | grep --line-buffered -E "^\[[0-9]*0\]|agree|Error|assert" || exit 1
done
