#!/bin/bash
# CPU-only: builds the oracle with AddressSanitizer + UBSan and runs the CPU test-suite and every parity / snow
# case's oracle side through it (sanitizers are not available on the GPU pool; the oracle is the CPU build).
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -std=c99 -shared \
    -o /tmp/liborc_asan.so oracle/oracle_unit.c -lm
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MCF_ORACLE_LIB=/tmp/liborc_asan.so
python -m pytest tests -x -q -m "not gpu" --deselect tests/test_branch_coverage_cpu.py --deselect tests/test_distributed_cpu.py
python - <<'PY'
import sys; sys.path.insert(0, "tests")
import parity_cases as P, snow_cases as SC
from oracle import oracle as O
n = 0
for name in P.CASES:
    a, af = P.build(name); O.run_grid(**a, array_forcing=af); n += 1
for name in SC.SNOW_CASES:
    sw, af = SC.build_snow(name)
    smod = O.run_snowmodel(**SC.model_args(sw), array_forcing=af)
    snowm, micro = SC.microsnow_state(sw, smod)
    for h in SC.MICRO_HEIGHTS:
        O.run_microsnow(h, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0, [1] * 10, array_forcing=af)
    n += 1
print("cases through the sanitizer build:", n)
PY
