#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot host API (mcf_runmicro1: H2D of the inputs, solve, D2H of
every requested output into host arrays).  Reported in DESIGN.md §3; never bench.py's `value`."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from microclimf_amd import synthetic          # noqa: E402
from microclimf_amd.api import runmicro1Cpp   # noqa: E402

rows, cols, T = (int(v) for v in (sys.argv[1:4] or (512, 512, 240)))
for out in ([1] * 10, [1] + [0] * 9):
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, out=out)
    runmicro1Cpp(**synthetic.workload(64, 64, 48, reqhgt=0.05, out=out))      # warm up
    r = None                                   # (freeing a previous 10 GB result is not part of the call)
    t0 = time.perf_counter()
    r = runmicro1Cpp(**a)
    dt = time.perf_counter() - t0
    nbytes = sum(v.nbytes for v in r.values())
    valid = int((~np.isnan(a["vegp"]["hgt"])).sum())
    print(f"{rows}x{cols}x{T} outputs={sum(out)}: {dt:.3f} s, {valid * T / dt:.3e} cell-steps/s end to end, "
          f"{nbytes / dt / 1e9:.1f} GB/s of output into host memory")
