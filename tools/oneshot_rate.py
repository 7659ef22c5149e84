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

from microclimf_amd.api import runmicro2Cpp   # noqa: E402

af = "--af" in sys.argv                        # array climate: the 15 forcing arrays cross PCIe too
coarse = "--coarse" in sys.argv                # coarse array climate (8 x 8), interpolated inside the solver
argv = [v for v in sys.argv[1:] if v not in ("--af", "--coarse")]
rows, cols, T = (int(v) for v in (argv[0:3] or (512, 512, 240)))
for out in ([1] * 10, [1] + [0] * 9):
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, out=out, array_forcing=af)
    w = synthetic.workload(64, 64, 48, reqhgt=0.05, out=out, array_forcing=af)
    fn = runmicro1Cpp
    if af:
        fn = runmicro2Cpp
        for d in (a, w):
            d["lats"], d["lons"] = d.pop("lat"), d.pop("lon")
    if coarse:
        from microclimf_amd.api import runmicro2Cpp_coarse
        a, rp, cp = synthetic.coarse_workload(rows, cols, T, 8, 8, reqhgt=0.05, out=out)
        w, _, _ = synthetic.coarse_workload(64, 64, 48, 8, 8, reqhgt=0.05, out=out)
        fn = runmicro2Cpp_coarse
        for d in (a, w):
            d["lats"], d["lons"] = d.pop("lat"), d.pop("lon")
    fn(**w)                                    # warm up
    r = None                                   # (freeing a previous 10 GB result is not part of the call)
    t0 = time.perf_counter()
    r = fn(**a)
    dt = time.perf_counter() - t0
    nbytes = sum(v.nbytes for v in r.values())
    valid = int((~np.isnan(a["vegp"]["hgt"])).sum())
    if af:
        nin = sum(np.asarray(v).nbytes for v in list(a["climdata"].values()) + list(a["pointm"].values()))
        print(f"  array climate: {nin / 1e9:.2f} GB of forcing uploaded")
    print(f"{rows}x{cols}x{T} outputs={sum(out)}: {dt:.3f} s, {valid * T / dt:.3e} cell-steps/s end to end, "
          f"{nbytes / dt / 1e9:.1f} GB/s of output into host memory")
