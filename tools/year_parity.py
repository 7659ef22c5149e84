"""A whole year value by value: the HIP path against the oracle on 96 x 96 cells x 8760 h (8 x 10^7 cell-steps, all ten
outputs; the oracle needs about a minute on one core).  python tools/year_parity.py [--rows 96 --cols 96]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import runmicro1Cpp  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=96)
ap.add_argument("--cols", type=int, default=96)
ap.add_argument("--reqhgt", type=float, default=0.05)
ap.add_argument("--coarse", type=str, default="", help="CRxCC: coarse array forcing against expand-then-solve through the oracle")
ap.add_argument("--array", action="store_true", help="array forcing (runmicro2Cpp's geometry): every forcing value per cell")
a = ap.parse_args()
if a.coarse:
    from microclimf_amd.api import runmicro2Cpp_coarse
    from oracle import coarse_oracle as CO
    cr, cc = (int(v) for v in a.coarse.split("x"))
    w, rp, cp = synthetic.coarse_workload(a.rows, a.cols, 8760, cr, cc, reqhgt=a.reqhgt, variety=True, na_frac=0.02)
    order = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact", "complete",
             "mat", "out")
    t0 = time.perf_counter()
    got = runmicro2Cpp_coarse(*[w[k] for k in order], rowpos=rp, colpos=cp)
    t1 = time.perf_counter()
    clim, pm = CO.expand(w["climdata"], w["pointm"], rp, cp)
    b = dict(w)
    b.update(climdata=clim, pointm=pm)
    want = O.run_grid(**b, array_forcing=True)
    t2 = time.perf_counter()
elif a.array:
    from microclimf_amd.api import runmicro2Cpp
    w = synthetic.workload(a.rows, a.cols, 8760, reqhgt=a.reqhgt, variety=True, na_frac=0.02, array_forcing=True)
    order = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact", "complete",
             "mat", "out")
    t0 = time.perf_counter()
    got = runmicro2Cpp(*[w[k] for k in order])
    t1 = time.perf_counter()
    want = O.run_grid(**w, array_forcing=True)
    t2 = time.perf_counter()
else:
    w = synthetic.workload(a.rows, a.cols, 8760, reqhgt=a.reqhgt, variety=True, na_frac=0.02)
    t0 = time.perf_counter()
    got = runmicro1Cpp(**w)
    t1 = time.perf_counter()
    want = O.run_grid(**w)
    t2 = time.perf_counter()
print(f"{a.rows} x {a.cols} x 8760, reqhgt {a.reqhgt}{', coarse ' + a.coarse if a.coarse else ', array forcing' if a.array else ''}: HIP one-shot {t1 - t0:.2f} s, oracle {t2 - t1:.1f} s")
worst = 0.0
for k, x in want.items():
    g = got[k]
    assert np.array_equal(np.isnan(g), np.isnan(x)), k
    fin = np.isfinite(x)
    e = float((np.abs(g[fin] - x[fin]) / (1 + np.abs(x[fin]))).max()) if fin.any() else 0.0
    worst = max(worst, e)
    print(f"  {k:10s} max scaled |HIP - oracle| = {e:.3e} over {int(fin.sum())} values")
print(f"worst {worst:.3e}")
assert worst < 1e-6
