"""Timing and long-series parity of the snow branch on one GPU (not part of bench.py's contract):
  python tools/snow_rate.py [--rows R --cols C --tsteps T --array-forcing --check]
Prints the device rate of k_snowmodel (MCF_TIMING line on stderr), the end-to-end rate through
the host-pointer C ABI and, with --check, the worst scaled error against the CPU oracle."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("MCF_TIMING", "1")

from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.snow import gridmicrosnow1, gridmicrosnow2, gridmodelsnow1, gridmodelsnow2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=512)
    ap.add_argument("--cols", type=int, default=512)
    ap.add_argument("--tsteps", type=int, default=120)
    ap.add_argument("--array-forcing", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--reqhgt", type=float, default=0.05)
    ap.add_argument("--driver", action="store_true", help="time mcf_snowmodel1 (the 5-day chunk loop) instead")
    a = ap.parse_args()
    af = a.array_forcing
    if a.driver:
        return driver(a)
    sw = synthetic.snow_workload(a.rows, a.cols, a.tsteps, array_forcing=af, cold=3.0, zref=3.5)
    fn = gridmodelsnow2 if af else gridmodelsnow1
    args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"])
    fn(*args)                                   # warm-up (module load)
    t = time.time()
    r = fn(*args)
    dt = time.time() - t
    n = a.rows * a.cols * a.tsteps
    print(f"gridmodelsnow{2 if af else 1}: {a.rows}x{a.cols}x{a.tsteps} end-to-end {dt:.3f} s = {n / dt:.3e} cell-steps/s")
    snowm, micro = synthetic.microsnow_inputs(sw, r)
    mfn = gridmicrosnow2 if af else gridmicrosnow1
    margs = (a.reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0, [1] * 10)
    t = time.time()
    mo = mfn(*margs)
    dt = time.time() - t
    print(f"gridmicrosnow{2 if af else 1}: end-to-end {dt:.3f} s = {n / dt:.3e} cell-steps/s")
    if a.check:
        from oracle import oracle as O
        t = time.time()
        w = O.run_snowmodel(**sw, array_forcing=af)
        dtc = time.time() - t
        print(f"oracle gridmodelsnow: {dtc:.3f} s = {n / dtc:.3e} cell-steps/s (1 core)")
        worst = 0.0
        for k in ("Tc", "Tg", "sdepc", "sdepg", "sden", "agec", "ageg"):
            g, ww = r[k], w[k]
            assert np.array_equal(np.isnan(g), np.isnan(ww)), k
            f = np.isfinite(ww)
            e = float(np.max(np.abs(g[f] - ww[f]) / (1 + np.abs(ww[f])))) if f.any() else 0.0
            print(f"  {k}: max scaled error {e:.3e}")
            worst = max(worst, e)
        t = time.time()
        wm = O.run_microsnow(*margs, array_forcing=af)
        dtc = time.time() - t
        print(f"oracle gridmicrosnow: {dtc:.3f} s = {n / dtc:.3e} cell-steps/s (1 core)")
        for k in wm:
            g, ww = mo[k], wm[k]
            assert np.array_equal(np.isnan(g), np.isnan(ww)), k
            f = np.isfinite(ww)
            e = float(np.max(np.abs(g[f] - ww[f]) / (1 + np.abs(ww[f])))) if f.any() else 0.0
            print(f"  micro {k}: max scaled error {e:.3e}")
            worst = max(worst, e)
        print(f"worst scaled error {worst:.3e}")


def driver(a):
    from microclimf_amd.snow import snowmodel1_chunks
    sw = synthetic.snow_workload(a.rows, a.cols, a.tsteps, cold=3.0, zref=3.5)
    _, _, dtm = synthetic.rasters(a.rows, a.cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
    snowmodel1_chunks(*args)
    t = time.time()
    r = snowmodel1_chunks(*args)
    dt = time.time() - t
    n = a.rows * a.cols * a.tsteps
    print(f"snowmodel1 chunk loop: {a.rows}x{a.cols}x{a.tsteps} end-to-end {dt:.3f} s = {n / dt:.3e} cell-steps/s")
    if a.check:
        from oracle import snowdriver_oracle as SD
        t = time.time()
        w = SD.snowmodel1_chunks(*args)
        print(f"oracle chunk loop: {time.time() - t:.3f} s (numpy terrain + C snow model, 1 core)")
        for k in w:
            assert np.array_equal(np.isnan(r[k]), np.isnan(w[k])), k
            f = np.isfinite(w[k])
            print(f"  {k}: max scaled error {float(np.max(np.abs(r[k][f] - w[k][f]) / (1 + np.abs(w[k][f])))):.3e}")


if __name__ == "__main__":
    main()
