"""Where the solver's time goes, by input-side ablation (no code changes): the same 1024 x 1024 x 5-day launch with
inputs that switch whole sections of the cell-step off.  python tools/ablation.py [--rows 1024 --cols 1024]"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--cols", type=int, default=1024)
a = ap.parse_args()
DAYS = 5


def run(label, reqhgt=0.05, out=None, tweak=None, start_doy=152):
    w = synthetic.workload(a.rows, a.cols, DAYS * 24 * 2, reqhgt=reqhgt, start_doy=start_doy, out=out)
    if tweak:
        tweak(w)
    with Plan(**w, ring_days=DAYS, ring_slots=1) as p:
        p.run_days(0, DAYS)
        p.sync()
        p.kernel_timing(True)
        for _ in range(4):
            p.run_days(DAYS, DAYS)
        p.sync()
        ms, n = p.kernel_stats()
        valid = p.valid_cells
    t = ms / n
    print(f"{label:72s} {t:7.3f} ms  {valid * DAYS * 24 / (t * 1e-3):9.3e} cell-steps/s")
    return t


def night(w):
    w["climdata"]["swdown"][:] = 0.0
    w["climdata"]["difrad"][:] = 0.0


def bare(w):
    for k in ("pai", "hgt", "paia"):
        w["vegp"][k][~np.isnan(w["vegp"]["hgt"])] = 0.0


base = run("all ten outputs, below canopy, June (baseline)")
run("above canopy (reqhgt 1.6 m): no leaf / Lagrangian block", reqhgt=1.6)
run("ground surface (reqhgt 0): Tz from the soil model, no TVabove for Tz", reqhgt=0.0)
run("polar night (swdown = 0): no short-wave block, stomata shut", tweak=night)
run("December (short days)", start_doy=345)
run("bare ground everywhere (pai = hgt = 0)", tweak=bare)
run("pass 1 only (soilm, windspeed, short-wave outputs)", out=[0, 0, 0, 1, 1, 1, 1, 0, 1, 0])
run("Tz only", out=[1] + [0] * 9)
run("Tz + soilm (the bioclim pair)", out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0])
