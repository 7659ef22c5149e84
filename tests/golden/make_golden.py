"""Generates tests/golden/*.npz: small seeded inputs + the ORACLE's outputs.

These are regression vectors produced by oracle/mcf_oracle.c (the CPU restatement),
NOT outputs of the reference itself: the reference cannot be built or run in this
image (it needs R/Rcpp).  They freeze the oracle so that later edits to it, or to the
HIP kernels, are caught by value; they do not add reference pinning.

Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from microclimf_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (workload kwargs, array_forcing)
    "vec_below_canopy": (dict(rows=8, cols=6, tsteps=72, reqhgt=0.05, variety=True, start_doy=170), False),
    "vec_above_canopy": (dict(rows=8, cols=6, tsteps=48, reqhgt=5.0, zref=10.0, hgt_range=(0.5, 9.0),
                              variety=True, start_doy=170), False),
    "vec_ground": (dict(rows=8, cols=6, tsteps=48, reqhgt=0.0, variety=True, start_doy=20, cold=10.0), False),
    "vec_soil": (dict(rows=8, cols=6, tsteps=96, reqhgt=-0.1, variety=True, start_doy=100,
                      out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False),
    "arr_below_canopy": (dict(rows=6, cols=5, tsteps=48, reqhgt=0.05, variety=True, start_doy=170,
                              array_forcing=True), True),
}


def build(name):
    kw, af = CASES[name]
    a = synthetic.workload(**kw)
    a["vegp"]["hgt"][0, 0] = np.nan
    return a, af


def flatten(prefix, d, out):
    for k, v in d.items():
        out[f"{prefix}.{k}"] = np.asarray(v)


if __name__ == "__main__":
    here = Path(__file__).resolve().parent
    for name in CASES:
        a, af = build(name)
        res = O.run_grid(**a, array_forcing=af)
        blob = {}
        for grp in ("obstime", "climdata", "pointm", "vegp", "soilc"):
            flatten(grp, a[grp], blob)
        for k in ("reqhgt", "zref", "Sminp", "Smaxp", "tfact", "complete", "mat", "out", "lat", "lon"):
            blob[f"arg.{k}"] = np.asarray(a[k])
        for k, v in res.items():
            blob[f"expect.{k}"] = v
        np.savez_compressed(here / f"{name}.npz", **blob)
        print(name, {k: v.shape for k, v in res.items()})
