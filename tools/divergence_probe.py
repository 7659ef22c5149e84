#!/usr/bin/env python3
"""How much of k_solve's time is the canopy-class divergence inside a tile?  The synthetic raster draws vegetation per cell
(5 % bare cells, i.i.d.), so two thirds of the 21-cell tiles hold both classes and their waves run the above- and the
below-canopy code.  Same launch on rasters without bare cells / with the bare cells gathered into whole tiles."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

R = C = 1024
T = 480


def run(label, mutate=None):
    w = synthetic.workload(R, C, T, reqhgt=0.05)
    if mutate:
        mutate(w)
    with Plan(w["obstime"], w["climdata"], w["pointm"], w["vegp"], w["soilc"], w["reqhgt"], w["zref"], w["lat"], w["lon"],
              w["Sminp"], w["Smaxp"], w["tfact"], True, w["mat"], w["out"], ring_days=10) as p:
        p.run_days(0, 10)
        p.sync()
        p.kernel_timing(True)
        for _ in range(3):
            p.run_days(0, 10)
            p.run_days(10, 10)
        p.sync()
        ms, n = p.kernel_stats()
        valid = p.valid_cells
    print(f"{label:50s} {ms / n:8.3f} ms per 10-day launch, {valid * 240 / (ms / n) * 1e3:.4e} cell-steps/s", flush=True)


def gather_bare(w):
    """the same cells, the bare ones moved to the front of the column-major order (whole tiles of one class)"""
    v = w["vegp"]
    hgt = v["hgt"]
    order = np.argsort(~(hgt.ravel(order="F") == 0.0), kind="stable")
    for d in (w["vegp"], w["soilc"]):
        for k, a in d.items():
            a = np.asarray(a)
            if a.ndim == 2:
                d[k] = np.asfortranarray(a.ravel(order="F")[order].reshape(a.shape, order="F"))
            elif a.ndim == 3 and a.shape[:2] == hgt.shape:
                d[k] = np.asfortranarray(np.stack([a[:, :, i].ravel(order="F")[order].reshape(hgt.shape, order="F")
                                                   for i in range(a.shape[2])], axis=2))


def no_bare(w):
    v = w["vegp"]
    b = v["hgt"] == 0.0
    v["hgt"] = np.asfortranarray(np.where(b, 0.5, v["hgt"]))
    v["pai"] = np.asfortranarray(np.where(b, 2.0, v["pai"]))
    v["paia"] = np.asfortranarray(np.where(b, 1.4, v["paia"]))
    v["leafden"] = np.asfortranarray(np.where(b, 4.0, v["leafden"]))


def all_bare(w):
    v = w["vegp"]
    ok = ~np.isnan(v["hgt"])
    v["hgt"] = np.asfortranarray(np.where(ok, 0.0, v["hgt"]))
    v["pai"] = np.asfortranarray(np.where(ok, 0.0, v["pai"]))
    v["paia"] = np.asfortranarray(np.where(ok, 0.0, v["paia"]))
    v["leafden"] = np.asfortranarray(np.where(ok, np.nan, v["leafden"]))


run("shipped raster (5 % bare cells, i.i.d.)")
run("no bare cells", mutate=no_bare)
run("bare cells gathered into whole tiles", mutate=gather_bare)
run("every cell bare (above-canopy code only)", mutate=all_bare)
