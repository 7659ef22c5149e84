"""Seeded random configurations of the one-shot solver entries against the oracle: raster shape (incl. single rows /
columns and sizes that do not fill a workgroup), series length (whole days + a ragged tail), height class, season,
latitude, cold spells, NA share, output mask, forcing geometry, workgroup geometry, day chunking — the cross product the
hand-written cases of tests/parity_cases.py sample only along its axes."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import runmicro1Cpp, runmicro2Cpp
from test_parity_gpu import compare

pytestmark = pytest.mark.gpu


def draw(i):
    rng = np.random.default_rng(7000 + i)
    rows, cols = (int(rng.integers(1, 40)), int(rng.integers(1, 40)))
    if i % 7 == 0:
        rows = 1
    if i % 11 == 0:
        cols = 1
    days = int(rng.integers(1, 5))
    tail = int(rng.choice([0, 0, 5, 23]))
    reqhgt = float(rng.choice([0.02, 0.05, 0.4, 1.0, 1.9, 0.0, -0.03, -0.2]))
    af = bool(rng.random() < 0.35)
    out = [bool(b) for b in rng.random(10) < 0.6]
    if not any(out):
        out[0] = True
    kw = dict(reqhgt=reqhgt, start_doy=int(rng.integers(1, 360)), variety=bool(rng.random() < 0.7),
              cold=float(rng.choice([0.0, 0.0, 12.0])), hgt_range=(0.05, float(rng.choice([0.3, 1.5, 3.0]))),
              lat=float(rng.choice([-35.0, 5.0, 50.0, 68.0])), lon=float(rng.choice([-5.0, 100.0])),
              na_frac=float(rng.choice([0.0, 0.02, 0.3])), out=out, array_forcing=af,
              complete=bool(rng.random() < 0.5) if reqhgt < 0 else True, seed=int(rng.integers(1, 1 << 30)))
    if kw["hgt_range"][1] > 1.9:
        kw["zref"] = 3.5                      # the model needs the reference height above the canopy
    extra = dict(cells_per_block=int(rng.choice([0, 16, 21, 32, 42])), days_per_chunk=int(rng.choice([0, 1, 2])))
    if reqhgt < 0:
        extra["days_per_chunk"] = 0          # the below-ground smoother needs the whole series in one slot
    return rows, cols, days * 24 + tail, kw, extra


@pytest.mark.parametrize("i", range(96))
def test_random_configuration(oracle, i):
    rows, cols, T, kw, extra = draw(i)
    a = synthetic.workload(rows, cols, T, **kw)
    af = kw["array_forcing"]
    want = oracle.run_grid(**a, array_forcing=af)
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runmicro2Cpp(**a, **extra)
    else:
        got = runmicro1Cpp(**a, **extra)
    compare(got, want)


@pytest.mark.parametrize("af,reqhgt", [(False, 0.05), (True, 0.6), (False, -0.05)])
def test_medium_raster_against_the_oracle(oracle, af, reqhgt):
    """~900 workgroups: more than one XCD round and both hour rotations of the tile schedule (blockIdx >> 8), which the
    small cases above never reach, compared value by value"""
    a = synthetic.workload(160, 120, 72, reqhgt=reqhgt, variety=True, start_doy=200, array_forcing=af, na_frac=0.03)
    want = oracle.run_grid(**a, array_forcing=af)
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runmicro2Cpp(**a)
    else:
        got = runmicro1Cpp(**a)
    compare(got, want)
