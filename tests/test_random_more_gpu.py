"""Seeded random configurations of the remaining entries: time-varying vegetation (runmicro3/4Cpp), the terrain
pre-compute on odd raster shapes and block cuts, and the plan route (random chunk sizes, ring slots, fetch ranges) against
the one-shot route."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import Plan, runmicro1Cpp, runmicro3Cpp, runmicro4Cpp
from microclimf_amd.terrain import precompute_terrain
from oracle import terrain_oracle as TO
from test_parity_gpu import compare
from test_terrain_cpu import synth_dtm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(32))
def test_random_layered_vegetation(oracle, i):
    rng = np.random.default_rng(3300 + i)
    rows, cols, days = int(rng.integers(1, 30)), int(rng.integers(1, 30)), int(rng.integers(2, 8))
    layers = int(rng.integers(1, days + 1))
    af = bool(rng.random() < 0.35)
    reqhgt = float(rng.choice([0.05, 0.3, 1.5, 0.0, -0.1]))
    out = [1] + [int(b) for b in rng.random(9) < 0.5]
    a = synthetic.workload(rows, cols, days * 24 + int(rng.choice([0, 6])), reqhgt=reqhgt, variety=True,
                           start_doy=int(rng.integers(1, 350)), array_forcing=af, na_frac=float(rng.choice([0, 0.05])),
                           out=out, seed=int(rng.integers(1, 1 << 30)))
    cover = int(rng.integers(layers, days + 1))
    a = synthetic.layered(a, layers, cover_days=cover if reqhgt >= 0 else None)
    want = oracle.run_grid(**a, array_forcing=af)
    dfsel = a.pop("dfsel")
    chunk = int(rng.choice([0, 1, 3])) if reqhgt >= 0 else 0
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runmicro4Cpp(dfsel, **a, days_per_chunk=chunk)
    else:
        got = runmicro3Cpp(dfsel, **a, days_per_chunk=chunk)
    compare(got, want)


@pytest.mark.parametrize("i", range(16))
def test_random_terrain_shapes_and_blocks(i):
    rng = np.random.default_rng(4400 + i)
    rows, cols = int(rng.integers(3, 260)), int(rng.integers(3, 90))
    res = float(rng.choice([0.5, 1.0, 2.5, 30.0]))
    z = synth_dtm(rows, cols) * float(rng.choice([0.2, 1.0, 5.0]))
    for _ in range(int(rng.integers(0, 6))):                               # NA elevations
        z[int(rng.integers(0, rows)), int(rng.integers(0, cols))] = np.nan
    zref = float(rng.choice([2.0, 10.0]))
    want = TO.terrain(z, res, zref)
    got = precompute_terrain(z, res, zref)
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w, rtol=0, atol=2e-9, err_msg=k)     # steep (x5) relief amplifies rounding in tan()
    if rows > 40:                                                           # a row block (boundaries on multiples of 10)
        r0 = 10 * int(rng.integers(1, rows // 20 + 1))
        n = min(10 * int(rng.integers(1, 8)), rows - r0)
        hn, hs = min(128, r0), min(128, rows - r0 - n)
        blk = precompute_terrain(z[r0 - hn:r0 + n + hs], res, zref, halo_north=hn, halo_south=hs, row0=r0,
                                 rows_total=rows)
        for k, w in want.items():
            np.testing.assert_allclose(blk[k], w[r0:r0 + n], rtol=0, atol=2e-9, err_msg=f"block {k}")


@pytest.mark.parametrize("i", range(16))
def test_random_plan_schedules_equal_the_one_shot_call(i):
    """any way of cutting the series into ring slots gives the bits of the one-shot entry"""
    rng = np.random.default_rng(5500 + i)
    rows, cols, days = int(rng.integers(2, 40)), int(rng.integers(2, 40)), int(rng.integers(2, 9))
    out = [1] + [int(b) for b in rng.random(9) < 0.5]
    a = synthetic.workload(rows, cols, days * 24, reqhgt=float(rng.choice([0.05, 1.0, 0.0])), variety=True,
                           start_doy=int(rng.integers(1, 350)), out=out, seed=int(rng.integers(1, 1 << 30)))
    ref = runmicro1Cpp(**a)
    ring_days, slots = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    with Plan(**a, ring_days=ring_days, ring_slots=slots, cells_per_block=int(rng.choice([0, 16, 21, 32, 42]))) as p:
        slot = 0
        for d0 in range(0, days, ring_days):
            nd = min(ring_days, days - d0)
            p.run_days(d0, nd, slot)
            p.sync()
            for k in ref:
                s0 = int(rng.integers(0, nd * 24))
                n = int(rng.integers(1, nd * 24 - s0 + 1))
                got = p.fetch(slot, k, s0, n)
                want = ref[k][:, :, d0 * 24 + s0:d0 * 24 + s0 + n]
                assert np.array_equal(got.view(np.uint64), np.asfortranarray(want).view(np.uint64)), k   # bit for bit
            slot = (slot + 1) % slots


@pytest.mark.parametrize("i", range(12))
def test_random_bioclim_selection(oracle, i):
    from microclimf_amd.api import runbioclim1Cpp, runbioclim2Cpp
    rng = np.random.default_rng(6600 + i)
    nq = [int(rng.integers(1, 4)) * 24 for _ in range(4)]                    # quarter lengths (the divisor stays 72, cpp:3325)
    T = 336 + sum(nq)
    af = bool(rng.random() < 0.3)
    a = synthetic.workload(int(rng.integers(1, 16)), int(rng.integers(1, 16)), T, reqhgt=float(rng.choice([0.05, 1.0])),
                           variety=True, start_doy=int(rng.integers(1, 300)), array_forcing=af,
                           na_frac=float(rng.choice([0.0, 0.1])), seed=int(rng.integers(1, 1 << 30)))
    for k in ("complete", "out"):
        a.pop(k)
    edges = 336 + np.concatenate([[0], np.cumsum(nq)])
    q = [np.arange(edges[j], edges[j + 1]) for j in range(4)]
    out = [int(b) for b in rng.random(19) < 0.5]
    if not any(out):
        out[0] = 1
    air = bool(rng.random() < 0.5)
    kw = dict(out=out, wetq=q[0], dryq=q[1], hotq=q[2], colq=q[3], air=air)
    want = oracle.run_bioclim(**a, **kw, array_forcing=af)
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runbioclim2Cpp(**a, **kw)
    else:
        got = runbioclim1Cpp(**a, **kw)
    assert list(got) == [f"bio{j + 1}" for j in range(19) if out[j]]
    for k, w in want.items():
        assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
        np.testing.assert_allclose(got[k], w, rtol=1e-9, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("i", range(10))
def test_random_nc_files_device_route_equals_host_route(i, tmp_path):
    from microclimf_amd import ncsink
    rng = np.random.default_rng(7700 + i)
    rows, cols, days = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 5))
    reqhgt = float(rng.choice([0.05, 1.0, 0.0]))
    allowed = ncsink.DEFAULT_VARS_ABOVE + ("soilm",) if reqhgt > 0 else ncsink.DEFAULT_VARS_SURFACE
    names = tuple(v for v in allowed if rng.random() < 0.6) or ("Tz",)
    a = synthetic.workload(rows, cols, days * 24, reqhgt=reqhgt, variety=True, start_doy=int(rng.integers(1, 350)),
                           na_frac=float(rng.choice([0.0, 0.2])), seed=int(rng.integers(1, 1 << 30)))
    east, north = ncsink.coords_from_extent(0.0, cols * 3.0, -50.0, -50.0 + rows * 3.0, 3.0)
    hours = ncsink.hours_since_epoch(a["obstime"])
    puts = bool(rng.random() < 0.3)
    with Plan(**a, ring_days=days) as p:
        p.run_days(0, days)
        p.sync()
        with ncsink.NcWriter(tmp_path / "d.nc", rows, cols, hours, east, north, reqhgt, names, "x", puts) as w:
            cut = int(rng.integers(0, days * 24 + 1))
            w.write_plan(p, 0, cut, cut, days * 24 - cut)
            w.write_plan(p, 0, 0, 0, cut)
        with ncsink.NcWriter(tmp_path / "h.nc", rows, cols, hours, east, north, reqhgt, names, "x", puts) as w:
            w.write_host(0, {k: p.fetch(0, k, 0, days * 24) for k in names})
    assert (tmp_path / "d.nc").read_bytes() == (tmp_path / "h.nc").read_bytes()
