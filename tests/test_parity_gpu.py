"""GPU parity tests: libmcfhip (through the C ABI) against the CPU oracle on the
same seeded inputs.  Acceptance bar from BASELINE.json's north_star: 1e-4 degC /
1e-4 relative; the tests assert TOL = 1e-6 * (1 + |x|), two orders tighter."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import Plan, runmicro1Cpp, runmicro2Cpp

pytestmark = pytest.mark.gpu
TOL = 1e-6
NA_BITS = 0x7FF00000000007A2


def compare(got, want, tol=TOL):
    assert list(got) == list(want)            # same variables, same (reference) order
    worst = {}
    for k, w in want.items():
        g = got[k]
        assert g.shape == w.shape, k
        assert np.array_equal(np.isnan(g), np.isnan(w)), f"{k}: NA pattern differs"
        fin = np.isfinite(w)
        assert np.array_equal(np.isfinite(g), fin), f"{k}: inf pattern differs"
        err = np.abs(g[fin] - w[fin]) / (1.0 + np.abs(w[fin]))
        worst[k] = float(err.max()) if err.size else 0.0
        assert worst[k] <= tol, f"{k}: max scaled error {worst[k]:.3e}"
        na = np.isnan(w) & (w.view(np.uint64) == NA_BITS)
        if na.any():                          # NA cells / steps carry R's NA_real_ payload, not just any NaN
            assert (g[na].view(np.uint64) == NA_BITS).all(), k
    return worst


from parity_cases import CASES, build, with_na  # noqa: E402,F401  (re-exported for the other GPU tests)


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_oracle(oracle, name):
    """every workload of tests/parity_cases.py (which together flip every reachable branch of the
    path, tests/test_branch_coverage_cpu.py) through the C ABI against the oracle"""
    a, af = build(name)
    want = oracle.run_grid(**a, array_forcing=af)
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runmicro2Cpp(**a)
    else:
        got = runmicro1Cpp(**a)
    compare(got, want)


@pytest.mark.parametrize("cpb", [16, 21, 32, 42])
@pytest.mark.parametrize("name", ["below_canopy", "mixed_canopy", "ground", "soil_5cm"])
def test_workgroup_geometries(oracle, name, cpb):
    a, af = build(name)
    compare(runmicro1Cpp(**a, cells_per_block=cpb), oracle.run_grid(**a))


def test_out_mask_returns_only_requested(oracle):
    """only requested variables come back, in the reference's order; steps past the last whole day
    stay NA (cpp:2116)"""
    a, _ = build("partial_day_mask")
    got = runmicro1Cpp(**a)
    assert list(got) == ["Tz", "relhum", "windspeed", "Rlwdown"]
    assert np.isnan(got["Tz"][:, :, 48:]).all() and np.isfinite(got["Tz"][1, 1, :48]).all()


def test_chunking_is_bitwise_invariant():
    a = synthetic.workload(32, 9, 24 * 7, reqhgt=0.05, variety=True, start_doy=150)
    r1 = runmicro1Cpp(**a, days_per_chunk=7)
    r2 = runmicro1Cpp(**a, days_per_chunk=2)
    r3 = runmicro1Cpp(**a, days_per_chunk=3, cells_per_block=32)
    r4 = runmicro1Cpp(**a, days_per_chunk=4, cells_per_block=42)
    for k in r1:
        assert np.array_equal(r1[k], r2[k], equal_nan=True), k
        assert np.array_equal(r1[k], r3[k], equal_nan=True), k
        assert np.array_equal(r1[k], r4[k], equal_nan=True), k


def test_plan_ring_and_twi_mean(oracle):
    """the plan API: device-resident ring, partial twi reduction reinstalled (multi-GPU path)."""
    a = synthetic.workload(16, 16, 96, reqhgt=0.05, variety=True, start_doy=190)
    want = oracle.run_grid(**a)
    with Plan(**a, ring_days=2, ring_slots=2) as p:
        s, n = p.twi_partial()
        assert n == 256
        assert s == pytest.approx(np.sum(np.log(a["soilc"]["twi"]) / a["tfact"]), rel=1e-12)
        p.set_twi_mean(s / n)
        p.run_days(0, 2, 0)
        p.run_days(2, 2, 1)
        p.sync()
        tz = np.concatenate([p.fetch(0, "Tz", 0, 48), p.fetch(1, "Tz", 0, 48)], axis=2)
    np.testing.assert_allclose(tz, want["Tz"], rtol=0, atol=1e-6)


def test_large_grid_properties():
    """size-independent properties at a size the oracle cannot sweep: NA pattern, bounds,
    and equality of a sub-tile solved alone with the same tile solved inside the large grid
    (cells are independent given the raster-wide twi mean)."""
    R, Cn, T = 512, 256, 48
    a = synthetic.workload(R, Cn, T, reqhgt=0.05, start_doy=172)
    with Plan(**a, ring_days=2, ring_slots=1) as big:
        s, n = big.twi_partial()
        big.run_days(0, 2, 0)
        tz_big = big.fetch(0, "Tz", 0, T)
        rh_big = big.fetch(0, "relhum", 0, T)
    na = np.isnan(a["vegp"]["hgt"])
    assert np.array_equal(np.isnan(tz_big[:, :, 0]), na)
    assert np.nanmax(rh_big) <= 100.0 and np.nanmin(rh_big) > 0
    assert np.nanmax(np.abs(tz_big - a["climdata"]["temp"][None, None, :])) < 50   # the reference caps dT at -0.6273*mxtc + 49.79 (cpp:1236)
    sub = synthetic.workload(64, Cn, T, reqhgt=0.05, start_doy=172, row0=128, rows_total=R)
    for k in sub["vegp"]:
        assert np.array_equal(sub["vegp"][k], a["vegp"][k][128:192], equal_nan=True)
    with Plan(**sub, ring_days=2, ring_slots=1) as small:
        small.set_twi_mean(s / n)
        small.run_days(0, 2, 0)
        tz_small = small.fetch(0, "Tz", 0, T)
    assert np.array_equal(tz_small, tz_big[128:192], equal_nan=True)
