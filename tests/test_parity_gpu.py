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
        na = np.isnan(g)
        if na.any():                          # R's NA_real_ payload, not just any NaN
            assert (g[na].view(np.uint64) == NA_BITS).all(), k
    return worst


def with_na(a, frac_cells=((0, 0), (3, 2))):
    for (i, j) in frac_cells:
        a["vegp"]["hgt"][i, j] = np.nan
    return a


@pytest.mark.parametrize("reqhgt,zref,hgt_range", [
    (0.05, 2.0, (0.05, 1.5)),      # below canopy for ~all cells
    (1.0, 2.0, (0.05, 1.9)),       # mixed above / below canopy
    (5.0, 10.0, (0.5, 9.0)),       # above canopy, shrub + tree stomatal classes
    (0.0, 2.0, (0.05, 1.5)),       # ground surface
])
@pytest.mark.parametrize("cpb", [16, 21, 32, 42])
def test_runmicro1_matches_oracle(oracle, reqhgt, zref, hgt_range, cpb):
    a = with_na(synthetic.workload(21, 13, 96, reqhgt=reqhgt, zref=zref, hgt_range=hgt_range,
                                   variety=True, start_doy=170))
    want = oracle.run_grid(**a)
    got = runmicro1Cpp(**a, cells_per_block=cpb)
    compare(got, want)


@pytest.mark.parametrize("start_doy,cold", [(1, 12.0), (355, 4.0), (80, 0.0)])
def test_runmicro1_cold_and_low_sun(oracle, start_doy, cold):
    """sub-zero branches of satvap / latent heat, long nights, low sun (large Rbeam)."""
    a = with_na(synthetic.workload(17, 9, 72, reqhgt=0.05, variety=True, start_doy=start_doy, cold=cold))
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))


def test_runmicro1_tropical_latitude(oracle):
    """near-zenith sun exercises the un-saturated branch of the cankCpp degrees call (cpp:1425)
    and the C4 / tropical stomatal classes (cpp:399, 415)."""
    a = synthetic.workload(16, 8, 48, reqhgt=0.05, zref=12.0, hgt_range=(0.2, 11.0), variety=True,
                           start_doy=80, lat=0.5, lon=0.0)
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))


@pytest.mark.parametrize("reqhgt", [-0.05, -0.4, -3.0, -40.0])
@pytest.mark.parametrize("complete", [True, False])
def test_runmicro1_below_ground(oracle, reqhgt, complete):
    """rolling-mean window <= 48 h, daily route, and n >= tsteps (cpp:1483-1492); the blends
    of the incomplete route (cpp:1495-1536)."""
    a = with_na(synthetic.workload(18, 7, 240, reqhgt=reqhgt, variety=True, start_doy=100,
                                   out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=complete))
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))


def test_out_mask_and_partial_day(oracle):
    """only requested variables come back; steps past the last whole day stay NA (cpp:2116)."""
    a = with_na(synthetic.workload(16, 5, 60, reqhgt=0.05, variety=True, start_doy=200,
                                   out=[1, 0, 1, 0, 1, 0, 0, 1, 0, 0]))
    got = runmicro1Cpp(**a)
    assert list(got) == ["Tz", "relhum", "windspeed", "Rlwdown"]
    assert np.isnan(got["Tz"][:, :, 48:]).all() and np.isfinite(got["Tz"][1, 1, :48]).all()
    compare(got, oracle.run_grid(**a))


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


@pytest.mark.parametrize("reqhgt", [0.05, 0.0, -0.2])
def test_runmicro2_matches_oracle(oracle, reqhgt):
    out = [1, 0, 0, 1, 0, 0, 0, 0, 0, 0] if reqhgt < 0 else [1] * 10
    a = with_na(synthetic.workload(19, 6, 72, reqhgt=reqhgt, variety=True, start_doy=170,
                                   array_forcing=True, out=out))
    want = oracle.run_grid(**a, array_forcing=True)
    a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
    compare(runmicro2Cpp(**a), want)


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
