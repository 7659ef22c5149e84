"""The host-side point model of libmcfhip (mcf_pointmodel.cpp) against the oracle's restatement
(oracle/pointmodel.c) on the inputs of the reference's own tests and on the synthetic weather.  Both are host code;
no GPU is involved.  Same algorithm in the same evaluation order: agreement to rounding (1e-11 asserted)."""
import ctypes as C

import numpy as np
import pytest

from microclimf_amd import pointmodel as PM
from microclimf_amd import synthetic
from oracle import pointchain
from oracle import replay_reference_tests as RT

DP = C.POINTER(C.c_double)


def _test_inputs():
    hrs, n, obst, Tair, RH, Pk = RT._forcing(2024, 3, 21, True)
    SWd = np.maximum(0, 600 * np.sin((hrs - 6) / 12 * np.pi))
    clim = {"temp": Tair, "relhum": RH, "pres": Pk, "swdown": SWd, "difrad": np.minimum(SWd, 0.3 * SWd),
            "lwdown": np.full(n, 350.0), "windspeed": np.full(n, 2.0), "precip": np.zeros(n)}
    return obst, clim, n


def test_bigleaf_equals_oracle_on_the_reference_test_inputs(oracle):
    obst, clim, n = _test_inputs()
    vegp = np.array([0.5, 2.0, 1.0, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100])
    groundp = np.array([0.15, 0, 180, 0.97, 1.53, 0.509, 0.06, 0.5422, 5.2, -5.6, 0.42, 0.074])
    want = RT.bigleaf(obst, clim, vegp, groundp, np.full(n, 0.3), 50.0, -5.0, 25, 2, 50, 0.5, 0.5, 0.1, False)
    got = PM.BigLeafCpp(obst, clim, vegp, groundp, np.full(n, 0.3), 50.0, -5.0, 25, 2, 50, 0.5, 0.5, 0.1, False)
    assert got["iters"] == want["iters"]
    assert got["err"] == pytest.approx(want["err"], abs=1e-11)
    for k in ("Tc", "Tg", "H", "G", "psih", "psim", "phih", "OL", "uf", "RabsG", "albedo"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-11, atol=1e-11, err_msg=k)
    # and the reference's own bounds (test-BigLeafCpp.R) hold for the product as they do for the oracle
    assert got["err"] < 0.5 and got["Tc"].max() <= clim["temp"].max() + 10


def test_weatherhgt_and_soilm_equal_oracle(oracle):
    lib = oracle.load()
    obst, clim, n = _test_inputs()
    got = PM.weatherhgtCpp(obst, clim, 2, 2, 10, 50, -5)
    Tz, Rh, Uz = np.zeros(n), np.zeros(n), np.zeros(n)
    d = lambda a: a.ctypes.data_as(DP)                                                  # noqa: E731
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))                                  # noqa: E731
    lib.orc_weatherhgt(C.c_int(n), i(obst["year"]), i(obst["month"]), i(obst["day"]), d(obst["hour"]), d(clim["temp"]),
                       d(clim["relhum"]), d(clim["pres"]), d(clim["swdown"]), d(clim["difrad"]), d(clim["lwdown"]),
                       d(clim["windspeed"]), C.c_double(2.0), C.c_double(2.0), C.c_double(10.0), C.c_double(50.0),
                       C.c_double(-5.0), d(Tz), d(Rh), d(Uz))
    np.testing.assert_allclose(got["temp"], Tz, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(got["relhum"], Rh, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(got["windspeed"], Uz, rtol=1e-11, atol=1e-11)
    ratio = got["windspeed"] / clim["windspeed"]
    assert 1.2 <= ratio.min() and ratio.max() <= 1.4                                   # test-weatherhgtCpp.R
    assert np.array_equal(got["swdown"], clim["swdown"]) and got["swdown"] is not clim["swdown"]
    # soilmCpp on two days (test-soilmCpp.R's call)
    c2 = {k: np.concatenate([v, v]) for k, v in clim.items()}
    p = pointchain.SOILM_PARAMS
    sm = PM.soilmCpp(c2, p["rmu"], p["mult"], p["pwr"], p["Smax"], p["Smin"], p["Ksat"], p["a"])
    want = np.zeros(2)
    lib.orc_soilm.restype = C.c_int
    lib.orc_soilm(C.c_int(2 * n), d(c2["temp"]), d(c2["swdown"]), d(c2["lwdown"]), d(c2["precip"]), C.c_double(p["rmu"]),
                  C.c_double(p["mult"]), C.c_double(p["pwr"]), C.c_double(p["Smax"]), C.c_double(p["Smin"]),
                  C.c_double(p["Ksat"]), C.c_double(p["a"]), d(want))
    assert len(sm) == 2 and np.allclose(sm, want, rtol=1e-13) and 0.35 <= sm.min() and sm.max() <= 0.419


def test_chain_to_pointm_equals_oracle_chain(oracle):
    """weather -> soilmCpp -> BigLeafCpp -> pointmprocess -> pointm, product against oracle/pointchain.py"""
    a = synthetic.workload(4, 4, 240, reqhgt=0.05, start_doy=140)
    c = a["climdata"]
    T = len(c["temp"])
    weather = {"temp": c["temp"], "relhum": 100 * c["ea"] / c["es"], "pres": c["pres"], "swdown": c["swdown"],
               "difrad": c["difrad"], "lwdown": c["lwdown"], "windspeed": c["windspeed"],
               "precip": np.where(np.arange(T) % 17 == 0, 1.0, 0.0)}
    want, werr = pointchain.pointm_chain(a["obstime"], weather, a["lat"], a["lon"], a["zref"])
    got = PM.runpointmodel_chain(a["obstime"], weather, pointchain.VEGP_P, pointchain.GROUNDP_P, a["lat"], a["lon"],
                                 a["zref"], soilparams=pointchain.SOILM_PARAMS, yearG=False)
    assert got["bigleaf"]["err"] == pytest.approx(werr, abs=1e-10)
    for k, w in want.items():
        np.testing.assert_allclose(got["pointm"][k], w, rtol=1e-10, atol=1e-10, err_msg=k)


def test_guards_against_the_reference_out_of_bounds_means():
    from microclimf_amd import _abi
    obst, clim, n = _test_inputs()
    vegp = np.array([0.5, 2.0, 1.0, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100])
    groundp = np.array([0.15, 0, 180, 0.97, 1.53, 0.509, 0.06, 0.5422, 5.2, -5.6, 0.42, 0.074])
    ob5 = {k: np.tile(v, 5) for k, v in obst.items()}
    cl5 = {k: np.tile(v, 5) for k, v in clim.items()}
    with pytest.raises(_abi.McfError, match="yearG"):
        PM.BigLeafCpp(ob5, cl5, vegp, groundp, np.full(5 * n, 0.3), 50.0, -5.0, yearG=True)      # 5 days: 91-day mean
    r = PM.BigLeafCpp(obst, clim, vegp, groundp, np.full(n, 0.3), 50.0, -5.0, yearG=True)       # one day is fine
    assert np.isfinite(r["Tc"]).all()


@pytest.mark.parametrize("i", range(12))
def test_random_point_model_inputs_equal_oracle(oracle, i):
    """random sites, canopies, soils and weather (1 - 6 days, and whole-year-like >= 90-day series with yearG)"""
    rng = np.random.default_rng(8800 + i)
    days = int(rng.choice([1, 2, 3, 6])) if i % 4 else 95
    yearG = days >= 90 or days == 1
    a = synthetic.workload(2, 2, days * 24, reqhgt=0.05, start_doy=int(rng.integers(1, 250)),
                           lat=float(rng.choice([-40.0, 10.0, 50.0, 65.0])), lon=float(rng.choice([-5.0, 120.0])),
                           cold=float(rng.choice([0.0, 10.0])), seed=int(rng.integers(1, 1 << 30)))
    c = a["climdata"]
    n = days * 24
    clim = {"temp": c["temp"], "relhum": np.clip(100 * c["ea"] / c["es"], 5, 100), "pres": c["pres"], "swdown": c["swdown"],
            "difrad": c["difrad"], "lwdown": c["lwdown"], "windspeed": np.maximum(c["windspeed"], 0.5),
            "precip": np.where(rng.random(n) < 0.1, rng.uniform(0, 5, n), 0.0)}
    hgt = float(rng.uniform(0.1, 1.8))
    vegp = np.array([hgt, rng.uniform(0.2, 4), rng.uniform(0.5, 2), rng.uniform(0, 0.5), rng.uniform(0.3, 0.45),
                     rng.uniform(0.1, 0.25), rng.uniform(0.01, 0.1), 0.97, rng.uniform(0.2, 0.4), 100.0])
    groundp = np.array([rng.uniform(0.1, 0.2), rng.uniform(0, 20), rng.uniform(0, 360), 0.97, 1.53, 0.509, 0.06, 0.5422,
                        5.2, -5.6, 0.42, 0.074])
    soilm = rng.uniform(0.1, 0.4, n)
    zref = float(hgt + rng.uniform(0.3, 2.0))
    args = (a["obstime"], clim, vegp, groundp, soilm, a["lat"], a["lon"], 25.0, zref, int(rng.choice([20, 100])), 0.5, 0.5,
            0.1, yearG)
    want = RT.bigleaf(*args)
    got = PM.BigLeafCpp(*args)
    assert got["iters"] == want["iters"]
    for k in ("Tc", "Tg", "H", "G", "psih", "psim", "phih", "OL", "uf", "RabsG", "albedo"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-10, atol=1e-10, err_msg=k)
    # height adjustment of the same weather
    wh = PM.weatherhgtCpp(a["obstime"], clim, zref, zref, zref + 8.0, a["lat"], a["lon"])
    assert np.isfinite(wh["temp"]).all() and (wh["windspeed"] >= clim["windspeed"] - 1e-12).all()


def _snow_test_inputs(oracle):
    """the inputs of tests/testthat/test-pointmodelsnow.R as oracle/replay_reference_tests.py rebuilds them"""
    RT.replay_pointmodelsnow_test()
    lib = oracle.load()
    hrs = np.arange(24.0)
    n = 24
    obst = {"year": np.full(n, 2024, dtype=np.int32), "month": np.full(n, 3, dtype=np.int32),
            "day": np.full(n, 21, dtype=np.int32), "hour": hrs.copy()}
    Tair = -5 + 5 * np.sin((hrs - 8) / 24 * 2 * np.pi)
    lib.orc_satvap.restype = C.c_double
    lib.orc_satvap.argtypes = [C.c_double]
    ea = 0.7 * lib.orc_satvap(float(np.mean(Tair)))
    RH = np.array([ea / lib.orc_satvap(float(t)) * 100 for t in Tair])
    return obst, Tair, RH, n


def test_pointmodelsnow_equals_oracle_on_the_reference_test(oracle):
    want = dict(RT.LAST_POINTSNOW) if RT.LAST_POINTSNOW else None
    obst, Tair, RH, n = _snow_test_inputs(oracle)
    want = dict(RT.LAST_POINTSNOW)
    # the replay's radiation inputs are not kept: rebuild the same weather through the oracle's helpers
    lib = oracle.load()
    Pk = np.full(n, 101.3)
    csr, zen, azi, si = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    d = lambda a: a.ctypes.data_as(DP)                                                  # noqa: E731
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))                                  # noqa: E731
    lib.orc_clearskyrad.restype = None
    lib.orc_clearskyrad.argtypes = None
    lib.orc_clearskyrad(C.c_int(n), i(obst["year"]), i(obst["month"]), i(obst["day"]), d(obst["hour"]), C.c_double(50.0),
                        C.c_double(-5.0), d(Tair), d(RH), d(Pk), d(csr))
    lib.orc_solpositionv.restype = None
    lib.orc_solpositionv(C.c_int(n), i(obst["year"]), i(obst["month"]), i(obst["day"]), d(obst["hour"]), C.c_double(50.0),
                         C.c_double(-5.0), C.c_double(0.0), C.c_double(180.0), d(zen), d(azi), d(si))
    SWd = 0.5 * csr
    clim = {"temp": Tair, "relhum": RH, "pres": Pk, "swdown": SWd, "difrad": SWd - 0.3 * csr * si, "lwdown": np.full(n, 350.0),
            "windspeed": np.full(n, 2.0), "precip": np.full(n, 1.0)}
    got = PM.pointmodelsnow(obst, clim, [2, 0.5, 0.05, 0], [0, 180, 50, -5, 2, 0, 0], "Taiga")
    assert got["iters"] == want["iters"]
    for k in ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "sdeng", "G", "RswabsG", "RlwabsG", "tr", "umu", "sublmelt", "tempmelt",
              "rainmelt", "sstemp"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-10, atol=1e-10, err_msg=k)
    # test-pointmodelsnow.R's own bounds hold for the product as they do for the oracle
    assert np.isfinite(got["Tc"]).all() and (got["sdepc"] >= 0).all() and got["sdepc"][-1] > 0


@pytest.mark.parametrize("i", range(8))
def test_random_snow_point_series_equal_oracle(oracle, i):
    rng = np.random.default_rng(9300 + i)
    days = int(rng.integers(1, 12))
    a = synthetic.workload(2, 2, days * 24, reqhgt=0.05, start_doy=int(rng.choice([5, 40, 340])),
                           lat=float(rng.choice([46.0, 57.0, 68.0])), cold=float(rng.choice([4.0, 9.0, 14.0])),
                           seed=int(rng.integers(1, 1 << 30)))
    c = a["climdata"]
    n = days * 24
    clim = {"temp": c["temp"], "relhum": np.clip(100 * c["ea"] / c["es"], 5, 100), "pres": c["pres"], "swdown": c["swdown"],
            "difrad": c["difrad"], "lwdown": c["lwdown"], "windspeed": np.maximum(c["windspeed"], 0.5),
            "precip": np.where(rng.random(n) < 0.2, rng.uniform(0, 3, n), 0.0)}
    hgt = float(rng.choice([0.0, 0.3, 1.5, 3.0]))
    vegp = [float(rng.uniform(0.2, 3)) if hgt > 0 else 0.0, hgt, float(rng.uniform(0.02, 0.3)), float(rng.uniform(0, 0.5))]
    other = [float(rng.uniform(0, 20)), float(rng.uniform(0, 360)), a["lat"], a["lon"], hgt + 2.0, float(rng.uniform(0, 0.6)),
             float(rng.integers(0, 200))]
    env = str(rng.choice(["Alpine", "Maritime", "Prairie", "Tundra", "Taiga", "Ephemeral"]))
    want = RT.pointmodelsnow(a["obstime"], clim, np.array(vegp), np.array(other), env, 0.5, 30)
    got = PM.pointmodelsnow(a["obstime"], clim, vegp, other, env, 0.5, 30)
    assert got["iters"] == want["iters"]
    for k in ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "G", "RswabsG", "RlwabsG", "tr", "umu", "sublmelt", "tempmelt", "rainmelt"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-9, err_msg=k)


def test_man_refuses_windows_longer_than_the_series():
    """the reference's circular index leaves the array there (cpp:561-572; found by tools/host_sanitizers.sh)"""
    from microclimf_amd import _abi
    x = np.arange(48.0)
    assert PM.manCpp(x, 48)[-1] == pytest.approx(x.mean())
    assert np.isfinite(PM.manCpp(x, 49)).all()                          # 2 days of daily means: still inside
    for win in (72, 500):
        with pytest.raises(_abi.McfError, match="longer than the series"):
            PM.manCpp(x, win)
    with pytest.raises(_abi.McfError, match="longer than the series"):
        PM.manCpp(np.arange(10.0), 11)
    assert np.isfinite(PM.manCpp(np.arange(240.0), 100)).all()          # 4-day mean of daily means: fine
