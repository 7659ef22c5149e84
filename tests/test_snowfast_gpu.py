"""The fast snow method (`runsnowmodel(method = "fast")` -> `.snowmodelq1`, R/internal.R:2627-2776) on the device: the
position index entry against the oracle's `.tpicalc`, the whole chain against the oracle's restatement of the day loop, and
the red curve of the reference's published figure."""
import numpy as np
import pytest

from bundled import load
from microclimf_amd import frontend as F
from microclimf_amd import snow as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,af", [((50, 50), 3), ((50, 50), 7), ((37, 64), 4), ((20, 31), 1), ((12, 40), 6), ((12, 40), 30)])
def test_tpicalc_matches_the_oracle(oracle, shape, af):
    from oracle import snowdriver_oracle as SD
    rng = np.random.default_rng(af * 100 + shape[0])
    r, c = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    z = 40 * np.sin(r / 9.0) * np.cos(c / 7.0) + 3 * rng.standard_normal(shape) + 100
    z[1:4, 2:5] = np.nan
    z[-1, -1] = np.nan
    for tfact in (0.01, 0.6):                           # 0.6: both clamps of `tpic` are reached
        got = S.tpicalc(af, z, tfact)
        want = SD.tpicalc(af, min(shape), z, tfact)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got, want, rtol=1e-11, equal_nan=True)
        assert abs(np.nanmean(got) - 1) < 1e-12


def test_tpicalc_refuses_what_terra_refuses():
    from microclimf_amd import _abi
    with pytest.raises(_abi.McfError, match="aggregation factor"):
        S.tpicalc(0, np.zeros((5, 5)), 0.01)


def test_fast_method_matches_the_oracle_chain(oracle):
    """50 days of the bundled site made colder, five selected days of which two are consecutive (R's `a:b` then counts
    down); the product (`runsnowmodel(method = "fast")`) against the oracle's point model, terrain, grid model, position
    index and day loop"""
    from oracle import replay_reference_tests as RT
    from oracle import snowfast_oracle as SF
    weather, vegp, soilc, dtm = load(50 * 24)
    weather = dict(weather, temp=weather["temp"] - 9.0)
    mp = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc), days=[4, 11, 12, 30, 47])
    got = F.runsnowmodel(weather, mp, vegp, soilc, dtm)                  # the reference's default: method = "fast"
    assert list(got) == ["Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden", "umu"] and got["Tc"].shape == (50, 50, 120)
    vg = F.cleanvegp(vegp)
    vp = F.sortvegp_point(vg)
    z = np.asarray(dtm["z"])
    obst = {k: np.asarray(v) for k, v in weather["obstime"].items()}
    w = {k: np.asarray(weather[k], dtype=np.float64) for k in F.WEATHER}
    pm = RT.pointmodelsnow(obst, w, np.array([vp[1], vp[0], vp[5], vp[3]]), np.array([0, 0, mp["lat"], mp["long"], 2.0, 0, 0]),
                           "Taiga", maxiter=20)
    n = len(w["temp"])
    ai = np.asarray(mp["subs"]) - 1
    pointm = {"Gp": pm["G"], "Tc": pm["Tc"], "RswabsG": pm["RswabsG"], "RlwabsG": pm["RlwabsG"], "umu": pm["umu"], "tr": pm["tr"]}
    vs = F.sortl(vg, pm["sdepc"][:n])
    vs["leaft"] = np.where(np.isnan(vs["leaft"]), 0.01, vs["leaft"])
    other = {"zref": 2.0, "lat": mp["lat"], "lon": mp["long"], "isnowdc": z * 0, "isnowac": z * 0, "isnowag": z * 0}
    rows = lambda d: {k: np.asarray(v)[ai] for k, v in d.items()}      # noqa: E731
    want = SF.snowmodelq1_days(rows(obst), rows(w), rows(pointm), pm, w["temp"], np.where(w["temp"] > 2, 0.0, w["precip"]),
                               mp["subs"], vs, other, "Taiga", z, dtm["res"], 0.01)
    np.testing.assert_allclose(got["umu"], pm["umu"][ai], rtol=1e-10)
    for k in want:
        g, x = got[k], want[k]
        assert np.array_equal(np.isnan(g), np.isnan(x)), k
        err = np.nanmax(np.abs(g - x) / (1 + np.abs(x)))
        assert err < 1e-6, (k, err)
    assert np.nanmax(got["groundsnowdepth"]) > 0.01


def test_fast_method_cannot_start_on_the_first_day(oracle):
    weather, vegp, soilc, dtm = load(10 * 24)
    mp = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc), days=[1, 6])
    with pytest.raises(ValueError, match="first day"):
        F.runsnowmodel(weather, mp, vegp, soilc, dtm)
    assert F.runsnowmodel(weather, mp, vegp, soilc, dtm, method="slow")["Tc"].shape == (50, 50, 48)


def test_vignette_fast_snow_depth_steps_match_the_published_figure():
    """vignettes/images/image14p.png, the red curve (running-microclimf.Rmd:663-683): climdata$temp - 12, the point model
    subset to each month's coldest day, `runsnowmodel(method = "fast")`; raster-mean depth = totalSWE / snowden at the start
    of the 12 selected days as read off the figure (to about 0.01 m): below the slow method's curve from February to May"""
    weather, vegp, soilc, dtm = load()
    cold = dict(weather, temp=weather["temp"] - 12.0)
    mp = F.subsetpointmodel(F.runpointmodel(cold, 0.05, dtm, vegp, soilc), tstep="month", what="tmin")
    smod = F.runsnowmodel(cold, mp, vegp, soilc, dtm, method="fast")
    with np.errstate(invalid="ignore", divide="ignore"):
        depth = np.nanmean(smod["totalSWE"] / smod["snowden"], axis=(0, 1))
    start = [0.022, 0.33, 0.57, 0.285, 0.12, 0.04, 0.03, 0.0, 0.005, 0.0, 0.06, 0.335]
    for m, want in enumerate(start):
        assert abs(depth[m * 24] - want) < 0.015, (m + 1, depth[m * 24], want)
    assert abs(depth[3 * 24 - 1] - 0.605) < 0.015
