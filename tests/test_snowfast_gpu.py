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


def _crop(vegp, soilc, dtm, r0, r1, c0, c1):
    cut = lambda a: np.asarray(a)[r0:r1, c0:c1]                        # noqa: E731
    return {k: cut(v) for k, v in vegp.items()}, {k: cut(v) for k, v in soilc.items()}, dict(dtm, z=cut(dtm["z"]))


@pytest.mark.parametrize("case", [
    dict(days=[4, 11, 12, 30, 47], window=(0, 50, 0, 50)),                                             # two consecutive days
    dict(days=[3, 20, 44], window=(5, 28, 10, 47), snowenv="Alpine", snowinitd=0.002, snowinita=30.0, stfact=0.03),
    dict(days=[6, 7, 8, 35], window=(20, 50, 0, 19), snowenv="Tundra", zref=3.0, windhgt=2.0),
    dict(days=[10, 40], window=(12, 13, 0, 50), snowenv="Prairie", cold=-14.0),                        # a single row of cells
])
def test_fast_method_matches_the_oracle_chain(oracle, case):
    """50 days of the bundled site made colder, a few selected days (consecutive ones too: R's `a:b` then counts down),
    windows of the raster down to one row, snow environments, initial pack, other reference heights; the product
    (`runsnowmodel(method = "fast")`) against the oracle's point model, terrain, grid model, position index and day loop"""
    from oracle import replay_reference_tests as RT
    from oracle import snowfast_oracle as SF
    weather, vegp, soilc, dtm = load(50 * 24)
    vegp, soilc, dtm = _crop(vegp, soilc, dtm, *case["window"])
    weather = dict(weather, temp=weather["temp"] + case.get("cold", -9.0))
    env, sd0, sa0 = case.get("snowenv", "Taiga"), case.get("snowinitd", 0.0), case.get("snowinita", 0.0)
    zref, windhgt, stfact = case.get("zref", 2.0), case.get("windhgt", case.get("zref", 2.0)), case.get("stfact", 0.01)
    mp = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc), days=case["days"])
    got = F.runsnowmodel(weather, mp, vegp, soilc, dtm, snowenv=env, snowinitd=sd0, snowinita=sa0, zref=zref, windhgt=windhgt,
                         stfact=stfact)                                 # the reference's default: method = "fast"
    z = np.asarray(dtm["z"])
    n = 24 * len(case["days"])
    assert list(got) == ["Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden", "umu"] and got["Tc"].shape == z.shape + (n,)
    vg = F.cleanvegp(vegp)
    vp = F.sortvegp_point(vg)
    obst = {k: np.asarray(v) for k, v in weather["obstime"].items()}
    w = {k: np.array(weather[k], dtype=np.float64) for k in F.WEATHER}
    if zref != windhgt:
        w["windspeed"] = w["windspeed"] * np.log(67.8 * zref - 5.42) / np.log(67.8 * windhgt - 5.42)
    assert np.nanmax(vg["hgt"]) <= zref                                  # no weather height adjustment in these cases
    sdep, sage = z * 0 + sd0, z * 0 + sa0
    pm = RT.pointmodelsnow(obst, w, np.array([vp[1], vp[0], vp[5], vp[3]]),
                           np.array([0, 0, mp["lat"], mp["long"], zref, np.nanmean(sdep), np.nanmean(sage)]), env, maxiter=20)
    T = len(w["temp"])
    ai = np.asarray(mp["subs"]) - 1
    pointm = {"Gp": pm["G"], "Tc": pm["Tc"], "RswabsG": pm["RswabsG"], "RlwabsG": pm["RlwabsG"], "umu": pm["umu"], "tr": pm["tr"]}
    vs = F.sortl(vg, pm["sdepc"][:T])
    vs["leaft"] = np.where(np.isnan(vs["leaft"]), 0.01, vs["leaft"])
    other = {"zref": zref, "lat": mp["lat"], "lon": mp["long"], "isnowdc": sd0 * z, "isnowac": sage, "isnowag": sage}
    rows = lambda d: {k: np.asarray(v)[ai] for k, v in d.items()}      # noqa: E731
    want = SF.snowmodelq1_days(rows(obst), rows(w), rows(pointm), pm, w["temp"], np.where(w["temp"] > 2, 0.0, w["precip"]),
                               mp["subs"], vs, other, env, z, dtm["res"], stfact)
    np.testing.assert_allclose(got["umu"], pm["umu"][ai], rtol=1e-10)
    for k in want:
        g, x = got[k], want[k]
        assert np.array_equal(np.isnan(g), np.isnan(x)), k
        assert np.array_equal(np.isinf(g), np.isinf(x)), k
        fin = np.isfinite(x)
        err = np.max(np.abs(g[fin] - x[fin]) / (1 + np.abs(x[fin])))
        assert err < 1e-6, (k, err)
    assert np.nanmax(got["groundsnowdepth"][np.isfinite(got["groundsnowdepth"])]) > 0.01


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


def _slow(SA, obst, clim_c, pointm_c, vg, other, z, dtmc, dtm, api, cr, cc, altcorrect):
    return SA.snowmodel2_chunks(obst, clim_c, pointm_c, F.sortl(vg, np.max(pointm_c["sdepc"], axis=(0, 1))), other, "Taiga", z, dtmc,
                                dtm["res"], 0.01, api.coarse_positions(50, cr), api.coarse_positions(50, cc), altcorrect=altcorrect,
                                agg=1)


@pytest.mark.parametrize("altcorrect,subset", [(0, False), (2, False), (1, True), (0, "fast"), (2, "fast")])
def test_array_weather_snow_model_matches_the_oracle_chain(oracle, altcorrect, subset):
    """`runsnowmodel()` with array weather (`.snowmodel2`): a 2 x 3 grid of perturbed cold climate cells over the bundled
    site, 11 days (two 5-day chunks and a ragged day); the snow point model per climate cell, resampling, altitude
    correction, terrain refresh, gridmodelsnow2, position index and hand-over against the oracle's pieces"""
    from oracle import replay_reference_tests as RT
    from oracle import snowarray_oracle as SA
    from microclimf_amd import api
    weather, vegp, soilc, dtm = load(11 * 24)
    cr, cc, T = 2, 3, 11 * 24
    rng = np.random.default_rng(9)
    climarray = {}
    for k in F.WEATHER:
        base = np.broadcast_to(weather[k][None, None, :], (cr, cc, T)).copy()
        if k == "temp":
            base += -9.0 + rng.uniform(-1.5, 1.5, (cr, cc, 1))
        elif k in ("swdown", "difrad", "windspeed", "precip"):
            base *= rng.uniform(0.9, 1.1, (cr, cc, 1))
        elif k == "winddir":
            base = (base + rng.integers(-1, 2, (cr, cc, T)) * 10.0) % 360
        climarray[k] = np.asfortranarray(base)
    climarray["difrad"] = np.minimum(climarray["difrad"], climarray["swdown"])
    clat = dtm["lat"] + 1e-4 * np.arange(cr)[:, None] + 0 * np.arange(cc)[None, :]
    clon = dtm["long"] + 1e-4 * np.arange(cc)[None, :] + 0 * np.arange(cr)[:, None]
    lats = dtm["lat"] + 9e-6 * np.arange(50)[::-1, None] + 0 * np.arange(50)[None, :]
    lons = dtm["long"] + 1.4e-5 * np.arange(50)[None, :] + 0 * np.arange(50)[:, None]
    z = np.asarray(dtm["z"])
    dtmc = np.array([[np.nanmean(z[:25, :17]), np.nanmean(z[:25, 17:34]), np.nanmean(z[:25, 34:])],
                     [np.nanmean(z[25:, :17]), np.nanmean(z[25:, 17:34]), np.nanmean(z[25:, 34:])]]) + 40.0
    mpa = F.runpointmodela(climarray, weather["obstime"], 0.05, dtm, vegp, soilc, lats=clat, lons=clon)
    if subset:
        mpa = [F.subsetpointmodel(m, days=[2, 7, 8]) for m in mpa]
    got = F.runsnowmodela(climarray, weather["obstime"], mpa, vegp, soilc, dtm, dtmc=dtmc, lats_c=clat, lons_c=clon, lats=lats,
                          lons=lons, altcorrect=altcorrect, method="fast" if subset == "fast" else "slow")
    assert list(got) == ["Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden", "umu"]
    # the same through the oracle
    vg = F.cleanvegp(vegp)
    obst = {k: np.asarray(v) for k, v in weather["obstime"].items()}
    wdir = np.array([F.getmode(climarray["winddir"][:, :, k]) for k in range(T)])
    vc = {k: F.block_reduce(vg[k], cr, cc) for k in ("pai", "hgt", "leaft", "clump")}
    clim_c = {k: np.array(climarray[k], copy=True) for k in F.WEATHER if k != "winddir"}
    clim_c["winddir"] = wdir
    names = {"Gp": "G", "Tc": "Tc", "RswabsG": "RswabsG", "RlwabsG": "RlwabsG", "umu": "umu", "tr": "tr", "sdepc": "sdepc"}
    if subset == "fast":
        names.update({k: k for k in ("sublmelt", "tempmelt", "rainmelt", "sstemp", "sdenc", "sdeng")})
    pointm_c = {k: np.empty((cr, cc, T)) for k in names}
    for i in range(cr):
        for j in range(cc):
            w = {k: np.ascontiguousarray(clim_c[k][i, j, :]) for k in clim_c if k != "winddir"}
            pm = RT.pointmodelsnow(obst, w, np.array([np.mean(vc[k][i, j, :]) for k in ("pai", "hgt", "leaft", "clump")]),
                                   np.array([0, 0, clat[i, j], clon[i, j], 2.0, 0, 0]), "Taiga", maxiter=10)
            for k, v in names.items():
                pointm_c[k][i, j, :] = pm[v][1:T + 1] if subset == "fast" and k == "sdepc" else pm[v][:T]
    other = {"zref": 2.0, "lats": lats, "lons": lons, "isnowdc": z * 0, "isnowac": z * 0, "isnowdg": z * 0, "isnowag": z * 0}
    if subset == "fast":
        from oracle import snowfast_oracle as SF
        subs = np.asarray(mpa[0]["subs"])
        ai = subs - 1
        sel = lambda d: {k: (np.asarray(v)[ai] if np.ndim(v) == 1 else np.asarray(v)[:, :, ai]) for k, v in d.items()}   # noqa: E731
        pm2 = {k: pointm_c[k] for k in ("sublmelt", "tempmelt", "rainmelt", "sstemp", "sdenc", "sdeng")}
        pm2["tc"] = clim_c["temp"]
        pm2["snow"] = np.where(clim_c["temp"] > 2, 0.0, clim_c["precip"])
        pm_s = sel({k: pointm_c[k] for k in ("Gp", "Tc", "RswabsG", "RlwabsG", "umu", "tr", "sdepc")})
        want = SF.snowmodelq2_days(sel(obst), sel(clim_c), pm_s, pm2, subs, F.sortl(vg, np.max(pm_s["sdepc"], axis=(0, 1))),
                                   {k: other[k] for k in ("zref", "lats", "lons", "isnowdc", "isnowac", "isnowag")}, "Taiga", z, dtmc,
                                   dtm["res"], 0.01, api.coarse_positions(50, cr), api.coarse_positions(50, cc), altcorrect=altcorrect)
        subset = None                                                     # already the selected hours
        assert got["Tc"].shape == (50, 50, 72)
    else:
        want = _slow(SA, obst, clim_c, pointm_c, vg, other, z, dtmc, dtm, api, cr, cc, altcorrect)
    if subset:
        i = np.asarray(mpa[0]["subs"]) - 1
        want = {k: v[:, :, i] for k, v in want.items()}
        assert got["Tc"].shape == (50, 50, 72)
    for k in want:
        g, x = got[k], want[k]
        assert np.array_equal(np.isnan(g), np.isnan(x)), k
        with np.errstate(invalid="ignore"):
            err = np.nanmax(np.abs(g - x) / (1 + np.abs(x)))
        assert err < 1e-6, (k, err)
    assert np.nanmax(got["groundsnowdepth"]) > 0.005
    if not subset and got["Tc"].shape[2] == T:
        assert np.all(np.isnan(got["Tc"][:, :, 240:])) and not np.all(np.isnan(got["umu"][:, :, 240:]))   # past the last chunk


def test_array_weather_snow_model_refusals(oracle):
    weather, vegp, soilc, dtm = load(6 * 24)
    mp = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
    sub = F.subsetpointmodel(mp, days=[2, 4])
    one = np.ones((1, 2))
    arr = {k: np.broadcast_to(weather[k][None, None, :], (1, 2, 144)).copy() for k in F.WEATHER}
    kw = dict(dtmc=one * 50, lats_c=one * dtm["lat"], lons_c=one * dtm["long"], lats=np.full((50, 50), dtm["lat"]),
              lons=np.full((50, 50), dtm["long"]))
    assert F.runsnowmodela(arr, weather["obstime"], [sub, sub], vegp, soilc, dtm, **kw)["Tc"].shape == (50, 50, 48)   # method = "fast"
    with pytest.raises(ValueError, match="tallest vegetation"):
        F.runsnowmodela(arr, weather["obstime"], [sub, sub], vegp, soilc, dtm, method="slow", zref=1.0, **kw)
    with pytest.raises(ValueError, match="needs a micropoint"):
        F.runsnowmodela(arr, weather["obstime"], [mp, None], vegp, soilc, dtm, **kw)


@pytest.mark.parametrize("altcorrect", [0, 2])
def test_runmicro_with_snow_and_array_weather(oracle, altcorrect):
    """`runmicro(snow = TRUE)` with array weather (`.runmicrosnow2`): a cold spell followed by a thaw over a 2 x 2 climate
    grid — no-snow days through the coarse-array solver, snow days through gridmicrosnow2, merged by day; the product
    against the same orchestration with the oracle's solver, snow microclimate and terrain behind it"""
    from functools import partial
    from oracle import coarse_oracle as CO
    from oracle import terrain_oracle as TO
    from microclimf_amd import api
    weather, vegp, soilc, dtm = load(15 * 24)
    cr, cc, T = 2, 2, 15 * 24
    t = np.arange(T)
    rng = np.random.default_rng(3)
    climarray = {}
    for k in F.WEATHER:
        base = np.broadcast_to(weather[k][None, None, :], (cr, cc, T)).copy()
        if k == "temp":
            base += -9.0 + 8.0 * (t > 5 * 24) + 7.0 * (t > 10 * 24) + rng.uniform(-1.0, 1.0, (cr, cc, 1))
        elif k in ("swdown", "difrad", "windspeed", "precip"):
            base *= rng.uniform(0.9, 1.1, (cr, cc, 1))
        climarray[k] = np.asfortranarray(base)
    climarray["difrad"] = np.minimum(climarray["difrad"], climarray["swdown"])
    clat = dtm["lat"] + 1e-4 * np.arange(cr)[:, None] + 0 * np.arange(cc)[None, :]
    clon = dtm["long"] + 1e-4 * np.arange(cc)[None, :] + 0 * np.arange(cr)[:, None]
    lats = dtm["lat"] + 9e-6 * np.arange(50)[::-1, None] + 0 * np.arange(50)[None, :]
    lons = dtm["long"] + 1.4e-5 * np.arange(50)[None, :] + 0 * np.arange(50)[:, None]
    z = np.asarray(dtm["z"])
    dtmc = np.array([[np.nanmean(z[:25, :25]), np.nanmean(z[:25, 25:])], [np.nanmean(z[25:, :25]), np.nanmean(z[25:, 25:])]]) + 30.0
    mpa = F.runpointmodela(climarray, weather["obstime"], 0.05, dtm, vegp, soilc, lats=clat, lons=clon)
    smod = F.runsnowmodela(climarray, weather["obstime"], mpa, vegp, soilc, dtm, dtmc=dtmc, lats_c=clat, lons_c=clon, lats=lats,
                           lons=lons, altcorrect=altcorrect)
    sd = S.snowdaysfun(S.applycpp3(np.nan_to_num(smod["totalSWE"]), "max"), S.applycpp3(np.nan_to_num(smod["totalSWE"]), "min"))
    assert sd["snowdays"].sum() > 0 and sd["nosnowdays"].sum() > 0

    def solve_oracle(mpx, reqhgt, **kw):
        tf = kw.pop("tfact", 1.5)
        zz = F.cleanvars(vegp, soilc, dtm["z"])[2]
        ter = TO.terrain(zz, dtm["res"], mpx[0]["zref"])
        a = F.prepare_grid_inputs_array(mpx, cr, cc, reqhgt, vegp, soilc, dtm, lats=lats, lons=lons, slr=ter["slope"],
                                        apr=ter["aspect"], hor=ter["hor"], svf=ter["svfa"], wsa=ter["wsa"], **kw)
        a["tfact"] = tf
        clim, pm = CO.expand(a["climdata"], a["pointm"], api.coarse_positions(50, cr), api.coarse_positions(50, cc),
                             altcorrect=altcorrect, dtmc=dtmc, dtm=zz)
        a.update(climdata=clim, pointm=pm)
        return oracle.run_grid(**a, array_forcing=True)

    kw = dict(dtmc=dtmc, lats=lats, lons=lons, altcorrect=altcorrect)
    for reqhgt in (0.05, 0.0):
        got = F.runmicro_snow_array(mpa, cr, cc, reqhgt, vegp, soilc, dtm, smod, **kw)
        want = F.runmicro_snow_array(mpa, cr, cc, reqhgt, vegp, soilc, dtm, smod, _solve=solve_oracle,
                                     _microsnow=partial(oracle.run_microsnow, array_forcing=True), _terrain=TO.terrain, **kw)
        assert list(got) == list(want)
        for k in want:
            assert got[k].shape == (50, 50, T)
            assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
            with np.errstate(invalid="ignore"):
                err = np.nanmax(np.abs(got[k] - want[k]) / (1 + np.abs(want[k])))
            assert err < 1e-6, (reqhgt, k, err)
