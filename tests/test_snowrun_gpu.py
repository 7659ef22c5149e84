"""mcf_runmicrosnow1 / mcf_runmicrosnow1_multi / mcf_snowrun_* (include/mcf.h): `runmicro(..., snow = TRUE)` with data.frame
weather as ONE library call — `.snowmodel1`'s chunk loop (R/internal.R:2498-2619) and `.runmicrosnow1`'s two models and merge
(R/internal.R:3581-3659) device-resident.  Held against
  (1) the reference's orchestration on the host through the one-shot entry points (whole-series snow arrays in host memory,
      the solver and gridmicrosnow1 on day SUBSETS, merge_snow_outputs): HIP vs HIP, 1e-12;
  (2) the same orchestration with the ORACLE's solver and snow microclimate behind it (oracle/mcf_oracle.c, snow_oracle.c)
      on the device's snow series: 1e-6 — every one of the merged ten outputs, every cell-step;
  (3) row blocks (more blocks than devices, two host threads on one device): 1e-9 / bitwise for one block."""
import numpy as np
import pytest

from microclimf_amd import snow as S
from microclimf_amd import synthetic
from microclimf_amd.api import runmicro1Cpp

pytestmark = pytest.mark.gpu
ARGS = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact",
        "complete", "mat", "out")
MAT = 7.5


def _steps(days0):
    return (np.repeat(np.asarray(days0) * 24, 24) + np.tile(np.arange(24), len(days0))).astype(np.int64)


def _sub(d, idx):
    return {k: (np.asarray(v)[idx] if np.ndim(v) == 1 else v) for k, v in d.items()}


def _case(reqhgt, cold, doy, rows=22, cols=13, ndays=20):
    T = ndays * 24
    sw = synthetic.snow_workload(rows, cols, T, cold=cold, zref=3.5, start_doy=doy)
    a = synthetic.workload(rows, cols, T, reqhgt=reqhgt, zref=3.5, hgt_range=(0.05, 3.0), start_doy=doy, variety=True)
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    snow = dict(sw, dtm=dtm, res=1.0, tfact=0.02)
    micro = {"obstime": sw["obstime"], "climdata": sw["climdata"], "vegp": sw["vegp"], "other": sw["other"]}
    return sw, a, dtm, snow, micro


def _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, solve, microsnow, oracle_merge=False):
    """`.runmicrosnow1` steps (3)-(5) on host arrays: `solve(args of the day subset)`, `microsnow(...)` on the snow-day subset.
    oracle_merge: the template and the merge by oracle/snowmerge_oracle.py (its own restatement of R/internal.R:3565-3578,
    3625-3656) instead of the product's snow.merge_snow_outputs — the oracle-backed leg checks the product's merge with it"""
    rows, cols = dtm.shape
    ni, si = _steps(ndays_), _steps(sdays)
    outm = [1] * 10 if reqhgt > 0 else [1 if i in (0, 3, 5, 6, 7, 8, 9) else 0 for i in range(10)]
    an = dict(a, obstime=_sub(a["obstime"], ni), climdata=_sub(a["climdata"], ni), pointm=_sub(a["pointm"], ni))
    moutn = solve(an)
    micro = {}
    if oracle_merge:
        from oracle import snowmerge_oracle as MO
        micro = MO.prep_micro(moutn, sdays + 1, ndays_ + 1, rows, cols)
    else:
        s1 = np.arange(si.size)[np.repeat(np.isin(sdays, ndays_), 24)]
        s2 = np.arange(ni.size)[np.repeat(np.isin(ndays_, sdays), 24)]
        for k, v in moutn.items():
            m = np.full((rows, cols, si.size), np.nan, order="F")
            m[:, :, s1] = v[:, :, s2]
            micro[k] = m
    swe = smod["totalSWE"].copy()
    swe[np.isnan(swe)] = 0.0
    swe[np.isnan(dtm)] = np.nan
    smods = {k: np.asfortranarray((swe if k == "totalSWE" else v)[:, :, si]) for k, v in smod.items()}
    mouts = microsnow(reqhgt, _sub(sw["obstime"], si), _sub(sw["climdata"], si), smods, micro, sw["vegp"], sw["other"], MAT, outm)
    for k in moutn:
        if k not in mouts:
            mouts[k] = micro[k]
    if oracle_merge:
        return MO.merge(moutn, mouts, sdays + 1, ndays_ + 1, rows, cols)
    return S.merge_snow_outputs(moutn, mouts, sdays + 1, ndays_ + 1, rows, cols)


def _close(got, want, tol, what):
    assert list(got) == list(want)
    for k in want:
        g, w = got[k], want[k]
        assert g.shape == w.shape, (what, k)
        assert np.array_equal(np.isnan(g), np.isnan(w)), (what, k)
        fin = np.isfinite(w)
        if fin.any():
            err = float(np.max(np.abs(g[fin] - w[fin]) / (1 + np.abs(w[fin]))))
            assert err < tol, (what, k, err)


@pytest.mark.parametrize("reqhgt,cold,doy", [(0.05, 0.0, 90), (0.3, -3.0, 20), (0.0, 3.0, 120)])
def test_one_call_equals_the_host_orchestration_and_the_oracle_backed_one(oracle, reqhgt, cold, doy):
    sw, a, dtm, snow, micro = _case(reqhgt, cold, doy)
    got, smod = S.runmicrosnow1(a, snow, micro, MAT, want_smod=True)
    # the snow series the entry returns are mcf_snowmodel1's
    want_smod = S.snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
    for k in smod:
        assert np.array_equal(smod[k], want_smod[k], equal_nan=True), k
    # the staged form gives the day classes — and the same output
    with S.SnowRun(a, snow) as run:
        sd, nd = run.pass1()
        got2 = run.pass2(micro, MAT)
    for k in got:
        assert np.array_equal(got[k], got2[k], equal_nan=True), k
    swe = smod["totalSWE"].copy()
    swe[np.isnan(swe)] = 0.0
    swe[np.isnan(dtm)] = np.nan
    days = S.snowdaysfun(S.applycpp3(swe, "max"), S.applycpp3(swe, "min"))
    assert np.array_equal(sd, days["snowdays"]) and np.array_equal(nd, days["nosnowdays"])
    sdays, ndays_ = np.flatnonzero(sd), np.flatnonzero(nd)
    assert sdays.size >= 3 and ndays_.size >= 3 and (sd & nd).sum() >= 1 and (sd | nd).all()
    # (1) the reference's orchestration on the host, HIP behind it
    want = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: runmicro1Cpp(*[an[k] for k in ARGS]), S.gridmicrosnow1)
    _close(got, want, 1e-12, "host-orchestrated HIP")
    # (2) ... and with the oracle's solver and snow microclimate behind it
    want_o = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: oracle.run_grid(**{k: an[k] for k in ARGS}),
                          oracle.run_microsnow, oracle_merge=True)
    _close(got, want_o, 1e-6, "oracle-backed orchestration")
    # every class of cell-step is in the comparison: snow-covered, snow-free on a snow day, no-snow day
    si = _steps(sdays)
    covered = swe[:, :, si] > 0
    assert covered.any() and (~covered & ~np.isnan(dtm)[:, :, None]).any()
    if doy == 90:
        assert np.setdiff1d(ndays_, sdays).size > 0 and np.setdiff1d(sdays, ndays_).size > 0       # days of one class only, both ways


def test_row_blocks_and_host_threads(oracle):
    reqhgt, cold, doy = 0.05, 0.0, 90
    sw, a, dtm, snow, micro = _case(reqhgt, cold, doy, rows=320, cols=24, ndays=10)
    one = S.runmicrosnow1(a, snow, micro, MAT)
    same = S.runmicrosnow1(a, snow, micro, MAT, devices=[0], n_blocks=1)
    for k in one:
        assert np.array_equal(one[k], same[k], equal_nan=True), k            # one block: bit for bit
    for devices, nb in (([0], 2), ([0, 0], 2)):                               # blocks > devices; two host threads on one device
        multi, smod_m = S.runmicrosnow1(a, snow, micro, MAT, devices=devices, n_blocks=nb, want_smod=True)
        _close(multi, one, 1e-9, f"{nb} blocks on {devices}")
    smod_1 = S.snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
    for k in smod_m:
        w = smod_1[k]
        assert np.array_equal(np.isnan(smod_m[k]), np.isnan(w)), k
        assert np.nanmax(np.abs(smod_m[k] - w) / (1 + np.abs(w))) < 1e-9, k


def test_tiles_wholly_under_snow_are_left_out_and_nothing_changes(oracle):
    """A deep pack everywhere but on the northern rows, no snowfall: every day is a snow day and a no-snow day, and pass 2's
    solver leaves out the tiles inside the pack (mcf_snowrun_stats says so) — the one call still equals the reference's
    orchestration on the host, which solves every cell of every no-snow day (1e-12), and the oracle-backed one (1e-6)."""
    reqhgt, rows, cols, ndays = 0.05, 64, 24, 10
    sw, a, dtm, snow, micro = _case(reqhgt, -2.0, 60, rows=rows, cols=cols, ndays=ndays)
    deep = np.asfortranarray(np.where(np.arange(rows)[:, None] >= 9, 0.9, 0.0) * np.ones((1, cols)))
    other = dict(sw["other"], isnowdc=deep, isnowdg=np.asfortranarray(0.7 * deep))
    clim = dict(sw["climdata"], precip=np.zeros(ndays * 24))
    sw = dict(sw, other=other, climdata=clim)
    snow = dict(snow, other=other, climdata=clim)
    micro = dict(micro, other=other, climdata=clim)
    with S.SnowRun(a, snow) as run:
        sd, nd, smod = run.pass1(want_smod=True)
        got = run.pass2(micro, MAT)
        st = run.stats()
    assert (sd & nd).sum() >= 5 and st["tile_days_left_out"] >= 20 and st["tile_days_left_out"] < st["tile_days"], (sd, nd, st)
    sdays, ndays_ = np.flatnonzero(sd), np.flatnonzero(nd)
    want = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: runmicro1Cpp(*[an[k] for k in ARGS]), S.gridmicrosnow1)
    _close(got, want, 1e-12, "host-orchestrated HIP")
    want_o = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: oracle.run_grid(**{k: an[k] for k in ARGS}),
                          oracle.run_microsnow, oracle_merge=True)
    _close(got, want_o, 1e-6, "oracle-backed orchestration")


def test_a_year_without_snow_and_bad_arguments():
    sw, a, dtm, snow, micro = _case(0.05, -25.0, 170, rows=10, cols=9, ndays=7)       # midsummer, 25 K warmer: no pack survives
    snow["other"] = dict(snow["other"], isnowdc=np.zeros_like(dtm), isnowdg=np.zeros_like(dtm))
    with S.SnowRun(a, snow) as run:
        sd, nd = run.pass1()
        assert not sd.any() and nd.all()
        got = run.pass2(None, MAT)                     # no snow day: gridmicrosnow1's inputs are not needed
    want = runmicro1Cpp(*[a[k] for k in ARGS])
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k     # the solver's own output (mxtc of all days = the series')
    with pytest.raises(Exception, match="reqhgt < 0"):
        S.runmicrosnow1(dict(a, reqhgt=-0.1), snow, micro, MAT)
    with S.SnowRun(a, snow) as run:
        with pytest.raises(Exception, match="pass1 first"):
            run.pass2(micro, MAT)


def _subset_dfsel(layer_of_day, days0):
    """`.runmodel3Cpp` on a day subset (R/internal.R:1391-1399): the subset's days keep the layer the whole series gives them
    (`.sortvegp`, :252-270), consecutive days of one layer form one row of dfsel, the layers are renumbered 1.. in order —
    st / ed are step positions IN THE SUBSET.  -> (dfsel, the whole-series layers used, in order)"""
    lay = [int(layer_of_day[d]) for d in days0]
    used, st, ed = [], [], []
    for k, l in enumerate(lay):
        if not used or used[-1] != l:
            used.append(l); st.append(k * 24); ed.append(k * 24 + 23)
        else:
            ed[-1] = k * 24 + 23
    return {"lyr": np.arange(1, len(used) + 1), "st": np.array(st), "ed": np.array(ed)}, used


def test_time_varying_vegetation_runs_on_the_no_snow_days_with_the_whole_series_layers(oracle):
    """Round 5: a layered `vegp` — what the reference's BUNDLED example data is — behind the one-call entry.
    The reference solves the no-snow-day subset with runmicro3Cpp and a dfsel built on the subset (`.runmicronosnow`,
    R/internal.R:3333-3342); held against exactly that, HIP behind it (1e-12) and the oracle behind it (1e-6, all ten outputs)."""
    from microclimf_amd.api import runmicro3Cpp
    reqhgt, L = 0.05, 4
    sw, a, dtm, snow, micro = _case(reqhgt, 0.0, 90)
    al = synthetic.layered(a, L)
    ndays = len(a["obstime"]["year"]) // 24
    layer_of_day = np.zeros(ndays, int)
    for l in range(L):
        layer_of_day[al["dfsel"]["st"][l] // 24:(al["dfsel"]["ed"][l] + 1) // 24] = l
    got = S.runmicrosnow1(al, snow, micro, MAT)
    with S.SnowRun(al, snow) as run:
        sd, nd = run.pass1()
    sdays, ndays_ = np.flatnonzero(sd), np.flatnonzero(nd)
    assert len(set(layer_of_day[ndays_])) >= 3          # the no-snow days span several layers
    smod = S.snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
    dfs, used = _subset_dfsel(layer_of_day, ndays_)
    veg_sub = {k: np.asfortranarray(v[:, :, used]) for k, v in al["vegp"].items()}

    def solve_with(fn):
        def solve(an):
            an = dict(an, vegp=veg_sub)
            return fn(dfs, an)
        return solve
    want = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, solve_with(lambda d, an: runmicro3Cpp(d, *[an[k] for k in ARGS])),
                        S.gridmicrosnow1)
    _close(got, want, 1e-12, "host-orchestrated HIP, layered")
    want_o = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt,
                          solve_with(lambda d, an: oracle.run_grid(**{k: an[k] for k in ARGS}, dfsel=d)), oracle.run_microsnow,
                          oracle_merge=True)
    _close(got, want_o, 1e-6, "oracle-backed orchestration, layered")
    # and the layers matter: the single-layer run differs
    flat = S.runmicrosnow1(a, snow, micro, MAT)
    assert np.nanmax(np.abs(flat["Tz"] - got["Tz"])) > 1e-3


def test_kept_chunks_are_pooled_across_the_handles_years_and_change_no_bit(oracle):
    """mcf_snowrun_keep (round 5): pass 1's snow chunks stay in HBM up to a byte budget, pooled in the handle — the second
    year allocates nothing, pass 2 re-runs only what did not fit — and every output is bit for bit the unkept run's."""
    sw, a, dtm, snow, micro = _case(0.05, 0.0, 90)
    plain = S.runmicrosnow1(a, snow, micro, MAT)
    one = 120 * 22 * 13 * 8 * 5                      # one chunk's five series
    with S.SnowRun(a, snow) as run:
        run.keep(1.2 * one / 2 ** 30)                # room for ONE set: the other snow chunks are re-run
        for year in range(2):
            run.pass1()
            got = run.pass2(micro, MAT)
            st = run.stats()
            for k in plain:
                assert np.array_equal(got[k], plain[k], equal_nan=True), (year, k)
        assert st["chunks_kept"] >= 2 and st["chunks_rerun"] >= 2, st        # (counters run over both years)
        run.keep(0)                                  # off again: everything re-run, same bits
        run.pass1()
        got = run.pass2(micro, MAT)
        for k in plain:
            assert np.array_equal(got[k], plain[k], equal_nan=True), k
