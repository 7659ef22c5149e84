"""mcf_runmicrosnow2 (include/mcf.h): `runmicro(..., snow = TRUE)` with ARRAY weather as one library call — `.snowmodel2`'s chunk
loop (R/internal.R:2950-3008) and `.runmicrosnow2`'s two models and merge (:3661-3745) device-resident, every input at the
raster's resolution.  Held against `.runmicrosnow2`'s orchestration on host arrays:
  (1) HIP behind it — runmicro2Cpp on the no-snow-day subset arrays, gridmicrosnow2 on the snow-day subset arrays, the merge —
      1e-12;
  (2) the ORACLE's solver and snow microclimate behind it (oracle/mcf_oracle.c, snow_oracle.c, snowmerge_oracle.py): 1e-6 on all
      ten merged outputs of every cell-step."""
import numpy as np
import pytest

from microclimf_amd import snow as S
from microclimf_amd import synthetic
from microclimf_amd.api import runmicro2Cpp
from test_snowrun_gpu import _close, _steps

pytestmark = pytest.mark.gpu
ARGS = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact",
        "complete", "mat", "out")
MAT = 6.5


def _case(reqhgt, cold, doy, rows=18, cols=11, ndays=15):
    T = ndays * 24
    sw = synthetic.snow_workload(rows, cols, T, array_forcing=True, cold=cold, zref=3.5, start_doy=doy)
    a = synthetic.workload(rows, cols, T, reqhgt=reqhgt, zref=3.5, hgt_range=(0.05, 3.0), start_doy=doy, variety=True, array_forcing=True)
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    clim = sw["climdata"]
    wd = np.asarray(clim["winddir"], dtype=np.float64) * np.pi / 180
    wuv, wvv = np.nanmean(clim["windspeed"] * np.cos(wd), axis=(0, 1)), np.nanmean(clim["windspeed"] * np.sin(wd), axis=(0, 1))
    other = {k: sw["other"][k] for k in ("zref", "lats", "lons", "isnowdc", "isnowdg", "isnowac", "isnowag")}
    snow = dict(obstime=sw["obstime"], climdata=clim, pointm=sw["pointm"], vegp=sw["vegp"], other=other, snowenv=sw["snowenv"], dtm=dtm,
                res=1.0, tfact=0.02, af_wind=np.sqrt(wuv ** 2 + wvv ** 2), wsa_s=10)
    micro = {"obstime": sw["obstime"], "climdata": clim, "vegp": sw["vegp"], "other": sw["other"]}
    return sw, a, dtm, snow, micro


def _sub3(d, idx):
    """day-subset of a dict of vectors [T] and arrays [rows, cols, T]"""
    out = {}
    for k, v in d.items():
        v = np.asarray(v)
        out[k] = v[idx] if v.ndim == 1 else np.asfortranarray(v[:, :, idx]) if v.ndim == 3 and v.shape[2] >= idx.max() + 1 else v
    return out


def _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, solve, microsnow, oracle_merge):
    rows, cols = dtm.shape
    ni, si = _steps(ndays_), _steps(sdays)
    outm = [1] * 10 if reqhgt > 0 else [1 if i in (0, 3, 5, 6, 7, 8, 9) else 0 for i in range(10)]
    an = dict(a, obstime=_sub3(a["obstime"], ni), climdata=_sub3(a["climdata"], ni), pointm=_sub3(a["pointm"], ni))
    moutn = solve(an)
    if oracle_merge:
        from oracle import snowmerge_oracle as MO
        micro = MO.prep_micro(moutn, sdays + 1, ndays_ + 1, rows, cols)
    else:
        micro = {}
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
    mouts = microsnow(reqhgt, _sub3(sw["obstime"], si), _sub3(sw["climdata"], si), smods, micro, sw["vegp"], sw["other"], MAT, outm)
    for k in moutn:
        if k not in mouts:
            mouts[k] = micro[k]
    if oracle_merge:
        return MO.merge(moutn, mouts, sdays + 1, ndays_ + 1, rows, cols)
    return S.merge_snow_outputs(moutn, mouts, sdays + 1, ndays_ + 1, rows, cols)


@pytest.mark.parametrize("reqhgt,cold,doy", [(0.05, 0.0, 90), (0.3, -3.0, 30), (0.0, 3.0, 120)])
def test_array_weather_run_equals_the_host_orchestration_and_the_oracle_backed_one(oracle, reqhgt, cold, doy):
    sw, a, dtm, snow, micro = _case(reqhgt, cold, doy)
    got, smod = S.runmicrosnow2(a, snow, micro, MAT, want_smod=True)
    want_smod = S.snowmodel2_device(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], snow["other"], sw["snowenv"], dtm, 1.0, 0.02,
                                    af_wind=snow["af_wind"], wsa_s=10)
    for k in smod:
        assert np.array_equal(smod[k], want_smod[k], equal_nan=True), k
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
    assert sdays.size >= 2 and ndays_.size >= 2 and (sd | nd).all()
    want = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: runmicro2Cpp(*[an[k] for k in ARGS]), S.gridmicrosnow2, False)
    _close(got, want, 1e-12, "host-orchestrated HIP")
    want_o = _orchestrate(a, sw, dtm, smod, sdays, ndays_, reqhgt,
                          lambda an: oracle.run_grid(**{k: an[k] for k in ARGS}, array_forcing=True),
                          lambda *x: oracle.run_microsnow(*x, array_forcing=True), True)
    _close(got, want_o, 1e-6, "oracle-backed orchestration")
    covered = swe[:, :, _steps(sdays)] > 0
    assert covered.any() and (~covered & ~np.isnan(dtm)[:, :, None]).any()


def test_geometry_checks():
    from microclimf_amd import McfError
    sw, a, dtm, snow, micro = _case(0.05, 0.0, 90, rows=8, cols=6, ndays=5)
    v = synthetic.workload(8, 6, 120, reqhgt=0.05, zref=3.5)
    with pytest.raises((McfError, ValueError)):
        S.runmicrosnow1(v, snow, micro, MAT)                 # vector solver inputs, array snow model
    with pytest.raises(McfError, match="array weather"):
        S.runmicrosnow1(a, dict(snow), micro, MAT, n_blocks=2)      # row blocks: not with array weather
