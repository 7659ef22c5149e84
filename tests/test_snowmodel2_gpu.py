"""mcf_snowmodel2 (include/mcf.h): `.snowmodel2`'s chunk loop (R/internal.R:2950-3008) device-resident, given array weather at
the raster's resolution.  Held against
  (1) oracle/snowarray_oracle.py (its terrain, gridmodelsnow2 and `.tpicalc` restatements) fed the SAME fine arrays through an
      identity coarse grid: 1e-6;
  (2) the product's host-orchestrated loop (snow.snowmodel2_chunks: one-shot gridmodelsnow2, terrain and tpi calls per chunk,
      whole-series host arrays) on the same identity grid: 1e-9 (HIP vs HIP; the loop's orchestration is what differs)."""
import numpy as np
import pytest

from microclimf_amd import snow as S
from microclimf_amd import synthetic

pytestmark = pytest.mark.gpu


def _case(rows=24, cols=17, ndays=12, cold=2.0, doy=40):
    T = ndays * 24
    sw = synthetic.snow_workload(rows, cols, T, array_forcing=True, cold=cold, zref=3.5, start_doy=doy)
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    clim = dict(sw["climdata"])
    wd = np.asarray(clim["winddir"], dtype=np.float64) * np.pi / 180
    wu, wv = clim["windspeed"] * np.cos(wd), clim["windspeed"] * np.sin(wd)
    wuv, wvv = np.nanmean(wu, axis=(0, 1)), np.nanmean(wv, axis=(0, 1))
    af_wind = np.sqrt(wuv ** 2 + wvv ** 2)
    other = {k: sw["other"][k] for k in ("zref", "lats", "lons", "isnowdc", "isnowdg", "isnowac", "isnowag")}
    return sw, clim, dtm, af_wind, other


def _close(got, want, tol, what):
    for k in ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden"):
        g, w = got[k], want[k]
        assert np.array_equal(np.isnan(g), np.isnan(w)), (what, k)
        fin = np.isfinite(w)
        err = float(np.max(np.abs(g[fin] - w[fin]) / (1 + np.abs(w[fin])))) if fin.any() else 0.0
        assert err < tol, (what, k, err)


@pytest.mark.parametrize("wsa_s,res", [(10, 1.0), (1, 1.0), (0, 120.0)])
def test_chunk_loop_with_array_weather(oracle, wsa_s, res):
    from oracle import snowarray_oracle as SA
    sw, clim, dtm, af_wind, other = _case()
    R, Cc = dtm.shape
    got = S.snowmodel2_device(sw["obstime"], clim, sw["pointm"], sw["vegp"], other, sw["snowenv"], dtm, res, 0.02, af_wind=af_wind,
                              wsa_s=wsa_s)
    assert np.nanmax(got["groundsnowdepth"]) > 0.01 and got["Tc"].shape == (R, Cc, 12 * 24)
    assert np.all(np.isnan(got["Tc"][:, :, 240:]))                   # `1:n5days` truncates: the last two days belong to no chunk
    agg = wsa_s if wsa_s else (10 if res <= 100 else 1)
    rp, cp = np.arange(R, dtype=np.float64), np.arange(Cc, dtype=np.float64)
    pm = dict(sw["pointm"], tr=sw["pointm"]["umu"])                  # (`tr` only rides along)
    want = SA.snowmodel2_chunks(sw["obstime"], clim, pm, sw["vegp"], other, sw["snowenv"], dtm, np.nan_to_num(dtm), res, 0.02, rp, cp,
                                altcorrect=0, agg=agg)
    _close(got, want, 1e-6, "oracle")
    host = S.snowmodel2_chunks(sw["obstime"], clim, pm, sw["vegp"], other, sw["snowenv"], dtm, np.nan_to_num(dtm), res, 0.02, rowpos=rp,
                               colpos=cp, altcorrect=0, agg=agg)
    _close(got, host, 1e-9, "host-orchestrated HIP")


def test_argument_checks():
    from microclimf_amd import McfError
    sw, clim, dtm, af_wind, other = _case(8, 6, 5)
    with pytest.raises((McfError, ValueError)):            # (the Python mirror's shape check comes first; the C entry says "array weather")
        S.snowmodel1_chunks(sw["obstime"], clim, sw["pointm"], sw["vegp"], dict(other, lat=50.0, lon=0.0), sw["snowenv"], dtm, 1.0)
