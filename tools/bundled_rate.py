"""End-to-end wall time of BASELINE.json configs[0] — runpointmodel() + runmicro() on the reference's bundled example data
(50 x 50 cells, 12 pai layers, 8760 hourly steps) through the host front end: python tools/bundled_rate.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
w96 = load(96)
mp = F.runpointmodel(w96[0], 0.05, dtm, vegp, soilc)
F.runmicro(mp, 0.05, vegp, soilc, dtm)                       # warm-up: library load, first kernels
for reqhgt in (0.05, 1.0, 0.0):
    t0 = time.perf_counter()
    mp = F.runpointmodel(weather, reqhgt, dtm, vegp, soilc)
    t1 = time.perf_counter()
    a = F.prepare_grid_inputs(mp, reqhgt, vegp, soilc, dtm)
    t2 = time.perf_counter()
    out = F.runmicro(mp, reqhgt, vegp, soilc, dtm)
    t3 = time.perf_counter()
    valid = int((~np.isnan(a["vegp"]["hgt"][:, :, 0])).sum())
    print(f"reqhgt {reqhgt}: runpointmodel {t1 - t0:.3f} s, input preparation (terrain on the device, wetness index, vegetation "
          f"layers) {t2 - t1:.3f} s, runmicro incl. the same preparation {t3 - t2:.3f} s; {valid} cells x 8760 h, "
          f"{len(out)} outputs = {sum(v.nbytes for v in out.values()) / 1e9:.2f} GB; whole chain {t1 - t0 + t3 - t2:.2f} s = "
          f"{valid * 8760 / (t1 - t0 + t3 - t2):.3e} cell-steps/s end to end")

# the reference's snow example (R/Cppwrappers.R:701-704: "takes ~90 seconds"): climdata$temp - 8, runsnowmodel for the year
wc = dict(weather, temp=weather["temp"] - 8.0)
mpc = F.runpointmodel(wc, 0.05, dtm, vegp, soilc)
F.runsnowmodel({k: (v[:240] if k != "obstime" else {q: x[:240] for q, x in v.items()}) for k, v in wc.items()},
               F.runpointmodel({k: (v[:240] if k != "obstime" else {q: x[:240] for q, x in v.items()}) for k, v in wc.items()},
                               0.05, dtm, vegp, soilc), vegp, soilc, dtm)          # warm-up
t0 = time.perf_counter()
sm = F.runsnowmodel(wc, mpc, vegp, soilc, dtm)
dt = time.perf_counter() - t0
d = np.nanmean(sm["groundsnowdepth"], axis=(0, 1))
print(f"runsnowmodel, bundled site at -8 K, 8760 h (73 five-day chunks): {dt:.2f} s; mean ground snow depth peaks at "
      f"{d.max():.3f} m on step {int(d.argmax())}, {int((d > 0).sum())} h with snow")

# the same on a subset micropoint (R/Cppwrappers.R:712-716: method = "slow" "takes ~90 seconds again", "fast" "~4 seconds")
mpsub = F.subsetpointmodel(mpc)
for meth in ("slow", "fast"):
    t0 = time.perf_counter()
    sm = F.runsnowmodel(wc, mpsub, vegp, soilc, dtm, method=meth)
    dt = time.perf_counter() - t0
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.nanmean(sm["totalSWE"] / sm["snowden"], axis=(0, 1))
    print(f"runsnowmodel(method = \"{meth}\") on the monthly subset (12 days of 24 h returned): {dt:.2f} s; mean depth peaks at "
          f"{np.nanmax(d):.3f} m")

# the array-weather routes on the same site: a 2 x 2 climate grid (the vignette's dummy arrays, Rmd:560-577), cold year
cr = cc = 2
arr = {k: np.asfortranarray(np.broadcast_to(wc[k][None, None, :], (cr, cc, len(wc[k]))).copy()) for k in F.WEATHER}
cl = dtm["lat"] + 1e-4 * np.arange(cr)[:, None] + 0 * np.arange(cc)[None, :]
co = dtm["long"] + 1e-4 * np.arange(cc)[None, :] + 0 * np.arange(cr)[:, None]
la = dtm["lat"] + 9e-6 * np.arange(50)[::-1, None] + 0 * np.arange(50)[None, :]
lo = dtm["long"] + 1.4e-5 * np.arange(50)[None, :] + 0 * np.arange(50)[:, None]
zz = np.asarray(dtm["z"])
dc = np.array([[np.nanmean(zz[:25, :25]), np.nanmean(zz[:25, 25:])], [np.nanmean(zz[25:, :25]), np.nanmean(zz[25:, 25:])]])
t0 = time.perf_counter()
mpa = F.runpointmodela(arr, wc["obstime"], 0.05, dtm, vegp, soilc, lats=cl, lons=co)
t1 = time.perf_counter()
sma = F.runsnowmodela(arr, wc["obstime"], mpa, vegp, soilc, dtm, dtmc=dc, lats_c=cl, lons_c=co, lats=la, lons=lo)
t2 = time.perf_counter()
mo = F.runmicro_snow_array(mpa, cr, cc, 0.05, vegp, soilc, dtm, sma, dtmc=dc, lats=la, lons=lo)
t3 = time.perf_counter()
mpas = F.subsetpointmodela(mpa)
t4 = time.perf_counter()
smf = F.runsnowmodela(arr, wc["obstime"], mpas, vegp, soilc, dtm, dtmc=dc, lats_c=cl, lons_c=co, lats=la, lons=lo, method="fast")
t5 = time.perf_counter()
print(f"array weather, 2 x 2 climate cells, 8760 h: runpointmodela {t1 - t0:.2f} s, runsnowmodel (.snowmodel2; resampling on the "
      f"host, terrain / gridmodelsnow2 / position index on the device) {t2 - t1:.2f} s, runmicro(snow = TRUE) (.runmicrosnow2) "
      f"{t3 - t2:.2f} s, Tz mean {np.nanmean(mo['Tz']):.2f} degC; on the monthly subset runsnowmodel(method = \"fast\") "
      f"(.snowmodelq2) {t5 - t4:.2f} s")

# the reference's runmicro() example (R/Cppwrappers.R:355-362, "takes ~20 seconds" for two calls): the point model subset
# to the hottest day of each month, then runmicro at 5 cm and at 1 m
mps = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc))
t0 = time.perf_counter()
o1 = F.runmicro(mps, 0.05, vegp, soilc, dtm)
o2 = F.runmicro(mps, 1.0, vegp, soilc, dtm)
print(f"two runmicro() calls on the monthly subset (50 x 50 cells x 288 h each, incl. terrain / wetness-index preparation): "
      f"{time.perf_counter() - t0:.3f} s")

# runbioclim() on the bundled year (vignettes/running-microclimf.Rmd:648: "~20 seconds")
F.runbioclim(weather, 0.05, vegp, soilc, dtm)
t0 = time.perf_counter()
b = F.runbioclim(weather, 0.05, vegp, soilc, dtm)
print(f"runbioclim(), 19 layers from 14 modelled days: {time.perf_counter() - t0:.3f} s; bio1 mean {np.nanmean(b['bio1']):.2f} degC")
