"""runmicro_big() at GPU scale: a synthetic raster through the whole front end — point model, universal terrain on the
device, wetness index on the host, tiles solved straight into netCDF files.  python tools/big_rate.py [--n 2048 --days 30]"""
import argparse
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import frontend as F, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--days", type=int, default=30)
ap.add_argument("--tilesize", type=int, default=1000)
ap.add_argument("--vars", default="Tz")
ap.add_argument("--dir", default="/tmp")
a = ap.parse_args()
T = a.days * 24
obst, clim, _ = synthetic.forcing_vectors(T, 50.0, -5.0, 2023, 152, synthetic.SEED, 0.0)
weather = {"temp": clim["temp"], "relhum": np.clip(100 * clim["ea"] / clim["es"], 5, 100), "pres": clim["pres"],
           "swdown": clim["swdown"], "difrad": clim["difrad"], "lwdown": clim["lwdown"], "windspeed": clim["windspeed"],
           "winddir": clim["winddir"], "precip": np.zeros(T), "obstime": obst}
vegp0, soilc0, z = synthetic.rasters(a.n, a.n)
vegp = {k: vegp0[k] for k in F.VEG_KEYS}
soilc = {"soiltype": np.where(np.isnan(vegp["hgt"]), np.nan, 7.0), "groundr": soilc0["gref"]}
dtm = {"z": np.where(np.isnan(vegp["hgt"]), np.nan, z), "res": 1.0, "lat": 50.0, "long": -5.0, "xmin": 0.0, "ymax": float(a.n)}
out = tempfile.mkdtemp(dir=a.dir)
need = a.n * a.n * T * 4 * len(a.vars.split(","))
if need > 0.6 * shutil.disk_usage(a.dir).free:
    raise SystemExit("not enough room for the files")
t0 = time.perf_counter()
mp = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
t1 = time.perf_counter()
files = F.runmicro_big(mp, 0.05, out, vegp, soilc, dtm, tilesize=a.tilesize, vars=tuple(a.vars.split(",")))
t2 = time.perf_counter()
size = sum(Path(f).stat().st_size for f in files)
shutil.rmtree(out)
valid = int((~np.isnan(vegp["hgt"])).sum())
print(f"{a.n} x {a.n} cells x {T} h, variables {a.vars}: point model {t1 - t0:.2f} s, runmicro_big {t2 - t1:.2f} s -> {len(files)} files, "
      f"{size / 1e9:.2f} GB; {valid * T / (t2 - t0):.3e} cell-steps/s end to end")
