"""writetonc sink rate on one GPU: a ring slot of `--days` days of every default variable written to a netCDF file
(python tools/nc_rate.py [--rows 1024 --cols 1024 --days 2 --dir /tmp])."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import ncsink, synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--cols", type=int, default=1024)
ap.add_argument("--days", type=int, default=2)
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--format", default="classic", choices=("classic", "netcdf4"))
ap.add_argument("--deflate", type=int, default=0, help="netcdf4: 0 = the reference's level 9, 1..9, -1 = none")
a = ap.parse_args()
T = a.days * 24
w = synthetic.workload(a.rows, a.cols, T, reqhgt=0.05)
east, north = ncsink.coords_from_extent(0, a.cols * 1.0, 0, a.rows * 1.0, 1.0)
names = ncsink.default_vars(0.05)
for where in ((a.dir, "/dev/shm") if a.format == "classic" else (a.dir,)):
    path = os.path.join(where, "mcf_nc_rate.nc")
    with Plan(w["obstime"], w["climdata"], w["pointm"], w["vegp"], w["soilc"], w["reqhgt"], w["zref"], w["lat"], w["lon"],
              w["Sminp"], w["Smaxp"], w["tfact"], True, w["mat"], w["out"], ring_days=a.days) as p:
        p.run_days(0, a.days)
        p.sync()
        with ncsink.NcWriter(path, a.rows, a.cols, np.arange(T) + 473352.0, east, north, 0.05, names,
                             format=a.format, deflate_level=a.deflate) as nc:
            t = time.time()
            ms = nc.write_plan(p, 0, 0, 0, T, timing=True)
            dt = time.time() - t
        n = a.rows * a.cols * T * len(names)
        size = os.path.getsize(path)
        os.remove(path)
        raw = n * 4
        print(f"{where} [{a.format}{'' if a.format == 'classic' else ' deflate ' + str(a.deflate or 9)}]: {raw / 1e9:.2f} GB of int32 -> "
              f"{size / 1e9:.2f} GB ({raw / size:.2f}:1), {raw / dt / 1e9:.2f} GB/s of values")
        print(f"{where}: {len(names)} variables x {T} steps x {a.rows}x{a.cols}: {size / 1e9:.2f} GB file in {dt:.3f} s = "
              f"{size / dt / 1e9:.2f} GB/s, {n / dt:.3e} values/s; k_pack_nc {ms:.3f} ms = {n * 12 / (ms * 1e-3) / 1e9:.0f} GB/s "
              f"of HBM traffic (8 B read + 4 B written per value)")
