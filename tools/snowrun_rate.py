"""What a host session gets from ONE call of mcf_runmicrosnow1 (`runmicro(..., snow = TRUE)`, data.frame weather): a synthetic
raster for a whole year, `Tz` only, into a numpy array — PCIe and the host's page faults included.
    python tools/snowrun_rate.py [--rows 512 --cols 512 --days 365 --out Tz,relhum]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import snow as S, synthetic  # noqa: E402

NAMES = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=512)
ap.add_argument("--cols", type=int, default=512)
ap.add_argument("--days", type=int, default=365)
ap.add_argument("--out", type=str, default="Tz")
ap.add_argument("--cold", type=float, default=0.0)
ap.add_argument("--keep-gb", type=float, default=0.0, help="mcf_snowrun_keep: pass 1's snow chunks stay in HBM up to this much; the "
                                                           "handle then runs a SECOND year, which finds the sets pooled")
a = ap.parse_args()
T = a.days * 24
want = a.out.split(",")
out = [1 if n in want else 0 for n in NAMES]
sw = synthetic.snow_workload(a.rows, a.cols, T, cold=a.cold, zref=3.5, start_doy=1)
g = synthetic.workload(a.rows, a.cols, T, reqhgt=0.05, zref=3.5, hgt_range=(0.05, 3.0), start_doy=1, variety=True, out=out)
_, _, dtm = synthetic.rasters(a.rows, a.cols)
dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
snow = dict(sw, dtm=dtm, res=1.0, tfact=0.02)
micro = {"obstime": sw["obstime"], "climdata": sw["climdata"], "vegp": sw["vegp"], "other": sw["other"]}
S.runmicrosnow1(dict(g, obstime={k: v[:240] for k, v in g["obstime"].items()}, climdata={k: v[:240] for k, v in g["climdata"].items()},
                     pointm={k: v[:240] for k, v in g["pointm"].items()}),
                dict(snow, obstime={k: v[:240] for k, v in sw["obstime"].items()}, climdata={k: v[:240] for k, v in sw["climdata"].items()},
                     pointm={k: v[:240] for k, v in sw["pointm"].items()}),
                {"obstime": {k: v[:240] for k, v in sw["obstime"].items()}, "climdata": {k: v[:240] for k, v in sw["climdata"].items()},
                 "vegp": sw["vegp"], "other": sw["other"]}, 7.5)          # warm-up: library, clocks
print("inputs made, library warm", flush=True)
t = time.perf_counter()
with S.SnowRun(g, snow) as run:
    sd, nd = run.pass1()
    t1 = time.perf_counter()
    got = run.pass2(micro, 7.5)
    st = run.stats()
dt = time.perf_counter() - t
valid = int(np.isfinite(dtm).sum())
gb = sum(v.nbytes for v in got.values()) / 1e9
print(f"{a.rows} x {a.cols} x {a.days} days, outputs {want}: {dt:.2f} s ({t1 - t:.2f} s pass 1) = {valid * T / dt:.3e} cell-steps/s, "
      f"{gb:.1f} GB into host arrays; snow days {int(sd.sum())}, no-snow days {int(nd.sum())}, {st}", flush=True)
if a.keep_gb > 0:
    # (an output array of a 1024^2 year is 73.5 GB of host memory: only ONE set is alive at a time — the unkept run's is reduced
    # to a 64-bit digest per variable first; the box allows a command 270 GB)
    def digest(d):       # (xor and wrapping sum of the 64-bit patterns: bitwise equality up to collisions, at memory speed)
        out = {}
        for k, v in d.items():
            u = np.ascontiguousarray(v).reshape(-1).view(np.uint64)
            out[k] = (int(np.bitwise_xor.reduce(u)), int(np.add.reduce(u, dtype=np.uint64)))
        return out
    ref = digest(got)
    del got
    with S.SnowRun(g, snow) as run:
        run.keep(a.keep_gb)
        for year in (1, 2):
            t = time.perf_counter()
            sd, nd = run.pass1()
            t1 = time.perf_counter()
            got2 = run.pass2(micro, 7.5)
            dt = time.perf_counter() - t
            st2 = run.stats()
            same = digest(got2) == ref
            del got2
            print(f"  keep {a.keep_gb:g} GB, year {year} of one handle: {dt:.2f} s ({t1 - t:.2f} s pass 1) = {valid * T / dt:.3e} cell-steps/s, {st2}; "
                  f"outputs bitwise the unkept run's: {same}", flush=True)
