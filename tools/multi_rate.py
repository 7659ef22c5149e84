"""Rate of the one-process multi-device entries on whatever devices the box has (not part of bench.py's contract):
  python tools/multi_rate.py [--rows 2048 --cols 2048 --tsteps 240 --devices 0,1,2,3 --blocks 0 --what solver,terrain,snow,bioclim]
One host process, a host thread per device (include/mcf.h mcf_runmicro1_multi, mcf_precompute_terrain_multi,
mcf_snowmodel1_multi, mcf_runbioclim1_multi).  The one-shot entries move their inputs and outputs over PCIe, so these are
PCIe-inclusive rates; with one device listed the line is the single-device call through the same driver.  No multi-GPU node
was available to the rounds that wrote this: on such a node it is the first thing to run."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import _abi, synthetic  # noqa: E402
from microclimf_amd.api import runbioclim1Cpp, runmicro1Cpp  # noqa: E402
from microclimf_amd.snow import snowmodel1_chunks  # noqa: E402
from microclimf_amd.terrain import precompute_terrain  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2048)
ap.add_argument("--cols", type=int, default=2048)
ap.add_argument("--tsteps", type=int, default=240)
ap.add_argument("--devices", default="", help="comma-separated HIP ordinals; empty = every visible device")
ap.add_argument("--blocks", type=int, default=0)
ap.add_argument("--what", default="solver,terrain,snow,bioclim")
a = ap.parse_args()
lib = _abi.load()
ndev = lib.mcf_device_count()
devs = [int(d) for d in a.devices.split(",") if d != ""] or list(range(ndev))
print(f"{ndev} visible device(s); using {devs}, {a.blocks or len(devs)} row blocks")
R, C, T = a.rows, a.cols, a.tsteps
what = set(a.what.split(","))


def timed(label, units, fn):
    fn()                                     # warm-up: library load, first allocations
    t = time.perf_counter()
    fn()
    dt = time.perf_counter() - t
    print(f"{label:9s} {dt:8.3f} s  {units / dt:.3e} {'cell-steps' if label != 'terrain' else 'cells'}/s")


if "solver" in what:
    w = synthetic.workload(R, C, T, reqhgt=0.05, out=[1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    keys = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact", "complete", "mat", "out")
    timed("solver", R * C * T, lambda: runmicro1Cpp(*[w[k] for k in keys], devices=devs, n_blocks=a.blocks))
if "terrain" in what:
    _, _, dtm = synthetic.rasters(R, C)
    timed("terrain", R * C, lambda: precompute_terrain(dtm, 1.0, 2.0, devices=devs, n_blocks=a.blocks))
if "snow" in what:
    Ts = max(120, (T // 120) * 120)
    sw = synthetic.snow_workload(R, C, Ts, cold=3.0, zref=3.5)
    _, _, dtm = synthetic.rasters(R, C)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    timed("snow", R * C * Ts, lambda: snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"],
                                                      dtm, 1.0, 0.02, devices=devs, n_blocks=a.blocks))
if "bioclim" in what:
    Tb = 336 + 4 * 72
    b = synthetic.workload(R, C, Tb, reqhgt=0.05)
    for k in ("complete", "out"):
        b.pop(k)
    q = [np.arange(336 + 72 * i, 336 + 72 * (i + 1)) for i in range(4)]
    timed("bioclim", R * C * Tb, lambda: runbioclim1Cpp(**b, out=[1] * 19, wetq=q[0], dryq=q[1], hotq=q[2], colq=q[3], air=True,
                                                        devices=devs, n_blocks=a.blocks))
