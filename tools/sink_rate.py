"""Output-sink rates on one GPU: plain fp64 fetch vs the writetonc-packed int32 fetch of one 5-day ring slot
(python tools/sink_rate.py [--rows 1024 --cols 1024])."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--cols", type=int, default=1024)
ap.add_argument("--days", type=int, default=5)
a = ap.parse_args()
T = a.days * 24
w = synthetic.workload(a.rows, a.cols, T, reqhgt=0.05)
with Plan(w["obstime"], w["climdata"], w["pointm"], w["vegp"], w["soilc"], w["reqhgt"], w["zref"], w["lat"], w["lon"],
          w["Sminp"], w["Smaxp"], w["tfact"], True, w["mat"], w["out"], ring_days=a.days) as p:
    p.run_days(0, a.days)
    p.sync()
    n = a.rows * a.cols * T
    p.fetch(0, "Tz", 0, T); p.fetch_packed(0, "Tz", 0, T)          # warm-up (page faults of the host buffers)
    t = time.time(); p.fetch(0, "Tz", 0, T); dt = time.time() - t
    print(f"plain fetch  : {n * 8 / 1e9:.2f} GB in {dt:.3f} s = {n * 8 / dt / 1e9:.1f} GB/s, {n / dt:.3e} values/s")
    t = time.time(); _, ms = p.fetch_packed(0, "Tz", 0, T, timing=True); dt = time.time() - t
    print(f"packed fetch : {n * 4 / 1e9:.2f} GB in {dt:.3f} s = {n / dt:.3e} values/s; pack kernel {ms:.3f} ms = "
          f"{n * 12 / (ms * 1e-3) / 1e9:.0f} GB/s of HBM traffic (8 B read + 4 B written per value)")
