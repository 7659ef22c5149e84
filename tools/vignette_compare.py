"""GPU box: the front end's series for every digitised figure against tests/golden/vignette_points.json, distances in pixels.
python tools/vignette_compare.py  -> prints one line per curve (and writes gpurun_out/vignette_compare.txt)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402
import vignette_fixture as V  # noqa: E402

lines = []


def report(fig, k, curve, x, y):
    p = V.panel(fig, k)
    d = V.distances(p, curve, x, y)
    s = (f"{fig}[{k}] {curve:8s} fig->model max {d['fig_to_model_max']:6.2f} p99 {d['fig_to_model_p99']:6.2f} | model->fig max "
         f"{d['model_to_fig_max']:6.2f} p99 {d['model_to_fig_p99']:6.2f} px   (1 px = {p['px']['x']:.4g} x {p['px']['y']:.4g}, "
         f"{d['fig_pixels']} px)")
    print(s, flush=True)
    lines.append(s)
    return d


def flat_site(pai, hgt):
    _, _, soilc, dtm = load()
    one = np.ones((5, 5))
    vegp2 = {"pai": pai * one, "hgt": hgt * one, "x": one, "gsmax": 0.1 * one, "leafr": 0.3 * one, "clump": 0 * one,
             "leafd": 0.05 * one, "leaft": 0.15 * one}
    return {"z": 0 * one, "res": 10.0, "lat": dtm["lat"], "long": dtm["long"]}, vegp2, {"soiltype": 7 * one, "groundr": 0.15 * one}


weather, vegp, soilc, dtm = load()
# image7 / image8: height profiles
dem, vegp2, soilc2 = flat_site(0.05, 0.005)
mp = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dem, vegp2, soilc2), tstep="month", what="tmax")
hs = [0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0]
t = [F.runmicro(mp, h, vegp2, soilc2, dem)["Tz"][1, 1, 131] for h in hs]
report("image7", 0, "profile", t, hs)
dem, vegp2, soilc2 = flat_site(3.0, 10.0)
mp = F.subsetpointmodel(F.runpointmodel(weather, 10.0, dem, vegp2, soilc2), tstep="month", what="tmax")
heights = 10 ** (np.arange(-10, 11) / 10)
t = np.array([F.runmicro(mp, float(h), vegp2, soilc2, dem)["Tz"][1, 1, 131] for h in heights])
report("image8", 0, "profile", t, heights)
# image9: soil temperatures over the year
for depth, curve in ((-0.05, "d005"), (-0.2, "d020"), (-1.0, "d100")):
    mpd = F.runpointmodel(weather, depth, dem, vegp2, soilc2)
    tz = F.runmicro(mpd, depth, vegp2, soilc2, dem)["Tz"][1, 1, :]
    report("image9", 0, curve, np.arange(1, tz.size + 1), tz)
# image1b: the point model's three temperatures (envelope of everything drawn)
mp0 = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
print("point model keys:", sorted(mp0.keys()) if isinstance(mp0, dict) else type(mp0))
# image14a: snow year
cold = dict(weather, temp=weather["temp"] - 12.0)
mpc = F.runpointmodel(cold, 0.05, dtm, vegp, soilc)
smod = F.runsnowmodel(cold, mpc, vegp, soilc, dtm, snowenv="Maritime")
with np.errstate(invalid="ignore", divide="ignore"):
    swe = np.nanmean(smod["totalSWE"], axis=(0, 1))
    depth = np.nanmean(smod["totalSWE"] / smod["snowden"], axis=(0, 1))
hours = np.arange(swe.size, dtype=float)
report("image14a", 0, "swe", hours, swe)
report("image14a", 1, "depth", hours, depth)
# image14p: subset snow, slow and fast
mps = F.subsetpointmodel(mpc, tstep="month", what="tmin")
for method, curve in (("slow", "slow"), ("fast", "fast")):
    sm = F.runsnowmodel(cold, mps, vegp, soilc, dtm, method=method)
    with np.errstate(invalid="ignore", divide="ignore"):
        dep = np.nanmean(sm["totalSWE"] / sm["snowden"], axis=(0, 1))
    report("image14p", 0, curve, np.arange(1, dep.size + 1), dep)
# image14b: runmicro with / without snow at -8 K (see the test's docstring) and at the text's -12 K
for off in (8.0, 12.0):
    c2 = dict(weather, temp=weather["temp"] - off)
    mp2 = F.subsetpointmodel(F.runpointmodel(c2, 0.05, dtm, vegp, soilc), tstep="month", what="tmin")
    sm2 = F.runsnowmodel(c2, mp2, vegp, soilc, dtm, snowenv="Maritime", method="slow")
    m1 = F.runmicro_snow(mp2, 0.05, vegp, soilc, dtm, sm2)
    m2 = F.runmicro(mp2, 0.05, vegp, soilc, dtm)
    with np.errstate(invalid="ignore"):
        tz1, tz2 = np.nanmean(m1["Tz"], axis=(0, 1)), np.nanmean(m2["Tz"], axis=(0, 1))
        s1, s2 = np.nanmean(m1["soilm"], axis=(0, 1)), np.nanmean(m2["soilm"], axis=(0, 1))
    idx = np.arange(1, tz1.size + 1)
    print(f"-- image14b with climdata$temp - {off:g}")
    report("image14b", 0, "nosnow", idx, tz2)
    report("image14b", 0, "snow", idx, tz1)
    report("image14b", 1, "nosnow", idx, s2)
    report("image14b", 1, "snow", idx, s1)
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
(out / "vignette_compare.txt").write_text("\n".join(lines) + "\n")
