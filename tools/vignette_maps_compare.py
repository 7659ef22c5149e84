"""GPU box: the front end's rasters for every digitised MAP of the reference's vignette against tests/golden/vignette_points.json
(tests/vignette_fixture.py map_compare: distance of each model cell from the published cell's colour class, in class widths).
python tools/vignette_maps_compare.py  -> one line per map (and gpurun_out/vignette_maps_compare.txt)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402
import vignette_fixture as V  # noqa: E402

lines = []


def report(fig, k, raster):
    m = V.map_panel(fig, k)
    d = V.map_compare(m, raster)
    s = (f"{fig}[{k}] {m['what'][:52]:52s} NA equal {d['na_equal']}  outside class by (class widths of {d['class_width']:.3g}): median "
         f"{d['median']:.2f} p90 {d['p90']:.2f} p99 {d['p99']:.2f} max {d['max']:.2f}; within 1: {100 * d['within_1']:.1f} % within 3: "
         f"{100 * d['within_3']:.1f} %; range model {d['model_min']:.5g} .. {d['model_max']:.5g} figure {d['legend_min']:.5g} .. {d['legend_max']:.5g}")
    print(s, flush=True)
    lines.append(s)
    return d


weather, vegp, soilc, dtm = load()
mp = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
mx, mn = F.subsetpointmodel(mp, what="tmax"), F.subsetpointmodel(mp, what="tmin")
mo = F.runmicro(mx, 0.05, vegp, soilc, dtm)
tmx, tmn = mo["Tz"], F.runmicro(mn, 0.05, vegp, soilc, dtm)["Tz"]
report("image1a", 0, tmx[:, :, 133])
report("image1a", 1, ((tmn + tmx) / 2).mean(axis=2))
report("image6", 0, tmx[:, :, 133])
report("image2", 0, mo["soilm"][:, :, 133])
with np.errstate(invalid="ignore"):
    report("image3b", 0, (mo["Rdirdown"] + mo["Rdifdown"])[:, :, 130])
report("image3b", 1, mo["Rswup"][:, :, 130])
report("image4", 0, mo["windspeed"][:, :, 99])
report("image5", 0, F.runmicro(mx, 0.0, vegp, soilc, dtm)["Tz"][:, :, 133])
# image10: the monthly-tmax run written by writetonc and read back (packed int / 100): layer 12
mpm = F.subsetpointmodel(mp, tstep="month", what="tmax")
mout = F.runmicro(mpm, 0.05, vegp, soilc, dtm)
report("image10", 0, np.rint(mout["Tz"][:, :, 11] * 100) / 100)
report("image11", 0, F.runbioclim(weather, 0.05, vegp, soilc, dtm, temp="air")["bio12"])
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
(out / "vignette_maps_compare.txt").write_text("\n".join(lines) + "\n")
