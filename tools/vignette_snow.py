"""The reference vignette's snow example (vignettes/running-microclimf.Rmd:685-703, images/image14a.png) through the front
end: climdata$temp - 12, runsnowmodel(..., snowenv = "Maritime") for the year on the bundled site; raster means of snow
water equivalent and of depth = SWE / density over time.  The published curves: SWE peaks near 190 mm and depth near
0.53 m in early April, the pack is gone by early June, and builds again to about 115 mm / 0.34 m by 31 December."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
cold = dict(weather, temp=weather["temp"] - 12.0)
mp = F.runpointmodel(cold, 0.05, dtm, vegp, soilc)
smod = F.runsnowmodel(cold, mp, vegp, soilc, dtm, snowenv="Maritime")
with np.errstate(invalid="ignore", divide="ignore"):
    swe = np.nanmean(smod["totalSWE"], axis=(0, 1))
    depth = np.nanmean(smod["totalSWE"] / smod["snowden"], axis=(0, 1))
ob = weather["obstime"]
k = int(np.argmax(swe))
print(f"SWE peak {swe[k]:.1f} mm on {int(ob['month'][k]):02d}-{int(ob['day'][k]):02d}; depth peak {np.nanmax(depth):.3f} m; "
      f"on 31 Dec: SWE {swe[-1]:.1f} mm, depth {depth[-1]:.3f} m")
for m, d in ((2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (6, 15), (7, 1), (8, 15), (10, 1), (11, 15), (12, 15)):
    i = int(np.nonzero((ob["month"] == m) & (ob["day"] == d) & (ob["hour"] == 12))[0][0])
    print(f"  {m:02d}-{d:02d}: SWE {swe[i]:7.2f} mm, depth {depth[i]:.3f} m")
np.savez_compressed(ROOT / "gpurun_out" / "vignette_snow.npz", swe=swe, depth=depth)
