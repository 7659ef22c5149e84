"""vignettes/running-microclimf.Rmd:663-683 (images/image14p.png), the blue curve: climdata$temp - 12, the point model subset
to the coldest day of each month, runsnowmodel(method = "slow") with the default snow environment, raster-mean depth =
totalSWE / snowden over the 288 selected hours.  Read off the figure, month by month: 0.02, 0.35, 0.64 -> 0.68, 0.48,
0.20, 0.00, 0.08 -> 0.055, 0.00, 0.025, 0.01, 0.095, 0.375 m."""
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
mp = F.subsetpointmodel(F.runpointmodel(cold, 0.05, dtm, vegp, soilc), tstep="month", what="tmin")
for env, meth in (("Taiga", "slow"), ("Taiga", "fast")):
    smod = F.runsnowmodel(cold, mp, vegp, soilc, dtm, snowenv=env, method=meth)
    with np.errstate(invalid="ignore", divide="ignore"):
        depth = np.nanmean(smod["totalSWE"] / smod["snowden"], axis=(0, 1))
        swe = np.nanmean(smod["totalSWE"], axis=(0, 1))
    print(env, meth, "steps", depth.size, "days", sorted(set(zip(mp["obstime"]["month"].astype(int), mp["obstime"]["day"].astype(int)))))
    for m in range(12):
        d = depth[m * 24:(m + 1) * 24]
        print(f"  month {m + 1:2d}: depth {d[0]:.3f} -> {d[-1]:.3f} m (min {np.nanmin(d):.3f}, max {np.nanmax(d):.3f}); SWE {swe[m*24]:.1f} mm")
    np.savez_compressed(ROOT / "gpurun_out" / f"vignette_snow_subset_{meth}.npz", depth=depth, swe=swe)
