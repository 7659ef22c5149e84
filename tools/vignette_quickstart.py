"""The reference vignette's quick start (vignettes/running-microclimf.Rmd:113-126) on the bundled data through the front
end: the two maps of images/image1a.png — air temperature 5 cm above ground on the hottest hour of the year, and the mean
of the monthly maximum and minimum days — summarised as ranges for comparison with the published colour scales
(left ~26..53 degC, right ~10.2..14.1 degC).  python tools/vignette_quickstart.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
mp = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
mx = F.subsetpointmodel(mp, tstep="month", what="tmax")
mn = F.subsetpointmodel(mp, tstep="month", what="tmin")
mout_mx = F.runmicro(mx, 0.05, vegp, soilc, dtm)
mout_mn = F.runmicro(mn, 0.05, vegp, soilc, dtm)
ob = mx["obstime"]
k = 133                                                       # mout_mx$Tz[,,134]
print(f"step 134 of the tmax subset: {int(ob['year'][k])}-{int(ob['month'][k]):02d}-{int(ob['day'][k]):02d} {int(ob['hour'][k]):02d}:00 "
      f"(the vignette: 2017-06-20 13:00)")
hot = mout_mx["Tz"][:, :, k]
mairt = ((mout_mn["Tz"] + mout_mx["Tz"]) / 2).mean(axis=2)
for name, a, pub in (("Tz on the hottest hour", hot, "26 .. 53"), ("mean of monthly max and min days", mairt, "10.2 .. 14.1")):
    q = np.nanpercentile(a, [0, 1, 50, 99, 100])
    print(f"{name}: min {q[0]:.2f}, 1 % {q[1]:.2f}, median {q[2]:.2f}, 99 % {q[3]:.2f}, max {q[4]:.2f} degC "
          f"(published colour scale: {pub}); NA cells {int(np.isnan(a).sum())}")
na = np.isnan(hot)
print("NA block in the south-west corner (the white area of the figure):", bool(na[38:, :12].mean() > 0.8), "share of NA there", float(na[38:, :12].mean()))
np.savez_compressed(ROOT / "gpurun_out" / "vignette_maps.npz", hot=hot, mairt=mairt)
