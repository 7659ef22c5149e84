"""vignettes/running-microclimf.Rmd:707-729 (images/image14b.png): the monthly-minimum subset, runmicro with and without snow;
raster means of Tz and soilm over the 288 steps.  The text says `climdata$temp - 12`; the published figure is met with - 8 K
(the offset of the help-file examples, R/Cppwrappers.R:701), see tools/probe_image14b.py.  usage: [what [offset]]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
offset = float(sys.argv[2]) if len(sys.argv) > 2 else -8.0
cold = dict(weather, temp=weather["temp"] + offset)
mp = F.subsetpointmodel(F.runpointmodel(cold, 0.05, dtm, vegp, soilc), tstep="month", what=(sys.argv[1] if len(sys.argv) > 1 else "tmin"))
smod = F.runsnowmodel(cold, mp, vegp, soilc, dtm, snowenv="Maritime", method="slow")
m1 = F.runmicro_snow(mp, 0.05, vegp, soilc, dtm, smod)
m2 = F.runmicro(mp, 0.05, vegp, soilc, dtm)
with np.errstate(invalid="ignore"):
    tz1, tz2 = np.nanmean(m1["Tz"], axis=(0, 1)), np.nanmean(m2["Tz"], axis=(0, 1))
    s1, s2 = np.nanmean(m1["soilm"], axis=(0, 1)), np.nanmean(m2["soilm"], axis=(0, 1))
print("Tz with snow   : min %.2f max %.2f at %d" % (tz1.min(), tz1.max(), int(tz1.argmax()) + 1))
print("Tz without snow: min %.2f max %.2f at %d" % (tz2.min(), tz2.max(), int(tz2.argmax()) + 1))
for i in (12, 36, 60, 85, 90, 110, 134, 182, 197, 205, 220, 230, 254, 278):
    print(f"  step {i:3d}: Tz snow {tz1[i - 1]:6.2f}  no snow {tz2[i - 1]:6.2f}   soilm snow {s1[i - 1]:.3f}  no snow {s2[i - 1]:.3f}")
np.savez_compressed(ROOT / "gpurun_out" / "vignette_snow_micro.npz", tz1=tz1, tz2=tz2, s1=s1, s2=s2)
