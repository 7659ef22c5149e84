import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from bundled import load
from microclimf_amd import frontend as F
weather, vegp, soilc, dtm = load()
print("figure (red, no snow): Tz 12->6.5 36->-0.5 85->11 110->14 134->27.5 182->22.5 230->8 278->5.5 ; soilm 60->0.41 85->0.295 134->0.295 182->0.24 205->0.275 230->0.32 254->0.37 278->0.40")
for off in (0.0, -8.0, -12.0):
    w = dict(weather, temp=weather["temp"] + off)
    mp0 = F.runpointmodel(w, 0.05, dtm, vegp, soilc)
    for what in ("tmin", "tmax", "tmedian"):
        mp = F.subsetpointmodel(mp0, tstep="month", what=what)
        m = F.runmicro(mp, 0.05, vegp, soilc, dtm)
        with np.errstate(invalid="ignore"):
            tz = np.nanmean(m["Tz"], axis=(0, 1)); sm = np.nanmean(m["soilm"], axis=(0, 1))
        print(f"off {off:5.1f} {what:7s} Tz", " ".join(f"{i}:{tz[i-1]:.1f}" for i in (12, 36, 85, 110, 134, 182, 230, 278)),
              "| soilm", " ".join(f"{i}:{sm[i-1]:.3f}" for i in (60, 85, 134, 182, 205, 230, 254, 278)))
