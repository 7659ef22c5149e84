"""The component figures of the reference vignette (vignettes/running-microclimf.Rmd:322-395, images/image2..5; drawn there
with the package's R-language model path) from the solver's outputs on the same monthly-maximum subset: soil moisture and
ground temperature on the hottest hour (step 134), short-wave fluxes at 10:00 on 20 June (step 131), wind speed at step 100.
Published colour scales: image3b downward short wave ~50..950, upward ~10..285 W/m2; image4 wind ~0.15..3.15 m/s;
image5 soil surface temperature ~25..59 degC."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
mx = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc), what="tmax")
m = F.runmicro(mx, 0.05, vegp, soilc, dtm)
g = F.runmicro(mx, 0.0, vegp, soilc, dtm)
maps = {"soilm[134]": m["soilm"][:, :, 133], "Rdown[131]": (m["Rdirdown"] + m["Rdifdown"])[:, :, 130], "Rswup[131]": m["Rswup"][:, :, 130],
        "windspeed[100]": m["windspeed"][:, :, 99], "Tg[134]": g["Tz"][:, :, 133], "Tz[134]": m["Tz"][:, :, 133]}
for k, a in maps.items():
    q = np.nanpercentile(a, [0, 1, 50, 99, 100])
    print(f"{k:15s} min {q[0]:8.3f}  1% {q[1]:8.3f}  median {q[2]:8.3f}  99% {q[3]:8.3f}  max {q[4]:8.3f}")
np.savez_compressed(ROOT / "gpurun_out" / "vignette_components.npz", **{k.split("[")[0]: v for k, v in maps.items()})
