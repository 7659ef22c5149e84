"""vignettes/running-microclimf.Rmd:408-467 (images/image7.png, image8.png): temperature-height profiles in the hottest
hour of the monthly-tmax subset (entry 132) over a flat, uniform 5 x 5 raster.
image7 (pai 0.05, 5 mm sward; heights 0.01 .. 1 m), read off the figure: 1 m 27.3, 0.5 m 29.7, 0.2 m 32.8, 0.1 m 35.2,
0.05 m 37.6, 0.02 m 40.8, 0.01 m 43.4 degC.
image8 (pai 3, 10 m canopy; heights 0.1 .. 10 m): 18.1 degC at 0.1 m, a maximum of about 24.95 near 3 m, 24.1 at 6.3 m,
20.8 at 10 m."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
one = np.ones((5, 5))
dem = {"z": 0 * one, "res": 10.0, "lat": dtm["lat"], "long": dtm["long"]}
soilc2 = {"soiltype": 7 * one, "groundr": 0.15 * one}


def uniform(pai, hgt):
    return {"pai": pai * one, "hgt": hgt * one, "x": one, "gsmax": 0.1 * one, "leafr": 0.3 * one, "clump": 0 * one,
            "leafd": 0.05 * one, "leaft": 0.15 * one}


res = {}
for name, vegp2, rq0, heights in (("image7", uniform(0.05, 0.005), 0.05, [0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0]),
                                  ("image8", uniform(3.0, 10.0), 10.0, list(10 ** (np.arange(-10, 11) / 10)))):
    mp = F.subsetpointmodel(F.runpointmodel(weather, rq0, dem, vegp2, soilc2), tstep="month", what="tmax")
    temps = [float(F.runmicro(mp, h, vegp2, soilc2, dem)["Tz"][1, 1, 131]) for h in heights]
    res[name + "_h"] = np.array(heights)
    res[name + "_t"] = np.array(temps)
    print(name, "zref", mp["zref"], "entries", len(mp["weather"]["temp"]))
    for h, t in zip(heights, temps):
        print(f"  {h:7.3f} m  {t:6.2f} degC")
np.savez_compressed(ROOT / "gpurun_out" / "vignette_profiles.npz", **res)
