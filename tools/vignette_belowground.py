"""vignettes/running-microclimf.Rmd:470-494 (images/image9.png): soil temperature over 2017 at 5 cm, 20 cm and 1 m depth
under a uniform 10 m canopy (pai 3) on a flat 5 x 5 raster.  Published curves: 5 cm about 2.7 .. 18.6 degC, 20 cm about
5 .. 15.4, 1 m about 6.8 .. 14.3 (a smooth annual wave peaking in late July / August)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402

weather, vegp, soilc, dtm = load()
st = soilc["soiltype"]
print("soil types of the site:", np.unique(st[~np.isnan(st)]))
one = np.ones((5, 5))
vegp2 = {"pai": 3 * one, "hgt": 10 * one, "x": one, "gsmax": 0.1 * one, "leafr": 0.3 * one, "clump": 0 * one, "leafd": 0.05 * one,
         "leaft": 0.15 * one}
soilc2 = {"soiltype": np.full((5, 5), float(np.round(np.nanmean(st)))), "groundr": 0.15 * one}
dem = {"z": 0 * one, "res": 10.0, "lat": dtm["lat"], "long": dtm["long"]}
out = {}
for depth in (-0.05, -0.2, -1.0):
    mp = F.runpointmodel(weather, depth, dem, vegp2, soilc2)
    t = F.runmicro(mp, depth, vegp2, soilc2, dem)["Tz"][1, 1, :]
    out[str(depth)] = t
    k = int(np.argmax(t))
    print(f"depth {-depth:4.2f} m: min {t.min():.2f}  max {t.max():.2f} degC (hour {k}), mean {t.mean():.2f}; zref {mp['zref']}")
np.savez_compressed(ROOT / "gpurun_out" / "vignette_belowground.npz", **out)
