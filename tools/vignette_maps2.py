"""vignettes/running-microclimf.Rmd:392-397 (image6.png: air temperature 5 cm above ground, entry 134 of the monthly-tmax
subset, colour scale about 25 .. 54 degC) and :540-549 (image10.png: the same run written with writetonc and layer 12 read
back /100, colour scale about 7.2 .. 19.4 degC)."""
import sys
import tempfile
from pathlib import Path

import numpy as np
from scipy.io import netcdf_file

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bundled import load  # noqa: E402
from microclimf_amd import frontend as F  # noqa: E402
from microclimf_amd.ncsink import writetonc  # noqa: E402

weather, vegp, soilc, dtm = load()
mp = F.subsetpointmodel(F.runpointmodel(weather, 0.05, dtm, vegp, soilc), tstep="month", what="tmax")
mout = dict(F.runmicro(mp, 0.05, vegp, soilc, dtm))
t6 = mout["Tz"][:, :, 133]
print(f"image6: Tz[,,134] {np.nanmin(t6):.2f} .. {np.nanmax(t6):.2f} degC")
mout["tme"] = mp["obstime"]
xmin, xmax, ymin, ymax = dtm["extent"]
with tempfile.TemporaryDirectory() as d:
    f = str(Path(d) / "modelout.nc")
    writetonc(mout, f, {"xmin": xmin, "xmax": xmax, "ymin": ymin, "ymax": ymax, "res": dtm["res"]}, 0.05)
    with netcdf_file(f, "r", mmap=False) as nc:
        v = nc.variables["Tz"]
        print("nc Tz", v.shape, v.data.dtype, {k: getattr(v, k) for k in v._attributes})
        lay = np.array(v.data[11], dtype=np.float64)
        miss = getattr(v, "_FillValue", None)
lay[lay == miss] = np.nan
lay /= 100
print(f"image10: Tz[[12]]/100 {np.nanmin(lay):.2f} .. {np.nanmax(lay):.2f} degC; NaN cells {int(np.isnan(lay).sum())}")
# the figure: warm upper-left half (13 .. 19), cold lower-right half (7 .. 11) split along the NE-facing scarp
print("  mean of rows 10-25 x cols 0-20 (warm slope):", np.nanmean(lay[10:25, 0:20]).round(2),
      " rows 20-45 x cols 20-30 (shaded):", np.nanmean(lay[20:45, 20:30]).round(2))
np.savez_compressed(ROOT / "gpurun_out" / "vignette_maps2.npz", image6=t6, image10=lay)
