"""The reference package's bundled example data (BASELINE.json configs[0]) as the front end's inputs, from the fixture
tests/golden/bundled_caerth.npz (made by tests/golden/make_bundled_inputs.py)."""
from pathlib import Path

import numpy as np

FIXTURE = Path(__file__).resolve().parent / "golden" / "bundled_caerth.npz"


def load(hours: int | None = None):
    d = np.load(FIXTURE)
    sl = slice(0, hours)
    weather = {k[5:]: d[k][sl] for k in d.files if k.startswith("clim_")}
    weather["obstime"] = {"year": d["time_year"][sl], "month": d["time_month"][sl], "day": d["time_day"][sl],
                          "hour": d["time_hour"][sl]}
    vegp = {k[5:]: d[k] for k in d.files if k.startswith("vegp_")}
    soilc = {k[6:]: d[k] for k in d.files if k.startswith("soilc_")}
    dtm = {"z": d["dtm"], "res": float(d["res"][0]), "lat": float(d["latlong"][0]), "long": float(d["latlong"][1]),
           "extent": d["extent"]}
    return weather, vegp, soilc, dtm
