"""Turns the reference package's bundled example DATA (data/*.rda: dtmcaerth, vegp, soilc, climdata and the two soil
parameter tables — BASELINE.json configs[0]) into tests/golden/bundled_caerth.npz.  Data only: no code of the reference
is read or stored.  Run in the build container (needs /root/reference): python tests/golden/make_bundled_inputs.py"""
import math
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tools"))
from read_rda import read_rda  # noqa: E402

DATA = Path("/root/reference/data")


def raster(obj):
    """PackedSpatRaster -> ([rows, cols] or [rows, cols, layers] array, definition string)"""
    d = obj.attrs["definition"].value[0]
    nc, nr, nl = (int(re.search(rf"{k}=(\d+)", d).group(1)) for k in ("ncols", "nrows", "nlyrs"))
    v = obj.attrs["values"].value.reshape((nl, nr * nc)).T if obj.attrs["values"].attrs["dim"].value[1] == nl else None
    a = obj.attrs["values"].value.reshape((nl, nr, nc))            # R matrix [cells, layers] column-major = layer-major
    a = np.transpose(a, (1, 2, 0))                                  # cells are row-major within a layer
    return (a[:, :, 0] if nl == 1 else a), d


def tm_inverse(E, N, a=6377563.396, rf=299.3249646, lat0=49.0, lon0=-2.0, k0=0.9996012717, fe=400000.0, fn=-100000.0):
    """Transverse Mercator -> geographic on the projection's own ellipsoid (Airy 1830), the series of the Ordnance Survey
    guide 'A guide to coordinate systems in Great Britain', annex C.  The raster's CRS names no datum, so no datum shift."""
    f = 1 / rf
    b = a * (1 - f)
    e2 = (a * a - b * b) / (a * a)
    n = (a - b) / (a + b)
    phi0, lam0 = math.radians(lat0), math.radians(lon0)

    def M(phi):
        return b * k0 * ((1 + n + 1.25 * n ** 2 + 1.25 * n ** 3) * (phi - phi0)
                         - (3 * n + 3 * n ** 2 + 21 / 8 * n ** 3) * math.sin(phi - phi0) * math.cos(phi + phi0)
                         + (15 / 8 * n ** 2 + 15 / 8 * n ** 3) * math.sin(2 * (phi - phi0)) * math.cos(2 * (phi + phi0))
                         - 35 / 24 * n ** 3 * math.sin(3 * (phi - phi0)) * math.cos(3 * (phi + phi0)))
    phi = (N - fn) / (a * k0) + phi0
    while abs(N - fn - M(phi)) >= 1e-5:
        phi = (N - fn - M(phi)) / (a * k0) + phi
    s = math.sin(phi)
    nu = a * k0 / math.sqrt(1 - e2 * s * s)
    rho = a * k0 * (1 - e2) * (1 - e2 * s * s) ** -1.5
    eta2 = nu / rho - 1
    t = math.tan(phi)
    c = 1 / math.cos(phi)
    VII = t / (2 * rho * nu)
    VIII = t / (24 * rho * nu ** 3) * (5 + 3 * t * t + eta2 - 9 * t * t * eta2)
    IX = t / (720 * rho * nu ** 5) * (61 + 90 * t * t + 45 * t ** 4)
    X = c / nu
    XI = c / (6 * nu ** 3) * (nu / rho + 2 * t * t)
    XII = c / (120 * nu ** 5) * (5 + 28 * t * t + 24 * t ** 4)
    XIIA = c / (5040 * nu ** 7) * (61 + 662 * t * t + 1320 * t ** 4 + 720 * t ** 6)
    dE = E - fe
    lat = phi - VII * dE ** 2 + VIII * dE ** 4 - IX * dE ** 6
    lon = lam0 + X * dE - XI * dE ** 3 + XII * dE ** 5 - XIIA * dE ** 7
    return math.degrees(lat), math.degrees(lon)


def main():
    out = {}
    dtm, d = raster(read_rda(DATA / "dtmcaerth.rda")["dtmcaerth"])
    out["dtm"] = dtm
    ext = {k: float(re.search(rf"{k}=([-\d.]+)", d).group(1)) for k in ("xmin", "xmax", "ymin", "ymax")}
    out["extent"] = np.array([ext["xmin"], ext["xmax"], ext["ymin"], ext["ymax"]])
    out["res"] = np.array([(ext["xmax"] - ext["xmin"]) / dtm.shape[1], (ext["ymax"] - ext["ymin"]) / dtm.shape[0]])
    prm = {k: float(v) for k, v in re.findall(r'PARAMETER\["([^"]+)",([-\d.]+)', d)}
    lat, lon = tm_inverse((ext["xmin"] + ext["xmax"]) / 2, (ext["ymin"] + ext["ymax"]) / 2,
                          lat0=prm["Latitude of natural origin"], lon0=prm["Longitude of natural origin"],
                          k0=prm["Scale factor at natural origin"], fe=prm["False easting"], fn=prm["False northing"])
    out["latlong"] = np.array([lat, lon])
    vegp = read_rda(DATA / "vegp.rda")["vegp"]
    for k in vegp.names():
        out["vegp_" + k] = raster(vegp[k])[0]
    soilc = read_rda(DATA / "soilc.rda")["soilc"]
    for k in soilc.names():
        out["soilc_" + k] = raster(soilc[k])[0]
    clim = read_rda(DATA / "climdata.rda")["climdata"]
    for k in clim.names():
        if k == "obs_time":
            t = clim[k]
            out["time_year"] = t["year"].value + 1900
            out["time_month"] = t["mon"].value + 1
            out["time_day"] = t["mday"].value
            out["time_hour"] = t["hour"].value + t["min"].value / 60.0 + t["sec"].value / 3600.0
        else:
            out["clim_" + k] = np.asarray(clim[k].value, dtype=np.float64)
    for tab in ("soilparameters", "soilparamsp"):
        t = read_rda(DATA / f"{tab}.rda")[tab]
        for k in t.names():
            v = t[k]
            out[f"{tab}_{k}"] = np.array(v.value) if v.kind == "character" else np.asarray(v.value, dtype=np.float64)
    dst = Path(__file__).resolve().parent / "bundled_caerth.npz"
    np.savez_compressed(dst, **out)
    print(dst, dst.stat().st_size, "bytes;", "lat/long", out["latlong"], "dtm", dtm.shape, "NA", int(np.isnan(dtm).sum()))
    for k, v in out.items():
        print(f"  {k:28s} {getattr(v, 'shape', '')} {v.dtype}")


if __name__ == "__main__":
    main()
