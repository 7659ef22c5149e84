"""Loader for tests/golden/*.npz (oracle-generated regression vectors)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = ("vec_below_canopy", "vec_above_canopy", "vec_ground", "vec_soil", "arr_below_canopy")


def load(name):
    z = np.load(GOLDEN / f"{name}.npz")
    args = {g: {} for g in ("obstime", "climdata", "pointm", "vegp", "soilc")}
    expect = {}
    scal = {}
    for key in z.files:
        grp, k = key.split(".", 1)
        if grp in args:
            args[grp][k] = z[key]
        elif grp == "expect":
            expect[k] = z[key]
        else:
            scal[k] = z[key]
    a = dict(args)
    for k in ("reqhgt", "zref", "Sminp", "Smaxp", "tfact", "mat"):
        a[k] = float(scal[k])
    a["complete"] = bool(scal["complete"])
    a["out"] = [bool(x) for x in scal["out"]]
    af = scal["lat"].ndim == 2
    a["lat"] = scal["lat"] if af else float(scal["lat"])
    a["lon"] = scal["lon"] if af else float(scal["lon"])
    names = [n for n, on in zip(("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown",
                                 "Rlwdown", "Rswup", "Rlwup"), a["out"]) if on]
    expect = {n: expect[n] for n in names}
    return a, af, expect


SNOW_CASES = ("snow_vec", "snow_arr")


def load_snow(name):
    """(snow_workload-like dict, array_forcing, reqhgt, mat, micro, expected gridmodelsnow, expected gridmicrosnow)"""
    z = np.load(GOLDEN / f"{name}.npz")
    sw = {g: {} for g in ("obstime", "climdata", "pointm", "vegp", "other")}
    micro, smod, mout, scal = {}, {}, {}, {}
    for key in z.files:
        grp, k = key.split(".", 1)
        v = z[key]
        if grp in sw:
            sw[grp][k] = v if v.ndim else v.item()
        elif grp == "micro":
            micro[k] = v
        elif grp == "smod":
            smod[k] = v
        elif grp == "mout":
            mout[k] = v
        else:
            scal[k] = v
    sw["snowenv"] = str(scal["snowenv"])
    names = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
    micro = {n: micro[n] for n in names}
    mout = {n: mout[n] for n in names}
    smod = {n: smod[n] for n in ("Tc", "Tg", "sdepc", "sdepg", "sden", "agec", "ageg", "meltc", "meltg")}
    return sw, bool(scal["array_forcing"]), float(scal["reqhgt"]), float(scal["mat"]), micro, smod, mout
