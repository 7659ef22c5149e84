"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE — see mcf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It reuses the product's argument marshalling so that the oracle
and libmcfhip see byte-identical inputs.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from microclimf_amd import _abi
from microclimf_amd.marshal import alloc_outputs, marshal

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "libmcf_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    srcs = [_DIR / "oracle_unit.c", _DIR / "mcf_oracle.c", _DIR / "mcf_oracle.h", _DIR / "pointmodel.c",
            _DIR / "pointmodel.h", _DIR / "snow_oracle.c", _DIR / "snow_oracle.h", _DIR / "Makefile", _DIR.parent / "include" / "mcf.h"]
    srcs = [s for s in srcs if s.exists()]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_DIR), "-B", "libmcf_oracle.so"], check=True,
                       capture_output=True)
    return LIB_PATH


class Solmodel(C.Structure):
    _fields_ = [("zend", C.c_double), ("zenr", C.c_double), ("azid", C.c_double), ("azir", C.c_double)]


class Kstruct(C.Structure):
    _fields_ = [("k", C.c_double), ("kd", C.c_double), ("Kc", C.c_double)]


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        import os
        alt = os.environ.get("MCF_ORACLE_LIB")       # e.g. an AddressSanitizer build (CPU only)
        if not alt:
            build()
        lib = C.CDLL(alt or str(LIB_PATH))
        lib.orc_run_grid.restype = C.c_int
        lib.orc_run_grid.argtypes = [C.POINTER(_abi.GridInputs), C.POINTER(_abi.Options),
                                     C.POINTER(_abi.Outputs)]
        lib.orc_solposition.restype = Solmodel
        lib.orc_solposition.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]
        lib.orc_solarindex.restype = C.c_double
        lib.orc_solarindex.argtypes = [C.c_double] * 4 + [C.c_int]
        lib.orc_cank.restype = Kstruct
        lib.orc_cank.argtypes = [C.c_double] * 3
        lib.orc_satvap.restype = C.c_double
        lib.orc_satvap.argtypes = [C.c_double]
        lib.orc_julday.restype = C.c_int
        lib.orc_julday.argtypes = [C.c_int] * 3
        lib.orc_na_real.restype = C.c_double
        lib.orc_soild.restype = C.c_double
        lib.orc_soild.argtypes = [C.c_double] * 4
        lib.orc_zeroplanedis.restype = C.c_double
        lib.orc_zeroplanedis.argtypes = [C.c_double] * 2
        lib.orc_roughlength.restype = C.c_double
        lib.orc_roughlength.argtypes = [C.c_double] * 4
        lib.orc_man.restype = None
        lib.orc_man.argtypes = [_abi.c_double_p, C.c_int, C.c_int, _abi.c_double_p]
        _lib = lib
    return _lib


def run_grid(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact,
             complete, mat, out, array_forcing=False, dfsel=None, lib=None):
    """Oracle for runmicro1Cpp (array_forcing=False) / runmicro2Cpp (True); with `dfsel`
    the time-varying-vegetation variants runmicro3Cpp / runmicro4Cpp.  `lib` substitutes another
    build of the oracle (the gcov-instrumented one of tests/test_branch_coverage_cpu.py)."""
    if lib is None:
        lib = load()
    else:
        lib.orc_run_grid.restype = C.c_int
        lib.orc_run_grid.argtypes = [C.POINTER(_abi.GridInputs), C.POINTER(_abi.Options),
                                     C.POINTER(_abi.Outputs)]
    m = marshal(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact,
                complete, mat, out, array_forcing, dfsel=dfsel)
    outs, arrays = alloc_outputs(m)
    rc = lib.orc_run_grid(C.byref(m.inputs), C.byref(m.options), C.byref(outs))
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return arrays


def solposition(lat, lon, year, month, day, hour):
    s = load().orc_solposition(lat, lon, int(year), int(month), int(day), hour)
    return s.zend, s.zenr, s.azid, s.azir


def satvap(tc):
    f = load().orc_satvap
    return np.vectorize(f, otypes=[float])(tc)


def run_bioclim(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, mat,
                out, wetq, dryq, hotq, colq, air, array_forcing=False, dfsel=None):
    """Oracle for runbioclim1Cpp / runbioclim2Cpp (cpp:3563-3616): the grid oracle with the reference's
    output mask, then runbioclimCpp's reductions cell by cell."""
    lib = load()
    mask = [0] * 10
    mask[0 if air else 1] = 1
    mask[3] = 1
    res = run_grid(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact,
                   True, mat, mask, array_forcing, dfsel=dfsel)
    tz = res["Tz" if air else "tleaf"]
    sm = res["soilm"]
    R, Cc, T = tz.shape
    qs = [np.ascontiguousarray(np.asarray(q, dtype=np.int32)) for q in (wetq, dryq, hotq, colq)]
    bio = np.full((19, R, Cc), lib.orc_na_real())
    tmp = np.zeros(19)
    IP, DP = C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.orc_bioclim_cell.restype = None
    for i in range(R):
        for j in range(Cc):
            if np.isnan(tz[i, j, 0]):
                continue
            a = np.ascontiguousarray(tz[i, j, :]); b = np.ascontiguousarray(sm[i, j, :])
            lib.orc_bioclim_cell(a.ctypes.data_as(DP), b.ctypes.data_as(DP), C.c_int(T),
                                 qs[0].ctypes.data_as(IP), C.c_int(len(qs[0])), qs[1].ctypes.data_as(IP),
                                 C.c_int(len(qs[1])), qs[2].ctypes.data_as(IP), C.c_int(len(qs[2])),
                                 qs[3].ctypes.data_as(IP), C.c_int(len(qs[3])), tmp.ctypes.data_as(DP))
            bio[:, i, j] = tmp
    return {f"bio{v + 1}": bio[v] for v in range(19) if out[v]}


# ---- snow branch (snow_oracle.c) ----
def run_snowmodel(obstime, climdata, pointm, vegp, other, snowenv, array_forcing=False, lib=None):
    """Oracle for gridmodelsnow1 (array_forcing=False) / gridmodelsnow2 (True).  `lib`: another build
    of the oracle (the gcov-instrumented one)."""
    from microclimf_amd import snow as S
    lib = lib or load()
    m = S.marshal_snow(obstime, climdata, vegp, other, array_forcing, pointm=pointm, snowenv=snowenv)
    out, arrays = S.alloc_snowmodel_out(m)
    lib.orc_gridmodelsnow.restype = C.c_int
    lib.orc_gridmodelsnow.argtypes = [C.POINTER(_abi.SnowInputs), C.POINTER(_abi.SnowModelOut)]
    rc = lib.orc_gridmodelsnow(C.byref(m.inputs), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"snow oracle failed: {rc}")
    return arrays


def run_microsnow(reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out, array_forcing=False, lib=None):
    """Oracle for gridmicrosnow1 / gridmicrosnow2; returns updated copies of the requested fields."""
    from microclimf_amd import snow as S
    lib = lib or load()
    m = S.marshal_snow(obstime, climdata, vegp, other, array_forcing, micro=True)
    sm = S.marshal_snowm(m, snowm)
    sel, outs, arrays = S.marshal_micro(m, micro, out)
    lib.orc_gridmicrosnow.restype = C.c_int
    lib.orc_gridmicrosnow.argtypes = [C.POINTER(_abi.SnowInputs), C.POINTER(_abi.Snowm), C.c_double, C.c_double,
                                      C.POINTER(C.c_int32 * _abi.NOUT), C.POINTER(_abi.Outputs)]
    rc = lib.orc_gridmicrosnow(C.byref(m.inputs), C.byref(sm), float(reqhgt), float(mat), C.byref(sel),
                               C.byref(outs))
    if rc != 0:
        raise RuntimeError(f"snow oracle failed: {rc}")
    return arrays
