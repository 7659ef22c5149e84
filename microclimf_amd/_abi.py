"""ctypes mirror of include/mcf.h and the loader for libmcfhip.so.

The structs here are a field-for-field transcription of the C header; the
shared library is the product (hand-written HIP for gfx950).  There is no CPU
fallback: `load()` raises if the library is not built, and every solve raises
if no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NOUT = 10
OUT_NAMES = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown",
             "Rdifdown", "Rlwdown", "Rswup", "Rlwup")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class Obstime(C.Structure):
    _fields_ = [("year", c_int32_p), ("month", c_int32_p), ("day", c_int32_p),
                ("hour", c_double_p)]


CLIM_FIELDS = ("tc", "es", "ea", "tdew", "pk", "swdown", "difrad", "lwdown",
               "windspeed", "winddir")
POINTM_FIELDS = ("soilm", "Tg", "Tbp", "G", "umu", "kp", "muGp", "dtrp")
VEGP_FIELDS = ("hgt", "pai", "x", "gsmax", "leafr", "leaft", "clump", "leafd",
               "paia", "leafden")
SOILC_FIELDS = ("Smin", "Smax", "gref", "soilb", "Psie", "Vq", "Vm", "Mc", "rho",
                "slope", "aspect", "twi", "svfa", "wsa", "hor")


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, c_double_p) for f in fields]})


Climate = _ptr_struct("Climate", CLIM_FIELDS)
Pointm = _ptr_struct("Pointm", POINTM_FIELDS)
Vegp = _ptr_struct("Vegp", VEGP_FIELDS)
Soilc = _ptr_struct("Soilc", SOILC_FIELDS)


class GridInputs(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("tsteps", C.c_int64),
                ("array_forcing", C.c_int32), ("veg_layers", C.c_int32),
                ("obstime", Obstime), ("clim", Climate), ("pointm", Pointm),
                ("vegp", Vegp), ("soilc", Soilc),
                ("lat", C.c_double), ("lon", C.c_double),
                ("lats", c_double_p), ("lons", c_double_p),
                ("lyr_st", c_int32_p), ("lyr_ed", c_int32_p),
                ("coarse_rows", C.c_int32), ("coarse_cols", C.c_int32),
                ("coarse_rowpos", c_double_p), ("coarse_colpos", c_double_p),
                ("coarse_relhum", c_double_p), ("coarse_winddir", c_double_p),
                ("coarse_altcorrect", C.c_int32), ("coarse_dtm", c_double_p), ("fine_dtm", c_double_p),
                ("row_pitch", C.c_int64)]


class Multi(C.Structure):
    """include/mcf.h mcf_multi: devices and row blocks of the one-process multi-device entry points."""
    _fields_ = [("n_devices", C.c_int32), ("devices", c_int32_p), ("n_blocks", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("reqhgt", C.c_double), ("zref", C.c_double),
                ("Sminp", C.c_double), ("Smaxp", C.c_double),
                ("tfact", C.c_double), ("mat", C.c_double),
                ("complete", C.c_int32), ("out", C.c_int32 * NOUT),
                ("device", C.c_int32), ("days_per_chunk", C.c_int32),
                ("cells_per_block", C.c_int32)]


class TerrainIn(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("halo_north", C.c_int32),
                ("halo_south", C.c_int32), ("dtm", c_double_p), ("res", C.c_double),
                ("zref", C.c_double), ("agg", C.c_int32), ("reserved0", C.c_int32),
                ("row0", C.c_int64), ("rows_total", C.c_int64)]


class TerrainOut(C.Structure):
    _fields_ = [("slope", c_double_p), ("aspect", c_double_p), ("hor", c_double_p),
                ("svfa", c_double_p), ("wsa", c_double_p)]


NBIO = 19


class BioclimSel(C.Structure):
    _fields_ = [("wetq", c_int32_p), ("dryq", c_int32_p), ("hotq", c_int32_p), ("colq", c_int32_p),
                ("nwet", C.c_int32), ("ndry", C.c_int32), ("nhot", C.c_int32), ("ncol", C.c_int32),
                ("air", C.c_int32), ("out", C.c_int32 * NBIO)]


class BioclimOut(C.Structure):
    _fields_ = [("bio", c_double_p * NBIO)]


class Outputs(C.Structure):
    _fields_ = [("var", c_double_p * NOUT)]


# ---- snow branch (include/mcf.h "snow branch") ----
SNOWENV = {"Alpine": 0, "Maritime": 1, "Prairie": 2, "Tundra": 3, "Taiga": 4}
SNOW_CLIM_FIELDS = ("temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir", "precip", "umu")
SNOW_POINTM_FIELDS = ("Gp", "Tc", "RswabsG", "RlwabsG", "umu")
SNOW_VEGP_FIELDS = ("pai", "hgt", "leaft", "clump", "paia", "leafd", "leafden")
SNOWM_FIELDS = ("Tc", "Tg", "totalSWE", "groundsnowdepth", "snowden")
SNOWMODEL_OUT3 = ("Tc", "Tg", "sdepc", "sdepg", "sden")
SNOWMODEL_OUT2 = ("agec", "ageg", "meltc", "meltg")

SnowClimate = _ptr_struct("SnowClimate", SNOW_CLIM_FIELDS)
SnowPointm = _ptr_struct("SnowPointm", SNOW_POINTM_FIELDS)
SnowVegp = _ptr_struct("SnowVegp", SNOW_VEGP_FIELDS)
Snowm = _ptr_struct("Snowm", SNOWM_FIELDS)
SnowModelOut = _ptr_struct("SnowModelOut", SNOWMODEL_OUT3 + SNOWMODEL_OUT2)


class SnowOther(C.Structure):
    _fields_ = [("slope", c_double_p), ("aspect", c_double_p), ("skyview", c_double_p), ("wsa", c_double_p),
                ("hor", c_double_p), ("lat", C.c_double), ("lon", C.c_double), ("lats", c_double_p),
                ("lons", c_double_p), ("zref", C.c_double), ("isnowdc", c_double_p), ("isnowdg", c_double_p),
                ("isnowac", c_int32_p), ("isnowag", c_int32_p), ("Smax", c_double_p)]


class SnowInputs(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("tsteps", C.c_int64), ("array_forcing", C.c_int32),
                ("snowenv", C.c_int32), ("obstime", Obstime), ("clim", SnowClimate), ("pointm", SnowPointm),
                ("vegp", SnowVegp), ("other", SnowOther)]


POINT_WEATHER_FIELDS = ("temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "precip")
PointWeather = _ptr_struct("PointWeather", POINT_WEATHER_FIELDS)
BIGLEAF_FIELDS = ("Tc", "Tg", "H", "G", "psih", "psim", "phih", "OL", "uf", "RabsG", "albedo")


class BigLeafOut(C.Structure):
    _fields_ = [(k, c_double_p) for k in BIGLEAF_FIELDS] + [("err", C.c_double), ("iters", C.c_int32)]


class NcSpec(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nsteps", C.c_int64), ("east", c_double_p),
                ("north", c_double_p), ("time_hours", c_double_p), ("crs_wkt", C.c_char_p), ("reqhgt", C.c_double),
                ("vars", C.c_int32 * 10), ("reference_puts_only", C.c_int32), ("format", C.c_int32),
                ("deflate_level", C.c_int32)]


POINTSNOW_FIELDS = ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "sdeng", "G", "RswabsG", "RlwabsG", "tr", "umu", "sublmelt",
                    "tempmelt", "rainmelt", "sstemp")


class PointSnowOut(C.Structure):
    _fields_ = [(k, c_double_p) for k in POINTSNOW_FIELDS] + [("mxdif", C.c_double), ("iters", C.c_int32)]


class SnowDriverIn(C.Structure):
    _fields_ = [("base", SnowInputs), ("dtm", c_double_p), ("res", C.c_double), ("tfact", C.c_double),
                ("chunk_steps", C.c_int32), ("af_wsa_s", C.c_int32), ("af_wind", c_double_p)]


SNOWDRIVER_OUT = ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden")
SnowDriverOut = _ptr_struct("SnowDriverOut", SNOWDRIVER_OUT)


class MicrosnowIn(C.Structure):
    """include/mcf.h mcf_microsnow_in"""
    _fields_ = [("grid", C.POINTER(GridInputs)), ("snow", C.POINTER(SnowDriverIn)), ("micro", C.POINTER(SnowInputs)),
                ("mat", C.c_double)]

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "csrc" / "libmcfhip.so"

# every symbol include/mcf.h declares (tests check the .so exports all of them)
EXPORTS = (
    "mcf_abi_version", "mcf_last_error", "mcf_device_count",
    "mcf_runmicro1", "mcf_runmicro2", "mcf_runmicro3", "mcf_runmicro4",
    "mcf_plan_create", "mcf_plan_destroy", "mcf_plan_twi_partial",
    "mcf_plan_set_twi_mean", "mcf_plan_upload_forcing_days", "mcf_plan_run_days",
    "mcf_plan_belowground", "mcf_plan_sync", "mcf_plan_fetch", "mcf_plan_fetch_cells", "mcf_plan_fetch_packed", "mcf_plan_slot_ptr",
    "mcf_plan_ring_layout", "mcf_ring_index", "mcf_plan_run_days_at", "mcf_plan_run_days_masked", "mcf_plan_set_mxtc",
    "mcf_snowplan_covered_tiles", "mcf_plan_run_days_cells", "mcf_snowplan_free_cells",
    "mcf_runmicro1_multi", "mcf_runmicro2_multi", "mcf_runmicro3_multi", "mcf_runmicro4_multi", "mcf_plan_fetch_pitched",
    "mcf_snowplan_reset", "mcf_snowplan_checkpoint", "mcf_snowplan_restore", "mcf_snowplan_fetch_cells", "mcf_snowplan_keep_chunk", "mcf_snowplan_can_keep", "mcf_snowplan_set_keep_budget", "mcf_plan_set_mxtc_days", "mcf_snowplan_set_series",
    "mcf_snowplan_release_kept", "mcf_snowplan_meand_accumulate", "mcf_snowplan_micro_setup", "mcf_snowplan_microsnow",
    "mcf_plan_timer_start", "mcf_plan_timer_stop", "mcf_plan_kernel_timing",
    "mcf_plan_kernel_stats", "mcf_plan_dispatch_stats", "mcf_plan_valid_cells", "mcf_plan_bytes", "mcf_selftest_math",
    "mcf_precompute_terrain", "mcf_precompute_terrain_multi", "mcf_runbioclim1_multi", "mcf_runbioclim2_multi",
    "mcf_runbioclim3_multi", "mcf_runbioclim4_multi", "mcf_runbioclim1", "mcf_runbioclim2", "mcf_runbioclim3", "mcf_runbioclim4",
    "mcf_snowenv_from_name", "mcf_gridmodelsnow1", "mcf_gridmodelsnow2", "mcf_gridmicrosnow1",
    "mcf_gridmicrosnow2", "mcf_snowmodel1", "mcf_snowmodel2", "mcf_snowmodel1_multi", "mcf_applycpp3",
    "mcf_snowplan_create", "mcf_snowplan_destroy", "mcf_snowplan_chunks", "mcf_snowplan_surface", "mcf_snowplan_handover", "mcf_snowplan_apply3",
    "mcf_snowplan_surface_partial", "mcf_snowplan_prepare_chunk", "mcf_snowplan_run_chunk", "mcf_snowplan_pack_halo",
    "mcf_snowplan_prepare_chunk_dev",
    "mcf_bigleaf", "mcf_soilm", "mcf_pointmprocess", "mcf_weatherhgt", "mcf_man", "mcf_pointmodelsnow", "mcf_canintfrac", "mcf_meltmu", "mcf_meltmu2", "mcf_tpicalc",
    "mcf_nc_create", "mcf_nc_write_host", "mcf_nc_write_plan", "mcf_nc_close",
    "mcf_flowacc", "mcf_topidx",
    "mcf_runmicrosnow1", "mcf_runmicrosnow2", "mcf_runmicrosnow1_multi", "mcf_snowrun_create", "mcf_snowrun_destroy", "mcf_snowrun_days", "mcf_snowrun_stats", "mcf_snowrun_keep",
    "mcf_snowrun_pass1", "mcf_snowrun_pass2", "mcf_snowplan_run_chunk_pitched", "mcf_snowplan_chunk_af",
)

ABI_VERSION = 6     # include/mcf.h MCF_ABI_VERSION this mirror was written against
_lib = None


class McfError(RuntimeError):
    """Raised when libmcfhip reports a non-zero status."""


class DispatchStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("fast_tiles", "slow_tiles", "irregular_days", "fast_launches", "slow_launches",
                                         "canary_trips")]


class RingLayout(C.Structure):
    """include/mcf.h mcf_ring_layout: how a ring slot variable is addressed on the device."""
    _fields_ = [("tiled", C.c_int32), ("cells_per_tile", C.c_int32), ("block_doubles", C.c_int32), ("slot_days", C.c_int32),
                ("cells", C.c_int64), ("tile_stride", C.c_int64), ("day_stride", C.c_int64)]


def _needed_hip_soname(lib_path: Path):
    """DT_NEEDED entry of libmcfhip.so that names the HIP runtime (e.g. 'libamdhip64.so.7'), read from the ELF's dynamic
    section's strings; None if it cannot be found."""
    import re
    try:
        data = lib_path.read_bytes()
    except OSError:
        return None
    m = re.search(rb"libamdhip64\.so\.\d+", data)
    return m.group(0).decode() if m else None


def _share_hip_runtime(lib_path: Path):
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's); the copy
    that is loaded first serves every later user.  If libmcfhip pulled in /opt/rocm's first, a later `import torch` would
    bring a SECOND runtime into the process and find no device ("No HIP GPUs are available").  Where torch is installed but
    not imported yet, its copy is therefore loaded first — the same state as importing torch before this package.  Hosts
    without torch (an R session) use /opt/rocm's.

    Only when the wheel's runtime carries the SONAME libmcfhip was linked against (a different major would leave two runtimes
    in the process after all, silently): otherwise a warning says so and nothing is preloaded.  MCF_NO_HIP_PRELOAD=1 opts out
    (processes that will never import torch)."""
    import importlib.util
    import sys
    import warnings
    if "torch" in sys.modules or os.environ.get("MCF_NO_HIP_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if not cand.exists():
        return
    need, have = _needed_hip_soname(lib_path), _needed_hip_soname(cand)      # (a library's own SONAME string matches too)
    if need and have and need != have:
        warnings.warn(f"libmcfhip is linked against {need}, the installed torch bundles {have}: not preloading torch's HIP "
                      "runtime — import torch BEFORE microclimf_amd if both are used in this process", RuntimeWarning)
        return
    try:
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def load() -> C.CDLL:
    """Load libmcfhip.so (built by `make -C microclimf_amd/csrc` / build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MCF_LIB", LIB_PATH))
    if not path.exists():
        raise McfError(
            f"{path} not found: the HIP extension is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C microclimf_amd/csrc`). There is no CPU fallback.")
    _share_hip_runtime(path)
    lib = C.CDLL(str(path))
    lib.mcf_abi_version.restype = C.c_int
    lib.mcf_last_error.restype = C.c_char_p
    lib.mcf_device_count.restype = C.c_int
    GI, OP, OU = C.POINTER(GridInputs), C.POINTER(Options), C.POINTER(Outputs)
    for fn in (lib.mcf_runmicro1, lib.mcf_runmicro2, lib.mcf_runmicro3, lib.mcf_runmicro4):
        fn.restype = C.c_int
        fn.argtypes = [GI, OP, OU]
    P = C.c_void_p
    lib.mcf_plan_create.restype = C.c_int
    lib.mcf_plan_create.argtypes = [GI, OP, C.c_int32, C.c_int32, C.POINTER(P)]
    lib.mcf_plan_destroy.restype = None
    lib.mcf_plan_destroy.argtypes = [P]
    lib.mcf_plan_twi_partial.restype = C.c_int
    lib.mcf_plan_twi_partial.argtypes = [P, c_double_p, C.POINTER(C.c_int64)]
    lib.mcf_plan_set_twi_mean.restype = C.c_int
    lib.mcf_plan_set_twi_mean.argtypes = [P, C.c_double]
    lib.mcf_plan_upload_forcing_days.restype = C.c_int
    lib.mcf_plan_upload_forcing_days.argtypes = [P, GI, C.c_int32, C.c_int32, C.c_int32]
    lib.mcf_plan_run_days.restype = C.c_int
    lib.mcf_plan_run_days.argtypes = [P, C.c_int32, C.c_int32, C.c_int32]
    if hasattr(lib, "mcf_plan_run_days_at"):     # (absent from an older library named by MCF_LIB for an A/B run)
        for fn in (lib.mcf_runmicro1_multi, lib.mcf_runmicro2_multi, lib.mcf_runmicro3_multi, lib.mcf_runmicro4_multi):
            fn.restype = C.c_int
            fn.argtypes = [GI, OP, C.POINTER(Multi), OU]
        lib.mcf_plan_fetch_pitched.restype = C.c_int
        lib.mcf_plan_fetch_pitched.argtypes = [P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, c_double_p, C.c_int64]
        lib.mcf_plan_run_days_at.restype = C.c_int
        lib.mcf_plan_run_days_at.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        lib.mcf_plan_set_mxtc.restype = C.c_int
        lib.mcf_plan_set_mxtc.argtypes = [P, C.c_double]
        lib.mcf_snowplan_reset.restype = C.c_int
        lib.mcf_snowplan_reset.argtypes = [P]
        lib.mcf_snowplan_fetch_cells.restype = C.c_int
        lib.mcf_snowplan_fetch_cells.argtypes = [P, C.c_int32, C.POINTER(C.c_int64), C.c_int32, c_double_p, C.POINTER(C.c_int32)]
        lib.mcf_snowplan_can_keep.restype = C.c_int
        lib.mcf_snowplan_can_keep.argtypes = [P, C.c_int64, C.POINTER(C.c_int32)]
        lib.mcf_snowplan_set_series.restype = C.c_int
        lib.mcf_snowplan_set_series.argtypes = [P, C.c_uint32]
        lib.mcf_snowplan_keep_chunk.restype = C.c_int
        lib.mcf_snowplan_keep_chunk.argtypes = [P, C.c_int32, C.c_int64, C.POINTER(C.c_int32)]
        lib.mcf_snowplan_release_kept.restype = C.c_int
        lib.mcf_snowplan_release_kept.argtypes = [P]
        for fn in (lib.mcf_snowplan_checkpoint, lib.mcf_snowplan_restore):
            fn.restype = C.c_int
            fn.argtypes = [P, C.c_int32]
        lib.mcf_snowplan_meand_accumulate.restype = C.c_int
        lib.mcf_snowplan_meand_accumulate.argtypes = [P, C.c_int32, c_int32_p]
        lib.mcf_snowplan_micro_setup.restype = C.c_int
        lib.mcf_snowplan_micro_setup.argtypes = [P, C.POINTER(SnowInputs), c_int32_p, C.c_int32, C.c_double, C.c_double,
                                                 C.POINTER(C.c_int32 * NOUT), C.c_int32]
        lib.mcf_plan_run_days_masked.restype = C.c_int
        lib.mcf_plan_run_days_masked.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_int64]
        lib.mcf_snowplan_covered_tiles.restype = C.c_int
        lib.mcf_snowplan_covered_tiles.argtypes = [P, P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_int64, C.POINTER(C.c_int64)]
        lib.mcf_plan_run_days_cells.restype = C.c_int
        lib.mcf_plan_run_days_cells.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.mcf_snowplan_free_cells.restype = C.c_int
        lib.mcf_snowplan_free_cells.argtypes = [P, P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.mcf_snowplan_microsnow.restype = C.c_int
        lib.mcf_snowplan_microsnow.argtypes = [P, P, C.c_int32, C.c_int32, c_int32_p]
    lib.mcf_plan_belowground.restype = C.c_int
    lib.mcf_plan_belowground.argtypes = [P]
    lib.mcf_plan_sync.restype = C.c_int
    lib.mcf_plan_sync.argtypes = [P]
    lib.mcf_plan_fetch.restype = C.c_int
    lib.mcf_plan_fetch.argtypes = [P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, c_double_p]
    lib.mcf_plan_fetch_cells.restype = C.c_int
    lib.mcf_plan_fetch_cells.argtypes = [P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int64,
                                         c_double_p]
    lib.mcf_plan_fetch_packed.restype = C.c_int
    lib.mcf_plan_fetch_packed.argtypes = [P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_double, c_int32_p,
                                          C.POINTER(C.c_float)]
    lib.mcf_plan_slot_ptr.restype = C.c_int
    lib.mcf_plan_slot_ptr.argtypes = [P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    try:
        lib.mcf_plan_ring_layout.restype = C.c_int
        lib.mcf_plan_ring_layout.argtypes = [P, C.POINTER(RingLayout)]
        lib.mcf_ring_index.restype = C.c_int64
        lib.mcf_ring_index.argtypes = [C.POINTER(RingLayout), C.c_int64, C.c_int64]
    except AttributeError:
        if "MCF_LIB" not in os.environ:     # an older library named for a same-box A/B run (tools/ab_bench2.sh) may lack them
            raise
    lib.mcf_plan_timer_start.restype = C.c_int
    lib.mcf_plan_timer_start.argtypes = [P]
    lib.mcf_plan_timer_stop.restype = C.c_int
    lib.mcf_plan_timer_stop.argtypes = [P, C.POINTER(C.c_float)]
    lib.mcf_plan_kernel_timing.restype = C.c_int
    lib.mcf_plan_kernel_timing.argtypes = [P, C.c_int32]
    lib.mcf_plan_kernel_stats.restype = C.c_int
    lib.mcf_plan_kernel_stats.argtypes = [P, c_double_p, C.POINTER(C.c_int64)]
    lib.mcf_plan_dispatch_stats.restype = C.c_int
    lib.mcf_plan_dispatch_stats.argtypes = [P, C.POINTER(DispatchStats)]
    lib.mcf_plan_valid_cells.restype = C.c_int64
    lib.mcf_plan_valid_cells.argtypes = [P]
    lib.mcf_plan_bytes.restype = C.c_int64
    lib.mcf_plan_bytes.argtypes = [P]
    lib.mcf_selftest_math.restype = C.c_int
    lib.mcf_selftest_math.argtypes = [C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int32]
    for fn in (lib.mcf_runbioclim1, lib.mcf_runbioclim2, lib.mcf_runbioclim3, lib.mcf_runbioclim4):
        fn.restype = C.c_int
        fn.argtypes = [GI, OP, C.POINTER(BioclimSel), C.POINTER(BioclimOut)]
    for fn in (lib.mcf_runbioclim1_multi, lib.mcf_runbioclim2_multi, lib.mcf_runbioclim3_multi, lib.mcf_runbioclim4_multi):
        fn.restype = C.c_int
        fn.argtypes = [GI, OP, C.POINTER(BioclimSel), C.POINTER(Multi), C.POINTER(BioclimOut)]
    lib.mcf_snowenv_from_name.restype = C.c_int32
    lib.mcf_snowenv_from_name.argtypes = [C.c_char_p]
    SI = C.POINTER(SnowInputs)
    for fn in (lib.mcf_gridmodelsnow1, lib.mcf_gridmodelsnow2):
        fn.restype = C.c_int
        fn.argtypes = [SI, C.POINTER(SnowModelOut), C.c_int32]
    for fn in (lib.mcf_gridmicrosnow1, lib.mcf_gridmicrosnow2):
        fn.restype = C.c_int
        fn.argtypes = [SI, C.POINTER(Snowm), C.c_double, C.c_double, C.POINTER(C.c_int32 * NOUT), OU, C.c_int32]
    PW, OT = C.POINTER(PointWeather), C.POINTER(Obstime)
    lib.mcf_bigleaf.restype = C.c_int
    lib.mcf_bigleaf.argtypes = [C.c_int64, OT, PW, c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(BigLeafOut)]
    lib.mcf_soilm.restype = C.c_int
    lib.mcf_soilm.argtypes = [C.c_int64, PW] + [C.c_double] * 7 + [c_double_p, C.POINTER(C.c_int64)]
    lib.mcf_pointmprocess.restype = C.c_int
    lib.mcf_pointmprocess.argtypes = [C.c_int64] + [c_double_p] * 7 + [C.c_double] * 7 + [c_double_p] * 6
    lib.mcf_weatherhgt.restype = C.c_int
    lib.mcf_weatherhgt.argtypes = [C.c_int64, OT, PW] + [C.c_double] * 5 + [c_double_p] * 3
    lib.mcf_pointmodelsnow.restype = C.c_int
    lib.mcf_pointmodelsnow.argtypes = [C.c_int64, C.POINTER(Obstime), C.POINTER(PointWeather), c_double_p, c_double_p,
                                       C.c_int32, C.c_double, C.c_double, C.POINTER(PointSnowOut)]
    lib.mcf_man.restype = C.c_int
    lib.mcf_man.argtypes = [C.c_int64, c_double_p, C.c_int32, c_double_p]
    lib.mcf_flowacc.restype = C.c_int
    lib.mcf_flowacc.argtypes = [C.c_int64, C.c_int64, c_double_p, c_double_p]
    lib.mcf_topidx.restype = C.c_int
    lib.mcf_topidx.argtypes = [C.c_int64, C.c_int64, c_double_p, C.c_double, C.c_double, c_double_p]
    lib.mcf_nc_create.restype = C.c_int
    lib.mcf_nc_create.argtypes = [C.c_char_p, C.POINTER(NcSpec), C.POINTER(C.c_void_p)]
    lib.mcf_nc_write_host.restype = C.c_int
    lib.mcf_nc_write_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(c_double_p * 10)]
    lib.mcf_nc_write_plan.restype = C.c_int
    lib.mcf_nc_write_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                      C.POINTER(C.c_float)]
    lib.mcf_nc_close.restype = C.c_int
    lib.mcf_nc_close.argtypes = [C.c_void_p]
    lib.mcf_applycpp3.restype = C.c_int
    lib.mcf_applycpp3.argtypes = [c_double_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_double_p,
                                  C.c_int32]
    lib.mcf_snowplan_create.restype = C.c_int
    lib.mcf_snowplan_create.argtypes = [C.POINTER(SnowDriverIn), C.c_int64, C.c_int64, C.c_int32, C.POINTER(P)]
    lib.mcf_snowplan_destroy.restype = None
    lib.mcf_snowplan_destroy.argtypes = [P]
    lib.mcf_snowplan_chunks.restype = C.c_int32
    lib.mcf_snowplan_chunks.argtypes = [P]
    lib.mcf_snowplan_surface.restype = C.c_int
    lib.mcf_snowplan_surface.argtypes = [P, c_double_p]
    lib.mcf_snowplan_apply3.restype = C.c_int
    lib.mcf_snowplan_apply3.argtypes = [P, C.c_int32, C.c_int32, c_double_p, c_double_p]
    lib.mcf_snowplan_handover.restype = C.c_int
    lib.mcf_snowplan_handover.argtypes = [P, c_double_p]
    lib.mcf_snowplan_surface_partial.restype = C.c_int
    lib.mcf_snowplan_surface_partial.argtypes = [P, c_double_p, c_double_p]
    lib.mcf_snowplan_prepare_chunk.restype = C.c_int
    lib.mcf_snowplan_prepare_chunk.argtypes = [P, C.c_int32, c_double_p, C.c_int32, C.c_int32, C.c_double, c_double_p,
                                               c_double_p]
    lib.mcf_snowplan_pack_halo.restype = C.c_int
    lib.mcf_snowplan_pack_halo.argtypes = [P, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.mcf_snowplan_prepare_chunk_dev.restype = C.c_int
    lib.mcf_snowplan_prepare_chunk_dev.argtypes = [P, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, c_double_p,
                                                   c_double_p]
    lib.mcf_snowplan_run_chunk.restype = C.c_int
    lib.mcf_snowplan_run_chunk.argtypes = [P, C.c_int32, C.c_double, C.POINTER(SnowDriverOut)]
    lib.mcf_snowmodel1.restype = C.c_int
    lib.mcf_snowmodel1.argtypes = [C.POINTER(SnowDriverIn), C.POINTER(SnowDriverOut), C.c_int32]
    lib.mcf_snowmodel2.restype = C.c_int
    lib.mcf_snowmodel2.argtypes = [C.POINTER(SnowDriverIn), C.POINTER(SnowDriverOut), C.c_int32]
    lib.mcf_snowmodel1_multi.restype = C.c_int
    lib.mcf_snowmodel1_multi.argtypes = [C.POINTER(SnowDriverIn), C.POINTER(SnowDriverOut), C.POINTER(Multi)]
    if hasattr(lib, "mcf_runmicrosnow1"):     # (absent from an older library named by MCF_LIB for an A/B run)
        MI, SO = C.POINTER(MicrosnowIn), C.POINTER(SnowDriverOut)
        lib.mcf_runmicrosnow1.restype = C.c_int
        lib.mcf_runmicrosnow1.argtypes = [MI, OP, OU, SO]
        lib.mcf_runmicrosnow2.restype = C.c_int
        lib.mcf_runmicrosnow2.argtypes = [MI, OP, OU, SO]
        lib.mcf_plan_set_mxtc_days.restype = C.c_int
        lib.mcf_plan_set_mxtc_days.argtypes = [P, C.POINTER(GridInputs), c_int32_p, C.c_int32]
        lib.mcf_runmicrosnow1_multi.restype = C.c_int
        lib.mcf_runmicrosnow1_multi.argtypes = [MI, OP, C.POINTER(Multi), OU, SO]
        lib.mcf_snowrun_create.restype = C.c_int
        lib.mcf_snowrun_create.argtypes = [MI, OP, C.POINTER(Multi), C.POINTER(P)]
        lib.mcf_snowrun_destroy.restype = None
        lib.mcf_snowrun_destroy.argtypes = [P]
        lib.mcf_snowrun_stats.restype = C.c_int
        lib.mcf_snowrun_stats.argtypes = [P, C.POINTER(C.c_int64)]
        lib.mcf_snowrun_keep.restype = C.c_int
        lib.mcf_snowrun_keep.argtypes = [P, C.c_int64]
        lib.mcf_snowplan_set_keep_budget.restype = C.c_int
        lib.mcf_snowplan_set_keep_budget.argtypes = [P, C.c_int64]
        lib.mcf_snowrun_days.restype = C.c_int32
        lib.mcf_snowrun_days.argtypes = [P]
        lib.mcf_snowrun_pass1.restype = C.c_int
        lib.mcf_snowrun_pass1.argtypes = [P, SO, c_int32_p, c_int32_p]
        lib.mcf_snowrun_pass2.restype = C.c_int
        lib.mcf_snowrun_pass2.argtypes = [P, C.POINTER(SnowInputs), C.c_double, OU]
        lib.mcf_snowplan_run_chunk_pitched.restype = C.c_int
        lib.mcf_snowplan_run_chunk_pitched.argtypes = [P, C.c_int32, C.c_double, SO, C.c_int64]
        lib.mcf_snowplan_chunk_af.restype = C.c_int
        lib.mcf_snowplan_chunk_af.argtypes = [P, C.c_int32, C.POINTER(C.c_int32)]
    lib.mcf_precompute_terrain.restype = C.c_int
    lib.mcf_precompute_terrain.argtypes = [C.POINTER(TerrainIn), C.POINTER(TerrainOut), C.c_int32]
    lib.mcf_precompute_terrain_multi.restype = C.c_int
    lib.mcf_precompute_terrain_multi.argtypes = [C.POINTER(TerrainIn), C.POINTER(TerrainOut), C.POINTER(Multi)]
    if lib.mcf_abi_version() != ABI_VERSION:
        raise McfError("libmcfhip ABI version mismatch")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().mcf_last_error()
        raise McfError(f"libmcfhip error {status}: {msg.decode() if msg else '?'}")
