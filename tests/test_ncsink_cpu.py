"""The `writetonc` sink's file side on the host (mcf_nc_create / mcf_nc_write_host, no device): the file is read back
with scipy's netCDF reader — an independent implementation of the classic format — and compared with a numpy
restatement of writetonc's `atonc` packing (R/dataprep.R:1064-1069) and of its dataset definition."""
import numpy as np
import pytest
from scipy.io import netcdf_file

from microclimf_amd import _abi, ncsink

SCALE = {"Tz": 100, "tleaf": 100, "relhum": 1, "soilm": 100, "windspeed": 100, "Rdirdown": 1, "Rdifdown": 1,
         "Rlwdown": 1, "Rswup": 1, "Rlwup": 1}


def atonc(a, rd):
    """aperm(a, c(2,1,3)); round half even; as.integer; NA -> the variable's missval on ncvar_put"""
    with np.errstate(invalid="ignore"):
        q = np.rint(np.transpose(a, (1, 0, 2)) * rd)
    return np.where(np.isfinite(q), q, -9999).astype(np.int32)


def mout_for(rows, cols, n, names, seed=0):
    rng = np.random.default_rng(seed)
    m = {}
    for k in names:
        a = rng.uniform(-30, 40, (rows, cols, n))
        a[rng.random((rows, cols)) < 0.1] = np.nan                 # NA cells
        a.flat[::7] = np.round(a.flat[::7], 2) + 0.005              # ties of round(x * 100)
        m[k] = np.asfortranarray(a)
    return m


DTM = {"xmin": 1000.0, "xmax": 1000.0 + 7 * 25, "ymin": 5000.0, "ymax": 5000.0 + 5 * 25, "res": 25.0, "crs": "EPSG:27700 (test)"}


def read(path):
    f = netcdf_file(str(path), "r", mmap=False)
    return f


def test_dataset_definition_and_values(tmp_path):
    rows, cols, n = 5, 7, 30
    names = ncsink.default_vars(0.05)
    m = mout_for(rows, cols, n, names)
    obst = {"year": np.full(n, 2024), "month": np.full(n, 3), "day": 21 + np.arange(n) // 24, "hour": (np.arange(n) % 24).astype(float)}
    m["tme"] = obst
    p = tmp_path / "a.nc"
    ncsink.writetonc(m, p, DTM, 0.05)
    f = read(p)
    assert f.version_byte == 2
    assert list(f.dimensions) == ["east", "north", "time"] and f.dimensions["east"] == cols and f.dimensions["north"] == rows
    assert f.dimensions["time"] is None                                        # record dimension
    assert list(f.variables) == ["east", "north", "crs", "time", *names]
    assert np.array_equal(f.variables["east"][:], 1012.5 + 25 * np.arange(cols))
    assert np.array_equal(f.variables["north"][:], 5012.5 + 25 * np.arange(rows))    # ascending, as dataprep.R:1073
    assert f.variables["east"].units == b"metres" and f.variables["north"].long_name == b"Northings"
    t = f.variables["time"]
    assert t.units == b"hours since 1970-01-01 00:00" and t.calendar == b"gregorian" and t.standard_name == b"time"
    assert t[0] == 1710979200 / 3600 and np.array_equal(np.diff(t[:]), np.ones(n - 1))      # 2024-03-21 00:00 UTC
    crs = f.variables["crs"]
    assert crs[()] == 1 and crs.crs_wkt == b"EPSG:27700 (test)" and crs.grid_mapping_name == b"longitude_latitude"
    longname = {"Tz": b"Air temperature at height 0.05 m", "tleaf": b"Leaf temperature at height 0.05 m",
                "relhum": b"Relative humidity at height 0.05 m", "windspeed": b"Wind speed at height 0.05 m",
                "Rswup": b"Upward shortwave radiation"}
    units = {"Tz": b"deg C x 100", "relhum": b"Percentage", "windspeed": b"m/s x 100", "Rlwdown": b"W/m^2"}
    for k in names:
        v = f.variables[k]
        assert v.dimensions == ("time", "north", "east") and v.data.dtype == np.dtype(">i4")
        assert v._FillValue == -9999 and v.grid_mapping == b"crs"
        if k in longname:
            assert v.long_name == longname[k]
        if k in units:
            assert v.units == units[k]
        want = atonc(m[k], SCALE[k])                                            # [east, north, time]
        assert np.array_equal(np.transpose(v[:], (2, 1, 0)), want), k
    assert (f.variables["Tz"][:] == -9999).any()
    f.close()


def test_reference_puts_only_leaves_the_unput_variables_at_missval(tmp_path):
    rows, cols, n = 4, 3, 5
    names = ncsink.default_vars(2.0)
    m = mout_for(rows, cols, n, names, 1)
    m["tme"] = np.arange(n) + 400000.0
    dtm = {"xmin": 0, "xmax": 3, "ymin": 0, "ymax": 4, "res": 1.0}
    ncsink.writetonc(m, tmp_path / "q.nc", dtm, 2.0, reference_puts_only=True)
    f = read(tmp_path / "q.nc")
    assert f.variables["Tz"].long_name == b"Air temperature at height 2 m"
    for k in names:
        got = np.transpose(f.variables[k][:], (2, 1, 0))
        if k.startswith("R"):                                                   # dataprep.R:1163-1167 never runs
            assert (got == -9999).all(), k
        else:
            assert np.array_equal(got, atonc(m[k], SCALE[k])), k
    f.close()


def test_chunks_in_any_order_surface_and_below(tmp_path):
    rows, cols, n = 6, 4, 48
    for reqhgt, tlong, sunits in ((0.0, b"Soil surface temperature", b"Volume percentage soil moisture in top 10 cm of soil"),
                                  (-0.1, b"Soil temperature at depth 0.1 m", b"Percentage volume")):
        names = ncsink.default_vars(reqhgt)
        m = mout_for(rows, cols, n, names, 2)
        east, north = ncsink.coords_from_extent(0, cols * 10, 0, rows * 10, 10)
        p = tmp_path / f"c{reqhgt}.nc"
        with ncsink.NcWriter(p, rows, cols, np.arange(n) + 1.0, east, north, reqhgt) as w:
            assert w.vars == names
            w.write_host(24, {k: m[k][:, :, 24:] for k in names})
            w.write_host(0, {k: m[k][:, :, :24] for k in names})
        f = read(p)
        assert f.variables["Tz"].long_name == tlong and f.variables["soilm"].units == sunits
        for k in names:
            assert np.array_equal(np.transpose(f.variables[k][:], (2, 1, 0)), atonc(m[k], SCALE[k])), k
        assert np.array_equal(f.variables["time"][:], np.arange(n) + 1.0)
        f.close()


def test_errors(tmp_path):
    east, north = ncsink.coords_from_extent(0, 3, 0, 2, 1)
    with pytest.raises(_abi.McfError, match="tleaf"):
        ncsink.NcWriter(tmp_path / "x.nc", 2, 3, np.arange(3.0), east, north, 0.0, ("Tz", "tleaf"))
    with pytest.raises(_abi.McfError, match="Rswup"):
        ncsink.NcWriter(tmp_path / "x.nc", 2, 3, np.arange(3.0), east, north, -0.5, ("Tz", "Rswup"))
    with pytest.raises(_abi.McfError, match="cannot create"):
        ncsink.NcWriter(tmp_path / "no_such_dir" / "x.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",))
    with ncsink.NcWriter(tmp_path / "y.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",)) as w:
        with pytest.raises(_abi.McfError, match="step range"):
            w.write_host(2, {"Tz": np.zeros((2, 3, 2))})
        with pytest.raises(ValueError):
            w.write_host(0, {"Tz": np.zeros((3, 2, 1))})
