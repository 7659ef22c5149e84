"""The `writetonc` sink's netCDF-4 container (format="netcdf4": the reference's own, ncvar_def(..., compression = 9),
R/dataprep.R:1110-1111) on the host — mcf_nc_create / mcf_nc_write_host, no device.  The file is read back through the HDF5
library's own read path (tests/h5mini.py) and held against a numpy restatement of `atonc` (dataprep.R:1064-1069), against
the dataset definition of writetonc, against the netCDF-4 conventions a netCDF reader relies on (dimension scales and their
attachment, _Netcdf4Dimid, _FillValue of the variable's type, NULLTERM text attributes, creation-order indices), and against
the classic container written from the same arrays."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

import h5mini
from microclimf_amd import _abi, ncsink
from test_ncsink_cpu import DTM, SCALE, atonc, mout_for

pytestmark = pytest.mark.skipif(h5mini.load() is None, reason="no HDF5 library on this host (the netCDF-4 container needs one)")


def test_dataset_definition_values_and_conventions(tmp_path):
    rows, cols, n = 5, 7, 30
    names = ncsink.default_vars(0.05)
    m = mout_for(rows, cols, n, names)
    m["tme"] = {"year": np.full(n, 2024), "month": np.full(n, 3), "day": 21 + np.arange(n) // 24, "hour": (np.arange(n) % 24).astype(float)}
    p = tmp_path / "a4.nc"
    ncsink.writetonc(m, p, DTM, 0.05, format="netcdf4")
    assert p.read_bytes()[:8] == b"\x89HDF\r\n\x1a\n"
    f = h5mini.File(p)
    # ncdf4's order: the dimension variables, the data variables in writetonc's list order, `crs` last (dataprep.R:1159)
    assert f.names_in_creation_order() == ["east", "north", "time", *names, "crs"]
    with pytest.raises(KeyError):
        f.attr(None, "_NCProperties")                  # libnetcdf's own provenance stamp: optional, and not ours to write
    # dimensions: scales named after themselves, with ncdf4's dimension ids and coordinate values
    for dimid, (d, length) in enumerate((("east", cols), ("north", rows), ("time", n))):
        assert f.is_scale(d) and f.scale_name(d) == d and f.attr(d, "CLASS") == b"DIMENSION_SCALE"
        assert f.attr(d, "_Netcdf4Dimid") == dimid and f.shape(d) == (length,) and f.is_type(d, "H5T_IEEE_F64LE_g")
    assert np.array_equal(f.read("east"), 1012.5 + 25 * np.arange(cols))
    assert np.array_equal(f.read("north"), 5012.5 + 25 * np.arange(rows))
    t = f.read("time")
    assert t[0] == 1710979200 / 3600 and np.array_equal(np.diff(t), np.ones(n - 1))
    assert f.attr("east", "units") == b"metres" and f.attr("north", "long_name") == b"Northings"
    assert f.attr("time", "units") == b"hours since 1970-01-01 00:00" and f.attr("time", "calendar") == b"gregorian"
    assert f.attr("time", "standard_name") == b"time"
    assert f.shape("crs") == () and f.read("crs")[()] == 1
    assert f.attr("crs", "crs_wkt") == b"EPSG:27700 (test)" and f.attr("crs", "grid_mapping_name") == b"longitude_latitude"
    longname = {"Tz": b"Air temperature at height 0.05 m", "tleaf": b"Leaf temperature at height 0.05 m",
                "relhum": b"Relative humidity at height 0.05 m", "windspeed": b"Wind speed at height 0.05 m",
                "Rswup": b"Upward shortwave radiation"}
    units = {"Tz": b"deg C x 100", "relhum": b"Percentage", "windspeed": b"m/s x 100", "Rlwdown": b"W/m^2"}
    for k in names:
        assert f.shape(k) == (n, rows, cols) and f.is_type(k, "H5T_STD_I32BE_g")
        chunk, filters = f.chunk_and_filters(k)
        assert chunk == (1, rows, cols) and filters == [(1, (9,))]              # H5Z_FILTER_DEFLATE, level 9
        assert f.attr(k, "_FillValue") == -9999 and f.attr(k, "grid_mapping") == b"crs"
        for dim, scale in enumerate(("time", "north", "east")):
            assert f.num_scales(k, dim) == 1 and f.attached(k, scale, dim)
        if k in longname:
            assert f.attr(k, "long_name") == longname[k]
        if k in units:
            assert f.attr(k, "units") == units[k]
        assert np.array_equal(np.transpose(f.read(k), (2, 1, 0)), atonc(m[k], SCALE[k])), k
    assert (f.read("Tz") == -9999).any()
    f.close()
    # the same arrays in the classic container: identical values through an independent reader
    ncsink.writetonc(m, tmp_path / "a3.nc", DTM, 0.05)
    c = netcdf_file(str(tmp_path / "a3.nc"), "r", mmap=False)
    f = h5mini.File(p)
    for k in names:
        assert np.array_equal(f.read(k), c.variables[k][:]), k
    f.close()
    c.close()


def test_row_strips_edge_chunks_any_order_and_unwritten_steps(tmp_path, monkeypatch):
    """rows x cols above the strip limit would need a 1 Mi-cell raster; the strip height is derived from the column count, so a
    wide raster gives several strips with a ragged last one.  Pieces arrive out of order, from several deflate threads, and a
    step that is never written reads as the fill value."""
    rows, cols, n = 5, 300000, 3
    rng = np.random.default_rng(3)
    tz = np.asfortranarray(rng.integers(-2000, 3000, (rows, cols, n)) / 100.0)
    tz[:, ::17, :] = np.nan
    east, north = np.arange(cols) + 0.5, np.arange(rows) + 0.5
    p = tmp_path / "wide.nc"
    monkeypatch.setenv("MCF_NC_DEFLATE_THREADS", "3")
    with ncsink.NcWriter(p, rows, cols, np.arange(n) + 1.0, east, north, -0.1, ("Tz",), format="netcdf4", deflate_level=1) as w:
        w.write_host(2, {"Tz": tz[:, :, 2:]})
        w.write_host(0, {"Tz": tz[:, :, :1]})
    f = h5mini.File(p)
    chunk, filters = f.chunk_and_filters("Tz")
    assert chunk == (1, 3, cols) and filters == [(1, (1,))]                    # 2^20 // 300000 = 3 rows per strip: strips of 3 + 2 rows
    got = np.transpose(f.read("Tz"), (2, 1, 0))
    want = atonc(tz, 100)
    assert np.array_equal(got[:, :, 0], want[:, :, 0]) and np.array_equal(got[:, :, 2], want[:, :, 2])
    assert (got[:, :, 1] == -9999).all()
    assert f.storage_size("Tz") < 0.75 * 2 * rows * cols * 4                   # deflated (random digits, level 1)
    f.close()


def test_no_compression_surface_and_reference_puts_only(tmp_path):
    rows, cols, n = 4, 3, 5
    names = ncsink.default_vars(0.0)
    m = mout_for(rows, cols, n, names, 1)
    m["tme"] = np.arange(n) + 400000.0
    dtm = {"xmin": 0, "xmax": 3, "ymin": 0, "ymax": 4, "res": 1.0}
    p = tmp_path / "q4.nc"
    ncsink.writetonc(m, p, dtm, 0.0, reference_puts_only=True, format="netcdf4", deflate_level=-1)
    f = h5mini.File(p)
    assert f.attr("Tz", "long_name") == b"Soil surface temperature"
    assert f.attr("soilm", "units") == b"Volume percentage soil moisture in top 10 cm of soil"
    for k in names:
        chunk, filters = f.chunk_and_filters(k)
        assert chunk == (1, rows, cols) and filters == []
        got = np.transpose(f.read(k), (2, 1, 0))
        if k.startswith("R") or k == "soilm":                                   # dataprep.R:1161, 1163-1167 never run
            assert (got == -9999).all(), k
        else:
            assert np.array_equal(got, atonc(m[k], SCALE[k])), k
    f.close()


def test_h5dump_reads_the_file(tmp_path):
    """the HDF5 distribution's own tool, where it is installed: the header lists the deflate filter and the scales"""
    exe = next((e for e in ("h5dump", "/opt/conda/bin/h5dump") if subprocess.run(["sh", "-c", f"command -v {e}"], capture_output=True).returncode == 0), None)
    if exe is None:
        pytest.skip("no h5dump")
    m = mout_for(3, 4, 2, ("Tz",), 5)
    m["tme"] = np.arange(2) + 1.0
    p = tmp_path / "d.nc"
    ncsink.writetonc(m, p, {"xmin": 0, "xmax": 4, "ymin": 0, "ymax": 3, "res": 1.0}, -0.2, vars=("Tz",), format="netcdf4")
    r = subprocess.run([exe, "-H", "-p", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "COMPRESSION DEFLATE { LEVEL 9 }" in r.stdout and "DIMENSION_SCALE" not in r.stderr
    assert 'ATTRIBUTE "DIMENSION_LIST"' in r.stdout and 'ATTRIBUTE "REFERENCE_LIST"' in r.stdout
    r = subprocess.run([exe, "-d", "/Tz", "-y", str(p)], capture_output=True, text=True)      # full data read + inflate
    assert r.returncode == 0, r.stderr


def test_errors(tmp_path, monkeypatch):
    east, north = ncsink.coords_from_extent(0, 3, 0, 2, 1)
    with pytest.raises(ValueError, match="format"):
        ncsink.NcWriter(tmp_path / "x.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",), format="hdf")
    with pytest.raises(_abi.McfError, match="deflate_level"):
        ncsink.NcWriter(tmp_path / "x.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",), format="netcdf4", deflate_level=12)
    with pytest.raises(_abi.McfError, match="cannot create"):
        ncsink.NcWriter(tmp_path / "no_such_dir" / "x.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",), format="netcdf4")
    with pytest.raises(_abi.McfError, match="at least one time step"):
        ncsink.NcWriter(tmp_path / "x.nc", 2, 3, np.arange(0.0), east, north, 1.0, ("Tz",), format="netcdf4")
    with ncsink.NcWriter(tmp_path / "y.nc", 2, 3, np.arange(3.0), east, north, 1.0, ("Tz",), format="netcdf4") as w:
        with pytest.raises(_abi.McfError, match="step range"):
            w.write_host(2, {"Tz": np.zeros((2, 3, 2))})


def test_a_host_without_hdf5_gets_an_error_that_says_so(tmp_path):
    """the binding is made once per process, so the missing-library case runs in a child process: MCF_HDF5_LIB names THE
    library to use, and one that cannot be loaded is an error from mcf_nc_create; the classic container needs no library"""
    code = (
        "import numpy as np, sys\n"
        "from microclimf_amd import ncsink, _abi\n"
        "try:\n"
        "    ncsink.NcWriter(sys.argv[1], 2, 3, np.arange(3.0), np.arange(3.0), np.arange(2.0), 1.0, ('Tz',), format='netcdf4')\n"
        "    print('created')\n"
        "except _abi.McfError as e:\n"
        "    print('ERR', e)\n"
        "w = ncsink.NcWriter(sys.argv[1], 2, 3, np.arange(3.0), np.arange(3.0), np.arange(2.0), 1.0, ('Tz',)); w.close(); print('classic ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCF_HDF5_LIB=str(tmp_path / "nothing.so"))
    r = subprocess.run([os.sys.executable, "-c", code, str(tmp_path / "z.nc")], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stderr
    assert "created" not in r.stdout and "classic ok" in r.stdout
    assert "ERR" in r.stdout and "HDF5 library" in r.stdout and "nothing.so" in r.stdout
