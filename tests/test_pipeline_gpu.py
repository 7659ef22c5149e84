"""Solver -> netCDF file end to end (microclimf_amd.pipeline.run_to_nc, the body of runmicro_big's tile loop) against
the oracle's outputs packed by a numpy `atonc`.  The packing rounds value x 100 to an integer, so a device / oracle
difference of 1e-13 may flip a tie: the file may differ from the oracle's packing by one unit in a handful of values."""
import numpy as np
import pytest
from scipy.io import netcdf_file

from microclimf_amd import ncsink, pipeline, synthetic

pytestmark = pytest.mark.gpu
SCALE = {"Tz": 100, "tleaf": 100, "relhum": 1, "soilm": 100, "windspeed": 100, "Rdirdown": 1, "Rdifdown": 1,
         "Rlwdown": 1, "Rswup": 1, "Rlwup": 1}


def pack(a, rd):
    with np.errstate(invalid="ignore"):
        q = np.rint(np.transpose(a, (1, 0, 2)) * rd)
    return np.where(np.isfinite(q), q, -9999).astype(np.int64)


@pytest.mark.parametrize("reqhgt,af", [(0.05, False), (0.0, False), (2.5, True)])
def test_year_slice_to_file_equals_packed_oracle(reqhgt, af, oracle, tmp_path):
    rows, cols, T = 23, 31, 7 * 24
    a = synthetic.workload(rows, cols, T, reqhgt=reqhgt, variety=True, start_doy=200, na_frac=0.04, array_forcing=af)
    dtm = {"xmin": 0.0, "xmax": cols * 2.0, "ymin": 10.0, "ymax": 10.0 + rows * 2.0, "res": 2.0, "crs": "local"}
    names = ncsink.default_vars(reqhgt) + (("soilm",) if reqhgt > 0 else ())
    info = pipeline.run_to_nc(a, tmp_path / "t.nc", dtm, vars=names, days_per_chunk=3, array_forcing=af)
    assert info["steps"] == T and info["vars"] == names
    want = oracle.run_grid(**a, array_forcing=af)
    f = netcdf_file(str(tmp_path / "t.nc"), "r", mmap=False)
    assert np.array_equal(f.variables["time"][:], ncsink.hours_since_epoch(a["obstime"]))
    for k in names:
        got = np.transpose(f.variables[k][:], (2, 1, 0)).astype(np.int64)
        w = pack(want[k], SCALE[k])
        d = np.abs(got - w)
        assert d.max() <= 1, k
        assert (d != 0).mean() < 1e-4, (k, (d != 0).mean())
        assert np.array_equal(got == -9999, w == -9999), k
    f.close()


def test_the_reference_container_end_to_end_holds_the_classic_files_values(tmp_path):
    """run_to_nc(format="netcdf4"): solver -> device-packed records -> deflated chunks -> HDF5; read back through the HDF5
    library it holds exactly the values of the classic file of the same run"""
    import h5mini
    if h5mini.load() is None:
        pytest.skip("no HDF5 library on this host")
    rows, cols, T = 23, 31, 4 * 24
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, variety=True, start_doy=200, na_frac=0.04)
    dtm = {"xmin": 0.0, "xmax": cols * 2.0, "ymin": 10.0, "ymax": 10.0 + rows * 2.0, "res": 2.0, "crs": "local"}
    pipeline.run_to_nc(a, tmp_path / "c.nc", dtm, days_per_chunk=3)
    info = pipeline.run_to_nc(a, tmp_path / "h.nc", dtm, days_per_chunk=3, format="netcdf4", deflate_level=4)
    c, h = netcdf_file(str(tmp_path / "c.nc"), "r", mmap=False), h5mini.File(tmp_path / "h.nc")
    for k in info["vars"]:
        assert np.array_equal(h.read(k), c.variables[k][:]), k
        assert h.chunk_and_filters(k) == ((1, rows, cols), [(1, (4,))])
    assert np.array_equal(h.read("time"), c.variables["time"][:]) and h.attr("crs", "crs_wkt") == b"local"
    c.close(); h.close()


def test_below_ground_is_refused(tmp_path):
    a = synthetic.workload(8, 8, 48, reqhgt=-0.1)
    with pytest.raises(ValueError, match="reqhgt < 0"):
        pipeline.run_to_nc(a, tmp_path / "b.nc", {"xmin": 0, "xmax": 8, "ymin": 0, "ymax": 8, "res": 1.0})
