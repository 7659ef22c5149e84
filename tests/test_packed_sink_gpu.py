"""The writetonc-packed output sink: mcf_plan_fetch_packed against a numpy
restatement of `atonc` (R/dataprep.R:1064-1069): aperm(a, c(2,1,3)); round(a * rd, 0) (half to even);
as.integer (NA -> NA_integer_).  Integer output: the comparison is bit-exact."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import Plan

pytestmark = pytest.mark.gpu
NA_INT = np.iinfo(np.int32).min


def atonc(a, rd):
    with np.errstate(invalid="ignore"):
        x = np.rint(np.transpose(a, (1, 0, 2)) * rd)          # numpy rint = round half to even, as R's round(x, 0)
    out = np.full(x.shape, NA_INT, dtype=np.int32)
    ok = np.isfinite(x) & (np.abs(x) < 2147483648.0)
    out[ok] = x[ok].astype(np.int32)
    return np.asfortranarray(out)


@pytest.mark.parametrize("rows,cols", [(37, 29), (64, 32), (5, 70)])
def test_packed_fetch_is_atonc_of_plain_fetch(rows, cols):
    a = synthetic.workload(rows, cols, 72, reqhgt=0.05, variety=True, start_doy=120, na_frac=0.05)
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=3) as p:
        p.run_days(0, 3)
        p.sync()
        for var in ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rlwup"):
            plain = p.fetch(0, var, 0, 72)
            packed = p.fetch_packed(0, var, 0, 72)
            assert packed.dtype == np.int32 and packed.shape == (cols, rows, 72)
            want = atonc(plain, Plan.NC_SCALE[var])
            assert np.array_equal(packed, want), var
            assert (packed[np.isnan(np.transpose(plain, (1, 0, 2)))] == NA_INT).all()
        # a sub-range of steps and an explicit scale
        part = p.fetch_packed(0, "Tz", 24, 30, scale=10.0)
        assert np.array_equal(part, atonc(p.fetch(0, "Tz", 24, 30), 10.0))


def test_packed_fetch_round_half_even_and_range():
    """ties round to even (R >= 4.0 round()), values beyond int32 become NA like as.integer()"""
    a = synthetic.workload(8, 8, 24, reqhgt=0.05, na_frac=0.0)
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=1) as p:
        p.run_days(0, 1)
        p.sync()
        plain = p.fetch(0, "Rlwdown", 0, 24)
        huge = p.fetch_packed(0, "Rlwdown", 0, 24, scale=1e8)          # ~3e10 > 2^31
        assert (huge == NA_INT).all()
        # scale chosen so that value * scale lands exactly on .5 for one element
        v = plain[3, 4, 7]
        s = 2.5 / v
        got = p.fetch_packed(0, "Rlwdown", 0, 24, scale=s)[4, 3, 7]
        assert got == int(np.rint(v * s))


def test_large_fetch_through_the_host_pipe_is_bitwise_the_plain_copy():
    """fetches of >= 64 MB go through the pinned ring + host copy threads (mcf_hostpipe.hpp); smaller ones through
    a plain hipMemcpy: the same slot fetched both ways must be bit-identical, including odd tail sizes"""
    rows, cols, T = 500, 517, 48
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, start_doy=100, out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=2) as p:
        p.run_days(0, 2)
        p.sync()
        for var in ("Tz", "soilm"):
            big = p.fetch(0, var, 0, T)                                   # 99 MB: host pipe
            assert big.nbytes >= 64 << 20
            parts = [p.fetch(0, var, k, 8) for k in range(0, T, 8)]       # 16.5 MB each: plain copies
            assert parts[0].nbytes < 64 << 20
            small = np.concatenate(parts, axis=2)
            assert np.array_equal(big.view(np.uint64), small.view(np.uint64)), var
            odd = p.fetch(0, var, 3, 41)                                   # 84.8 MB, not a multiple of the piece size
            assert np.array_equal(odd.view(np.uint64), small[:, :, 3:44].view(np.uint64)), var


# ---- the writetonc file sink: records packed and byte-ordered on the device (k_pack_nc) --------------------------
def _nc_inputs(rows, cols, T):
    from microclimf_amd import ncsink
    east, north = ncsink.coords_from_extent(0.0, cols * 5.0, 100.0, 100.0 + rows * 5.0, 5.0)
    return ncsink, east, north, 473352.0 + np.arange(T)


@pytest.mark.parametrize("rows,cols,puts_only", [(37, 29, False), (64, 32, True), (5, 70, False)])
def test_nc_file_from_the_device_ring_is_bytewise_the_host_written_file(rows, cols, puts_only, tmp_path):
    """two routes to the same file: device ring -> k_pack_nc -> records (chunks of 1 and 2 days, second chunk first),
    and plain fp64 fetch -> mcf_nc_write_host (itself checked against scipy + numpy in tests/test_ncsink_cpu.py)"""
    T = 72
    ncsink, east, north, hours = _nc_inputs(rows, cols, T)
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, variety=True, start_doy=120, na_frac=0.05)
    names = ncsink.default_vars(0.05) + ("soilm",)
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=3) as p:
        p.run_days(0, 3)
        p.sync()
        with ncsink.NcWriter(tmp_path / "dev.nc", rows, cols, hours, east, north, 0.05, names, "wkt", puts_only) as w:
            ms = w.write_plan(p, 0, 24, 24, 48, timing=True)
            w.write_plan(p, 0, 0, 0, 24)
            assert ms > 0
        with ncsink.NcWriter(tmp_path / "host.nc", rows, cols, hours, east, north, 0.05, names, "wkt", puts_only) as w:
            w.write_host(0, {k: p.fetch(0, k, 0, T) for k in names})
        tz = p.fetch(0, "Tz", 0, T)
    dev, host = (tmp_path / "dev.nc").read_bytes(), (tmp_path / "host.nc").read_bytes()
    assert len(dev) == len(host) and dev == host
    from scipy.io import netcdf_file
    f = netcdf_file(str(tmp_path / "dev.nc"), "r", mmap=False)
    got = np.transpose(f.variables["Tz"][:], (2, 1, 0))
    want = atonc(tz, 100.0)
    want[want == NA_INT] = -9999
    assert np.array_equal(got, want)
    if puts_only:
        assert (f.variables["Rswup"][:] == -9999).all() and (f.variables["soilm"][:] == -9999).all()
    else:
        assert (f.variables["Rlwup"][:] != -9999).any()
    f.close()


def test_netcdf4_file_from_the_device_ring_holds_the_host_written_values(tmp_path):
    """the reference's container (netCDF-4, deflate 9) fed by the same device records: read back through the HDF5 library
    (tests/h5mini.py) it equals the host-written netCDF-4 file, the classic file and numpy's atonc of the fetched doubles"""
    import h5mini
    if h5mini.load() is None:
        pytest.skip("no HDF5 library on this host")
    rows, cols, T = 37, 29, 72
    ncsink, east, north, hours = _nc_inputs(rows, cols, T)
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, variety=True, start_doy=120, na_frac=0.05)
    names = ncsink.default_vars(0.05) + ("soilm",)
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=3) as p:
        p.run_days(0, 3)
        p.sync()
        with ncsink.NcWriter(tmp_path / "dev4.nc", rows, cols, hours, east, north, 0.05, names, "wkt", format="netcdf4") as w:
            w.write_plan(p, 0, 24, 24, 48)
            w.write_plan(p, 0, 0, 0, 24)
        full = {k: p.fetch(0, k, 0, T) for k in names}
        with ncsink.NcWriter(tmp_path / "host4.nc", rows, cols, hours, east, north, 0.05, names, "wkt", format="netcdf4") as w:
            w.write_host(0, full)
        with ncsink.NcWriter(tmp_path / "host3.nc", rows, cols, hours, east, north, 0.05, names, "wkt") as w:
            w.write_host(0, full)
    from scipy.io import netcdf_file
    dev, host, classic = h5mini.File(tmp_path / "dev4.nc"), h5mini.File(tmp_path / "host4.nc"), netcdf_file(str(tmp_path / "host3.nc"), "r", mmap=False)
    for k in names:
        got = dev.read(k)
        assert np.array_equal(got, host.read(k)) and np.array_equal(got, classic.variables[k][:]), k
        want = atonc(full[k], Plan.NC_SCALE[k])
        want[want == NA_INT] = -9999
        assert np.array_equal(np.transpose(got, (2, 1, 0)), want), k
        assert dev.chunk_and_filters(k) == ((1, rows, cols), [(1, (9,))])
    assert np.array_equal(dev.read("time"), hours)
    dev.close(); host.close(); classic.close()


def test_nc_sink_needs_the_files_variables_in_the_plan(tmp_path):
    from microclimf_amd import _abi
    ncsink, east, north, hours = _nc_inputs(8, 8, 24)
    a = synthetic.workload(8, 8, 24, reqhgt=0.05, out=[1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    with Plan(a["obstime"], a["climdata"], a["pointm"], a["vegp"], a["soilc"], a["reqhgt"], a["zref"], a["lat"],
              a["lon"], a["Sminp"], a["Smaxp"], a["tfact"], True, a["mat"], a["out"], ring_days=1) as p:
        p.run_days(0, 1)
        p.sync()
        with ncsink.NcWriter(tmp_path / "a.nc", 8, 8, hours, east, north, 0.05, ("Tz", "tleaf")) as w:
            with pytest.raises(_abi.McfError, match="not requested"):
                w.write_plan(p, 0, 0, 0, 24)
        with ncsink.NcWriter(tmp_path / "b.nc", 8, 9, hours, np.arange(9.0), north, 0.05, ("Tz",)) as w:
            with pytest.raises(_abi.McfError, match="grid"):
                w.write_plan(p, 0, 0, 0, 24)
        with ncsink.NcWriter(tmp_path / "c.nc", 8, 8, hours, east, north, 0.05, ("Tz",)) as w:
            w.write_plan(p, 0, 0, 0, 24)                                    # Tz alone is fine
