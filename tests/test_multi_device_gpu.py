"""One process, several devices (include/mcf.h mcf_runmicro1_multi / mcf_runmicro2_multi): the raster in row blocks, each
block solved by its device's host thread in place through the row pitch.  On a one-GPU box the blocks are time-sliced on
device 0 — the partition, the whole-raster twi mean installed in every block, the pitched uploads and fetches are the same
code; the result must be the single-device one BIT FOR BIT (no scaling is measured here: unmeasured on N > 1 hardware)."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import runmicro1Cpp, runmicro2Cpp

pytestmark = pytest.mark.gpu
ARGS = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact",
        "complete", "mat", "out")


def _bits_equal(a, b):
    assert list(a) == list(b)
    for k in a:
        assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), k


@pytest.mark.parametrize("rows,cols,T,reqhgt,nb", [(67, 23, 72, 0.05, 3), (40, 31, 55, 1.0, 4), (9, 12, 48, 0.05, 9),
                                                    (53, 17, 72, -0.08, 2), (120, 8, 24, 0.0, 5)])
def test_row_blocks_on_one_device_equal_the_whole_raster_bitwise(rows, cols, T, reqhgt, nb):
    a = synthetic.workload(rows, cols, T, reqhgt=reqhgt, start_doy=160, variety=True, na_frac=0.08)
    whole = runmicro1Cpp(*[a[k] for k in ARGS])
    parts = runmicro1Cpp(*[a[k] for k in ARGS], devices=[0], n_blocks=nb)
    _bits_equal(whole, parts)


def test_large_row_blocks_take_the_pinned_ring_scatter_bitwise():
    """>= 64 MB per block and variable: the pitched fetch goes through HostPipe::copy_pitched (contiguous DMA into the pinned
    ring, rows scattered by the copy threads) instead of hipMemcpy2D; odd block heights, so no row run is page-aligned"""
    a = synthetic.workload(1030, 131, 240, reqhgt=0.05, start_doy=160, variety=True, na_frac=0.03)
    out = [1, 0, 0, 0, 0, 0, 1, 0, 0, 0]
    args = [a[k] for k in ARGS[:-1]] + [out]
    whole = runmicro1Cpp(*args)
    _bits_equal(whole, runmicro1Cpp(*args, devices=[0], n_blocks=3))
    _bits_equal(whole, runmicro1Cpp(*args, devices=[0, 0], n_blocks=2))


def test_array_forcing_row_blocks_bitwise():
    a = synthetic.workload(45, 19, 72, reqhgt=0.05, start_doy=170, variety=True, array_forcing=True, na_frac=0.05)
    whole = runmicro2Cpp(*[a[k] for k in ARGS])
    parts = runmicro2Cpp(*[a[k] for k in ARGS], devices=[0, 0], n_blocks=3)          # two host threads on the one device
    _bits_equal(whole, parts)


@pytest.mark.parametrize("altcorrect", [0, 2])
def test_coarse_array_forcing_row_blocks_bitwise(altcorrect):
    """array_forcing == 2: the coarse climate grid is shared by the blocks, the cells' positions in it (coarse_rowpos) and the
    fine elevations of the altitude correction are offset by the block's first row"""
    from microclimf_amd.api import runmicro2Cpp_coarse
    a, rp, cp = synthetic.coarse_workload(57, 21, 72, 4, 3, reqhgt=0.05, start_doy=165)
    kw = dict(rowpos=rp, colpos=cp)
    if altcorrect:
        _, _, dtm = synthetic.rasters(57, 21)
        kw.update(altcorrect=altcorrect, dtmc=200.0 + 30.0 * np.arange(12, dtype=np.float64).reshape(4, 3), dtm=dtm)
    args = [a[k] for k in ARGS]
    whole = runmicro2Cpp_coarse(*args, **kw)
    for devices, nb in (([0], 4), ([0, 0], 3)):
        _bits_equal(whole, runmicro2Cpp_coarse(*args, **kw, devices=devices, n_blocks=nb))


def test_time_varying_vegetation_row_blocks_bitwise():
    """mcf_runmicro3_multi / 4_multi: the layered vegetation arrays [rows, cols, layers] go through the same row pitch; a day
    no layer covers stays NA in every block"""
    from microclimf_amd.api import runmicro3Cpp, runmicro4Cpp
    a = synthetic.layered(synthetic.workload(47, 13, 24 * 7, reqhgt=0.05, variety=True, start_doy=150, na_frac=0.05), 3, cover_days=6)
    dfsel = a.pop("dfsel")
    whole = runmicro3Cpp(dfsel, **a)
    parts = runmicro3Cpp(dfsel, **a, devices=[0], n_blocks=4)
    _bits_equal(whole, parts)
    assert np.isnan(parts["Tz"][:, :, 144:]).all() and np.isfinite(parts["Tz"][:, :, :144]).any()
    b = synthetic.layered(synthetic.workload(33, 11, 96, reqhgt=0.05, variety=True, start_doy=170, array_forcing=True), 2)
    dfsel = b.pop("dfsel")
    b["lats"], b["lons"] = b.pop("lat"), b.pop("lon")
    _bits_equal(runmicro4Cpp(dfsel, **b), runmicro4Cpp(dfsel, **b, devices=[0, 0], n_blocks=3))


def test_a_sea_of_na_cells_moves_the_block_boundaries_not_the_result():
    a = synthetic.workload(80, 14, 48, reqhgt=0.05, start_doy=200)
    a["vegp"]["hgt"][:55, :] = np.nan                 # the valid cells sit in the last 25 rows
    whole = runmicro1Cpp(*[a[k] for k in ARGS])
    parts = runmicro1Cpp(*[a[k] for k in ARGS], devices=[0], n_blocks=4)
    _bits_equal(whole, parts)


def test_multi_argument_checks():
    from microclimf_amd import McfError
    a = synthetic.workload(12, 12, 24, reqhgt=0.05)
    with pytest.raises(McfError, match="device ordinal"):
        runmicro1Cpp(*[a[k] for k in ARGS], devices=[7])


@pytest.mark.parametrize("rows,cols,res,nb", [(420, 37, 1.0, 3), (300, 24, 2.5, 5), (131, 64, 1.0, 2)])
def test_terrain_row_blocks_on_one_device_equal_the_whole_raster_bitwise(rows, cols, res, nb):
    """mcf_precompute_terrain_multi: the whole raster cut into row blocks, each with the halo its stencils need gathered in the
    library — slope, aspect, 24 horizons, sky view and 8 wind-shelter coefficients are bit for bit the single-device call's"""
    from microclimf_amd.terrain import precompute_terrain
    from test_terrain_cpu import synth_dtm
    z = synth_dtm(rows, cols)
    z[3, 4] = z[rows // 2, cols // 3] = np.nan
    want = precompute_terrain(z, res, 2.0)
    got = precompute_terrain(z, res, 2.0, devices=[0], n_blocks=nb)
    for k, w in want.items():
        assert np.array_equal(got[k], w, equal_nan=True), k
    two = precompute_terrain(z, res, 2.0, devices=[0, 0], n_blocks=nb + 1, what=("hor", "svfa"))      # two host threads on one device
    assert set(two) == {"hor", "svfa"} and np.array_equal(two["hor"], want["hor"], equal_nan=True)


def test_terrain_row_blocks_random_geometries_bitwise():
    """seeded sweep over raster shapes, cell sizes, aggregation factors and block counts (blocks narrower than the halo, a
    single row per block, more blocks than rows asked for)"""
    from microclimf_amd.terrain import precompute_terrain
    from test_terrain_cpu import synth_dtm
    rng = np.random.default_rng(11)
    for _ in range(8):
        rows, cols = int(rng.integers(3, 260)), int(rng.integers(2, 40))
        res, agg, nb = float(rng.choice([0.5, 1.0, 2.5, 30.0])), int(rng.choice([1, 4, 10])), int(rng.integers(2, 12))
        z = synth_dtm(rows, cols)
        z[rng.random((rows, cols)) < 0.03] = np.nan
        want = precompute_terrain(z, res, 2.0, agg=agg)
        got = precompute_terrain(z, res, 2.0, agg=agg, devices=[0], n_blocks=nb)
        for k, w in want.items():
            assert np.array_equal(got[k], w, equal_nan=True), (k, rows, cols, res, agg, nb)


def test_terrain_multi_argument_checks():
    from microclimf_amd import McfError
    from microclimf_amd.terrain import precompute_terrain
    from test_terrain_cpu import synth_dtm
    z = synth_dtm(60, 12)
    with pytest.raises(McfError, match="whole raster"):
        precompute_terrain(z, 1.0, 2.0, halo_north=5, halo_south=5, devices=[0])
    with pytest.raises(McfError, match="ordinal"):
        precompute_terrain(z, 1.0, 2.0, devices=[99])
