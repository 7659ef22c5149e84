"""mcf_precompute_terrain (HIP) against the numpy terrain oracle."""
import numpy as np
import pytest

from microclimf_amd.terrain import precompute_terrain
from oracle import terrain_oracle as TO
from test_terrain_cpu import synth_dtm

pytestmark = pytest.mark.gpu


def check(got, want, sl=slice(None)):
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w[sl], rtol=0, atol=1e-10, err_msg=k)


@pytest.mark.parametrize("rows,cols,res", [(57, 43, 1.0), (131, 64, 2.5), (20, 9, 1.0)])
def test_whole_raster(rows, cols, res):
    z = synth_dtm(rows, cols)
    z[3, 4] = np.nan                      # NA elevation: 0 for the stencils, NA -> 0 slope around it
    got = precompute_terrain(z, res, 2.0)
    check(got, TO.terrain(z, res, 2.0))


def test_row_block_with_halos_equals_whole():
    z = synth_dtm(420, 37)
    want = TO.terrain(z, 1.0, 2.0)
    row0, rows, hn, hs = 150, 110, 128, 128
    got = precompute_terrain(z[row0 - hn:row0 + rows + hs], 1.0, 2.0, halo_north=hn, halo_south=hs, row0=row0,
                             rows_total=420)
    check(got, want, slice(row0, row0 + rows))
    # a block touching the northern raster edge needs no northern halo
    got = precompute_terrain(z[0:90 + 128], 1.0, 2.0, halo_north=0, halo_south=128, row0=0, rows_total=420)
    check(got, want, slice(0, 90))


def test_insufficient_halo_is_rejected():
    from microclimf_amd import McfError
    z = synth_dtm(300, 16)
    with pytest.raises(McfError, match="halo"):
        precompute_terrain(z[100:200], 1.0, 2.0, halo_north=10, halo_south=10, row0=110, rows_total=300)
