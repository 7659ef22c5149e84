"""mcf_flowacc / mcf_topidx (host code of libmcfhip, mcf_hydro.cpp) against the statement-by-statement restatement
of flowaccCpp and `.topidx` in oracle/hydro_oracle.py.  Integer-valued counts: exact; the index to rounding."""
import numpy as np
import pytest

from microclimf_amd import _abi
from microclimf_amd.terrain import flowaccCpp, topidx
from oracle import hydro_oracle as HO
from test_terrain_cpu import synth_dtm


def rasters():
    rng = np.random.default_rng(11)
    yield "smooth", synth_dtm(37, 29)
    z = synth_dtm(24, 31)
    z[5, 6] = z[17, 20] = np.nan
    z[:, 0] = np.nan                                     # a no-data column at the edge
    yield "with_na", z
    yield "ties_and_plateaus", np.round(synth_dtm(30, 22) / 5.0) * 5.0          # many equal elevations
    yield "pits", np.where(rng.random((20, 20)) < 0.05, -50.0, synth_dtm(20, 20))
    hi = synth_dtm(18, 16) + 9990.0                       # straddles the 9999.99 m ceiling of flowdirCpp
    yield "above_the_ceiling", hi
    yield "single_row", synth_dtm(1, 25)
    yield "single_cell", np.array([[12.0]])
    yield "random", rng.uniform(0, 300, (26, 19))


@pytest.mark.parametrize("name,z", list(rasters()), ids=[n for n, _ in rasters()])
def test_flowacc_equals_the_restatement(name, z):
    got = flowaccCpp(z)
    want = HO.flowacc(z)
    assert np.array_equal(got, want), name
    assert (got[np.isnan(z)] == -2147483648.0).all()                       # (double)NA_INTEGER, cpp:5374
    if name == "smooth":
        assert got.max() > 20 and got.min() == 1.0


@pytest.mark.parametrize("name,z", [(n, z) for n, z in rasters() if z.shape[0] >= 3 and z.shape[1] >= 3],
                         ids=[n for n, z in rasters() if z.shape[0] >= 3 and z.shape[1] >= 3])
@pytest.mark.parametrize("res", [1.0, (2.0, 5.0), 30.0])
def test_topidx_equals_the_restatement(name, z, res):
    xres, yres = (res, res) if np.isscalar(res) else res
    got = topidx(z, res)
    want = HO.topidx(z, xres, yres)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    na = np.isnan(z)
    assert (got[na].view(np.uint64) == 0x7FF00000000007A2).all()           # masked cells are R's NA_real_
    assert (got[~na] > 0).all()


def test_degenerate_inputs():
    allna = np.full((6, 5), np.nan)
    assert (flowaccCpp(allna) == -2147483648.0).all()                      # the reference underflows `size() - 1` here
    assert np.isnan(topidx(allna, 1.0)).all()
    with pytest.raises(_abi.McfError, match="resolution"):
        topidx(np.zeros((4, 4)), 0.0)
    flat = np.zeros((7, 7))
    t = topidx(flat, 10.0)                                                  # slope floor atan(0.02 / res)
    assert np.isfinite(t).all() and t.min() >= 100.0 / np.tan(np.arctan(0.002)) - 1e-6
