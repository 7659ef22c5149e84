"""Host entries of the fast snow method (`.snowmodelq1`, R/internal.R:2627-2776): mcf_canintfrac and mcf_meltmu against
the oracle's restatement of src/microclimfCpp.cpp:5417-5492, and the R `a:b` helper of the day loop."""
import numpy as np
import pytest

from microclimf_amd import snow as S
from oracle import snowfast_oracle as SF


@pytest.fixture(scope="module", autouse=True)
def _lib():
    import __graft_entry__ as g
    g.build_library()


def test_canintfrac_matches_the_oracle():
    rng = np.random.default_rng(5)
    hgt = rng.uniform(0.0, 25.0, (7, 9))
    pai = rng.uniform(0.0, 6.0, (7, 9))
    hgt[2, 3] = np.nan
    hgt[0, 0], pai[0, 1] = 0.0, 0.0                      # the 0.001 floors
    for prec, tc in ((1.7, -3.0), (0.2, 1.5), (12.0, -20.0)):
        got = S.canintfrac(hgt, pai, 2.0, prec, tc, 0.0)
        want = SF.canintfrac(hgt, pai, 2.0, prec, tc, 0.0)
        assert np.isnan(got[2, 3]) and np.isnan(want[2, 3])
        np.testing.assert_allclose(got, want, rtol=1e-12, equal_nan=True)
        assert np.nanmax(got) <= 1.0 and np.nanmin(got) > 0.0
    for prec in (0.0, float("nan")):                     # no snowfall in the series: `mean(snow[snow > 0])` is NaN in R
        got = S.canintfrac(hgt, pai, 2.0, prec, -3.0, 0.0)
        assert np.isnan(got[2, 3]) and np.all(got[~np.isnan(hgt)] == 0.5)


def test_meltmu_matches_the_oracle():
    rng = np.random.default_rng(6)
    sv = rng.uniform(0.3, 1.0, (6, 5))
    sv[1, 1] = np.nan
    stemp = rng.normal(0.5, 3.0, 200)
    tc = stemp - rng.uniform(0.0, 4.0, 200)
    got, want = S.meltmu(sv, stemp, tc), SF.meltmu(sv, stemp, tc)
    np.testing.assert_allclose(got, want, rtol=1e-12, equal_nan=True)
    assert np.isnan(got[1, 1]) and np.nanmin(got) >= 0.0
    # open sky = the point model itself; a frozen surface gives 1 everywhere, the NA cell included (cpp:5484-5490)
    np.testing.assert_allclose(S.meltmu(np.ones((2, 2)), stemp, tc), 1.0, rtol=1e-12)
    assert np.all(S.meltmu(sv, -np.abs(stemp), tc) == 1.0)
    assert np.all(S.meltmu(sv, stemp[:0], tc[:0]) == 1.0)


def test_meltmu2_matches_the_oracle():
    rng = np.random.default_rng(8)
    mu = rng.uniform(0.3, 1.0, (5, 4))
    mu[2, 2] = np.nan
    stemp = rng.normal(0.3, 3.0, (5, 4, 60))
    tc = stemp - rng.uniform(0.0, 4.0, (5, 4, 60))
    stemp[0, 0, :] = -1.0                                # never thaws: 0.5 (cpp:5518-5520)
    stemp[4, 3, 5] = np.nan                              # a masked step counts for nothing
    got, want = S.meltmu2(mu, stemp, tc), SF.meltmu2(mu, stemp, tc)
    np.testing.assert_allclose(got, want, rtol=1e-12, equal_nan=True)
    assert np.isnan(got[2, 2]) and got[0, 0] == 0.5
    np.testing.assert_allclose(S.meltmu2(mu, stemp[:, :, :0], tc[:, :, :0])[~np.isnan(mu)], 0.5)


def test_r_colon_counts_down_like_R():
    assert list(S.r_colon(3, 6)) == [2, 3, 4, 5]
    assert list(S.r_colon(25, 24)) == [24, 23]          # two consecutive selected days: `(ped + 1):(subs[st] - 1)`
    assert list(S.r_colon(4, 4)) == [3]
