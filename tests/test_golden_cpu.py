"""The oracle against the committed golden vectors (tests/golden/, oracle-generated:
a regression freeze, see tests/golden/make_golden.py)."""
import numpy as np
import pytest

from golden_util import CASES, load


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    a, af, expect = load(name)
    got = oracle.run_grid(**a, array_forcing=af)
    assert list(got) == list(expect)
    for k, w in expect.items():
        assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
        np.testing.assert_allclose(got[k], w, rtol=1e-12, atol=1e-12, err_msg=k)


from golden_util import GOLDEN, SNOW_CASES, load_snow  # noqa: E402
from microclimf_amd import synthetic  # noqa: E402


@pytest.mark.parametrize("name", SNOW_CASES)
def test_snow_oracle_reproduces_golden(oracle, name):
    sw, af, reqhgt, mat, micro, smod, mout = load_snow(name)
    got = oracle.run_snowmodel(**sw, array_forcing=af)
    for k, w in smod.items():
        assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
        np.testing.assert_allclose(got[k], w, rtol=1e-12, atol=1e-12, err_msg=k)
    snowm, _ = synthetic.microsnow_inputs(sw, smod)
    gm = oracle.run_microsnow(reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], mat,
                              [1] * 10, array_forcing=af)
    for k, w in mout.items():
        assert np.array_equal(np.isnan(gm[k]), np.isnan(w)), k
        np.testing.assert_allclose(gm[k], w, rtol=1e-12, atol=1e-12, err_msg=k)


def test_pointmodelsnow_replay_reproduces_golden():
    """the oracle's pointmodelsnow on the inputs of the reference's test-pointmodelsnow.R, frozen"""
    from oracle import replay_reference_tests as RT
    RT.replay_pointmodelsnow_test()
    z = np.load(GOLDEN / "pointmodelsnow_test.npz")
    for k in ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "G", "RswabsG", "RlwabsG", "tr", "umu"):
        np.testing.assert_allclose(RT.LAST_POINTSNOW[k], z[k], rtol=1e-12, atol=1e-12, err_msg=k)
