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
