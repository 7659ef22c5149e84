"""runmicro3Cpp / runmicro4Cpp through the C ABI against the oracle."""
import numpy as np
import pytest

from microclimf_amd import McfError, synthetic
from microclimf_amd.api import runmicro3Cpp, runmicro4Cpp
from test_parity_gpu import compare, with_na

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layers,chunk", [(3, 0), (4, 2), (2, 5)])
def test_runmicro3_matches_oracle(oracle, layers, chunk):
    a = with_na(synthetic.workload(19, 7, 24 * 7, reqhgt=0.05, variety=True, start_doy=150))
    a = synthetic.layered(a, layers, cover_days=6)       # last day not covered by any layer
    want = oracle.run_grid(**a)
    got = runmicro3Cpp(a.pop("dfsel"), **a, days_per_chunk=chunk)
    assert np.isnan(got["Tz"][:, :, 144:]).all()
    compare(got, want)


def test_runmicro3_below_ground(oracle):
    a = with_na(synthetic.workload(17, 5, 24 * 6, reqhgt=-0.1, variety=True, start_doy=120,
                                   out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]))
    a = synthetic.layered(a, 3)
    want = oracle.run_grid(**a)
    compare(runmicro3Cpp(a.pop("dfsel"), **a), want)


def test_runmicro4_matches_oracle(oracle):
    a = with_na(synthetic.workload(18, 6, 96, reqhgt=0.05, variety=True, start_doy=170, array_forcing=True))
    a = synthetic.layered(a, 2)
    want = oracle.run_grid(**a, array_forcing=True)
    dfsel = a.pop("dfsel")
    a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
    compare(runmicro4Cpp(dfsel, **a), want)


def test_too_many_layers_is_the_reference_error():
    a = synthetic.layered(synthetic.workload(4, 4, 48, reqhgt=0.05), 2)
    a["dfsel"]["ed"] = np.array([11, 47])           # first layer shorter than a day
    with pytest.raises(McfError, match="Too many layers in vegp"):
        runmicro3Cpp(a.pop("dfsel"), **a)
