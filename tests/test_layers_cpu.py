"""Time-varying vegetation (runmicro3Cpp / runmicro4Cpp) in the oracle."""
import numpy as np

from microclimf_amd import synthetic


def test_identical_layers_equal_static_vegetation(oracle):
    a = synthetic.workload(5, 4, 96, reqhgt=0.05, variety=True, start_doy=170)
    want = oracle.run_grid(**a)
    b = dict(a)
    b["vegp"] = {k: np.stack([v, v, v], axis=2) for k, v in a["vegp"].items()}
    b["dfsel"] = {"lyr": [1, 2, 3], "st": [0, 24, 72], "ed": [23, 71, 95]}
    got = oracle.run_grid(**b)
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k


def test_uncovered_days_stay_na(oracle):
    a = synthetic.layered(synthetic.workload(4, 3, 96, reqhgt=0.05, start_doy=170), 2, cover_days=3)
    r = oracle.run_grid(**a)
    assert np.isnan(r["Tz"][:, :, 72:]).all() and np.isfinite(r["Tz"][1, 1, :72]).all()
    # the layers differ, so the two halves differ from a static run with layer 0
    s = dict(a)
    s.pop("dfsel")
    s["vegp"] = {k: v[:, :, 0] for k, v in a["vegp"].items()}
    r0 = oracle.run_grid(**s)
    assert np.array_equal(r["Tz"][:, :, :24], r0["Tz"][:, :, :24], equal_nan=True)
    assert not np.allclose(r["Tz"][1, 1, 48:72], r0["Tz"][1, 1, 48:72])
