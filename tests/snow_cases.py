"""Named workloads of the snow branch, shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np

from microclimf_amd import synthetic

# name -> (kwargs of synthetic.snow_workload, array_forcing)
SNOW_CASES = {
    "alpine_5day": (dict(rows=14, cols=9, tsteps=120, cold=3.0, zref=3.5), False),
    "taiga_cold": (dict(rows=9, cols=7, tsteps=96, cold=9.0, zref=3.5, snowenv="Taiga"), False),
    "tundra_thaw": (dict(rows=8, cols=8, tsteps=144, cold=-1.0, zref=3.5, snowenv="Tundra", start_doy=60), False),
    "maritime_partial_day": (dict(rows=7, cols=5, tsteps=61, cold=3.0, zref=3.5, snowenv="Maritime"), False),
    "prairie_short": (dict(rows=6, cols=6, tsteps=17, cold=4.0, zref=3.5, snowenv="Prairie"), False),
    "veg_above_zref": (dict(rows=8, cols=6, tsteps=48, cold=3.0, zref=2.0), False),      # log of a negative: NaN parity
    "unknown_env": (dict(rows=5, cols=5, tsteps=48, cold=3.0, zref=3.5, snowenv="Ephemeral"), False),
    "array_5day": (dict(rows=10, cols=8, tsteps=120, cold=3.0, zref=3.5), True),
    "array_partial_day": (dict(rows=6, cols=7, tsteps=53, cold=2.0, zref=3.5, snowenv="Tundra"), True),
    "array_thaw": (dict(rows=7, cols=6, tsteps=96, cold=-1.0, zref=3.5, start_doy=75), True),
}


def build_snow(name):
    kw, af = SNOW_CASES[name]
    return synthetic.snow_workload(array_forcing=af, **kw), af


def assert_close(got, want, tol, what=""):
    """NaN pattern identical, finite values within tol * (1 + |x|)."""
    assert got.shape == want.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), f"{what}: inf pattern differs"
    if fin.any():
        err = np.abs(got[fin] - want[fin]) / (1.0 + np.abs(want[fin]))
        assert err.max() <= tol, f"{what}: max scaled error {err.max():.3e}"
        return float(err.max())
    return 0.0
