"""Generates tests/golden/snow_*.npz: small seeded snow-branch inputs + the ORACLE's outputs
(oracle/snow_oracle.c).  Regression vectors, NOT reference outputs (the reference cannot be built
here); they freeze the oracle and let the GPU tests check libmcfhip without the oracle at run time.

Run from the repo root:  python tests/golden/make_golden_snow.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from microclimf_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import replay_reference_tests as RT  # noqa: E402

CASES = {
    "snow_vec": (dict(rows=7, cols=5, tsteps=72, cold=3.0, zref=3.5, snowenv="Alpine"), False, 0.05),
    "snow_arr": (dict(rows=5, cols=6, tsteps=48, cold=2.0, zref=3.5, snowenv="Tundra"), True, 1.0),
}

if __name__ == "__main__":
    here = Path(__file__).resolve().parent
    for name, (kw, af, reqhgt) in CASES.items():
        sw = synthetic.snow_workload(array_forcing=af, **kw)
        smod = O.run_snowmodel(**sw, array_forcing=af)
        snowm, micro = synthetic.microsnow_inputs(sw, smod)
        mo = O.run_microsnow(reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0,
                             [1] * 10, array_forcing=af)
        blob = {"arg.snowenv": np.asarray(sw["snowenv"]), "arg.array_forcing": np.asarray(af),
                "arg.reqhgt": np.asarray(reqhgt), "arg.mat": np.asarray(3.0)}
        for grp in ("obstime", "climdata", "pointm", "vegp", "other"):
            for k, v in sw[grp].items():
                blob[f"{grp}.{k}"] = np.asarray(v)
        for k, v in micro.items():
            blob[f"micro.{k}"] = v
        for k, v in smod.items():
            blob[f"smod.{k}"] = v
        for k, v in mo.items():
            blob[f"mout.{k}"] = v
        np.savez_compressed(here / f"{name}.npz", **blob)
        print(name, {k: v.shape for k, v in smod.items()})
    # the oracle's output on the inputs of the reference's own test-pointmodelsnow.R
    checks, info = RT.replay_pointmodelsnow_test()
    assert all(c[1] for c in checks)
    np.savez_compressed(here / "pointmodelsnow_test.npz", **{k: np.asarray(v) for k, v in RT.LAST_POINTSNOW.items()})
    print("pointmodelsnow_test", info)
