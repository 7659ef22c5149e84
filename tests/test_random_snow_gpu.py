"""Seeded random configurations of the snow entries (mcf_gridmodelsnow1/2, mcf_gridmicrosnow1/2) against
oracle/snow_oracle.c: raster shape, series length with ragged tails, season, cold offset, environment, latitude,
NA / bare shares, forcing geometry, sensor height and output mask."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.snow import gridmicrosnow1, gridmicrosnow2, gridmodelsnow1, gridmodelsnow2
from snow_cases import assert_close, model_args

pytestmark = pytest.mark.gpu
TOL = 1e-6
ENVS = ("Alpine", "Maritime", "Prairie", "Taiga", "Tundra", "Ephemeral")


def draw(i):
    rng = np.random.default_rng(9100 + i)
    rows, cols = int(rng.integers(1, 24)), int(rng.integers(1, 24))
    if i % 9 == 0:
        rows = 1
    T = int(rng.integers(1, 6)) * 24 + int(rng.choice([0, 0, 7, 23]))
    if i % 13 == 0:
        T = int(rng.integers(1, 24))                       # less than a day
    af = bool(rng.random() < 0.4)
    kw = dict(rows=rows, cols=cols, tsteps=T, cold=float(rng.choice([-1.0, 2.0, 3.0, 6.0, 11.0])), zref=3.5,
              snowenv=str(rng.choice(ENVS)), start_doy=int(rng.choice([5, 40, 75, 330, 355])),
              lat=float(rng.choice([45.0, 57.0, 69.0, -44.0])), lon=float(rng.choice([-4.0, 20.0, 170.0])),
              na_frac=float(rng.choice([0.0, 0.02, 0.4])), bare_frac=float(rng.choice([0.0, 0.1, 0.6])),
              year=int(rng.choice([2023, 2024])), seed=int(rng.integers(1, 1 << 30)), array_forcing=af)
    reqhgt = float(rng.choice([0.0, 0.02, 0.05, 0.6, 1.0, 2.5, 3.4, -0.05, -1.0, -8.0]))
    out = [int(b) for b in rng.random(10) < 0.6]
    if not any(out):
        out[int(rng.integers(0, 10))] = 1
    return kw, af, reqhgt, out


@pytest.mark.parametrize("i", range(64))
def test_random_snow_configuration(oracle, i):
    kw, af, reqhgt, out = draw(i)
    sw = synthetic.snow_workload(**kw)
    want = oracle.run_snowmodel(**model_args(sw), array_forcing=af)
    got = (gridmodelsnow2 if af else gridmodelsnow1)(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"],
                                                      sw["other"], sw["snowenv"])
    for k in ("Tc", "Tg", "sdepc", "sdepg", "sden", "meltc", "meltg"):
        assert_close(got[k], want[k], TOL, f"{i}:{k}")
    for k in ("agec", "ageg"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    # the snow microclimate on the oracle's snowpack state
    snowm, micro = synthetic.microsnow_inputs(sw, want)
    args = (reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0, out)
    mwant = oracle.run_microsnow(*args, array_forcing=af)
    mgot = (gridmicrosnow2 if af else gridmicrosnow1)(*args)
    assert list(mgot) == list(mwant)
    with np.errstate(invalid="ignore"):
        covered = snowm["totalSWE"] > 0
    for k in mwant:
        assert_close(mgot[k], mwant[k], TOL, f"{i}:{reqhgt}:{k}")
        assert np.array_equal(mgot[k][~covered], micro[k][~covered]), k


@pytest.mark.parametrize("i", range(10))
def test_random_snow_driver_cases(oracle, i):
    """`.snowmodel1`'s chunk loop on the device against its oracle for random rasters, resolutions and chunk lengths
    (at most three chunks: over long series the reference's hand-over amplifies rounding residue)"""
    from microclimf_amd.snow import applycpp3, snowmodel1_chunks
    from oracle import snowdriver_oracle as SD
    rng = np.random.default_rng(9900 + i)
    rows, cols = int(rng.integers(12, 60)), int(rng.integers(12, 60))
    chunk = int(rng.choice([24, 48, 120]))
    T = chunk * int(rng.integers(1, 4)) + int(rng.choice([0, 0, 10]))
    sw = synthetic.snow_workload(rows, cols, T, cold=float(rng.choice([2.0, 3.0, 6.0])), zref=3.5,
                                 snowenv=str(rng.choice(ENVS[:5])), start_doy=int(rng.choice([5, 40, 340])),
                                 seed=int(rng.integers(1, 1 << 30)))
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm * float(rng.choice([0.5, 1.0, 3.0])))
    args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm,
            float(rng.choice([1.0, 2.0, 10.0])), float(rng.choice([0.02, 0.05])))
    want = SD.snowmodel1_chunks(*args, chunk_steps=chunk)
    got = snowmodel1_chunks(*args, chunk_steps=chunk)
    for k in want:
        assert_close(got[k], want[k], TOL, f"{i}:{k}")
    # applycpp3 over the result, as .runmicrosnow uses it (R/internal.R:3592-3593)
    for fun in ("max", "min", "mean", "sum"):
        swe = np.nan_to_num(got["totalSWE"], nan=0.0)
        ref = getattr(np, fun)(swe.reshape(-1, swe.shape[2]), axis=0)
        np.testing.assert_allclose(applycpp3(np.asfortranarray(swe), fun), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("af", [False, True])
def test_medium_snow_raster_against_the_oracle(oracle, af):
    """tens of workgroups per kernel instead of the two or three of the small cases"""
    sw = synthetic.snow_workload(150, 110, 96, cold=3.0, zref=3.5, array_forcing=af, na_frac=0.03)
    want = oracle.run_snowmodel(**model_args(sw), array_forcing=af)
    got = (gridmodelsnow2 if af else gridmodelsnow1)(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"],
                                                      sw["other"], sw["snowenv"])
    for k in ("Tc", "Tg", "sdepc", "sdepg", "sden", "meltc", "meltg", "agec", "ageg"):
        assert_close(got[k], want[k], TOL, k)
    snowm, micro = synthetic.microsnow_inputs(sw, want)
    for reqhgt in (0.05, 2.5):
        args = (reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0, [1] * 10)
        mwant = oracle.run_microsnow(*args, array_forcing=af)
        mgot = (gridmicrosnow2 if af else gridmicrosnow1)(*args)
        for k in mwant:
            assert_close(mgot[k], mwant[k], TOL, f"{reqhgt}:{k}")
