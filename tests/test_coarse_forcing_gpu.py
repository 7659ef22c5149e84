"""Coarse array forcing (include/mcf.h, array_forcing == 2): `.runmodel2Cpp`'s resampling of coarse climate and
point-model arrays fused into the solver, against the reference's order of work — expand to full resolution
(oracle/coarse_oracle.py), then the array-forcing oracle."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import Plan, runmicro2Cpp, runmicro2Cpp_coarse
from oracle import coarse_oracle as CO
from test_parity_gpu import compare

pytestmark = pytest.mark.gpu
ARGS = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact",
        "complete", "mat", "out")


def expanded(a, rp, cp):
    clim, pm = CO.expand(a["climdata"], a["pointm"], rp, cp)
    b = dict(a)
    b.update(climdata=clim, pointm=pm)
    return b


@pytest.mark.parametrize("rows,cols,cr,cc,reqhgt", [(37, 29, 5, 4, 0.05), (64, 40, 2, 3, 1.0), (20, 33, 1, 1, 0.05),
                                                     (31, 18, 31, 18, 0.0), (26, 26, 4, 4, -0.1)])
def test_coarse_forcing_matches_expand_then_solve(oracle, rows, cols, cr, cc, reqhgt):
    a, rp, cp = synthetic.coarse_workload(rows, cols, 72, cr, cc, reqhgt=reqhgt, variety=True, start_doy=170, na_frac=0.03)
    got = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp)
    b = expanded(a, rp, cp)
    want = oracle.run_grid(**b, array_forcing=True)
    compare(got, want)
    # and the device's own array-forcing path given the expanded arrays
    full = runmicro2Cpp(*[b[k] for k in ARGS])
    for k in want:
        np.testing.assert_allclose(got[k], full[k], rtol=1e-9, atol=1e-9, err_msg=k)


def test_identity_grid_equals_plain_array_forcing_bitwise_inputs(oracle):
    """a coarse grid as fine as the raster interpolates nothing: positions are integers, weights 0"""
    a, rp, cp = synthetic.coarse_workload(24, 16, 48, 24, 16, variety=True, start_doy=100)
    assert np.array_equal(rp, np.arange(24.0)) and np.array_equal(cp, np.arange(16.0))
    b = expanded(a, rp, cp)
    assert np.array_equal(b["climdata"]["tc"], a["climdata"]["temp"])
    compare(runmicro2Cpp_coarse(*[a[k] for k in ARGS]), oracle.run_grid(**b, array_forcing=True))


def test_plan_keeps_the_coarse_series_resident_and_needs_no_uploads(oracle):
    a, rp, cp = synthetic.coarse_workload(40, 30, 120, 3, 3, variety=True, start_doy=200)
    want = oracle.run_grid(**expanded(a, rp, cp), array_forcing=True)
    with Plan(**a, ring_days=2, ring_slots=2, coarse={"rowpos": rp, "colpos": cp}) as p:
        slot = 0
        for d0 in (0, 2, 4):
            nd = min(2, 5 - d0)
            p.run_days(d0, nd, slot)                 # no upload_forcing_days
            p.sync()
            for k in ("Tz", "relhum", "Rswup"):
                got = p.fetch(slot, k, 0, nd * 24)
                w = want[k][:, :, d0 * 24:(d0 + nd) * 24]
                assert np.array_equal(np.isnan(got), np.isnan(w))
                assert np.nanmax(np.abs(got - w) / (1 + np.abs(w))) < 1e-6, k
            slot ^= 1


def test_coarse_mode_argument_checks():
    from microclimf_amd import McfError
    a, rp, cp = synthetic.coarse_workload(12, 12, 24, 3, 3)
    with pytest.raises(McfError, match="coarse_rowpos"):
        runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp + 5.0, colpos=cp)
    a2, rp2, cp2 = synthetic.coarse_workload(12, 12, 48, 3, 3, reqhgt=-0.1, complete=False)
    a2["pointm"] = dict(a2["pointm"], Tg=a2["pointm"]["soilm"], Tbp=a2["pointm"]["soilm"])
    with pytest.raises(McfError, match="complete"):
        runmicro2Cpp_coarse(*[a2[k] for k in ARGS], rowpos=rp2, colpos=cp2)


@pytest.mark.parametrize("i", range(24))
def test_random_coarse_configurations(oracle, i):
    rng = np.random.default_rng(12000 + i)
    rows, cols = int(rng.integers(1, 45)), int(rng.integers(1, 45))
    cr, cc = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    T = int(rng.integers(1, 4)) * 24 + int(rng.choice([0, 0, 7]))
    reqhgt = float(rng.choice([0.02, 0.05, 0.6, 1.9, 0.0, -0.08]))
    out = [int(b) for b in rng.random(10) < 0.6]
    out[0] = 1
    a, rp, cp = synthetic.coarse_workload(rows, cols, T, cr, cc, reqhgt=reqhgt, variety=bool(rng.random() < 0.7),
                                          start_doy=int(rng.integers(1, 350)), cold=float(rng.choice([0.0, 12.0])),
                                          na_frac=float(rng.choice([0.0, 0.05])), out=out, seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.3:                       # a climate grid that covers more than the raster: positions inside one cell
        rp = np.clip(rp * 0.3 + 0.2 * (cr - 1), 0, cr - 1)
        cp = np.clip(cp * 0.5, 0, cc - 1)
    got = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp, days_per_chunk=int(rng.choice([0, 1])) if reqhgt >= 0 else 0)
    compare(got, oracle.run_grid(**expanded(a, rp, cp), array_forcing=True))


@pytest.mark.parametrize("altcorrect", [1, 2])
@pytest.mark.parametrize("rows,cols,cr,cc", [(33, 27, 3, 4), (40, 40, 1, 2)])
def test_altitude_correction_is_fused_too(oracle, altcorrect, rows, cols, cr, cc):
    """`.runmodel2Cpp`'s altcorrect (R/internal.R:1233-1251): pressure through sea level, temperature by a fixed or a
    humidity-dependent lapse rate times the elevation difference to the (interpolated) climate cell; es / ea / tdew stay
    those of the uncorrected temperature"""
    a, rp, cp = synthetic.coarse_workload(rows, cols, 72, cr, cc, reqhgt=0.05, variety=True, start_doy=150, na_frac=0.02)
    _, _, z = synthetic.rasters(rows, cols)
    z = 300.0 + 8.0 * (z - np.nanmean(z))                                  # a few hundred metres of relief
    rng = np.random.default_rng(3)
    zc = 350.0 + 200.0 * rng.random((cr, cc))
    zc[0, 0] = np.nan                                                       # dtmc[is.na(dtmc)] <- 0
    got = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp, altcorrect=altcorrect, dtmc=zc, dtm=z)
    clim, pm = CO.expand(a["climdata"], a["pointm"], rp, cp, altcorrect, zc, z)
    b = dict(a)
    b.update(climdata=clim, pointm=pm)
    compare(got, oracle.run_grid(**b, array_forcing=True))
    plain = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp)
    assert np.nanmax(np.abs(got["Tz"] - plain["Tz"])) > 0.3                  # the correction does something


@pytest.mark.parametrize("kind", ["flipped", "zigzag"])
def test_row_positions_that_are_not_monotone_take_the_per_lane_taps(oracle, kind, monkeypatch):
    """the LDS-staged taps assume row positions that do not decrease down a column (a tile's wrap into the next
    column is found where the position falls back).  A north/south-flipped coarse grid, or any other order, is valid input —
    mcf_plan_create must route it to the per-lane taps: results equal expand-then-solve, and equal the run with the staged
    taps switched off."""
    a, rp, cp = synthetic.coarse_workload(64, 24, 48, 4, 3, reqhgt=0.05, variety=True, start_doy=170, na_frac=0.02)
    rp = rp[::-1].copy() if kind == "flipped" else np.where(np.arange(64) % 2 == 0, rp, rp[::-1])
    got = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp)
    compare(got, oracle.run_grid(**expanded(a, rp, cp), array_forcing=True))
    monkeypatch.setenv("MCF_NO_COARSE_LDS", "1")
    ref = runmicro2Cpp_coarse(*[a[k] for k in ARGS], rowpos=rp, colpos=cp)
    for k in got:
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
