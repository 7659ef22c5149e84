"""Fused bioclim sink (mcf_runbioclim1/2) against the oracle's runbioclimCpp restatement."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import runbioclim1Cpp, runbioclim2Cpp

pytestmark = pytest.mark.gpu
T = 336 + 4 * 72          # 12 monthly days, hottest, coldest, four quarters of three days


def quarters():
    base = 336
    return [np.arange(base + 72 * i, base + 72 * (i + 1)) for i in range(4)]


def args(array_forcing, reqhgt=0.05):
    a = synthetic.workload(11, 7, T, reqhgt=reqhgt, variety=True, start_doy=120, array_forcing=array_forcing)
    a["vegp"]["hgt"][0, 0] = np.nan
    for k in ("complete", "out"):
        a.pop(k)
    return a


@pytest.mark.parametrize("air", [True, False])
def test_runbioclim1(oracle, air):
    a = args(False)
    wq, dq, hq, cq = quarters()
    out = [1] * 19
    want = oracle.run_bioclim(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=air)
    got = runbioclim1Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=air)
    assert list(got) == [f"bio{i}" for i in range(1, 20)]
    for k, w in want.items():
        assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
        np.testing.assert_allclose(got[k], w, rtol=1e-9, atol=1e-9, err_msg=k)
    assert np.isnan(got["bio1"][0, 0])


def test_runbioclim2_and_selection(oracle):
    a = args(True)
    wq, dq, hq, cq = quarters()
    out = [0] * 19
    for i in (0, 2, 6, 14, 18):
        out[i] = 1
    want = oracle.run_bioclim(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True, array_forcing=True)
    a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
    got = runbioclim2Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True)
    assert list(got) == ["bio1", "bio3", "bio7", "bio15", "bio19"]
    for k, w in want.items():
        np.testing.assert_allclose(got[k], w, rtol=1e-9, atol=1e-9, err_msg=k)


def test_runbioclim_rejects_short_series():
    from microclimf_amd import McfError
    a = synthetic.workload(4, 4, 240, reqhgt=0.05)
    for k in ("complete", "out"):
        a.pop(k)
    with pytest.raises(McfError, match="336"):
        runbioclim1Cpp(**a, out=[1] * 19, wetq=[0], dryq=[0], hotq=[0], colq=[0], air=True)


@pytest.mark.parametrize("af", [False, True])
def test_runbioclim3_and_4_layered_vegetation(oracle, af):
    """runbioclim3Cpp / 4Cpp: fourteen one-day vegetation layers (cpp:3634-3646); the quarter days lie past the last
    layer, so the quarter statistics see NA exactly as in the reference"""
    from microclimf_amd.api import BIOCLIM_DFSEL, runbioclim3Cpp, runbioclim4Cpp
    a = synthetic.workload(9, 6, T, reqhgt=0.05, variety=True, start_doy=100, array_forcing=af)
    a["vegp"]["hgt"][2, 1] = np.nan
    a = synthetic.layered(a, 14)
    a.pop("dfsel")
    for k in ("complete", "out"):
        a.pop(k)
    wq, dq, hq, cq = quarters()
    out = [1] * 19
    want = oracle.run_bioclim(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True, array_forcing=af, dfsel=BIOCLIM_DFSEL)
    if af:
        a["lats"], a["lons"] = a.pop("lat"), a.pop("lon")
        got = runbioclim4Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True)
    else:
        got = runbioclim3Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True)
    for k, w in want.items():
        assert np.array_equal(np.isnan(got[k]), np.isnan(w)), k
        np.testing.assert_allclose(got[k], w, rtol=1e-9, atol=1e-9, err_msg=k)
    assert np.isfinite(got["bio1"][0, 0]) and np.isnan(got["bio1"][2, 1])


def test_bioclim_row_blocks_on_one_device_give_the_same_bits():
    """mcf_runbioclim1_multi ... 4_multi: the fused sink over row blocks from one process (whole-raster twi mean installed in
    every block, the nineteen matrices written in place through the row pitch) — bit for bit the single-device matrices"""
    from microclimf_amd.api import runbioclim3Cpp
    wq, dq, hq, cq = quarters()
    out = [1] * 19
    a = synthetic.workload(37, 9, T, reqhgt=0.05, variety=True, start_doy=120, na_frac=0.05)
    for k in ("complete", "out"):
        a.pop(k)
    whole = runbioclim1Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True)
    parts = runbioclim1Cpp(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True, devices=[0, 0], n_blocks=3)
    assert list(whole) == list(parts)
    for k in whole:
        assert np.array_equal(whole[k].view(np.uint64), parts[k].view(np.uint64)), k
    b = synthetic.workload(23, 8, T, reqhgt=0.05, variety=True, start_doy=100, array_forcing=True)
    for k in ("complete", "out"):
        b.pop(k)
    b["lats"], b["lons"] = b.pop("lat"), b.pop("lon")
    sel = [1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1]
    w2 = runbioclim2Cpp(**b, out=sel, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=False)
    p2 = runbioclim2Cpp(**b, out=sel, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=False, devices=[0], n_blocks=4)
    for k in w2:
        assert np.array_equal(w2[k].view(np.uint64), p2[k].view(np.uint64)), k
    c = synthetic.layered(synthetic.workload(21, 6, T, reqhgt=0.05, variety=True, start_doy=100), 14)
    for k in ("complete", "out", "dfsel"):
        c.pop(k)
    w3 = runbioclim3Cpp(**c, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True)
    p3 = runbioclim3Cpp(**c, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=True, devices=[0], n_blocks=2)
    for k in w3:
        assert np.array_equal(w3[k].view(np.uint64), p3[k].view(np.uint64)), k


@pytest.mark.parametrize("layered", [False, True])
def test_streamed_sink_is_bit_for_bit_the_whole_series_sink(monkeypatch, layered):
    """Round 5: mcf_runbioclim1 / 3 fold the solver's day chunks into per-cell running state (k_bioclim_acc / k_bioclim_fin)
    instead of keeping cells x steps of output; same accumulation order on the same operands.  One-day chunks, chunks that
    cut the quarters, the whole series in one chunk: all equal MCF_BIOCLIM_WHOLE=1 (k_bioclim on the whole series) bit for
    bit — including the variance's second pass, which makes the soil moisture series again from the cell's constants."""
    from microclimf_amd.api import runbioclim3Cpp
    wq, dq, hq, cq = quarters()
    out = [1] * 19
    a = synthetic.workload(29, 13, T, reqhgt=0.05, variety=True, start_doy=150, na_frac=0.06)
    if layered:
        a = synthetic.layered(a, 14)
        a.pop("dfsel")
    for k in ("complete", "out"):
        a.pop(k)
    fn = runbioclim3Cpp if layered else runbioclim1Cpp
    run = lambda air: fn(**a, out=out, wetq=wq, dryq=dq, hotq=hq, colq=cq, air=air)      # noqa: E731
    monkeypatch.setenv("MCF_BIOCLIM_WHOLE", "1")
    whole = {air: run(air) for air in (True, False)}
    monkeypatch.delenv("MCF_BIOCLIM_WHOLE")
    for gb in ("1e-9", "3.6e-5", "8"):                      # one day per chunk / a few (29 x 13 cells: 9.2 KB a day) / everything
        monkeypatch.setenv("MCF_BIOCLIM_RING_GB", gb)
        for air in (True, False):
            got = run(air)
            for k in whole[air]:
                assert np.array_equal(got[k], whole[air][k], equal_nan=True), (gb, air, k)
                fin = np.isfinite(got[k])
                assert np.array_equal(got[k][fin].view(np.uint64), whole[air][k][fin].view(np.uint64)), (gb, air, k)
    # a NaN forcing step: its launch runs the reference's clamp forms (k_solve F = false), so WHICH days do depends on the
    # chunking and the last bits with it — the sink itself must still agree, NaN pattern and all
    a["climdata"]["temp"][40] = np.nan
    monkeypatch.setenv("MCF_BIOCLIM_RING_GB", "1e-9")
    got = run(True)
    monkeypatch.setenv("MCF_BIOCLIM_WHOLE", "1")
    ref = run(True)
    monkeypatch.delenv("MCF_BIOCLIM_WHOLE")
    for k in ref:
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    a["climdata"]["temp"][40] = 10.0
    # quarter lists that are not ascending take the whole-series form (accumulation order = list order)
    monkeypatch.delenv("MCF_BIOCLIM_RING_GB")
    rev = fn(**a, out=out, wetq=wq[::-1].copy(), dryq=dq, hotq=hq, colq=cq, air=True)
    assert np.allclose(rev["bio8"], whole[True]["bio8"], rtol=1e-12, atol=1e-12, equal_nan=True)
