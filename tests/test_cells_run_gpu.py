"""mcf_plan_run_days_cells (include/mcf.h): the solver for a SUBSET of the cells — gathered into dense tiles of their own, solved
into a ring of their own, copied to their places in the slot — against the plain launch: the same bits at the marked cells,
nothing touched elsewhere.  And its user, the snow run's days that are snow days and no-snow days at once
(mcf_snowplan_free_cells): the merged output is the tile-masked run's and the plain run's, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from microclimf_amd import snow as S  # noqa: E402
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

OUT = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")


def _dev(mask):
    return torch.from_numpy(np.ascontiguousarray(mask.reshape(-1, order="F"), dtype=np.uint8)).to("cuda:0")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("case", ["plain", "irregular_cells", "irregular_day", "layers", "ground", "few", "all", "cpb16", "cpb32", "cpb42"])
def test_marked_cells_get_the_plain_launch_s_bits_and_the_others_keep_theirs(case, monkeypatch):
    rows, cols, nd = (57, 31, 4)
    T = 8 * 24
    rq = 0.0 if case == "ground" else 0.05
    a = synthetic.workload(rows, cols, T, reqhgt=rq, start_doy=150, variety=True, hgt_range=(0.03, 3.0))
    rng = np.random.default_rng(11)
    if case == "irregular_cells":           # cells whose constants are not finite take the reference-form clamps: tiles of their own
        veg = dict(a["vegp"])
        clump = np.array(veg["clump"], order="F")
        clump[rng.random((rows, cols)) < 0.08] = np.nan
        veg["clump"] = clump
        a["vegp"] = veg
    if case == "irregular_day":             # a NaN forcing step: the whole launch runs the reference-form instantiation
        cd = dict(a["climdata"])
        t = np.array(cd["temp"])
        t[30] = np.nan
        cd["temp"] = t
        a["climdata"] = cd
    if case == "layers":
        a = synthetic.layered(a, 3)
    need = rng.random((rows, cols)) < (0.002 if case == "few" else 1.1 if case == "all" else 0.07)
    if case == "few":
        need[:] = False
        need[5, 7] = need[40, 30] = True
    nsel = [k for k in OUT if not (rq == 0.0 and k in ("tleaf", "relhum"))]
    a["out"] = [k in nsel for k in OUT]
    cpb = int(case[3:]) if case.startswith("cpb") else 0     # (other tile sizes: the block's places are hour-major)
    with Plan(**a, ring_days=5, ring_slots=2, cells_per_block=cpb) as p:
        assert cpb == 0 or p.ring_layout()["cells_per_tile"] == cpb
        p.run_days_at(0, nd, 0, 1)                      # the plain launch of days 0 .. 3 at day 1 of slot 0
        p.run_days_at(4, nd, 1, 1)                      # slot 1 holds days 4 .. 7 there
        held = {k: p.fetch(1, k, 24, nd * 24).copy() for k in nsel}
        want = {k: p.fetch(0, k, 24, nd * 24).copy() for k in nsel}
        flags = _dev(need)
        n = p.run_days_cells(0, nd, 1, 1, flags.data_ptr())
        assert n == int(need.sum())
        for k in nsel:
            got = p.fetch(1, k, 24, nd * 24)
            assert np.array_equal(_bits(got[need]), _bits(want[k][need])), (case, k)
            assert np.array_equal(_bits(got[~need]), _bits(held[k][~need])), (case, k)
        assert np.isfinite(want["Tz"][need]).any()
        if case == "plain":
            # ... in passes of one day when the budget is small, and again on the same buffers
            p.run_days_at(4, nd, 1, 1)
            monkeypatch.setenv("MCF_CELLS_RING_GB", "0.000001")
            n2 = p.run_days_cells(0, nd, 1, 1, flags.data_ptr())
            assert n2 == n
            for k in nsel:
                got = p.fetch(1, k, 24, nd * 24)
                assert np.array_equal(_bits(got[need]), _bits(want[k][need])), k
                assert np.array_equal(_bits(got[~need]), _bits(held[k][~need])), k
            # no cell marked: nothing happens
            last = {k: p.fetch(1, k, 24, nd * 24).copy() for k in nsel}
            none = _dev(np.zeros((rows, cols), bool))
            assert p.run_days_cells(0, nd, 1, 1, none.data_ptr()) == 0
            for k in nsel:
                assert np.array_equal(_bits(p.fetch(1, k, 24, nd * 24)), _bits(last[k])), k


def test_argument_checks():
    a = synthetic.workload(12, 9, 48, reqhgt=-0.05, start_doy=150)
    flags = _dev(np.ones((12, 9), bool))
    with Plan(**a, ring_days=2, ring_slots=1) as p:
        with pytest.raises(RuntimeError, match="cell subset"):
            p.run_days_cells(0, 1, 0, 0, flags.data_ptr())
    a = synthetic.workload(12, 9, 48, reqhgt=0.05, start_doy=150)
    with Plan(**a, ring_days=2, ring_slots=1) as p:
        with pytest.raises(RuntimeError, match="null"):
            p.run_days_cells(0, 1, 0, 0, 0)
        with pytest.raises(RuntimeError, match="ring slot holds"):
            p.run_days_cells(0, 2, 0, 1, flags.data_ptr())
        with pytest.raises(RuntimeError, match="day range"):
            p.run_days_cells(1, 2, 0, 0, flags.data_ptr())


MAT = 10.0


def _snow_year(rows, cols, ndays, mode, monkeypatch):
    """The one-call snow run over a deep pack with holes (one cell in fifty starts bare, no snowfall): most days are snow days AND
    no-snow days, few cells are the solver's, and a third of the tiles hold one."""
    for k in ("MCF_SNOW_NO_TILE_SKIP", "MCF_SNOW_NO_CELL_GATHER"):
        monkeypatch.delenv(k, raising=False)
    if mode == "tiles":
        monkeypatch.setenv("MCF_SNOW_NO_CELL_GATHER", "1")
    if mode == "dense":
        monkeypatch.setenv("MCF_SNOW_NO_TILE_SKIP", "1")
    T = ndays * 24
    sw = synthetic.snow_workload(rows, cols, T, cold=-2.0, zref=3.5, start_doy=60)
    a = synthetic.workload(rows, cols, T, reqhgt=0.05, zref=3.5, hgt_range=(0.05, 3.0), start_doy=60, variety=True)
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    rng = np.random.default_rng(3)
    deep = np.where(rng.random((rows, cols)) < 0.98, 2.0, 0.0)
    sw["other"] = dict(sw["other"], isnowdc=np.asfortranarray(deep), isnowdg=np.asfortranarray(0.7 * deep))
    sw["climdata"] = dict(sw["climdata"], precip=np.zeros(T))
    snow = dict(sw, dtm=dtm, res=1.0, tfact=0.02)
    micro = {"obstime": sw["obstime"], "climdata": sw["climdata"], "vegp": sw["vegp"], "other": sw["other"]}
    with S.SnowRun(a, snow) as run:
        sd, nsd = run.pass1()
        out = run.pass2(micro, MAT)
        st = run.stats()
    return out, np.asarray(sd), np.asarray(nsd), st


def test_snow_run_gathers_the_cells_the_merge_keeps_and_the_output_is_the_same(monkeypatch):
    rows, cols, ndays = 64, 24, 10
    cells, sd, nsd, st = _snow_year(rows, cols, ndays, "cells", monkeypatch)
    tiles, sd2, nsd2, st2 = _snow_year(rows, cols, ndays, "tiles", monkeypatch)
    dense, sd3, nsd3, st3 = _snow_year(rows, cols, ndays, "dense", monkeypatch)
    assert np.array_equal(sd, sd2) and np.array_equal(sd, sd3) and np.array_equal(nsd, nsd2) and np.array_equal(nsd, nsd3)
    assert (sd & nsd).sum() >= 5
    assert st3["tile_days_left_out"] == 0 and st["tile_days_left_out"] > st2["tile_days_left_out"], (st, st2, st3)
    assert st["tile_days_left_out"] > 0.3 * st["tile_days"], st               # (days without snow anywhere are in the total)
    for k in cells:
        assert np.array_equal(_bits(cells[k]), _bits(tiles[k])), k
        assert np.array_equal(_bits(cells[k]), _bits(dense[k])), k
    assert np.isfinite(cells["Tz"]).any()
