"""The solver's two clamp variants (include/mcf.h mcf_dispatch_stats; mcf_device.hpp `cap` / `flr`): tiles of REGULAR cells
run one v_min_f64 / v_max_f64 per clamp, everything else the reference's compare-and-select form, and a fast wave that meets
a NaN at a watched clamp has its tile redone by the fix-up kernel.  Whatever path runs, the result is the oracle's — value
for value and NaN for NaN; these tests pin WHICH path ran on inputs built to need each of them."""
import numpy as np
import pytest

from microclimf_amd import synthetic
from microclimf_amd.api import Plan

pytestmark = pytest.mark.gpu
NAMES = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")


def _solve(a, days, cpb=21):
    with Plan(**a, ring_days=days, cells_per_block=cpb) as p:
        p.run_days(0, days)
        p.sync()
        got = {k: p.fetch(0, k, 0, days * 24) for k in NAMES}
        st = p.dispatch_stats()
    return got, st


def _same(got, want):
    worst = 0.0
    for k, w in want.items():
        g = got[k]
        assert np.array_equal(np.isnan(g), np.isnan(w)), f"NaN pattern of {k}"
        fin = np.isfinite(w)
        if fin.any():
            worst = max(worst, float((np.abs(g[fin] - w[fin]) / (1 + np.abs(w[fin]))).max()))
        inf = np.isinf(w)
        assert np.array_equal(g[inf], w[inf]), k
    assert worst < 1e-6, worst
    return worst


def test_a_regular_raster_runs_the_fast_clamps(oracle):
    a = synthetic.workload(30, 21, 72, reqhgt=0.05, start_doy=170, variety=True)     # bare and NA cells are regular too
    got, st = _solve(a, 3)
    ntiles = -(-30 * 21 // 21)
    assert st["fast_tiles"] == ntiles and st["slow_tiles"] == 0, st
    assert st["fast_launches"] == 1 and st["slow_launches"] == 0 and st["canary_trips"] == 0, st
    _same(got, oracle.run_grid(**a))


def test_cells_with_non_finite_constants_go_to_the_reference_form(oracle):
    a = synthetic.workload(42, 10, 48, reqhgt=0.05, start_doy=170)
    a["soilc"]["twi"][3, 0] = np.nan          # tile 0: NA twi in a valid cell
    a["vegp"]["leafr"][5, 2] = np.nan         # tile (5 + 42*2) // 21 = 4
    a["vegp"]["x"][7, 4] = np.inf             # tile (7 + 42*4) // 21 = 8
    a["soilc"]["hor"][9, 6, 3] = np.nan       # tile (9 + 42*6) // 21 = 12
    got, st = _solve(a, 2)
    assert st["slow_tiles"] == 4 and st["fast_tiles"] == 20 - 4, st
    assert st["fast_launches"] == 1 and st["slow_launches"] == 1, st
    _same(got, oracle.run_grid(**a))


def test_a_step_with_non_finite_forcing_sends_its_launch_to_the_reference_form(oracle):
    a = synthetic.workload(21, 6, 96, reqhgt=0.05, start_doy=170)
    a["climdata"]["lwdown"] = a["climdata"]["lwdown"].copy()
    a["climdata"]["lwdown"][30] = np.nan      # day 1
    with Plan(**a, ring_days=1, ring_slots=1) as p:
        got = {k: [] for k in NAMES}
        for d in range(4):
            p.run_days(d, 1, 0)
            p.sync()
            for k in NAMES:
                got[k].append(p.fetch(0, k, 0, 24))
        st = p.dispatch_stats()
    got = {k: np.concatenate(v, axis=2) for k, v in got.items()}
    assert st["irregular_days"] == 1 and st["slow_launches"] == 1 and st["fast_launches"] == 3, st
    _same(got, oracle.run_grid(**a))


def test_a_nan_born_inside_a_regular_cell_trips_the_canary_and_is_redone(oracle):
    """Soil volume fractions that make the conductivity negative: every cell constant is finite (the cell is REGULAR), but
    the damping depth is the square root of a negative number, so the ground heat flux meets its clamp as a NaN.  The
    reference's `if (G > 0.6*Rmx)` leaves it NaN; v_min_f64 would not — the watched clamp trips and the tile is redone."""
    a = synthetic.workload(21, 4, 48, reqhgt=0.05, start_doy=170)
    a["soilc"]["Vq"][4, 1] = 1.6              # 1 - 0.74*Vq - 0.49*Vm < 0: c1 < 0 (src/microclimfCpp.cpp:628-636)
    got, st = _solve(a, 2)
    want = oracle.run_grid(**a)
    assert np.isnan(want["Tz"][4, 1]).all() and np.isfinite(want["Tz"][5, 1]).all()
    assert st["slow_tiles"] == 0 and st["canary_trips"] >= 1, st
    _same(got, want)


@pytest.mark.parametrize("cpb", [16, 21, 32, 42])
def test_every_tile_geometry_has_both_variants(oracle, cpb):
    a = synthetic.workload(23, 9, 48, reqhgt=1.0, start_doy=200, variety=True, hgt_range=(0.1, 1.9))   # above and below canopy
    a["soilc"]["twi"][0, 0] = np.nan
    got, st = _solve(a, 2, cpb)
    assert st["slow_tiles"] == 1 and st["fast_tiles"] == -(-23 * 9 // cpb) - 1, st
    _same(got, oracle.run_grid(**a))


def test_soil_state_shared_per_day_or_per_lane_gives_the_same_bits(oracle):
    """Days on which pointm$soilm has one value (what soilmCpp's daily bucket model gives) compute the soil-only state once
    per tile and day; a day on which it varies within the day must go through the hour lanes.  One series with such a day
    in the middle, solved as ONE launch (no day shared) and day by day (days 0 and 2 shared): bit for bit the same, and
    the oracle's values."""
    a = synthetic.workload(25, 7, 72, reqhgt=0.05, start_doy=170, variety=True)
    sm = a["pointm"]["soilm"].copy()
    assert (sm[:24] == sm[0]).all()                       # the synthetic series is daily
    sm[24:48] = sm[24] + 0.01 * np.sin(np.arange(24))     # day 1 varies hour by hour
    a["pointm"]["soilm"] = sm
    whole, _ = _solve(a, 3)
    with Plan(**a, ring_days=1) as p:
        parts = {k: [] for k in NAMES}
        for d in range(3):
            p.run_days(d, 1, 0)
            p.sync()
            for k in NAMES:
                parts[k].append(p.fetch(0, k, 0, 24))
    for k in NAMES:
        by_day = np.concatenate(parts[k], axis=2)
        assert np.array_equal(by_day.view(np.uint64), whole[k].view(np.uint64)), k
    _same(whole, oracle.run_grid(**a))


# ---- array forcing: no per-step table the host could classify; every lane checks its own forcing values ---------------
def _solve_af(a, days, **kw):
    with Plan(**a, array_forcing=kw.pop("mode", True), ring_days=days, ring_slots=1, **kw) as p:
        if not kw.get("coarse_rows"):
            p.upload_forcing_days(0, days, 0)
        p.run_days(0, days, 0)
        p.sync()
        got = {k: p.fetch(0, k, 0, days * 24) for k in NAMES}
        st = p.dispatch_stats()
    return got, st


def test_array_forcing_runs_the_fast_clamps_too(oracle):
    a = synthetic.workload(33, 8, 72, reqhgt=0.05, start_doy=170, variety=True, array_forcing=True)
    got, st = _solve_af(a, 3)
    assert st["fast_tiles"] == -(-33 * 8 // 32) and st["slow_tiles"] == 0, st
    assert st["fast_launches"] == 1 and st["slow_launches"] == 0 and st["canary_trips"] == 0, st
    _same(got, oracle.run_grid(**a, array_forcing=True))


@pytest.mark.parametrize("series,value", [("lwdown", np.nan), ("tc", np.inf), ("pk", -101.3), ("Gp", np.nan), ("dtrp", 0.0),
                                          ("kp", np.inf)])
def test_a_lane_with_bad_array_forcing_has_its_tile_redone(oracle, series, value):
    """One value of one cell's series is not finite, or has the sign that breaks what the min / max clamps rely on (a negative
    pressure; a zero or infinite divisor of the ground-heat-flux factor): the lane's check trips the canary, the fix-up
    kernel redoes the tile with the reference's clamps, and the result is the oracle's — NaN for NaN."""
    a = synthetic.workload(32, 5, 48, reqhgt=0.05, start_doy=170, array_forcing=True)
    grp = "pointm" if series in a["pointm"] else "climdata"
    a[grp][series] = a[grp][series].copy(order="F")
    a[grp][series][7, 2, 30] = value            # tile 2, day 1, hour 6
    got, st = _solve_af(a, 2)
    assert st["slow_tiles"] == 0 and st["fast_launches"] == 1 and st["canary_trips"] >= 1, st
    _same(got, oracle.run_grid(**a, array_forcing=True))


def test_coinciding_lagrangian_resistances_are_redone_with_the_reference_form(oracle):
    """A wind speed so large (finite, so the step is REGULAR) that both canopy resistances fall under their 0.001 floor
    (rhcanopy, src/microclimfCpp.cpp:1378-1379): Rc - Rz = 0 and the reference's far field is inf * 0 = NaN (cpp:1388-1395).
    The fast variant's single-reciprocal form has the finite limit there, so it must hand such tiles to the reference form —
    the result is the oracle's, NaN for NaN."""
    a = synthetic.workload(21, 4, 48, reqhgt=0.05, start_doy=170)
    a["climdata"]["windspeed"] = a["climdata"]["windspeed"].copy()
    a["climdata"]["windspeed"][30] = 1e8
    got, st = _solve(a, 2)
    want = oracle.run_grid(**a)
    assert np.isnan(want["Tz"][:, :, 30]).sum() > 60 and np.isfinite(want["Tz"][:, :, 31]).all()
    assert st["irregular_days"] == 0 and st["fast_launches"] == 1 and st["canary_trips"] >= 1, st
    _same(got, want)


# ---- round 3 ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cpb", [16, 21, 42])
def test_coarse_forcing_solves_every_cell_whatever_cells_per_block_was_asked_for(oracle, cpb):
    """Coarse array forcing is built for 32-cell tiles only; a plan asked for another tile size used to launch
    ceil(N / cpb) 32-cell tiles — a quarter of the raster never solved.  The plan now forces 32: tile classes, tile lists,
    the ring's blocks and the kernel agree, with an irregular cell (slow list) in the raster."""
    a, rp, cp = synthetic.coarse_workload(45, 11, 48, 3, 4, reqhgt=0.05)
    a["soilc"]["twi"][3, 0] = np.nan           # one irregular tile
    with Plan(**a, ring_days=2, ring_slots=1, cells_per_block=cpb, coarse={"rowpos": rp, "colpos": cp}) as p:
        p.run_days(0, 2, 0)
        p.sync()
        got = {k: p.fetch(0, k, 0, 48) for k in NAMES}
        st = p.dispatch_stats()
        lay = p.ring_layout()
    assert lay["cells_per_tile"] == 32 and lay["block_doubles"] == 768, lay
    assert st["fast_tiles"] + st["slow_tiles"] == -(-45 * 11 // 32) and st["slow_tiles"] == 1, st
    from oracle import coarse_oracle as CO
    clim, pm = CO.expand(a["climdata"], a["pointm"], rp, cp)
    b = dict(a)
    b.update(climdata=clim, pointm=pm)
    _same(got, oracle.run_grid(**b, array_forcing=True))


def test_a_tile_whose_waves_all_trip_is_pushed_once(oracle):
    """Every cell of one tile makes a NaN inside a regular cell (negative soil conductivity): all eight waves of the tile's
    workgroup trip, the tile is listed ONCE (canary_trips counts tiles per launch, not waves)."""
    a = synthetic.workload(21, 4, 48, reqhgt=0.05, start_doy=170)
    a["soilc"]["Vq"][:, 1] = 1.6               # tile 1, all 21 cells
    got, st = _solve(a, 2)
    assert st["slow_tiles"] == 0 and st["canary_trips"] == 1, st
    _same(got, oracle.run_grid(**a))


def test_more_tripped_tiles_than_the_fix_list_holds_redoes_the_raster(oracle):
    """fix_cap = 8192 entries; 8400 tripped tiles overflow it and k_solve_fix redoes every tile of the launch."""
    rows, cols = 21 * 8400 // 16, 16
    a = synthetic.workload(rows, cols, 24, reqhgt=0.05, start_doy=170)
    a["soilc"]["Vq"][:, :] = 1.6
    a["soilc"]["Vq"][:21, 0] = a["soilc"]["Vq"][0, 0] * 0 + 0.3        # tile 0 stays clean
    got, st = _solve(a, 1)
    assert st["canary_trips"] == 8400 - 1, st
    # a sample of cells against the oracle (the whole raster is 176 400 cells)
    sub = dict(a)
    pick = np.r_[0:42, 5000:5042, rows * cols - 42:rows * cols]
    def take(m):
        m = np.asarray(m)
        flat = m.reshape((rows * cols,) + m.shape[2:], order="F")[pick]
        return np.asfortranarray(flat.reshape((pick.size, 1) + m.shape[2:]))
    sub["vegp"] = {k: take(v) for k, v in a["vegp"].items()}
    sub["soilc"] = {k: take(v) for k, v in a["soilc"].items()}
    import ctypes as C
    lib = oracle.load()
    lib.orc_set_twi_mean_override.argtypes = [C.c_double, C.c_int]
    tw = a["soilc"]["twi"]
    lib.orc_set_twi_mean_override(float(np.mean(np.log(tw) / a["tfact"])), 1)
    try:
        want = oracle.run_grid(**sub)
    finally:
        lib.orc_set_twi_mean_override(0.0, 0)
    sample = {k: np.asfortranarray(v.reshape(rows * cols, 24, order="F")[pick].reshape(pick.size, 1, 24)) for k, v in got.items()}
    _same(sample, want)
