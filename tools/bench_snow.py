"""bench.py --config 4: BASELINE.json configs[4] — 4096 x 4096 cells x 8760 hourly steps with the snow branch, the raster
dealt to 8 ranks in row blocks (strong scaling; with fewer GPUs every rank runs its block of the 8-block partition).

One "step" = one simulated year of a rank's block through the reference's snow pipeline, device-resident:
  per 5-day chunk (R/internal.R:2553-2617 `.snowmodel1`, 73 chunks):
    128 boundary rows of the snow surface point-to-point to the neighbouring ranks, device to device (RCCL send/recv; the
    terrain stencil and `.tpicalc`'s block means reach +-128 rows), (sum, count) all-reduce of the surface
    -> terrain refresh from dtm + snow depth (slope, aspect, 24 horizons, sky view, 8 wind-shelter maps), tpic
    -> (sum, count) all-reduce of tpic -> gridmodelsnow1 on the chunk (k_snowmodel) -> redistribution, hand-over
    -> applycpp3 min / max of totalSWE per step on the device, all-reduced over the ranks (R/internal.R:3592-3593)
    -> snowdaysfun: the chunk's snow / no-snow days
The reference (`.runmicrosnow1`, R/internal.R:3581-3659) then solves the no-snow days with the grid solver, the snow days with
gridmicrosnow1 (src/microclimfCpp.cpp:4894-5056) and merges by day — with the year's snow series in memory.  Here they exist
one chunk at a time, and gridmicrosnow1 needs the per-cell mean snow damping depth of the WHOLE snow-day series (cpp:4713-4737)
and the day list itself before its first step, so the year is walked TWICE (include/mcf.h mcf_snowplan_micro_*):
  pass 1  the chunk loop above + the running sum behind the mean damping depth
  between gridmicrosnow1's set-up for the snow-day subset, the solver's maximum temperature over the no-snow subset, reset
  pass 2  the chunk loop again (without apply3) + k_solve on the chunk's no-snow days at their place in the ring slot +
          k_microsnow_ring: the snow-day microclimate written over it — the slot holds `.runmicrosnow1`'s merged output
value = valid cells x 8760 / seconds of BOTH passes: every cell-step went through the snowpack model twice, and through the
solver or gridmicrosnow1 (or both, on days that have snow somewhere and bare cells elsewhere)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np


def cpu_legs(args):
    """The three kernels of the pipeline as the oracle runs them on one host core, on a bounded sample (before torch / HIP are
    touched): gridmodelsnow1 (oracle/snow_oracle.c), the grid solver (oracle/mcf_oracle.c) and gridmicrosnow1, each over every
    step of the sample.  run_snow_config weights them by the share of days each one covers in the measured year — the
    reference's single pass over the snow model, its solver on the no-snow days, gridmicrosnow1 on the snow days."""
    from microclimf_amd import synthetic
    from oracle import oracle as O
    r, c, t = 128, 128, 240
    sw = synthetic.snow_workload(r, c, t, cold=3.0, zref=3.5, start_doy=1)
    a = synthetic.workload(r, c, t, reqhgt=args.reqhgt, zref=3.5, hgt_range=(0.05, 3.0))
    O.load()
    t0 = time.perf_counter()
    smod = O.run_snowmodel(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"])
    t_snow = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.run_grid(**a)
    t_solver = time.perf_counter() - t0
    snowm, micro = synthetic.microsnow_inputs(sw, smod)
    t0 = time.perf_counter()
    O.run_microsnow(args.reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 7.5, [1] * 10)
    t_micro = time.perf_counter() - t0
    n = float(np.count_nonzero(~np.isnan(sw["vegp"]["hgt"]))) * t
    return {"cell_steps": n, "snowmodel_s": t_snow, "solver_s": t_solver, "microsnow_s": t_micro,
            "sample": f"{r}x{c} cells x {t} hourly steps, oracle (gcc -O2, 1 thread): gridmodelsnow1 {t_snow:.1f} s, grid solver "
                      f"{t_solver:.1f} s, gridmicrosnow1 {t_micro:.1f} s (every step of the sample each)"}


def run_snow_config(args, world, rank, local_rank):
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.stdout.flush()
        _saved = os.dup(1)
        os.dup2(2, 1)
        cpu = cpu_legs(args)
        os.dup2(_saved, 1)
        os.close(_saved)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MCF_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks share a device,
    # the collectives run over gloo on host tensors); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("MCF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("MCF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build_library()
    if use_dist:
        dist.barrier()
    from microclimf_amd import synthetic
    from microclimf_amd.api import Plan
    from microclimf_amd.distributed import allreduce_apply3, allreduce_max, allreduce_sum, allreduce_twi_mean, row_block
    from microclimf_amd.snow import DeviceHalo, SnowPlan, snowdaysfun

    rows_total, cols, T = args.rows, args.cols, args.tsteps
    nblocks = max(world, args.share or 0)
    row0, rows = row_block(rank, nblocks, rows_total)
    exchange_ok = nblocks == world
    ndays, chunk_days = T // 24, 5
    t0 = time.perf_counter()
    sw = synthetic.snow_workload(rows, cols, T, cold=3.0, zref=3.5, row0=row0, rows_total=rows_total, start_doy=1)
    _, _, dtm = synthetic.rasters(rows, cols, row0, rows_total)
    dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
    from microclimf_amd.terrain import HALO
    halo_n = halo_s = None
    if not exchange_ok:
        # a rank's share without its neighbours: the halo rows are the neighbouring blocks' snow-FREE surface (generated from
        # the seeded DTM, uploaded once), not their snow surface — the exchange itself only runs when every block has its rank
        hn_, hs_ = min(HALO, row0), min(HALO, rows_total - (row0 + rows))
        dev = torch.device("cuda", local_rank)
        if hn_:     # a piece is column-major [h, cols] = a contiguous [cols, h] tensor
            halo_n = torch.from_numpy(np.ascontiguousarray(synthetic.rasters(hn_, cols, row0 - hn_, rows_total)[2].T)).to(dev)
        if hs_:
            halo_s = torch.from_numpy(np.ascontiguousarray(synthetic.rasters(hs_, cols, row0 + rows, rows_total)[2].T)).to(dev)
    a = synthetic.workload(rows, cols, T, reqhgt=args.reqhgt, row0=row0, rows_total=rows_total, zref=3.5, hgt_range=(0.05, 3.0))
    setup_s = time.perf_counter() - t0
    sp = SnowPlan(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02,
                  row0=row0, rows_total=rows_total, device=local_rank, keep_results=False)
    plan = Plan(**a, ring_days=chunk_days, ring_slots=2, device=local_rank)
    halo = DeviceHalo(sp, rank, world) if exchange_ok and world > 1 else None
    s, n = plan.twi_partial()
    twi_mean = allreduce_twi_mean(s, float(n))
    plan.set_twi_mean(twi_mean)
    valid = plan.valid_cells
    stats = {"solver_days": 0, "snow_days": 0, "years": 0}

    def fence():
        plan.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    stage_s = {}
    timing = os.environ.get("MCF_BENCH_STAGES") == "1"

    def lap(name, t_last):
        if not timing:
            return t_last
        plan.sync()
        torch.cuda.synchronize()
        now = time.perf_counter()
        stage_s[name] = stage_s.get(name, 0.0) + now - t_last
        return now

    def snow_chunk(ch):
        tl = time.perf_counter()
        # the surface never leaves the device: its 128 boundary rows go point-to-point to the neighbouring ranks
        pn, ps = halo.exchange() if halo is not None else (halo_n, halo_s)
        tl = lap("halo", tl)
        ss, sn = sp.surface_partial()
        smean = allreduce_twi_mean(ss, sn)                      # (sum, count) -> mean over the whole raster
        ts, tn = sp.prepare_chunk_dev(ch, pn, ps, smean)
        tl = lap("terrain+tpic", tl)
        sp.run_chunk(ch, allreduce_twi_mean(ts, tn))
        return lap("snowmodel+redistribute", tl)

    def steps_of(days0):
        return (np.repeat(np.asarray(days0, dtype=np.int64) * 24, 24) + np.tile(np.arange(24), len(days0))).astype(np.int64)

    def sub(d, idx):
        return {k: (np.asarray(v)[idx] if np.ndim(v) == 1 else v) for k, v in d.items()}

    # pass 1's series of a snow chunk stay on the device while this much of it remains free (the solver's ring, the snow plan
    # and the micro set-up are allocated by then; MCF_SNOW_KEEP_RESERVE_GB=1e9 switches the cache off)
    keep_reserve = int(float(os.environ.get("MCF_SNOW_KEEP_RESERVE_GB", "24")) * 2**30)
    outm = [1] * 10 if args.reqhgt > 0 else [1 if i in (0, 3, 5, 6, 7, 8, 9) else 0 for i in range(10)]
    ncd = sp.chunks * chunk_days
    tile_skip = os.environ.get("MCF_SNOW_NO_TILE_SKIP") is None
    cell_gather = os.environ.get("MCF_SNOW_NO_CELL_GATHER") is None

    def one_year(probe=None):
        """probe (the untimed verification year): {"cells": ...} — the sample cells' five snow series are read back behind
        every chunk of pass 1 and their ten merged outputs behind every chunk of pass 2 (nothing is kept in HBM then: the
        read-back sees the plan's working buffers)"""
        # ---- pass 1: snow series chunk by chunk -> day classes, running sum of the snow damping depth
        snowday, nosnowday = np.zeros(ncd, np.int32), np.zeros(ncd, np.int32)
        sp.release_kept()
        kept = [False] * sp.chunks
        for ch in range(sp.chunks):
            sp.checkpoint(ch)                                   # 24 B per cell: pass 2 starts any chunk from here
            # a chunk that could not stay in HBM anyway is re-run by pass 2 if it holds a snow day: pass 1 writes only what it
            # reads itself of such a chunk — totalSWE (day classes) and the density (mean damping depth), 16 of 40 B per cell-step
            full = probe is not None or sp.can_keep(keep_reserve)
            sp.set_series(sp.SERIES_ALL if full else sp.SERIES_PASS1)
            tl = snow_chunk(ch)
            mx, cmx = sp.apply3(ch, "max")
            mn, cmn = sp.apply3(ch, "min")
            mx, mn = allreduce_apply3(mx, cmx, "max"), allreduce_apply3(mn, cmn, "min")
            days = snowdaysfun(mx, mn)
            d0 = ch * chunk_days
            snowday[d0:d0 + chunk_days], nosnowday[d0:d0 + chunk_days] = days["snowdays"], days["nosnowdays"]
            tl = lap("apply3", tl)
            sp.meand_accumulate(ch, days["snowdays"])
            if probe is not None:
                for name in probe["smod"]:
                    probe["smod"][name][:, ch * 120:(ch + 1) * 120] = sp.fetch_cells(name, probe["cells"])
            elif full and days["snowdays"].any():               # its five series stay in HBM for pass 2 while room remains
                kept[ch] = sp.keep_chunk(ch, reserve_bytes=keep_reserve)
            tl = lap("meanD", tl)
        # ---- between: gridmicrosnow1's set-up on the snow-day subset, the solver's maximum temperature on the no-snow subset
        tl = time.perf_counter()
        sp.set_series(sp.SERIES_ALL)                            # (pass 2's re-runs feed the snow microclimate)
        sdays, ndays_ = np.flatnonzero(snowday), np.flatnonzero(nosnowday)
        stats["snow_days"] += int(sdays.size)
        if sdays.size:
            si = steps_of(sdays)
            sod = np.full(ncd, -1, np.int32)
            sod[sdays] = np.arange(sdays.size)
            sp.micro_setup(args.reqhgt, sub(sw["obstime"], si), sub(sw["climdata"], si), sw["vegp"], sw["other"], 7.5, outm, sod,
                           reuse_static=stats["years"] > 0)       # vegetation / terrain matrices go up once per plan
        stats["years"] += 1
        if ndays_.size:
            plan.set_mxtc(float(np.max(a["climdata"]["temp"][steps_of(ndays_[ndays_ < ndays])])))
        tl = lap("micro set-up", tl)
        # ---- pass 2: the series again + the solver on the no-snow days + the snow-day microclimate over it
        slot = 0
        for ch in range(sp.chunks):
            d0 = ch * chunk_days
            nos = nosnowday[d0:d0 + chunk_days]
            has_snow = bool(snowday[d0:d0 + chunk_days].any())
            tl = time.perf_counter()
            if has_snow and not kept[ch]:                         # a chunk without a snow day is the solver's alone
                sp.restore(ch)
                tl = snow_chunk(ch)
            elif has_snow:
                stats["chunks_kept"] = stats.get("chunks_kept", 0) + 1
            else:
                stats["chunks_skipped"] = stats.get("chunks_skipped", 0) + 1
            k = 0
            while k < len(nos):                                       # runs of consecutive no-snow days of one class -> one launch each
                if not nos[k]:
                    k += 1
                    continue
                both = bool(snowday[d0 + k])
                e = k
                while e < len(nos) and nos[e] and bool(snowday[d0 + e]) == both:
                    e += 1
                if d0 + k < ndays:
                    nd = min(e, ndays - d0) - k
                    # a day that is a snow day too: the solver's values survive the merge (`.runmicrosnow1`, step 5) only where a cell
                    # is not under snow — those cells gathered into tiles of their own (mcf_plan_run_days_cells), or, where they
                    # are many, the tiles whose cells ALL lie under snow for the whole run left out of the launch
                    if tile_skip and sdays.size and both:
                        need, n_need = sp.free_cells(plan, ch, k, nd) if cell_gather else (0, plan.rows * plan.cols)
                        if os.environ.get("MCF_SNOW_DEBUG"):
                            print(f"[cells] chunk {ch} days {k}+{nd}: {n_need / (plan.rows * plan.cols):.3f} of the cells not under snow throughout",
                                  file=sys.stderr, flush=True)
                        if 16 * n_need <= plan.rows * plan.cols:      # (profiles/r05_cells_rate.txt: pays below ~ 8 % of the cells)
                            plan.run_days_cells(d0 + k, nd, slot, k, need)
                            stats["tile_days_skipped"] = stats.get("tile_days_skipped", 0) + (plan.n_tiles - -(-n_need // plan.ring_layout()["cells_per_tile"])) * nd
                            stats["cell_days_gathered"] = stats.get("cell_days_gathered", 0) + n_need * nd
                        else:
                            sk, ncov = sp.covered_tiles(plan, ch, k, nd)
                            if ncov:
                                plan.run_days_masked(d0 + k, nd, slot, k, sk)
                                stats["tile_days_skipped"] = stats.get("tile_days_skipped", 0) + ncov * nd
                            else:
                                plan.run_days_at(d0 + k, nd, slot, k)
                    else:
                        plan.run_days_at(d0 + k, nd, slot, k)
                    stats["tile_days"] = stats.get("tile_days", 0) + plan.n_tiles * nd
                    stats["solver_days"] += nd
                k = e
            tl = lap("solver", tl)
            if sdays.size and has_snow:
                sp.microsnow(plan, ch, slot, nos)
            tl = lap("microsnow", tl)
            if probe is not None:
                for name in probe["out"]:
                    probe["out"][name][:, ch * 120:(ch + 1) * 120] = plan.fetch_cells(slot, name, 0, 120, probe["cells"])
            slot = (slot + 1) % 2
        if probe is not None:
            probe["snowday"], probe["nosnowday"] = snowday.copy(), nosnowday.copy()
        return snowday

    def verify_sample(snowday, ncells=256):
        """The snow model of the year's LAST chunk with a snow day, re-run from the timed run's own checkpoint and held against
        the oracle for a sample of cells: the oracle gets the cells' hand-over state and the chunk's terrain (slope, aspect,
        sky view, wind shelter, horizons — products of neighbourhood operators the oracle cannot redo for a sample) from the
        device and runs gridmodelsnow1 on the chunk's 120 steps.  Every rank re-runs the chunk (its halo exchange and
        reductions are collective); rank 0 compares its own block."""
        chs = [ch for ch in range(sp.chunks) if snowday[ch * chunk_days:(ch + 1) * chunk_days].any()]
        if not chs:
            return {"ok": None, "note": "no snow day in the year"}
        ch = chs[-1]
        sp.restore(ch)
        hg = np.asarray(sw["vegp"]["hgt"]).ravel(order="F")
        ok_cells = np.flatnonzero(~np.isnan(hg))
        cells = ok_cells[np.linspace(0, ok_cells.size - 1, min(ncells, ok_cells.size)).astype(np.int64)]
        state = {k: sp.fetch_cells(k, cells)[:, 0] for k in ("isnowdc", "isnowac", "isnowag")}
        snow_chunk(ch)
        if rank != 0:
            return None
        from oracle import oracle as O
        O.load()
        K = cells.size
        k0, ns = ch * chunk_days * 24, chunk_days * 24
        sl = slice(k0, k0 + ns)
        col = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float64).ravel(order="F")[cells].reshape(K, 1))
        oth = dict(sw["other"])
        for k in ("slope", "aspect", "skyview"):
            oth[k] = np.asfortranarray(sp.fetch_cells(k, cells).reshape(K, 1))
        oth["wsa"] = np.asfortranarray(sp.fetch_cells("wsa", cells).reshape(K, 1, 8))
        oth["hor"] = np.asfortranarray(sp.fetch_cells("hor", cells).reshape(K, 1, 24))
        oth["isnowdc"] = np.asfortranarray(state["isnowdc"].reshape(K, 1))
        oth["isnowdg"] = col(sw["other"]["isnowdg"])
        oth["isnowac"] = np.asfortranarray(state["isnowac"].reshape(K, 1))
        oth["isnowag"] = np.asfortranarray(state["isnowag"].reshape(K, 1))
        vg = {k: col(v) for k, v in sw["vegp"].items()}
        cut = lambda d: {k: (np.asarray(v)[sl] if np.ndim(v) == 1 else v) for k, v in d.items()}
        want = O.run_snowmodel(cut(sw["obstime"]), cut(sw["climdata"]), cut(sw["pointm"]), vg, oth, sw["snowenv"])
        worst, bad = 0.0, []
        # (ground snow depth and totalSWE leave the chunk redistributed by the topographic position index: not comparable)
        for name, key in (("Tc", "Tc"), ("Tg", "Tg"), ("snowden", "sden")):
            g = sp.fetch_cells(name, cells)
            w = want[key].reshape(K, ns, order="F")
            if not np.array_equal(np.isnan(g), np.isnan(w)):
                bad.append(name + ": NA pattern")
                continue
            f = np.isfinite(w)
            e = float(np.max(np.abs(g[f] - w[f]) / (1 + np.abs(w[f])))) if f.any() else 0.0
            worst = max(worst, e)
            if e > 1e-6:
                bad.append(f"{name}: {e:.2e}")
        return {"ok": not bad, "max_scaled_err": worst, "tolerance": 1e-6, "cells": int(K), "steps": int(ns), "chunk": int(ch),
                "what": "gridmodelsnow1 of the year's last chunk with a snow day (Tc, Tg and snow density of every step), re-run "
                        "from the timed run's checkpoint, against oracle/snow_oracle.c given the cells' hand-over state and the "
                        "chunk's terrain from the device; the terrain refresh, the tpi redistribution (which rewrites the chunk's snow depths) and the snow-day microclimate are "
                        "held against the oracle by tests/test_snow_gpu.py, test_terrain_gpu.py and test_snow_micro_pipeline_gpu.py "
                        "on small rasters", "mismatches": bad or None}

    def verify_merged(ncells=256):
        """What the line's cell-steps count, against the oracle: one more (untimed) year with a probe on a sample of cells —
        their five snow series out of pass 1 and their ten MERGED outputs out of the ring slot behind every chunk of pass 2
        — and, on the host, `.runmicrosnow1`'s orchestration (R/internal.R:3581-3659) with the oracle behind it on exactly
        those cells: oracle/mcf_oracle.c on the no-snow-day subset, oracle/snow_oracle.c (gridmicrosnow1) on the snow-day
        subset given the device's snow series, merged by day.  Every rank runs the year (its exchanges are collective);
        rank 0 compares its own block."""
        import ctypes as C
        from oracle import snowmerge_oracle as MO          # the merge's own restatement (R/internal.R:3565-3578, 3625-3656)
        hg = np.asarray(sw["vegp"]["hgt"]).ravel(order="F")
        ok_cells = np.flatnonzero(~np.isnan(hg))
        na_cells = np.flatnonzero(np.isnan(hg))[:4]
        cells = np.unique(np.concatenate([ok_cells[np.linspace(0, ok_cells.size - 1, min(ncells, ok_cells.size)).astype(np.int64)],
                                          na_cells])).astype(np.int64)
        K, TT = cells.size, sp.chunks * 120
        names = [n for i, n in enumerate(("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup",
                                          "Rlwup")) if a["out"][i]]
        probe = {"cells": cells, "smod": {k: np.full((K, TT), np.nan) for k in ("Tc", "Tg", "groundsnowdepth", "snowden", "totalSWE")},
                 "out": {k: np.full((K, TT), np.nan) for k in names}}
        one_year(probe)
        if rank != 0:
            return None
        from oracle import oracle as O
        lib = O.load()
        snowday, nosnowday = probe["snowday"][:ndays], probe["nosnowday"][:ndays]
        nd_cov = min(ndays, sp.chunks * chunk_days)              # days past the last whole chunk are the solver's alone: not probed
        sdays = np.flatnonzero(snowday[:nd_cov])
        ndays_ = np.flatnonzero(nosnowday[:nd_cov])
        neither = np.flatnonzero((snowday[:nd_cov] == 0) & (nosnowday[:nd_cov] == 0))
        col = lambda m: np.asfortranarray(np.asarray(m, dtype=np.float64).reshape(-1, order="F")[cells].reshape(K, 1))      # noqa: E731
        col3 = lambda m: np.asfortranarray(np.asarray(m).reshape(rows * cols, -1, order="F")[cells].reshape(K, 1, -1))    # noqa: E731
        sub = lambda d, idx: {k: (np.asarray(v)[idx] if np.ndim(v) == 1 else v) for k, v in d.items()}                       # noqa: E731
        # ---- the no-snow solver on the no-snow-day subset (its temperature cap takes the SUBSET's maximum, cpp:2159-2168)
        ni_all = steps_of(np.flatnonzero(nosnowday[:ndays]))
        an = dict(a, obstime=sub(a["obstime"], ni_all), climdata=sub(a["climdata"], ni_all), pointm=sub(a["pointm"], ni_all))
        an["vegp"] = {k: col(v) for k, v in a["vegp"].items()}
        an["soilc"] = {k: (col3(v) if np.ndim(v) == 3 else col(v)) for k, v in a["soilc"].items()}
        lib.orc_set_twi_mean_override.argtypes = [C.c_double, C.c_int]
        lib.orc_set_twi_mean_override(float(twi_mean), 1)
        t0 = time.perf_counter()
        try:
            moutn = O.run_grid(**an)
        finally:
            lib.orc_set_twi_mean_override(0.0, 0)
        keepn = np.repeat(np.flatnonzero(nosnowday[:ndays]) < nd_cov, 24)
        moutn = {k: np.asfortranarray(v[:, :, keepn]) for k, v in moutn.items()}
        # ---- gridmicrosnow1 on the snow-day subset, the device's snow series behind it
        si, ni = steps_of(sdays), steps_of(ndays_)
        rank_of_day = np.cumsum((snowday[:nd_cov] | nosnowday[:nd_cov]).astype(np.int64)) - 1
        micro = MO.prep_micro(moutn, rank_of_day[sdays] + 1, rank_of_day[ndays_] + 1, K, 1)
        swe = probe["smod"]["totalSWE"].copy()
        swe[np.isnan(swe)] = 0.0
        swe[np.isnan(hg[cells])] = np.nan
        smods = {k: np.asfortranarray((swe if k == "totalSWE" else v)[:, si].reshape(K, 1, si.size)) for k, v in probe["smod"].items()}
        vg = {k: col(v) for k, v in sw["vegp"].items()}
        oth = {k: (col3(v) if np.ndim(v) == 3 else col(v) if np.ndim(v) == 2 else v) for k, v in sw["other"].items()}
        mouts = O.run_microsnow(args.reqhgt, sub(sw["obstime"], si), sub(sw["climdata"], si), smods, micro, vg, oth, 7.5, outm)
        for k in moutn:
            if k not in mouts:
                mouts[k] = micro[k]
        # (days in neither class — a melted pack's negative rounding residue — have no place in the reference's merge, which
        # indexes by absolute hour, R/internal.R:3650-3655: the days are renumbered without them)
        want = MO.merge(moutn, mouts, rank_of_day[sdays] + 1, rank_of_day[ndays_] + 1, K, 1)
        t_or = time.perf_counter() - t0
        # ---- the comparison: every day that is in a class (the reference's merge has no place for the others)
        inclass = np.repeat((snowday[:nd_cov] | nosnowday[:nd_cov]).astype(bool), 24)
        worst, bad, nvals = 0.0, [], 0
        cls = {"snow_covered": 0, "snow_free_on_a_snow_day": 0, "no_snow_day": 0}
        for k, w in want.items():
            g = probe["out"][k][:, :nd_cov * 24][:, inclass]
            w = w.reshape(K, -1)
            if g.shape != w.shape:
                bad.append(f"{k}: shape {g.shape} vs {w.shape}")
                continue
            if not np.array_equal(np.isnan(g), np.isnan(w)):
                bad.append(k + ": NA pattern")
                continue
            f = np.isfinite(w)
            e = float(np.max(np.abs(g[f] - w[f]) / (1 + np.abs(w[f])))) if f.any() else 0.0
            worst = max(worst, e)
            nvals += int(f.sum())
            if e > 1e-6:
                bad.append(f"{k}: {e:.2e}")
        day_of = np.repeat(np.arange(nd_cov), 24)[inclass]
        cov = (swe[:, :nd_cov * 24][:, inclass] > 0)
        valid_row = ~np.isnan(hg[cells])[:, None]
        cls["snow_covered"] = int((cov & valid_row).sum())
        cls["snow_free_on_a_snow_day"] = int((~cov & valid_row & snowday[day_of].astype(bool)[None, :]).sum())
        cls["no_snow_day"] = int((valid_row & (snowday[day_of] == 0)[None, :]).sum())
        return {"ok": not bad, "max_scaled_err": worst, "tolerance": 1e-6, "cells": int(K), "days": int(nd_cov - neither.size),
                "days_in_neither_class": int(neither.size), "values": nvals, "cell_steps_by_class": cls, "outputs": names,
                "what": "the ten MERGED outputs of a whole (untimed, probed) year for a sample of cells — read out of the solver's "
                        "ring slot behind every chunk of pass 2 — against `.runmicrosnow1`'s orchestration with the oracle behind it: "
                        "oracle/mcf_oracle.c on the no-snow-day subset, oracle/snow_oracle.c gridmicrosnow1 on the snow-day subset "
                        f"given the device's snow series of those cells, merged by day ({t_or:.1f} s of oracle)",
                "mismatches": bad or None}

    for _ in range(args.warmup):
        one_year()
    stage_s.clear()             # (MCF_BENCH_STAGES=1: the timed years only — the warm-up year holds the allocations)
    fence()
    stats["solver_days"] = stats["snow_days"] = 0
    t0 = time.perf_counter()
    last_days = None
    for _ in range(args.steps):
        last_days = one_year()
    fence()
    dt = allreduce_max(time.perf_counter() - t0)
    timed_stats = dict(stats)           # (the verification below walks one more year)
    verified = None
    if not getattr(args, "no_verify", False) and last_days is not None:
        vm = verify_merged()
        vs = verify_sample(last_days)
        if rank == 0:
            verified = dict(vm)
            verified["snowmodel"] = vs
            verified["ok"] = bool(vm["ok"] and (vs.get("ok") is not False))
    valid_all = allreduce_sum(float(valid))
    value = valid_all * ndays * 24 * args.steps / dt
    if rank == 0:
        sd = timed_stats["solver_days"] / max(args.steps, 1)
        snd = timed_stats["snow_days"] / max(args.steps, 1)
        # 5 snow series per cell-step in each of the two passes + 10 outputs per cell-step of a solver day or a snow day
        passes = 1.0 + (sp.chunks - (stats.get("chunks_skipped", 0) + stats.get("chunks_kept", 0)) / max(stats["years"], 1)) / sp.chunks
        alg = valid_all * 24 * (passes * ndays * 40.0 + (sd + snd) * 80.05) * args.steps
        # counter bytes of the pipeline's kernels for ONE simulated year (the profiled command runs one), beside the algorithmic ones
        import bench as B
        ptraf, pnote = B.committed_pipeline_traffic()
        alg_year = alg / max(args.steps, 1)
        line = {
            "metric": "cell-steps/s", "value": value, "unit": "cell-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{rows_total}x{cols} synthetic DTM in {nblocks} row blocks, {world} of them solved by {world} GPU(s), {T} hourly "
                            "steps, snow branch: `.snowmodel1` chunk loop (terrain refresh from dtm + snow every 5 days, gridmodelsnow1, "
                            "`.tpicalc` redistribution) + applycpp3 min/max of totalSWE, walked twice; the grid solver on the no-snow days, "
                            "gridmicrosnow1 on the snow days, merged in the device ring as `.runmicrosnow1` does "
                            "[BASELINE.json configs[4]]",
                "baseline_config": 4, "rows_per_gpu": rows, "cols": cols, "tsteps": T, "valid_cells": int(valid_all),
                "solver_days_per_year": sd,
                "solver_tile_days_left_out": (f"{stats.get('tile_days_skipped', 0) / max(stats.get('tile_days', 0), 1):.3f} of the solver's tile-days: "
                                              "on a day that is a snow day as well, gridmicrosnow1 overwrites every value of a cell under snow "
                                              "(`.runmicrosnow1`'s merge): only the cells not under snow throughout are solved, gathered into tiles "
                                              "of their own (mcf_plan_run_days_cells; where they are many: the tiles wholly under snow left out) — "
                                              "the merged output is the same (MCF_SNOW_NO_CELL_GATHER=1: tiles as the unit; MCF_SNOW_NO_TILE_SKIP=1 "
                                              "solves every cell)") if tile_skip else "none (MCF_SNOW_NO_TILE_SKIP)",
                "solver_cell_days_gathered": stats.get("cell_days_gathered", 0) // max(stats.get("years", 1), 1),
                "snow_days_per_year": snd,
                "halo": (("RCCL" if backend == "nccl" else backend + " (REHEARSAL: ranks share a GPU)")
                         + " send/recv of 128 surface rows per neighbour and chunk, packed and unpacked on the device") if exchange_ok and world > 1 else
                        "generated, not exchanged: the neighbouring blocks' snow-free surface, resident on the device (a rank's share of "
                        "the partition without its neighbours)" if not exchange_ok else "single block",
                "collectives": "per chunk: 2 (sum, count) all-reduces + min / max all-reduce of [120] doubles; once: twi (sum, count)",
                "passes": "2 over the snow series (gridmicrosnow1 needs the whole series' mean snow damping depth and day list "
                          "first); pass 1 checkpoints every chunk's start state (24 B per cell) and, of a chunk that cannot stay in HBM, "
                          "stores only the two series it reads itself (totalSWE, density: 16 of 40 B per cell-step); pass 2 restores and re-runs only the "
                          f"chunks that hold a snow day ({stats.get('chunks_skipped', 0) // max(stats['years'], 1)} of {sp.chunks} skipped per year) "
                          f"and whose series did not stay in HBM ({stats.get('chunks_kept', 0) // max(stats['years'], 1)} kept per year, 10 GB each "
                          "for a 512 x 4096 block; the cache is allocated once per plan — in the warm-up year here, ~0.25 s per chunk — and "
                          "reused by every later year: it pays in multi-year runs)",
                "sink": "solver: HBM ring (2 slots x 5 days); snow series: chunk buffers on the device, no D2H",
            },
            "input_setup_s": setup_s,
            "stage_seconds": stage_s or None,
            "verified": verified,
            "roofline": {"bound": "fp64_valu", "achieved": alg / dt / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": alg / dt / 1e9 / (8000.0 * world), "traffic": (ptraf or {}).get("bytes_per_profiled_run"),
                         "algorithmic_bytes_per_year": alg_year,
                         "traffic_basis": ({**ptraf, "per": "one simulated year of one rank's block (the profiled command; k_solve's fix-up and the "
                                                       "small reduction kernels are not in the sum)",
                                            "ratio_to_algorithmic": ptraf["bytes_per_profiled_run"] / alg_year} if ptraf else None),
                         "counters": pnote,
                         "kernel": "pipeline: 2 x (k_snowmodel + terrain) + k_solve + k_microsnow_tiles",
                         "frac_is": "algorithmic bytes (40 B per snow-model cell-step of pass 1 and of the chunks pass 2 re-runs + 80 B per solver or snow-microclimate cell-step) / time / 8 TB/s per GPU"},
        }
        if cpu is not None:
            # one pass over the snow model for every step + the solver / gridmicrosnow1 on their shares of the year's days
            per_cs = (cpu["snowmodel_s"] + cpu["solver_s"] * sd / ndays + cpu["microsnow_s"] * snd / ndays) / cpu["cell_steps"]
            line["cpu_baseline"] = {"value": 1.0 / per_cs, "unit": "cell-steps/s", "cores": 1, "kind": "port",
                                    "sample": cpu["sample"] + f"; weighted as the measured year: solver on {sd:.0f}, "
                                              f"gridmicrosnow1 on {snd:.0f} of {ndays} days, the snow model once per step"}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    rc = 0
    if rank == 0 and verified is not None and verified.get("ok") is False:
        print("bench.py --config 4: the timed run's snow series do NOT match the oracle: " + json.dumps(verified), file=sys.stderr)
        rc = 3
    plan.close()
    sp.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc
