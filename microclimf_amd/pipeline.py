"""Solver -> file, the body of the reference's `runmicro_big` loop for one tile (R/Cppwrappers.R:520-531:
`runmicro(...)` then `writetonc(mout, fo, dtmi, reqhgt)`), without `mout` ever existing on the host: day chunks are
solved into the plan's device ring and leave it as finished netCDF records (`mcf_nc_write_plan`).

The file write bounds the whole job (a 5-day chunk of 1024 x 1024 cells is solved in 0.03 s and takes 0.5 s to land
in the page cache), so the loop is deliberately plain: solve chunk k, write chunk k.
"""
from __future__ import annotations

import time
from typing import Mapping, Sequence

import numpy as np

from . import ncsink
from .api import Plan


def run_to_nc(inputs: Mapping, fileout: str, dtm: Mapping, *, vars: Sequence[str] | None = None,
              days_per_chunk: int = 5, device: int = 0, array_forcing: bool = False,
              reference_puts_only: bool = False, twi_mean: float | None = None, format: str = "classic",
              deflate_level: int = 0) -> dict:
    """`inputs`: the 15 arguments of runmicro1Cpp / runmicro2Cpp by name (as `synthetic.workload` returns them);
    with "dfsel" added for time-varying vegetation (runmicro3Cpp); `dtm`: {"xmin","xmax","ymin","ymax","res"[, "crs"]} of
    the tile.  Variables default to writetonc's for the height.
    `twi_mean`: the raster-wide mean of log(twi)/tfact when this tile is part of a larger raster
    (`distributed.allreduce_twi_mean`).  `format` / `deflate_level`: the file's container (ncsink: "classic", or "netcdf4" —
    the reference's, deflate 9 by default).  Returns timings and sizes."""
    reqhgt = float(inputs["reqhgt"])
    if reqhgt < 0:
        raise ValueError("reqhgt < 0 needs the whole series in the ring (Plan.belowground); fetch and use writetonc")
    names = tuple(ncsink.default_vars(reqhgt) if vars is None else vars)
    obst = inputs["obstime"]
    T = len(np.asarray(obst["hour"]))
    ndays = T // 24
    hours = ncsink.hours_since_epoch(obst)
    res = dtm["res"]
    xres, yres = (res, res) if np.isscalar(res) else res
    east, north = ncsink.coords_from_extent(dtm["xmin"], dtm["xmax"], dtm["ymin"], dtm["ymax"], xres, yres)
    a = dict(inputs)
    # solve only what the file holds (plus nothing else: the ring is sized by the requested outputs)
    from . import _abi
    a["out"] = [1 if n in names else 0 for n in _abi.OUT_NAMES]
    t_solve = t_write = 0.0
    t_begin = time.perf_counter()
    with Plan(**a, ring_days=days_per_chunk, ring_slots=1, device=device, array_forcing=array_forcing) as plan:
        if twi_mean is not None:
            plan.set_twi_mean(twi_mean)
        rows, cols = plan.rows, plan.cols
        with ncsink.NcWriter(fileout, rows, cols, hours[:ndays * 24], east, north, reqhgt, names, dtm.get("crs", ""),
                             reference_puts_only, format=format, deflate_level=deflate_level) as nc:
            t_setup = time.perf_counter() - t_begin
            for d0 in range(0, ndays, days_per_chunk):
                nd = min(days_per_chunk, ndays - d0)
                t0 = time.perf_counter()
                if array_forcing:
                    plan.upload_forcing_days(d0, nd, 0)
                plan.run_days(d0, nd, 0)
                plan.sync()
                t1 = time.perf_counter()
                nc.write_plan(plan, 0, 0, d0 * 24, nd * 24)
                t_write += time.perf_counter() - t1
                t_solve += t1 - t0
            t_loop_end = time.perf_counter()
        t_close_file = time.perf_counter() - t_loop_end
        valid = plan.valid_cells
    return {"rows": rows, "cols": cols, "steps": ndays * 24, "vars": names, "valid_cells": int(valid),
            "solve_s": t_solve, "write_s": t_write, "setup_s": t_setup, "close_file_s": t_close_file,
            "total_s": time.perf_counter() - t_begin, "values": rows * cols * ndays * 24 * len(names)}
