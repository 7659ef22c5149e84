#!/usr/bin/env python3
"""What the solver's launch over a SUBSET of the cells costs (mcf_plan_run_days_cells: the cells gathered into tiles of their
own) beside the plain launch and the launch that leaves out whole tiles, by the share of cells marked — scattered at random,
the worst case for tiles as the unit.  1024 x 1024 cells, runs of 1 and 5 days."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402

R = C = 1024
w = synthetic.workload(R, C, 240, reqhgt=0.05)
rng = np.random.default_rng(5)
u = rng.random(R * C)
with Plan(**w, ring_days=5, ring_slots=2) as p:
    cpt = p.ring_layout()["cells_per_tile"]
    nt = p.n_tiles

    def timed(f, reps=5):
        f()
        p.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        p.sync()
        return 1e3 * (time.perf_counter() - t0) / reps

    for nd in (1, 5):
        dense = timed(lambda: p.run_days_at(0, nd, 0, 0))
        print(f"{nd} day(s): plain launch {dense:.3f} ms", flush=True)
        for share in (0.01, 0.02, 0.05, 0.10, 0.25, 0.50):
            need = (u < share).astype(np.uint8)
            flags = torch.from_numpy(need).to("cuda:0")
            tile_has = np.zeros(nt, bool)
            np.logical_or.at(tile_has, np.arange(R * C) // cpt, need.astype(bool))
            skip = (~tile_has).astype(np.uint8)
            cells = timed(lambda: p.run_days_cells(0, nd, 1, 0, flags.data_ptr()))
            tiles = timed(lambda: p.run_days_masked(0, nd, 1, 0, skip))
            print(f"   {share:5.2f} of the cells marked: gathered {cells:7.3f} ms ({cells / dense:4.2f} x plain), whole tiles left out "
                  f"{tiles:7.3f} ms ({tiles / dense:4.2f} x; {skip.mean():.2f} of the tiles hold no marked cell)", flush=True)
