"""The N>1 path on CPU: two ranks over gloo.  Each rank generates its own row block of
the raster, contributes its partial (sum, count) of log(twi)/tfact to the all-reduce,
solves its block with the GLOBAL mean, and the gathered blocks must equal the
single-process solve of the whole raster.  The oracle stands in for the device kernels
(which need a GPU); partition, seeding-by-global-index and the collective are the code
under test."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from microclimf_amd import synthetic
from microclimf_amd.distributed import allreduce_max, allreduce_twi_mean, row_block

ROWS, COLS, T = 13, 7, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        lib = O.load()
        row0, rows = row_block(rank, world, ROWS)
        a = synthetic.workload(rows, COLS, T, reqhgt=0.05, row0=row0, rows_total=ROWS, variety=True,
                               start_doy=170)
        twi = a["soilc"]["twi"]
        ok = ~np.isnan(twi)
        s, n = float(np.sum(np.log(twi[ok]) / a["tfact"])), float(ok.sum())   # stands in for mcf_plan_twi_partial
        mean = allreduce_twi_mean(s, n)
        lib.orc_set_twi_mean_override.argtypes = [C.c_double, C.c_int]
        lib.orc_set_twi_mean_override(mean, 1)
        res = O.run_grid(**a)
        lib.orc_set_twi_mean_override(0.0, 0)
        tmax = allreduce_max(float(rank + 1))
        q.put((rank, row0, rows, mean, tmax, {k: np.ascontiguousarray(v) for k, v in res.items()}))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_row_block_partition():
    for world in (1, 2, 3, 8):
        blocks = [row_block(r, world, 1030) for r in range(world)]
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == 1030
        for (r0, n0), (r1, _) in zip(blocks, blocks[1:]):
            assert r0 + n0 == r1
        assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1


def test_blocks_are_slices_of_the_whole_raster():
    whole = synthetic.workload(ROWS, COLS, T, variety=True)
    r0, n = row_block(1, 2, ROWS)
    blk = synthetic.workload(n, COLS, T, row0=r0, rows_total=ROWS, variety=True)
    for grp in ("vegp", "soilc"):
        for k, v in whole[grp].items():
            assert np.array_equal(blk[grp][k], v[r0:r0 + n], equal_nan=True), (grp, k)


def test_two_ranks_gloo_equal_single_process(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = synthetic.workload(ROWS, COLS, T, reqhgt=0.05, variety=True, start_doy=170)
    want = oracle.run_grid(**whole)
    twi = whole["soilc"]["twi"]
    gmean = np.sum(np.log(twi) / whole["tfact"]) / twi.size
    for rank, row0, rows, mean, tmax, res in got:
        assert mean == pytest.approx(gmean, rel=1e-13)
        assert tmax == 2.0
        for k, v in res.items():
            np.testing.assert_allclose(v, want[k][row0:row0 + rows], rtol=1e-12, atol=1e-12, err_msg=f"{k} rank {rank}")
