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
from microclimf_amd.distributed import allreduce_apply3, allreduce_max, allreduce_twi_mean, row_block

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


def test_balanced_row_blocks_follow_the_valid_cells():
    from microclimf_amd.distributed import balanced_row_blocks
    rng = np.random.default_rng(5)
    valid = np.full(1030, 1024.0)
    valid[:400] = 0                                   # sea in the north
    valid[400:520] = rng.integers(0, 300, 120)        # a ragged coast
    for world in (1, 2, 3, 8):
        blocks = balanced_row_blocks(valid, world)
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == 1030
        for (a, n), (b, _) in zip(blocks, blocks[1:]):
            assert a + n == b and n > 0 and b % 10 == 0
        work = np.array([valid[a:a + n].sum() for a, n in blocks])
        assert work.max() <= valid.sum() / world + 10 * 1024          # within one 10-row group of the ideal share
    eq = np.array([valid[a:a + n].sum() for a, n in [row_block(r, 8, 1030) for r in range(8)]])
    bal = np.array([valid[a:a + n].sum() for a, n in balanced_row_blocks(valid, 8)])
    assert bal.max() < 0.75 * eq.max()                                 # equal rows would leave three ranks idle
    # degenerate inputs: all-NA raster, fewer row groups than ranks, fewer rows than ranks
    assert [n for _, n in balanced_row_blocks(np.zeros(80), 4)] == [20, 20, 20, 20]
    b = balanced_row_blocks(np.ones(25), 4)
    assert sum(n for _, n in b) == 25 and all(n > 0 for _, n in b)
    b = balanced_row_blocks(np.ones(3), 4)
    assert sum(n for _, n in b) == 3


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


def _apply3_numpy(a, fun):
    """applycpp3 (cpp:5553-5588) in numpy: NA-skipping reduction over space per time step."""
    with np.errstate(invalid="ignore", all="ignore"):
        flat = a.reshape(-1, a.shape[2], order="F")
        ok = ~np.isnan(flat)
        if fun == "sum":
            return np.where(ok, flat, 0.0).sum(axis=0), ok.sum(axis=0).astype(float)
        if fun == "max":
            return np.where(ok, flat, -np.inf).max(axis=0), ok.sum(axis=0).astype(float)
        return np.where(ok, flat, np.inf).min(axis=0), ok.sum(axis=0).astype(float)


def _apply3_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        swe = _swe_field()
        row0, rows = row_block(rank, world, swe.shape[0])
        blk = swe[row0:row0 + rows]
        out = {}
        for fun in ("sum", "max", "min"):
            loc, cnt = _apply3_numpy(blk, fun)            # stands in for mcf_applycpp3 on the rank's GPU
            out[fun] = allreduce_apply3(loc, cnt, fun)
        loc, cnt = _apply3_numpy(blk, "sum")
        out["mean"] = allreduce_apply3(loc, cnt, "mean")
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _swe_field():
    rng = np.random.default_rng(5)
    swe = np.asfortranarray(np.maximum(rng.normal(2.0, 3.0, (11, 6, 30)), 0.0))
    swe[2, 3, :] = np.nan
    swe[:, :, 7] = np.nan                 # an all-NA step: max -Inf, min +Inf, mean NaN
    swe[:, :, 12] = 0.0
    return swe


def test_applycpp3_partials_combine_over_two_ranks():
    """the snow branch's per-step min / max of totalSWE over space (R/internal.R:3592-3593) for a
    row-tiled raster: per-rank partials + one all-reduce equal the single-process reduction"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_apply3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    swe = _swe_field()
    want = {f: _apply3_numpy(swe, f)[0] for f in ("sum", "max", "min")}
    s, n = _apply3_numpy(swe, "sum")
    with np.errstate(invalid="ignore", divide="ignore"):
        want["mean"] = np.where(n > 0, s / n, np.nan)
    for rank, out in got:
        for f in want:
            assert np.allclose(out[f], want[f], rtol=1e-13, equal_nan=True), (rank, f)
    assert np.isneginf(want["max"][7]) and np.isposinf(want["min"][7]) and np.isnan(want["mean"][7])


class _FakeSnowPlan:
    """Stands in for microclimf_amd.snow.SnowPlan (which needs a GPU): a surface that changes per chunk, the
    same partial sums, and checks that the halo rows handed to prepare_chunk are the neighbours' rows."""

    def __init__(self, rank, world, rows_total, cols, chunks):
        self.rank, self.world, self.cols, self.chunks = rank, world, cols, chunks
        self.row0, self.rows = row_block(rank, world, rows_total)
        self.rows_total = rows_total
        self.t = 0
        self.means = []
        self.result = {"log": self.means}

    def _global(self):
        i = np.arange(self.rows_total, dtype=np.float64)[:, None]
        j = np.arange(self.cols, dtype=np.float64)[None, :]
        return 100 + np.sin(i / 7.0) * 5 + j * 0.1 + self.t

    def surface(self):
        return np.asfortranarray(self._global()[self.row0:self.row0 + self.rows])

    def surface_partial(self):
        s = self.surface()
        return float(s.sum()), float(s.size)

    def prepare_chunk(self, ch, ext, hn, hs, smean):
        g = self._global()
        want = g[self.row0 - hn:self.row0 + self.rows + hs]
        assert ext is not None and np.array_equal(np.asarray(ext), want), "halo rows are not the neighbours' rows"
        assert abs(smean - g.mean()) < 1e-12
        t = np.exp((smean - self.surface()) * 0.02)
        return float(t.sum()), float(t.size)

    def run_chunk(self, ch, tmean):
        self.means.append(tmean)
        self.t += 1


def _snowtile_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from microclimf_amd.snow import snowmodel1_chunks_tiled
        plan = _FakeSnowPlan(rank, world, 301, 9, 3)
        res = snowmodel1_chunks_tiled(plan, rank, world)
        q.put((rank, list(res["log"])))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_snow_chunk_loop_collectives_over_two_ranks():
    """snowmodel1_chunks_tiled: per chunk one point-to-point halo exchange of the snow surface and two (sum, count)
    all-reduces; both ranks end up with the raster-wide tpic mean of every chunk"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_snowtile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _FakeSnowPlan(0, 1, 301, 9, 3)
    want = []
    for _ in range(3):
        g = ref._global()
        want.append(float(np.exp((g.mean() - g) * 0.02).mean()))
        ref.t += 1
    for rank in range(world):
        assert np.allclose(got[rank], want, rtol=1e-13), rank
