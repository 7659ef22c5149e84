"""Terrain pre-compute: known answers of the numpy oracle (oracle/terrain_oracle.py) and the
2-rank halo-exchange path over gloo with the oracle as the compute stage."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from microclimf_amd.distributed import row_block
from microclimf_amd.terrain import HALO, precompute_terrain_tiled
from oracle import terrain_oracle as TO


def synth_dtm(rows, cols, seed=3):
    i = np.arange(rows, dtype=np.float64)[:, None]
    j = np.arange(cols, dtype=np.float64)[None, :]
    rng = np.random.default_rng(seed)
    return 100 + 40 * np.sin(2 * np.pi * i / 257) * np.cos(2 * np.pi * j / 193) + 12 * np.sin(2 * np.pi * (i + j) / 61) \
        + rng.uniform(0, 1, (rows, cols))


def test_flat_terrain():
    z = np.full((30, 25), 7.0)
    t = TO.terrain(z, 1.0, 2.0)
    inner = (slice(12, 18), slice(12, 14))          # away from the zero padding of the raster edge
    assert np.all(t["hor"][inner] == 0) and np.allclose(t["svfa"][inner], 1.0)
    assert np.all(t["slope"] == 0)
    # at the edge the reference's zero padding makes the outside look 7 m lower: no horizon either
    assert t["hor"].max() == 0 and np.allclose(t["wsa"], 1.0)


def test_R_index_truncation_rule():
    # azimuth 90 deg: cos = 6e-17 so 101 - cos*s^2 rounds to 101 -> no row shift; cols shift by s^2
    sh = TO._shifts(90.0)
    assert [s[0] for s in sh] == [0] * 10 and [s[1] for s in sh] == [k * k for k in range(1, 11)]
    sh = TO._shifts(0.0)      # north: rows decrease
    assert [s[0] for s in sh] == [-k * k for k in range(1, 11)] and all(s[1] == 0 for s in sh)
    sh = TO._shifts(45.0)     # trunc toward zero of 101 -+ 0.7071 s^2
    assert sh[2][0] == int(np.trunc(101 - np.cos(np.pi / 4) * 9)) - 101 == -7
    assert sh[2][1] == int(np.trunc(101 + np.sin(np.pi / 4) * 9)) - 101 == 6


def test_step_ridge_horizon():
    z = np.zeros((40, 40))
    z[:, 30:] = 10.0                                  # a 10 m wall to the east, res 1
    h = TO.horizon(z, 90.0)
    assert h[20, 29] == pytest.approx(10.0)           # one cell away: 10/1
    assert h[20, 26] == pytest.approx(10.0 / 4)       # steps 1 (col 27: 0), 4 (col 30: 10/4)
    assert h[20, 35] == 0                             # on the plateau looking east: padding is lower


def test_tilted_plane_slope_aspect():
    i = np.arange(20, dtype=np.float64)[:, None]
    j = np.arange(20, dtype=np.float64)[None, :]
    z = 0.1 * j + 0 * i                               # rises eastwards at 10 %: faces west
    s, a = TO.slope_aspect(z, 1.0)
    assert s[5, 5] == pytest.approx(np.degrees(np.arctan(0.1))) and a[5, 5] == pytest.approx(270.0)
    assert s[0, 5] == 0 and a[5, 0] == 0              # raster edge: NA -> 0
    z = 0.2 * i + 0 * j                               # rises southwards (row index grows south): faces north
    s, a = TO.slope_aspect(z, 2.0)
    assert s[5, 5] == pytest.approx(np.degrees(np.arctan(0.1))) and a[5, 5] % 360 == pytest.approx(0.0)


def test_block_mean_and_bilinear():
    a = np.arange(23 * 12, dtype=np.float64).reshape(23, 12)
    c = TO.block_mean(a, 10)
    assert c.shape == (3, 2) and c[0, 0] == a[:10, :10].mean() and c[2, 1] == a[20:, 10:].mean()
    f = TO.bilinear_from_blocks(c, 10, 23, 12)
    assert f[4, 4] == pytest.approx(0.5 * (c[0, 0] + c[0, 0]))    # clamped towards the first centre
    assert f[0, 0] == c[0, 0]
    top = c[1, 0] * 0.35 + c[1, 1] * 0.65                          # row 22 -> t = 1.75, col 11 -> t = 0.65
    bot = c[2, 0] * 0.35 + c[2, 1] * 0.65
    assert f[22, 11] == pytest.approx(0.25 * top + 0.75 * bot)
    mid = f[9, 4]                                                  # 0.45 of the way from centre 4.5 to 14.5
    assert mid == pytest.approx(c[0, 0] * 0.55 + c[1, 0] * 0.45)


def _oracle_compute(ext, res, zref, *, agg, halo_north, halo_south, row0, rows_total, what, device):
    rows = ext.shape[0] - halo_north - halo_south
    full = np.zeros((rows_total, ext.shape[1]))
    full[row0 - halo_north:row0 + rows + halo_south] = ext
    t = TO.terrain(full, res, zref, agg)
    return {k: v[row0:row0 + rows] for k, v in t.items()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


ROWS, COLS = 310, 40


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = synth_dtm(ROWS, COLS)
        row0, rows = row_block(rank, world, ROWS)
        out = precompute_terrain_tiled(z[row0:row0 + rows], 1.0, 2.0, rank, world, row0, ROWS,
                                       compute=_oracle_compute)
        q.put((rank, row0, rows, {k: np.ascontiguousarray(v) for k, v in out.items()}))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_halo_exchange_two_ranks_gloo():
    """blocks solved behind the halo exchange equal the rows of the whole-raster result: the
    128-row halo covers the +-100-cell stencil and the wind-shelter smoothing."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = TO.terrain(synth_dtm(ROWS, COLS), 1.0, 2.0)
    assert HALO >= 125
    for rank, row0, rows, out in got:
        for k, v in out.items():
            np.testing.assert_allclose(v, want[k][row0:row0 + rows], rtol=0, atol=1e-14, err_msg=f"{k} rank {rank}")
