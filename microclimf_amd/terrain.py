"""Host side of the on-device terrain pre-compute (include/mcf.h, mcf_precompute_terrain).

`precompute_terrain(dtm, res, zref)` builds what the reference's marshaller builds in R
before it calls the solver (R/internal.R:1124-1154): slope, aspect, hor[,,24], svfa and
wsa[,,8].  `precompute_terrain_tiled` does the same for one rank's row block of a larger
raster: the +-100-cell stencil (and the wind-shelter smoothing) needs HALO rows from the
neighbouring ranks, exchanged point-to-point with torch.distributed (RCCL send/recv over
xGMI when the backend is nccl; gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi

HALO = 128          # >= 100 + 2.5 * s for s = 10
WHAT = ("slope", "aspect", "hor", "svfa", "wsa")


def precompute_terrain(dtm, res: float, zref: float, *, agg: int = 10, halo_north: int = 0,
                       halo_south: int = 0, row0: int = 0, rows_total: int = 0, what=WHAT, device: int = 0,
                       devices=None, n_blocks: int = 0):
    """dtm: [(halo_north + rows + halo_south), cols] elevations (NaN = NA).
    `devices` (a list of HIP ordinals, [] = all visible) / `n_blocks`: the WHOLE raster in row blocks over several devices
    from this one process (include/mcf.h mcf_precompute_terrain_multi), same values."""
    lib = _abi.load()
    z = np.asfortranarray(np.asarray(dtm, dtype=np.float64))
    rb, cols = z.shape
    rows = rb - halo_north - halo_south
    tin = _abi.TerrainIn()
    tin.rows, tin.cols = rows, cols
    tin.halo_north, tin.halo_south = halo_north, halo_south
    tin.dtm = z.ctypes.data_as(_abi.c_double_p)
    tin.res, tin.zref, tin.agg = float(res), float(zref), int(agg)
    tin.row0, tin.rows_total = int(row0), int(rows_total)
    shapes = {"slope": (rows, cols), "aspect": (rows, cols), "hor": (rows, cols, 24),
              "svfa": (rows, cols), "wsa": (rows, cols, 8)}
    tout = _abi.TerrainOut()
    res_arrays = {}
    for k in WHAT:
        if k in what:
            a = np.empty(shapes[k], dtype=np.float64, order="F")
            res_arrays[k] = a
            setattr(tout, k, a.ctypes.data_as(_abi.c_double_p))
        else:
            setattr(tout, k, None)
    if devices is not None or n_blocks:
        mu = _abi.Multi()
        devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
        mu.n_devices, mu.devices, mu.n_blocks = int(devs.size), devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
        _abi.check(lib.mcf_precompute_terrain_multi(C.byref(tin), C.byref(tout), C.byref(mu)))
    else:
        _abi.check(lib.mcf_precompute_terrain(C.byref(tin), C.byref(tout), device))
    return res_arrays


def snow_terrain(dtm, res: float, zref: float, *, agg: int | None = None, mask=None, device: int = 0) -> dict:
    """The terrain block of `.snowmodelq1` (R/internal.R:2690-2706): as the marshaller's, but terra's NA aspects (raster
    edge, NA neighbour) become 180 and slope / aspect are masked by the dtm."""
    z = np.asarray(dtm, dtype=np.float64)
    t = precompute_terrain(z, res, zref, agg=(10 if res <= 100 else 1) if agg is None else agg, device=device)
    pad = np.pad(z, 1, constant_values=np.nan)
    na = np.zeros(z.shape, dtype=bool)
    for dr in (0, 1, 2):
        for dc in (0, 1, 2):
            if (dr, dc) != (1, 1):
                na |= np.isnan(pad[dr:dr + z.shape[0], dc:dc + z.shape[1]])
    hole = np.isnan(z) if mask is None else np.isnan(np.asarray(mask, dtype=np.float64))     # `mask(slope, dtm)`
    return {"slope": np.where(hole, np.nan, t["slope"]), "aspect": np.where(hole, np.nan, np.where(na, 180.0, t["aspect"])),
            "hor": t["hor"], "skyview": t["svfa"], "wsa": t["wsa"]}


def exchange_halo(block: np.ndarray, rank: int, world: int, halo: int = HALO, device=None):
    """Returns (extended block, halo_north, halo_south): up to `halo` rows from each neighbouring
    rank's block (row blocks are ordered north to south by rank).  Point-to-point
    isend/irecv between ring neighbours only; nothing is exchanged across the raster's edge."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return block, 0, 0
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rows, cols = block.shape
    # every rank learns its neighbours' row counts (they cap the halo actually available)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = rows
    dist.all_reduce(counts)
    counts = [int(v) for v in counts.tolist()]
    block = np.asarray(block)
    # only the boundary rows travel: row slices as row-major tensors (a raster block arrives column-major, so this is the one
    # transposing copy, of `halo` rows, not of the block)
    ops, recv_n, recv_s = [], None, None
    if rank > 0:
        hn = min(halo, counts[rank - 1])
        recv_n = torch.empty((hn, cols), dtype=torch.float64, device=device)
        ops.append(dist.P2POp(dist.irecv, recv_n, rank - 1))
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(block[:min(halo, rows)])).to(device), rank - 1))
    if rank < world - 1:
        hs = min(halo, counts[rank + 1])
        recv_s = torch.empty((hs, cols), dtype=torch.float64, device=device)
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(block[max(rows - halo, 0):])).to(device), rank + 1))
        ops.append(dist.P2POp(dist.irecv, recv_s, rank + 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    hn = 0 if recv_n is None else recv_n.shape[0]
    hs = 0 if recv_s is None else recv_s.shape[0]
    return assemble_halo(block, None if recv_n is None else recv_n.cpu().numpy(), None if recv_s is None else recv_s.cpu().numpy()), hn, hs


def assemble_halo(block, north=None, south=None, out=None) -> np.ndarray:
    """[north; block; south] as ONE column-major array (what mcf_precompute_terrain / mcf_snowplan_prepare_chunk take): the
    block is copied column by column, never transposed.  `out`: an array of that shape to reuse."""
    hn = 0 if north is None else north.shape[0]
    hs = 0 if south is None else south.shape[0]
    rows, cols = block.shape
    ext = np.empty((hn + rows + hs, cols), dtype=np.float64, order="F") if out is None else out
    if hn:
        ext[:hn] = north
    ext[hn:hn + rows] = block
    if hs:
        ext[hn + rows:] = south
    return ext


def precompute_terrain_tiled(block, res, zref, rank, world, row0, rows_total, *, agg=10, what=WHAT,
                             device=0, compute=precompute_terrain):
    """One rank's share of the terrain pre-compute: halo exchange, then the local kernels.
    `compute` is injectable so that the CPU tests can put the numpy oracle behind the same
    exchange."""
    ext, hn, hs = exchange_halo(np.asarray(block, dtype=np.float64), rank, world)
    return compute(ext, res, zref, agg=agg, halo_north=hn, halo_south=hs, row0=row0,
                   rows_total=rows_total, what=what, device=device)


def flowaccCpp(dm) -> np.ndarray:
    """Drop-in for flowaccCpp (src/microclimfCpp.cpp:5368-5408): flow accumulation of an elevation matrix (host)."""
    lib = _abi.load()
    z = np.asfortranarray(np.asarray(dm, dtype=np.float64))
    fa = np.empty(z.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_flowacc(z.shape[0], z.shape[1], z.ctypes.data_as(_abi.c_double_p), fa.ctypes.data_as(_abi.c_double_p)))
    return fa


def topidx(dtm, res) -> np.ndarray:
    """`.topidx(dtm)` (R/internal.R:861-874): the topographic wetness index the solver takes as soilc$twi (host)."""
    lib = _abi.load()
    z = np.asfortranarray(np.asarray(dtm, dtype=np.float64))
    xres, yres = (res, res) if np.isscalar(res) else res
    twi = np.empty(z.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_topidx(z.shape[0], z.shape[1], z.ctypes.data_as(_abi.c_double_p), float(xres), float(yres),
                              twi.ctypes.data_as(_abi.c_double_p)))
    return twi
