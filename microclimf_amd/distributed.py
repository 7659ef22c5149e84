"""Row-block partition of a raster over ranks and the solver's one collective.

Cells are independent inside the grid solver (no neighbour reads in
src/microclimfCpp.cpp:2180-2323), so each GPU owns a contiguous block of raster rows.
The single data-path exchange is the raster-wide mean of log(twi)/tfact over non-NA
cells (cpp:993-1004): every rank contributes its (sum, count) — computed on the device
by mcf_plan_twi_partial — to one all-reduce (RCCL when the backend is nccl, gloo in the
CPU tests) and installs the global mean with mcf_plan_set_twi_mean.
"""
from __future__ import annotations


def row_block(rank: int, world: int, rows_total: int):
    """(row0, rows) of `rank`'s block when `rows_total` rows are dealt to `world` ranks in
    contiguous blocks whose sizes differ by at most one row."""
    base, extra = divmod(rows_total, world)
    rows = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, rows


def balanced_row_blocks(valid_per_row, world: int, multiple: int = 10):
    """[(row0, rows)] per rank: contiguous row blocks holding about the same number of VALID cells (non-NA `hgt`),
    the solver's unit of work — NA cells cost nothing, so equal row counts leave ranks idle on rasters with sea or
    no-data areas.  Interior boundaries fall on multiples of `multiple` rows (10 keeps the wind-shelter
    pre-compute's 10 x 10 aggregation blocks inside one tile, R/internal.R:980); every rank gets at least one such
    group while there are enough rows."""
    import numpy as np
    w = np.asarray(valid_per_row, dtype=np.float64)
    rows_total = len(w)
    if world < 1 or multiple < 1:
        raise ValueError("world and multiple must be >= 1")
    ngroups = -(-rows_total // multiple)
    if ngroups < world:                       # fewer groups than ranks: fall back to single rows
        if multiple > 1:
            return balanced_row_blocks(w, world, 1)
        return [row_block(r, world, rows_total) for r in range(world)]
    gw = np.add.reduceat(w, np.arange(0, rows_total, multiple))
    cum = np.concatenate([[0.0], np.cumsum(gw)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        if total > 0:
            g = int(np.searchsorted(cum, total * r / world, side="left"))
            # the nearer of the two neighbouring group boundaries
            if g > 0 and abs(cum[g - 1] - total * r / world) <= abs(cum[min(g, ngroups)] - total * r / world):
                g -= 1
        else:
            g = ngroups * r // world
        g = min(max(g, cuts[-1] + 1), ngroups - (world - r))      # strictly increasing, leave a group for each later rank
        cuts.append(g)
    cuts.append(ngroups)
    out = []
    for r in range(world):
        row0 = cuts[r] * multiple
        row1 = min(cuts[r + 1] * multiple, rows_total)
        out.append((row0, row1 - row0))
    return out


def allreduce_twi_mean(local_sum: float, local_count: float, device=None) -> float:
    """Global mean from per-rank partial (sum, count); a no-op without an initialised
    process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_sum / local_count
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(local_sum), float(local_count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0].item()) / float(t[1].item())


def allreduce_max(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_sum(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_apply3(local, local_count, fun_name: str, device=None):
    """Raster-wide applycpp3 (src/microclimfCpp.cpp:5553-5588) from per-rank row-block partials:
    `local` is the rank's result of applycpp3(block, "sum" | "max" | "min") and `local_count` its
    non-NA counts; "mean" is formed from the all-reduced sum and count (pass the local SUM).  One
    all-reduce of a [tsteps] (or [2, tsteps]) fp64 tensor."""
    import numpy as np
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.float64)
    cnt = np.asarray(local_count, dtype=np.float64)
    active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if fun_name in ("mean", "sum"):
        t = np.stack([local, cnt])
        if active:
            if device is None:
                device = "cuda" if dist.get_backend() == "nccl" else "cpu"
            tt = torch.from_numpy(t).to(device)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            t = tt.cpu().numpy()
        if fun_name == "sum":
            return t[0]
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(t[1] > 0, t[0] / t[1], np.nan)
    if fun_name not in ("max", "min"):
        raise ValueError("Unknown function name")
    if not active:
        return local
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    tt = torch.from_numpy(local.copy()).to(device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX if fun_name == "max" else dist.ReduceOp.MIN)
    return tt.cpu().numpy()
