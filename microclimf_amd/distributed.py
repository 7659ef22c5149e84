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
