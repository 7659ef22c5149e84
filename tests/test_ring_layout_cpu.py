"""The tiled output ring's addressing (include/mcf.h mcf_ring_layout / mcf_ring_index; mcf_kernels.h RingView): every
(cell, step) of a slot has its own place, a tile-day block is whole 128-byte lines, a wave's 64 lanes are 64 consecutive
doubles.  Host-only: mcf_ring_index needs no device."""
import ctypes as C

import numpy as np
import pytest

from microclimf_amd import _abi


def _layout(cells, cpb, days, nvars=10):
    blk = ((cpb * 24 + 63) // 64) * 64
    lay = _abi.RingLayout()
    lay.tiled, lay.cells_per_tile, lay.block_doubles, lay.slot_days = 1, cpb, blk, days
    lay.cells = cells
    lay.day_stride = nvars * blk
    lay.tile_stride = days * nvars * blk
    return lay, blk


@pytest.mark.parametrize("cpb", [16, 21, 32, 42])
def test_every_cell_step_has_its_own_place_inside_its_tile_day_block(cpb):
    lib = _abi.load()
    cells, days = 3 * cpb + 5, 2          # a ragged last tile
    lay, blk = _layout(cells, cpb, days)
    idx = np.array([[lib.mcf_ring_index(C.byref(lay), c, k) for k in range(days * 24)] for c in range(cells)])
    assert (idx >= 0).all() and np.unique(idx).size == idx.size
    tile = np.arange(cells)[:, None] // cpb
    day = np.arange(days * 24)[None, :] // 24
    base = tile * lay.tile_stride + day * lay.day_stride
    assert ((idx - base) >= 0).all() and ((idx - base) < blk).all()       # inside the (tile, day, variable 0) block
    assert blk % 16 == 0 and lay.tile_stride % 16 == 0                     # blocks are whole 128-byte lines
    assert lib.mcf_ring_index(C.byref(lay), cells, 0) == -1 and lib.mcf_ring_index(C.byref(lay), 0, days * 24) == -1


def test_the_21_cell_block_is_the_solver_lane_order():
    """Wave w of k_solve<21> holds hours 3w .. 3w+2: lanes 0-47 = 3 hours x cells 0-15, lanes 48-62 = 3 hours x cells
    16-20, lane 63 padding (mcf_kernels.hip solve_tile)."""
    lib = _abi.load()
    lay, blk = _layout(21, 21, 1)
    assert blk == 512
    seen = set()
    for h in range(24):
        w, hh = divmod(h, 3)
        for cell in range(21):
            lane = 16 * hh + cell if cell < 16 else 48 + 5 * hh + (cell - 16)
            got = lib.mcf_ring_index(C.byref(lay), cell, h)
            assert got == 64 * w + lane
            seen.add(got)
    assert seen == set(range(512)) - {64 * w + 63 for w in range(8)}


def test_the_linear_layout_is_the_reference_layout():
    lib = _abi.load()
    lay = _abi.RingLayout()
    lay.tiled, lay.cells, lay.slot_days = 0, 50, 3
    assert lib.mcf_ring_index(C.byref(lay), 7, 5) == 7 + 50 * 5
