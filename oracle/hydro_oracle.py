"""TEST INFRASTRUCTURE — CPU restatement of the reference's flow accumulation and topographic wetness index, used
only by tests/ to check microclimf_amd's mcf_flowacc / mcf_topidx.  Pure-Python loops: small rasters only.

  flowdir / flowacc : src/microclimfCpp.cpp:5326-5366 / 5368-5408, statement by statement (padded matrix, index
                      enumeration, (value, index) sort with std::greater, `size() - 1` iterations)
  topidx            : R/internal.R:861-874 (`.topidx`) with terra::terrain(unit = "radians") restated as Horn's
                      8-neighbour slope, NA on the raster edge and beside NA cells (terra's behaviour; parity with terra
                      itself is unpinned, as for oracle/terrain_oracle.py)
"""
import math

import numpy as np

NA_INTEGER_AS_DOUBLE = -2147483648.0


def flowdir(md):
    md = np.asarray(md, dtype=np.float64)
    nrow, ncol = md.shape
    md2 = np.full((nrow + 2, ncol + 2), np.nan)
    md2[1:-1, 1:-1] = md
    fd = np.zeros((nrow, ncol), dtype=np.int64)
    for i in range(nrow):
        for j in range(ncol):
            if math.isnan(md[i, j]):
                continue
            minval = 9999.99
            indx = 1
            for jj in range(3):
                for ii in range(3):
                    val2 = md2[i + ii, j + jj]
                    if not math.isnan(val2) and val2 < minval:
                        minval = val2
                        fd[i, j] = indx
                    indx += 1
    return fd


def flowacc(dm):
    dm = np.asarray(dm, dtype=np.float64)
    nrow, ncol = dm.shape
    fd = flowdir(dm)
    fa = np.where(np.isnan(dm), NA_INTEGER_AS_DOUBLE, 1.0)
    order = [(dm[i, j], i * ncol + j) for i in range(nrow) for j in range(ncol) if not math.isnan(dm[i, j])]
    order.sort(reverse=True)                       # std::greater on (value, index) pairs
    for k in range(len(order) - 1):
        index = order[k][1]
        y, x = index // ncol, index % ncol
        f = int(fd[y, x])
        if f < 1 or f > 9:
            continue
        y2 = y + (f - 1) % 3 - 1
        x2 = x + (f - 1) // 3 - 1
        if 0 <= x2 < ncol and 0 <= y2 < nrow and fa[y2, x2] != NA_INTEGER_AS_DOUBLE:
            fa[y2, x2] += fa[y, x]
    return fa


def slope_radians(dtm, xres, yres):
    z = np.asarray(dtm, dtype=np.float64)
    out = np.full(z.shape, np.nan)
    zn, zc, zs = z[:-2, :], z[1:-1, :], z[2:, :]
    nw, n_, ne = zn[:, :-2], zn[:, 1:-1], zn[:, 2:]
    w_, e_ = zc[:, :-2], zc[:, 2:]
    sw, s_, se = zs[:, :-2], zs[:, 1:-1], zs[:, 2:]
    dzdx = ((ne + 2 * e_ + se) - (nw + 2 * w_ + sw)) / (8 * xres)
    dzdy = ((nw + 2 * n_ + ne) - (sw + 2 * s_ + se)) / (8 * yres)
    with np.errstate(invalid="ignore"):
        sl = np.arctan(np.sqrt(dzdx ** 2 + dzdy ** 2))
    out[1:-1, 1:-1] = np.where(np.isnan(zc[:, 1:-1]), np.nan, sl)
    return out


def topidx(dtm, xres, yres):
    dtm = np.asarray(dtm, dtype=np.float64)
    minslope = math.atan(0.02 / ((xres + yres) / 2))
    B = slope_radians(dtm, xres, yres)
    with np.errstate(invalid="ignore"):
        B[B < minslope] = minslope
    if np.isfinite(B).any():
        B[np.isnan(B)] = np.nanmedian(B)
    a = flowacc(dtm) + 1
    a = a * xres * yres
    a[a < 1] = 1
    with np.errstate(invalid="ignore", divide="ignore"):
        tpx = a / np.tan(B)
    return np.where(np.isnan(dtm), np.nan, tpx)
