"""Terrain pre-compute oracle (TEST INFRASTRUCTURE): numpy restatement of the R code that
builds the solver's terrain inputs in the reference's marshaller.

  horizon()       .horizon          R/internal.R:909-925  (called 24x at R/internal.R:1144)
  skyview()       svf from hor      R/internal.R:1147-1148
  windcoef()      .windcoef         R/internal.R:949-968
  windsheltera()  .windsheltera     R/internal.R:970-991
  slope_aspect()  terra::terrain(v = "slope" / "aspect"), NA -> 0 as R/internal.R:1124-1136

The first three are plain array arithmetic in R and are restated exactly, including R's
index arithmetic: `a:b` with a fractional start yields a, a+1, ... and subsetting truncates
each index toward zero, so the shift of step s is trunc(101 -+ cos/sin(azi)*s^2) - 101 in
fp64 (e.g. azimuth 90 deg: cos = 6e-17 -> 101 - 6e-17*s^2 rounds to 101 -> shift 0).

PARITY UNPINNED for the parts that live in terra (C++/GDAL, not in the reference repo):
`aggregate(fact = s, fun = "mean")` + `resample()` inside .windsheltera and `terrain()`.
They are restated from terra's documented behaviour: block means on an s x s grid anchored
at the raster's top-left (last partial block averaged over the cells it has), bilinear
interpolation between block centres clamped at the outermost centres, and Horn's (1981)
8-neighbour slope/aspect with NA on the raster edge.
"""
from __future__ import annotations

import numpy as np

HALO = 128   # rows a tile needs from each neighbour for an exact wind-shelter result


def _shifts(azimuth_deg: float):
    azi = azimuth_deg * (np.pi / 180)          # .ar(), R/internal.R:113-115
    out = []
    for step in range(1, 11):
        s2 = float(step * step)
        dr = int(np.trunc(101 - np.cos(azi) * s2)) - 101
        dc = int(np.trunc(101 + np.sin(azi) * s2)) - 101
        out.append((dr, dc, s2))
    return out


def _padded(dtm, reso):
    z = np.array(dtm, dtype=np.float64)
    z[np.isnan(z)] = 0.0                       # dtm[is.na(dtm)] <- 0
    z = z / reso
    x, y = z.shape
    p = np.zeros((x + 200, y + 200))
    p[100:100 + x, 100:100 + y] = z
    return z, p


def horizon(dtm, azimuth_deg: float, reso: float = 1.0):
    """tan(horizon angle) in one direction, R/internal.R:909-925."""
    z, p = _padded(dtm, reso)
    x, y = z.shape
    hor = np.zeros_like(z)
    for dr, dc, s2 in _shifts(azimuth_deg):
        shifted = p[100 + dr:100 + dr + x, 100 + dc:100 + dc + y]
        hor = np.maximum(hor, (shifted - z) / s2)
    return hor


def horizons24(dtm, reso: float = 1.0):
    return np.stack([horizon(dtm, 15.0 * i, reso) for i in range(24)], axis=2)


def skyview(hor):
    """svfa = 0.5*cos(2*tan(mean(atan(hor)))) + 0.5, R/internal.R:1147-1148 (sic)."""
    msl = np.tan(np.mean(np.arctan(hor), axis=2))
    return 0.5 * np.cos(2 * msl) + 0.5


def windcoef(dsm, direction_deg: float, hgt: float = 1.0, reso: float = 1.0):
    """Wind-shelter coefficient in one direction, R/internal.R:949-968."""
    z, p = _padded(dsm, reso)
    x, y = z.shape
    h = hgt / reso
    hor = np.zeros_like(z)
    for dr, dc, s2 in _shifts(direction_deg):
        shifted = p[100 + dr:100 + dr + x, 100 + dc:100 + dc + y]
        hor = np.maximum(hor, (shifted - z) / s2)
        hor = np.where(hor < (h / s2), 0.0, hor)
    return 1 - np.arctan(0.17 * 100 * hor) / 1.65


def block_mean(a, s: int):
    """terra::aggregate(fact = s, fun = "mean"): s x s blocks from the top-left corner."""
    x, y = a.shape
    nx, ny = -(-x // s), -(-y // s)
    out = np.empty((nx, ny))
    for i in range(nx):
        for j in range(ny):
            out[i, j] = a[i * s:(i + 1) * s, j * s:(j + 1) * s].mean()
    return out


def bilinear_from_blocks(c, s: int, x: int, y: int):
    """terra::resample(coarse, fine) (bilinear): fine cell centres against block centres at
    s*I + (s-1)/2, clamped to the outermost centres."""
    def axis(n_f, n_c):
        t = (np.arange(n_f) - (s - 1) / 2) / s
        i0 = np.floor(t).astype(int)
        w = t - i0
        lo = np.clip(i0, 0, n_c - 1)
        hi = np.clip(i0 + 1, 0, n_c - 1)
        return lo, hi, w
    r0, r1, wr = axis(x, c.shape[0])
    c0, c1, wc = axis(y, c.shape[1])
    top = c[r0][:, c0] * (1 - wc)[None, :] + c[r0][:, c1] * wc[None, :]
    bot = c[r1][:, c0] * (1 - wc)[None, :] + c[r1][:, c1] * wc[None, :]
    return top * (1 - wr)[:, None] + bot * wr[:, None]


def windsheltera(dtm, whgt: float, s: int = 10, reso: float = 1.0):
    """8-direction wind-shelter array, R/internal.R:970-991."""
    x, y = np.shape(dtm)
    a = np.empty((x, y, 16))
    for i in range(16):
        wc = windcoef(dtm, i * 360.0 / 16, whgt, reso)
        a[:, :, i] = bilinear_from_blocks(block_mean(wc, s), s, x, y)
    a2 = np.empty((x, y, 8))
    for i in range(1, 9):                       # R's 1-based loop
        if i == 1:
            m = 0.5 * a[:, :, 0] + 0.25 * a[:, :, 1] + 0.25 * a[:, :, 15]
        else:
            m = 0.5 * a[:, :, i * 2 - 2] + 0.25 * a[:, :, i * 2 - 1] + 0.25 * a[:, :, i * 2 - 3]
        a2[:, :, i - 1] = m
    return a2


def slope_aspect(dtm, reso: float = 1.0, aspect_na: float = 0.0):
    """Horn 8-neighbour slope and aspect in degrees (aspect clockwise from north, downslope
    direction, 90 where flat); raster-edge cells and cells with an NA neighbour are NA in
    terra and become 0 in the marshaller (R/internal.R:1132-1133)."""
    z = np.array(dtm, dtype=np.float64)
    x, y = z.shape
    slope = np.zeros((x, y))
    aspect = np.full((x, y), float(aspect_na))
    zn = z[:-2, :]; zs = z[2:, :]; zc = z[1:-1, :]
    # row index grows southwards, column index eastwards
    nw, n_, ne = zn[:, :-2], zn[:, 1:-1], zn[:, 2:]
    w_, e_ = zc[:, :-2], zc[:, 2:]
    sw, s_, se = zs[:, :-2], zs[:, 1:-1], zs[:, 2:]
    dzdx = ((ne + 2 * e_ + se) - (nw + 2 * w_ + sw)) / (8 * reso)          # eastward
    dzdy = ((nw + 2 * n_ + ne) - (sw + 2 * s_ + se)) / (8 * reso)          # northward
    sl = np.degrees(np.arctan(np.sqrt(dzdx ** 2 + dzdy ** 2)))
    asp = np.degrees(np.arctan2(-dzdx, -dzdy)) % 360.0
    asp = np.where((dzdx == 0) & (dzdy == 0), 90.0, asp)
    bad = np.isnan(sl)
    slope[1:-1, 1:-1] = np.where(bad, 0.0, sl)
    aspect[1:-1, 1:-1] = np.where(bad, float(aspect_na), asp)
    return slope, aspect


def terrain(dtm, reso: float, zref: float, s: int = 10):
    hor = horizons24(dtm, reso)
    slope, aspect = slope_aspect(dtm, reso)
    return {"slope": slope, "aspect": aspect, "hor": hor, "svfa": skyview(hor),
            "wsa": windsheltera(dtm, zref, s, reso)}
