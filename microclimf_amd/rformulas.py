"""R-side formulas of the reference that define solver inputs: `.satvap` and `.dewpoint`
(R/internal.R:501-521).  They differ from the C++ satvapCpp / dewpointCpp (ice branch at tc < 0 instead of tc <= 0,
other constants), and the front ends apply THESE to the weather before the solver is called."""
import numpy as np


def satvap_R(tc):
    tc = np.asarray(tc, dtype=np.float64)
    es = 0.61078 * np.exp(17.27 * tc / (tc + 237.3))
    ei = 0.61078 * np.exp(21.875 * tc / (tc + 265.5))
    return np.where(tc < 0, ei, es)


def dewpoint_R(ea, tc):
    ea = np.asarray(ea, dtype=np.float64)
    e0 = 611.2 / 1000
    L = (2.501e6) - (2340 * tc)
    it = 1 / 273.15 - (461.5 / L) * np.log(ea / e0)
    tdew = 1 / it - 273.15
    e0 = 610.78 / 1000
    L = 2.834e6
    it = 1 / 273.15 - (461.5 / L) * np.log(ea / e0)
    tfrost = 1 / it - 273.15
    return np.where(tdew < 0, tfrost, tdew)


def lapserate_R(tc, ea, pk):
    """`.lapserate` (R/internal.R:545-550): moist adiabatic lapse rate (K / m)"""
    rv = 0.622 * ea / (pk - ea)
    return 9.8076 * (1 + (2501000 * rv) / (287 * (tc + 273.15))) / (1003.5 + (0.622 * 2501000 ** 2 * rv) / (287 * (tc + 273.15) ** 2))


def upsample_coarse(a, rowpos, colpos):
    """`resample(.rast(a, dtmc), dtm)` for [crows, ccols(, T)] held on the host: bilinear between the four neighbouring
    coarse cell centres, positions as `api.coarse_positions` gives them (edge replication) — the same taps the solver's
    coarse array forcing uses on the device (mcf_device.hpp CoarseTap)."""
    a = np.asarray(a, dtype=np.float64)
    two = a.ndim == 2
    if two:
        a = a[:, :, None]
    cr, cc = a.shape[:2]
    fr, fc = np.floor(rowpos), np.floor(colpos)
    r0, c0 = fr.astype(np.int64), fc.astype(np.int64)
    r1, c1 = np.minimum(r0 + 1, cr - 1), np.minimum(c0 + 1, cc - 1)
    wy, wx = (rowpos - fr)[:, None, None], (colpos - fc)[None, :, None]
    top = (1.0 - wx) * a[np.ix_(r0, c0)] + wx * a[np.ix_(r0, c1)]
    bot = (1.0 - wx) * a[np.ix_(r1, c0)] + wx * a[np.ix_(r1, c1)]
    out = np.asfortranarray((1.0 - wy) * top + wy * bot)
    return out[:, :, 0] if two else out
