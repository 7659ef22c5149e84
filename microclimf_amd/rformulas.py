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
