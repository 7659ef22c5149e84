#!/usr/bin/env python3
"""Prints the lookup tables of mcf_device.hpp's table-based exp / log (correctly rounded, from 60-digit decimal arithmetic).

    python tools/gen_math_tables.py log     -> kLogTab[512]: 256 pairs (c_j, l_j), see flog_tab in mcf_device.hpp
    python tools/gen_math_tables.py exp     -> kExp2Tab[256]: 2^(j/256)
"""
import sys
from decimal import Decimal, getcontext

getcontext().prec = 60
JS = 106          # intervals below this index reduce towards 0.5 (i.e. 2m towards 1), the others towards 1


def rows(vals, per=4):
    out = []
    for i in range(0, len(vals), per):
        out.append("    " + ", ".join(float.hex(v) for v in vals[i:i + per]) + ("," if i + per < len(vals) else "};"))
    return "\n".join(out)


def log_table():
    vals = []
    for j in range(256):
        if j < JS:                      # M = 2m in [1 + 2j/512, 1 + (2j+2)/512): c' ~ 1/centre, stored c = 2c'
            cp = 1.0 if j == 0 else float(Decimal(1) / (Decimal(1) + Decimal(2 * j + 1) / 512))
            c = 2.0 * cp
        else:                           # m in [0.5 + j/512, 0.5 + (j+1)/512): c ~ 1/centre
            cp = 1.0 if j == 255 else float(Decimal(1) / (Decimal(1) / 2 + (Decimal(j) + Decimal(1) / 2) / 512))
            c = cp
        l = 0.0 if cp == 1.0 else float(-(Decimal(cp).ln()))          # -log of the STORED reciprocal
        vals += [c, l]
    return vals


def exp_table():
    ln2 = Decimal(2).ln()
    return [float((ln2 * j / 256).exp()) for j in range(256)]


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "log"
    if which == "log":
        print("__device__ const double kLogTab[512] = {")
        print(rows(log_table()))
    else:
        print("__device__ const double kExp2Tab[256] = {")
        print(rows(exp_table()))
