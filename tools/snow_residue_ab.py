#!/usr/bin/env python3
"""do device and oracle part ways at the snow loop's ill-conditioned hand-over because hipcc contracts a*b+c
into FMAs in mcf_snow.hip (the oracle is built -ffp-contract=off)?  Runs the full-year chunk loop of
tests/test_snow_gpu.py::test_full_year_chunk_loop... on the library named by MCF_LIB (default: the shipped one) and prints the
hand-overs at which the `sdepcp > 0` gate differs from the oracle's.  A/B: the shipped build vs one whose mcf_snow.o is built
with -ffp-contract=off (UNIT=mcf_snow tools/build_variant.sh snow_nofma -ffp-contract=off)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from microclimf_amd.snow import SnowPlan           # noqa: E402
from oracle import oracle as O                     # noqa: E402
from oracle import snowdriver_oracle as SD         # noqa: E402
from test_snow_gpu import _driver_case             # noqa: E402

O.load()
sw, dtm = _driver_case(50, 50, 8760)
args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
hand = []
with SnowPlan(*args) as p:
    for ch in range(p.chunks):
        s, n = p.surface_partial()
        ts, tn = p.prepare_chunk(ch, None, 0, 0, s / n)
        p.run_chunk(ch, ts / tn)
        hand.append(p.handover())
    got = {k: v.copy() for k, v in p.result.items()}
events, first_diff = [], []


def hook(ch, o):
    d = hand[ch]
    with np.errstate(invalid="ignore"):
        differs = (o > 0) != (d > 0)
        bits = (o.view(np.uint64) != d.view(np.uint64)) & ~(np.isnan(o) & np.isnan(d))
    if bits.any() and not first_diff:
        first_diff.append((ch, int(bits.sum()), float(np.nanmax(np.abs(o - d)))))
    if differs.any():
        events.append((ch, int(differs.sum())))
    return np.where(differs, d, o)


want = SD.snowmodel1_chunks(*args, handover=hook)
worst = max(float(np.nanmax(np.abs(got[k] - want[k]) / (1 + np.abs(want[k])))) for k in want)
print("gate events (chunk, cells):", events)
print("first hand-over whose bits differ at all (chunk, cells, max |diff|):", first_diff)
print("max scaled difference over the year with the gate aligned: %.3e" % worst)
