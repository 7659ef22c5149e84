"""Diagnostic (GPU box): where the device chunk loop of `.snowmodel1` and its oracle part over a full year.
The loop hands the pack depth over as `(asc + cdsnow + dsnow2)[last]` (R/internal.R:2607): when the pack has
melted this is a rounding residue (0 or +-1e-17 m) and `sdepcp > 0` (cpp:4337) then decides whether the model
runs on the next chunk - ill-conditioned in the reference itself; see DESIGN.md section 8."""
import sys, numpy as np
sys.path.insert(0, ".")
from microclimf_amd import synthetic
from microclimf_amd.snow import snowmodel1_chunks
from oracle import snowdriver_oracle as SD
sw = synthetic.snow_workload(50, 50, 8760, cold=3.0, zref=3.5)
_, _, dtm = synthetic.rasters(50, 50)
dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
r = snowmodel1_chunks(*args); w = SD.snowmodel1_chunks(*args)
with np.errstate(invalid="ignore"):
    bad = np.zeros((50, 50, 8760), bool)
    for k in w:
        d = np.abs(r[k] - w[k]) / (1 + np.abs(w[k]))
        bad |= np.nan_to_num(d) > 1e-6
    cells = np.argwhere(bad.any(axis=2))
    print("cells diverged:", len(cells), "of", int((~np.isnan(dtm)).sum()), "; cell-steps diverged:", int(bad.sum()))
    for i, j in cells[:20]:
        t0 = int(np.argmax(bad[i, j]))
        ch = t0 // 120
        prev = ch * 120 - 1
        tot_g = r["totalSWE"][i, j, prev] / r["snowden"][i, j, prev]
        tot_o = w["totalSWE"][i, j, prev] / w["snowden"][i, j, prev]
        print(f"cell ({i},{j}) first diverges at step {t0} (chunk {ch}, offset {t0 % 120}); pack depth handed over: gpu {tot_g:.3e} oracle {tot_o:.3e}")
