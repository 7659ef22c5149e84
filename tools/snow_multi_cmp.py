"""mcf_snowmodel1 against mcf_snowmodel1_multi on ONE device (host-side overhead of the row-block driver): python tools/snow_multi_cmp.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from microclimf_amd import synthetic
from microclimf_amd.snow import snowmodel1_chunks
R = C = 512; T = 120
sw = synthetic.snow_workload(R, C, T, cold=3.0, zref=3.5)
_, _, dtm = synthetic.rasters(R, C)
dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
args = (sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
for label, kw in (("single", {}), ("multi 1 block", dict(devices=[0], n_blocks=1)), ("multi 2 blocks", dict(devices=[0], n_blocks=2)), ("multi 2 threads", dict(devices=[0, 0], n_blocks=2))):
    snowmodel1_chunks(*args, **kw)
    t = time.perf_counter(); snowmodel1_chunks(*args, **kw); print(label, "%.3f s" % (time.perf_counter() - t))
