import numpy as np, sys
sys.path.insert(0, '.')
from microclimf_amd import snow as S, synthetic
rows, cols, ndays = 22, 13, 20
T = ndays * 24
for cold in (3.0, 0.0, -3.0, -6.0):
    for doy in (20, 90, 120):
        sw = synthetic.snow_workload(rows, cols, T, cold=cold, zref=3.5, start_doy=doy)
        _, _, dtm = synthetic.rasters(rows, cols)
        dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
        with S.SnowPlan(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02, keep_results=False) as sp:
            sd, nd = [], []
            for ch in range(sp.chunks):
                ss, sn = sp.surface_partial()
                ts, tn = sp.prepare_chunk(ch, None, 0, 0, ss / sn)
                sp.run_chunk(ch, ts / tn)
                mx, _ = sp.apply3(ch, "max"); mn, _ = sp.apply3(ch, "min")
                d = S.snowdaysfun(mx, mn)
                sd += list(d["snowdays"]); nd += list(d["nosnowdays"])
        print(cold, doy, "snow", "".join(map(str, sd)), "nosnow", "".join(map(str, nd)), flush=True)
