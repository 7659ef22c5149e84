"""One pass over the kernels outside k_solve (terrain pre-compute, snow branch, output sinks) at bench-like sizes,
for `rocprofv3 --kernel-trace --stats` (tools/profile_aux.sh): their device times end up in
profiles/<tag>_aux_kernel_stats.csv."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import synthetic  # noqa: E402
from microclimf_amd.api import Plan  # noqa: E402
from microclimf_amd.snow import applycpp3, gridmicrosnow1, gridmodelsnow1, snowmodel1_chunks  # noqa: E402
from microclimf_amd.terrain import precompute_terrain  # noqa: E402

R = C = 1024
# terrain pre-compute (f-1)
_, _, dtm = synthetic.rasters(R, C)
precompute_terrain(dtm, 1.0, 2.0)
# solver into a ring slot + packed sink (f-3)
w = synthetic.workload(R, C, 120, reqhgt=0.05)
with Plan(w["obstime"], w["climdata"], w["pointm"], w["vegp"], w["soilc"], w["reqhgt"], w["zref"], w["lat"], w["lon"],
          w["Sminp"], w["Smaxp"], w["tfact"], True, w["mat"], w["out"], ring_days=5) as p:
    p.run_days(0, 5)
    p.sync()
    for var in ("Tz", "relhum"):
        p.fetch_packed(0, var, 0, 120)
# snow branch (f-4): recurrence, chunk loop (terrain refresh + tpi + redistribution), microclimate, applycpp3
sw = synthetic.snow_workload(R, C, 240, cold=3.0, zref=3.5)
smod = gridmodelsnow1(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"])
dtm = np.where(np.isnan(sw["vegp"]["hgt"]), np.nan, dtm)
snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"], dtm, 1.0, 0.02)
sw2 = synthetic.snow_workload(512, 512, 240, cold=3.0, zref=3.5)
sm2 = gridmodelsnow1(sw2["obstime"], sw2["climdata"], sw2["pointm"], sw2["vegp"], sw2["other"], sw2["snowenv"])
snowm, micro = synthetic.microsnow_inputs(sw2, sm2)
gridmicrosnow1(0.05, sw2["obstime"], sw2["climdata"], snowm, micro, sw2["vegp"], sw2["other"], 3.0, [1] * 10)
for fun in ("max", "min"):
    applycpp3(snowm["totalSWE"], fun)
print("aux workload done")
