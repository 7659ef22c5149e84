"""Max scaled |HIP - oracle| over a few small workloads (python tools/parity_probe.py): a quick number for A/B builds."""
import numpy as np, sys
sys.path.insert(0, '.')
from microclimf_amd import synthetic
from microclimf_amd.api import runmicro1Cpp, runmicro2Cpp
from oracle import oracle as O
O.load()
for reqhgt, af in ((0.05, False), (2.5, False), (0.0, False), (0.05, True)):
    a = synthetic.workload(24, 20, 96, reqhgt=reqhgt, variety=True, start_doy=170, array_forcing=af)
    got = (runmicro2Cpp if af else runmicro1Cpp)(*[a[k] for k in ('obstime','climdata','pointm','vegp','soilc','reqhgt','zref','lat','lon','Sminp','Smaxp','tfact','complete','mat','out')])
    want = O.run_grid(**a, array_forcing=af)
    worst = 0
    for k, w in want.items():
        g = got[k]
        assert np.array_equal(np.isnan(g), np.isnan(w)), k
        if np.isfinite(w).any():
            worst = max(worst, float(np.nanmax(np.abs(g - w) / (1 + np.abs(w)))))
    print(f"reqhgt {reqhgt} af {af}: max scaled diff {worst:.3e}")
