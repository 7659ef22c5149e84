"""Random small configurations of mcf_runmicrosnow1 (one call, possibly over row blocks) and — round 5 — of its layered-vegetation
form and of mcf_runmicrosnow2 (array weather) against the reference's orchestration on the host with the library's one-shot entries
behind it (tests/test_snowrun_gpu.py / test_snowrun2_gpu.py `_orchestrate`): rasters, lengths, heights, seasons, deep packs beside
bare rows, block counts, layer counts.  python tools/fuzz_snowrun.py [--n 40 --seed 1]"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from microclimf_amd import snow as S, synthetic  # noqa: E402
from microclimf_amd.api import runmicro1Cpp, runmicro2Cpp, runmicro3Cpp  # noqa: E402
import test_snowrun_gpu as TS  # noqa: E402
import test_snowrun2_gpu as TS2  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
done = skipped = 0
worst = 0.0
for it in range(a.n):
    rows, cols = int(rng.integers(8, 70)), int(rng.integers(5, 40))
    ndays = int(rng.integers(6, 26))
    reqhgt = float(rng.choice([0.0, 0.05, 0.3, 1.5]))
    cold = float(rng.choice([-6.0, -3.0, 0.0, 3.0]))
    doy = int(rng.choice([20, 60, 90, 120, 330]))
    nb = int(rng.choice([1, 1, 2, 3]))
    deep = bool(rng.integers(0, 2))
    kind = str(rng.choice(["vector", "vector", "layered", "array"]))
    if kind == "array":
        rows, cols, ndays, nb = min(rows, 30), min(cols, 20), min(ndays, 16), 1          # (fifteen [rows, cols, T] arrays on the host)
        sw, g, dtm, snow, micro = TS2._case(reqhgt, cold, doy, rows=rows, cols=cols, ndays=ndays)
    else:
        sw, g, dtm, snow, micro = TS._case(reqhgt, cold, doy, rows=rows, cols=cols, ndays=ndays)
    layers, layer_of_day = 0, None
    if kind == "layered":
        layers = int(rng.integers(2, 7))
        g = synthetic.layered(g, layers)
        layer_of_day = np.zeros(ndays, int)
        for l in range(layers):
            layer_of_day[g["dfsel"]["st"][l] // 24:(g["dfsel"]["ed"][l] + 1) // 24] = l
    if deep:
        cut = int(rng.integers(1, max(2, rows // 2)))
        pack = np.asfortranarray(np.where(np.arange(rows)[:, None] >= cut, float(rng.uniform(0.2, 1.2)), 0.0) * np.ones((1, cols)))
        other = dict(sw["other"], isnowdc=pack, isnowdg=np.asfortranarray(0.7 * pack))
        sw = dict(sw, other=other)
        snow = dict(snow, other=other)
        micro = dict(micro, other=other)
    kw = {} if nb == 1 else {"devices": [0], "n_blocks": nb}
    try:
        with S.SnowRun(g, snow, **kw) as run:
            sd, nd, smod = run.pass1(want_smod=True)
            if not (sd | nd).all():
                skipped += 1           # a day in neither class: the reference's merge indexes past its arrays
                continue
            got = run.pass2(micro if sd.any() else None, TS2.MAT if kind == "array" else TS.MAT)      # (the mean annual temperature each module's `_orchestrate` uses)
            st = run.stats()
    except Exception as e:     # noqa: BLE001
        print(f"[{it}] rows={rows} cols={cols} ndays={ndays} reqhgt={reqhgt} cold={cold} doy={doy} nb={nb} deep={deep}: RAISED {e}")
        raise
    sdays, ndays_ = np.flatnonzero(sd), np.flatnonzero(nd)
    if ndays_.size == 0 or (sdays.size == 0 and kind != "vector"):
        skipped += 1
        continue
    if sdays.size == 0:
        want = runmicro1Cpp(*[g[k] for k in TS.ARGS])
    elif kind == "array":
        want = TS2._orchestrate(g, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: runmicro2Cpp(*[an[k] for k in TS2.ARGS]), S.gridmicrosnow2,
                                False)
    elif kind == "layered":
        dfs, used = TS._subset_dfsel(layer_of_day, ndays_)
        veg_sub = {k: np.asfortranarray(v[:, :, used]) for k, v in g["vegp"].items()}
        flat = {k: v for k, v in g.items() if k != "dfsel"}
        want = TS._orchestrate(flat, sw, dtm, smod, sdays, ndays_, reqhgt,
                               lambda an: runmicro3Cpp(dfs, *[dict(an, vegp=veg_sub)[k] for k in TS.ARGS]), S.gridmicrosnow1)
    else:
        want = TS._orchestrate(g, sw, dtm, smod, sdays, ndays_, reqhgt, lambda an: runmicro1Cpp(*[an[k] for k in TS.ARGS]), S.gridmicrosnow1)
    tol = 1e-12 if nb == 1 else 1e-9
    err = 0.0
    for k in want:
        gk, wk = got[k], want[k]
        assert np.array_equal(np.isnan(gk), np.isnan(wk)), (it, k, "NA pattern")
        fin = np.isfinite(wk)
        if fin.any():
            err = max(err, float(np.max(np.abs(gk[fin] - wk[fin]) / (1 + np.abs(wk[fin])))))
    assert err < tol, (it, kind, rows, cols, ndays, reqhgt, cold, doy, nb, deep, err)
    worst = max(worst, err)
    done += 1
    print(f"[{it}] {kind}{layers or ''} {rows}x{cols}x{ndays}d reqhgt={reqhgt} cold={cold} doy={doy} blocks={nb} deep={deep}: snow {int(sd.sum())} / no-snow {int(nd.sum())} days, "
          f"err {err:.1e}, left out {st['tile_days_left_out']}/{st['tile_days']}")
print(f"{done} configurations agree (worst {worst:.2e}), {skipped} skipped")
