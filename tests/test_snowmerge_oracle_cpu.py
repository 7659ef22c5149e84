"""oracle/snowmerge_oracle.py — the numpy restatement of `.runmicrosnow1`'s template / merge steps (R/internal.R:3565-3578,
3625-3656) — on hand-made day lists whose expected layout is written out day by day, and against the product's own
`snow.merge_snow_outputs` (two independent writings of the same R lines must agree)."""
import numpy as np

from microclimf_amd import snow as S
from oracle import snowmerge_oracle as M


def _series(days, tag):
    """[1, 1, 24 * len(days)]: value = tag + day + hour / 100"""
    v = np.concatenate([tag + d + np.arange(24) / 100.0 for d in days]) if len(days) else np.zeros(0)
    return v.reshape(1, 1, -1)


def test_template_and_merge_day_by_day():
    snowdays, nosnowdays = [1, 2, 4, 5], [2, 3, 5, 6]            # days 2 and 5 in both classes, 3 and 6 snow-free, 1 and 4 snow only
    moutn = {"Tz": _series(nosnowdays, 100.0), "soilm": _series(nosnowdays, 200.0)}
    micro = M.prep_micro(moutn, snowdays, nosnowdays, 1, 1)
    tz = micro["Tz"][0, 0].reshape(4, 24)
    assert np.isnan(tz[0]).all() and np.isnan(tz[2]).all()                     # days 1, 4: blank
    assert np.array_equal(tz[1], 100.0 + 2 + np.arange(24) / 100.0)             # day 2: the no-snow model's day 2
    assert np.array_equal(tz[3], 100.0 + 5 + np.arange(24) / 100.0)
    mouts = {"Tz": _series(snowdays, 300.0), "soilm": micro["soilm"]}
    out = M.merge(moutn, mouts, snowdays, nosnowdays, 1, 1)
    tz = out["Tz"][0, 0].reshape(6, 24)
    for d in (1, 2, 4, 5):
        assert np.array_equal(tz[d - 1], 300.0 + d + np.arange(24) / 100.0), d   # snow days: the snow microclimate, all of it
    for d in (3, 6):
        assert np.array_equal(tz[d - 1], 100.0 + d + np.arange(24) / 100.0), d   # days without any snow: the no-snow model
    sm = out["soilm"][0, 0].reshape(6, 24)
    assert np.isnan(sm[0]).all() and np.isnan(sm[3]).all()                      # a variable the snow model does not return: blank on snow-only days
    assert np.array_equal(sm[1], 200.0 + 2 + np.arange(24) / 100.0) and np.array_equal(sm[2], 200.0 + 3 + np.arange(24) / 100.0)


def test_one_class_empty_returns_the_other():
    a = {"Tz": _series([1, 2], 1.0)}
    assert M.merge({}, a, [1, 2], [], 1, 1)["Tz"] is a["Tz"]
    assert M.merge(a, {}, [], [1, 2], 1, 1)["Tz"] is a["Tz"]


def test_agrees_with_the_products_merge_on_random_day_classes():
    rng = np.random.default_rng(11)
    for _ in range(20):
        nd = int(rng.integers(3, 12))
        cls = rng.integers(1, 4, nd)                       # 1 snow only, 2 no-snow only, 3 both
        snowdays = [d + 1 for d in range(nd) if cls[d] & 1]
        nosnowdays = [d + 1 for d in range(nd) if cls[d] & 2]
        if not snowdays or not nosnowdays:
            continue
        moutn = {"Tz": rng.normal(size=(3, 2, 24 * len(nosnowdays)))}
        mouts = {"Tz": rng.normal(size=(3, 2, 24 * len(snowdays)))}
        want = M.merge(moutn, mouts, snowdays, nosnowdays, 3, 2)
        got = S.merge_snow_outputs(moutn, mouts, np.array(snowdays), np.array(nosnowdays), 3, 2)
        assert np.array_equal(got["Tz"], want["Tz"], equal_nan=True)
