"""bench.py as the driver runs it, on a small workload: ONE JSON line with the contract's keys, the timed output checked
against the oracle inside the run (`verified`), the roofline / cpu_baseline objects, the secondary block — and the N = 2
path rehearsed over gloo with both ranks on this GPU (partition, twi all-reduce, terrain halo exchange, verification)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_one_line_with_roofline_baseline_and_in_run_verification():
    d = _bench("--config", "2", "--rows", "96", "--cols", "64", "--tsteps", "240", "--steps", "2", "--warmup", "1",
               "--cpu-sample", "16x16x48", "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "terrain_precompute_s"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 0
    assert d["config"]["terrain"].startswith("on-device")
    r = d["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    v = d["verified"]
    assert v["ok"] and v["na_pattern_equal"] and v["max_scaled_err"] < 1e-6 and v["cells"] >= 200 and v["steps"] >= 24
    assert d["config"]["dispatch"]["fast_launches"] > 0


def test_headline_raster_at_full_size_one_launch_is_verified():
    """BASELINE.json configs[2]'s raster at FULL size — 4096 x 4096 cells, terrain inputs built on the device — for one 7-day
    launch (the year is 53 of them): the timed output of 264 sampled cells against the oracle, every tile through the fast path"""
    d = _bench("--config", "2", "--tsteps", "168", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary")
    assert d["config"]["rows_per_gpu"] == 4096 and d["config"]["cols"] == 4096 and d["config"]["baseline_config"] == 2 and d["config"]["terrain"].startswith("on-device")
    v = d["verified"]
    assert v["ok"] and v["na_pattern_equal"] and v["max_scaled_err"] < 1e-6 and v["cells"] >= 200
    assert d["config"]["dispatch"]["fast_launches"] > 0 and d["value"] > 1e10


def test_secondary_block_covers_the_other_geometries():
    d = _bench("--config", "2", "--rows", "64", "--cols", "64", "--tsteps", "240", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    s = d["secondary"]
    assert set(s) == {"configs[1]", "array_forcing", "coarse_forcing_8x8"}
    for name, e in s.items():
        assert "error" not in e, (name, e)
        assert e["value"] > 0 and 0 < e["hbm_frac"] < 1
    for name in s:
        assert s[name]["verified"]["ok"], (name, s[name]["verified"])


def test_two_ranks_over_gloo_share_this_gpu():
    d = _bench("--gpus", "2", "--config", "2", "--rows", "160", "--cols", "96", "--tsteps", "240", "--steps", "1", "--warmup", "0",
               "--no-secondary", env={"MCF_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["verified"]["ok"] and "REHEARSAL" in d["config"]["partition"]
    d = _bench("--gpus", "2", "--config", "4", "--rows", "320", "--cols", "96", "--share", "2", "--tsteps", "720", "--steps", "1",
               "--warmup", "0", env={"MCF_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["baseline_config"] == 4
    assert d["verified"]["ok"], d["verified"]          # the last snow chunk's model against the oracle (rank 0's block)


def test_snow_config_line_is_checked_against_the_oracle_in_the_run():
    d = _bench("--config", "4", "--rows", "160", "--cols", "96", "--tsteps", "1440", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    v = d["verified"]
    # the ten merged outputs of a probed year against the oracle-backed orchestration ...
    assert v["ok"] and v["max_scaled_err"] < 1e-6 and v["cells"] > 100 and v["days"] >= 55 and len(v["outputs"]) == 10, v
    assert v["values"] > 1e6 and v["cell_steps_by_class"]["snow_covered"] > 0
    # ... and the snow model of the last snow chunk
    vs = v["snowmodel"]
    assert vs["ok"] and vs["max_scaled_err"] < 1e-6 and vs["cells"] > 100 and vs["steps"] == 120, vs
    assert "checkpoint" in d["config"]["passes"]
