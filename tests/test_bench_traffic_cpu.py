"""bench.py's counter accounting: `roofline.traffic` is put on the SAME launch basis as
`algorithmic_bytes_per_launch` — writes (and array forcing's reads) scale with the cell-steps of the run's mean launch, vector
forcing's per-launch constant images do not — and counters taken from other kernel sources are withheld with the reason."""
import json

import bench


def _fake(monkeypatch, tmp_path, khash):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "traffic.json").write_text(json.dumps({"entries": [
        {"rows": 64, "cols": 64, "ring_days": 7, "tag": "t", "kernel_hash": khash, "hbm_bytes_per_launch": 110.0, "read_bytes": 10.0,
         "write_bytes": 100.0, "cell_steps_per_launch": 1000.0}]}))
    (prof / "t_pmc_summary.json").write_text(json.dumps({"cell_steps_per_launch": 1000.0, "per_launch_mean": {}}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)


def test_traffic_is_scaled_to_the_runs_own_launch(monkeypatch, tmp_path):
    monkeypatch.setattr(bench, "kernel_hash", lambda: "abc")
    _fake(monkeypatch, tmp_path, "abc")
    # this run's mean launch covers 10 cells x 90 steps = 900 cell-steps (a year ends in a shorter launch)
    rb = bench.roofline_block(valid=10, T=8760, steps_per_launch=90, avg_ms=1.0, klaunches=5, af=False, rows=64, cols=64, ring_days=7,
                              rate_per_gpu=1e9, coarse=None)
    assert abs(rb["traffic"] - (10.0 + 100.0 * 0.9)) < 1e-9              # reads per launch as they are, writes x 900 / 1000
    tb = rb["traffic_basis"]
    assert tb["counter_run_cell_steps_per_launch"] == 1000.0 and tb["this_run_cell_steps_per_launch"] == 900.0
    assert abs(tb["ratio_to_algorithmic"] - rb["traffic"] / rb["algorithmic_bytes_per_launch"]) < 1e-12


def test_stale_counters_are_withheld(monkeypatch, tmp_path):
    monkeypatch.setattr(bench, "kernel_hash", lambda: "new")
    _fake(monkeypatch, tmp_path, "old")
    rb = bench.roofline_block(valid=10, T=8760, steps_per_launch=90, avg_ms=1.0, klaunches=5, af=False, rows=64, cols=64, ring_days=7,
                              rate_per_gpu=1e9, coarse=None)
    assert rb["traffic"] is None and rb["valu"] is None and "stale" in rb["counters"]


def test_pipeline_traffic_needs_the_snow_sources_hash(monkeypatch, tmp_path):
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "x_c4_aux_pmc_summary.json").write_text(json.dumps({
        "_meta": {"kernel_hash": "h1", "command": "python3 bench.py --config 4"},
        "k_a": {"launches": 2, "hbm_bytes_per_launch": {"read": 1.0, "write": 3.0}},
        "k_b": {"launches": 1, "per_launch_mean": {}}}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "snow_kernel_hash", lambda: "h1")
    t, note = bench.committed_pipeline_traffic()
    assert note is None and t["bytes_per_profiled_run"] == 8.0 and list(t["kernels"]) == ["k_a"]
    monkeypatch.setattr(bench, "snow_kernel_hash", lambda: "h2")
    t, note = bench.committed_pipeline_traffic()
    assert t is None and "stale" in note


def test_another_rasters_counters_are_scaled_reads_by_cells_writes_by_cell_steps(monkeypatch, tmp_path):
    """A raster with no counter run of its own (a rank's block of configs[3]): k_solve's bytes per cell-step do not depend on the
    raster's shape — the current-source entry is used, said so, with the reads (per-launch constant images) scaled by the cells
    of a launch and the writes by its cell-steps."""
    monkeypatch.setattr(bench, "kernel_hash", lambda: "abc")
    _fake(monkeypatch, tmp_path, "abc")          # 64 x 64, 7-day launches: 1000 cell-steps per launch
    # this run: 3 x as many cells, 14-day launches
    cells_pmc = 1000.0 / (24 * 7)
    valid, spl = 3 * cells_pmc, 24 * 14
    rb = bench.roofline_block(valid=valid, T=8760, steps_per_launch=spl, avg_ms=1.0, klaunches=5, af=False, rows=128, cols=96,
                              ring_days=14, rate_per_gpu=1e9, coarse=None)
    assert abs(rb["traffic"] - (10.0 * 3 + 100.0 * 6)) < 1e-9
    tb = rb["traffic_basis"]
    assert "ANOTHER raster" in tb["source"] and abs(tb["reads_scaled_by_cells"] - 3.0) < 1e-12 and abs(tb["writes_scaled_by_cell_steps"] - 6.0) < 1e-12
    # ... but never counters of other sources
    monkeypatch.setattr(bench, "kernel_hash", lambda: "new")
    rb = bench.roofline_block(valid=valid, T=8760, steps_per_launch=spl, avg_ms=1.0, klaunches=5, af=False, rows=128, cols=96,
                              ring_days=14, rate_per_gpu=1e9, coarse=None)
    assert rb["traffic"] is None and "no counters" in rb["counters"]
