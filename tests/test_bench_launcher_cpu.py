"""bench.py's own rank fan-out (`--gpus N` without an external launcher), driven on CPU: the parent starts
`torch.distributed.run` BEFORE torch / HIP are imported, the ranks meet over gloo, deal the raster, all-reduce the twi
partials, the slowest rank's time and the unit counts, and rank 0's single JSON line comes back through the parent.
`--stub` replaces the solver (which needs a GPU) by a stand-in; launcher, partition and collectives are the code under
test.  The line is flagged "stub": true so that it can never be mistaken for a measurement."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

from microclimf_amd import synthetic

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, timeout=300):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                       cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout            # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def _valid(rows, cols):
    vegp, _, _ = synthetic.rasters(rows, cols, 0, rows)
    return int((~np.isnan(vegp["hgt"])).sum())


def test_defaults_are_the_headline_config():
    sys.path.insert(0, str(ROOT))
    import bench
    a = bench.parse([])
    assert (a.config, a.rows, a.cols, a.terrain, a.scaling, a.gpus, a.tsteps) == (2, 4096, 4096, "device", "weak", 1, 8760)
    a = bench.parse(["--config", "3"])
    assert (a.rows, a.cols, a.terrain, a.scaling, a.share) == (8192, 8192, "device", "strong", 8)
    a = bench.parse(["--config", "1", "--rows", "512"])
    assert (a.rows, a.cols, a.terrain) == (512, 1024, "random")


def test_parent_does_not_import_torch_before_launching():
    code = ("import sys; sys.argv=['bench.py']; import bench; bench.parse(['--gpus','2']); "
            "assert 'torch' not in sys.modules, 'torch imported at module level'")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=str(ROOT))


def test_gpus_2_starts_two_ranks_weak():
    line = _run("--gpus", "2", "--stub", "--config", "1", "--rows", "40", "--cols", "16", "--steps", "2", "--warmup", "0")
    assert line["n_gpus"] == 2 and line["stub"] is True and line["scaling"] == "weak"
    # weak scaling: 40 rows PER rank; the two blocks are rows 0..39 and 40..79 of ONE 80-row raster (seeded by global index)
    assert line["config"]["valid_cells"] == _valid(80, 16)
    _, soilc, _ = synthetic.rasters(80, 16, 0, 80)
    assert abs(line["config"]["twi_mean"] - float(np.mean(np.log(soilc["twi"]) / 1.5))) < 1e-12


def test_gpus_2_strong_share_of_eight_blocks():
    line = _run("--gpus", "2", "--stub", "--config", "3", "--rows", "80", "--cols", "16", "--steps", "1", "--warmup", "0")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    # 80 rows in 8 blocks of 10; ranks 0 and 1 hold rows 0..19
    vegp, _, _ = synthetic.rasters(20, 16, 0, 80)
    assert line["config"]["valid_cells"] == int((~np.isnan(vegp["hgt"])).sum())


def test_single_rank_needs_no_launcher():
    line = _run("--gpus", "1", "--stub", "--config", "1", "--rows", "24", "--cols", "8", "--steps", "1", "--warmup", "0")
    assert line["n_gpus"] == 1 and line["config"]["valid_cells"] == _valid(24, 8)
