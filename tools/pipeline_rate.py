"""End-to-end rate of the tile pipeline (solve in day chunks -> netCDF file): python tools/pipeline_rate.py
[--rows 512 --cols 512 --days 60 --vars Tz,tleaf,...] — the body of runmicro_big's loop for one tile."""
import argparse
import os
import shutil
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from microclimf_amd import ncsink, pipeline, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=512)
ap.add_argument("--cols", type=int, default=512)
ap.add_argument("--days", type=int, default=60)
ap.add_argument("--vars", default="")
ap.add_argument("--dir", default="/tmp")
a = ap.parse_args()
names = tuple(a.vars.split(",")) if a.vars else ncsink.default_vars(0.05)
free = shutil.disk_usage(a.dir).free
need = a.rows * a.cols * a.days * 24 * 4 * len(names)
if need > 0.5 * free:
    raise SystemExit(f"{need / 1e9:.1f} GB would not fit {a.dir} comfortably ({free / 1e9:.1f} GB free)")
w = synthetic.workload(a.rows, a.cols, a.days * 24, reqhgt=0.05)
dtm = {"xmin": 0.0, "xmax": a.cols * 1.0, "ymin": 0.0, "ymax": a.rows * 1.0, "res": 1.0, "crs": "local"}
path = os.path.join(a.dir, "mcf_pipeline_rate.nc")
pipeline.run_to_nc(synthetic.workload(64, 64, 48, reqhgt=0.05), path, {**dtm, "xmax": 64.0, "ymax": 64.0}, vars=names)   # warm-up
t0 = time.perf_counter()
info = pipeline.run_to_nc(w, path, dtm, vars=names, days_per_chunk=5)
dt = time.perf_counter() - t0
size = os.path.getsize(path)
os.remove(path)
cs = info["valid_cells"] * info["steps"]
print(f"{a.rows}x{a.cols} cells x {info['steps']} steps, {len(names)} variables -> {size / 1e9:.2f} GB file in {dt:.2f} s "
      f"(setup {info['setup_s']:.2f} s, solve {info['solve_s']:.2f} s, pack + copy + write {info['write_s']:.2f} s, close {info['close_file_s']:.2f} s): {cs / dt:.3e} cell-steps/s end to end, "
      f"{size / dt / 1e9:.2f} GB/s to {a.dir} ({free / 1e9:.0f} GB free)")
