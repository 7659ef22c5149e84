#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into the files kept
under profiles/: the rocprofv3 kernel stats of the bench command, the per-launch PMC means of
k_solve and profiles/traffic.json (HBM bytes per launch, gfx950 FETCH_SIZE correction applied
as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE x2, WRITE_SIZE as is)."""
import csv
import glob
import os
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path


def newest(pattern):
    """the files of the LATEST run matching the pattern: gpurun merges a repeated call's output next to the earlier one's"""
    fs = glob.glob(pattern)
    return [max(fs, key=os.path.getmtime)] if fs else []

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"
dst.mkdir(exist_ok=True)

stats = newest(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst / f"{tag}_bench_kernel_stats.csv")
for name in ("bench_under_rocprof.json", "bench.json"):
    if (src / name).exists():
        shutil.copy(src / name, dst / f"{tag}_{name}")

pmc = {}
dur = {}
kname = "k_solve"
nlaunch = 0
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc"):
    cc = newest(str(src / d / "*" / "*_counter_collection.csv"))
    if not cc:
        continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if "k_solve<" in r["Kernel_Name"]:            # not k_solve_fix (the empty fix-up launches)
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
        nlaunch = len(v)
    kt = newest(str(src / d / "*" / "*_kernel_trace.csv"))[0]
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))
          if "k_solve<" in r["Kernel_Name"]]
    dur[d] = sum(ds) / len(ds)
# days per launch and raster: from the bench line written during the PMC run itself ("HBM ring (S slots x N days)")
import re
sys.path.insert(0, str(root))
import bench  # noqa: E402  (kernel_hash(): the stamp bench.py checks before it reports these counters)
ring_days, valid_cells, rows, cols, tsteps = 5, 1038103, 1024, 1024, 1920
for bj in (src / "pmc_fetch.json", dst / f"{tag}_bench_under_rocprof.json"):
    if bj.exists():
        try:
            cfg = json.loads(bj.read_text().strip().splitlines()[-1])["config"]
            ring_days = int(re.search(r"x (\d+) days", cfg["sink"]).group(1))
            valid_cells = int(cfg["valid_cells"])
            rows, cols = int(cfg["rows_per_gpu"]), int(cfg["cols"])
            tsteps = int(cfg["tsteps"])
            break
        except Exception:
            pass
bargs = (src / "bench_args.txt").read_text().strip() if (src / "bench_args.txt").exists() else ""
summary = {"kernel": kname.replace("void mcf::", "").replace("(mcf::SolveArgs)", ""), "per_launch_mean": pmc, "avg_launch_ms_under_pmc": dur,
           "ring_days": ring_days, "rows": rows, "cols": cols,
           # the series need not be a whole number of launches: the MEAN launch covers total / launches cell-steps
           "launches": nlaunch, "cell_steps_per_launch": valid_cells * (tsteps // 24) * 24 / max(nlaunch, 1),
           "kernel_hash": bench.kernel_hash(),
           "command": f"python3 bench.py {bargs} --tsteps 1920 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "
                      f"--no-verify ({ring_days}-day launches; 80 days: a whole number of launches for 1, 2, 4, 5, 8 and 10-day slots)"}
special = any(f in bargs for f in ("--array-forcing", "--coarse"))     # other geometries: summary only, traffic.json is bench.py's
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and special:
    summary["hbm_bytes_per_launch"] = {"read": pmc["FETCH_SIZE"] * 1024 * 2, "write": pmc["WRITE_SIZE"] * 1024,
                                       "total": pmc["FETCH_SIZE"] * 1024 * 2 + pmc["WRITE_SIZE"] * 1024}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and not special:
    fetch_b = pmc["FETCH_SIZE"] * 1024 * 2       # gfx950: FETCH_SIZE reports half of a coalesced stream
    write_b = pmc["WRITE_SIZE"] * 1024
    summary["hbm_bytes_per_launch"] = {"read": fetch_b, "write": write_b, "total": fetch_b + write_b}
    tf = dst / "traffic.json"
    try:
        old = json.loads(tf.read_text())
        entries = old.get("entries", [old])
    except Exception:
        entries = []
    entries = [e for e in entries if (e.get("rows"), e.get("cols"), e.get("ring_days")) != (rows, cols, ring_days)]
    entries.append({"rows": rows, "cols": cols, "ring_days": ring_days, "tag": tag, "kernel_hash": bench.kernel_hash(),
                    "hbm_bytes_per_launch": fetch_b + write_b, "read_bytes": fetch_b, "write_bytes": write_b,
                    "cell_steps_per_launch": summary["cell_steps_per_launch"]})
    tf.write_text(json.dumps({
        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB -> B; FETCH_SIZE x2 (gfx950); "
                  "kernel_hash = bench.kernel_hash() of the sources the counters were taken from",
        "entries": entries}, indent=1))
if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    cyc = pmc["GRBM_GUI_ACTIVE"] / 8          # summed over 8 XCDs
    summary["valu_busy_fraction"] = pmc["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)   # quad-cycles, 1024 SIMDs
    summary["clock_ghz"] = cyc / (dur.get("pmc_misc", 1) * 1e-3) / 1e9
(dst / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
