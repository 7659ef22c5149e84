#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into the files kept
under profiles/: the rocprofv3 kernel stats of the bench command, the per-launch PMC means of
k_solve and profiles/traffic.json (HBM bytes per launch, gfx950 FETCH_SIZE correction applied
as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE x2, WRITE_SIZE as is)."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"
dst.mkdir(exist_ok=True)

stats = glob.glob(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst / f"{tag}_bench_kernel_stats.csv")
for name in ("bench_under_rocprof.json", "bench.json"):
    if (src / name).exists():
        shutil.copy(src / name, dst / f"{tag}_{name}")

pmc = {}
dur = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc"):
    cc = glob.glob(str(src / d / "*" / "*_counter_collection.csv"))
    if not cc:
        continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if "k_solve" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
    kt = glob.glob(str(src / d / "*" / "*_kernel_trace.csv"))[0]
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))
          if "k_solve" in r["Kernel_Name"]]
    dur[d] = sum(ds) / len(ds)
# days per launch: from the bench line written during the profiled run ("HBM ring (2 slots x N days)")
import re
ring_days, valid_cells = 5, 1038103
bj = dst / f"{tag}_bench_under_rocprof.json"
if bj.exists():
    try:
        cfg = json.loads(bj.read_text())["config"]
        ring_days = int(re.search(r"x (\d+) days", cfg["sink"]).group(1))
        valid_cells = int(cfg["valid_cells"])
    except Exception:
        pass
summary = {"kernel": "k_solve<21,0,false>", "per_launch_mean": pmc, "avg_launch_ms_under_pmc": dur,
           "ring_days": ring_days, "cell_steps_per_launch": valid_cells * ring_days * 24,
           "command": f"python3 bench.py --tsteps 1200 --steps 1 --warmup 0 --no-cpu-baseline ({ring_days}-day launches)"}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch_b = pmc["FETCH_SIZE"] * 1024 * 2       # gfx950: FETCH_SIZE reports half of a coalesced stream
    write_b = pmc["WRITE_SIZE"] * 1024
    summary["hbm_bytes_per_launch"] = {"read": fetch_b, "write": write_b, "total": fetch_b + write_b}
    (dst / "traffic.json").write_text(json.dumps({
        "rows": 1024, "cols": 1024, "ring_days": ring_days, "tag": tag,
        "hbm_bytes_per_launch": fetch_b + write_b, "read_bytes": fetch_b, "write_bytes": write_b,
        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB -> B; FETCH_SIZE x2 (gfx950)"},
        indent=1))
if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    cyc = pmc["GRBM_GUI_ACTIVE"] / 8          # summed over 8 XCDs
    summary["valu_busy_fraction"] = pmc["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)   # quad-cycles, 1024 SIMDs
    summary["clock_ghz"] = cyc / (dur.get("pmc_misc", 1) * 1e-3) / 1e9
(dst / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
