#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>_aux/ (tools/profile_aux.sh) into profiles/<tag>_aux_kernel_stats.csv and
profiles/<tag>_aux_pmc_summary.json: per kernel the mean duration, VALU instructions per lane and the VALU-busy fraction."""
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

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = Path(__file__).resolve().parents[1]
src = root / "gpurun_out" / f"prof_{tag}_aux"
dst = root / "profiles"
st = newest(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if st:
    shutil.copy(st[0], dst / f"{tag}_aux_kernel_stats.csv")
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in ("pmc_sq", "pmc_misc", "pmc_fetch", "pmc_write"):
    for f in newest(str(src / d / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in newest(str(src / d / "*" / "*_kernel_trace.csv")):
        if d == "pmc_misc":
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {}
for k, c in agg.items():
    if not any(s in k for s in ("k_snowmodel", "k_microsnow", "k_snow_redistribute", "k_horizon", "k_windcoef", "k_apply3_part",
                                "k_pack", "k_tpi", "k_tiles_covered", "k_meand", "k_solve<", "k_apply3_minmax", "k_surface")):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"launches": len(next(iter(c.values()))), "mean_ms": sum(dur[k]) / max(len(dur[k]), 1), "per_launch_mean": m}
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        e["valu_busy_fraction"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:        # KB -> B; FETCH_SIZE x2 on gfx950 (the guide's correction)
        e["hbm_bytes_per_launch"] = {"read": m["FETCH_SIZE"] * 1024 * 2, "write": m["WRITE_SIZE"] * 1024}
        if e["mean_ms"] > 0:
            e["hbm_TBps"] = (m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024) / (e["mean_ms"] * 1e-3) / 1e12
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
        e["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    out[k.replace("(anonymous namespace)::", "")[:90]] = e
sys.path.insert(0, str(root))
import bench  # noqa: E402  (snow_kernel_hash(): the stamp bench.py --config 4 checks before it reports these bytes)
wl = (src / "workload.txt").read_text().strip() if (src / "workload.txt").exists() else ""
out["_meta"] = {"kernel_hash": bench.snow_kernel_hash(), "command": "python3 " + wl}
(dst / f"{tag}_aux_pmc_summary.json").write_text(json.dumps(out, indent=1))
for k, e in out.items():
    if k == "_meta":
        continue
    print(f"{k[:60]:60s} {e['launches']:4d} x {e['mean_ms']:8.3f} ms  VALU/wave {e.get('valu_insts_per_wave', 0):9.0f}  busy {e.get('valu_busy_fraction', 0):.2f}  HBM {e.get('hbm_TBps', 0):.2f} TB/s")
