"""Comparison of model series with the machine-digitised curves of the reference's published figures
(tests/golden/vignette_points.json, written by tools/digitize_vignette.py — see there for how).

Everything is done in PIXEL space of the original figure: the model's series is mapped through the figure's axis
calibration and drawn as a polyline; the figure's curve is the set of pixels of its colour.  Two directed distances:
  fig -> model   how far the published curve strays from the model's polyline (every published pixel accounted for)
  model -> fig   how far the model's polyline strays from the published curve (only meaningful where the curve is not hidden
                 behind another one drawn later)
One pixel is the digitisation error; a line of width 1-2 px, the rasterisation and the axis fit (<= 0.5 px) add ~1.5 px."""
import json
from pathlib import Path

import numpy as np
from scipy.spatial import cKDTree

FIXTURE = Path(__file__).resolve().parent / "golden" / "vignette_points.json"


def panel(fig: str, k: int = 0) -> dict:
    return json.loads(FIXTURE.read_text())["figures"][fig]["panels"][k]


def curve_pixels(p: dict, curve: str) -> np.ndarray:
    pts = []
    for x, flat in p["curves"][curve]["columns"]:
        for a, b in zip(flat[0::2], flat[1::2]):
            pts += [(x, r) for r in range(a, b + 1)]
    return np.array(pts, dtype=np.float64)


def to_px(p: dict, x, y):
    ax, bx, ay, by = p["x"]["per_px"], p["x"]["at_px0"], p["y"]["per_px"], p["y"]["at_px0"]
    return (np.asarray(x, float) - bx) / ax, (np.asarray(y, float) - by) / ay


def polyline_px(p: dict, x, y, step: float = 0.25) -> np.ndarray:
    """the model's series as R draws it: straight segments between consecutive points, clipped to the frame"""
    px, py = to_px(p, x, y)
    ok = np.isfinite(px) & np.isfinite(py)
    out = []
    for i in range(len(px) - 1):
        if not (ok[i] and ok[i + 1]):
            continue
        n = max(2, int(np.hypot(px[i + 1] - px[i], py[i + 1] - py[i]) / step) + 1)
        t = np.linspace(0.0, 1.0, n)
        out.append(np.column_stack([px[i] + t * (px[i + 1] - px[i]), py[i] + t * (py[i + 1] - py[i])]))
    q = np.concatenate(out)
    f = p["frame_px"]
    keep = (q[:, 0] >= f["left"]) & (q[:, 0] <= f["right"]) & (q[:, 1] >= f["top"]) & (q[:, 1] <= f["bottom"])
    return q[keep]


def distances(p: dict, curve: str, x, y) -> dict:
    """directed distances in pixels (max and 99th percentile) between the published curve and the model's polyline"""
    fig = curve_pixels(p, curve)
    mod = polyline_px(p, x, y)
    d_fm = cKDTree(mod).query(fig)[0]
    d_mf = cKDTree(fig).query(mod)[0]
    return {"fig_to_model_max": float(d_fm.max()), "fig_to_model_p99": float(np.percentile(d_fm, 99)),
            "model_to_fig_max": float(d_mf.max()), "model_to_fig_p99": float(np.percentile(d_mf, 99)),
            "fig_pixels": int(len(fig)), "px": p["px"], "d_fm": d_fm, "d_mf": d_mf, "fig": fig, "mod": mod}


def envelope_columns(p: dict, curve: str):
    """per pixel column: (x pixel, lowest value, highest value) of the curve's colour"""
    ay, by = p["y"]["per_px"], p["y"]["at_px0"]
    out = []
    for x, flat in p["curves"][curve]["columns"]:
        out.append((x, ay * max(flat) + by, ay * min(flat) + by))
    return np.array(out)


# ---- raster maps -----------------------------------------------------------------------------------------------------------
def map_panel(fig: str, k: int = 0) -> dict:
    """A digitised map: `lo` / `hi` [rows, cols] = the value range of the colour class each cell was drawn in (NaN = NA)."""
    p = json.loads(FIXTURE.read_text())["maps"][fig]["panels"][k]
    cls = np.array(p["classes"], dtype=np.float64)
    idx = np.array(p["cells"], dtype=np.int64)
    lo = np.where(idx >= 0, cls[np.maximum(idx, 0), 0], np.nan)
    hi = np.where(idx >= 0, cls[np.maximum(idx, 0), 1], np.nan)
    return {"lo": lo, "hi": hi, "class_width": float(np.median(cls[:, 1] - cls[:, 0])), "legend": p["legend"], "what": p["what"]}


def map_compare(m: dict, raster) -> dict:
    """How far the model's raster lies outside the published cells' colour classes, in class widths: 0 inside the class."""
    r = np.asarray(raster, dtype=np.float64)
    na_equal = bool(np.array_equal(np.isnan(r), np.isnan(m["lo"])))
    with np.errstate(invalid="ignore"):
        d = np.maximum(np.maximum(m["lo"] - r, r - m["hi"]), 0.0) / m["class_width"]
    ok = np.isfinite(d)
    d = d[ok]
    return {"na_equal": na_equal, "cells": int(ok.sum()), "max": float(d.max()), "p99": float(np.percentile(d, 99)), "p90": float(np.percentile(d, 90)),
            "median": float(np.median(d)), "within_1": float((d <= 1.0).mean()), "within_3": float((d <= 3.0).mean()),
            "class_width": m["class_width"], "legend_min": m["legend"]["min"], "legend_max": m["legend"]["max"],
            "model_min": float(np.nanmin(r)), "model_max": float(np.nanmax(r))}
