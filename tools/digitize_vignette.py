#!/usr/bin/env python3
"""Machine digitisation of the reference's published result figures (vignettes/images/*.png) into a committed fixture.

    python tools/digitize_vignette.py            # writes tests/golden/vignette_points.json   (build container only)

The reference holds no numeric outputs of its grid model — its tests assert intervals only — but its
vignette prints base-R line plots of model output on the bundled example data.  Those PNGs are the only results of the
reference that exist in this environment.  This script turns them into numbers WITHOUT a human reading values off a
screen:

  * plot boxes are found as the long black rectangles; axis ticks as the short black strokes just outside a box;
  * each axis is calibrated by a least-squares line through (tick pixel, tick label): the tick PIXELS are detected, the
    tick LABELS are the printed numbers, typed once into SPEC below (no OCR engine in the image) and cross-checked by
    the fit — equally spaced labels must sit on equally spaced pixels to within `max_fit_px`;
  * every curve is extracted by colour: per pixel column inside the box, the lowest and highest pixel of that colour,
    converted to data units.  One pixel is the digitisation error (`px` per axis in the fixture), e.g. 0.046 degC in
    image9, 0.39 mm of snow water equivalent in image14a.

Raster maps (terra::plot with a colour legend) are digitised too, cell by cell: the legend's ramp gives colour -> value, every
raster cell's colour then gives its value to the width of one colour class (~0.1 degC on the temperature maps) — see MAPS.

The fixture carries the calibration, the per-column envelopes of every curve and where they were taken from;
tests/test_frontend_gpu.py::test_vignette_* draw the model's own series into the same pixel columns and compare
envelopes.  The PNGs themselves are not copied (they stay under /root/reference); the fixture is data derived from them.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[1]
IMAGES = Path("/root/reference/vignettes/images")
OUT = ROOT / "tests" / "golden" / "vignette_points.json"

HOURS_2017 = {"Jan 2017": 0, "Apr 2017": 24 * 90, "Jul 2017": 24 * 181, "Oct 2017": 24 * 273, "Jan 2018": 24 * 365}

# Printed tick labels of each panel (top to bottom panels; x ticks left to right, y ticks bottom to top) and the curves
# to extract.  `x_is`: what the x coordinate of the plotted series is (0-based hour of 2017, R's 1-based index, or a
# data value).  Source lines: vignettes/running-microclimf.Rmd.
SPEC = {
    "image7": dict(rmd="408-432", what="Tz above a 5 mm sward, entry 132 of the monthly-tmax subset, at 0.01 .. 1 m",
                   panels=[dict(xticks=[30, 35, 40], yticks=[0.0, 0.2, 0.4, 0.6, 0.8, 1.0], x_is="temperature", y_is="height",
                                curves={"profile": "black"})]),
    "image8": dict(rmd="440-466", what="Tz under a 10 m canopy of pai 3, entry 132 of the monthly-tmax subset, 0.1 .. 10 m",
                   panels=[dict(xticks=[18, 19, 20, 21, 22, 23, 24, 25], yticks=[0, 2, 4, 6, 8, 10], x_is="temperature", y_is="height",
                                curves={"profile": "black"})]),
    "image9": dict(rmd="470-494", what="soil temperature under that canopy over 2017 at 5 cm (red), 20 cm (black), 1 m (grey)",
                   panels=[dict(xticks=[0, 2000, 4000, 6000, 8000], yticks=[0, 5, 10, 15, 20, 25], x_is="index1", y_is="degC",
                                curves={"d005": "red", "d020": "black", "d100": "grey"})]),
    "image14a": dict(rmd="685-703", what="runsnowmodel at -12 K, snowenv Maritime: raster-mean snow water equivalent and depth",
                     panels=[dict(xticks=list(HOURS_2017.values()), yticks=[0, 50, 100, 150], x_is="hour", y_is="mm",    # R printed 0, 50, 100
                                  curves={"swe": "black"}),
                             dict(xticks=list(HOURS_2017.values()), yticks=[0.0, 0.1, 0.2, 0.3, 0.4, 0.5], x_is="hour", y_is="m",   # printed: 0.0, 0.2, 0.4
                                  curves={"depth": "black"})]),
    "image14p": dict(rmd="663-683", what="subset snow runs at -12 K (each month's coldest day): depth, slow (blue) and fast (red)",
                     panels=[dict(xticks=[0, 50, 100, 150, 200, 250], yticks=[0.0, 0.2, 0.4, 0.6, 0.8, 1.0], x_is="index1", y_is="m",
                                  curves={"slow": "blue", "fast": "red"})]),
    "image14b": dict(rmd="707-731", what="runmicro on the monthly-tmin subset without (red) and with snow (blue): mean Tz and soil moisture",
                     panels=[dict(xticks=[0, 50, 100, 150, 200, 250], yticks=[-10, 0, 10, 20, 30], x_is="index1", y_is="degC",
                                  curves={"nosnow": "red", "snow": "blue"}),
                             dict(xticks=[0, 50, 100, 150, 200, 250], yticks=[0.0, 0.1, 0.2, 0.3, 0.4, 0.5], x_is="index1", y_is="fraction",   # printed: 0.0, 0.2, 0.4
                                  curves={"nosnow": "red", "snow": "blue"})]),
    "image1b": dict(rmd="127-140", what="runpointmodel on the bundled year: canopy (teal), air (grey) and ground (red) temperature",
                    panels=[dict(xticks=list(HOURS_2017.values()), yticks=[0, 10, 20, 30, 40, 50], x_is="hour", y_is="degC",
                                 curves={"all": "any"})]),
}
MAX_FIT_PX = 0.75


def colour_mask(img, name):
    r, g, b = (img[:, :, k].astype(np.int32) for k in range(3))
    if name == "black":
        return (r < 90) & (g < 90) & (b < 90)
    if name == "red":
        return (r > 180) & (g < 90) & (b < 90)
    if name == "blue":
        return (b > 180) & (r < 90) & (g < 90)
    if name == "grey":
        return (abs(r - g) < 14) & (abs(g - b) < 14) & (r > 120) & (r < 215)
    if name == "any":
        return (r < 235) | (g < 235) | (b < 235)
    raise ValueError(name)


def runs(mask1d, min_len):
    """[(start, stop)] of the True runs of at least min_len"""
    out, start = [], None
    for i, v in enumerate(np.append(mask1d, False)):
        if v and start is None:
            start = i
        elif not v and start is not None:
            if i - start >= min_len:
                out.append((start, i))
            start = None
    return out


def find_boxes(img):
    """Plot regions as (top, bottom, left, right) pixel coordinates of the black frame, top to bottom."""
    dk = colour_mask(img, "black")
    H, W = dk.shape
    lines = []
    for y in range(H):
        for x0, x1 in runs(dk[y], int(0.5 * W)):
            lines.append((y, x0, x1 - 1))
    # merge adjacent rows of the same stroke
    merged = []
    for y, x0, x1 in lines:
        if merged and y - merged[-1][0] <= 1 and abs(x0 - merged[-1][1]) <= 1:
            continue
        merged.append((y, x0, x1))
    assert len(merged) % 2 == 0 and merged, f"expected pairs of horizontal frame lines, got {merged}"
    boxes = []
    for (ya, xa0, xa1), (yb, xb0, xb1) in zip(merged[0::2], merged[1::2]):
        assert abs(xa0 - xb0) <= 1 and abs(xa1 - xb1) <= 1, "top and bottom of a frame do not line up"
        boxes.append((ya, yb, xa0, xa1))
    return boxes


def tick_pixels(img, box, side, depth=(3, 6)):
    """Centres of the tick strokes outside the frame: side 'left' -> rows, 'bottom' -> columns.  `depth`: the pixels beyond
    the frame line that a stroke must cover in full (base R's strokes are 7 px long, terra's 4)."""
    dk = colour_mask(img, "black")
    top, bot, left, right = box
    d0, d1 = depth
    if side == "left":
        strip = dk[:, left - d1 + 1:left - d0 + 1].all(axis=1)        # a tick covers these columns in full; label glyphs are further out
        strip[:max(top - 2, 0)] = False
        strip[bot + 3:] = False
    else:
        strip = dk[bot + d0:bot + d1, :].all(axis=0)
        strip[:max(left - 2, 0)] = False
        strip[right + 3:] = False
    return [0.5 * (a + b - 1) for a, b in runs(strip, 1)]


def calibrate(pix, labels, what):
    pix, labels = np.asarray(pix, float), np.asarray(labels, float)
    assert len(pix) == len(labels), f"{what}: {len(pix)} ticks detected at {pix}, {len(labels)} labels given"
    a, b = np.polyfit(pix, labels, 1)             # value = a * pixel + b
    resid_px = np.abs((labels - b) / a - pix).max()
    assert resid_px <= MAX_FIT_PX, f"{what}: labels do not sit on a line through the tick pixels ({resid_px:.2f} px)"
    return float(a), float(b), float(resid_px)


def digitize(name, spec):
    img = np.array(Image.open(IMAGES / f"{name}.png").convert("RGB"))
    boxes = find_boxes(img)
    assert len(boxes) == len(spec["panels"]), f"{name}: {len(boxes)} plot frames found, {len(spec['panels'])} expected"
    panels = []
    for box, ps in zip(boxes, spec["panels"]):
        top, bot, left, right = box
        ypix = tick_pixels(img, box, "left")
        xpix = tick_pixels(img, box, "bottom")
        if ps["xticks"] is None or ps["yticks"] is None:
            raise SystemExit(f"{name}: fill in the printed tick labels; detected {len(xpix)} x ticks at {xpix}, "
                             f"{len(ypix)} y ticks at {ypix}")
        ax, bx, rx = calibrate(xpix, ps["xticks"], f"{name} x")
        ay, by, ry = calibrate(ypix[::-1], ps["yticks"], f"{name} y")          # rows grow downwards: bottom label first
        interior = img[top + 1:bot, left + 1:right]
        curves = {}
        for cname, colour in ps["curves"].items():
            m = colour_mask(interior, colour)
            cols = []
            for j in range(m.shape[1]):
                rr = runs(m[:, j], 1)
                if rr:
                    flat = []
                    for a, b in rr:
                        flat += [top + 1 + a, top + b]            # first and last pixel row of the run (inclusive)
                    cols.append([left + 1 + j, flat])
            curves[cname] = {"colour": colour, "columns": cols, "pixels": int(m.sum())}
        panels.append({
            "frame_px": {"top": top, "bottom": bot, "left": left, "right": right},
            "x": {"is": ps["x_is"], "tick_px": xpix, "tick_labels": ps["xticks"], "per_px": ax, "at_px0": bx, "fit_resid_px": rx},
            "y": {"is": ps["y_is"], "tick_px": ypix[::-1], "tick_labels": ps["yticks"], "per_px": ay, "at_px0": by, "fit_resid_px": ry},
            "px": {"x": abs(ax), "y": abs(ay)},
            "curves": curves,
        })
    return {"source": f"vignettes/images/{name}.png", "rmd_lines": spec["rmd"], "what": spec["what"], "panels": panels}


# ---- raster maps (terra::plot of a 50 x 50 SpatRaster with a continuous colour legend) ---------------------------------------
# Per panel: the printed tick labels of the two axes and of the legend, top to bottom / left to right as for the line plots.
# Everything else is detected: the frames (black rectangles: wide = map, narrow = legend), the tick strokes, the legend's
# colour ramp (one colour per pixel row -> value by the legend's calibration) and the colour of every raster cell.  A cell's
# value is then known to the width of its colour class (terra cuts the data range into equal classes): the fixture stores
# the classes ([lo, hi] in data units) and the class index of each cell, -1 = NA (white).
XT50 = [0, 10, 20, 30, 40, 50]
MAPS = {
    "image1a": dict(rmd="113-126", nrow=50, ncol=50, panels=[
        dict(what="Tz at 5 cm on the hottest hour of the monthly-tmax subset", xticks=XT50, yticks=[0, 20, 40], legend=[30, 35, 40, 45, 50]),
        dict(what="mean of the monthly tmax / tmin subsets' Tz", xticks=XT50, yticks=[0, 20, 40],
             legend=[10.5, 11.0, 11.5, 12.0, 12.5, 13.0, 13.5, 14.0])]),
    "image2": dict(rmd="322-330", nrow=50, ncol=50, panels=[
        dict(what="soil moisture on the hottest hour", xticks=XT50, yticks=XT50, legend=[0.15, 0.20, 0.25, 0.30, 0.35, 0.40])]),
    "image4": dict(rmd="370-378", nrow=50, ncol=50, panels=[
        dict(what="wind speed at step 100 of the monthly-tmax subset", xticks=XT50, yticks=XT50, legend=[0.5, 1.0, 1.5, 2.0, 2.5, 3.0])]),
    "image3b": dict(rmd="340-360", nrow=50, ncol=50, panels=[
        dict(what="downward short wave at 10:00 on 20 June", xticks=XT50, yticks=[0, 20, 40], legend=[200, 400, 600, 800]),
        dict(what="upward short wave at 10:00 on 20 June", xticks=XT50, yticks=[0, 20, 40], legend=[50, 100, 150, 200, 250])]),
    "image5": dict(rmd="380-390", nrow=50, ncol=50, panels=[
        dict(what="soil surface temperature on the hottest hour", xticks=XT50, yticks=XT50, legend=[25, 30, 35, 40, 45, 50, 55])]),
    "image6": dict(rmd="392-397", nrow=50, ncol=50, panels=[
        dict(what="Tz[,,134] of the monthly-tmax subset", xticks=XT50, yticks=XT50, legend=[30, 35, 40, 45, 50])]),
    "image10": dict(rmd="540-549", nrow=50, ncol=50, panels=[
        dict(what="layer 12 of Tz read back from the writetonc file, / 100", xticks=[169480, 169490, 169500, 169510, 169520],
             yticks=[12480, 12490, 12500, 12510, 12520], legend=[8, 10, 12, 14, 16, 18])]),
    "image11": dict(rmd="590-600", nrow=50, ncol=50, panels=[
        dict(what="runbioclim(..., temp = 'air')[[12]]", xticks=[169480, 169490, 169500, 169510, 169520],
             yticks=[12480, 12490, 12500, 12510, 12520], legend=[0.390, 0.395, 0.400, 0.405, 0.410, 0.415])]),
}


def _runs2d(mask, axis, min_len):
    """black strokes: [(fixed index, start, stop)] of runs of at least min_len along `axis`"""
    out = []
    m = mask if axis == 1 else mask.T
    for i in range(m.shape[0]):
        for a, b in runs(m[i], min_len):
            out.append((i, a, b - 1))
    return out


def find_rectangles(img, min_side=60):
    """Black axis-aligned frames as (top, bottom, left, right): pairs of vertical strokes joined by horizontal ones."""
    dk = colour_mask(img, "black")
    vert = _runs2d(dk, 0, min_side)           # (column, row0, row1)
    # merge neighbouring columns of one stroke
    vert.sort()
    strokes = []
    for c, a, b in vert:
        if strokes and c - strokes[-1][0] <= 1 and abs(a - strokes[-1][1]) <= 2 and abs(b - strokes[-1][2]) <= 2:
            continue
        strokes.append((c, a, b))
    rects = []
    for i, (c0, a0, b0) in enumerate(strokes):
        for c1, a1, b1 in strokes[i + 1:]:
            if abs(a0 - a1) <= 2 and abs(b0 - b1) <= 2 and c1 - c0 >= 6:
                # the horizontal strokes that close the frame (corner ticks may carry the vertical strokes past them)
                full = [y for y in range(min(a0, a1), max(b0, b1) + 1) if dk[y, c0:c1 + 1].mean() > 0.97]
                if len(full) >= 2 and full[-1] - full[0] >= min_side:
                    rects.append((full[0], full[-1], c0, c1))
                    break
    return rects


def digitize_map(name, spec):
    img = np.array(Image.open(IMAGES / f"{name}.png").convert("RGB"))
    rects = find_rectangles(img)
    maps = sorted([r for r in rects if r[3] - r[2] > 100], key=lambda r: r[2])
    bars = sorted([r for r in rects if r[3] - r[2] <= 40], key=lambda r: r[2])
    assert len(maps) == len(bars) == len(spec["panels"]), f"{name}: {len(maps)} map frames, {len(bars)} legends, {len(spec['panels'])} expected"
    nrow, ncol = spec["nrow"], spec["ncol"]
    panels = []
    for box, bar, ps in zip(maps, bars, spec["panels"]):
        top, bot, left, right = box
        ax, bx, rx = calibrate(tick_pixels(img, box, "bottom", (1, 4)), ps["xticks"], f"{name} x")
        ay, by, ry = calibrate(tick_pixels(img, box, "left", (1, 4))[::-1], ps["yticks"], f"{name} y")
        # legend: ticks to the right of the bar
        bt, bb, bl, br = bar
        dk = colour_mask(img, "black")
        strip = dk[:, br + 2:br + 5].all(axis=1)
        strip[:max(bt - 2, 0)] = False
        strip[bb + 3:] = False
        lpix = [0.5 * (a + b - 1) for a, b in runs(strip, 1)]
        al, bl0, rl = calibrate(lpix[::-1], ps["legend"], f"{name} legend")
        ramp = np.median(img[bt + 1:bb, bl + 2:br - 1].astype(np.int32), axis=1).astype(np.int32)      # one colour per pixel row
        rows = np.arange(bt + 1, bb)
        # colour classes: maximal runs of rows with one colour
        classes, start = [], 0
        for i in range(1, len(ramp) + 1):
            if i == len(ramp) or (ramp[i] != ramp[start]).any():
                # a class spans pixel rows [start, i): its values run from the lower edge of the last row to the upper edge of the first
                v_hi = al * (rows[start] - 0.5) + bl0
                v_lo = al * (rows[i - 1] + 0.5) + bl0
                classes.append({"rgb": [int(v) for v in ramp[start]], "lo": float(min(v_lo, v_hi)), "hi": float(max(v_lo, v_hi))})
                start = i
        cols_rgb = np.array([c["rgb"] for c in classes], dtype=np.int32)
        # raster cells: the commonest colour of the cell's inner pixels
        cw, ch = (right - left) / ncol, (bot - top) / nrow
        extent = [ax * left + bx, ax * right + bx, ay * bot + by, ay * top + by]          # xmin, xmax, ymin, ymax by the axes
        assert abs((extent[1] - extent[0]) - ncol * round((extent[1] - extent[0]) / ncol)) < 0.25 and \
            abs((extent[3] - extent[2]) - nrow * round((extent[3] - extent[2]) / nrow)) < 0.25, f"{name}: frame is not the raster's extent {extent}"
        idx = np.full((nrow, ncol), -1, dtype=np.int32)
        worst = 0
        for r in range(nrow):
            for c in range(ncol):
                # the frame lines are the raster's edges (checked against the axis calibration below): row 0 is the top row
                x0, x1 = left + (c + 0.25) * cw, left + (c + 0.75) * cw
                y0, y1 = top + (r + 0.25) * ch, top + (r + 0.75) * ch
                xa, xb = int(np.ceil(x0)), int(np.floor(x1))
                ya, yb = int(np.ceil(y0)), int(np.floor(y1))
                if xb < xa or yb < ya:                                  # cells of a few pixels: the centre pixel
                    xa = xb = int(round(left + (c + 0.5) * cw))
                    ya = yb = int(round(top + (r + 0.5) * ch))
                px = img[ya:yb + 1, xa:xb + 1].reshape(-1, 3).astype(np.int32)
                vals, counts = np.unique(px, axis=0, return_counts=True)
                col = vals[counts.argmax()]                             # cells are flat-coloured: the commonest colour
                if (col > 245).all():
                    continue                                           # white: NA
                d = np.abs(cols_rgb - col).sum(axis=1)
                k = int(d.argmin())
                worst = max(worst, int(d[k]))
                idx[r, c] = k
        # (a legend shorter than the palette skips colours: a cell may then sit one palette step from the nearest legend row)
        assert worst <= (6 if len(ramp) >= 300 else 16), f"{name}: a cell colour is {worst} away from every legend colour"
        panels.append({
            "what": ps["what"], "frame_px": {"top": top, "bottom": bot, "left": left, "right": right},
            "legend_px": {"top": bt, "bottom": bb, "left": bl, "right": br}, "extent_by_axes": [round(v, 3) for v in extent],
            "x": {"tick_labels": ps["xticks"], "per_px": ax, "at_px0": bx, "fit_resid_px": rx},
            "y": {"tick_labels": ps["yticks"], "per_px": ay, "at_px0": by, "fit_resid_px": ry},
            "legend": {"tick_px": lpix[::-1], "tick_labels": ps["legend"], "per_px": al, "at_px0": bl0, "fit_resid_px": rl,
                       "min": float(al * (bb - 0.5) + bl0) if al < 0 else float(al * (bt + 0.5) + bl0),
                       "max": float(al * (bt + 0.5) + bl0) if al < 0 else float(al * (bb - 0.5) + bl0)},
            "classes": [[round(c["lo"], 6), round(c["hi"], 6)] for c in classes],
            "cells": idx.tolist(), "worst_colour_distance": worst,
        })
    return {"source": f"vignettes/images/{name}.png", "rmd_lines": spec["rmd"], "panels": panels}


def main():
    if not IMAGES.exists():
        raise SystemExit("the reference's vignette images are not here: this script runs in the build container only")
    only = sys.argv[1:]
    out = {"_about": "written by tools/digitize_vignette.py from the reference's published figures; columns = [x pixel, [first row, "
                     "last row, ...]] = the runs of pixels of the curve's colour in that pixel column (rows count downwards); "
                     "value(pixel) = per_px * pixel + at_px0 on each axis; one pixel (`px`) is the digitisation error",
           "figures": {}}
    for name, spec in SPEC.items():
        if only and name not in only:
            continue
        out["figures"][name] = digitize(name, spec)
        f = out["figures"][name]
        for k, p in enumerate(f["panels"]):
            print(f"{name} panel {k}: frame {p['frame_px']}, 1 px = {p['px']['x']:.4g} ({p['x']['is']}) x {p['px']['y']:.4g} "
                  f"({p['y']['is']}); fit residual {p['x']['fit_resid_px']:.2f} / {p['y']['fit_resid_px']:.2f} px; "
                  + ", ".join(f"{c}: {len(v['columns'])} columns" for c, v in p["curves"].items()))
    out["_about_maps"] = ("maps: terra::plot rasters; `classes` = [lo, hi] of every legend colour in data units (value = per_px * "
                          "pixel row + at_px0 on the legend), `cells` = [row][col] class index of the raster cell drawn there (row 0 "
                          "= northernmost), -1 = NA; a cell's value is known to the width of its class")
    out["maps"] = {}
    for name, spec in MAPS.items():
        if only and name not in only:
            continue
        out["maps"][name] = digitize_map(name, spec)
        for k, p in enumerate(out["maps"][name]["panels"]):
            w = np.median([b - a for a, b in p["classes"]])
            print(f"{name} map {k}: legend {p['legend']['min']:.5g} .. {p['legend']['max']:.5g}, {len(p['classes'])} colour classes of "
                  f"{w:.3g}, fit residual {p['legend']['fit_resid_px']:.2f} px, {sum(v < 0 for r in p['cells'] for v in r)} NA cells")
    if not only:
        OUT.write_text(json.dumps(out, separators=(",", ":")))
        print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
