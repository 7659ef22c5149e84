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


def tick_pixels(img, box, side):
    """Centres of the tick strokes outside the frame: side 'left' -> rows, 'bottom' -> columns."""
    dk = colour_mask(img, "black")
    top, bot, left, right = box
    if side == "left":
        strip = dk[:, left - 5:left - 2].all(axis=1)        # a tick covers these columns in full; label glyphs are further out
        strip[:max(top - 2, 0)] = False
        strip[bot + 3:] = False
    else:
        strip = dk[bot + 3:bot + 6, :].all(axis=0)
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
    if not only:
        OUT.write_text(json.dumps(out, separators=(",", ":")))
        print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
