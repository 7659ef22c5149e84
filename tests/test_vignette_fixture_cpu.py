"""tests/golden/vignette_points.json — the reference's published result figures, digitised by tools/digitize_vignette.py.
Here (no GPU): the fixture is self-consistent, regenerates bit for bit where the reference's images are present (the build
container), and the one published series that needs no GPU — the host-side point model of image1b — meets it.  The grid
solver's and the snow branch's curves — and the ten published raster maps, cell by cell — are compared in
tests/test_frontend_gpu.py::test_vignette_*."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import vignette_fixture as V
from bundled import load
from microclimf_amd import frontend as F

ROOT = Path(__file__).resolve().parent.parent
IMAGES = Path("/root/reference/vignettes/images")


def test_fixture_is_self_consistent():
    fx = json.loads(V.FIXTURE.read_text())["figures"]
    assert set(fx) == {"image1b", "image7", "image8", "image9", "image14a", "image14p", "image14b"}
    for name, f in fx.items():
        for p in f["panels"]:
            for ax in ("x", "y"):
                a = p[ax]
                assert a["fit_resid_px"] <= 0.75, (name, ax)            # printed labels sit on the detected ticks
                back = (np.array(a["tick_labels"]) - a["at_px0"]) / a["per_px"]
                assert np.abs(back - np.array(a["tick_px"])).max() <= 0.75
            fr = p["frame_px"]
            for c in p["curves"].values():
                assert c["columns"], name
                for x, flat in c["columns"]:
                    assert fr["left"] < x < fr["right"] and len(flat) % 2 == 0
                    assert all(fr["top"] < r < fr["bottom"] for r in flat)
    # the digitisation error per figure: one pixel in data units
    assert abs(fx["image9"]["panels"][0]["px"]["y"] - 0.0503) < 1e-3          # degC
    assert abs(fx["image14a"]["panels"][0]["px"]["y"] - 1.30) < 0.01          # mm of snow water equivalent


def test_map_fixtures_are_self_consistent():
    """the digitised raster maps: 50 x 50 cells each, the bundled site's 128 no-data cells white, the printed legend labels
    on the detected tick strokes, colour classes that tile the legend without gaps, every cell matched to a legend colour"""
    _, _, _, dtm = load()
    na = np.isnan(np.asarray(dtm["z"], dtype=float))
    mp = json.loads(V.FIXTURE.read_text())["maps"]
    assert set(mp) == {"image1a", "image2", "image3b", "image4", "image5", "image6", "image10", "image11"}
    assert sum(len(f["panels"]) for f in mp.values()) == 10
    for name, f in mp.items():
        for k, p in enumerate(f["panels"]):
            idx = np.array(p["cells"])
            assert idx.shape == (50, 50) and np.array_equal(idx < 0, na), (name, k)
            cls = np.array(p["classes"])
            assert idx.max() < len(cls) and (cls[:, 1] > cls[:, 0]).all()
            order = np.argsort(cls[:, 0])
            assert np.abs(cls[order][1:, 0] - cls[order][:-1, 1]).max() < 1e-5, (name, k)      # classes tile the legend
            for a in (p["x"], p["y"], p["legend"]):
                assert a["fit_resid_px"] <= 0.75, (name, k)
            assert p["worst_colour_distance"] <= 16
            m = V.map_panel(name, k)
            assert np.nanmin(m["lo"]) >= p["legend"]["min"] - 1e-6 and np.nanmax(m["hi"]) <= p["legend"]["max"] + 1e-6
    # the resolution this buys, in data units: one colour class
    assert abs(V.map_panel("image6")["class_width"] - 0.097) < 0.002          # degC
    assert abs(V.map_panel("image4")["class_width"] - 0.0100) < 0.0005        # m/s


@pytest.mark.skipif(not IMAGES.exists(), reason="the reference's images only exist in the build container")
def test_fixture_regenerates_from_the_reference_images(tmp_path):
    sys.path.insert(0, str(ROOT / "tools"))
    import digitize_vignette as D
    fx = json.loads(V.FIXTURE.read_text())
    for name, spec in D.SPEC.items():
        assert D.digitize(name, spec) == fx["figures"][name], name
    for name, spec in D.MAPS.items():
        assert json.loads(json.dumps(D.digitize_map(name, spec))) == fx["maps"][name], name


def test_point_model_series_meet_the_published_figure():
    """vignettes/images/image1b.png (running-microclimf.Rmd:283-292): ground and canopy temperature of `runpointmodel` over
    2017 — host code, so this pin needs no GPU.  1 px = 18.2 h x 0.115 degC; both directions within 2 px."""
    weather, vegp, soilc, dtm = load()
    dfo = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)["dfo"]
    hours = np.arange(len(dfo["Tg"]), dtype=float)
    d = V.distances(V.panel("image1b"), "all", np.concatenate([hours, [np.nan], hours]),
                    np.concatenate([dfo["Tg"], [np.nan], dfo["Tc"]]))
    assert d["fig_to_model_max"] < 2.0 and d["model_to_fig_max"] < 2.0, (d["fig_to_model_max"], d["model_to_fig_max"])
