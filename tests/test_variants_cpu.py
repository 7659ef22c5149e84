"""The timing / ablation variants of k_solve live as patches under tools/variants/ (tools/make_variant_patches.py writes them,
tools/build_variant.sh applies them to a scratch copy): they must keep applying to the source that ships, and that source
must carry no experiment switch of its own."""
import re
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "microclimf_amd" / "csrc"


def test_every_variant_patch_applies_to_the_shipped_kernel_source(tmp_path):
    patches = sorted((ROOT / "tools" / "variants").glob("*.patch"))
    assert len(patches) >= 6
    for p in patches:
        d = tmp_path / p.stem / "microclimf_amd" / "csrc"
        d.mkdir(parents=True)
        for f in ("mcf_kernels.hip", "mcf_device.hpp"):
            shutil.copy(CSRC / f, d / f)
        r = subprocess.run(["patch", "-s", "-p1", "--dry-run", "-i", str(p)], cwd=tmp_path / p.stem, capture_output=True, text=True)
        assert r.returncode == 0, f"{p.name} no longer applies (run tools/make_variant_patches.py): {r.stdout}{r.stderr}"


def test_the_hot_kernel_source_has_no_experiment_switches():
    for f in ("mcf_kernels.hip", "mcf_device.hpp"):
        src = (CSRC / f).read_text()
        assert not re.search(r"MCF_EXPERIMENT|MCF_PERSISTENT_TILES|MCF_NT_STORES|MCF_DAYPRIO|MCF_STORES_AFTER_STAGE", src), f
        assert "template <int CPB, int AF, bool BG, bool F, bool SSREQ, bool PT>" not in src
