"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/mcf.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from microclimf_amd import _abi, synthetic

ROOT = Path(__file__).resolve().parent.parent


def _build():
    import __graft_entry__ as g
    g.build_library()


def test_header_symbols_are_exported():
    _build()
    hdr = (ROOT / "include" / "mcf.h").read_text()
    declared = set(re.findall(r"\b(mcf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    lib = C.CDLL(str(_abi.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"libmcfhip.so does not export {name}"
    assert lib.mcf_abi_version() == _abi.ABI_VERSION == int(re.search(r"#define MCF_ABI_VERSION (\d+)", hdr).group(1))


def test_struct_layout_matches_header():
    # sizes implied by include/mcf.h on LP64
    assert C.sizeof(_abi.Obstime) == 4 * 8
    assert C.sizeof(_abi.Climate) == 10 * 8
    assert C.sizeof(_abi.Pointm) == 8 * 8
    assert C.sizeof(_abi.Vegp) == 10 * 8
    assert C.sizeof(_abi.Soilc) == 15 * 8
    assert C.sizeof(_abi.GridInputs) == 3 * 8 + 8 + (4 + 10 + 8 + 10 + 15) * 8 + 2 * 8 + 2 * 8 + 2 * 8 + 8 + 4 * 8 + 8 + 2 * 8 + 8     # + row_pitch


def test_ctypes_structs_agree_with_the_compiled_header(tmp_path):
    """sizeof / offsetof from gcc on include/mcf.h against the ctypes mirrors"""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    probes = {"mcf_grid_inputs": (_abi.GridInputs, ["tsteps", "array_forcing", "veg_layers", "clim", "soilc", "lat", "lats",
                                                    "lyr_ed", "coarse_rows", "coarse_cols", "coarse_rowpos", "coarse_winddir",
                                                    "coarse_altcorrect", "coarse_dtm", "fine_dtm"]),
              "mcf_options": (_abi.Options, ["tfact", "complete", "out", "device", "cells_per_block"]),
              "mcf_nc_spec": (_abi.NcSpec, ["nsteps", "east", "crs_wkt", "reqhgt", "vars", "reference_puts_only", "format", "deflate_level"]),
              "mcf_multi": (_abi.Multi, ["n_devices", "devices", "n_blocks"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mcf.h"', 'int main(void) {']
    for name, (_, fields) in probes.items():
        src.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in fields:
            src.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src.append('return 0; }')
    (tmp_path / "probe.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-I", str(root / "include"), "-o", str(tmp_path / "probe"), str(tmp_path / "probe.c")], check=True)
    out = subprocess.run([str(tmp_path / "probe")], check=True, capture_output=True, text=True).stdout.split("\n")
    got = dict(line.split() for line in out if line)
    for name, (cls, fields) in probes.items():
        assert int(got[name]) == C.sizeof(cls), name
        for f in fields:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"
    assert C.sizeof(_abi.Options) == 6 * 8 + 4 + 40 + 3 * 4
    assert C.sizeof(_abi.Outputs) == 80


def test_no_cpu_fallback_without_device():
    _build()
    lib = _abi.load()
    if lib.mcf_device_count() > 0:
        pytest.skip("a GPU is present")
    from microclimf_amd.api import runmicro1Cpp
    a = synthetic.workload(4, 4, 24)
    with pytest.raises(_abi.McfError, match="no HIP device"):
        runmicro1Cpp(**a)
    # ... nor through the one-process multi-device entries (solver, terrain pre-compute, snow chunk loop, bioclim sink)
    with pytest.raises(_abi.McfError, match="no HIP device"):
        runmicro1Cpp(**a, devices=[0], n_blocks=2)
    from microclimf_amd.terrain import precompute_terrain
    with pytest.raises(_abi.McfError, match="no HIP device"):
        precompute_terrain(np.zeros((8, 8)), 1.0, 2.0, devices=[0], n_blocks=2)
    from microclimf_amd.snow import snowmodel1_chunks
    sw = synthetic.snow_workload(6, 5, 120, cold=3.0, zref=3.5)
    with pytest.raises(_abi.McfError, match="no HIP device"):
        snowmodel1_chunks(sw["obstime"], sw["climdata"], sw["pointm"], sw["vegp"], sw["other"], sw["snowenv"],
                          np.zeros((6, 5)), 1.0, 0.02, devices=[0], n_blocks=2)
    from microclimf_amd.api import runbioclim1Cpp
    b = synthetic.workload(4, 4, 336)
    for k in ("complete", "out"):
        b.pop(k)
    with pytest.raises(_abi.McfError, match="no HIP device"):
        runbioclim1Cpp(**b, out=[1] * 19, wetq=[0], dryq=[0], hotq=[0], colq=[0], air=True, devices=[0], n_blocks=2)


def test_marshal_rejects_bad_shapes():
    from microclimf_amd.marshal import marshal
    a = synthetic.workload(4, 4, 24)
    a["vegp"] = dict(a["vegp"], pai=np.zeros((3, 4)))
    with pytest.raises(ValueError):
        marshal(**a, array_forcing=False)
