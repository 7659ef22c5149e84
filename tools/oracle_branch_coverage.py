#!/usr/bin/env python3
"""Branch coverage of the CPU oracle under the parity workloads (tests/parity_cases.py).

Builds oracle/oracle_unit.c with gcov instrumentation (-O0 --coverage), runs every parity case
(plus the time-varying-vegetation and reference-test replays) through it and reports, per source
line of mcf_oracle.c, the conditional branches that were only ever taken one way.  Used by
tests/test_branch_coverage_cpu.py; run directly for the full listing."""
from __future__ import annotations

import ctypes as C
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def measure(workdir: Path):
    import parity_cases as P
    from microclimf_amd import synthetic
    from oracle import oracle as O

    src = ROOT / "oracle" / "oracle_unit.c"
    lib_path = workdir / "libcov.so"
    subprocess.run(["gcc", "-O0", "--coverage", "-DORC_COVERAGE", "-fPIC", "-shared", "-std=c99", "-o", str(lib_path), str(src), "-lm"],
                   check=True, cwd=workdir)
    lib = C.CDLL(str(lib_path))
    for name in P.CASES:
        a, af = P.build(name)
        O.run_grid(**a, array_forcing=af, lib=lib)
    a = synthetic.layered(synthetic.workload(6, 5, 96, reqhgt=0.05, variety=True, start_doy=150), 3, cover_days=3)
    O.run_grid(**a, lib=lib)
    lib.orc_cov_dump()
    subprocess.run(["gcov", "-b", "-c", "-o", str(workdir / "libcov.so-oracle_unit.gcno"), str(src)],
                   check=True, cwd=workdir, capture_output=True)
    return parse(workdir / "mcf_oracle.c.gcov")


def measure_snow(workdir: Path):
    """The same for oracle/snow_oracle.c under tests/snow_cases.py (snowpack model, snow microclimate at
    several sensor heights) and the replayed test-pointmodelsnow.R."""
    import snow_cases as SC
    from microclimf_amd import synthetic
    from oracle import oracle as O
    from oracle import replay_reference_tests as RT

    src = ROOT / "oracle" / "oracle_unit.c"
    lib_path = workdir / "libcovs.so"
    subprocess.run(["gcc", "-O0", "--coverage", "-DORC_COVERAGE", "-fPIC", "-shared", "-std=c99", "-o", str(lib_path), str(src), "-lm"],
                   check=True, cwd=workdir)
    lib = C.CDLL(str(lib_path))
    for name in SC.SNOW_CASES:
        sw, af = SC.build_snow(name)
        smod = O.run_snowmodel(**SC.model_args(sw), array_forcing=af, lib=lib)
        snowm, micro = SC.microsnow_state(sw, smod)
        for reqhgt in SC.MICRO_HEIGHTS:
            O.run_microsnow(reqhgt, sw["obstime"], sw["climdata"], snowm, micro, sw["vegp"], sw["other"], 3.0, [1] * 10,
                            array_forcing=af, lib=lib)
    lib.orc_satvap.restype = C.c_double
    lib.orc_satvap.argtypes = [C.c_double]
    saved = O._lib
    O._lib = lib                                   # the replay goes through oracle.load()
    try:
        RT.replay_pointmodelsnow_test()
    finally:
        O._lib = saved
    lib.orc_cov_dump()
    subprocess.run(["gcov", "-b", "-c", "-o", str(workdir / "libcovs.so-oracle_unit.gcno"), str(src)],
                   check=True, cwd=workdir, capture_output=True)
    return parse(workdir / "snow_oracle.c.gcov")


def parse(gcov_file: Path):
    """-> (lines_with_branches, one_sided): one_sided = [(lineno, source, [counts])] for executed
    lines where some branch outcome was never taken."""
    total, one_sided = 0, []
    cur = None
    for raw in gcov_file.read_text().splitlines():
        m = re.match(r"\s*([0-9#=\-]+)\*?:\s*(\d+):(.*)", raw)
        if m:
            if cur and cur["br"]:
                total += 1
                if cur["exec"] and any(c == 0 for c in cur["br"]):
                    one_sided.append((cur["no"], cur["src"].strip(), cur["br"]))
            cnt = m.group(1)
            cur = {"no": int(m.group(2)), "src": m.group(3), "br": [],
                   "exec": cnt not in ("-", "#####", "=====")}
            continue
        b = re.match(r"branch\s+\d+\s+(taken (\d+)|never executed)", raw)
        if b and cur is not None:
            cur["br"].append(int(b.group(2)) if b.group(2) else 0)
    if cur and cur["br"]:
        total += 1
        if cur["exec"] and any(c == 0 for c in cur["br"]):
            one_sided.append((cur["no"], cur["src"].strip(), cur["br"]))
    return total, one_sided


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        total, one = measure(Path(d))
    print(f"{total} source lines with conditional branches, {len(one)} taken one way only:")
    for no, src, br in one:
        print(f"  mcf_oracle.c:{no}: {src[:110]}   {br}")
    with tempfile.TemporaryDirectory() as d:
        total, one = measure_snow(Path(d))
    print(f"snow_oracle.c: {total} source lines with conditional branches, {len(one)} taken one way only:")
    for no, src, br in one:
        print(f"  snow_oracle.c:{no}: {src[:110]}   {br}")
