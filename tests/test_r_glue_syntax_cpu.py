"""SYNTAX CHECK of the R `.Call` shim r/mcfhip_glue.c — not parity evidence, pins nothing about the reference.

R is not installed in the build image, so the 500-line glue can never be compiled against the real headers there.
This test keeps it from rotting: `gcc -fsyntax-only -Werror` against declarations-only stand-ins of R.h / Rinternals.h /
R_ext/Rdynload.h (tests/r_api_decls/, written from R's documented C API) and the real include/mcf.h — so every mcf_*
call in the glue is type-checked against the ABI, and the `_Static_assert`s on the struct layouts the fill loops rely on
are evaluated.  Also: the registration table lists every `.Call` name r/mcfhip_overrides.R uses, with the arity of
the C definition and of the R call (the reference's table: src/RcppExports.cpp:835-891)."""
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
GLUE = ROOT / "r" / "mcfhip_glue.c"
OVERRIDES = ROOT / "r" / "mcfhip_overrides.R"


def test_glue_passes_a_syntax_and_type_check():
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I", str(ROOT / "tests" / "r_api_decls"), "-I", str(ROOT / "include"), str(GLUE)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def _call_entries(src):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(mcfhip_\w+)",\s*\(DL_FUNC\)&\1,\s*(\d+)\}', src)}


def _c_arity(src, name):
    m = re.search(r"\bSEXP\s+" + name + r"\s*\(([^)]*)\)\s*\{", src)
    assert m, f"{name} is registered but not defined"
    return len(re.findall(r"\bSEXP\b", m.group(1)))


def test_registration_table_matches_definitions_and_r_calls():
    src = GLUE.read_text()
    entries = _call_entries(src)
    assert len(entries) >= 14
    for name, n in entries.items():
        assert _c_arity(src, name) == n, name
    rsrc = OVERRIDES.read_text()
    # literal .Call("mcfhip_x", a, b, ...) sites
    used = {}
    for m in re.finditer(r'\.Call\(\s*"(mcfhip_\w+)"\s*,([^)]*)\)', rsrc):
        used[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    # .Call(sym, ...) sites inside the `for (nm in c(...))` loops: sym = paste0("mcfhip_", nm | sub("Cpp$", "", nm))
    for m in re.finditer(r'for \(nm in c\(([^)]*)\)\) local\(\{(.*?)\}\)', rsrc, re.S):
        names = re.findall(r'"(\w+)"', m.group(1))
        call = re.search(r"\.Call\(sym,([^)]*)\)", m.group(2))
        assert call
        nargs = len([a for a in call.group(1).split(",") if a.strip()])
        for nm in names:
            used["mcfhip_" + re.sub(r"Cpp$", "", nm)] = nargs
    assert used, "no .Call sites found"
    for name, nargs in used.items():
        assert name in entries, f"{name} is called from R but not registered"
        assert entries[name] == nargs, f"{name}: R passes {nargs} arguments, the table says {entries[name]}"


def test_layout_asserts_cover_every_pointer_cast():
    src = GLUE.read_text()
    casts = set(re.findall(r"\(const double \*\*\)&in->(\w+)", src))
    guarded = {"vegp", "soilc", "clim", "pointm"}
    assert casts <= guarded, casts - guarded
    for t in ("mcf_vegp", "mcf_soilc", "mcf_snow_climate", "mcf_snow_pointm", "mcf_snow_vegp"):
        assert f"MCF_PTR_STRUCT({t}," in src
