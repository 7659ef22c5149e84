#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop, from hipcc's gfx950 assembly (runs in the build container, no GPU).

    python tools/isa_mix.py                       # k_solve<21,0,false>, the vector-forcing solver
    python tools/isa_mix.py --kernel 'k_solve<32,1,false>' --out profiles/r02_isa_mix_af.txt

Compiles microclimf_amd/csrc/mcf_kernels.hip (or --src) to assembly with the Makefile's flags, takes the named kernel,
finds its day loop (the backward branch that spans the most instructions) and counts opcodes by class.  The count is
STATIC: every instruction of the loop body once — i.e. the path of a lane that takes every branch (a daytime step of a
vegetated below-canopy cell).  rocprofv3's SQ_INSTS_VALU per cell-step (profiles/*_pmc_summary.json) is the dynamic
counterpart, averaged over day and night."""
import argparse
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "microclimf_amd" / "csrc"


def hipflags():
    mk = (CSRC / "Makefile").read_text()
    m = re.search(r"^HIPFLAGS \?= (.*?)(?<!\\)\n", mk, re.S | re.M)
    flags = m.group(1).replace("\\\n", " ").replace("$(ARCH)", "gfx950").split()
    return [f for f in flags if f != "-fPIC"]


def assemble(src: Path, extra):
    out = Path("/tmp") / (src.stem + ".isa_mix.s")
    cmd = ["/opt/rocm/bin/hipcc", *hipflags(), *extra, "-S", "--cuda-device-only", "-o", str(out), str(src)]
    subprocess.run(cmd, check=True, cwd=str(CSRC), stderr=subprocess.DEVNULL)
    return out.read_text()


def mangle_match(name, sym):
    """'k_solve<21,0,false>' against _ZN3mcf7k_solveILi21ELi0ELb0EEEv..."""
    m = re.match(r"(\w+)(?:<(.*)>)?$", name)
    base, targs = m.group(1), m.group(2)
    if f"{len(base)}{base}" not in sym:
        return False
    if targs is None:
        return True
    enc = ""
    for t in targs.split(","):
        t = t.strip()
        enc += {"true": "Lb1E", "false": "Lb0E"}.get(t, f"Li{t}E")
    return f"I{enc}E" in sym


CLASSES = [
    ("fp64 fma/mul/add", r"v_(fma|fmac|mul|add)_f64"),
    ("fp64 min/max", r"v_(min|max)_f64"),
    ("fp64 rcp/rsq/sqrt (quarter rate)", r"v_(rcp|rsq|sqrt)_f64"),
    ("fp64 rndne/cvt/ldexp/frexp/trunc/floor", r"v_(rndne|cvt_\w+|ldexp|frexp_\w+|trunc|floor|ceil|fract)_f64|v_cvt_f64_\w+|v_cvt_\w+_f64"),
    ("fp64 div helpers (div_scale/fmas/fixup)", r"v_div_\w+_f64"),
    ("v_cmp*", r"v_cmpx?_\w+"),
    ("v_cndmask", r"v_cndmask_b32"),
    ("v_mov", r"v_mov_b(32|64)|v_accvgpr_\w+"),
    ("lane ops (readlane/writelane/bpermute/dpp)", r"v_(readlane|writelane|readfirstlane)_b32|ds_bpermute_b32|ds_permute_b32|v_permlane\w*"),
    ("integer / bit VALU", r"v_(and|or|xor|not|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|add|sub|subrev|mul|mad|bfe|bfi|add3|lshl_add|"
                           r"add_lshl|lshl_or|and_or|or3|mul_lo|mul_hi|mad_u64|addc|subb|subbrev|min|max|med3|alignbit|perm|cvt|mbcnt_lo|mbcnt_hi)_[a-z0-9_]+"),
    ("LDS (ds_read/ds_write)", r"ds_\w+"),
    ("global / scratch memory", r"(global|flat|buffer|scratch)_\w+"),
    ("SALU s_mov", r"s_mov_b(32|64)"),
    ("SALU other", r"s_(?!waitcnt|nop|barrier|cbranch|branch|endpgm|setprio|sleep|mov_b)\w+"),
    ("s_cbranch/s_branch", r"s_c?branch\w*"),
    ("s_waitcnt", r"s_waitcnt\w*"),
    ("s_nop", r"s_nop"),
    ("s_barrier", r"s_barrier"),
]
VALU_CLASSES = {c for c, _ in CLASSES[:10]}


def classify(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    for name, pat in CLASSES:
        if re.fullmatch(pat, op):
            return name
    return "other: " + op


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_solve<21,0,false>")
    ap.add_argument("--src", default=str(CSRC / "mcf_kernels.hip"))
    ap.add_argument("--out", default="")
    ap.add_argument("-D", action="append", default=[], help="extra -D defines")
    a = ap.parse_args()
    text = assemble(Path(a.src), [f"-D{d}" for d in a.D])
    lines = text.splitlines()
    start = end = None
    sym = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and mangle_match(a.kernel, m.group(1)):
            start, sym = i, m.group(1)
        if start is not None and end is None and ln.strip().startswith(".Lfunc_end") and i > start:
            end = i
            break
    if start is None:
        sys.exit(f"kernel {a.kernel} not found")
    body = lines[start:end]
    ins, labels, headers = [], {}, set()
    last_label = None
    for ln in body:
        s = ln.split(";")[0].strip()
        if not s:
            if ("Loop Header" in ln or "in Loop:" in ln) and last_label:      # (the annotation can sit on its own line)
                headers.add(last_label)
            continue
        m = re.match(r"^(\.L\w+):$", s)
        if m:
            labels[m.group(1)] = len(ins)
            last_label = m.group(1)
            if "Loop Header" in ln or "in Loop:" in ln:          # LLVM's own annotation: a real loop, not an exec-mask waterfall
                headers.add(m.group(1))
            continue
        if s.startswith("."):
            continue
        ins.append(s)
    # the day loop: the backward branch spanning the most instructions
    best = (0, 0, 0)
    for i, s in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.L\w+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i and (not headers or m.group(1) in headers):
            span = i - labels[m.group(1)]
            if span > best[0]:
                best = (span, labels[m.group(1)], i)
    _, lo, hi = best
    loop = ins[lo:hi + 1]
    cnt = collections.Counter(classify(s.split()[0]) for s in loop)
    ops = collections.Counter(s.split()[0] for s in loop)
    meta = {}
    for key in ("num_vgpr", "numbered_sgpr", "private_seg_size"):
        m = re.search(r"\.set " + re.escape(sym) + r"\." + key + r", (\d+)", text)
        meta[key] = int(m.group(1)) if m else None
    m = re.search(r"\.amdhsa_kernel " + re.escape(sym) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
    lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", m.group(1)) if m else None
    spills = len([s for s in loop if re.match(r"v_(readlane|writelane)_b32", s)])
    out = []
    out.append(f"kernel {a.kernel}  ({sym})")
    out.append(f"VGPRs {meta['num_vgpr']}  SGPRs {meta['numbered_sgpr']}  scratch {meta['private_seg_size']} B  "
               f"LDS {lds.group(1) if lds else '?'} B  |  whole kernel {len(ins)} instructions, day loop {len(loop)}")
    valu = sum(n for c, n in cnt.items() if c in VALU_CLASSES)
    out.append(f"day loop, static (every branch taken once): {valu} VALU, {spills} of them SGPR-spill lane moves")
    out.append("")
    out.append(f"{'class':48s} {'count':>6s}  {'% of loop':>9s}")
    for name, _ in CLASSES:
        if cnt.get(name):
            out.append(f"{name:48s} {cnt[name]:6d}  {100.0 * cnt[name] / len(loop):8.1f}%")
    for name, n in sorted(cnt.items()):
        if name.startswith("other"):
            out.append(f"{name:48s} {n:6d}")
    out.append("")
    out.append("top opcodes: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(28)))
    res = "\n".join(out)
    print(res)
    if a.out:
        Path(a.out).write_text(res + "\n")


if __name__ == "__main__":
    main()
