#!/usr/bin/env python3
"""Static VALU cost of a kernel's day loop by SOURCE LINE (hipcc -gline-tables-only assembly; build container, no GPU).

    python tools/isa_lines.py --kernel 'k_solve<21, 0, false, true, true, false>' [--top 40] [--bucket 10]

Cost units: an fp64 VALU instruction 2 (4 cycles per wave on a 16-lane fp64 pipe), quarter-rate fp64 (rcp/rsq/sqrt) 8,
any other VALU instruction 1 — per wave64 instruction.  Static: every instruction of the loop body once."""
import argparse
import collections
import re
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import isa_mix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_solve<21, 0, false, true, true, false>")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--bucket", type=int, default=1, help="group source lines into buckets of this many lines")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--src", default=str(isa_mix.CSRC / "mcf_kernels.hip"))
    a = ap.parse_args()
    src = Path(a.src)
    out = Path("/tmp/isa_lines.s")
    cmd = ["/opt/rocm/bin/hipcc", *isa_mix.hipflags(), *[f"-D{d}" for d in a.D], "-gline-tables-only", "-S", "--cuda-device-only",
           "-o", str(out), str(src)]
    subprocess.run(cmd, check=True, cwd=str(isa_mix.CSRC), stderr=subprocess.DEVNULL)
    s = out.read_text()
    files = {}
    for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    sym = None
    for m in re.finditer(r"^(_Z\w+):", s, re.M):
        if isa_mix.mangle_match(a.kernel.replace(" ", ""), m.group(1)):
            sym = m.group(1)
            break
    if sym is None:
        raise SystemExit("kernel not found")
    i = s.index(sym + ":")
    j = s.index(".Lfunc_end", i)
    rows = []            # (index, opcode, loc, label?)
    cur = None
    labels = {}
    headers = set()
    last_label = None
    for line in s[i:j].split("\n"):
        t = line.strip()
        if t.startswith(".loc"):
            p = t.split()
            cur = (files.get(int(p[1]), p[1]), int(p[2]))
        elif re.match(r"^\.LBB\w+:", t):
            labels[t.split(":")[0]] = len(rows)
            last_label = t.split(":")[0]
            if "Loop Header" in t or "in Loop:" in t:           # LLVM's own annotation: a real loop, not an exec-mask waterfall
                headers.add(last_label)
        elif t.startswith(";") and ("Loop Header" in t or "in Loop:" in t) and last_label:   # (the annotation can sit on its own line)
            headers.add(last_label)
        elif line.startswith("\t") and not t.startswith((".", ";")):
            rows.append((t.split()[0], t, cur))
    # the day loop: the backward branch (to a loop header) spanning the most instructions
    best = (0, 0, 0)
    for n, (op, t, _) in enumerate(rows):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < n and n - labels[tgt] > best[0] and (not headers or tgt in headers):
                best = (n - labels[tgt], labels[tgt], n)
    _, lo, hi = best
    cost = collections.Counter()
    count = collections.Counter()
    for op, t, loc in rows[lo:hi + 1]:
        if not op.startswith("v_"):
            continue
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            c = 8
        elif op.endswith("_f64") or "_f64_" in op:
            c = 2
        else:
            c = 1
        key = (loc[0], loc[1] // a.bucket * a.bucket) if loc else ("?", 0)
        cost[key] += c
        count[key] += 1
    total = sum(cost.values())
    print(f"{a.kernel}: day loop {hi - lo + 1} instructions, VALU cost {total} units ({sum(count.values())} VALU instructions)")
    for key, c in cost.most_common(a.top):
        print(f"  {key[0]}:{key[1]:<5d} {c:5d} units {100.0 * c / total:5.1f} %  ({count[key]} instr)")


if __name__ == "__main__":
    main()
