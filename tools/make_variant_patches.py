#!/usr/bin/env python3
"""Regenerates tools/variants/*.patch against the current microclimf_amd/csrc/mcf_kernels.hip.

The timing / ablation variants of k_solve (results wrong on purpose — only the launch time is read) are kept OUT of the
shipped source: each is a list of (old, new) text replacements, written here as a unified diff that
tools/build_variant.sh applies to a scratch copy of csrc/.  Run this after editing the kernel (the diffs carry context)."""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "microclimf_amd" / "csrc" / "mcf_kernels.hip"
OUT = ROOT / "tools" / "variants"

P1 = '''            if (AF) pass1<F, false>(C, TR, SL, g, flags, dTcap, cy, p1, MK, cn);
            else pass1<F, SS>(C, TL, SL, g, flags, dTcap, cy, p1, MK, cn);
'''
P2 = '''            if (AF) pass2<F, false>(C, TR, SL, g, flags, dTcap, cy, dtr, Rmx, need_tv, p2, MK, cn);
            else pass2<F, SS>(C, TL, SL, g, flags, dTcap, cy, dtr, Rmx, need_tv, p2, MK, cn);
'''
ST = '''                asm("" : "+v"(posb));
                *(double*)((char*)ring_day + ((size_t)sel * (NT * 8)) + posb) = val;'''
BAR = '''        __syncthreads();
        if (tid < 3 * CPB)      // reset the buffer of the day after next (see s_ext)'''
LOOP = '''    for (int dl = 0; dl < ndays; ++dl, ++run) {
        const int dabs = day0 + dl;'''
SECT_GLOBAL = ('''namespace mcf {

// ------------------------------------------------------------------------------------
__global__ void k_fill(''', '''namespace mcf {
// TIMING VARIANT: shader-clock cycles per wave, by hour group (hour / 3): per day {pass 1 + its stores, wait at the day
// barrier, soil producer + pass 2 + its stores, whole day}, and per tile the prologue (entry -> top of the first day), split
// at the barrier behind the staging loads.  Summed in registers over the launch's days; one tile in 16 reports (a few
// atomics per wave: the atomics must not become the load).
__device__ unsigned long long g_sect[8 * 8];

// ------------------------------------------------------------------------------------
__global__ void k_fill(''')

VARIANTS = {
    # the launch's stores with no physics in front of them
    "storeonly": [(P1, "            cy.soilm = cy.Rbdown = cy.Rddown = p1.uz = p1.Rdup = p1.Tg0 = p1.absRnet = (double)dl;   // TIMING VARIANT: no physics\n"),
                  (P2, "            p2.Tz = p2.Tg = p2.tleaf = p2.rh = p2.lwdn = p2.lwup = dtr + Rmx;   // TIMING VARIANT: no physics\n")],
    # every store elided (the compiler cannot prove the predicate false)
    "nostore": [(ST, '''                asm("" : "+v"(posb));
                if (val == 1.2345e300) *(double*)((char*)ring_day + ((size_t)sel * (NT * 8)) + posb) = val;   // TIMING VARIANT: stores elided''')],
    # every store issued, all into an L2-resident window
    "storehot": [(ST, '''                asm("" : "+v"(posb));
                *(double*)((char*)a.out_base + (((size_t)tile & 63) * 40960 + (size_t)sel * (NT * 8)) + posb) = val;   // TIMING VARIANT: every store issued, into an L2-resident window''')],
    "nobarrier": [(BAR, '''        // TIMING VARIANT: no day barrier
        if (tid < 3 * CPB)      // reset the buffer of the day after next (see s_ext)''')],
    # (No "no time-table prefetch" variant: without the rows the physics runs on whatever the LDS slots hold — zeros send every
    # step down the night path — so its launch time says nothing about the prefetch's cost.  Two real reorderings were measured
    # in round 3 instead, same box: the rows written in front of pass 1's stores, and staged two days ahead and written in front
    # of pass 2's stores: -0.5 % and -1 %.  The prefetch is not where time goes.)
    # an explicit drain of the memory counter behind each batch of five stores: if the launch time does not move, the waits
    # are already there (the compiler's vmcnt(0) in front of the first LDS / memory load into a pending store's data register)
    "drain_after_stores": [("        if (!BG) ring_day += a.out_day_stride;\n    }", "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");     // TIMING VARIANT\n        if (!BG) ring_day += a.out_day_stride;\n    }")],
    "prologue_only": [(LOOP, '''    for (int dl = 0; dl < (a.N < 0 ? ndays : 0); ++dl, ++run) {     // TIMING VARIANT: prologue only
        const int dabs = day0 + dl;''')],
    "sections": [SECT_GLOBAL,
                 ('''    constexpr int NT = solve_threads(CPB);
    static_assert(NT == RING_BLOCK(CPB), "tile-day block = one value per lane");''', '''    constexpr int NT = solve_threads(CPB);
    static_assert(NT == RING_BLOCK(CPB), "tile-day block = one value per lane");
    const long long t_entry = clock64();'''),
                 ('''    enter_layer(day0);
    // the tile's first block of this launch''', '''    const long long t_staged = clock64();
    enter_layer(day0);
    // the tile's first block of this launch'''),
                 (LOOP, '''    long long acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0;
    const long long t_loop = clock64();
    for (int dl = 0; dl < ndays; ++dl, ++run) {
        const long long t_top = clock64();
        const int dabs = day0 + dl;'''),
                 (BAR, '''        const long long t_p1 = clock64();
        __syncthreads();
        const long long t_bar = clock64();
        if (tid < 3 * CPB)      // reset the buffer of the day after next (see s_ext)'''),
                 ('''        if (!BG) ring_day += a.out_day_stride;
    }''', '''        if (!BG) ring_day += a.out_day_stride;
        const long long t_end = clock64();
        acc1 += t_p1 - t_top; acc2 += t_bar - t_p1; acc3 += t_end - t_bar; acc4 += t_end - t_top;
    }
    if (F && SS && (tid & 63) == 0 && (tile & 15) == 0) {
        const int grp = hr / 3;
        atomicAdd(&g_sect[grp * 8 + 0], (unsigned long long)acc1);
        atomicAdd(&g_sect[grp * 8 + 1], (unsigned long long)acc2);
        atomicAdd(&g_sect[grp * 8 + 2], (unsigned long long)acc3);
        atomicAdd(&g_sect[grp * 8 + 3], (unsigned long long)acc4);
        atomicAdd(&g_sect[grp * 8 + 4], (unsigned long long)ndays);
        atomicAdd(&g_sect[grp * 8 + 5], (unsigned long long)(t_staged - t_entry));
        atomicAdd(&g_sect[grp * 8 + 6], (unsigned long long)(t_loop - t_staged));
        atomicAdd(&g_sect[grp * 8 + 7], 1ull);
    }'''),
                 ('''void print_variant_stats() {}''', '''void print_variant_stats() {
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sect), sizeof h) != hipSuccess) return;
    fprintf(stderr, "[mcf sections] hours  wave-days   pass1  barrier    pass2      day | tiles  staging  soil+flags  (shader-clock cycles per wave: per day | per tile)\\n");
    for (int g = 0; g < 8; ++g) {
        const double n = (double)(h[g * 8 + 4] ? h[g * 8 + 4] : 1), m = (double)(h[g * 8 + 7] ? h[g * 8 + 7] : 1);
        fprintf(stderr, "[mcf sections] %2d-%2d %10llu %8.0f %8.0f %8.0f %8.0f | %8llu %8.0f %8.0f\\n", 3 * g, 3 * g + 2, h[g * 8 + 4], h[g * 8] / n,
                h[g * 8 + 1] / n, h[g * 8 + 2] / n, h[g * 8 + 3] / n, h[g * 8 + 7], h[g * 8 + 5] / m, h[g * 8 + 6] / m);
    }
}''')],
    # static priority for one half of the workgroup's waves (MI355X_MICROARCH.md, "Two waves per SIMD" item 4): waves w and
    # w + 4 share a SIMD; results bitwise the shipped ones
    "setprio_hi": [(LOOP, '''    if ((tid >> 6) >= 4) __builtin_amdgcn_s_setprio(1);     // VARIANT
    for (int dl = 0; dl < ndays; ++dl, ++run) {
        const int dabs = day0 + dl;''')],
    "setprio_lo": [(LOOP, '''    if ((tid >> 6) < 4) __builtin_amdgcn_s_setprio(1);     // VARIANT
    for (int dl = 0; dl < ndays; ++dl, ++run) {
        const int dabs = day0 + dl;''')],
    # the ten stores as non-temporal stores
    "store_nt": [(ST, '''                asm("" : "+v"(posb));
                {   // VARIANT: nt
                    double* const sb = (double*)((char*)ring_day + ((size_t)sel * (NT * 8)));
                    asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(posb), "v"(val), "s"(sb) : "memory");
                }''')],
    # what the output selector costs: every variable held, in order (sel = v known at compile time) — bench.py's mask
    "allout": [("            const unsigned sel = (unsigned)(osel >> (4 * v)) & 15u;\n", "            const unsigned sel = (unsigned)v;     // VARIANT: all ten outputs, identity order\n")],
    # persistent workgroups (a fixed grid walking the tile sequence): measured a loss in rounds 2 and 3 (128 VGPRs, scratch)
    # (an edit may name its file: (file, old, new); two-element edits are on mcf_kernels.hip)
    "tab_noconflict": [("mcf_device.hpp", '''            : "=v"(off) : "v"(K.sh3), "v"(t));
        T = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(K.tab) + off);''', '''            : "=v"(off) : "v"(K.sh3), "v"(t));
        off &= 8u;      // TIMING VARIANT: every lane reads one of two neighbouring entries (no bank conflicts)
        T = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(K.tab) + off);'''),
                       ("mcf_device.hpp", "    const double2 cl = *reinterpret_cast<const double2*>(K.ltab + 2 * j);",
                        "    const double2 cl = *reinterpret_cast<const double2*>(K.ltab + 2 * (j & 1));     // TIMING VARIANT")],
    # one more LDS round trip behind every operand batch (17 per cell-step): what an exposed LDS latency costs
    "extra_lds_wait": [("mcf_device.hpp", """    (pin1(a), ...);
}""", """    (pin1(a), ...);
    {   // TIMING VARIANT: a dependent LDS read and its wait behind the batch
        double d;
        unsigned z = 0;
        asm volatile("ds_read_b64 %0, %1\\n\\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(z));
    }
}""")],
    # ten more scalar moves behind every operand batch (170 per cell-step, +40 % scalar instructions): what a scalar instruction costs here
    "extra_salu": [("mcf_device.hpp", """    (pin1(a), ...);
}""", """    (pin1(a), ...);
    {   // TIMING VARIANT
        unsigned t;
        asm volatile("s_mov_b32 %0, 0x3f1a36e2\\n\\ts_mov_b32 %0, 0x3f1a36e3\\n\\ts_mov_b32 %0, 0x3f1a36e4\\n\\ts_mov_b32 %0, 0x3f1a36e5\\n\\t"
                     "s_mov_b32 %0, 0x3f1a36e6\\n\\ts_mov_b32 %0, 0x3f1a36e7\\n\\ts_mov_b32 %0, 0x3f1a36e8\\n\\ts_mov_b32 %0, 0x3f1a36e9\\n\\t"
                     "s_mov_b32 %0, 0x3f1a36ea\\n\\ts_mov_b32 %0, 0x3f1a36eb" : "=s"(t));
    }
}""")],
    # the coarse-forcing kernel with the bounded exp's two VGPR residents (three registers: 12 B of scratch in the LDS-staged kernel)
    "coarse_bounded_exp": [("    MK.pin(true, AF == 0, AF == 1);", "    MK.pin(true, AF == 0, AF != 0);     // VARIANT")],
    "persistent_loop": [('''    const int rot = (int)((blockIdx.x >> 8) & 1);
    const int64_t pos = tile_position(a.ntiles_launch);
    if (pos < 0) return;
    const int64_t tile = a.tile_list ? (int64_t)a.tile_list[pos] : pos;
    solve_tile<CPB, AF, BG, F, SSREQ>(a, tile, a.day0, a.ndays, rot);''', '''    // VARIANT: persistent workgroups — a fixed grid (gridDim.x = 8 R), workgroup (x, r) walks positions r, r + R, ... of XCD
    // x's eighth of the tile sequence; no dispatch gap between a workgroup's tiles
    const int rot = (int)((blockIdx.x >> 8) & 1);
    const int64_t per_xcd = (a.ntiles_launch + 7) / 8;
    const int64_t lo = (int64_t)(blockIdx.x & 7) * per_xcd, hi = lo + per_xcd < a.ntiles_launch ? lo + per_xcd : a.ntiles_launch;
    const int64_t R = gridDim.x >> 3;
    for (int64_t pos = lo + (blockIdx.x >> 3); pos < hi; pos += R) {
        const int64_t tile = a.tile_list ? (int64_t)a.tile_list[pos] : pos;
        __syncthreads();
        solve_tile<CPB, AF, BG, F, SSREQ>(a, tile, a.day0, a.ndays, rot);
    }'''),
                        ('''static dim3 solve_grid(int64_t ntiles) { return dim3((unsigned)(8 * ((ntiles + 7) / 8))); }     // tile_position()''',
                         '''static dim3 solve_grid(int64_t ntiles) {
    static const int wg = getenv("MCF_PERSIST_WGS") ? atoi(getenv("MCF_PERSIST_WGS")) : 512;
    return dim3((unsigned)std::min<int64_t>(wg, 8 * ((ntiles + 7) / 8)));
}''')],
}


def main():
    files = {f: (SRC.parent / f).read_text() for f in ("mcf_kernels.hip", "mcf_device.hpp")}
    OUT.mkdir(exist_ok=True)
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for name, edits in VARIANTS.items():
            cur = dict(files)
            ok = True
            for e in edits:
                f, old, new = e if len(e) == 3 else ("mcf_kernels.hip",) + tuple(e)
                if cur[f].count(old) != 1:
                    print(f"{name}: anchor not found exactly once in {f}: {old[:60]!r}", file=sys.stderr)
                    bad += 1
                    ok = False
                    break
                cur[f] = cur[f].replace(old, new)
            if not ok:
                continue
            d = ""
            for f in files:
                if cur[f] == files[f]:
                    continue
                a, b = Path(td) / ("a_" + f), Path(td) / ("b_" + f)
                a.write_text(files[f])
                b.write_text(cur[f])
                d += subprocess.run(["diff", "-u", "--label", f"a/microclimf_amd/csrc/{f}", "--label", f"b/microclimf_amd/csrc/{f}",
                                     str(a), str(b)], capture_output=True, text=True).stdout
            (OUT / f"{name}.patch").write_text(d)
            print(f"{name}: {len(d.splitlines())} lines")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
