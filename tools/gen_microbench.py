#!/usr/bin/env python3
"""Generates tools/microbench_ops.hip: issue cost of single VALU / LDS instructions on gfx950, one wave per SIMD on one CU
(four independent chains per wave unless the name says 'dep'), timed with s_memtime inside the wave and with HIP events
around the launch.  Build + run: hipcc -O2 --offload-arch=gfx950 tools/microbench_ops.hip -o tools/microbench_ops.bin (the
binary travels with gpurun)."""
from pathlib import Path

# name, list of asm lines for ONE group (4 chains), registers: a,b,c,d doubles (0-3), ia..id ints (4-7), m,q double consts (8,9),
# one (10) int const, s (11) sgpr pair mask
OPS = [
    ("REAL 64-bit select: cmp vcc; cnd_e32 lo; cnd_e32 hi (independent)", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_cndmask_b32 %{j}, {ONE}, {TWO}, vcc", "v_cndmask_b32 %{h}, {TWO}, {ONE}, vcc"]),
    ("same with 2 fma between cmp and the selects", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_fma_f64 %{i}, %{i}, {M}, {Q}", "v_fma_f64 %{i}, %{i}, {M}, {Q}", "v_cndmask_b32 %{j}, {ONE}, {TWO}, vcc", "v_cndmask_b32 %{h}, {TWO}, {ONE}, vcc"]),
    ("same, e64 selects with vcc", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_cndmask_b32_e64 %{j}, {ONE}, {TWO}, vcc", "v_cndmask_b32_e64 %{h}, {TWO}, {ONE}, vcc"]),
    ("same, cmp -> sgpr pair, e64 selects", ["v_cmp_lt_f64 {MASK}, %{i}, {M}", "v_cndmask_b32_e64 %{j}, {ONE}, {TWO}, {MASK}", "v_cndmask_b32_e64 %{h}, {TWO}, {ONE}, {MASK}"]),
    ("two e32 selects back to back, no cmp (independent)", ["v_cndmask_b32 %{j}, {ONE}, {TWO}, vcc", "v_cndmask_b32 %{h}, {TWO}, {ONE}, vcc"]),
    ("cnd_e32, fma, cnd_e32, fma (independent)", ["v_cndmask_b32 %{j}, {ONE}, {TWO}, vcc", "v_fma_f64 %{i}, %{i}, {M}, {Q}", "v_cndmask_b32 %{h}, {TWO}, {ONE}, vcc", "v_fma_f64 %{i}, %{i}, {M}, {Q}"]),
    ("v_bfi_b32", ["v_bfi_b32 %{j}, %{h}, {ONE}, {TWO}"]),
    ("select by mask: cmp sgpr; cnd_e64 mask; 2 bfi", ["v_cmp_lt_f64 {MASK}, %{i}, {M}", "v_cndmask_b32_e64 %{j}, 0, -1, {MASK}", "v_bfi_b32 %{h}, %{j}, {ONE}, {TWO}", "v_bfi_b32 %{h}, %{j}, {TWO}, {ONE}"]),
    ("v_fma_f64", ["v_fma_f64 %{i}, %{i}, {M}, {Q}" for _ in range(1)]),
    ("v_fmac_f64", ["v_fmac_f64 %{i}, {M}, {Q}"]),
    ("v_mul_f64", ["v_mul_f64 %{i}, %{i}, {M}"]),
    ("v_rcp_f64", ["v_rcp_f64 %{i}, %{i}"]),
    ("v_rcp_f32", ["v_rcp_f32 %{j}, %{j}"]),
    ("v_cndmask_b32 vcc (dst=src0)", ["v_cndmask_b32 %{j}, %{j}, {ONE}, vcc"]),
    ("v_cndmask_b32 vcc (dst!=src)", ["v_cndmask_b32 %{j}, {ONE}, {ONE}, vcc"]),
    ("v_cndmask_b32_e64 sgpr mask", ["v_cndmask_b32_e64 %{j}, %{j}, {ONE}, {MASK}"]),
    ("1 v_cmp vcc : 2 v_cndmask vcc (64-bit select)", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc", "v_cndmask_b32 %{j}, {ONE}, %{j}, vcc"]),
    ("1 v_cmp vcc : 4 v_cndmask vcc", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc", "v_cndmask_b32 %{j}, {ONE}, %{j}, vcc", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc", "v_cndmask_b32 %{j}, {ONE}, %{j}, vcc"]),
    ("1 v_cmp sgpr : 2 v_cndmask_e64 sgpr", ["v_cmp_lt_f64 {MASK}, %{i}, {M}", "v_cndmask_b32_e64 %{j}, %{j}, {ONE}, {MASK}", "v_cndmask_b32_e64 %{j}, {ONE}, %{j}, {MASK}"]),
    ("v_cndmask_b32_e64 with vcc", ["v_cndmask_b32_e64 %{j}, %{j}, {ONE}, vcc"]),
    ("s_mov vcc then 4 v_cndmask vcc", ["s_mov_b64 vcc, {MASK}", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc", "v_cndmask_b32 %{j}, {ONE}, %{j}, vcc", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc", "v_cndmask_b32 %{j}, {ONE}, %{j}, vcc"]),
    ("v_cmp vcc, fma, fma, v_cndmask vcc", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_fma_f64 %{i}, %{i}, {M}, {Q}", "v_fma_f64 %{i}, %{i}, {M}, {Q}", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc"]),
    ("v_addc_co_u32 vcc", ["v_addc_co_u32 %{j}, vcc, %{j}, {ONE}, vcc"]),
    ("v_cmp_lt_f64 vcc + v_cndmask pair", ["v_cmp_lt_f64 vcc, %{i}, {M}", "v_cndmask_b32 %{j}, %{j}, {ONE}, vcc"]),
    ("v_cmp_lt_f64 sgpr + v_cndmask_e64 pair", ["v_cmp_lt_f64 {MASK}, %{i}, {M}", "v_cndmask_b32_e64 %{j}, %{j}, {ONE}, {MASK}"]),
    ("v_cmp_lt_f64 vcc", ["v_cmp_lt_f64 vcc, %{i}, {M}"]),
    ("v_cmp_class_f64 vcc", ["v_cmp_class_f64 vcc, %{i}, {ONE}"]),
    ("v_max_f64", ["v_max_f64 %{i}, %{i}, {M}"]),
    ("v_mov_b32", ["v_mov_b32 %{j}, {ONE}"]),
    ("v_mov_b64", ["v_mov_b64 %{i}, {M}"]),
    ("v_and_b32", ["v_and_b32 %{j}, %{j}, {ONE}"]),
    ("v_lshlrev_b32", ["v_lshlrev_b32 %{j}, 3, %{j}"]),
    ("v_lshlrev_b32_sdwa byte0", ["v_lshlrev_b32_sdwa %{j}, {ONE}, %{j} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"]),
    ("v_ashrrev_i32", ["v_ashrrev_i32 %{j}, 8, %{j}"]),
    ("v_add_u32", ["v_add_u32 %{j}, %{j}, {ONE}"]),
    ("v_mul_lo_u32", ["v_mul_lo_u32 %{j}, %{j}, {ONE}"]),
    ("v_lshl_add_u64", ["v_lshl_add_u64 %{i}, %{i}, 3, {M}"]),
    ("v_cvt_f64_i32", ["v_cvt_f64_i32 %{i}, %{j}"]),
    ("v_cvt_i32_f64", ["v_cvt_i32_f64 %{j}, %{i}"]),
    ("v_rndne_f64", ["v_rndne_f64 %{i}, %{i}"]),
    ("v_trunc_f64", ["v_trunc_f64 %{i}, %{i}"]),
    ("v_floor_f64", ["v_floor_f64 %{i}, %{i}"]),
    ("v_ldexp_f64", ["v_ldexp_f64 %{i}, %{i}, %{j}"]),
    ("v_frexp_mant_f64", ["v_frexp_mant_f64 %{i}, %{i}"]),
    ("v_frexp_exp_i32_f64", ["v_frexp_exp_i32_f64 %{j}, %{i}"]),
    ("v_div_scale_f64", ["v_div_scale_f64 %{i}, vcc, %{i}, {M}, %{i}"]),
    ("v_div_fmas_f64", ["v_div_fmas_f64 %{i}, %{i}, {M}, {Q}"]),
    ("v_div_fixup_f64", ["v_div_fixup_f64 %{i}, %{i}, {M}, {Q}"]),
    ("v_readlane_b32", ["v_readlane_b32 {SG}, %{j}, 3"]),
    ("v_readfirstlane_b32", ["v_readfirstlane_b32 {SG}, %{j}"]),
    ("v_mov_b32 dpp row_shr", ["v_mov_b32_dpp %{j}, %{j} row_shr:1 row_mask:0xf bank_mask:0xf"]),
    ("ds_bpermute_b32 + wait", ["ds_bpermute_b32 %{j}, %{j}, %{j}", "s_waitcnt lgkmcnt(0)"]),
    ("ds_read_b64 + wait (latency)", ["ds_read_b64 %{i}, %{j}", "s_waitcnt lgkmcnt(0)"]),
    ("ds_read_b64 x4 then wait", None),
    ("ds_read2_b64 x4 then wait", None),
    ("ds_max_f64 (atomic, no return)", ["ds_max_f64 %{j}, %{i}"]),
    ("s_nop 0", ["s_nop 0"]),
    ("s_mov_b32", ["s_mov_b32 {SG}, 5"]),
]

HEAD = r'''// GENERATED by tools/gen_microbench.py — do not edit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
__shared__ double lds_buf[2048];
template <int KIND>
__global__ void k(double* out, long long* cyc, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = a + 1, c = a + 2, d = a + 3;
    const double m = 0.999999, q = 1e-7;
    int ia = (threadIdx.x & 63) * 8, ib = ia + 512, ic = ia + 1024, id = ia + 1536, ie = 1, if_ = 2, ig = 3, ih = 4;
    const int one = 1, two = 2;
    unsigned long long mask = 0x5555555555555555ull;
    int sg = 0;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds_buf[i] = 1.0 + i;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int w = 0; w < 2; ++w) {
        __builtin_amdgcn_s_waitcnt(0);
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
'''
TAIL = r'''        }
        __builtin_amdgcn_s_waitcnt(0);
        t1 = __builtin_amdgcn_s_memtime();
    }
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
    out[threadIdx.x] = a + b + c + d + ia + ib + ic + id + ie + if_ + ig + ih + sg;
}
template <int KIND>
static void run(const char* name, int per_iter, double* out, long long* cyc) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[4];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int i = 0; i < 4; ++i) mx = c[i] > mx ? c[i] : mx;
    printf("%-42s %7.2f ticks per instruction (one wave per SIMD; launch %.3f ms)\n", name, (double)mx / ((double)iters * per_iter), ms);
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 64);
'''


def body(lines):
    group = []
    for ch in range(4):
        for ln in lines:
            group.append(ln.replace("{i}", str(ch)).replace("{j}", str(4 + ch)).replace("{h}", str(8 + ch)))
    s = "\\n\\t".join(group) + "\\n\\t"
    return s, len(group)


def main():
    out = [HEAD]
    runs = []
    outs = '"+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "+v"(ie), "+v"(if_), "+v"(ig), "+v"(ih)'
    cons = (': ' + outs + ' : "v"(m), "v"(q), "v"(one), "v"(two), "s"(mask), "s"(sg) : "vcc", "memory"')
    NAMES = {"{M}": "%12", "{Q}": "%13", "{ONE}": "%14", "{TWO}": "%15", "{MASK}": "%16", "{SG}": "%17"}
    cons_sg = (': ' + outs + ', "+s"(sg), "+s"(mask) : "v"(m), "v"(q), "v"(one), "v"(two) : "vcc", "memory"')
    NAMES_SG = {"{SG}": "%12", "{MASK}": "%13", "{M}": "%14", "{Q}": "%15", "{ONE}": "%16", "{TWO}": "%17"}
    for k, (name, lines) in enumerate(OPS):
        if lines is None:
            op = "ds_read_b64" if "read_b64" in name else "ds_read2_b64"
            if op == "ds_read_b64":
                s = "\\n\\t".join(f"ds_read_b64 %{i}, %{4 + i}" for i in range(4)) + "\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"
                n = 4
                out.append(f'            if (KIND == {k}) asm volatile(REP8("{s}") {cons});\n')
            else:
                # ds_read2_b64 needs a 4-VGPR destination: use two 128-bit temporaries
                s = "ds_read2_b64 %0, %2 offset1:1\\n\\tds_read2_b64 %1, %3 offset1:1\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"
                n = 2
                out.append('            if (KIND == %d) { double __attribute__((ext_vector_type(2))) x0, x1; asm volatile(REP8("%s") : "=&v"(x0), "=&v"(x1) : "v"(ia), "v"(ib) : "memory"); a += x0.x + x1.y; }\n' % (k, s))
            runs.append((k, name, 8 * n))
            continue
        uses_sg = any("{SG}" in ln for ln in lines) or any(ln.startswith("v_cmp_lt_f64 {MASK}") for ln in lines)
        s, n = body(lines)
        for kk, vv in (NAMES_SG if uses_sg else NAMES).items():
            s = s.replace(kk, vv)
        out.append(f'            if (KIND == {k}) asm volatile(REP8("{s}") {cons_sg if uses_sg else cons});\n')
        instr = sum(1 for ln in lines if not ln.startswith("s_waitcnt")) * 4
        runs.append((k, name, 8 * instr))
    out.append(TAIL)
    for k, name, n in runs:
        out.append(f'    run<{k}>("{name}", {n}, out, cyc);\n')
    out.append("    return 0;\n}\n")
    Path(__file__).with_name("microbench_ops.hip").write_text("".join(out))


if __name__ == "__main__":
    main()
