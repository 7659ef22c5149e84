// tools/microbench_waves.hip — VALU / SALU issue throughput of ONE SIMD of gfx950 with 1, 2 and 4 resident waves.
//
// tools/microbench_ops.hip measured every instruction of k_solve with ONE wave per SIMD: a lone wave cannot issue faster
// than one instruction per ~5 cycles, so a 2-cycle and a 4-cycle instruction look alike there.  Here one workgroup of
// 256 x W threads (W waves per SIMD; the waves of a workgroup are dealt round-robin to the four SIMDs) runs an unrolled
// stream of 32 INDEPENDENT instructions per iteration (8 register chains x 4), every wave stamps s_memtime around its
// loop, and the figure reported is
//     cycles per wave-instruction per SIMD = longest wave time / (W x iterations x instructions per iteration)
// i.e. the reciprocal throughput of the SIMD's issue port for that instruction (mix).  HIP events around a 256-workgroup
// launch (every CU busy) give the same quantity in wall time.  `--pmc` mode: each mix once on the whole chip as its own
// kernel name, for rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (the unit of
// SQ_ACTIVE_INST_VALU).
//
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench_waves.hip -o tools/microbench_waves.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define R4(x) x x x x
// eight independent chains: doubles d0..d7 (operands %0..%7), ints i0..i7 (%8..%15); constants m (%16), q (%17), one (%18)
#define OUTS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
             "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
#define INS "v"(m), "v"(q), "v"(one), "s"(sm)

enum Kind {
    FMA64, MUL64, ADD64, MAX64, LDEXP64, RNDNE64, CVTI64, FREXPM64, RCP64, RSQ64, CMP64, MOV64,
    FMA32, MOV32, AND32, ADDU32, LSHL32, ASHR32, CND64E, CNDVCC, SDWA, MADU24,
    MIX_FMA_AND, MIX_FMA_MOV, MIX_FMA_SMOV, MIX_FMA_SNOP, MIX_FMA2_AND, MIX_FMA_CND, SMOV, SNOP, SAND,
    KIND_COUNT
};
struct Desc { const char* name; int per_iter; int valu_per_iter; };
static const Desc kDesc[KIND_COUNT] = {
    {"v_fma_f64", 32, 32}, {"v_mul_f64", 32, 32}, {"v_add_f64", 32, 32}, {"v_max_f64", 32, 32}, {"v_ldexp_f64", 32, 32},
    {"v_rndne_f64", 32, 32}, {"v_cvt_i32_f64", 32, 32}, {"v_frexp_mant_f64", 32, 32}, {"v_rcp_f64", 32, 32},
    {"v_rsq_f64", 32, 32}, {"v_cmp_lt_f64 (to an SGPR pair)", 32, 32}, {"v_mov_b64", 32, 32},
    {"v_fma_f32", 32, 32}, {"v_mov_b32", 32, 32}, {"v_and_b32", 32, 32}, {"v_add_u32", 32, 32}, {"v_lshlrev_b32", 32, 32},
    {"v_ashrrev_i32", 32, 32}, {"v_cndmask_b32_e64 (SGPR mask)", 32, 32}, {"v_cndmask_b32_e32 (vcc)", 32, 32},
    {"v_lshlrev_b32_sdwa", 32, 32}, {"v_mad_u32_u24", 32, 32},
    {"mix 1 v_fma_f64 : 1 v_and_b32", 64, 64}, {"mix 1 v_fma_f64 : 1 v_mov_b32", 64, 64},
    {"mix 1 v_fma_f64 : 1 s_mov_b32", 64, 32}, {"mix 1 v_fma_f64 : 1 s_nop 0", 64, 32},
    {"mix 2 v_fma_f64 : 1 v_and_b32", 48, 48}, {"mix 1 v_fma_f64 : 2 v_cndmask_b32_e64", 96, 96},
    {"s_mov_b32", 32, 0}, {"s_nop 0", 32, 0}, {"s_and_b32", 32, 0},
};

#define CH8(fmt) fmt(0, 8) fmt(1, 9) fmt(2, 10) fmt(3, 11) fmt(4, 12) fmt(5, 13) fmt(6, 14) fmt(7, 15)
#define S(x) #x
// one instruction per chain; D = double operand index, I = int operand index
#define I_FMA64(D, I) "v_fma_f64 %" S(D) ", %" S(D) ", %16, %17\n\t"
#define I_MUL64(D, I) "v_mul_f64 %" S(D) ", %" S(D) ", %16\n\t"
#define I_ADD64(D, I) "v_add_f64 %" S(D) ", %" S(D) ", %17\n\t"
#define I_MAX64(D, I) "v_max_f64 %" S(D) ", %" S(D) ", %17\n\t"
#define I_LDEXP64(D, I) "v_ldexp_f64 %" S(D) ", %" S(D) ", %18\n\t"
#define I_RNDNE64(D, I) "v_rndne_f64 %" S(D) ", %" S(D) "\n\t"
#define I_CVTI64(D, I) "v_cvt_i32_f64 %" S(I) ", %" S(D) "\n\t"
#define I_FREXPM64(D, I) "v_frexp_mant_f64 %" S(D) ", %" S(D) "\n\t"
#define I_RCP64(D, I) "v_rcp_f64 %" S(D) ", %" S(D) "\n\t"
#define I_RSQ64(D, I) "v_rsq_f64 %" S(D) ", %" S(D) "\n\t"
#define I_CMP64(D, I) "v_cmp_lt_f64 s[20:21], %" S(D) ", %16\n\t"
#define I_MOV64(D, I) "v_mov_b64 %" S(D) ", %16\n\t"
#define I_FMA32(D, I) "v_fma_f32 %" S(I) ", %" S(I) ", %18, %18\n\t"
#define I_MOV32(D, I) "v_mov_b32 %" S(I) ", %18\n\t"
#define I_AND32(D, I) "v_and_b32 %" S(I) ", %" S(I) ", %18\n\t"
#define I_ADDU32(D, I) "v_add_u32 %" S(I) ", %" S(I) ", %18\n\t"
#define I_LSHL32(D, I) "v_lshlrev_b32 %" S(I) ", 1, %" S(I) "\n\t"
#define I_ASHR32(D, I) "v_ashrrev_i32 %" S(I) ", 1, %" S(I) "\n\t"
#define I_CND64E(D, I) "v_cndmask_b32_e64 %" S(I) ", %" S(I) ", %18, %19\n\t"
#define I_CNDVCC(D, I) "v_cndmask_b32_e32 %" S(I) ", %" S(I) ", %18, vcc\n\t"
#define I_SDWA(D, I) "v_lshlrev_b32_sdwa %" S(I) ", %18, %" S(I) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
#define I_MADU24(D, I) "v_mad_u32_u24 %" S(I) ", %" S(I) ", %18, %18\n\t"
#define I_MIX_FMA_AND(D, I) I_FMA64(D, I) I_AND32(D, I)
#define I_MIX_FMA_MOV(D, I) I_FMA64(D, I) I_MOV32(D, I)
#define I_MIX_FMA_SMOV(D, I) I_FMA64(D, I) "s_mov_b32 s22, 5\n\t"
#define I_MIX_FMA_SNOP(D, I) I_FMA64(D, I) "s_nop 0\n\t"
#define I_MIX_FMA_CND(D, I) I_FMA64(D, I) I_CND64E(D, I) I_CND64E(D, I)
#define I_SMOV(D, I) "s_mov_b32 s22, 5\n\t"
#define I_SNOP(D, I) "s_nop 0\n\t"
#define I_SAND(D, I) "s_and_b32 s22, s22, 7\n\t"

#define BODY(K) if (KIND == K) asm volatile(R4(CH8(I_##K)) : OUTS : INS : "vcc", "s20", "s21", "s22", "memory");

template <int KIND>
__global__ void k_ops(double* out, long long* cyc, int iters) {
    const int t = threadIdx.x;
    double d0 = 1.0 + t * 1e-9, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    int i0 = t, i1 = t + 1, i2 = t + 2, i3 = t + 3, i4 = t + 4, i5 = t + 5, i6 = t + 6, i7 = t + 7;
    const double m = 0.999999, q = 1e-7;
    const int one = 1;
    unsigned long long sm = 0x5555555555555555ull;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int w = 0; w < 2; ++w) {     // first round warms the instruction cache
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            BODY(FMA64) BODY(MUL64) BODY(ADD64) BODY(MAX64) BODY(LDEXP64) BODY(RNDNE64) BODY(CVTI64) BODY(FREXPM64)
            BODY(RCP64) BODY(RSQ64) BODY(CMP64) BODY(MOV64) BODY(FMA32) BODY(MOV32) BODY(AND32) BODY(ADDU32) BODY(LSHL32)
            BODY(ASHR32) BODY(CND64E) BODY(CNDVCC) BODY(SDWA) BODY(MADU24) BODY(MIX_FMA_AND) BODY(MIX_FMA_MOV)
            BODY(MIX_FMA_SMOV) BODY(MIX_FMA_SNOP) BODY(MIX_FMA_CND) BODY(SMOV) BODY(SNOP) BODY(SAND)
            if (KIND == MIX_FMA2_AND)
                asm volatile(R4(I_FMA64(0, 8) I_FMA64(1, 9) I_AND32(0, 8) I_FMA64(2, 10) I_FMA64(3, 11) I_AND32(1, 9)
                                I_FMA64(4, 12) I_FMA64(5, 13) I_AND32(2, 10) I_FMA64(6, 14) I_FMA64(7, 15) I_AND32(3, 11))
                             : OUTS : INS : "vcc", "memory");
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if ((t & 63) == 0 && blockIdx.x == 0) cyc[t / 64] = t1 - t0;
    out[(size_t)blockIdx.x * blockDim.x + t] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int KIND>
static void run_one(double* out, long long* cyc, bool pmc) {
    const int iters = 2000;
    const Desc& d = kDesc[KIND];
    if (pmc) {
        hipLaunchKernelGGL(k_ops<KIND>, dim3(1024), dim3(1024), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        return;
    }
    printf("%-40s", d.name);
    for (int W : {1, 2, 4}) {
        hipLaunchKernelGGL(k_ops<KIND>, dim3(1), dim3(256 * W), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> c(4 * W);
        hipMemcpy(c.data(), cyc, sizeof(long long) * c.size(), hipMemcpyDeviceToHost);
        const long long mx = *std::max_element(c.begin(), c.end());
        // whole chip, every CU holding one such workgroup: wall time
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_ops<KIND>, dim3(256), dim3(256 * W), 0, 0, out, cyc, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ops<KIND>, dim3(256), dim3(256 * W), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
        const double n = 2.0 * iters * d.per_iter * W;      // both rounds run between the events
        printf("  W=%d %6.2f cyc (%.3f ns/instr/SIMD wall)", W, (double)mx / (iters * (double)d.per_iter * W), ms * 1e6 / n);
    }
    printf("\n");
}

template <int K>
static void run_all(double* out, long long* cyc, bool pmc) {
    run_one<K>(out, cyc, pmc);
    if constexpr (K + 1 < KIND_COUNT) run_all<K + 1>(out, cyc, pmc);
}

int main(int argc, char** argv) {
    const bool pmc = argc > 1 && !strcmp(argv[1], "--pmc");
    double* out; long long* cyc;
    hipMalloc(&out, 8ull * 1024 * 1024); hipMalloc(&cyc, 8 * 64);
    if (!pmc)
        printf("cycles per wave-instruction per SIMD (s_memtime of the slowest wave / (W x instructions)), W waves per SIMD on one CU;\n"
               "in brackets: the same from HIP events with one such workgroup on each of the 256 CUs\n");
    run_all<0>(out, cyc, pmc);
    return 0;
}
