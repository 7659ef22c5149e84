// Issue cost and dependent-issue latency of the fp64 VALU instructions k_solve is made of, measured by one wave on an
// otherwise idle CU (s_memtime around an unrolled block; gfx950).  Build + run: tools/microbench_fp64.sh (through gpurun).
//   dep  : every instruction reads the result of the one before it  -> cycles per instruction = its latency
//   ind4 : four independent chains interleaved                       -> cycles per instruction = its issue cost (if < latency)
// With W waves on the SIMD (second argument of the kernel) the same blocks show how many waves hide a dependent chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define BODY_DEP(op) REP64(op " %0, %0, %4, %5\n\t")
#define BODY_IND4(op) REP8(REP8(op " %0, %0, %4, %5\n\t" op " %1, %1, %4, %5\n\t" op " %2, %2, %4, %5\n\t" op " %3, %3, %4, %5\n\t"))

template <int KIND>
__global__ void k(double* out, long long* cyc, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = a + 1, c = a + 2, d = a + 3;
    const double m = 0.999999, q = 1e-7;
    int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;
    long long t0 = 0, t1 = 0;
    for (int w = 0; w < 2; ++w) {      // first round warms the instruction cache
        __builtin_amdgcn_s_waitcnt(0);
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) asm volatile(BODY_DEP("v_fma_f64") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 1) asm volatile(BODY_IND4("v_fma_f64") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 2) asm volatile(REP64("v_mul_f64 %0, %0, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 3) asm volatile(REP8(REP8("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 4) asm volatile(REP64("v_add_f64 %0, %0, %5\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 5) asm volatile(REP64("v_rcp_f64 %0, %0\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 6) asm volatile(REP8(REP8("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 7) asm volatile(REP8(REP8("v_rsq_f64 %0, %0\n\tv_rsq_f64 %1, %1\n\tv_rsq_f64 %2, %2\n\tv_rsq_f64 %3, %3\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 8) asm volatile(REP8(REP8("v_max_f64 %0, %0, %4\n\tv_max_f64 %1, %1, %4\n\tv_max_f64 %2, %2, %4\n\tv_max_f64 %3, %3, %4\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 9) asm volatile(REP8(REP8("v_ldexp_f64 %0, %0, %6\n\tv_ldexp_f64 %1, %1, %6\n\tv_ldexp_f64 %2, %2, %6\n\tv_ldexp_f64 %3, %3, %6\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q), "v"(0));
            if (KIND == 10) asm volatile(REP8(REP8("v_rndne_f64 %0, %0\n\tv_rndne_f64 %1, %1\n\tv_rndne_f64 %2, %2\n\tv_rndne_f64 %3, %3\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 11) asm volatile(REP8(REP8("v_cvt_i32_f64 %0, %4\n\tv_cvt_i32_f64 %1, %5\n\tv_cvt_i32_f64 %2, %4\n\tv_cvt_i32_f64 %3, %5\n\t")) : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(a), "v"(b));
            if (KIND == 12) asm volatile(REP8(REP8("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t")) : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(1));
            if (KIND == 13) asm volatile(REP64("v_add_u32 %0, %0, %4\n\t") : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(1));
            if (KIND == 14) asm volatile(REP8(REP8("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\t" "v_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %6\n\t")) : "+v"(a), "+v"(b), "+v"(ia), "+v"(ib) : "v"(m), "v"(q), "v"(1));
            if (KIND == 15) asm volatile(REP8(REP8("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
            if (KIND == 16) asm volatile(REP8(REP8("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc\n\t")) : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(1) : "vcc");
            if (KIND == 17) asm volatile(REP8(REP8("v_cmp_lt_f64 vcc, %0, %4\n\tv_cmp_lt_f64 vcc, %1, %4\n\tv_cmp_lt_f64 vcc, %2, %4\n\tv_cmp_lt_f64 vcc, %3, %4\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            if (KIND == 18) asm volatile(REP8(REP8("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %4\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
        }
        __builtin_amdgcn_s_waitcnt(0);
        t1 = __builtin_amdgcn_s_memtime();
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + ia + ib + ic + id;
}

template <int KIND>
static void run(const char* name, int per_iter, double* out, long long* cyc) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;      // one workgroup on one CU: waves are dealt round-robin to its 4 SIMDs
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long c[16];
        hipMemcpy(c, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int i = 0; i < threads / 64; ++i) mx = c[i] > mx ? c[i] : mx;
        // (the launch's wall time beside the ticks: two rounds of the block, so ticks / (ms / 2) is the counter's rate)
        printf("%-28s waves/SIMD %d: %8.2f memtime ticks per instruction per wave (x %d waves = %6.2f per SIMD-instruction), launch %.3f ms\n",
               name, waves_per_simd, (double)mx / ((double)iters * per_iter), waves_per_simd,
               (double)mx / ((double)iters * per_iter) / waves_per_simd, ms);
    }
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 8 * 4096);
    hipMalloc(&cyc, 8 * 64);
    run<0>("v_fma_f64 dependent", 64, out, cyc);
    run<1>("v_fma_f64 4 chains", 256, out, cyc);
    run<15>("v_fma_f64 2 chains", 128, out, cyc);
    run<2>("v_mul_f64 dependent", 64, out, cyc);
    run<3>("v_mul_f64 4 chains", 256, out, cyc);
    run<4>("v_add_f64 dependent", 64, out, cyc);
    run<5>("v_rcp_f64 dependent", 64, out, cyc);
    run<6>("v_rcp_f64 4 chains", 256, out, cyc);
    run<7>("v_rsq_f64 4 chains", 256, out, cyc);
    run<8>("v_max_f64 4 chains", 256, out, cyc);
    run<9>("v_ldexp_f64 4 chains", 256, out, cyc);
    run<10>("v_rndne_f64 4 chains", 256, out, cyc);
    run<11>("v_cvt_i32_f64 4 indep", 256, out, cyc);
    run<12>("v_add_u32 4 chains", 256, out, cyc);
    run<13>("v_add_u32 dependent", 64, out, cyc);
    run<14>("2 fma_f64 + 2 add_u32 mix", 256, out, cyc);
    run<16>("v_cndmask_b32 4 chains", 256, out, cyc);
    run<17>("v_cmp_lt_f64 4 indep", 256, out, cyc);
    run<18>("v_mov_b64 4 indep", 256, out, cyc);
    return 0;
}
