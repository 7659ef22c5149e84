// Does the VGPR bank of an fp64 instruction's operands change its issue cost on gfx950?  (tools/microbench_waves.hip prices
// instructions whose common operands the compiler placed; k_solve's FMAs read three different register pairs.)
// 1024 workgroups x 1024 threads = 4 waves per SIMD on every CU; eight independent instructions per group, physical registers
// named in the assembly (64-bit tuples sit at even registers on gfx950: a pair is "class 0" (reg = 0 mod 4) or "class 2");
// HIP-event time per launch; the ratio to the first pattern is what is read.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench_banks.bin tools/microbench_banks.hip && tools/microbench_banks.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(x) x x x x x x x x
#define FMA(d, a, b, c) "v_fma_f64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n\t"
#define MUL(d, a, b, c) "v_mul_f64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n\t"
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "v64","v65","v66","v67","v68","v69","v70","v71"
// eight instructions, destinations alternating between the classes (40, 42, ...), the same three sources in each
#define G_ALT(OP, a, b, c) OP(40, a, b, c) OP(42, a, b, c) OP(44, a, b, c) OP(46, a, b, c) OP(48, a, b, c) OP(50, a, b, c) OP(52, a, b, c) OP(54, a, b, c)
// ... destinations all class 0 / all class 2
#define G_D0(OP, a, b, c) OP(40, a, b, c) OP(44, a, b, c) OP(48, a, b, c) OP(52, a, b, c) OP(56, a, b, c) OP(60, a, b, c) OP(64, a, b, c) OP(68, a, b, c)
#define G_D2(OP, a, b, c) OP(42, a, b, c) OP(46, a, b, c) OP(50, a, b, c) OP(54, a, b, c) OP(58, a, b, c) OP(62, a, b, c) OP(66, a, b, c) OP(70, a, b, c)

struct Pat { const char* name; };
static const Pat kPat[] = {
    {"fma  src 0,0,0 same register x3 (8,8,8)       dst alt"}, {"fma  src 0,0,0 (8,12,16)                      dst alt"},
    {"fma  src 0,2,0 (8,10,12)                      dst alt"}, {"fma  src 0,0,2 (8,12,10)                      dst alt"},
    {"fma  src 2,0,0 (10,8,12)                      dst alt"}, {"fma  src 0,2,2 (8,10,14)                      dst alt"},
    {"fma  src 2,2,2 (10,14,18)                     dst alt"}, {"fma  src 0,0,0 (8,12,16)                      dst all 0"},
    {"fma  src 0,0,0 (8,12,16)                      dst all 2"}, {"fma  src 2,2,2 (10,14,18)                     dst all 0"},
    {"fma  src 0,2,0 (8,10,12)                      dst all 0"}, {"fma  src 0,2,0 (8,10,12)                      dst all 2"},
    {"fma  src a,a,b same class (8,8,12)            dst alt"}, {"fma  src a,a,b other class (8,8,10)           dst alt"},
    {"mul  src 0,0 (8,12)                           dst alt"}, {"mul  src 0,2 (8,10)                           dst alt"},
    {"mul  src a,a (8,8)                            dst alt"},
};
constexpr int NP = sizeof(kPat) / sizeof(kPat[0]);

template <int P>
__global__ void k(double* out, int iters) {
    for (int i = 0; i < iters; ++i) {
        if (P == 0) asm volatile(R8(G_ALT(FMA, 8, 8, 8)) ::: CLOB);
        if (P == 1) asm volatile(R8(G_ALT(FMA, 8, 12, 16)) ::: CLOB);
        if (P == 2) asm volatile(R8(G_ALT(FMA, 8, 10, 12)) ::: CLOB);
        if (P == 3) asm volatile(R8(G_ALT(FMA, 8, 12, 10)) ::: CLOB);
        if (P == 4) asm volatile(R8(G_ALT(FMA, 10, 8, 12)) ::: CLOB);
        if (P == 5) asm volatile(R8(G_ALT(FMA, 8, 10, 14)) ::: CLOB);
        if (P == 6) asm volatile(R8(G_ALT(FMA, 10, 14, 18)) ::: CLOB);
        if (P == 7) asm volatile(R8(G_D0(FMA, 8, 12, 16)) ::: CLOB);
        if (P == 8) asm volatile(R8(G_D2(FMA, 8, 12, 16)) ::: CLOB);
        if (P == 9) asm volatile(R8(G_D0(FMA, 10, 14, 18)) ::: CLOB);
        if (P == 10) asm volatile(R8(G_D0(FMA, 8, 10, 12)) ::: CLOB);
        if (P == 11) asm volatile(R8(G_D2(FMA, 8, 10, 12)) ::: CLOB);
        if (P == 12) asm volatile(R8(G_ALT(FMA, 8, 8, 12)) ::: CLOB);
        if (P == 13) asm volatile(R8(G_ALT(FMA, 8, 8, 10)) ::: CLOB);
        if (P == 14) asm volatile(R8(G_ALT(MUL, 8, 12, 0)) ::: CLOB);
        if (P == 15) asm volatile(R8(G_ALT(MUL, 8, 10, 0)) ::: CLOB);
        if (P == 16) asm volatile(R8(G_ALT(MUL, 8, 8, 0)) ::: CLOB);
    }
    if (threadIdx.x == 4096) out[0] = 1.0;
}

template <int P>
static double run(double* out) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(1024), dim3(1024), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(1024), dim3(1024), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 1024 workgroups x 16 waves / (256 CUs x 4 SIMDs) = 16 waves in turn, 64 instructions per iteration each
    return ms * 1e6 / (16.0 * iters * 64);      // ns per wave-instruction per SIMD
}
template <int P>
static void all(double* out, double* r) {
    r[P] = run<P>(out);
    if constexpr (P + 1 < NP) all<P + 1>(out, r);
}

int main() {
    double* out;
    (void)hipMalloc(&out, 8);
    // (the clock ramps up over the first launches of a process: three rounds over all patterns, the last two are read)
    double r[3][NP];
    for (int round = 0; round < 3; ++round) all<0>(out, r[round]);
    for (int i = 0; i < NP; ++i)
        printf("%-60s %.3f / %.3f / %.3f ns per wave-instruction per SIMD  (x %.3f of the first pattern, last round)\n", kPat[i].name, r[0][i],
               r[1][i], r[2][i], r[2][i] / r[2][0]);
    return 0;
}
