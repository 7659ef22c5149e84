// mcf_device.hpp — device-side physics of the grid microclimate solver (gfx950).
//
// From-scratch restatement of the arithmetic of runmicro1Cpp / runmicro2Cpp
// (reference: src/microclimfCpp.cpp, cited as cpp:LINE) organised by HOISTING
// CLASS instead of by the reference's call tree:
//
//   CellConst  (CF_*)  depends on the raster cell only; computed once per plan by
//                      k_cell_setup into an SoA table, staged in LDS by the solver.
//   TimeConst  (TF_*)  depends on the time step only (vector forcing); computed
//                      once per plan by k_time_setup; a day's 24 rows are staged
//                      in LDS.  For array forcing the same values are derived
//                      per cell-step in registers.
//   pass1 / pass2      the remaining cell x time work.  One lane = one
//                      (cell, hour); the two passes of a day are separated by
//                      one workgroup barrier that carries the day's
//                      max/min/|Rnet|max reduction through LDS.
//
// All arithmetic is IEEE fp64 (no fast-math): NaN comparison semantics are
// load-bearing for bare (pai = hgt = 0) cells, see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcf_kernels.h"


namespace mcf {

constexpr double kPi = 3.14159265358979323846;     // cpp:14
constexpr double kToRad = kPi / 180.0;             // cpp:15
constexpr double kSb = 5.67e-8;                    // cpp:16
constexpr double kThetam = 0.365;                  // cpp:17
constexpr double kKa = 0.4;                        // cpp:18
constexpr double kOmdy = (2.0 * kPi) / (24.0 * 3600.0);  // cpp:19
constexpr uint64_t kNaRealBits = 0x7FF00000000007A2ULL;  // R's NA_real_

// ---- cell flags -------------------------------------------------------------
enum : int {
    FL_VALID = 1,      // hgt is not NA                       cpp:2182-2183
    FL_PAI = 2,        // pai > 0                             cpp:1093,1165
    FL_BELOW = 4,      // !(reqhgt2 >= hgt): below canopy     cpp:1434
    FL_XONE = 8,       // x == 1                              cpp:109
    FL_XINF = 16,      // isinf(x)                            cpp:112
    FL_XZERO = 32,     // x == 0                              cpp:115
    FL_ABOVE1 = 64,    // reqhgt2 > d + zh                    cpp:1303 (TVabove at reqhgt)
    FL_ABOVE2 = 128,   // hgt > d + zh                        cpp:1303 (TVabove at hgt)
    FL_OMPNAN = 256,   // isnan(omp)                          cpp:464
    FL_STOM = 512,     // gsmax < 999.99                      cpp:1351
    FL_REGULAR = 1024  // every constant this cell's path reads is finite and in its physical range: no clamp of the
                       // hot loop can then meet a NaN operand (see `cap` / `flr` below); set by k_cell_setup
};
// Step flag packed into TF_IDX beside sindex / windex / ksat: the step's forcing is not finite or outside the range the
// fast clamps assume (pk, De, ghr > 0).
constexpr int kStepIrregular = 512;
// Day flag, set on all 24 rows of a day: the point model's soil moisture (pointm$soilm) has ONE value on this day — what
// soilmCpp's daily bucket model produces (src/microclimfCpp.cpp:931-972).  Everything that depends on the cell and on soil
// moisture only (spread soil moisture, matric potential, conductivity and damping depth, the stomatal water-stress
// factor) is then one value per cell-DAY, and k_solve computes it once per tile and day (SoilDay below) instead of in
// each of the 24 hour lanes.
constexpr int kSoilDaily = 1024;

// ---- per-cell constant table ------------------------------------------------
enum CellField : int {
    CF_FLAGS = 0,
    // solar index (cpp:85-102): si = cz*cs + sz*(ssca*caz + sssa*saz)
    CF_CS, CF_SSCA, CF_SSSA,
    // soil moisture spread (cpp:1021-1032): sm = th/(th + (1-th)*eta), eta = exp(-tadd)
    CF_SMIN, CF_RGE, CF_INVRGE, CF_ETA,
    // canopy extinction (cpp:104-132)
    CF_XX, CF_KDENINV,
    // two-stream diffuse constants (cpp:134-162, 1034-1084)
    // (KA1 .. KZ2, round 5.  The direct-beam coefficients p6, p7, p9, p10 (cpp:164-185) are linear in two per-step values each —
    // p6 = v1 k6a - g k6b, p7 = g k7b - v1 k7a with g = S2 v2; p9 = -(p8s k9a + w), p10 = p8s k10a + w with w = v3 / D2 — with
    // per-CELL factors k.. = (1/D1 | 1/D2) exp(-+h pait) (u -+ h | a + gma -+ h), and they are only ever used in four sums weighted
    // by per-cell exponentials (cpp:1102-1117): p6 + p7, p6 e^(-h paiaa) + p7 e^(+h paiaa), p9 S1 + p10 e^(h pait),
    // p9 e^(-h paiaa) + p10 e^(+h paiaa).  Multiplied out once per cell, each sum is two fmas of (v1, g) or (p8s, w).)
    CF_PAIT, CF_OM, CF_JDEL, CF_GMA, CF_GMA2, CF_AGM, CF_AGM2, CF_U1, CF_U2, CF_KA1, CF_KA2, CF_KB1, CF_KB2, CF_KG1, CF_KG2, CF_KZ1,
    CF_KZ2, CF_INVD2, CF_GREF, CF_GMAGREF, CF_LOGCLUMP, CF_LOGGI, CF_TRDN, CF_TRDU, CF_AMX,
    CF_PAIAA, CF_ALBD, CF_RDDNG, CF_RDDNZ, CF_RDUPZ, CF_SVFA,
    CF_HOM, CF_HOMP,
    // long wave (cpp:1165-1175)
    CF_TSV, CF_OMTRDIF,
    // wind (cpp:1179-1218)
    CF_UFC, CF_UZFAC, CF_GHAFAC,
    // soil surface (cpp:1262-1275) and conductivity (cpp:1249-1260, 628-636)
    CF_ABSPSIE, CF_INVSMAX, CF_SOILB, CF_RHO, CF_CSA, CF_C1, CF_C1MC4, CF_C3,
    // stomata (cpp:391-458)
    CF_GSMAX, CF_RSMX, CF_INV02RSMX, CF_RAT, CF_RATC, CF_PSIW0, CF_KK, CF_MUDENINV,
    // canopy conductance (cpp:460-477, 1425-1428)
    CF_PAI, CF_OMPC, CF_KSAT, CF_PSUNSAT, CF_SHADEFAC,
    // TVabove log-profile weights (cpp:1298-1313)
    CF_OML1, CF_OML2,
    // leaf temperature (cpp:1333-1364)
    CF_EMG, CF_EMA, CF_INVLEAFD,
    // below-canopy Lagrangian model (cpp:1365-1409)
    CF_A2H, CF_INTHH, CF_INTHZ, CF_HGT, CF_INVHGT, CF_INVHMZ, CF_NEARFAC, CF_LEAFDEN, CF_OMEMPAI,
    // array forcing: per-cell solar geometry (cpp:2497): latitude and the longitude part B of the
    // hour angle tt = A(time) + B(cell), B = 0.261799*4*lon/60 (cpp:44, 54)
    CF_SINLAT, CF_COSLAT, CF_COSB, CF_SINB,
    // coarse array forcing: the cell's position in the coarse grid (rows, columns)
    CF_CROWPOS, CF_CCOLPOS,
    // ... and its altitude correction: interpolated coarse elevation - own elevation; sea-level -> own-level pressure factor
    CF_ELEVD, CF_PKFAC,
    CF_COUNT
};
constexpr int kCellDirs = 32;  // 24 horizon + 8 wind-shelter values follow the CF_ rows

// ---- per-timestep table (vector forcing) --------------------------------------
enum TimeField : int {
    // raw forcing / point-model series
    TF_TC = 0, TF_ES, TF_EA, TF_TDEW, TF_PK, TF_RSW, TF_RDIF, TF_RLW, TF_U2,
    TF_SOILMP, TF_GP, TF_UMU, TF_KP, TF_MUGP, TF_DTRP,
    // solar geometry
    TF_CZ, TF_SZ, TF_CAZ, TF_SAZ, TF_TANSA, TF_ZEND,
    // canopy extinction operands, radians call (cpp:2231) ...
    TF_COSC, TF_TAN2C, TF_TANC, TF_INV2COSC,
    // ... and the degrees call inside TVaboveground (cpp:1425)
    TF_TAN2B, TF_TANB, TF_INV2COSB,
    // Penman-Monteith operands (cpp:1220-1247)
    TF_DE, TF_GHRRAD, TF_REM, TF_LAPK, TF_MUPM, TF_INVMUPM, TF_WFAC, TF_GFAC,
    // beam normalisation (cpp:1122-1124)
    TF_RBEAM, TF_RB,
    // packed ints: sindex | windex<<5 | ksat<<8
    TF_IDX,
    TF_COUNT
};

__device__ __forceinline__ double na_real() { return __longlong_as_double((long long)kNaRealBits); }

// ---- lean fp64 elementary functions -------------------------------------------------
// The hot loop spends most of its issue slots in exp / log / divide.  The device
// libm versions are correctly rounded over the whole domain and pay for it (log: 98
// VALU instructions, exp: 22 + constants, divide: 11, sqrt: 20).  The versions
// below are valid on the domains this path produces (finite, normal operands; exp
// also for very negative arguments, where it returns 0) and are accurate to ~1 ulp
// (checked against numpy by tests/test_math_gpu.py through mcf_selftest_math).
// v_rcp_f64 / v_rsq_f64 deliver ~23 bits.  With e = 1 - b*r0 (|e| < 2^-23) the CUBIC step r1 = r0*(1 + e + e^2) leaves a
// relative error of e^3 < 2^-69, i.e. a correctly rounded-to-nearest-ish reciprocal in three FMAs — one fewer than two
// Newton steps.  (Measured and not shipped: a single Newton step, 46 bits — 57 VALU fewer per cell-step, +2.6 %, at a 15 x
// wider margin to the oracle, 2.9e-12.)
__device__ __forceinline__ double frcp(double b) {           // 1/b, b finite normal non-zero
    double r = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r, 1.0);
    return fma(fma(e, e, e), r, r);
}
// 1/b to 46 bits (relative error < 2^-46 = 1.4e-14): ONE Newton step, two FMAs instead of three.  For the quotients whose error
// reaches an output unamplified — the Penman-Monteith temperature rises (|dT| <= 80 K: < 1.2e-12 K), the series conductances,
// the Lagrangian far field's normalisation.  Whatever feeds an exponential's argument (the saturation pressures) or the
// two-stream coefficient algebra (which amplifies a rounding error 1e5-fold) keeps frcp's 69 bits.
__device__ __forceinline__ double frcp_m(double b) {
    double r = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r, 1.0);
    return fma(e, r, r);
}
__device__ __forceinline__ double fdiv_m(double a, double b) { return a * frcp_m(b); }
__device__ __forceinline__ double fdiv(double a, double b) {  // a/b, b finite normal non-zero
    // a * (1/b) with the 69-bit reciprocal above: one FMA fewer; the two roundings (of 1/b and of the product) bound the
    // error by 1.5 ulp (3.3e-16), checked by tests/test_math_gpu.py
    return a * frcp(b);
}
__device__ __forceinline__ double fsqrt(double x) {           // sqrt(x), x finite normal positive
    // coupled Goldschmidt step from the 23-bit v_rsq_f64 (g ~ sqrt x and h ~ 1/(2 sqrt x) to 46 bits), then ONE residual
    // correction g += (x - g^2) * h: the residual is exact in the FMA and h's 46 bits leave an error far below an ulp
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, x);
    return fma(e, h, g);
}
// sqrt(x) to 46 bits: the coupled step above without the residual correction (four instructions behind v_rsq_f64 instead of
// seven) — the leaf boundary-layer conductance, which enters the leaf's energy balance linearly
__device__ __forceinline__ double fsqrt_m(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}
// A 64-bit literal used as a VALU operand has to live in a register pair.  Left alone,
// hipcc copies every polynomial coefficient into a VGPR pair (v_mov_b64) in front of each
// v_fmac: one extra VALU issue per term.  The two functions below are therefore written as
// single asm blocks: coefficients are pinned to SGPR pairs ("s" constraint: s_mov_b32 on the
// otherwise idle scalar unit) and consumed by the VOP3 forms; nothing inside needs wait
// states (plain dependent VALU ops).
//
// exp(x): Cody-Waite reduction + degree-11 minimax polynomial on [-ln2/2, ln2/2].
// No overflow/underflow branches: v_cvt_i32_f64 saturates and v_ldexp_f64 saturates to
// inf / flushes to 0 by itself, so very negative arguments return exactly 0.
//
// The coefficients are carried in a MathK: k_solve pins them into SGPR pairs once, in front of the day loop
// (MathK::pin), so that the ~25 exp and ~7 log evaluations of a cell-step share them.  Rematerialised at every
// call (22 + 14 s_mov_b32 each) they made up ~70 % of the kernel's scalar instructions, and the scalar issue slots
// they take from their own wave cost 9 % of the run time (measured by doubling them).
// k_solve routes exp and log through tables in LDS (fexp_tab, flog_tab); exp's c11 / c5 and log's Lg6, Lg7 / 1/7 live in VGPR
// pairs for the whole day loop there: the first Horner step reads TWO constants and a VOP3 can take only one from the scalar
// file, so the other was re-created by two v_mov_b32 in front of (nearly) every evaluation.  Kernels that do not set
// MathK::tab keep the degree-11 polynomial.
// 2^(j/64), j = 0 .. 63, correctly rounded (computed with 60 decimal digits)
__device__ const double kExp2Tab[256] = {
    0x1.0000000000000p+0, 0x1.00b1afa5abcbfp+0, 0x1.0163da9fb3335p+0, 0x1.02168143b0281p+0,
    0x1.02c9a3e778061p+0, 0x1.037d42e11bbccp+0, 0x1.04315e86e7f85p+0, 0x1.04e5f72f654b1p+0,
    0x1.059b0d3158574p+0, 0x1.0650a0e3c1f89p+0, 0x1.0706b29ddf6dep+0, 0x1.07bd42b72a836p+0,
    0x1.0874518759bc8p+0, 0x1.092bdf66607e0p+0, 0x1.09e3ecac6f383p+0, 0x1.0a9c79b1f3919p+0,
    0x1.0b5586cf9890fp+0, 0x1.0c0f145e46c85p+0, 0x1.0cc922b7247f7p+0, 0x1.0d83b23395decp+0,
    0x1.0e3ec32d3d1a2p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0fb66affed31bp+0, 0x1.1073028d7233ep+0,
    0x1.11301d0125b51p+0, 0x1.11edbab5e2ab6p+0, 0x1.12abdc06c31ccp+0, 0x1.136a814f204abp+0,
    0x1.1429aaea92de0p+0, 0x1.14e95934f312ep+0, 0x1.15a98c8a58e51p+0, 0x1.166a45471c3c2p+0,
    0x1.172b83c7d517bp+0, 0x1.17ed48695bbc0p+0, 0x1.18af9388c8deap+0, 0x1.1972658375d2fp+0,
    0x1.1a35beb6fcb75p+0, 0x1.1af99f8138a1cp+0, 0x1.1bbe084045cd4p+0, 0x1.1c82f95281c6bp+0,
    0x1.1d4873168b9aap+0, 0x1.1e0e75eb44027p+0, 0x1.1ed5022fcd91dp+0, 0x1.1f9c18438ce4dp+0,
    0x1.2063b88628cd6p+0, 0x1.212be3578a819p+0, 0x1.21f49917ddc96p+0, 0x1.22bdda27912d1p+0,
    0x1.2387a6e756238p+0, 0x1.2451ffb82140ap+0, 0x1.251ce4fb2a63fp+0, 0x1.25e85711ece75p+0,
    0x1.26b4565e27cddp+0, 0x1.2780e341ddf29p+0, 0x1.284dfe1f56381p+0, 0x1.291ba7591bb70p+0,
    0x1.29e9df51fdee1p+0, 0x1.2ab8a66d10f13p+0, 0x1.2b87fd0dad990p+0, 0x1.2c57e39771b2fp+0,
    0x1.2d285a6e4030bp+0, 0x1.2df961f641589p+0, 0x1.2ecafa93e2f56p+0, 0x1.2f9d24abd886bp+0,
    0x1.306fe0a31b715p+0, 0x1.31432edeeb2fdp+0, 0x1.32170fc4cd831p+0, 0x1.32eb83ba8ea32p+0,
    0x1.33c08b26416ffp+0, 0x1.3496266e3fa2dp+0, 0x1.356c55f929ff1p+0, 0x1.36431a2de883bp+0,
    0x1.371a7373aa9cbp+0, 0x1.37f26231e754ap+0, 0x1.38cae6d05d866p+0, 0x1.39a401b7140efp+0,
    0x1.3a7db34e59ff7p+0, 0x1.3b57fbfec6cf4p+0, 0x1.3c32dc313a8e5p+0, 0x1.3d0e544ede173p+0,
    0x1.3dea64c123422p+0, 0x1.3ec70df1c5175p+0, 0x1.3fa4504ac801cp+0, 0x1.40822c367a024p+0,
    0x1.4160a21f72e2ap+0, 0x1.423fb2709468ap+0, 0x1.431f5d950a897p+0, 0x1.43ffa3f84b9d4p+0,
    0x1.44e086061892dp+0, 0x1.45c2042a7d232p+0, 0x1.46a41ed1d0057p+0, 0x1.4786d668b3237p+0,
    0x1.486a2b5c13cd0p+0, 0x1.494e1e192aed2p+0, 0x1.4a32af0d7d3dep+0, 0x1.4b17dea6db7d7p+0,
    0x1.4bfdad5362a27p+0, 0x1.4ce41b817c114p+0, 0x1.4dcb299fddd0dp+0, 0x1.4eb2d81d8abffp+0,
    0x1.4f9b2769d2ca7p+0, 0x1.508417f4531eep+0, 0x1.516daa2cf6642p+0, 0x1.5257de83f4eefp+0,
    0x1.5342b569d4f82p+0, 0x1.542e2f4f6ad27p+0, 0x1.551a4ca5d920fp+0, 0x1.56070dde910d2p+0,
    0x1.56f4736b527dap+0, 0x1.57e27dbe2c4cfp+0, 0x1.58d12d497c7fdp+0, 0x1.59c0827ff07ccp+0,
    0x1.5ab07dd485429p+0, 0x1.5ba11fba87a03p+0, 0x1.5c9268a5946b7p+0, 0x1.5d84590998b93p+0,
    0x1.5e76f15ad2148p+0, 0x1.5f6a320dceb71p+0, 0x1.605e1b976dc09p+0, 0x1.6152ae6cdf6f4p+0,
    0x1.6247eb03a5585p+0, 0x1.633dd1d1929fdp+0, 0x1.6434634ccc320p+0, 0x1.652b9febc8fb7p+0,
    0x1.6623882552225p+0, 0x1.671c1c70833f6p+0, 0x1.68155d44ca973p+0, 0x1.690f4b19e9538p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6b052fa75173ep+0, 0x1.6c012750bdabfp+0, 0x1.6cfdcddd47645p+0,
    0x1.6dfb23c651a2fp+0, 0x1.6ef9298593ae5p+0, 0x1.6ff7df9519484p+0, 0x1.70f7466f42e87p+0,
    0x1.71f75e8ec5f74p+0, 0x1.72f8286ead08ap+0, 0x1.73f9a48a58174p+0, 0x1.74fbd35d7cbfdp+0,
    0x1.75feb564267c9p+0, 0x1.77024b1ab6e09p+0, 0x1.780694fde5d3fp+0, 0x1.790b938ac1cf6p+0,
    0x1.7a11473eb0187p+0, 0x1.7b17b0976cfdbp+0, 0x1.7c1ed0130c132p+0, 0x1.7d26a62ff86f0p+0,
    0x1.7e2f336cf4e62p+0, 0x1.7f3878491c491p+0, 0x1.80427543e1a12p+0, 0x1.814d2add106d9p+0,
    0x1.82589994cce13p+0, 0x1.8364c1eb941f7p+0, 0x1.8471a4623c7adp+0, 0x1.857f4179f5b21p+0,
    0x1.868d99b4492edp+0, 0x1.879cad931a436p+0, 0x1.88ac7d98a6699p+0, 0x1.89bd0a478580fp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8be05bad61778p+0, 0x1.8cf3216b5448cp+0, 0x1.8e06a5e0866d9p+0,
    0x1.8f1ae99157736p+0, 0x1.902fed0282c8ap+0, 0x1.9145b0b91ffc6p+0, 0x1.925c353aa2fe2p+0,
    0x1.93737b0cdc5e5p+0, 0x1.948b82b5f98e5p+0, 0x1.95a44cbc8520fp+0, 0x1.96bdd9a7670b3p+0,
    0x1.97d829fde4e50p+0, 0x1.98f33e47a22a2p+0, 0x1.9a0f170ca07bap+0, 0x1.9b2bb4d53fe0dp+0,
    0x1.9c49182a3f090p+0, 0x1.9d674194bb8d5p+0, 0x1.9e86319e32323p+0, 0x1.9fa5e8d07f29ep+0,
    0x1.a0c667b5de565p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a309bec4a2d33p+0, 0x1.a42c980460ad8p+0,
    0x1.a5503b23e255dp+0, 0x1.a674a8af46052p+0, 0x1.a799e1330b358p+0, 0x1.a8bfe53c12e59p+0,
    0x1.a9e6b5579fdbfp+0, 0x1.ab0e521356ebap+0, 0x1.ac36bbfd3f37ap+0, 0x1.ad5ff3a3c2774p+0,
    0x1.ae89f995ad3adp+0, 0x1.afb4ce622f2ffp+0, 0x1.b0e07298db666p+0, 0x1.b20ce6c9a8952p+0,
    0x1.b33a2b84f15fbp+0, 0x1.b468415b749b1p+0, 0x1.b59728de5593ap+0, 0x1.b6c6e29f1c52ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.b928cf22749e4p+0, 0x1.ba5b030a1064ap+0, 0x1.bb8e0b79a6f1fp+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bdf69c3f3a207p+0, 0x1.bf2c25bd71e09p+0, 0x1.c06286141b33dp+0,
    0x1.c199bdd85529cp+0, 0x1.c2d1cd9fa652cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c544778fafb22p+0,
    0x1.c67f12e57d14bp+0, 0x1.c7ba88988c933p+0, 0x1.c8f6d9406e7b5p+0, 0x1.ca3405751c4dbp+0,
    0x1.cb720dcef9069p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cdf0b555dc3fap+0, 0x1.cf3155b5bab74p+0,
    0x1.d072d4a07897cp+0, 0x1.d1b532b08c968p+0, 0x1.d2f87080d89f2p+0, 0x1.d43c8eacaa1d6p+0,
    0x1.d5818dcfba487p+0, 0x1.d6c76e862e6d3p+0, 0x1.d80e316c98398p+0, 0x1.d955d71ff6075p+0,
    0x1.da9e603db3285p+0, 0x1.dbe7cd63a8315p+0, 0x1.dd321f301b460p+0, 0x1.de7d5641c0658p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e11676b197d17p+0, 0x1.e264614f5a129p+0, 0x1.e3b333b16ee12p+0,
    0x1.e502ee78b3ff6p+0, 0x1.e653924676d76p+0, 0x1.e7a51fbc74c83p+0, 0x1.e8f7977cdb740p+0,
    0x1.ea4afa2a490dap+0, 0x1.eb9f4867cca6ep+0, 0x1.ecf482d8e67f1p+0, 0x1.ee4aaa2188510p+0,
    0x1.efa1bee615a27p+0, 0x1.f0f9c1cb6412ap+0, 0x1.f252b376bba97p+0, 0x1.f3ac948dd7274p+0,
    0x1.f50765b6e4540p+0, 0x1.f6632798844f8p+0, 0x1.f7bfdad9cbe14p+0, 0x1.f91d802243c89p+0,
    0x1.fa7c1819e90d8p+0, 0x1.fbdba3692d514p+0, 0x1.fd3c22b8f71f1p+0, 0x1.fe9d96b2a23d9p+0};
// log table (tools/gen_math_tables.py log): interval j of the mantissa m in [0.5, 1) is [0.5 + j/512, 0.5 + (j+1)/512); pairs
// (c_j, l_j) with m c_j - 1 = r, |r| <= 2^-8 and l_j = -log of the reciprocal used — see flog_tab.
constexpr int kLogSplit = 106;     // intervals below reduce 2m towards 1 (exponent - 1), the others m towards 1
__device__ const double kLogTab[512] = {
    0x1.0000000000000p+1, 0x0.0p+0, 0x1.fd04794a10e6ap+0, 0x1.7ee11ebd82ec4p-8,
    0x1.fb0c610d5e939p+0, 0x1.3e7295d25a7d5p-7, 0x1.f9182b6813bafp+0, 0x1.bcf712c743853p-7,
    0x1.f727cce5f530ap+0, 0x1.1d7f7eb9eebf1p-6, 0x1.f53b3a3fa204ep+0, 0x1.5c45a51b8d393p-6,
    0x1.f3526859b8cecp+0, 0x1.9ace7551cc515p-6, 0x1.f16d4c4401f17p+0, 0x1.d91a66c543cbep-6,
    0x1.ef8bdb389ebadp+0, 0x1.0b94f7c196173p-5, 0x1.edae0a9b3d3a5p+0, 0x1.2a7ec2214e879p-5,
    0x1.ebd3cff850b0cp+0, 0x1.494acc34d911dp-5, 0x1.e9fd21044e799p+0, 0x1.67f94f094bd92p-5,
    0x1.e829f39aef509p+0, 0x1.868a83083f6d0p-5, 0x1.e65a3dbe74d6bp+0, 0x1.a4fe9ffa3d233p-5,
    0x1.e48df596f3394p+0, 0x1.c355dd0921f2fp-5, 0x1.e2c511719ee16p+0, 0x1.e19070c276010p-5,
    0x1.e0ff87c01e100p+0, 0x1.ffae9119b92fbp-5, 0x1.df3d4f17de4dbp+0, 0x1.0ed839b5526fep-4,
    0x1.dd7e5e316d94cp+0, 0x1.1dcb263db1944p-4, 0x1.dbc2abe7d71d4p+0, 0x1.2cb0283f5de22p-4,
    0x1.da0a2f3803b41p+0, 0x1.3b87598b1b6f0p-4, 0x1.d854df401d855p+0, 0x1.4a50d3aa1b03fp-4,
    0x1.d6a2b33ef7448p+0, 0x1.590cafdf01c26p-4, 0x1.d4f3a293769cap+0, 0x1.67bb0726ec0fbp-4,
    0x1.d347a4bc01d34p+0, 0x1.765bf23a6be17p-4, 0x1.d19eb155f08a4p+0, 0x1.84ef898e82828p-4,
    0x1.cff8c01cff8c0p+0, 0x1.9375e55595edfp-4, 0x1.ce55c8eac7900p+0, 0x1.a1ef1d8061cd8p-4,
    0x1.ccb5c3b636e3ap+0, 0x1.b05b49bee4403p-4, 0x1.cb18a8930de60p+0, 0x1.beba818146764p-4,
    0x1.c97e6fb15e44dp+0, 0x1.cd0cdbf8c13e0p-4, 0x1.c7e7115d0ce95p+0, 0x1.db5270187d925p-4,
    0x1.c65285fd56843p+0, 0x1.e98b54967146bp-4, 0x1.c4c0c61456a8ep+0, 0x1.f7b79fec37de2p-4,
    0x1.c331ca3e91679p+0, 0x1.02ebb42bf3d4ap-3, 0x1.c1a58b327f576p+0, 0x1.09f561ee719c4p-3,
    0x1.c01c01c01c01cp+0, 0x1.10f8e422539b1p-3, 0x1.be9526d0769fap+0, 0x1.17f6458fca611p-3,
    0x1.bd10f365451b6p+0, 0x1.1eed90e2dc2c3p-3, 0x1.bb8f609879493p+0, 0x1.25ded0abc6ad3p-3,
    0x1.ba10679bd8488p+0, 0x1.2cca0f5f5f252p-3, 0x1.b89401b89401cp+0, 0x1.33af575770e4dp-3,
    0x1.b71a284ee6b34p+0, 0x1.3a8eb2d31a375p-3, 0x1.b5a2d4d5b081fp+0, 0x1.41682bf727bbfp-3,
    0x1.b42e00da17007p+0, 0x1.483bccce6e3dcp-3, 0x1.b2bba5ff26a23p+0, 0x1.4f099f4a230b1p-3,
    0x1.b14bbdfd760e6p+0, 0x1.55d1ad4232d70p-3, 0x1.afde42a2cb482p+0, 0x1.5c940075972b9p-3,
    0x1.ae732dd1c2a09p+0, 0x1.6350a28aaa759p-3, 0x1.ad0a798177693p+0, 0x1.6a079d0f7aad0p-3,
    0x1.aba41fbd2e5b1p+0, 0x1.70b8f97a1aa74p-3, 0x1.aa401aa401aa4p+0, 0x1.7764c128f2127p-3,
    0x1.a8de64688ebabp+0, 0x1.7e0afd630c276p-3, 0x1.a77ef750a56dap+0, 0x1.84abb75865137p-3,
    0x1.a621cdb4f8fdfp+0, 0x1.8b46f8223625bp-3, 0x1.a4c6e200d2637p+0, 0x1.91dcc8c340bdfp-3,
    0x1.a36e2eb1c432dp+0, 0x1.986d3228180c8p-3, 0x1.a217ae575ff2fp+0, 0x1.9ef83d2769a34p-3,
    0x1.a0c35b92ecdf1p+0, 0x1.a57df28244dcbp-3, 0x1.9f713117200d0p+0, 0x1.abfe5ae46124ap-3,
    0x1.9e2129a7d5f0ap+0, 0x1.b2797ee46320cp-3, 0x1.9cd34019cd340p+0, 0x1.b8ef670420c3bp-3,
    0x1.9b876f5262dd1p+0, 0x1.bf601bb0e44e0p-3, 0x1.9a3db2474fb98p+0, 0x1.c5cba543ae424p-3,
    0x1.98f603fe670a0p+0, 0x1.cc320c0176501p-3, 0x1.97b05f8d56652p+0, 0x1.d293581b6b3e7p-3,
    0x1.966cc01966cc0p+0, 0x1.d8ef91af31d5ep-3, 0x1.952b20d73ee97p+0, 0x1.df46c0c722d30p-3,
    0x1.93eb7d0aa6759p+0, 0x1.e598ed5a87e2ep-3, 0x1.92add0064ab74p+0, 0x1.ebe61f4dd7b0bp-3,
    0x1.9172152b841ddp+0, 0x1.f22e5e72f105cp-3, 0x1.903847ea1cec1p+0, 0x1.f871b28955045p-3,
    0x1.8f0063c018f00p+0, 0x1.feb0233e607cep-3, 0x1.8dca64397e408p+0, 0x1.0274dc16c232fp-2,
    0x1.8c9644f01efbcp+0, 0x1.058f3c703ebc5p-2, 0x1.8b64018b64019p+0, 0x1.08a73667c57aep-2,
    0x1.8a3395c018a34p+0, 0x1.0bbccdb0d24bcp-2, 0x1.8904fd503744bp+0, 0x1.0ed005f657da5p-2,
    0x1.87d8340ab6e97p+0, 0x1.11e0e2dad9cb6p-2, 0x1.86ad35cb59a84p+0, 0x1.14ef67f88685ap-2,
    0x1.8583fe7a7c018p+0, 0x1.17fb98e15095ep-2, 0x1.845c8a0ce5129p+0, 0x1.1b05791f07b4ap-2,
    0x1.8336d48397a24p+0, 0x1.1e0d0c33716bdp-2, 0x1.8212d9eba4018p+0, 0x1.211255986160cp-2,
    0x1.80f0965dfabcbp+0, 0x1.241558bfd1405p-2, 0x1.7fd005ff40180p+0, 0x1.27161913f853dp-2,
    0x1.7eb124ffa053bp+0, 0x1.2a1499f762bcap-2, 0x1.7d93ef9aa4b46p+0, 0x1.2d10dec508582p-2,
    0x1.7c7862170949fp+0, 0x1.300aead06350cp-2, 0x1.7b5e78c693733p+0, 0x1.3302c1658658ap-2,
    0x1.7a463005e918cp+0, 0x1.35f865c93293ep-2, 0x1.792f843c689c3p+0, 0x1.38ebdb38ed320p-2,
    0x1.781a71dc01782p+0, 0x1.3bdd24eb14b69p-2, 0x1.7706f5610d8d0p+0, 0x1.3ecc460ef5f50p-2,
    0x1.75f50b522b17cp+0, 0x1.41b941cce0beep-2, 0x1.74e4b040174e5p+0, 0x1.44a41b463c47bp-2,
    0x1.73d5e0c5899f7p+0, 0x1.478cd5959b3d8p-2, 0x1.72c899870f91fp+0, 0x1.4a7373cecf997p-2,
    0x1.71bcd732e940ap+0, 0x1.4d57f8fefe27fp-2, 0x1.70b29680e66fap+0, 0x1.503a682cb1cb3p-2,
    0x1.6fa9d43244380p+0, 0x1.531ac457ee77fp-2, 0x1.6ea28d118b474p+0, 0x1.55f9107a43ee2p-2,
    0x1.6d9cbdf26eaefp+0, 0x1.58d54f86e02f3p-2, 0x1.6c9863b1ab429p+0, 0x1.5baf846aa1b1ap-2,
    0x1.6b957b34e7803p+0, 0x1.5e87b20c2954ap-2, 0x1.6a94016a94017p+0, 0x1.615ddb4bec13cp-2,
    0x1.6993f349cc726p+0, -0x1.61965cdb02c1ep-2, 0x1.68954dd2390bap+0, -0x1.5ec433d5c35aep-2,
    0x1.67980e0bf08c7p+0, -0x1.5bf406b543db1p-2, 0x1.669c31075ab40p+0, -0x1.5925d2b112a59p-2,
    0x1.65a1b3dd13357p+0, -0x1.565995069514cp-2, 0x1.64a893adcd25fp+0, -0x1.538f4af8f72fcp-2,
    0x1.63b0cda236e1cp+0, -0x1.50c6f1d11b97bp-2, 0x1.62ba5eeade65ep+0, -0x1.4e0086dd8baccp-2,
    0x1.61c544c0161c5p+0, -0x1.4b3c077267e9ap-2, 0x1.60d17c61da198p+0, -0x1.487970e958771p-2,
    0x1.5fdf0317b5c6fp+0, -0x1.45b8c0a17df12p-2, 0x1.5eedd630a9fb3p+0, -0x1.42f9f3ff62641p-2,
    0x1.5dfdf303137b6p+0, -0x1.403d086cea79bp-2, 0x1.5d0f56ec91e57p+0, -0x1.3d81fb5946dbcp-2,
    0x1.5c21ff51ef005p+0, -0x1.3ac8ca38e5c5dp-2, 0x1.5b35e99f06714p+0, -0x1.3811728564cb2p-2,
    0x1.5a4b1346add2bp+0, -0x1.355bf1bd82c8bp-2, 0x1.596179c29d2cep+0, -0x1.32a84565120a9p-2,
    0x1.58791a9357ccep+0, -0x1.2ff66b04ea9d5p-2, 0x1.5791f34015792p+0, -0x1.2d46602adccefp-2,
    0x1.56ac0156ac015p+0, -0x1.2a982269a3dbep-2, 0x1.55c7426b79286p+0, -0x1.27ebaf58d8c9cp-2,
    0x1.54e3b4194ce66p+0, -0x1.25410494e56c8p-2, 0x1.5401540154015p+0, -0x1.22981fbef797ap-2,
    0x1.53201fcb02fb1p+0, -0x1.1ff0fe7cf47a9p-2, 0x1.5240152401524p+0, -0x1.1d4b9e796c245p-2,
    0x1.516131c015161p+0, -0x1.1aa7fd638d33ep-2, 0x1.508373590ec9cp+0, -0x1.180618ef18adep-2,
    0x1.4fa6d7aeb597cp+0, -0x1.1565eed455fc2p-2, 0x1.4ecb5c86b3d24p+0, -0x1.12c77cd00713cp-2,
    0x1.4df0ffac83c01p+0, -0x1.102ac0a35cc1bp-2, 0x1.4d17bef15cb4ep+0, -0x1.0d8fb813eb1efp-2,
    0x1.4c3f982c20723p+0, -0x1.0af660eb9e278p-2, 0x1.4b68893948d1cp+0, -0x1.085eb8f8ae799p-2,
    0x1.4a928ffad5b5cp+0, -0x1.05c8be0d9635ap-2, 0x1.49bdaa583b401p+0, -0x1.03346e0106062p-2,
    0x1.48e9d63e504d1p+0, -0x1.00a1c6adda472p-2, 0x1.4817119f3d325p+0, -0x1.fc218be620a5fp-3,
    0x1.47455a726abf2p+0, -0x1.f702d36777df0p-3, 0x1.4674aeb4717e9p+0, -0x1.f1e75fadf9bdep-3,
    0x1.45a50c670938fp+0, -0x1.eccf2c8fe920bp-3, 0x1.44d67190f8b43p+0, -0x1.e7ba35eb77e2ap-3,
    0x1.4408dc3e05b22p+0, -0x1.e2a877a6b2c0fp-3, 0x1.433c4a7ee52b4p+0, -0x1.dd99edaf6d7e9p-3,
    0x1.4270ba692bc4dp+0, -0x1.d88e93fb2f451p-3, 0x1.41a62a173e821p+0, -0x1.d38666871f467p-3,
    0x1.40dc97a843ae8p+0, -0x1.ce816157f1985p-3, 0x1.4014014014014p+0, -0x1.c97f8079d44ecp-3,
    0x1.3f4c65072bf74p+0, -0x1.c480c0005cccfp-3, 0x1.3e85c12a9d651p+0, -0x1.bf851c067555cp-3,
    0x1.3dc013dc013dcp+0, -0x1.ba8c90ae4ad19p-3, 0x1.3cfb5b51698ebp+0, -0x1.b5971a213acd9p-3,
    0x1.3c3795c553afbp+0, -0x1.b0a4b48fc1b44p-3, 0x1.3b74c1769aa5cp+0, -0x1.abb55c31693aep-3,
    0x1.3ab2dca869b81p+0, -0x1.a6c90d44b704cp-3, 0x1.39f1e5a22f36ep+0, -0x1.a1dfc40f1b7f1p-3,
    0x1.3931daaf8f721p+0, -0x1.9cf97cdce0ec1p-3, 0x1.3872ba2057e04p+0, -0x1.981634011aa74p-3,
    0x1.37b4824872744p+0, -0x1.9335e5d594985p-3, 0x1.36f7317fd9212p+0, -0x1.8e588ebac2dc1p-3,
    0x1.363ac622898b1p+0, -0x1.897e2b17b19a6p-3, 0x1.357f3e9078e5bp+0, -0x1.84a6b759f512dp-3,
    0x1.34c4992d87fd9p+0, -0x1.7fd22ff599d4cp-3, 0x1.340ad461776d3p+0, -0x1.7b0091651528bp-3,
    0x1.3351ee97dbfc6p+0, -0x1.7631d82935a84p-3, 0x1.3299e6401329ap+0, -0x1.716600c914055p-3,
    0x1.31e2b9cd37dc2p+0, -0x1.6c9d07d203fc4p-3, 0x1.312c67b6173eep+0, -0x1.67d6e9d785770p-3,
    0x1.3076ee7525c2cp+0, -0x1.6313a37335d76p-3, 0x1.2fc24c8874486p+0, -0x1.5e533144c1718p-3,
    0x1.2f0e8071a5703p+0, -0x1.59958ff1d52f4p-3, 0x1.2e5b88b5e3104p+0, -0x1.54dabc26105d3p-3,
    0x1.2da963ddd3cfbp+0, -0x1.5022b292f6a45p-3, 0x1.2cf8107590e67p+0, -0x1.4b6d6fefe22a5p-3,
    0x1.2c478d0c9c013p+0, -0x1.46baf0f9f5db8p-3, 0x1.2b97d835d548ep+0, -0x1.420b32740fdd6p-3,
    0x1.2ae8f087718d0p+0, -0x1.3d5e3126bc281p-3, 0x1.2a3ad49af0907p+0, -0x1.38b3e9e027477p-3,
    0x1.298d830d13780p+0, -0x1.340c59741142dp-3, 0x1.28e0fa7dd35a3p+0, -0x1.2f677cbbc0a98p-3,
    0x1.2835399057efdp+0, -0x1.2ac55095f5c5bp-3, 0x1.278a3eeaee650p+0, -0x1.2625d1e6ddf55p-3,
    0x1.26e009370049cp+0, -0x1.2188fd9807266p-3, 0x1.263697210aa18p+0, -0x1.1ceed09853755p-3,
    0x1.258de75895121p+0, -0x1.185747dbecf34p-3, 0x1.24e5f89029305p+0, -0x1.13c2605c398bfp-3,
    0x1.243ec97d49eaep+0, -0x1.0f301717cf0fbp-3, 0x1.239858d86b11fp+0, -0x1.0aa06912675d5p-3,
    0x1.22f2a55ce8fc5p+0, -0x1.06135354d4b19p-3, 0x1.224dadc900489p+0, -0x1.0188d2ecf613ep-3,
    0x1.21a970ddc5ba7p+0, -0x1.fa01c9db57ce7p-4, 0x1.2105ed5f1e336p+0, -0x1.f0f70cdd992e4p-4,
    0x1.20632213b6c6dp+0, -0x1.e7f1691a32d3ap-4, 0x1.1fc10dc4fce8bp+0, -0x1.def0d8d466dbbp-4,
    0x1.1f1faf3f16b64p+0, -0x1.d5f55659210e1p-4, 0x1.1e7f0550db594p+0, -0x1.ccfedbfee13a8p-4,
    0x1.1ddf0ecbcb841p+0, -0x1.c40d6425a5cb4p-4, 0x1.1d3fca840a074p+0, -0x1.bb20e936d6976p-4,
    0x1.1ca13750547fep+0, -0x1.b23965a52ff04p-4, 0x1.1c035409fc1dfp+0, -0x1.a956d3ecade60p-4,
    0x1.1b661f8cde833p+0, -0x1.a0792e9277cadp-4, 0x1.1ac998b75eb90p+0, -0x1.97a07024cbe6ep-4,
    0x1.1a2dbe6a5e3e4p+0, -0x1.8ecc933aeb6e2p-4, 0x1.19928f89362b7p+0, -0x1.85fd927506a46p-4,
    0x1.18f80af9b06dcp+0, -0x1.7d33687c293c8p-4, 0x1.185e2fa401186p+0, -0x1.746e100226edbp-4,
    0x1.17c4fc72bfcb9p+0, -0x1.6bad83c1883bap-4, 0x1.172c7052e1316p+0, -0x1.62f1be7d7774ap-4,
    0x1.16948a33b08fap+0, -0x1.5a3abb01ade21p-4, 0x1.15fd4906c96f1p+0, -0x1.5188742261311p-4,
    0x1.1566abc011567p+0, -0x1.48dae4bc3101dp-4, 0x1.14d0b155b19aep+0, -0x1.403207b414b79p-4,
    0x1.143b58c01143bp+0, -0x1.378dd7f74970fp-4, 0x1.13a6a0f9cf01ep+0, -0x1.2eee507b402ffp-4,
    0x1.131288ffbb3b6p+0, -0x1.26536c3d8c36cp-4, 0x1.127f0fd0d2295p+0, -0x1.1dbd2643d1913p-4,
    0x1.11ec346e36092p+0, -0x1.152b799bb3cd0p-4, 0x1.1159f5db29606p+0, -0x1.0c9e615ac4e19p-4,
    0x1.10c8531d0952ep+0, -0x1.0415d89e7444bp-4, 0x1.10374b3b480aap+0, -0x1.f723b517fc51fp-5,
    0x1.0fa6dd3f67322p+0, -0x1.e624c4a0b5e15p-5, 0x1.0f170834f27fap+0, -0x1.d52ed6405d87ap-5,
    0x1.0e87cb297a51ep+0, -0x1.c441e06f72a93p-5, 0x1.0df9252c8e5e6p+0, -0x1.b35dd9b58baa8p-5,
    0x1.0d6b154fb86f9p+0, -0x1.a282b8a936174p-5, 0x1.0cdd9aa677344p+0, -0x1.91b073efd7314p-5,
    0x1.0c50b446391f3p+0, -0x1.80e7023d8ccc8p-5, 0x1.0bc4614657569p+0, -0x1.70265a550e77bp-5,
    0x1.0b38a0c010b39p+0, -0x1.5f6e73078efc3p-5, 0x1.0aad71ce84d16p+0, -0x1.4ebf43349e26ap-5,
    0x1.0a22d38eaf2bfp+0, -0x1.3e18c1ca0ae99p-5, 0x1.0998c51f624d5p+0, -0x1.2d7ae5c3c5bb7p-5,
    0x1.090f45a1430aap+0, -0x1.1ce5a62bc3540p-5, 0x1.08865436c3cf7p+0, -0x1.0c58fa19dfaabp-5,
    0x1.07fdf0041ff7cp+0, -0x1.f7a9b16782855p-6, 0x1.0776182f57386p+0, -0x1.d6b272597981fp-6,
    0x1.06eecbe029155p+0, -0x1.b5cc258b718e7p-6, 0x1.06680a4010668p+0, -0x1.94f6b99a24473p-6,
    0x1.05e1d27a3ee9cp+0, -0x1.74321d3d006d2p-6, 0x1.055c23bb98e2ap+0, -0x1.537e3f45f354ep-6,
    0x1.04d6fd32b0c7bp+0, -0x1.32db0ea132e10p-6, 0x1.04525e0fc2fcbp+0, -0x1.12487a5507f68p-6,
    0x1.03ce4584b19a0p+0, -0x1.e38ce30333100p-7, 0x1.034ab2c50040dp+0, -0x1.a2a9c6c17044dp-7,
    0x1.02c7a505cffbfp+0, -0x1.61e77e8b53f9fp-7, 0x1.02451b7ddb2d2p+0, -0x1.2145e939ef1bcp-7,
    0x1.01c315657186bp+0, -0x1.c189cbb0e283fp-8, 0x1.014191f674111p+0, -0x1.40c8a7478788dp-8,
    0x1.00c0906c513cfp+0, -0x1.809048289860ap-9, 0x1.0000000000000p+0, 0x0.0p+0};
struct MathK {
    double e[12];   // exp: 1/ln2, -ln2_hi, -ln2_lo, c10 .. c2
    double l[7];    // log: Lg4, Lg5, Lg2, Lg3, Lg1, ln2_lo, ln2_hi
    double c11, lg6, lg7;   // VGPR residents (MCF_PIN_VCONST)
    double t[5];    // table exp: 256/ln2, -(ln2/256)_hi (30 bits: n * hi is exact), -(ln2/256)_lo, 1/24, -ln2/256 (53 bits: fexp_tab SHORT)
    double c5;      // 1/6 (VGPR resident: an instruction reads one scalar operand)
    double g[7];    // table log: (unused), -1/6, 1/5, -1/4, 1/3, ln2_lo, ln2_hi
    double g7v;     // 1/7 (VGPR resident: the first Horner step has two constants)
    double magic;   // 1.5 * 2^52 (VGPR resident): x * 256/ln2 + magic holds round(x * 256/ln2) in its low mantissa bits
    int sh3;        // 3 (VGPR resident): shift operand of the SDWA instruction that makes the table's byte offset
    bool vfast = false;   // magic / sh3 are pinned (k_solve's vector-forcing kernels; the array-forcing ones have no VGPR to spare)
    const double* ltab = nullptr;  // LDS copy of kLogTab
    const double* tab = nullptr;   // LDS copy of kExp2Tab
    bool logtab = false;           // flog goes through ltab (set with use_log_table; same reason)
    bool table = false;            // fexp goes through it (a compile-time fact after inlining: LDS address 0 is valid, so
                                   // the pointer cannot say)
    __device__ __forceinline__ void set() {
        c11 = 0x1.ade156a5dcb37p-26; lg6 = 1.531383769920937332e-01; lg7 = 1.479819860511658591e-01;
        magic = 0x1.8p52; sh3 = 3;
        t[0] = 0x1.71547652b82fep+8; t[1] = -0x1.62e42fec00000p-9; t[2] = -0x1.d1cf79abc9e3bp-40;
        t[3] = 1.0 / 24.0; t[4] = -0x1.62e42fefa39efp-9; c5 = 1.0 / 6.0;
        g[0] = 0.0; g7v = 1.0 / 7.0; g[1] = -1.0 / 6.0; g[2] = 1.0 / 5.0; g[3] = -0.25; g[4] = 1.0 / 3.0;
        g[5] = 1.90821492927058770002e-10; g[6] = 6.93147180369123816490e-01;
        e[0] = 0x1.71547652b82fep+0; e[1] = -0x1.62e42fefa39efp-1; e[2] = -0x1.abc9e3b39803fp-56;
        e[3] = 0x1.28af3fca7ab0cp-22; e[4] = 0x1.71dee623fde64p-19; e[5] = 0x1.a01997c89e6b0p-16;
        e[6] = 0x1.a01a014761f6ep-13; e[7] = 0x1.6c16c1852b7b0p-10; e[8] = 0x1.1111111122322p-7;
        e[9] = 0x1.55555555502a1p-5; e[10] = 0x1.5555555555511p-3; e[11] = 0x1.000000000000bp-1;
        l[0] = 2.222219843214978396e-01; l[1] = 1.818357216161805012e-01; l[2] = 3.999999999940941908e-01;
        l[3] = 2.857142874366239149e-01; l[4] = 6.666666666666735130e-01; l[5] = 1.90821492927058770002e-10;
        l[6] = 6.93147180369123816490e-01;
    }
    // makes the values opaque SGPR residents: the compiler can no longer re-create them from literals
    // copies the table into `lds` (64 doubles; the caller's barrier makes it visible) and switches fexp to it
    __device__ __forceinline__ void use_table(double* lds, int tid) {
        if (tid < 256) lds[tid] = kExp2Tab[tid];
        tab = lds;
        table = true;
    }
    // `lds`: 512 doubles for kLogTab; switches flog to the table route (with use_table)
    __device__ __forceinline__ void use_log_table(double* lds, int tid, int nthreads) {
        for (int i = tid; i < 512; i += nthreads) lds[i] = kLogTab[i];
        ltab = lds;
        logtab = true;
    }
    // tables already in LDS (k_solve fills them with its other prologue loads)
    __device__ __forceinline__ void tables(const double* exp_lds, const double* log_lds) {
        tab = exp_lds; table = true;
        ltab = log_lds; logtab = true;
    }
    // vconst: c5 / c11, lg6, lg7 held in VGPRs as well (an instruction reads one scalar operand, so a second constant costs
    // two moves wherever it is used) — not in the array-forcing kernels, which have no register to spare
    // lean_fast: only the bounded exp's two VGPR residents (three registers) — the array-forcing kernels
    __device__ __forceinline__ void pin(bool with_log, bool vconst = true, bool lean_fast = false) {
        if (table) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+s"(t[i]));
            if (vconst) asm volatile("" : "+s"(t[4]));
            if (vconst) asm volatile("" : "+v"(c5));
            if (vconst || lean_fast) {
                asm volatile("" : "+v"(magic));
                asm volatile("" : "+v"(sh3));
                vfast = true;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("" : "+s"(e[i]));
        }
        if (with_log) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (!logtab) asm volatile("" : "+s"(l[i]));
                else if (i > 0) asm volatile("" : "+s"(g[i]));
            if (logtab && vconst) asm volatile("" : "+v"(g7v));
        }
        if (!vconst) return;
        if (!table) asm volatile("" : "+v"(c11));
        if (with_log && !logtab) { asm volatile("" : "+v"(lg6)); asm volatile("" : "+v"(lg7)); }
    }
};
// exp(x) = 2^e * 2^(j/256) * exp(r), n = round(x * 256/ln2) = 256 e + j, r = x - n ln2/256, |r| <= ln2/512: the table value
// T comes from LDS (2 KB) while the degree-4 polynomial p = exp(r) - 1 is evaluated (r^5/120 < 3.8e-17), then T + T p and
// ldexp.  Same saturation behaviour as the polynomial route (v_cvt_i32_f64 and v_ldexp_f64 saturate; NaN stays NaN).
// BOUNDED = true: the caller guarantees |x| < 5e6 (then n = round(x * 256/ln2) fits 32 bits): n comes out of ONE fma as
// the low mantissa bits of x * 256/ln2 + 1.5 * 2^52 — no v_rndne, no v_cvt — and as a double by one subtraction.  Two VALU
// instructions fewer; no saturation for huge arguments, which is why the two-stream transmissions (arguments down to -inf)
// keep the general form.  NaN stays NaN in both.
// SHORT (with BOUNDED, round 5): r = x - n ln2/256 in ONE fma against the 53-bit constant instead of the hi / lo pair.  The
// constant's rounding (9e-20 of 2.7e-3) leaves an error of 3.3e-17 |x| in r, i.e. a RELATIVE error of |x| 2^-54 in the result
// on top of the polynomial's — for arguments that are themselves rounded quotients or products of modest size (the
// saturation pressures' a t / (t + b), |.| < 30; the stomatal and ground-wetness exponents): their own rounding error,
// >= |x| 2^-53, is amplified the same way, so the result loses less than half of what the argument already lost.
template <bool BOUNDED, bool SHORT = false>
__device__ __forceinline__ double fexp_tab(double x, const MathK& K) {
    double n, r, p, r2, out;
    int t;
    // (the table index is ready after the first (bounded) or second instruction: the LDS read is issued there and the
    // reduction's two FMAs run under its latency as well as the polynomial's four)
    if (BOUNDED && K.vfast) {
        double y;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"(x), "s"(K.t[0]), "v"(K.magic));    // x * 256/ln2 + 1.5 * 2^52: an integer in the low mantissa bits
        t = __double2loint(y);
        asm("v_add_f64 %0, %1, -%2" : "=v"(n) : "v"(y), "v"(K.magic));                      // n (exact)
    } else {
        asm("v_mul_f64 %0, %1, %2\n\t"
            "v_rndne_f64 %0, %0"
            : "=v"(n) : "v"(x), "s"(K.t[0]));
        asm("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(n));
    }
    if (BOUNDED && SHORT && K.vfast) {
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(n), "s"(K.t[4]), "v"(x));
    } else {
        asm("v_fma_f64 %0, %1, %2, %3\n\t"
            "v_fma_f64 %0, %1, %4, %0"
            : "=&v"(r) : "v"(n), "s"(K.t[1]), "v"(x), "s"(K.t[2]));
    }
    double T;
    if (K.vfast) {
        // byte offset of table entry t & 255 in ONE instruction (SDWA byte select + shift); its shift operand has to be a VGPR,
        // which the compiler re-created with a v_mov in front of every exp
        unsigned off;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
            : "=v"(off) : "v"(K.sh3), "v"(t));
        T = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(K.tab) + off);
    } else {
        T = K.tab[t & 255];
    }
    const int e = t >> 8;
    asm("v_fma_f64 %0, %2, %3, %4\n\t"        // p = r/24 + 1/6
        "v_fma_f64 %0, %2, %0, 0.5\n\t"       // p = r p + 1/2
        "v_mul_f64 %1, %2, %2\n\t"            // r^2
        "v_fma_f64 %0, %1, %0, %2"             // p = r^2 p + r
        : "=&v"(p), "=&v"(r2)
        : "v"(r), "s"(K.t[3]), "v"(K.c5));
    asm("v_fma_f64 %0, %1, %2, %1\n\t"        // T p + T
        "v_ldexp_f64 %0, %0, %3"
        : "=&v"(out)
        : "v"(T), "v"(p), "v"(e));
    return out;
}
// exp(x) for |x| < 5e6 (see fexp_tab) where the caller can vouch for the bound — the fast-clamp instantiations (F), whose
// operands are finite and in range by construction; the general form otherwise and where the table route is not in use
__device__ __forceinline__ double fexp(double x, const MathK& K);
template <bool F = true>
__device__ __forceinline__ double fexp_b(double x, const MathK& K) {
    if (F && K.table) return fexp_tab<true>(x, K);
    return fexp(x, K);
}
// ... and of modest size (fexp_tab SHORT): one instruction fewer
template <bool F = true>
__device__ __forceinline__ double fexp_s(double x, const MathK& K) {
    if (F && K.table) return fexp_tab<true, true>(x, K);
    return fexp(x, K);
}
__device__ __forceinline__ double fexp(double x, const MathK& K) {
    if (K.table) return fexp_tab<false>(x, K);
    double n, r, p, out;
    int t;
    const double c11 = K.c11;
    asm("v_mul_f64 %0, %5, %6\n\t"
        "v_rndne_f64 %0, %0\n\t"
        "v_fma_f64 %1, %0, %7, %5\n\t"
        "v_fma_f64 %1, %0, %8, %1\n\t"
        "v_fma_f64 %2, %1, %9, %10\n\t"
        "v_fma_f64 %2, %1, %2, %11\n\t"
        "v_fma_f64 %2, %1, %2, %12\n\t"
        "v_fma_f64 %2, %1, %2, %13\n\t"
        "v_fma_f64 %2, %1, %2, %14\n\t"
        "v_fma_f64 %2, %1, %2, %15\n\t"
        "v_fma_f64 %2, %1, %2, %16\n\t"
        "v_fma_f64 %2, %1, %2, %17\n\t"
        "v_fma_f64 %2, %1, %2, %18\n\t"
        "v_fma_f64 %2, %1, %2, 1.0\n\t"
        "v_fma_f64 %2, %1, %2, 1.0\n\t"
        "v_cvt_i32_f64 %3, %0\n\t"
        "v_ldexp_f64 %4, %2, %3"
        : "=&v"(n), "=&v"(r), "=&v"(p), "=&v"(t), "=v"(out)
        : "v"(x), "s"(K.e[0]), "s"(K.e[1]), "s"(K.e[2]), "v"(c11), "s"(K.e[3]), "s"(K.e[4]), "s"(K.e[5]), "s"(K.e[6]),
          "s"(K.e[7]), "s"(K.e[8]), "s"(K.e[9]), "s"(K.e[10]), "s"(K.e[11]));
    return out;
}
// log(x) for finite normal x > 0: m in [sqrt(1/2), sqrt(2)), s = f/(2+f), the
// classic 7-term series in s^2 with the hi/lo split of ln2.
// Table route: m in [0.5, 1) and e from frexp; the top 8 mantissa bits pick (c, l) with r = m c - 1, |r| <= 2^-8, and
//   log x = (e - [j < kLogSplit]) ln2 + l + log1p(r),   log1p(r) = r - r^2/2 + ... + r^7/7   (r^8/8 < 2^-67).
// Around x = 1 both neighbouring intervals have c = 1 (or 2) and l = 0 exactly, so the result is log1p(x - 1) there: no
// cancellation against e ln2.  No division: 14 fp64 + 5 integer instructions + one 16-byte LDS read, against 27 + rcp.
__device__ __forceinline__ double flog_tab(double x, const MathK& K) {
    const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const int j = (__double2hiint(m) >> 12) & 255;
    const double2 cl = *reinterpret_cast<const double2*>(K.ltab + 2 * j);
    e -= j < kLogSplit ? 1 : 0;
    const double dk = (double)e;
    double r, q, r2, out;
    asm("v_fma_f64 %0, %4, %5, -1.0\n\t"       // r = m c - 1
        "v_fma_f64 %1, %0, %7, %8\n\t"         // q = r/7 - 1/6
        "v_fma_f64 %1, %0, %1, %9\n\t"         // q = r q + 1/5
        "v_fma_f64 %1, %0, %1, %10\n\t"        // q = r q - 1/4
        "v_fma_f64 %1, %0, %1, %11\n\t"        // q = r q + 1/3
        "v_fma_f64 %1, %0, %1, -0.5\n\t"       // q = r q - 1/2
        "v_mul_f64 %2, %0, %0\n\t"             // r^2
        "v_fma_f64 %1, %2, %1, %0\n\t"         // p = r^2 q + r
        "v_fma_f64 %1, %6, %12, %1\n\t"        // + dk ln2_lo
        "v_fma_f64 %3, %6, %13, %14\n\t"       // dk ln2_hi + l   (dk ln2_hi is exact)
        "v_add_f64 %3, %3, %1"
        : "=&v"(r), "=&v"(q), "=&v"(r2), "=&v"(out)
        : "v"(m), "v"(cl.x), "v"(dk), "v"(K.g7v), "s"(K.g[1]), "s"(K.g[2]), "s"(K.g[3]), "s"(K.g[4]), "s"(K.g[5]), "s"(K.g[6]),
          "v"(cl.y));
    return out;
}
__device__ __forceinline__ double flog(double x, const MathK& K) {
    if (K.logtab) return flog_tab(x, K);
    double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const int lo = m < 0.70710678118654752440 ? 1 : 0;
    m = __builtin_amdgcn_ldexp(m, lo);
    e -= lo;
    const double f = m - 1.0;
    const double s = fdiv(f, 2.0 + f);
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double lg6 = K.lg6, lg7 = K.lg7;
    double z, w, t1, t2, out;
    asm("v_mul_f64 %0, %5, %5\n\t"              // z = s*s
        "v_mul_f64 %1, %0, %0\n\t"              // w = z*z
        "v_fma_f64 %2, %1, %9, %10\n\t"         // t1 = w*Lg6 + Lg4
        "v_fma_f64 %3, %1, %11, %12\n\t"        // t2 = w*Lg7 + Lg5
        "v_fma_f64 %2, %1, %2, %13\n\t"         // t1 = w*t1 + Lg2
        "v_fma_f64 %3, %1, %3, %14\n\t"         // t2 = w*t2 + Lg3
        "v_mul_f64 %2, %1, %2\n\t"              // t1 = w*t1
        "v_fma_f64 %3, %1, %3, %15\n\t"         // t2 = w*t2 + Lg1
        "v_fma_f64 %2, %0, %3, %2\n\t"          // R  = z*t2 + t1
        "v_add_f64 %2, %6, %2\n\t"              // hfsq + R
        "v_mul_f64 %2, %5, %2\n\t"              // s*(hfsq + R)
        "v_fma_f64 %2, %8, %16, %2\n\t"         // + dk*ln2_lo
        "v_add_f64 %2, %6, -%2\n\t"             // hfsq - (...)
        "v_add_f64 %2, %2, -%7\n\t"             // (...) - f
        "v_fma_f64 %4, %8, %17, -%2"              // dk*ln2_hi - (...)
        : "=&v"(z), "=&v"(w), "=&v"(t1), "=&v"(t2), "=v"(out)
        : "v"(s), "v"(hfsq), "v"(f), "v"(dk), "v"(lg6), "s"(K.l[0]), "v"(lg7), "s"(K.l[1]), "s"(K.l[2]), "s"(K.l[3]),
          "s"(K.l[4]), "s"(K.l[5]), "s"(K.l[6]));
    return out;
}
// pow(x, y) for the positive-base uses on this path, as exp(y*log(x)); the
// relative error (|y log x| * 2^-52) is far below the 1e-4 acceptance bar.
__device__ __forceinline__ double powxy(double x, double y, const MathK& K) { return fexp(y * flog(x, K), K); }
__device__ __forceinline__ double sq(double x) { return x * x; }
__device__ __forceinline__ double pow4(double x) { double x2 = x * x; return x2 * x2; }

// cpp:480-490 satvapCpp: 0.61078 exp(a tc / (tc + b)), (a, b) = (17.27, 237.3) over water (tc > 0), (21.875, 265.5) over ice.
// Chosen per lane the pair costs a compare, four 32-bit selects and the moves that put one constant of each pair into a
// VGPR — 9 of the function's ~26 VALU instructions.  The lanes of a wave (three consecutive hours of 21 neighbouring
// cells) are nearly always on the same side of 0 degrees C: then the constants are scalar operands of the multiply and the
// add, and only a wave that straddles the freezing point selects per lane.  Same operands, same operations: same bits.
__device__ __forceinline__ void satvap_nd(double tc, double& num, double& den) {
    const bool water = tc > 0;
    const uint64_t m = __builtin_amdgcn_ballot_w64(water);
    // (opaque results: or the compiler hoists the add out of the branches and selects its constant in VGPRs after all)
    if (m == __builtin_amdgcn_ballot_w64(true)) { num = 17.27 * tc; den = tc + 237.3; asm("" : "+v"(num), "+v"(den)); }
    else if (m == 0) { num = 21.875 * tc; den = tc + 265.5; asm("" : "+v"(num), "+v"(den)); }
    else { num = (water ? 17.27 : 21.875) * tc; den = tc + (water ? 237.3 : 265.5); }
}
__device__ __forceinline__ double satvap_arg(double tc) {
    double num, den;
    satvap_nd(tc, num, den);
    return fdiv(num, den);
}
// 1/a and 1/b out of ONE reciprocal: r = 1/(a b), 1/a = r b, 1/b = r a.  v_rcp_f64 runs at quarter rate (four issue slots) and
// its cubic refinement takes three more; a pair costs one reciprocal and three multiplications (10 slots) instead of two
// reciprocals (14).  For finite, normal a, b whose product stays in range — the fast-clamp instantiations; ~2 ulp.
__device__ __forceinline__ void frcp2(double a, double b, double& ra, double& rb) {
    const double r = frcp(a * b);
    ra = r * b;
    rb = r * a;
}
// ... to 46 bits each (frcp_m): the pair of the ground wetness factor and the canopy temperature's denominator
__device__ __forceinline__ void frcp2_m(double a, double b, double& ra, double& rb) {
    const double r = frcp_m(a * b);
    ra = r * b;
    rb = r * a;
}
__device__ __forceinline__ double satvap(double tc, const MathK& K) { return 0.61078 * fexp(satvap_arg(tc), K); }
// On REGULAR steps (kStepIrregular clear) air and dew-point temperature lie in (-150, 150) and every temperature of the path is a
// Penman-Monteith result in [tdew, tc + 80] (pm_temperature), so |a t / (t + b)| < 60: the bounded exp applies.
// F: 0.61078 exp(q) = exp(q + ln 0.61078), the addition folded into the quotient's last multiplication (one instruction
// fewer; the sum's rounding is of the size of the quotient's own), through the one-fma reduction (fexp_s).
constexpr double kLnSvp0 = -0x1.f8d9d41e4b1ffp-2;     // ln 0.61078
template <bool F>
__device__ __forceinline__ double satvap_f(double t, const MathK& K) {
    if (F && K.table) {
        double num, den;
        satvap_nd(t, num, den);
        return fexp_s<true>(fma(num, frcp(den), kLnSvp0), K);
    }
    return 0.61078 * fexp_b<F>(satvap_arg(t), K);
}
// ... from the quotient's parts, where its reciprocal was shared with another division (frcp2)
template <bool F>
__device__ __forceinline__ double satvap_rd(double num, double rden, const MathK& K) {
    if (F && K.table) return fexp_s<true>(fma(num, rden, kLnSvp0), K);
    return 0.61078 * fexp_b<F>(num * rden, K);
}
// cpp:24-26 with the 0.97*sb factor every caller applies
__device__ __forceinline__ double lw_emit(double tc) { return 0.97 * kSb * pow4(tc + 273.15); }
// cpp:1227-1232
__device__ __forceinline__ double latent(double tc) {
    return tc >= 0 ? 45068.7 - 42.8428 * tc : 51078.69 - 4.338 * tc - 0.06367 * tc * tc;
}

struct SolPos { double zend, zenr, azid; };

// cpp:28-37 juldayCpp
__host__ __device__ inline int julday(int year, int month, int day) {
    double dd = day + 0.5;
    int madj = month + (month < 3) * 12;
    int yadj = year + (month < 3) * -1;
    double j = trunc(365.25 * (yadj + 4716)) + trunc(30.6001 * (madj + 1)) + dd - 1524.5;
    int b = (int)(2 - trunc((double)(yadj / 100)) + trunc(trunc((double)(yadj / 100)) / 4));
    return (int)(j + (j > 2299160) * b);
}

// cpp:48-83 solpositionCpp, split so that the date-only part (dec, eot) can be
// tabulated per time step and the site part (lat, lon) applied per cell.
struct SolDate { double sindec, cosdec, eot; };
__device__ inline SolDate sol_date(int year, int month, int day) {
    int jd = julday(year, month, day);
    double m = 6.24004077 + 0.01720197 * (jd - 2451545.0);                       // cpp:42
    SolDate s;
    s.eot = -7.659 * sin(m) + 9.863 * sin(2 * m + 3.5932);                          // cpp:43
    double dec = (kPi * 23.5 / 180) * cos(2 * kPi * ((jd - 159.5) / 365.25));      // cpp:55
    s.sindec = sin(dec);
    s.cosdec = cos(dec);
    return s;
}
__device__ inline SolPos sol_site(const SolDate& sd, double lt, double sinlat, double coslat, double lon) {
    double st = lt + (4.0 * lon + sd.eot) / 60.0;                                   // cpp:44
    double tt = 0.261799 * (st - 12);                                               // cpp:54
    double ctt = cos(tt), stt = sin(tt);
    double coh = sd.sindec * sinlat + sd.cosdec * coslat * ctt;                     // cpp:56
    double z = acos(coh) * (180 / kPi);                                             // cpp:57
    double sh = coh;                                                                // cpp:59
    double hh = atan(sh / sqrt(1 - sh * sh));                                       // cpp:60
    double sazi = sd.cosdec * stt / cos(hh);                                        // cpp:61
    double num = sinlat * sd.cosdec * ctt - coslat * sd.sindec;
    double cazi = num / sqrt(sq(sd.cosdec * stt) + sq(num));                        // cpp:62-64
    double sqt = 1 - sazi * sazi;
    if (sqt < 0) sqt = 0;
    double azi = 180 + (180 * atan(sazi / sqrt(sqt))) / kPi;                        // cpp:67
    if (cazi < 0) azi = (sazi < 0) ? 180 - azi : 540 - azi;                         // cpp:68-75
    SolPos o;
    o.zend = z;
    o.zenr = z * kToRad;
    o.azid = azi;
    return o;
}

// half-away-from-zero round then C remainder (cpp:2166-2167); negative
// directions (out of bounds in the reference) are wrapped into range.
__device__ __forceinline__ int dir_index(double v, double step, int n) {
    int r = ((int)round(v / step)) % n;
    return r < 0 ? r + n : r;
}

// Register-resident per-timestep values; TF_ order.  Filled by derive_time().
struct TimeVals {
    double v[TF_COUNT];
};

// Fills the derived TF_ fields of `t` from its raw fields and the solar position.
__device__ inline void derive_time(TimeVals& t, const SolPos& sp, int windex) {
    MathK K;
    K.set();
    const double tc = t.v[TF_TC];
    const double zenr = sp.zenr;
    t.v[TF_ZEND] = sp.zend;
    t.v[TF_CZ] = cos(zenr);                       // cpp:1092 (unclamped), cpp:93/96
    t.v[TF_SZ] = sin(zenr);
    double azr = sp.azid * kToRad;
    t.v[TF_CAZ] = cos(azr);
    t.v[TF_SAZ] = sin(azr);
    t.v[TF_TANSA] = tan((kPi / 2.0) - zenr);      // cpp:2222-2223
    double zc = zenr > (kPi / 2.0) ? (kPi / 2.0) : zenr;    // cpp:106
    double cc = cos(zc), tn = tan(zc);
    t.v[TF_COSC] = cc;
    t.v[TF_TANC] = tn;
    t.v[TF_TAN2C] = tn * tn;
    t.v[TF_INV2COSC] = 1.0 / (2.0 * cc);
    // the reference passes the zenith in DEGREES to cankCpp at cpp:1425
    double zb = sp.zend > (kPi / 2.0) ? (kPi / 2.0) : sp.zend;
    double cb = cos(zb), tb = tan(zb);
    t.v[TF_TANB] = tb;
    t.v[TF_TAN2B] = tb * tb;
    t.v[TF_INV2COSB] = 1.0 / (2.0 * cb);
    t.v[TF_DE] = satvap(tc + 0.5, K) - satvap(tc - 0.5, K);                       // cpp:1223
    double tk = tc + 273.15;
    t.v[TF_GHRRAD] = (4 * 0.97 * kSb * (tk * tk * tk)) / 29.3;               // cpp:1224
    t.v[TF_REM] = lw_emit(tc);                                              // cpp:1225, 1167
    const double la = latent(tc);
    t.v[TF_LAPK] = la / t.v[TF_PK];                                         // m = la*(gV/pk), cpp:1233
    t.v[TF_MUPM] = la * (43.0 / t.v[TF_PK]);                                // cpp:1245
    t.v[TF_INVMUPM] = 1.0 / t.v[TF_MUPM];
    // G = Gp*(dtr/dtrp)*(k*muGp)/(kp*DD) = GFAC*dtr*k/DD, cpp:1282-1289
    t.v[TF_GFAC] = t.v[TF_GP] * t.v[TF_MUGP] / (t.v[TF_DTRP] * t.v[TF_KP]);
    t.v[TF_WFAC] = 0.018 / (8.31 * tk);                                     // cpp:1268
    double rbeam = (t.v[TF_RSW] - t.v[TF_RDIF]) / t.v[TF_CZ];               // cpp:1122
    if (rbeam > 1352.0) rbeam = 1352.0;
    t.v[TF_RBEAM] = rbeam;
    t.v[TF_RB] = rbeam * t.v[TF_CZ];                                        // cpp:1124
    int sindex = dir_index(sp.azid, 15.0, 24);
    int ksat = (sp.zend > (kPi / 2.0)) ? 1 : 0;
    // kStepIrregular: a value of this row is not finite, or one of the signs the fast clamps rely on does not hold
    // (Penman-Monteith's denominator 29.3*(g + ghr) + la/pk*g*De stays positive for ghr, De, la/pk > 0)
    bool ok = t.v[TF_DE] > 0.0 && t.v[TF_GHRRAD] > 0.0 && t.v[TF_LAPK] > 0.0 && t.v[TF_PK] > 0.0;
    ok = ok && tc > -150.0 && tc < 150.0 && t.v[TF_TDEW] > -150.0 && t.v[TF_TDEW] < 150.0;     // satvap_f's bounded exp
    for (int f = 0; f < TF_IDX; ++f) ok = ok && isfinite(t.v[f]);
    t.v[TF_IDX] = (double)(sindex | (windex << 5) | (ksat << 8) | (ok ? 0 : kStepIrregular));
}

// Array forcing (runmicro2Cpp geometry): the same TF_ values per CELL-step, from the date part of
// the solar position tabulated per time step (sin/cos of the declination and of the hour-angle part
// A = 0.261799*(hour + eot/60 - 12)) and per-cell constants.  cpp:48-83 is followed algebraically
// instead of through its inverse trig calls: with coh = cos(zenith),
//   cos(zenr) = coh, sin(zenr) = sqrt(1-coh^2), tan(pi/2 - zenr) = coh/sin(zenr),
//   cos(hh) = sin(zenr)  (hh = atan(sh/sqrt(1-sh^2)) = asin(coh)),
//   sin(azimuth) = -sazi, cos(azimuth) = -+sqrt(1-sazi^2) by the sign of cazi (cpp:65-75),
// and the 15-degree horizon sector round(azid/15) % 24 is found by comparing against tangents.
// Differences to the literal evaluation are rounding-level (1e-16 relative).
struct DateRow { double sindec, cosdec, cosA, sinA; };
__device__ __forceinline__ void derive_time_af(TimeVals& t, const DateRow& dr, double sinlat, double coslat,
                                               double cosB, double sinB, int windex, const MathK& K) {
    const double ctt = dr.cosA * cosB - dr.sinA * sinB;
    const double stt = dr.sinA * cosB + dr.cosA * sinB;
    const double coh = dr.sindec * sinlat + dr.cosdec * coslat * ctt;          // cpp:56
    double s2 = 1.0 - coh * coh;
    if (s2 < 0.0) s2 = 0.0;
    const double sz = fsqrt(s2 > 1e-300 ? s2 : 1e-300);
    t.v[TF_CZ] = coh;
    t.v[TF_SZ] = sz;
    t.v[TF_TANSA] = fdiv(coh, sz);
    // zenith in degrees is only compared with 90 (cpp:88) and with pi/2 (the degrees call, cpp:1425)
    const bool up = coh >= 0.0;                    // zenith <= 90 degrees
    const bool nearzen = coh > 0.99962422;         // zenith (deg) may be below pi/2 = 1.5708
    double zend = up ? 45.0 : 135.0;
    if (nearzen) zend = acos(coh) * (180 / kPi);
    t.v[TF_ZEND] = zend;
    // canopy extinction operands, radians call: zenr clamped to pi/2 (cpp:106)
    const double cc = up ? coh : 6.123233995736766e-17;                        // cos(pi/2) in fp64
    const double tn = up ? fdiv(sz, coh) : 1.633123935319537e16;               // tan(pi/2) in fp64
    t.v[TF_COSC] = cc;
    t.v[TF_TANC] = tn;
    t.v[TF_TAN2C] = tn * tn;
    t.v[TF_INV2COSC] = 0.5 * frcp(cc);
    const int ksat = zend > (kPi / 2.0) ? 1 : 0;       // TANB, TAN2B, INV2COSB (pass 2, !ksat only): derive_time_af_pass2
    // azimuth, cpp:59-75
    double sazi = fdiv(dr.cosdec * stt, sz);
    const double num = sinlat * dr.cosdec * ctt - coslat * dr.sindec;         // sign of cazi
    double sqt = 1.0 - sazi * sazi;
    if (sqt < 0.0) sqt = 0.0;
    if (sazi > 1.0) sazi = 1.0;
    if (sazi < -1.0) sazi = -1.0;
    const double rq = fsqrt(sqt > 1e-300 ? sqt : 1e-300);
    const double saz = -sazi;
    const double caz = num < 0.0 ? rq : -rq;
    t.v[TF_SAZ] = saz;
    t.v[TF_CAZ] = caz;
    // sindex = round(azid/15) % 24 (cpp:2501): rotate by +7.5 deg, quadrant, then tangent tests
    const double xr = caz * 0.99144486137381038 - saz * 0.13052619222005157;
    const double yr = saz * 0.99144486137381038 + caz * 0.13052619222005157;
    int q;
    double u, v;
    if (yr >= 0.0) {
        if (xr > 0.0) { q = 0; u = xr; v = yr; } else { q = 1; u = yr; v = -xr; }
    } else {
        if (xr < 0.0) { q = 2; u = -xr; v = -yr; } else { q = 3; u = -yr; v = xr; }
    }
    int n = (v >= u * 0.26794919243112270) + (v >= u * 0.57735026918962573) + (v >= u) +
            (v >= u * 1.7320508075688772) + (v >= u * 3.7320508075688776);
    const int sindex = (6 * q + n) % 24;
    t.v[TF_IDX] = (double)(sindex | (windex << 5) | (ksat << 8));
    // Penman-Monteith operands (cpp:1220-1247) and beam normalisation (cpp:1122-1124)
    const double tc = t.v[TF_TC];
    t.v[TF_DE] = satvap(tc + 0.5, K) - satvap(tc - 0.5, K);
    const double tk = tc + 273.15;
    t.v[TF_GHRRAD] = (4 * 0.97 * kSb * (tk * tk * tk)) * (1.0 / 29.3);
    t.v[TF_REM] = lw_emit(tc);
    const double ipk = frcp(t.v[TF_PK]);
    t.v[TF_LAPK] = latent(tc) * ipk;
    t.v[TF_WFAC] = fdiv(0.018, 8.31 * tk);
    double rbeam = fdiv(t.v[TF_RSW] - t.v[TF_RDIF], coh);
    if (rbeam > 1352.0) rbeam = 1352.0;
    t.v[TF_RBEAM] = rbeam;
    t.v[TF_RB] = rbeam * coh;
}
// ---- coarse array forcing (mcf_grid_inputs.array_forcing == 2) ------------------------------------------------
// Bilinear tap into a coarse [crows, ccols] field: the four neighbours and weights of one raster cell, from its
// position in coarse-grid units (clamped by the host, so that r1 / c1 fall back onto r0 / c0 at the far edges).
struct CoarseTap {
    uint32_t o00, o01, o10, o11;     // BYTE offsets of the four neighbours in a field: the field's base is wave-uniform, so a
                                     // tap is global_load v, v_off, s[base:base+1] — no 64-bit vector address arithmetic
    double wx, wy;
    // hour: the lane's hour of the day — `p` below is then the field at the DAY's first step (uniform over the workgroup)
    __device__ __forceinline__ CoarseTap(double rowpos, double colpos, int crows, int ccols, int hour = 0) {
        const double fr = floor(rowpos), fc = floor(colpos);
        const int r0 = (int)fr, c0 = (int)fc;
        const int r1 = r0 + 1 < crows ? r0 + 1 : r0, c1 = c0 + 1 < ccols ? c0 + 1 : c0;
        wy = rowpos - fr;
        wx = colpos - fc;
        const uint32_t h = (uint32_t)hour * (uint32_t)(crows * ccols);       // (24 x cells x 8 B < 2^32: checked by the host)
        o00 = 8u * (h + (uint32_t)(r0 + crows * c0)); o01 = 8u * (h + (uint32_t)(r0 + crows * c1));
        o10 = 8u * (h + (uint32_t)(r1 + crows * c0)); o11 = 8u * (h + (uint32_t)(r1 + crows * c1));
    }
    // p: the field at one time step
    __device__ __forceinline__ double operator()(const double* __restrict__ p) const {
        uint32_t a = o00, b = o01, c = o10, d = o11;
        asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // keeps the zero-extensions in the loads' own block (see k_solve `put`)
        const char* q = reinterpret_cast<const char*>(p);
        const double v00 = *reinterpret_cast<const double*>(q + a), v01 = *reinterpret_cast<const double*>(q + b),
                     v10 = *reinterpret_cast<const double*>(q + c), v11 = *reinterpret_cast<const double*>(q + d);
        return mix(mix(v00, v01, wx), mix(v10, v11, wx), wy);
    }
    // (1 - w) a + w b, with the contraction written out: the LDS-staged form of the taps (k_solve, CLDS) interpolates along the
    // columns once per tile-day and along the rows per lane, and must give the bits of this per-lane form
    static __device__ __forceinline__ double mix(double a, double b, double w) { return fma(w, b, (1.0 - w) * a); }
};
// `.satvap` and `.dewpoint` of the R side (R/internal.R:501-521), which `.runmodel2Cpp` applies to the resampled
// temperature and humidity (R/internal.R:1230-1232); NOT satvapCpp / dewpointCpp (other ice threshold, other constants).
__device__ __forceinline__ double satvap_r(double tc, const MathK& K) {
    const double a = tc < 0.0 ? 21.875 : 17.27, b = tc < 0.0 ? 265.5 : 237.3;
    return 0.61078 * fexp(fdiv(a * tc, tc + b), K);
}
// `.lapserate` (R/internal.R:545-550): moist adiabatic lapse rate, K per m
__device__ __forceinline__ double lapserate_r(double tc, double ea, double pk) {
    const double rv = fdiv(0.622 * ea, pk - ea), tk = tc + 273.15;
    return fdiv(9.8076 * (1.0 + fdiv(2501000.0 * rv, 287.0 * tk)),
                1003.5 + fdiv(0.622 * 2501000.0 * 2501000.0 * rv, 287.0 * (tk * tk)));
}
__device__ __forceinline__ double dewpoint_r(double ea, double tc, const MathK& K) {
    if (!(ea > 0.0)) return -273.15;                       // log(0) = -Inf in R: 1/Inf - 273.15
    const double lw = flog(ea * (1.0 / 0.6112), K), li = flog(ea * (1.0 / 0.61078), K);
    const double L = 2.501e6 - 2340.0 * tc;
    const double tdew = frcp(1.0 / 273.15 - fdiv(461.5, L) * lw) - 273.15;
    const double tfrost = frcp(1.0 / 273.15 - (461.5 / 2.834e6) * li) - 273.15;
    return tdew < 0.0 ? tfrost : tdew;
}

// Second half, evaluated in front of pass 2.  A lane carries every time value pass 2 reads in registers across the day's
// barrier (no table in LDS to re-read them from), so what is cheap to make again from a carried value is made again here
// instead of carried: the long-wave operands from tc (opaque to the compiler from here on, or it would carry pass 1's
// copies) and mu from la/pk.  (The degrees-call extinction operands of the rare lane whose zenith angle in DEGREES is
// below pi/2 are made inside pass 2, from the carried zenith.)
__device__ __forceinline__ void derive_time_af_pass2(TimeVals& t) {
    double tc = t.v[TF_TC];
    asm volatile("" : "+v"(tc));
    t.v[TF_TC] = tc;
    const double tk = tc + 273.15;
    t.v[TF_GHRRAD] = (4 * 0.97 * kSb * (tk * tk * tk)) * (1.0 / 29.3);       // cpp:1224
    t.v[TF_REM] = lw_emit(tc);                                                // cpp:1225
    const double mu = 43.0 * t.v[TF_LAPK];                                    // la*(43/pk), cpp:1245
    t.v[TF_MUPM] = mu;
    t.v[TF_INVMUPM] = frcp(mu);
}

// ---- accessors ----------------------------------------------------------------
// Cell constants in LDS, laid out [field][cells_per_block]; `dirs` holds the 24
// horizon + 8 wind-shelter values [dir][cells_per_block].
// One pointer per lane: the tile's image is [CF_COUNT cell fields, then 24 horizon + 8 wind-shelter rows][cells_per_block].
template <int CPB>
struct CellLds {
    const double* p;   // image + the lane's cell
    __device__ __forceinline__ double operator()(int field) const { return p[field * CPB]; }
    __device__ __forceinline__ double hor(int s) const { return p[(CF_COUNT + s) * CPB]; }
    __device__ __forceinline__ double wsa(int w) const { return p[(CF_COUNT + 24 + w) * CPB]; }
};
// A day's time table in LDS, laid out [field][24].
struct TimeLds {
    static constexpr bool in_registers = false;
    const double* row;  // + hour
    __device__ __forceinline__ double operator()(int field) const { return row[field * 24]; }
};
struct TimeReg {
    static constexpr bool in_registers = true;      // array forcing: the passes are short of registers (see pass2, section B)
    const TimeVals* t;
    __device__ __forceinline__ double operator()(int field) const { return t->v[field]; }
};

// Values a lane carries from pass 1 to pass 2 of the same cell-hour.
struct Carry {
    double soilm, num0, rden, radCsw, Rddown, Rbdown, X, uf;
};
struct Pass1Out {
    double Tg0, absRnet;   // to the day reduction
    double uz, Rdup;       // outputs only
};

// Pins values in registers at this point of the program: the compiler has to have
// finished the LDS loads that produce them.  Loading a section's operands in one batch
// and pinning them exposes ONE LDS latency per section instead of one per operand (hipcc
// otherwise places each ds_read right in front of its first use and waits on it at once).
__device__ __forceinline__ void pin1(double& a) { asm volatile("" : "+v"(a)); }
template <class... A>
__device__ __forceinline__ void pin(A&... a) {
    (pin1(a), ...);
}

// ---- clamps ---------------------------------------------------------------------------------------------------------
// The reference clamps with `if (x > hi) x = hi;`, which leaves a NaN x alone; v_min_f64 / v_max_f64 return the OTHER
// operand for a NaN.  On a 64-bit value the compare-and-select form costs v_cmp + 2 v_cndmask (+ moves for a literal
// bound), the min / max form one instruction — ~60 clamps per cell-step.  They are therefore written through cap / flr:
//   F = false  the reference's form, always right;
//   F = true   one v_min_f64 / v_max_f64 — equal to the reference's form whenever x is not NaN.
// k_solve runs the F = true instantiation of pass 1 / pass 2 only for waves whose lanes are all REGULAR (FL_REGULAR
// cells on steps without kStepIrregular): every cell constant and forcing value read is finite and in range, so the
// operands of most clamps are finite by construction (DESIGN.md lists the argument per site).  The few clamp operands
// that can still become NaN from finite inputs (a vanishing two-stream denominator, a non-positive soil diffusivity,
// coinciding Lagrangian resistances) are WATCHED: Canary::watch folds them into a value that is NaN iff any of them
// was NaN or infinite, and a wave with a tripped canary recomputes the pass with F = false.  Building with
// -DMCF_CANARY_ALL=1 watches every clamp operand instead (an audit build: mcf_plan_dispatch_stats then counts how often
// any clamp of a fast launch met a NaN).
#ifndef MCF_CANARY_ALL
#define MCF_CANARY_ALL 0
#endif
struct Canary {
    double c = 0.0;
    __device__ __forceinline__ void watch(double x) { c = fma(0.0, x, c); }     // 0*x: 0 for finite x, NaN otherwise
    __device__ __forceinline__ void trip() { c = __builtin_nan(""); }
    __device__ __forceinline__ bool tripped() const { return c != c; }
};
// one v_min_f64 / v_max_f64; a bound known at compile time is taken from the scalar file (two s_mov_b32 on the scalar
// unit) or as an inline constant instead of being copied into a VGPR pair
__device__ __forceinline__ void vmin64(double& x, double hi) {
    if (__builtin_constant_p(hi)) {
        if (hi == 1.0) asm("v_min_f64 %0, %0, 1.0" : "+v"(x));
        else asm("v_min_f64 %0, %0, %1" : "+v"(x) : "s"(hi));
    } else asm("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(hi));
}
__device__ __forceinline__ void vmax64(double& x, double lo) {
    if (__builtin_constant_p(lo)) {
        if (lo == 0.0) asm("v_max_f64 %0, %0, 0" : "+v"(x));
        else asm("v_max_f64 %0, %0, %1" : "+v"(x) : "s"(lo));
    } else asm("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(lo));
}
template <bool F>
__device__ __forceinline__ void cap(double& x, double hi, Canary& cn) {          // if (x > hi) x = hi;
    if (F) {
#if MCF_CANARY_ALL
        cn.watch(x);
#endif
        vmin64(x, hi);
    } else if (x > hi) x = hi;
}
template <bool F>
__device__ __forceinline__ void flr(double& x, double lo, Canary& cn) {          // if (x < lo) x = lo;
    if (F) {
#if MCF_CANARY_ALL
        cn.watch(x);
#endif
        vmax64(x, lo);
    } else if (x < lo) x = lo;
}

// Penman-Monteith surface temperature, cpp:1220-1247, from pre-assembled parts.  den > 0 on regular lanes (gHa >= 1e-4,
// ghr, De, lapk > 0), so dT is finite there.
// (`rden`: the reciprocal of the denominator — pass 1 and pass 2 solve for the ground with the SAME denominator, cpp:1272 /
// 1293, so the lane carries 1 / den across the day's barrier instead of den and pass 2 divides by a multiplication)
// F: `dTmx` is min(dTmx, 80) already (solve_tile): the two caps of cpp:1237-1238 are one v_min_f64.  (A NaN dTmx is ignored by
// the reference's comparison and by fmin alike.)
// M80: that merge has been done (vector forcing, where dTmx is one value per launch; array forcing's is per CELL — a second
// per-lane double alive through the whole day loop put scratch into kernels that sit at their register limit — and keeps two caps)
template <bool F, bool M80 = F>
__device__ __forceinline__ double pm_temperature_r(double num, double rden, double dTmx, double tc, double tdew, Canary& cn) {
    double dT = num * rden;
    cap<F>(dT, dTmx, cn);
    if (!M80) cap<F>(dT, 80.0, cn);
    double Ts = dT + tc;
    flr<F>(Ts, tdew, cn);
    return Ts;
}
template <bool F, bool M80 = F>
__device__ __forceinline__ double pm_temperature(double num, double den, double dTmx, double tc, double tdew, Canary& cn) {
    double dT = fdiv_m(num, den);
    cap<F>(dT, dTmx, cn);
    if (!M80) cap<F>(dT, 80.0, cn);
    double Ts = dT + tc;
    flr<F>(Ts, tdew, cn);
    return Ts;
}

// ---- per cell-day soil state --------------------------------------------------------------------------------------------
// cpp:1021-1032 (soildCpp), cpp:1264-1266 (matric potential of soiltempG0), cpp:1249-1260 (soilcondCpp),
// cpp:451-455 + 382-389 (the theta-only part of stomcondCpp).  The same inline functions serve the hour lanes (days
// whose soil moisture varies within the day, array forcing) and the tile's producer wave, so both give the same bits.
enum SoilField : int { SD_SOILM = 0, SD_MATRIC, SD_KSOIL, SD_RDD, SD_DD, SD_GS2, SD_COUNT };
template <bool F>
__device__ __forceinline__ double soil_spread(double soilmp, double smin, double invrge, double eta, double rge, Canary& cn) {
    double theta = (soilmp - smin) * invrge;
    cap<F>(theta, 0.9999, cn);
    flr<F>(theta, 0.0001, cn);
    double sm = theta / (theta + (1.0 - theta) * eta);
    return sm * rge + smin;
}
__device__ __forceinline__ double soil_ksoil_arg(double soilm, double c3) { return -pow4(c3 * soilm); }
__device__ __forceinline__ double soil_ksoil(double soilm, double rho, double c1, double c1mc4, double e) {
    double c2 = 1.06 * rho * soilm;
    return c1 + c2 * soilm - c1mc4 * e;
}
__device__ __forceinline__ void soil_damping(double soilm, double ksoil, double rho, double csa, double& DD, double& rdd) {
    double cs = csa + 4180.0 * soilm;
    double ph = (rho * (1.0 - soilm) + soilm) * 1000.0;
    double kap = fdiv(ksoil, cs * ph);
    DD = fsqrt(kap * (2.0 / kOmdy));
    rdd = frcp(DD);
}
template <bool F>
__device__ __forceinline__ double stom_se(double soilm, double rat, double ratc, double invsmax, Canary& cn) {
    double thetan = rat * soilm + ratc;
    double Se = thetan * invsmax;
    cap<F>(Se, 1.0, cn);
    return Se;
}
template <bool F>
__device__ __forceinline__ double stom_psiw(double pw, double abspsie, double psiw0, Canary& cn) {   // pw = Se^-b
    double psiw = -abspsie * pw * 0.01;
    flr<F>(psiw, psiw0, cn);
    return psiw;
}
__device__ __forceinline__ double stom_gs2(double e, double mudeninv, double gsmax) {               // e = exp(-kk*psiw)
    double mu = 1.0 - (e - 1.0) * mudeninv;
    return mu * gsmax;
}
// One tile's soil state for one day, by ONE wave: lanes 0 .. CPB-1 take the soil side of their cell (spread, matric
// potential, conductivity, damping depth), lanes CPB .. 2*CPB-1 the stomatal side, so that the two pow() (same exponent
// -b) and the two exp() each run as ONE wave instruction stream.  cellc: the tile's constants in LDS [field][CPB];
// dst: [SD_COUNT][CPB].
template <int CPB, bool F>
__device__ __forceinline__ void soil_day_produce(const double* cellc, double soilmp, double* dst, int lane, const MathK& K) {
    static_assert(2 * CPB <= 64, "two roles per cell must fit a wave");
    const int role = lane / CPB, cell = lane - role * CPB;
    if (role > 1) return;
    auto C = [&](int f) { return cellc[f * CPB + cell]; };
    Canary cn;
    const double soilm = soil_spread<F>(soilmp, C(CF_SMIN), C(CF_INVRGE), C(CF_ETA), C(CF_RGE), cn);
    const double abspsie = C(CF_ABSPSIE), invsmax = C(CF_INVSMAX);
    const double se = stom_se<F>(soilm, C(CF_RAT), C(CF_RATC), invsmax, cn);
    const double pw = powxy(role ? se : soilm * invsmax, -C(CF_SOILB), K);
    const double psiw = stom_psiw<F>(pw, abspsie, C(CF_PSIW0), cn);
    const double e = fexp(role ? -C(CF_KK) * psiw : soil_ksoil_arg(soilm, C(CF_C3)), K);
    if (role == 0) {
        const double rho = C(CF_RHO);
        const double ksoil = soil_ksoil(soilm, rho, C(CF_C1), C(CF_C1MC4), e);
        double DD, rdd;
        soil_damping(soilm, ksoil, rho, C(CF_CSA), DD, rdd);
        dst[SD_SOILM * CPB + cell] = soilm;
        dst[SD_MATRIC * CPB + cell] = -abspsie * pw;
        dst[SD_KSOIL * CPB + cell] = ksoil;
        dst[SD_RDD * CPB + cell] = rdd;
        dst[SD_DD * CPB + cell] = DD;
    } else {
        dst[SD_GS2 * CPB + cell] = stom_gs2(e, C(CF_MUDENINV), C(CF_GSMAX));
    }
}
// a lane's view of its cell's entry; SS = false: never read (array forcing)
template <int CPB>
struct SoilLds {
    const double* p;   // + cell
    __device__ __forceinline__ double operator()(int f) const { return p[f * CPB]; }
};

// ---------------------------------------------------------------------------------
// PASS 1 (cpp:2214-2262): terrain-adjusted solar index, soil moisture spread,
// two-stream radiation, wind, G = 0 soil surface temperature.
// ---------------------------------------------------------------------------------
template <bool F, bool SS, class CL, class TM, class SL>
__device__ __forceinline__ void pass1(const CL& C, const TM& T, const SL& S, const Globals& g, int flags, double dTmx,
                                      Carry& cy, Pass1Out& o, const MathK& K, Canary& cn) {
    // ---- section A operands: soil moisture spread + branch selectors
    double t_idx = T(TF_IDX), rsw = T(TF_RSW), rdif = T(TF_RDIF);
    pin(t_idx, rsw, rdif);
    const int idx = (int)t_idx;
    constexpr bool soil_shared = SS;     // the host launches the SS instantiation only for days flagged kSoilDaily
    // --- distributed soil moisture, cpp:1021-1032, and the matric potential of soiltempG0, cpp:1264-1266 ---------
    double soilm, matric;
    if (soil_shared) {
        soilm = S(SD_SOILM);
        matric = S(SD_MATRIC);
    } else {
        double t_soilmp = T(TF_SOILMP), c_smin = C(CF_SMIN), c_invrge = C(CF_INVRGE), c_eta = C(CF_ETA), c_rge = C(CF_RGE);
        pin(t_soilmp, c_smin, c_invrge, c_eta, c_rge);
        soilm = soil_spread<F>(t_soilmp, c_smin, c_invrge, c_eta, c_rge, cn);
        matric = -C(CF_ABSPSIE) * powxy(soilm * C(CF_INVSMAX), -C(CF_SOILB), K);
    }
    cy.soilm = soilm;
    // --- short wave, cpp:1086-1163 --------------------------------------------------
    double radGsw = 0.0, radCsw = 0.0, Rbdown = 0.0, Rddown = 0.0, Rdup = 0.0, X = 0.0;
    if (rsw > 0.0) {
        // ---- section B operands: solar index, horizon, canopy extinction
        double cz = T(TF_CZ), t_sz = T(TF_SZ), t_caz = T(TF_CAZ), t_saz = T(TF_SAZ), t_tansa = T(TF_TANSA),
               t_zend = T(TF_ZEND);
        double c_cs = C(CF_CS), c_ssca = C(CF_SSCA), c_sssa = C(CF_SSSA), c_hor = C.hor(idx & 31),
               svfa = C(CF_SVFA), gref = C(CF_GREF);
        pin(cz, t_sz, t_caz, t_saz, t_tansa, t_zend, c_cs, c_ssca, c_sssa, c_hor, svfa, gref);
        // solar index, cpp:85-102 + horizon shading cpp:2219-2223
        double si = cz * c_cs + t_sz * (c_ssca * t_caz + c_sssa * t_saz);
        if (!g.shadowmask && t_zend > 90.0) si = 0.0;
        flr<F>(si, 0.0, cn);
        if (c_hor > t_tansa) si = 0.0;
        if (flags & FL_PAI) {
            // ---- section C operands: extinction + direct-beam two-stream coefficients
            double t_tan2c = T(TF_TAN2C), t_cosc = T(TF_COSC);
            double c_xx = C(CF_XX), c_kdeninv = C(CF_KDENINV), om = C(CF_OM), gma = C(CF_GMA), agm = C(CF_AGM),
                   u1 = C(CF_U1), u2 = C(CF_U2), c_gma2 = C(CF_GMA2),
                   c_agm2 = C(CF_AGM2), c_jdel = C(CF_JDEL), c_pait = C(CF_PAIT), c_invd2 = C(CF_INVD2), c_gmagref = C(CF_GMAGREF);
            pin(t_tan2c, t_cosc, c_xx, c_kdeninv, om, gma, agm, u1, u2, c_gma2, c_agm2, c_jdel, c_pait, c_invd2, c_gmagref);
            // canopy extinction, cpp:104-132
            double k = fsqrt(c_xx + t_tan2c) * c_kdeninv;
            if (flags & (FL_XONE | FL_XINF | FL_XZERO))
                k = (flags & FL_XONE) ? T(TF_INV2COSC) : (flags & FL_XINF) ? 1.0 : T(TF_TANC);
            cap<F>(k, 6000.0, cn);
            double rsi = frcp(si);
            double kd = k * t_cosc * rsi;
            double Kc = rsi;
            if (si == 0.0) { kd = 1.0; Kc = 600.0; }
            // direct-beam two-stream coefficients, cpp:164-185
            double sig = kd * kd + c_gma2 - c_agm2;
            double ss = 0.5 * (om * kd + c_jdel);           // 0.5*(om + J*del/kd)*kd
            double sstr = om * kd - ss;
            double S2 = fexp(-kd * c_pait, K);
            double isig = frcp(sig);
            double p5 = -ss * (agm - kd) - gma * sstr;
            double p5s = p5 * isig;
            double v1 = ss - (p5 * (agm + kd)) * isig;
            double v2 = ss - gma - p5s * (u1 + kd);
            double gS2v2 = S2 * v2;
            double p8 = sstr * (agm + kd) - gma * ss;
            double p8s = -p8 * isig;  // p8 / (-sig)
            double v3 = (sstr + c_gmagref - p8s * (u2 - kd)) * S2;
            const double dv3 = c_invd2 * v3;
            // ---- section D operands: gap transmissions and fluxes
            double c_logclump = C(CF_LOGCLUMP), c_loggi = C(CF_LOGGI), amx = C(CF_AMX), trdn = C(CF_TRDN),
                   trdu = C(CF_TRDU), c_paiaa = C(CF_PAIAA), c_rddng = C(CF_RDDNG), c_albd = C(CF_ALBD), c_rddnz = C(CF_RDDNZ),
                   c_rdupz = C(CF_RDUPZ);
            // (array forcing holds its time values in registers and sits at its register limit: there the eight factors are read
            // where they are used, a pair at a time, instead of in this batch)
            constexpr bool late_k = TM::in_registers;
            double ka1 = late_k ? 0.0 : C(CF_KA1), ka2 = late_k ? 0.0 : C(CF_KA2), kb1 = late_k ? 0.0 : C(CF_KB1),
                   kb2 = late_k ? 0.0 : C(CF_KB2), kg1 = late_k ? 0.0 : C(CF_KG1), kg2 = late_k ? 0.0 : C(CF_KG2),
                   kz1 = late_k ? 0.0 : C(CF_KZ1), kz2 = late_k ? 0.0 : C(CF_KZ2);
            double Rbeam = T(TF_RBEAM), Rb = T(TF_RB);
            if (late_k) pin(c_logclump, c_loggi, amx, trdn, trdu, c_paiaa, c_rddng, c_albd, c_rddnz, c_rdupz, Rbeam, Rb);
            else pin(c_logclump, c_loggi, amx, trdn, trdu, c_paiaa, c_rddng, c_albd, c_rddnz, c_rdupz, ka1, ka2, kb1, kb2, kg1, kg2, kz1,
                     kz2, Rbeam, Rb);
            // gap transmissions, cpp:1095-1100
            // (F: an exponential is never negative — the floors at 0 cannot bind on a regular lane)
            double trbn = fexp(Kc * c_logclump, K);
            cap<F>(trbn, 0.999, cn);
            if (!F) flr<F>(trbn, 0.0, cn);
            double trb = fexp(Kc * c_loggi, K);
            cap<F>(trb, 0.999, cn);
            if (!F) flr<F>(trb, 0.0, cn);
            if (late_k) { ka1 = C(CF_KA1); ka2 = C(CF_KA2); }
            double albb = (1.0 - trdn * trbn) * (p5s + (v1 * ka1 + gS2v2 * ka2)) + trdn * trbn * gref;      // cpp:1102: p5s + p6 + p7
            if (F) cn.watch(albb);        // 1/sig, 1/D1, 1/D2 products: NaN if a two-stream denominator vanishes
            cap<F>(albb, amx, cn);
            flr<F>(albb, 0.01, cn);
            if (late_k) { kg1 = C(CF_KG1); kg2 = C(CF_KG2); }
            double Rdbdn_g = (1.0 - trbn) * (p8s * (S2 + kg1) + dv3 * kg2);                // cpp:1106: p8s S2 + p9 S1 + p10 e^(h pait)
            if (F) cn.watch(Rdbdn_g);
            cap<F>(Rdbdn_g, amx, cn);
            flr<F>(Rdbdn_g, 0.0, cn);
            double S2a = fexp(-kd * c_paiaa, K);
            if (late_k) { kb1 = C(CF_KB1); kb2 = C(CF_KB2); }
            double Rdbup_z = (1.0 - trdu * trbn) * (p5s * S2a + (v1 * kb1 + gS2v2 * kb2)) + trdu * trbn * gref;
            // (the same p5s, p6, p7 with finite weights: watched through albb)
            cap<F>(Rdbup_z, amx, cn);
            flr<F>(Rdbup_z, 0.0, cn);
            if (late_k) { kz1 = C(CF_KZ1); kz2 = C(CF_KZ2); }
            double Rdbdn_z = (1.0 - trb) * (p8s * (S2a + kz1) + dv3 * kz2);                 // cpp:1117
            // (p8s, p9, p10: watched through Rdbdn_g)
            cap<F>(Rdbdn_z, amx, cn);
            flr<F>(Rdbdn_z, 0.0, cn);
            double trg = trb + (1 - trb) * S2;                                              // cpp:1125
            double Rbc = (trg * si + (1 - trg) * cz) * Rbeam;
            double Rbdn_g = trbn + (1.0 - trbn) * S2;      // F: a convex combination of trbn in [0, 0.999] and S2 in [0, 1] stays in [0, 1]
            if (!F) {
                cap<F>(Rbdn_g, 1.0, cn);
                flr<F>(Rbdn_g, 0.0, cn);
            }
            const double rds = rdif * svfa;
            radGsw = (1.0 - gref) * (c_rddng * rds + Rdbdn_g * Rb + Rbdn_g * Rbeam * si);  // cpp:1131
            double maxg = (1.0 - gref) * (rds + Rbeam * si);
            cap<F>(radGsw, maxg, cn);
            radCsw = (1.0 - c_albd) * rds + (1.0 - albb) * Rbc;                              // cpp:1136
            Rbdown = (trb + (1.0 - trb) * S2a) * Rbeam;
            Rddown = c_rddnz * rds + Rdbdn_z * Rb;
            Rdup = c_rdupz * rds + Rdbup_z * Rb;
            X = Rddown + Rdup + k * cz * Rbdown;                                             // cpp:1142-1143
        } else {
            // bare ground, cpp:1145-1153
            Rbdown = (rsw - rdif) / cz;
            Rddown = rdif * svfa;
            Rdup = gref * (rdif * svfa + (rsw - rdif));
            radGsw = (1.0 - gref) * (svfa * rdif + si * Rbdown);
            radCsw = radGsw;
        }
    }
    cy.radCsw = radCsw;
    cy.Rddown = Rddown;
    cy.Rbdown = Rbdown;
    cy.X = X;
    o.Rdup = Rdup;
    // ---- section E operands: long wave, wind, G = 0 soil surface temperature
    double c_tsv = C(CF_TSV), c_omtrdif = C(CF_OMTRDIF), ws = C.wsa((idx >> 5) & 7), c_ufc = C(CF_UFC),
           c_uzfac = C(CF_UZFAC), c_ghafac = C(CF_GHAFAC);
    double t_rlw = T(TF_RLW), t_rem = T(TF_REM), u2m = T(TF_U2), t_umu = T(TF_UMU), t_wfac = T(TF_WFAC),
           t_lapk = T(TF_LAPK), t_es = T(TF_ES), t_ea = T(TF_EA), t_ghr = T(TF_GHRRAD), t_de = T(TF_DE),
           tc = T(TF_TC), tdew = T(TF_TDEW);
    pin(c_tsv, c_omtrdif, ws, c_ufc, c_uzfac, c_ghafac, t_rlw, t_rem, u2m, t_umu,
        t_wfac, t_lapk, t_es, t_ea, t_ghr, t_de, tc, tdew);
    // --- long wave absorbed by the ground, cpp:1165-1175 -----------------------------
    const double radGlw = 0.97 * (c_tsv * t_rlw + c_omtrdif * t_rem);
    // --- wind, cpp:1189-1218 ------------------------------------------------------------
    if (isnan(ws)) ws = 1.0;
    flr<F>(ws, 0.05, cn);
    double uf = u2m * c_ufc * t_umu * ws;
    flr<F>(uf, 0.001, cn);
    double uz = uf * c_uzfac;
    cap<F>(uz, u2m, cn);
    double gHa = uf * c_ghafac;
    flr<F>(gHa, 0.0001, cn);
    cy.uf = uf;
    o.uz = uz;
    // --- soil surface temperature with G = 0, cpp:1262-1275 ------------------------------
    const double radabs = radGsw + radGlw;
    double surfwet = fexp(matric * t_wfac, K);
    if (!F) cap<F>(surfwet, 1.0, cn);      // F: matric <= 0 (|psi_e| >= 0, a positive power) and wfac > 0: the exponent is <= 0
    const double m = t_lapk * gHa;
    const double num0 = radabs - t_rem - m * (t_es - t_ea) * surfwet;
    const double den = 29.3 * (gHa + t_ghr) + m * t_de;
    cy.num0 = num0;
    const double rden = frcp_m(den);
    cy.rden = rden;
    double Tg0 = pm_temperature_r<F, F && !TM::in_registers>(num0, rden, dTmx, tc, tdew, cn);
    o.Tg0 = Tg0;
    o.absRnet = fabs(radabs - lw_emit(Tg0));
}

// Stomatal operands shared by the three stomcondCpp evaluations of a cell-step.
struct Stom {
    double gsmax, rsmx, inv02rsmx, gs2;
};
// cpp:442-458 stomcondCpp with the soil-water factor `gs2 = mu*gsmax` passed in.
// Rswabs IS NaN for bare ground (its shade factor is 0/0): the two tests on it stay compare-and-branch; gs and gs2 are finite
template <bool F>
__device__ __forceinline__ double stomcond(double Rswabs, const Stom& s, const MathK& K, Canary& cn) {
    if (Rswabs <= 0.0) return 0.0;
    double gs = s.gsmax;          // light-saturated (Rswabs >= Rsmx): 2^0 = 1
    if (Rswabs < s.rsmx) gs = s.gsmax * fexp_s<F>((Rswabs - s.rsmx) * s.inv02rsmx, K);     // 2^-((Rsmx - R)/(0.2 Rsmx)): (-3.5, 0); inv02rsmx = ln2 / (0.2 Rsmx)
    cap<F>(gs, s.gs2, cn);
    return gs;
}

// cpp:1316-1331 mincondCpp: gmin = max(0.0463*(|Hf*Rnet|/leafd)^0.2, 0.05).  The two calls of a
// cell-step share Rnet and leafd, so (|Rnet|/leafd)^0.2 is evaluated once (`a02`) and each call
// supplies |Hf|^0.2 (`hf02`).
template <bool F>
__device__ __forceinline__ double mincond_a02(double Rnet, double invleafd, const MathK& K, Canary& cn) {
    double arg = fabs(Rnet) * invleafd;
    flr<F>(arg, 1e-300, cn);             // pow(0, 0.2) = 0 and tiny values end in the 0.05 floor alike
    return fexp_b<F>(0.2 * flog(arg, K), K);      // F: arg finite in [1e-300, DBL_MAX]: (-139, 142)
}
template <bool F>
__device__ __forceinline__ double mincond_gmin(double hf02, double a02, Canary& cn) {
    double gmin = 0.0463 * (hf02 * a02);
    flr<F>(gmin, 0.05, cn);
    return gmin;
}

// rhcanopy's limit on the near-field term (cpp:1400-1407): |near| <= mxnear, NaN -> 0.  On regular lanes `near` is a product of
// finite factors and mxnear a finite magnitude, so the limit is one v_max_f64 and one v_min_f64 instead of two compares and
// five selects (and the NaN case cannot arise).
template <bool F>
__device__ __forceinline__ void near_field_limit(double& near, double mxnear) {
    if (F) {
        asm("v_max_f64 %0, %0, -%1\n\tv_min_f64 %0, %0, %1" : "+v"(near) : "v"(mxnear));
    } else {
        if (fabs(near) > mxnear) near = near > 0.0 ? mxnear : -mxnear;
        if (isnan(near)) near = 0;
    }
}

struct Pass2Out {
    double Tg, DD, Tz, tleaf, rh, lwdn, lwup;
};

// ---------------------------------------------------------------------------------
// PASS 2 (cpp:2264-2305): ground temperature with the scaled ground heat flux,
// canopy / leaf / air temperature and humidity at reqhgt.
// ---------------------------------------------------------------------------------
// `midway` is called once, by every lane of pass 2, behind the canopy temperature: roughly the last third of the pass (leaf
// temperature and the Lagrangian profile of a below-canopy cell) is still to come and the register file is past its
// fullest — where the array-forcing kernel starts loading the next day's forcing (solve_tile).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <bool F, bool SS, class CL, class TM, class SL, class HK = NoHook>
__device__ __forceinline__ void pass2(const CL& C, const TM& T, const SL& S, const Globals& g, int flags, double dTmx,
                                      const Carry& cy, double dtr, double Rmx, bool above_ground,
                                      Pass2Out& o, const MathK& K, Canary& cn, HK&& midway = NoHook()) {
    const double soilm = cy.soilm;
    // ---- section A operands: soil conductivity, ground heat flux, ground temperature
    double t_gfac = T(TF_GFAC), tc = T(TF_TC), tdew = T(TF_TDEW), ea = T(TF_EA);
    pin(t_gfac, tc, tdew, ea);
    constexpr bool soil_shared = SS;
    // --- soil conductivity / damping depth, cpp:1249-1260 ---------------------------------
    double ksoil, DD, rdd;
    if (soil_shared) {
        ksoil = S(SD_KSOIL);
        rdd = S(SD_RDD);
        DD = S(SD_DD);
    } else {
        double rho = C(CF_RHO), c_csa = C(CF_CSA), c_c1 = C(CF_C1), c_c1mc4 = C(CF_C1MC4), c_c3 = C(CF_C3);
        pin(rho, c_csa, c_c1, c_c1mc4, c_c3);
        ksoil = soil_ksoil(soilm, rho, c_c1, c_c1mc4, fexp(soil_ksoil_arg(soilm, c_c3), K));
        soil_damping(soilm, ksoil, rho, c_csa, DD, rdd);
    }
    // --- ground heat flux and ground temperature, cpp:1277-1296 ------------------------------
    double G = (t_gfac * dtr * ksoil) * rdd;
    if (F) cn.watch(G);                  // sqrt of the soil diffusivity: NaN for a non-positive conductivity
    cap<F>(G, 0.6 * Rmx, cn);
    flr<F>(G, -0.6 * Rmx, cn);
    const double Tg = pm_temperature_r<F, F && !TM::in_registers>(cy.num0 - G, cy.rden, dTmx, tc, tdew, cn);
    o.Tg = Tg;
    o.DD = DD;
    if (!above_ground) return;
    // ---- section B operands: TVaboveground up to the canopy temperature
    // (with the time values in registers, the three constants used last are read where they are used: one LDS latency
    // each there, six registers fewer from here to there)
    constexpr bool lean = TM::in_registers;
    double c_ghafac = C(CF_GHAFAC), c_smin = C(CF_SMIN), c_invrge = C(CF_INVRGE), c_pai = C(CF_PAI),
           c_ksat = C(CF_KSAT), c_psunsat = C(CF_PSUNSAT), c_shadefac = lean ? 0.0 : C(CF_SHADEFAC),
           c_ompc = lean ? 0.0 : C(CF_OMPC), c_svfa = lean ? 0.0 : C(CF_SVFA);
    double rlw = T(TF_RLW), rsw = T(TF_RSW), rdif = T(TF_RDIF), t_idx = T(TF_IDX), lapk = T(TF_LAPK),
           es = T(TF_ES), De = T(TF_DE), rem = T(TF_REM), ghr = T(TF_GHRRAD);
    if (lean) pin(c_ghafac, c_smin, c_invrge, c_pai, c_ksat, c_psunsat);
    else pin(c_ghafac, c_smin, c_invrge, c_pai, c_ksat, c_psunsat, c_shadefac, c_ompc, c_svfa, rlw, rsw, rdif, t_idx,
             lapk, es, De, rem, ghr);
    // --- TVaboveground, cpp:1411-1472 ----------------------------------------------------------
    const double uf = cy.uf;
    double gHa = uf * c_ghafac;
    flr<F>(gHa, 0.0001, cn);
    // Reciprocals that do not depend on each other are taken in pairs from one v_rcp_f64 each (frcp2) in the fast-clamp
    // vector-forcing kernels: the ground's saturation pressure with the canopy's series conductance (round 5: the stomatal
    // block runs first), the ground wetness factor with the canopy temperature, the canopy's saturation pressure with the
    // Lagrangian time scale, the leaf's with the far field's normalisation.  Which divisions are paired is fixed per code path
    // and per cell, never by what the other lanes of the wave do: results stay bit-identical whatever the tile geometry,
    // chunking or raster partition.
    constexpr bool PAIR = F && !lean;
    const double surfwet = (soilm - c_smin) * c_invrge;
    double esTg = 0.0, dgw = 0.0, gwet = 0.0;
    // the ground's saturation pressure -> wetness factor of the ground's vapour source, cpp:1417-1423
    auto ground_wetness = [&]() {
        double eT = esTg - ea;
        flr<F>(eT, 0.001, cn);
        double plf = 0.8753 - 1.7126 * flog(eT, K);
        dgw = 1.0 + fexp_s<F>(-plf, K);                     // eT in [1e-3, DBL_MAX]: -plf in (-13, 1215)
    };
    if (!PAIR) {
        esTg = satvap_f<F>(Tg, K);
        ground_wetness();
        gwet = frcp(dgw);
        flr<F>(gwet, surfwet, cn);
    }
    // canopy conductance, cpp:1425-1428 + 460-477
    const int idx = (int)t_idx;
    double gS = 9999.99;
    Stom st;
    st.gs2 = 0.0;
    bool have_gs2 = false;
    auto load_stom = [&]() {
        // theta-only part of stomcondCpp, cpp:451-455 with psiwfromthetaCpp cpp:382-389
        if (soil_shared) {
            st.gsmax = C(CF_GSMAX);
            st.rsmx = C(CF_RSMX);
            st.inv02rsmx = C(CF_INV02RSMX);
            st.gs2 = S(SD_GS2);
            pin(st.gsmax, st.rsmx, st.inv02rsmx, st.gs2);
        } else {
            // two batches of LDS reads rather than one: 11 operands at once are 22 VGPRs at the kernel's tightest spot
            double c_rat = C(CF_RAT), c_ratc = C(CF_RATC), c_invsmax = C(CF_INVSMAX), c_abspsie = C(CF_ABSPSIE),
                   c_soilb = C(CF_SOILB), c_psiw0 = C(CF_PSIW0);
            pin(c_rat, c_ratc, c_invsmax, c_abspsie, c_soilb, c_psiw0);
            const double Se = stom_se<F>(soilm, c_rat, c_ratc, c_invsmax, cn);
            double psiw = stom_psiw<F>(powxy(Se, -c_soilb, K), c_abspsie, c_psiw0, cn);
            pin1(psiw);
            st.gsmax = C(CF_GSMAX);
            st.rsmx = C(CF_RSMX);
            st.inv02rsmx = C(CF_INV02RSMX);
            double c_kk = C(CF_KK), c_mudeninv = C(CF_MUDENINV);
            pin(c_kk, c_mudeninv, st.gsmax, st.rsmx, st.inv02rsmx);
            st.gs2 = stom_gs2(fexp(-c_kk * psiw, K), c_mudeninv, st.gsmax);
        }
        have_gs2 = true;
    };
    if (!(flags & FL_OMPNAN)) {
        double kb, P_sun;
        if (idx & 256) {  // zenith (in degrees) beyond pi/2: k is the per-cell saturated value
            kb = c_ksat;
            P_sun = c_psunsat;
        } else {
            // the degrees-call operands (cpp:1425): from the time table, or — array forcing, where this branch is the rare
            // lane whose zenith angle in DEGREES is below pi/2 — made here from the carried zenith rather than carried
            double tanb, tan2b, inv2cosb;
            if (lean) {
                const double zend = T(TF_ZEND);
                tanb = tan(zend);
                tan2b = tanb * tanb;
                inv2cosb = 1.0 / (2.0 * cos(zend));
            } else {
                tanb = T(TF_TANB); tan2b = T(TF_TAN2B); inv2cosb = T(TF_INV2COSB);
            }
            kb = fsqrt(C(CF_XX) + tan2b) * C(CF_KDENINV);
            if (flags & (FL_XONE | FL_XINF | FL_XZERO))
                kb = (flags & FL_XONE) ? inv2cosb : (flags & FL_XINF) ? 1.0 : tanb;
            cap<F>(kb, 6000.0, cn);
            P_sun = fdiv_m(1.0 - fexp(-kb * c_pai, K), kb);
        }
        double P_shade = c_pai - P_sun;
        if (lean) { c_shadefac = C(CF_SHADEFAC); c_ompc = C(CF_OMPC); }
        double Rshade_abs = rdif * c_shadefac;
        double Rsun_abs = (rsw - rdif) * kb * (1 - c_ompc) + Rshade_abs;
        double gs_sun = 0.0, gs_shade = 0.0;
        if (!(Rsun_abs <= 0.0) || !(Rshade_abs <= 0.0)) {
            load_stom();
            gs_sun = stomcond<F>(Rsun_abs, st, K, cn);
            gs_shade = stomcond<F>(Rshade_abs, st, K, cn);
        }
        gS = gs_sun * P_sun + gs_shade * P_shade;
    }
    double gV = 0.0;
    if (PAIR) {
        // gS >= 0 here (conductances times sunlit / shaded leaf areas, P_sun <= pai), so gHa + gS >= 1e-4
        double nTg, dTg, rTg, rG;
        satvap_nd(Tg, nTg, dTg);
        frcp2(dTg, gHa + gS, rTg, rG);
        esTg = satvap_rd<F>(nTg, rTg, K);
        if (gS > 0.0) gV = (gHa * gS) * rG;
        ground_wetness();
    } else if (gS > 0.0) gV = fdiv_m(gHa * gS, gHa + gS);  // 1/(1/gHa + 1/gS)
    // canopy temperature, cpp:1430-1432 (Penman-Monteith with the linear surface wetness)
    if (lean) c_svfa = C(CF_SVFA);
    const double Rabs = cy.radCsw + 0.97 * c_svfa * rlw;
    const double mC = lapk * gV;
    const double esw = (es - ea) * surfwet;                 // shared by the canopy's and the leaf's latent heat terms
    const double numC = Rabs - rem - mC * esw - G, denC = 29.3 * (gHa + ghr) + mC * De;
    double Tcan;
    if (PAIR) {
        double rC;
        frcp2_m(dgw, denC, gwet, rC);
        flr<F>(gwet, surfwet, cn);
        Tcan = pm_temperature_r<F, F && !lean>(numC, rC, dTmx, tc, tdew, cn);
    } else {
        Tcan = pm_temperature<F, F && !lean>(numC, denC, dTmx, tc, tdew, cn);
    }
    double esTcan, muR = 0.0;                               // muR: uf/(a2*h) / uf^2, the below-canopy profile's (cpp:1389)
    if (PAIR) {
        double nTc, dTc, rTc;
        satvap_nd(Tcan, nTc, dTc);
        if (flags & FL_BELOW) frcp2(dTc, C(CF_A2H) * uf, rTc, muR);
        else rTc = frcp(dTc);
        esTcan = satvap_rd<F>(nTc, rTc, K);
    } else {
        esTcan = satvap_f<F>(Tcan, K);
    }
    midway();
    double ez;
    if (!(flags & FL_BELOW)) {
        // above canopy: log profile, cpp:1298-1313 / 1434-1441
        double w = C(CF_OML1);
        bool prof = (flags & FL_ABOVE1) != 0;
        o.Tz = prof ? tc + (Tcan - tc) * w : Tcan;
        ez = prof ? ea + (esTcan - ea) * surfwet * w : ea + (esTcan - ea) * surfwet;
        o.tleaf = Tcan;
        o.lwup = lw_emit(Tcan);
        o.lwdn = rlw;
    } else {
        // ---- section C operands: leaf temperature
        double c_uzfac = C(CF_UZFAC), emg = C(CF_EMG), ema = C(CF_EMA), invleafd = C(CF_INVLEAFD),
               c_hom = C(CF_HOM), c_homp = C(CF_HOMP), t_u2 = T(TF_U2);
        pin(c_uzfac, emg, ema, invleafd, c_hom, c_homp, t_u2);
        // ---- leaf temperature, cpp:1333-1364 -------------------------------------------------
        double uz = uf * c_uzfac;
        cap<F>(uz, t_u2, cn);
        const double lwcan = lw_emit(Tcan), lwgro = lw_emit(Tg);
        const double lwup = emg * lwgro + (1 - emg) * lwcan;
        const double lwdn = ema * rlw + (1 - ema) * lwcan;
        const double lwabs = 0.97 * 0.5 * (lwup + lwdn);
        // radLsw / radLpar are set to exactly 0 at night and for pai == 0 (cpp:1151-1162); written this
        // way a NaN leaf reflectance stays confined to the daytime values, as in the reference
        // (F: X is exactly 0 at night and for bare ground — pass 1 sets it only inside `rsw > 0 && pai > 0` — and the reflectance finite,
        // so the products below ARE the selected values: same bits, no compare and selects)
        const bool lit = F || (rsw > 0.0 && (flags & FL_PAI));
        const double leafabs = (lit ? c_hom * cy.X : 0.0) + lwabs;   // radLsw + lwabs
        double gh = (0.135 * 1.4) * fsqrt_m(uz * invleafd);
        const double RnetL = leafabs - lwcan;                // cpp:1319 with tc = Tcan
        // mincondCpp's floor is gmin = max(0.0463 * (|Hf| * |Rnet| / leafd)^0.2, 0.05) with |Hf| = 1/(1 + exp(2 - Hlf)) < 1,
        // so gmin <= max(0.0463 * (|Rnet|/leafd)^0.2, 0.05) whatever the stomatal resistance.  When gh clears THAT bound —
        // tested without a pow as (gh/0.0463)^5 >= |Rnet|/leafd, with a margin six orders above any rounding — neither of
        // the two floors can bind, and the pow() of the bound and the three exp() / two log() of |Hf|^0.2 are not
        // evaluated.  Wave-uniform (every lane in this branch must clear it); a NaN operand fails the test.
        bool floors_idle = false;
        {
            const double tq = gh * (1.0 / 0.0463), tq2 = tq * tq;
            const bool clear = gh >= 0.0500001 && tq2 * tq2 * tq >= (fabs(RnetL) * invleafd) * 1.000001;
            floors_idle = __builtin_amdgcn_ballot_w64(!clear) == 0;
        }
        double a02 = 0.0;
        if (!floors_idle) {
            a02 = mincond_a02<F>(RnetL, invleafd, K, cn);
            double gmin = mincond_gmin<F>(g.hf0p, a02, cn);  // mincondCpp(leafabs, 999.99, Tcan, leafd)
            flr<F>(gh, gmin, cn);
        }
        double gVl = gh;
        if (flags & FL_STOM) {
            gVl = 0.0;
            const double PARabs = lit ? c_homp * cy.X : 0.0; // radLpar
            double gs = 0.0;
            if (PARabs > 0.0) {
                if (!have_gs2) load_stom();
                gs = stomcond<F>(PARabs, st, K, cn);
            }
            if (!floors_idle) {
                // rs = min(1/gs, 500) (500 when gs <= 0), Hlf = 1.09767*rs^0.2672778,
                // Hf = -1/(1 + exp(2 - Hlf)), cpp:1321-1325; |Hf|^0.2 = exp(-0.2*log(1 + exp(2 - Hlf)))
                double hf02 = g.hf500p;                       // rs = 500: a constant (night, closed stomata)
                if (gs > 0.002) {
                    // gs in (0.002, DBL_MAX): arguments in (-190, 1.7), (-inf .. 2] with Hlf >= 0 finite, (-0.5, 0)
                    double Hlf = 1.09767 * fexp_b<F>(-0.2672778 * flog(gs, K), K);
                    hf02 = fexp_b<F>(-0.2 * flog(1.0 + fexp_b<F>(2.0 - Hlf, K), K), K);
                }
                double gmin = mincond_gmin<F>(hf02, a02, cn); // mincondCpp(leafabs, gs, Tcan, leafd)
                flr<F>(gh, gmin, cn);
            }
            if (gs > 0.0) gVl = fdiv_m(gh * gs, gh + gs);
        }
        const double mL = lapk * gVl;
        const double tleaf = pm_temperature<F, F && !lean>(leafabs - rem - mL * esw - 0.0,
                                               29.3 * (gh + ghr) + mL * De, dTmx, tc, tdew, cn);
        o.tleaf = tleaf;
        o.lwdn = lwdn;
        o.lwup = lwup;
        // ---- section D operands: canopy-top source + Lagrangian near/far field
        double w2 = C(CF_OML2), hgt = C(CF_HGT), c_inthh = C(CF_INTHH), c_inthz = C(CF_INTHZ),
               c_invhgt = C(CF_INVHGT), c_invhmz = C(CF_INVHMZ), omem = C(CF_OMEMPAI), nf = C(CF_NEARFAC),
               lden = C(CF_LEAFDEN);
        double mu = T(TF_MUPM), invmu = T(TF_INVMUPM);
        pin(w2, hgt, c_inthh, c_inthz, c_invhgt, c_invhmz, omem, nf, lden, mu, invmu);
        // ---- canopy-top source, cpp:1449 ---------------------------------------------------------
        const double HC = 29.3 * gHa * (Tcan - tc);
        const double ecw = (esTcan - ea) * surfwet;          // shared with the canopy-top vapour pressure below
        const double LC = mC * ecw;
        // (whether the canopy top lies above d + zh is a property of the CELL: a wave whose cells all do — vegetation always
        // does — takes the profile form without the per-lane selects; same operations per lane either way)
        bool prof2 = (flags & FL_ABOVE2) != 0;
        double Th, eh;
        if (F && __builtin_amdgcn_ballot_w64(!prof2) == 0) {
            Th = tc + (Tcan - tc) * w2;
            eh = ea + ecw * w2;
        } else {
            Th = prof2 ? tc + (Tcan - tc) * w2 : Tcan;
            eh = prof2 ? ea + ecw * w2 : ea + ecw;
        }
        // ---- Lagrangian near/far field, cpp:1365-1409 ---------------------------------------------
        const double z = g.reqhgt2;
        if (!PAIR) muR = frcp(C(CF_A2H) * uf);
        double Rc = c_inthh * muR;
        flr<F>(Rc, 0.001, cn);
        double Rz = c_inthz * muR;
        flr<F>(Rz, 0.001, cn);
        const double rKc = Rc * c_invhgt;                    // 1/Kc
        // Far field = (Kg SG + Kh SH + Kc SC) / (Kg + Kh + Kc) with Kc = hgt/Rc, Kg = 1/(Rz z), Kh = (1/(h-z))/(Rc - Rz).
        // The reference's form costs four divisions.  Multiplied through by Rc (Rz z) (Rc - Rz) the three conductances
        // become products and ONE reciprocal is left — the fast variant's form.  It differs where Rc = Rz (both floors
        // binding): there the reference makes inf * 0 = NaN of it, this form the finite limit, so that case trips the
        // canary and the tile is redone with the reference's form below.
        double Kc, Kg, Kh, invK;
        if (F) {
            const double B = Rz * z, D = Rc - Rz;
            if (D == 0.0) cn.trip();
            Kg = Rc * D;
            Kh = c_invhmz * (Rc * B);
            Kc = hgt * (B * D);
        } else {
            Kc = fdiv(hgt, Rc);
            Kg = frcp(Rz * z);
            Kh = fdiv(c_invhmz, Rc - Rz);
        }
        // the leaf's saturated vapour pressure and the far field's normalisation share a reciprocal
        double esTl;
        if (PAIR) {
            // A leaf with closed stomata (every night step: gs = 0, so gVl = mL = 0) has no latent heat flux: LL = 0, the near
            // field of the vapour profile is 0 whatever its limit, and the leaf's saturation pressure — the only other use —
            // is not needed (round 5).  The branch follows the lane's own conductance, so results do not depend on the
            // lane's neighbours; a night wave (three consecutive hours) skips it whole.
            if (gVl > 0.0) {
                double nTl, dTl, rTl;
                satvap_nd(tleaf, nTl, dTl);
                frcp2(dTl, Kg + Kh + Kc, rTl, invK);
                esTl = satvap_rd<F>(nTl, rTl, K);
            } else {
                invK = frcp(Kg + Kh + Kc);
                esTl = eh;
            }
        } else {
            if (F && !(gVl > 0.0)) esTl = eh;                              // (as above; the reference's clamps keep the evaluation)
            else esTl = satvap_f<F>(tleaf, K);
            invK = frcp(Kg + Kh + Kc);
        }
        const double HL = 29.3 * gh * (tleaf - tc);                       // cpp:1242
        const double LL = mL * (esTl - ea) * surfwet;                     // cpp:1243
        const double cp43 = 29.3 * 43.0;
        // temperature
        {
            double Flux = HC * omem, SH = Th * cp43, SG = Tg * cp43;
            double mxnear = fabs(tleaf - Th) * cp43;
            double SC = SH + Flux * rKc;
            double farg = (Kg * SG + Kh * SH + Kc * SC) * invK;
            double near = nf * (HL * lden);
            near_field_limit<F>(near, mxnear);
            o.Tz = (near + farg) * (1.0 / cp43);
        }
        // vapour pressure
        {
            double Flux = LC * omem, SH = eh * mu, SG = esTg * gwet * mu;
            double mxnear = fabs(esTl - eh) * mu;
            double SC = SH + Flux * rKc;
            double farg = (Kg * SG + Kh * SH + Kc * SC) * invK;
            double near = nf * (LL * lden);
            near_field_limit<F>(near, mxnear);
            ez = (near + farg) * invmu;
        }
    }
    // ez / satvap(Tz) * 100, cpp:1463-1465; 1 / (0.61078 exp(u)) = exp(-u) / 0.61078 spares the fast variant the division
    double rh;
    if (F) {
        rh = (ez * (100.0 / 0.61078)) * fexp_s<true>(-satvap_arg(o.Tz), K);     // Tz lies within 2 K of the temperatures above
    } else {
        rh = fdiv(ez, satvap(o.Tz, K)) * 100.0;
    }
    if (F) cn.watch(rh);                 // NaN with Tz or ez: the Lagrangian far field divides by Rc - Rz, which both floors can make 0
    cap<F>(rh, 100.0, cn);
    o.rh = rh;
    // clamp Tz to the source temperatures +-2, cpp:1467-1470 (all four are Penman-Monteith results: finite)
    double tmx = o.tleaf;
    flr<F>(tmx, tc, cn);
    flr<F>(tmx, Tg, cn);
    flr<F>(tmx, Tcan, cn);
    double tmn = o.tleaf;
    cap<F>(tmn, tc, cn);
    cap<F>(tmn, Tg, cn);
    cap<F>(tmn, Tcan, cn);
    cap<F>(o.Tz, tmx + 2.0, cn);
    flr<F>(o.Tz, tmn - 2.0, cn);
}

}  // namespace mcf
