// mcf_kernels.hip — gfx950 kernels of the grid microclimate solver.
//
//   k_twi_partial   sum/count of log(twi)/tfact (the solver's one global reduction, cpp:993-1004)
//   k_cell_setup    per-cell constant table (CellConst), cpp:2185-2190 + everything hoistable
//   k_time_setup    per-timestep table (TimeConst) for vector forcing, cpp:2153-2169 + hoistable
//   k_solve         the hot loop cpp:2180-2306 / 2452-2586: one lane per (cell, hour)
//   k_mxtc          per-cell running max of air temperature (array forcing, cpp:2467-2471)
//   k_belowground   Tbelowgroundv, cpp:1474-1539, one lane per cell
//   k_fill          constant fill (NA_real_ for steps past the last whole day)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>

#include "mcf_device.hpp"
#include "mcf_kernels.h"

#ifndef MCF_WAVES_PER_EU
#define MCF_WAVES_PER_EU 4   // waves per SIMD the vector-forcing kernels are built for (<= 128 VGPRs)
#endif
#ifndef MCF_AF_WAVES
#define MCF_AF_WAVES 3       // ... and the array-forcing kernels (168 VGPRs)
#endif
// Timing experiments (results wrong on purpose, only the launch time is read) are not in this file: they are patches
// under tools/variants/, applied to a scratch copy by tools/build_variant.sh.

namespace mcf {

// ------------------------------------------------------------------------------------
__global__ void k_fill(double* __restrict__ p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// sparse read-back of a ring slot: one lane per (sampled cell, step)
__global__ void k_gather_cells(RingView src, int64_t step0, int64_t nsteps, const int64_t* __restrict__ cells,
                               int64_t ncells, double* __restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncells * nsteps) return;
    const int64_t ci = i % ncells, k = i / ncells;
    dst[i] = src.at(cells[ci], step0 + k);
}

// The reference's [rows, cols, steps] layout out of the tiled ring (host fetches: PCIe-bound, 30 x slower than this).
// One lane per cell, blockIdx.y = step: the stores are contiguous runs along the raster; the loads take a whole
// 128-byte line (cells 0..15 of an hour) + 40 bytes (cells 16..20) per 21-cell tile, the three hours that share the
// second line meet in the L2.
__global__ __launch_bounds__(256) void k_tile_series(const double* __restrict__ src, RingView dst) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dst.N) return;
    const int64_t k = blockIdx.y;
    const_cast<double*>(dst.base)[dst.index(c, k)] = src[c + dst.N * k];
}
__global__ __launch_bounds__(256) void k_untile(RingView src, int64_t step0, double* __restrict__ dst) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= src.N) return;
    const int64_t k = blockIdx.y;
    dst[c + src.N * k] = src.at(c, step0 + k);
}

// ------------------------------------------------------------------------------------
// Deterministic two-stage reduction: kTwiParts workgroups reduce fixed strided subsets with a fixed-shape LDS
// tree into out[2 + 2p], a single lane then adds the partials in order into out[0..1].
constexpr int kTwiParts = 128;
__global__ __launch_bounds__(256) void k_twi_partial(const double* __restrict__ twi, int64_t n, double tfact,
                                                     double* __restrict__ out /* [2 + 2*kTwiParts] */) {
    __shared__ double ssum[256];
    __shared__ double scnt[256];
    double s = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kTwiParts * 256) {
        double v = twi[i];
        if (!isnan(v)) {
            s += log(v) / tfact;
            c += 1.0;
        }
    }
    ssum[threadIdx.x] = s;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            ssum[threadIdx.x] += ssum[threadIdx.x + w];
            scnt[threadIdx.x] += scnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 + 2 * blockIdx.x] = ssum[0];
        out[3 + 2 * blockIdx.x] = scnt[0];
    }
}
__global__ void k_twi_finish(double* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double s = 0.0, c = 0.0;
    for (int p = 0; p < kTwiParts; ++p) { s += out[2 + 2 * p]; c += out[3 + 2 * p]; }
    out[0] = s;
    out[1] = c;
}

// ------------------------------------------------------------------------------------
// cpp:294-310
__device__ inline double zeroplanedis(double h, double pai) {
    if (pai < 0.001) pai = 0.001;
    return (1.0 - (1.0 - exp(-sqrt(7.5 * pai))) / sqrt(7.5 * pai)) * h;
}
__device__ inline double roughlength(double h, double pai, double d, double psi_h) {
    double Be = sqrt(0.003 + (0.2 * pai) / 2);
    double zm = (h - d) * exp(-kKa / Be) * exp(kKa * psi_h);
    if (zm > (0.9 * (h - d))) zm = 0.9 * (h - d);
    if (zm < 0.0005) zm = 0.0005;
    return zm;
}
// cpp:104-121 (k only)
__device__ inline double cank_k(double zenr, double x) {
    double k;
    if (zenr > (kPi / 2.0)) zenr = kPi / 2.0;
    if (x == 1.0) k = 1.0 / (2.0 * cos(zenr));
    else if (isinf(x)) k = 1.0;
    else if (x == 0.0) k = tan(zenr);
    else k = sqrt(x * x + (tan(zenr) * tan(zenr))) / (x + 1.774 * pow((x + 1.182), -0.733));
    if (k > 6000.0) k = 6000.0;
    return k;
}
// cpp:1365-1375, the height integral of rhcanopy
__device__ inline double rh_integral(double h, double z) {
    if (!(z != h)) return 4.293251 * h;
    double a = (kPi * z) / h;
    double s = sin(a), c1 = cos(a) + 1;
    return (2.0 * h *
            ((48 * atan((sqrt(5.0) * s) / c1)) / pow(5.0, 1.5) +
             (32.0 * s) / (c1 * ((25.0 * (s * s)) / (c1 * c1) + 5.0)))) / kPi;
}

// Which lanes read a cell constant (pass1 / pass2 in mcf_device.hpp): 0 every valid cell, 1 only cells with a canopy
// (FL_PAI), 2 only cells whose sensor is below the canopy top (FL_BELOW), 3 only cells at or above it.  Constants a
// cell's path never reads may be 0/0 by construction (the canopy block of bare ground, the above-canopy profile weight
// of a sensor below the displacement height) and do not count against FL_REGULAR.
__device__ constexpr int field_reader_class(int f) {
    switch (f) {
    case CF_XX: case CF_KDENINV: return 0;          // also read by pass 2's canopy conductance for every cell
    case CF_OM: case CF_JDEL: case CF_GMA: case CF_GMA2: case CF_AGM: case CF_AGM2: case CF_U1: case CF_U2:
    case CF_KA1: case CF_KA2: case CF_KB1: case CF_KB2: case CF_KG1: case CF_KG2: case CF_KZ1: case CF_KZ2: case CF_INVD2: case CF_GMAGREF: case CF_LOGCLUMP: case CF_LOGGI: case CF_TRDN:
    case CF_TRDU: case CF_AMX: case CF_PAIAA: case CF_ALBD: case CF_RDDNG:
    case CF_RDDNZ: case CF_RDUPZ: case CF_PAIT: case CF_SHADEFAC:
        return 1;
    case CF_EMG: case CF_EMA: case CF_INVLEAFD: case CF_HOM: case CF_HOMP: case CF_OML2: case CF_HGT: case CF_A2H:
    case CF_INTHH: case CF_INTHZ: case CF_INVHGT: case CF_INVHMZ: case CF_OMEMPAI: case CF_NEARFAC: case CF_LEAFDEN:
        return 2;
    case CF_OML1:
        return 3;
    default:
        return 0;
    }
}

__host__ __device__ constexpr int64_t tile_image_doubles_dev(int cpb) { return (((int64_t)(CF_COUNT + kCellDirs) * cpb + 15) / 16) * 16; }
int64_t tile_image_doubles(int cpb) { return tile_image_doubles_dev(cpb); }

__global__ __launch_bounds__(256) void k_cell_setup(CellSetupArgs a) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.N) return;
    const int64_t N = a.N;
    // the cell's column of its tile's image (CellSetupArgs)
    const int cpb = a.cpb;
    const int64_t tile = c / cpb;
    double* out = a.cellc + tile * tile_image_doubles_dev(cpb) + (c - tile * cpb);
    // FL_REGULAR (mcf_device.hpp): every constant the cell's path reads is finite (field_reader_class above)
    bool fin[4] = {true, true, true, true};     // by field_reader_class
    auto put = [&](int f, double v) {
        out[f * cpb] = v;
        if (!isfinite(v)) fin[field_reader_class(f)] = false;
    };
    const double hgt = a.hgt[c], pai = a.pai[c], x = a.x[c], lref = a.leafr[c], ltra = a.leaft[c],
                 clump = a.clump[c], gref = a.gref[c], paia = a.paia[c];
    int flags = 0;
    if (!isnan(a.hgt0[c])) flags |= FL_VALID;
    if (pai > 0.0) flags |= FL_PAI;
    if (!(a.g.reqhgt2 >= hgt)) flags |= FL_BELOW;
    if (x == 1.0) flags |= FL_XONE;
    else if (isinf(x)) flags |= FL_XINF;
    else if (x == 0.0) flags |= FL_XZERO;
    // ---- solar index operands, cpp:96-97
    const double slope = a.slope[c], aspect = a.aspect[c];
    double ss = sin(slope * kToRad);
    put(CF_CS, cos(slope * kToRad));
    put(CF_SSCA, ss * cos(aspect * kToRad));
    put(CF_SSSA, ss * sin(aspect * kToRad));
    // ---- soil moisture spread, cpp:1005-1032
    const double Smin = a.Smin[c], Smax = a.Smax[c];
    double tadd = log(a.twi[c]) / a.tfact - a.twi_mean;
    put(CF_SMIN, Smin);
    put(CF_RGE, Smax - Smin);
    put(CF_INVRGE, 1.0 / (Smax - Smin));
    put(CF_ETA, exp(-tadd));
    // ---- canopy extinction, cpp:119
    put(CF_XX, x * x);
    put(CF_KDENINV, 1.0 / (x + 1.774 * pow((x + 1.182), -0.733)));
    // ---- two-stream diffuse constants, cpp:134-162 and 1034-1084
    const double pait = pai / (1.0 - clump);
    const double om = lref + ltra, aa = 1.0 - om, del = lref - ltra;
    double J = 1.0 / 3.0;
    if (x != 1.0) {
        double mla = 9.65 * pow((3.0 + x), -1.65);
        if (mla > kPi / 2.0) mla = kPi / 2.0;
        J = cos(mla) * cos(mla);
    }
    const double gma = 0.5 * (om + J * del);
    const double h = sqrt(aa * aa + 2.0 * aa * gma);
    const double S1 = exp(-h * pait);
    const double u1 = aa + gma * (1.0 - 1.0 / gref);
    const double u2 = aa + gma * (1.0 - gref);
    const double D1 = (aa + gma + h) * (u1 - h) * 1.0 / S1 - (aa + gma - h) * (u1 + h) * S1;
    const double D2 = (u2 + h) * 1.0 / S1 - (u2 - h) * S1;
    const double p1 = (gma / (D1 * S1)) * (u1 - h);
    const double p2 = (-gma * S1 / D1) * (u1 + h);
    const double p3 = (1.0 / (D2 * S1)) * (u2 + h);
    const double p4 = (-S1 / D2) * (u2 - h);
    double gi = 0.0;
    if (clump > 0.0) gi = pow(clump, paia / pai);
    if (gi > 0.99) gi = 0.99;
    double giu = 0.0;
    if (clump > 0.0) giu = pow(clump, (pai - paia) / pai);
    if (giu > 0.99) giu = 0.99;
    const double trd = gi * gi, trdn = clump * clump, trdu = giu * giu;
    const double paiaa = paia / (1.0 - gi);
    double amx = gref;
    if (amx < lref) amx = lref;
    double albd = (1.0 - trdn * trdn) * (p1 + p2) + trdn * trdn * gref;
    if (albd > amx) albd = amx;
    if (albd < 0.01) albd = 0.01;
    const double ehp = exp(h * pait), ehpa = exp(h * paiaa), emhpa = exp(-h * paiaa);
    double Rddn_g = (1.0 - trdn) * (p3 * S1 + p4 * ehp) + trdn;
    if (Rddn_g > 1.0) Rddn_g = 1.0;
    if (Rddn_g < 0.0) Rddn_g = 0.0;
    double Rdup_z = (1.0 - trdu * trdn) * (p1 * emhpa + p2 * ehpa) + trdu * trdn * gref;
    if (Rdup_z > 1.0) Rdup_z = 1.0;
    if (Rdup_z < 0.0) Rdup_z = 0.0;
    double Rddn_z = (1.0 - trd) * (p3 * emhpa + p4 * ehpa) + trd;
    if (Rddn_z > 1.0) Rddn_z = 1.0;
    if (Rddn_z < 0.0) Rddn_z = 0.0;
    const double omp = 0.5 * om;
    if (isnan(omp)) flags |= FL_OMPNAN;
    put(CF_PAIT, pait); put(CF_OM, om); put(CF_JDEL, J * del); put(CF_GMA, gma); put(CF_GMA2, gma * gma);
    put(CF_AGM, aa + gma); put(CF_AGM2, (aa + gma) * (aa + gma)); put(CF_U1, u1); put(CF_U2, u2);
    put(CF_INVD2, 1.0 / D2);
    {   // the direct-beam coefficients' per-cell factors and their four weighted sums (mcf_device.hpp CF_KA1 ..)
        const double id1 = 1.0 / D1, id2 = 1.0 / D2, is1 = 1.0 / S1, agm = aa + gma;
        const double k6a = (id1 * is1) * (u1 - h), k6b = id1 * (agm - h), k7a = (id1 * S1) * (u1 + h), k7b = id1 * (agm + h);
        const double k9a = (id2 * is1) * (u2 + h), k10a = (id2 * S1) * (u2 - h);
        put(CF_KA1, k6a - k7a); put(CF_KA2, k7b - k6b);                                         // p6 + p7
        put(CF_KB1, k6a * emhpa - k7a * ehpa); put(CF_KB2, k7b * ehpa - k6b * emhpa);           // p6 e^(-h paiaa) + p7 e^(h paiaa)
        put(CF_KG1, k10a * ehp - k9a * S1); put(CF_KG2, ehp - S1);                              // p9 S1 + p10 e^(h pait)
        put(CF_KZ1, k10a * ehpa - k9a * emhpa); put(CF_KZ2, ehpa - emhpa);                      // p9 e^(-h paiaa) + p10 e^(h paiaa)
    }
    put(CF_GREF, gref); put(CF_GMAGREF, gma * gref); // pow(clump, Kc) is evaluated as exp(Kc*log(clump)); log(0) = -inf is stored as -1e5 so that
    // the product stays finite (exp still underflows to exactly 0, as pow(0, Kc) does)
    put(CF_LOGCLUMP, fmax(log(clump), -1e5)); put(CF_LOGGI, fmax(log(gi), -1e5));
    put(CF_TRDN, trdn); put(CF_TRDU, trdu); put(CF_AMX, amx); put(CF_PAIAA, paiaa);
    put(CF_ALBD, albd); put(CF_RDDNG, Rddn_g);
    put(CF_RDDNZ, Rddn_z); put(CF_RDUPZ, Rdup_z);
    const double svfa = a.svfa[c];
    put(CF_SVFA, svfa);
    put(CF_HOM, 0.5 * (1.0 - om));
    put(CF_HOMP, 0.5 * (1.0 - omp));
    // ---- long wave, cpp:1165-1175 (pai == 0 is the trdif = 1 special case)
    double trdif = 1.0;
    if (pai > 0.0) trdif = (1.0 - trdn) * exp(-pait) + trdn;
    put(CF_TSV, trdif * svfa);
    put(CF_OMTRDIF, 1.0 - trdif);
    // ---- wind, cpp:1179-1218
    const double d = zeroplanedis(hgt, pai);
    double zm = roughlength(hgt, pai, d, 0.0);
    if (zm < 1e-6) zm = 1e-6;
    const double aw = pai / hgt;
    const double zref = a.g.zref, z = a.g.reqhgt2;
    put(CF_UFC, kKa / log((zref - d) / zm));
    double uzfac;
    if (z >= hgt) {
        uzfac = log((z - d) / zm) / kKa;
    } else {
        double r = log((hgt - d) / zm) / kKa;   // uh = uf * r
        if (r < 1.0) r = 1.0;                   // cpp:1206
        double Be = 1.0 / r;
        if (Be < 0.001) Be = 0.001;
        double Lc = 1.0 / (0.25 * aw);
        double Lm = 2 * (Be * Be * Be) * Lc;
        uzfac = r * exp(Be * (z - hgt) / Lm);
    }
    put(CF_UZFAC, uzfac);
    {
        double z0 = 0.2 * zm + d;                // cpp:375-377
        put(CF_GHAFAC, (kKa * 43) / log((zref - d) / (z0 - d)));
    }
    // ---- soil, cpp:628-636, 1249-1268
    const double psie = a.Psie[c], soilb = a.soilb[c], Vq = a.Vq[c], Vm = a.Vm[c], Mc = a.Mc[c], rho = a.rho[c];
    put(CF_ABSPSIE, fabs(psie));
    put(CF_INVSMAX, 1.0 / Smax);
    put(CF_SOILB, soilb);
    put(CF_RHO, rho);
    put(CF_CSA, 2400 * rho / 2.64);
    {
        double frs = Vm + Vq;
        double c1 = (0.57 + 1.73 * Vq + 0.93 * Vm) / (1.0 - 0.74 * Vq - 0.49 * Vm) - 2.8 * frs * (1.0 - frs);
        double c3 = 1.0 + 2.6 * pow(Mc, -0.5);
        double c4 = 0.03 + 0.7 * frs * frs;
        put(CF_C1, c1);
        put(CF_C1MC4, c1 - c4);
        put(CF_C3, c3);
    }
    // ---- stomatal parameters, cpp:391-440
    {
        const double lat = a.lats ? a.lats[c] : a.lat;
        double Rsmx = 420.0, psiw0 = -3.1, kk = 0.34, rat = 0.9;
        if (hgt < 1.0 && fabs(lat) < 22.5) { Rsmx = 450.0; psiw0 = -2.7; kk = 0.39; rat = 0.9; }
        if (hgt >= 1.0 && hgt < 7.0) { Rsmx = 430.0; psiw0 = -4.0; kk = 0.28; rat = 0.75; }
        if (hgt >= 7.0) {
            if (fabs(lat) < 22.5) { Rsmx = 500.0; psiw0 = -1.75; kk = 0.67; rat = 0.4; }
            else if (x < 0.8 || fabs(lat) > 58.0) { Rsmx = 420.0; psiw0 = -4.09; kk = 0.29; rat = 0.6; }
            else { Rsmx = 500.0; psiw0 = -2.51; kk = 0.46; rat = 0.45; }
        }
        const double gsmax = a.gsmax[c];
        if (gsmax < 999.99) flags |= FL_STOM;
        put(CF_GSMAX, gsmax); put(CF_RSMX, Rsmx); put(CF_INV02RSMX, 0.693147180559945309417 / (0.2 * Rsmx)); put(CF_RAT, rat);
        put(CF_RATC, (1 - rat) * kThetam); put(CF_PSIW0, psiw0); put(CF_KK, kk);
        put(CF_MUDENINV, 1.0 / (exp(-kk * psiw0) - 1.0));
        put(CF_SINLAT, sin(lat * kPi / 180.0));
        put(CF_COSLAT, cos(lat * kPi / 180.0));
        const double B = 0.261799 * ((4.0 * (a.lons ? a.lons[c] : a.lon)) / 60.0);   // cpp:44, 54
        put(CF_COSB, cos(B));
        put(CF_SINB, sin(B));
        put(CF_CROWPOS, a.crowpos ? a.crowpos[c % a.rows] : 0.0);
        put(CF_CCOLPOS, a.ccolpos ? a.ccolpos[c / a.rows] : 0.0);
        put(CF_ELEVD, a.elevd ? a.elevd[c] : 0.0);
        put(CF_PKFAC, a.pkfac ? a.pkfac[c] : 1.0);
    }
    // ---- canopy conductance operands for the saturated (degrees) cankCpp call, cpp:1425, 466-469
    {
        double ksat = cank_k(kPi / 2.0, x);
        put(CF_PAI, pai);
        put(CF_OMPC, omp);
        put(CF_KSAT, ksat);
        put(CF_PSUNSAT, (1.0 - exp(-ksat * pai)) / ksat);
        put(CF_SHADEFAC, ((1.0 - exp(-pai)) / pai) * (1.0 - omp));
    }
    // ---- log-profile weights, cpp:1298-1313
    {
        double zh = 0.2 * zm;
        double lden = log((zref - d) / zh);
        if (z > (d + zh)) flags |= FL_ABOVE1;
        if (hgt > (d + zh)) flags |= FL_ABOVE2;
        put(CF_OML1, 1 - log((z - d) / zh) / lden);
        put(CF_OML2, 1 - log((hgt - d) / zh) / lden);
    }
    // ---- leaf / below-canopy constants, cpp:1341-1347, 1365-1398, 1447
    put(CF_EMG, exp(-(pai - paia)));
    put(CF_EMA, exp(-paia));
    put(CF_INVLEAFD, 1.0 / a.leafd[c]);
    {
        double a2 = 0.4 * (1.0 - (d / hgt)) / (1.25 * 1.25);
        put(CF_A2H, a2 * hgt);
        put(CF_INTHH, 4.293251 * hgt);
        put(CF_INTHZ, rh_integral(hgt, z));
        put(CF_HGT, hgt);
        put(CF_INVHGT, 1.0 / hgt);
        put(CF_INVHMZ, 1.0 / (hgt - z));
        put(CF_NEARFAC, 3.047519 + 0.128642 * log(pai));
        put(CF_LEAFDEN, a.leafden[c]);
        put(CF_OMEMPAI, 1.0 - exp(-pai));
    }
    {
        bool dirs_ok = true;                        // horizons finite; a NaN wind-shelter value is the reference's "1" (cpp:1193)
        for (int d = 0; d < 24; ++d) { const double v = a.hor[(int64_t)d * N + c]; out[(CF_COUNT + d) * cpb] = v; dirs_ok = dirs_ok && isfinite(v); }
        for (int d = 0; d < 8; ++d) { const double v = a.wsa[(int64_t)d * N + c]; out[(CF_COUNT + 24 + d) * cpb] = v; dirs_ok = dirs_ok && !isinf(v); }
        const bool bare = pai == 0.0 && hgt == 0.0;
        const bool veg = pai > 0.0 && hgt > 0.0 && clump >= 0.0 && clump < 1.0 && a.leafd[c] > 0.0 && isfinite(x) && x > 0.0;
        // soil moisture stays positive (Smin >= 0, Smax > Smin), which keeps the matric potential's log and pow real
        const bool soil = Smin >= 0.0 && Smax > Smin && a.gsmax[c] >= 0.0 && a.g.zref > 0.0;
        const bool below = (flags & FL_BELOW) != 0;
        const bool fields = fin[0] && (veg ? fin[1] : true) && (below ? fin[2] : fin[3]);
        if (dirs_ok && soil && (bare || veg) && fields) flags |= FL_REGULAR;
    }
    put(CF_FLAGS, (double)flags);
}

// ------------------------------------------------------------------------------------
// Vector forcing: one lane per time step.  Table layout [day][field][hour].
__global__ __launch_bounds__(256) void k_time_setup(TimeSetupArgs a) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nsteps) return;
    TimeVals t;
    for (int f = 0; f < 15; ++f) t.v[f] = a.raw[f][k];
    SolDate sd = sol_date(a.year[k], a.month[k], a.day[k]);
    SolPos sp = sol_site(sd, a.hour[k], sin(a.lat * kPi / 180.0), cos(a.lat * kPi / 180.0), a.lon);
    derive_time(t, sp, dir_index(a.winddir[k], 45.0, 8));
    int dy = k / 24, hr = k % 24;
    {   // kSoilDaily: the day's 24 point soil moistures are one value (bitwise; a NaN never equals itself)
        const double* sm = a.raw[TF_SOILMP] + dy * 24;
        bool daily = true;
        for (int h = 1; h < 24; ++h) daily = daily && sm[h] == sm[0];
        if (daily) t.v[TF_IDX] = (double)((int)t.v[TF_IDX] | kSoilDaily);
    }
    double* dst = a.tt + ((int64_t)dy * TF_COUNT) * 24 + hr;
    for (int f = 0; f < TF_COUNT; ++f) dst[f * 24] = t.v[f];
}

// Array forcing: per-timestep date part of the solar position + wind index.
__global__ __launch_bounds__(256) void k_date_setup(DateSetupArgs a) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nsteps) return;
    SolDate sd = sol_date(a.year[k], a.month[k], a.day[k]);
    const double A = 0.261799 * (a.hour[k] + sd.eot / 60.0 - 12.0);      // time part of tt, cpp:44, 54
    a.dt[4 * (int64_t)k + 0] = sd.sindec;
    a.dt[4 * (int64_t)k + 1] = sd.cosdec;
    a.dt[4 * (int64_t)k + 2] = cos(A);
    a.dt[4 * (int64_t)k + 3] = sin(A);
    a.windex[k] = dir_index(a.winddir[k], 45.0, 8);
}

// Array forcing: running per-cell max of tc over a slab of `nsteps` steps.
__global__ __launch_bounds__(256) void k_mxtc(const double* __restrict__ tc, int64_t N, int nsteps,
                                              double* __restrict__ mx) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    double m = mx[c];
    for (int k = 0; k < nsteps; ++k) {
        double v = tc[c + N * k];
        if (v > m) m = v;
    }
    mx[c] = m;
}

// coarse array forcing: the same maximum over the interpolated (and altitude-corrected) series; the coarse fields are
// L2-resident
__global__ __launch_bounds__(256) void k_mxtc_coarse(const double* __restrict__ force, int64_t stride, int crows, int ccols,
                                                     int nsteps, const double* __restrict__ rowpos,
                                                     const double* __restrict__ colpos, int64_t rows, int64_t N,
                                                     int altcorrect, const double* __restrict__ elevd,
                                                     const double* __restrict__ pkfac, double* __restrict__ mx) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const CoarseTap tap(rowpos[c % rows], colpos[c / rows], crows, ccols);
    const int64_t cN = (int64_t)crows * ccols;
    MathK MK;
    MK.set();
    const double ed = altcorrect ? elevd[c] : 0.0, pf = altcorrect ? pkfac[c] : 1.0;
    double m = -273.15;
    for (int k = 0; k < nsteps; ++k) {
        const double* q = force + cN * k;
        double v = tap(q + (int64_t)TF_TC * stride);
        if (altcorrect == 1) v += 0.005 * ed;
        if (altcorrect == 2) {
            const double ea = satvap_r(v, MK) * tap(q + (int64_t)TF_ES * stride) / 100.0;
            v += lapserate_r(v, ea, tap(q + (int64_t)TF_PK * stride) * pf) * ed;
        }
        if (v > m) m = v;
    }
    mx[c] = m;
}

// ------------------------------------------------------------------------------------
// The solver.  Workgroup = one tile of CPB consecutive cells x 24 hours, one lane per (cell, hour).
//   AF   0 vector forcing, 1 array forcing (runmicro2Cpp geometry), 2 coarse array forcing
//   BG   reqhgt < 0: store the ground temperature series and the damping-depth sum
//
// Output ring.  reqhgt >= 0 writes the TILED ring: the values one workgroup produces for one variable on one day —
// CPB cells x 24 hours — form one block of ring_block_doubles(CPB) doubles in the workgroup's own lane order
// (mcf_kernels.h `ring_pos`), blocks ordered [tile][day of slot][variable].  Every wave stores 64 consecutive doubles
// = four whole 128-byte lines per variable, a tile-day is one contiguous 40 KB run, and the address of a store is
// (uniform block base) + (the lane's constant position): the base lives in SGPRs, no per-store vector arithmetic.
// The reference's [rows, cols, steps] view (src/microclimfCpp.cpp:2292-2303) is re-created by the consumers
// (RingView: k_untile, k_gather_cells, k_pack_transpose, k_pack_nc, k_bioclim).  Until round 3 the ring itself was
// [variable][step][cell]: 240 row segments of 168 B per workgroup-day, consecutive hours 133 MB apart at 4096^2,
// partial lines completed by the neighbouring tiles — a pure store stream at 32 % of the HBM peak.
// reqhgt < 0 keeps the linear layout (k_belowground smooths whole series in place).
// ------------------------------------------------------------------------------------
// threads per workgroup: CPB*24 lanes rounded up to whole waves (21 cells: 8 waves, 32: 12, 16: 6 — two 16-cell workgroups of the
// 168-VGPR array-forcing kernel fill a CU's twelve wave slots)
constexpr int solve_threads(int cpb) { return ((cpb * 24 + 63) / 64) * 64; }
static_assert(solve_threads(21) == 512 && solve_threads(32) == 768, "ring_block_doubles() must agree");

// One tile (CPB consecutive cells) over days [day0, day0 + ndays) of the launch described by `a`.
//   F   fast clamps (mcf_device.hpp `cap`): only for tiles / days the host has classified REGULAR; a workgroup in which a
//       canary trips appends its tile to a.fix_list and k_solve_fix redoes the tile's days of this launch with F = false
//   SSREQ  per cell-day soil state shared through LDS (mcf_device.hpp SoilDay): only for launches whose days are all kSoilDaily
template <int CPB, int AF, bool BG, bool F, bool SSREQ>
__device__ __forceinline__ void solve_tile(const SolveArgs& a, const int64_t tile, const int day0, const int ndays, const int rot) {
    constexpr int NT = solve_threads(CPB);
    static_assert(NT == RING_BLOCK(CPB), "tile-day block = one value per lane");
    // the tile's LDS image: [CF_COUNT cell fields + 24 horizon + 8 wind-shelter rows][CPB], as it lies in the table
    constexpr int IMG = (int)tile_image_doubles_dev(CPB);
    __shared__ __attribute__((aligned(256))) double s_tile[IMG];
    double* const s_cell = s_tile;
    __shared__ double s_time[AF ? 1 : 3 * TF_COUNT * 24];
    // day-reduction staging: per (hour, cell) values for every lane to walk after the barrier, or — 21-cell tiles — ONE slot
    // per cell and statistic that the waves fold their partial extremes into with LDS atomics (ds_max_f64 / ds_min_f64):
    // after the barrier a lane reads its cell's three results instead of reducing 8 x 3 partials (24 LDS reads and 24
    // v_max / v_min per cell-step, by each of the cell's 24 hour lanes).  Three buffers: the one of day d + 2 is reset behind
    // the barrier of day d, when every wave is past reading it (it held day d - 1) and none can write it before the barrier of
    // day d + 1.
    constexpr bool PRE = CPB == 21;
    __shared__ double s_ext[3][3][CPB];        // [buffer][max Tg0, min Tg0, max |Rnet|][cell]
    __shared__ double s_dd[BG ? 24 * CPB : 1];
    // per cell-day soil state (mcf_device.hpp SoilDay): a ring of three days, filled two days ahead by one wave
    constexpr bool SS = SSREQ && (AF == 0) && (2 * CPB <= 64);
    __shared__ double s_soil[SS ? 3 * SD_COUNT * CPB : 1];
    __shared__ int s_trip;      // F: a wave of this workgroup tripped its canary
    // Coarse array forcing with the taps staged in LDS (CLDS = the SSREQ instantiation of AF == 2; mcf_plan_create chooses it
    // when every tile touches at most 4 coarse rows per raster column).  The 32 cells of a tile lie in at most two raster
    // columns (rows >= 32), and within one column every cell has the same two coarse columns and the same weight wx: the
    // bilinear interpolation's inner step — along the columns — is done ONCE per tile, day, series, hour and coarse row
    // (2 x 4 x 15 x 24 values, ~4 per lane and day: 8 loads) and a lane is left with the step along the rows, two LDS reads and
    // two instructions per series, instead of 4 global loads and 6 instructions (52 loads per cell-step before).  Double
    // buffered by day: the values of day d + 1 are written in front of pass 1 of day d, the day's barrier orders them before
    // their readers.  Same operations as CoarseTap in the same order (CoarseTap::mix): same bits.
    constexpr bool CLDS = (AF == 2) && SSREQ;
    constexpr int CU_ROWS = 4, CU_F = 15, CU_SLOT = CU_ROWS * CU_F * 24;
    __shared__ double s_cu[CLDS ? 2 * 2 * CU_SLOT : 1];      // [day parity][column slot][coarse row][series][hour]
    __shared__ double s_cw[2];                    // wx of the two column slots
    __shared__ int s_ci[9];                       // per slot: first coarse row, rows staged, coarse columns c0, c1; [8]: slot 1's first cell

    const int tid = threadIdx.x;
    int cl = tid % CPB;
    bool lane_on = tid < CPB * 24;
    int hr;
    uint32_t pos;       // the lane's place in a tile-day block of the ring = ring_pos(CPB, cl, hr) for the lanes that are on
    if (CPB == 21) {
        // With cell = t % 21 a 16-lane group (the unit in which ds_read_b64 resolves banks) straddles two hours in 15 of
        // 21 cases and then holds cells {16..20, 0..10}: cells k and 16+k fall in the same bank pair (2-way conflicts on
        // every per-cell table read; SQ_LDS_BANK_CONFLICT = 13 % of the kernel's cycles).  Each wave therefore takes 3
        // hours as three full groups (cells 0..15 of one hour each) plus one group holding cells 16..20 of the same three
        // hours (15 lanes, one idle): 16 distinct consecutive addresses, or 5 distinct ones read by 3 lanes each.
        const int w = tid >> 6, l = tid & 63;
        if (l < 48) { cl = l & 15; hr = 3 * w + (l >> 4); }
        else { const int j = l - 48; cl = 16 + (j % 5); hr = 3 * w + (j / 5); lane_on = j < 15; if (!lane_on) { cl = 20; hr = 3 * w + 2; } }
        // Waves w and w+4 of a workgroup share a SIMD and sit 12 hours apart: a day and a night wave at the equinox, but two
        // day waves (hours 6-8 and 18-20) on one SIMD in summer and two night waves in winter.  Two workgroups are resident
        // per CU; shifting every other one by six hours puts the complementary pattern on the same SIMDs.
        hr = hr + 6 * rot;
        hr -= hr >= 24 ? 24 : 0;
        pos = (uint32_t)(64 * (hr / 3) + l);        // lanes 0..47: 16 (hr % 3) + cell; 48..62: 48 + 5 (hr % 3) + cell - 16; 63: padding
    } else if (CPB == 32) {
        // Waves w, w+4, w+8 of a workgroup share a SIMD.  Daytime waves carry the short-wave block (about twice the work of
        // a night wave), so hour pairs are dealt to waves so that every SIMD gets a midday, a morning/evening and a night pair.
        const int wave = tid >> 6;
        const int tab = ((wave & 3) == 0) ? (wave == 0 ? 12 : wave == 4 ? 16 : 20)
                      : ((wave & 3) == 1) ? (wave == 1 ? 10 : wave == 5 ? 6 : 2)
                      : ((wave & 3) == 2) ? (wave == 2 ? 14 : wave == 6 ? 18 : 22)
                                          : (wave == 3 ? 8 : wave == 7 ? 4 : 0);
        hr = tab + ((tid >> 5) & 1);
        pos = (uint32_t)(hr * 32 + cl);
    } else {
        hr = tid / CPB;
        if (hr > 23) hr = 23;
        pos = (uint32_t)tid;                         // hour * CPB + cell; lanes past 24 * CPB: padding
    }
    uint32_t posb = pos * 8u;       // byte offset of the lane's value in a block
    const int64_t N = a.N;
    const int64_t c0 = tile * CPB;
    const int64_t c = c0 + cl;
    const bool in_grid = lane_on && c < N;   // the other lanes only help staging, keep the barriers and write the padding

    // ---- stage the tile's image (one contiguous, line-aligned run per (layer, tile): 16 bytes per lane and load, no index
    // arithmetic), the first day's time table and the exp / log tables in LDS.  ALL global loads of the prologue are issued
    // before the first one is waited for: written as plain copy loops, hipcc made load -> s_waitcnt vmcnt(0) -> ds_write of
    // every iteration — eight dependent trips through a memory system busy with this kernel's own stores, 13 700 cycles
    // (5.7 us, nearly a whole day of the workgroup's eight wave slots) from entry to the first barrier.
    auto image_of = [&](int layer) { return reinterpret_cast<const double2*>(a.cellc + ((int64_t)layer * a.ntiles_total + tile) * IMG); };
    // (a change of vegetation layer inside the launch restages the image with this plain loop: rare)
    auto stage_cells = [&](int layer) {
        const double2* src = image_of(layer);
        double2* dst = reinterpret_cast<double2*>(s_tile);
        for (int q = tid; q < IMG / 2; q += NT) dst[q] = src[q];
    };
    const CellLds<CPB> C{s_cell + cl};
    int flags = 0;
    bool valid = false;
    // vegetation layer of a day (runmicro3Cpp/4Cpp `dfsel`, cpp:2760-2768)
    int cur_layer = a.daylayer ? a.daylayer[day0] : 0;
    __shared__ double s_exptab[256];
    __shared__ __attribute__((aligned(16))) double s_logtab[512];
    {
        constexpr int NIMG = (IMG / 2 + NT - 1) / NT, NTT = AF ? 0 : (TF_COUNT * 24 + NT - 1) / NT, NLOG = (512 + NT - 1) / NT;
        double2 im[NIMG];
        double tt0[NTT ? NTT : 1], lg[NLOG];
        const double2* isrc = image_of(cur_layer >= 0 ? cur_layer : 0);
        const double* tsrc = a.tt + (int64_t)day0 * TF_COUNT * 24;
        // clamped indices instead of predicates: every lane loads, nothing is written into the loads' registers first
#pragma unroll
        for (int i = 0; i < NIMG; ++i) { const int q = tid + i * NT; im[i] = isrc[q < IMG / 2 ? q : IMG / 2 - 1]; }
#pragma unroll
        for (int i = 0; i < NTT; ++i) { const int q = tid + i * NT; tt0[i] = tsrc[q < TF_COUNT * 24 ? q : TF_COUNT * 24 - 1]; }
        const double ex = kExp2Tab[tid & 255];
#pragma unroll
        for (int i = 0; i < NLOG; ++i) lg[i] = kLogTab[(tid + i * NT) & 511];
        // (pinned: left alone, the compiler sinks each load into the guarded store that uses it and waits for it there)
#pragma unroll
        for (int i = 0; i < NIMG; ++i) pin(im[i].x, im[i].y);
#pragma unroll
        for (int i = 0; i < NTT; ++i) pin1(tt0[i]);
        if (cur_layer >= 0) {
#pragma unroll
            for (int i = 0; i < NIMG; ++i) { const int q = tid + i * NT; if (q < IMG / 2) reinterpret_cast<double2*>(s_tile)[q] = im[i]; }
        }
#pragma unroll
        for (int i = 0; i < NTT; ++i) { const int q = tid + i * NT; if (q < TF_COUNT * 24) s_time[q] = tt0[i]; }
        if (tid < 256) s_exptab[tid] = ex;
#pragma unroll
        for (int i = 0; i < NLOG; ++i) { const int q = tid + i * NT; if (q < 512) s_logtab[q] = lg[i]; }
    }
    // Every load of the prologue has landed — said explicitly, as an instruction the compiler's wait-count pass sees.  Some of
    // the LDS writes above sit behind lane predicates (`q < 512`); on the path that skips one, the pass keeps the load into
    // that register "possibly in flight" for ever, carries that into the day loop through the loop header, and drains the
    // memory counter (vmcnt(0): every store in flight as well) in front of the first write to that register in each region
    // of every day.  Hygiene: with the drains gone the launch time is the same to 0.1 % — a store's acknowledgement from the
    // L2 is quick; what the output stream costs is clock (profiles/r03_timing_experiments.txt).
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
    if (F && tid == 0) s_trip = 0;
    if (tid < 9 * CPB) (&s_ext[0][0][0])[tid] = ((tid / CPB) % 3 == 0) ? -999.0 : ((tid / CPB) % 3 == 1) ? 999.0 : -999.9;   // cpp:2196-2198
    Globals g = a.g;
    double dTmx = g.dTmx;
    if (AF && in_grid) dTmx = -0.6273 * a.mxtc[c] + 49.79;   // cpp:1236 with the per-cell mxtc
    // the fast clamps take the two caps of cpp:1237-1238 as one (pm_temperature); a NaN dTmx is ignored by both forms
    // (vector forcing only: mcf_device.hpp pm_temperature M80)
    const double dTcap = (F && !AF) ? fmin(dTmx, 80.0) : dTmx;
    // TVaboveground (cpp:2272) is only evaluated when one of its outputs was requested
    const bool need_tv = g.reqhgt >= 0.0 && a.need_tv != 0;
    MathK MK;
    MK.set();
    MK.tables(s_exptab, s_logtab);     // filled above, visible after the prologue's barrier
    MK.pin(true, AF == 0, AF == 1);     // (the coarse kernel is short of SGPRs: the two extra residents cost it 4 %)   // exp and log coefficients resident in SGPRs for the whole day loop (the array-forcing kernels
                             // have no VGPR to spare for the second constants)
    const double NA = na_real();

    static_assert(TF_TC == 0 && TF_SOILMP == 9 && TF_UMU < 15 && TF_GP < 15 && TF_KP < 15 && TF_MUGP < 15 && TF_DTRP < 15, "the 15 forcing series");
    Canary cn;      // F: NaN as soon as one watched clamp of this lane has met a NaN, on any day of the launch
    int run = 0;    // days this workgroup has started: indexes the time, reduction and soil rings
    // soil state of day `d` into ring slot `slot`, by the calling wave (all 64 lanes call)
    auto produce_soil = [&](int d, int slot) {
        if (SS) {
            // the day's point-model soil moisture: a wave-uniform address, read through the scalar cache (its own counter: a
            // vector load here would make the producing wave wait for the ten stores it has just issued)
            const double* pt = a.tt + ((int64_t)__builtin_amdgcn_readfirstlane(d) * TF_COUNT + TF_SOILMP) * 24;
            const double smp = *(const __attribute__((address_space(4))) double*)(uintptr_t)pt;
            soil_day_produce<SS ? CPB : 1, F>(s_cell, smp, s_soil + (slot % 3) * (SD_COUNT * CPB), tid & 63, MK);
        }
    };
    // after the tile's constants of `cur_layer` have landed in LDS (barrier before): the lane's flags, and the soil ring
    // (re)started with this layer's constants — day `d` and the next, one wave each; from there on the wave whose turn it
    // is fills, behind each day's barrier, the slot of the day after next
    auto enter_layer = [&](int d) {
        flags = (in_grid && cur_layer >= 0) ? (int)s_cell[CF_FLAGS * CPB + cl] : 0;
        valid = (flags & FL_VALID) != 0;
        if (SS && cur_layer >= 0) {
            const int wv = tid >> 6;
            if (wv == 0) produce_soil(d, run);
            if (wv == 1 && d + 1 < day0 + ndays) produce_soil(d + 1, run + 1);
            __syncthreads();
        }
    };
    __syncthreads();
    // CLDS: the column-interpolated coarse values of day `d` into buffer `buf` (all lanes; s_ci / s_cw set below).  The items —
    // (coarse row the tile touches, series, hour) — are dealt compactly: a tile inside one raster column with two coarse rows has
    // 720 of them, ONE per lane.  A lane's first item is only LOADED by stage_issue (two values held in registers) and written
    // by stage_commit in front of the day's barrier, so that the loads' latency runs under pass 1 (one 12-wave workgroup per
    // CU: nothing else hides it); the rare further items are loaded and written at once.
    auto stage_item = [&](int q, int d, int& dst, const double*& p0, const double*& p1, double& w) {
        const int nr0 = s_ci[1], nr1 = s_ci[5];
        const int rr = q / (CU_F * 24), rem = q - rr * (CU_F * 24), f = rem / 24, h = rem - f * 24;
        const int sl = rr >= nr0 ? 1 : 0, r = rr - (sl ? nr0 : 0);
        const bool on = rr < nr0 + nr1 && f != TF_TDEW;
        const int64_t cN = (int64_t)a.crows * a.ccols;
        const double* fp = a.af_base + cN * ((int64_t)d * 24) + (int64_t)f * a.af_stride + (int64_t)h * cN + (s_ci[4 * sl] + r);
        p0 = fp + (int64_t)a.crows * s_ci[4 * sl + 2];
        p1 = fp + (int64_t)a.crows * s_ci[4 * sl + 3];
        w = s_cw[sl];
        dst = on ? (sl * CU_ROWS + r) * (CU_F * 24) + rem : -1;
    };
    auto stage_issue = [&](int d, int buf, double& v0, double& v1) {
        int tq = tid;
        asm volatile("" : "+v"(tq));        // (opaque per call: or hipcc keeps the items' 64-bit addresses alive across the day loop)
        int dst;
        const double *p0, *p1;
        double w;
        stage_item(tq, d, dst, p0, p1, w);
        v0 = v1 = 0.0;
        if (dst >= 0) { v0 = *p0; v1 = *p1; }
        const int items = (s_ci[1] + s_ci[5]) * (CU_F * 24);
        for (int q = tq + NT; q < items; q += NT) {       // workgroup-uniform trip count; normally zero
            stage_item(q, d, dst, p0, p1, w);
            if (dst >= 0) s_cu[buf * (2 * CU_SLOT) + dst] = CoarseTap::mix(*p0, *p1, w);
        }
    };
    auto stage_commit = [&](int d, int buf, double v0, double v1) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        int dst;
        const double *p0, *p1;
        double w;
        stage_item(tq, d, dst, p0, p1, w);
        if (dst >= 0) s_cu[buf * (2 * CU_SLOT) + dst] = CoarseTap::mix(v0, v1, w);
    };
    if (CLDS) {
        // the tile's two column slots: slot 0 = the raster column of its first cell, slot 1 = the next one, from the cell at
        // which the tile wraps into it (rows restart: the row position falls, or — a one-row coarse grid — the column
        // position changes; if neither does, both columns read the same coarse cells with the same weights)
        if (tid < 64) {
            const int k = tid & 31;
            const bool have = c0 + k < N;
            const double rp = s_cell[CF_CROWPOS * CPB + k], cp = s_cell[CF_CCOLPOS * CPB + k];
            const double rpp = s_cell[CF_CROWPOS * CPB + (k ? k - 1 : 0)], cpp = s_cell[CF_CCOLPOS * CPB + (k ? k - 1 : 0)];
            const bool brk = have && k > 0 && tid < 32 && (rp < rpp || cp != cpp);
            const uint64_t mb = __builtin_amdgcn_ballot_w64(brk), mh = __builtin_amdgcn_ballot_w64(have && tid < 32);
            const int nk = 64 - __builtin_clzll(mh | 1ull);                 // cells of the tile inside the raster
            const int b = mb ? __builtin_ctzll(mb) : nk;                    // first cell of slot 1 (= nk: there is none)
            if (tid < 2) {
                const int fk = tid ? (b < nk ? b : 0) : 0, lk = tid ? nk - 1 : b - 1;
                const double rpf = s_cell[CF_CROWPOS * CPB + fk], rpl = s_cell[CF_CROWPOS * CPB + lk], cpx = s_cell[CF_CCOLPOS * CPB + fk];
                const int rb = (int)floor(rpf), re = (int)floor(rpl) + 1;
                const int cc0 = (int)floor(cpx), cc1 = cc0 + 1 < a.ccols ? cc0 + 1 : cc0;
                int nr = (re < a.crows ? re : a.crows - 1) - rb + 1;
                if (tid == 1 && b >= nk) nr = 0;
                s_ci[4 * tid + 0] = rb; s_ci[4 * tid + 1] = nr < CU_ROWS ? nr : CU_ROWS; s_ci[4 * tid + 2] = cc0; s_ci[4 * tid + 3] = cc1;
                s_cw[tid] = cpx - floor(cpx);
                if (tid == 0) s_ci[8] = b;
            }
        }
        __syncthreads();
        double v0, v1;
        stage_issue(day0, 0, v0, v1);
        stage_commit(day0, 0, v0, v1);
        __syncthreads();
    }
    enter_layer(day0);
    // the tile's first block of this launch in the tiled ring (uniform: SGPRs); a day's ten stores are
    // [block base + variable slab * block] + pos
    double* ring_day = BG ? nullptr : a.out_base + tile * a.out_tile_stride + (int64_t)a.slot_day0 * a.out_day_stride;
    for (int dl = 0; dl < ndays; ++dl, ++run) {
        const int dabs = day0 + dl;
        // the tile's cell constants are restaged whenever the day's vegetation layer changes — workgroup-uniform and rare
        const int layer = a.daylayer ? a.daylayer[dabs] : 0;
        if (layer != cur_layer) {
            __syncthreads();
            if (layer >= 0) stage_cells(layer);
            __syncthreads();
            cur_layer = layer;
            enter_layer(dabs);
        }
        const int64_t kl = (int64_t)(dabs - a.day0) * 24 + hr;   // step within the launch
        // Left alone, hipcc hoists a 64-bit "variable v is requested" mask and a 64-bit slab pointer per output variable out
        // of the day loop: 40 SGPRs that do not fit and are spilled to VGPR lanes (v_readlane + hazard nops around every
        // store).  Making the selector opaque once per day keeps two SGPRs live instead; the slab base is re-derived by a
        // handful of SALU ops per store.
        uint64_t osel = a.out_sel;
        asm volatile("" : "+s"(osel));
        double* lin = nullptr;                 // linear ring (reqhgt < 0): the lane's element of slab 0
        if (BG) lin = a.out_base + (c + N * (a.slot_step0 + kl));
        auto put = [&](int v, double val) {
            const unsigned sel = (unsigned)(osel >> (4 * v)) & 15u;
            if (sel == 15u) return;
            if (BG) { if (in_grid) lin[(int64_t)sel * a.out_stride] = val; }
            else {
                // SGPR base + the lane's 32-bit byte offset (global_store ... v_off, v_data, s[base]): no vector arithmetic per
                // store.  The empty asm keeps the zero-extension of the offset in the store's own basic block, where
                // instruction selection can fold it into the addressing mode.
                asm("" : "+v"(posb));
                *(double*)((char*)ring_day + ((size_t)sel * (NT * 8)) + posb) = val;
            }
        };
        // issue the loads of the next day's table rows now; they land in LDS after pass 1
        constexpr int TPER = (TF_COUNT * 24 + NT - 1) / NT;
        double pre[TPER];
        const bool stage = !AF && dl + 1 < ndays;
        if (stage) {
            const double* src = a.tt + (int64_t)(dabs + 1) * TF_COUNT * 24;
#pragma unroll
            for (int i = 0; i < TPER; ++i) {
                int q = tid + i * NT;
                // every lane loads (no default value to write into the load's registers first: that write made the compiler
                // drain the memory counter at the top of every day)
                pre[i] = src[q < TF_COUNT * 24 ? q : TF_COUNT * 24 - 1];
            }
        }
        double cs0 = 0.0, cs1 = 0.0;        // CLDS: the lane's staging item of the next day, in flight across pass 1
        if (CLDS && dl + 1 < ndays) stage_issue(dabs + 1, (run + 1) & 1, cs0, cs1);
        TimeVals tv;
        // the tile-day block of the tiled forcing ring (uniform) — the lane's value of series f is at [f][pos]
        const double* fday = (AF == 1) ? a.af_base + tile * a.af_tile_stride + (int64_t)(dabs - a.day0) * a.af_day_stride : nullptr;
        // (pb: an opaque copy of the lane offset made ONCE in the loads' own block — one move instead of one per load; see `put`)
        auto force = [&](int f, unsigned pb) { return *(const double*)((const char*)fday + (size_t)f * (NT * 8) + pb); };
        if (AF && valid) {
            const int64_t kabs = (int64_t)dabs * 24 + hr;
            // pass 2's ground heat flux takes four more series, through ONE value (cpp:1282-1289): loaded here with the rest,
            // so that one memory latency is exposed per day instead of two, and carried as that value
            double p2gp = 1.0, p2mugp = 1.0, p2dtrp = 1.0, p2kp = 1.0;
            if (AF == 2) {
                // coarse arrays: interpolate, then derive what `.runmodel2Cpp` derives after resampling
                // (slot TF_ES carries relhum, TF_U2 / TF_EA the wind components u, v)
                const CoarseTap tap(C(CF_CROWPOS), C(CF_CCOLPOS), a.crows, a.ccols, hr);
                const double* q = a.af_base + (int64_t)a.crows * a.ccols * ((int64_t)dabs * 24);     // the day's first step: uniform
                // CLDS: the lane's place in the day's staged buffer (column slot, the first of its two coarse rows) and its weight
                // along the rows — made again every day from the tile's image rather than carried (three registers short)
                const double* cu = s_cu;
                int cu_r1 = 0;
                double cu_wy = 0.0;
                if (CLDS) {
                    const double rp = C(CF_CROWPOS);
                    const int sl = cl >= s_ci[8] ? 1 : 0;
                    const double fr = floor(rp);
                    const int r0 = (int)fr;
                    cu_wy = rp - fr;
                    cu = s_cu + (run & 1) * (2 * CU_SLOT) + (sl * CU_ROWS + (r0 - s_ci[4 * sl])) * (CU_F * 24) + hr;
                    cu_r1 = r0 + 1 < a.crows ? CU_F * 24 : 0;
                }
                auto at = [&](int f) {
                    if (CLDS) return CoarseTap::mix(cu[f * 24], cu[f * 24 + cu_r1], cu_wy);
                    return tap(q + (int64_t)f * a.af_stride);
                };
                double tc = at(TF_TC);
                const double rh = at(TF_ES), wu = at(TF_U2), wv = at(TF_EA);
                const double es = satvap_r(tc, MK), ea = es * rh / 100.0;
                const double pk = at(TF_PK) * C(CF_PKFAC);
                tv.v[TF_ES] = es; tv.v[TF_EA] = ea; tv.v[TF_TDEW] = dewpoint_r(ea, tc, MK);   // from the uncorrected tc
                if (a.altcorrect == 1) tc += 0.005 * C(CF_ELEVD);
                if (a.altcorrect == 2) tc += lapserate_r(tc, ea, pk) * C(CF_ELEVD);
                tv.v[TF_TC] = tc;
                tv.v[TF_PK] = pk; tv.v[TF_RSW] = at(TF_RSW); tv.v[TF_RDIF] = at(TF_RDIF); tv.v[TF_RLW] = at(TF_RLW);
                const double s2 = wu * wu + wv * wv;
                tv.v[TF_U2] = s2 > 0.0 ? fsqrt(s2) : 0.0;
                tv.v[TF_SOILMP] = at(TF_SOILMP); tv.v[TF_UMU] = at(TF_UMU);
                if (a.need_pass2) { p2gp = at(TF_GP); p2mugp = at(TF_MUGP); p2dtrp = at(TF_DTRP); p2kp = at(TF_KP); }
            } else {
                // the 15 series, slots TF_TC .. TF_DTRP: ten raw inputs and umu feed pass 1; Gp, kp, muGp, dtrp feed GFAC below
                unsigned pb = posb;
                asm("" : "+v"(pb));
                for (int f = 0; f < 10; ++f) tv.v[f] = force(f, pb);
                tv.v[TF_UMU] = force(TF_UMU, pb);
                if (a.need_pass2) {
                    unsigned pc = posb;
                    asm("" : "+v"(pc));
                    p2gp = force(TF_GP, pc); p2mugp = force(TF_MUGP, pc);
                    p2dtrp = force(TF_DTRP, pc); p2kp = force(TF_KP, pc);
                }
            }
            DateRow dr{a.dt[4 * kabs + 0], a.dt[4 * kabs + 1], a.dt[4 * kabs + 2], a.dt[4 * kabs + 3]};
            derive_time_af(tv, dr, C(CF_SINLAT), C(CF_COSLAT), C(CF_COSB), C(CF_SINB), a.windex[kabs], MK);
            // (a zero or infinite divisor is a forcing value like any other here: IEEE division where the clamps are the
            // reference's — the fast reciprocal makes 0 * inf of it; the fast variant's check below sends such a lane there)
            tv.v[TF_GFAC] = F ? fdiv(p2gp * p2mugp, p2dtrp * p2kp) : (p2gp * p2mugp) / (p2dtrp * p2kp);
            if (F) {
                // a NaN or a zero denominator among the four shows in GFAC, an infinite one in their sum
                cn.watch(tv.v[TF_GFAC]);
                cn.watch((p2gp + p2mugp) + (p2dtrp + p2kp));
                // Array forcing has no per-step table the host could classify (kStepIrregular): every lane checks its own
                // forcing here, with derive_time's conditions — the values read are finite, the signs the fast clamps rely on
                // hold — and a failing lane trips the canary, so that k_solve_fix redoes its tile with the reference's clamps.
                for (int f = 0; f < 10; ++f) cn.watch(tv.v[f]);
                cn.watch(tv.v[TF_UMU]);
                cn.watch(tv.v[TF_RBEAM]);
                cn.watch(dTmx);                 // from the cell's mxtc, the maximum of its temperature series
                if (!(tv.v[TF_DE] > 0.0 && tv.v[TF_GHRRAD] > 0.0 && tv.v[TF_LAPK] > 0.0 && tv.v[TF_PK] > 0.0)) cn.trip();
                if (!(fabs(tv.v[TF_TC]) < 150.0 && fabs(tv.v[TF_TDEW]) < 150.0)) cn.trip();     // satvap_f's bounded exp
            }
        }
        TimeLds TL{s_time + (AF ? 0 : (run % 3) * (TF_COUNT * 24)) + hr};
        TimeReg TR{&tv};
        SoilLds<CPB> SL{s_soil + (SS ? (run % 3) * (SD_COUNT * CPB) + cl : 0)};

        Carry cy;
        Pass1Out p1;
        if (valid) {
            if (AF) pass1<F, false>(C, TR, SL, g, flags, dTcap, cy, p1, MK, cn);
            else pass1<F, SS>(C, TL, SL, g, flags, dTcap, cy, p1, MK, cn);
            if (!PRE) {
                // every hour lane folds its values into the cell's slots (the 21-cell lane map combines a wave's three hours
                // of a cell through the crossbar first, below); `if (m < x) m = x` ignores a NaN x, so does the LDS unit
                double (*ext)[CPB] = s_ext[run % 3];
                __hip_atomic_fetch_max(&ext[0][cl], p1.Tg0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(&ext[1][cl], p1.Tg0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&ext[2][cl], p1.absRnet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            put(3, cy.soilm);     // soilm      cpp:2227
            put(4, p1.uz);        // windspeed  cpp:2253
            put(5, cy.Rbdown);    // Rdirdown   cpp:2242
            put(6, cy.Rddown);    // Rdifdown   cpp:2243
            put(8, p1.Rdup);      // Rswup      cpp:2244
        } else {
            // NA cells, cells past the raster's end and a block's padding lanes: the ring's blocks are written whole
            put(3, NA);
            put(4, NA);
            put(5, NA);
            put(6, NA);
            put(8, NA);
        }
        if (PRE) {
            // The three hour lanes of a cell inside this wave (lanes l, l+16, l+32, or 48+j, 48+j+5, 48+j+10) first
            // combine their values through the crossbar, and one of them stores the wave's partial extremes:
            // 8 partials per cell instead of 24 values for every lane to walk after the barrier.  max / min with
            // NaN-ignoring v_max_f64 / v_min_f64 are order-independent, so this equals the hour-ordered scan.
            const int l = tid & 63;
            int q1, q2;
            if (l < 48) { q1 = l + 16; q1 -= q1 >= 48 ? 48 : 0; q2 = l + 32; q2 -= q2 >= 48 ? 48 : 0; }
            else if (l < 63) { const int j = l - 48; q1 = 48 + (j + 5) % 15; q2 = 48 + (j + 10) % 15; }
            else { q1 = l; q2 = l; }
            const double xt = valid ? p1.Tg0 : 0.0, xr = valid ? p1.absRnet : 0.0;
            const double t1 = __shfl(xt, q1), t2 = __shfl(xt, q2), r1 = __shfl(xr, q1), r2 = __shfl(xr, q2);
            double tmx3 = xt, tmn3 = xt, rmx3 = xr;
            asm("v_max_f64 %0, %0, %1" : "+v"(tmx3) : "v"(t1));
            asm("v_max_f64 %0, %0, %1" : "+v"(tmx3) : "v"(t2));
            asm("v_min_f64 %0, %0, %1" : "+v"(tmn3) : "v"(t1));
            asm("v_min_f64 %0, %0, %1" : "+v"(tmn3) : "v"(t2));
            asm("v_max_f64 %0, %0, %1" : "+v"(rmx3) : "v"(r1));
            asm("v_max_f64 %0, %0, %1" : "+v"(rmx3) : "v"(r2));
            // `if (m < x) m = x` ignores a NaN x; so does the LDS unit's float max / min (the slot is never NaN: it starts finite)
            if (valid && (l < 16 || (l >= 48 && l < 53))) {
                double (*ext)[CPB] = s_ext[run % 3];
                __hip_atomic_fetch_max(&ext[0][cl], tmx3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(&ext[1][cl], tmn3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&ext[2][cl], rmx3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (CLDS && dl + 1 < ndays) stage_commit(dabs + 1, (run + 1) & 1, cs0, cs1);
        if (stage) {
            double* dst = s_time + ((run + 1) % 3) * (TF_COUNT * 24);
#pragma unroll
            for (int i = 0; i < TPER; ++i) {
                int q = tid + i * NT;
                if (q < TF_COUNT * 24) dst[q] = pre[i];
            }
        }
        __syncthreads();
        if (tid < 3 * CPB)      // reset the buffer of the day after next (see s_ext)
            s_ext[(run + 2) % 3][tid / CPB][tid % CPB] = tid < CPB ? -999.0 : tid < 2 * CPB ? 999.0 : -999.9;
        if (SS) {
            // Every wave is past the day before now, so the soil ring slot of the day after next (= that of the day before)
            // is free; whoever fills it reaches the NEXT barrier before any wave starts that day.  The waves take turns.  A
            // change of layer on the way restarts the ring above, so a slot filled with the wrong layer's constants is never
            // read.
            if ((tid >> 6) == (run & 7) % (NT / 64) && dl + 2 < ndays) produce_soil(day0 + dl + 2, run + 2);
        }
        if (valid && a.need_pass2) {
            // day reductions with the reference's comparisons, cpp:2196-2198, 2256-2263.  `if (Rmx < rv) Rmx = rv` with a
            // finite start value ignores a NaN rv, exactly what v_max_f64 does (the accumulator is never NaN, rv is never a
            // signalling NaN: it was just computed): one VALU instruction instead of a compare and two 32-bit selects
            const double tmx = s_ext[run % 3][0][cl], tmn = s_ext[run % 3][1][cl], Rmx = s_ext[run % 3][2][cl];
            const double dtr = tmx - tmn;
            Pass2Out p2{};
            if (AF) derive_time_af_pass2(tv);
            if (AF) pass2<F, false>(C, TR, SL, g, flags, dTcap, cy, dtr, Rmx, need_tv, p2, MK, cn);
            else pass2<F, SS>(C, TL, SL, g, flags, dTcap, cy, dtr, Rmx, need_tv, p2, MK, cn);
            if (BG) {
                a.tgser[c + N * ((int64_t)dabs * 24 + hr)] = p2.Tg;
                s_dd[hr * CPB + cl] = p2.DD;
            } else {
                const bool pos_h = g.reqhgt > 0.0;
                put(0, pos_h ? p2.Tz : p2.Tg);        // cpp:2292-2297
                put(1, pos_h ? p2.tleaf : NA);        // cpp:2300-2303
                put(2, pos_h ? p2.rh : NA);
                put(7, p2.lwdn);                      // cpp:2298
                put(9, p2.lwup);                      // cpp:2299
            }
        } else if (!BG || in_grid) {
            if (!BG) put(0, NA);
            put(1, NA);
            put(2, NA);
            put(7, NA);
            put(9, NA);
        }
        if (BG) {
            // reqhgt < 0 never writes tleaf/relhum/Rlw* (cpp:2272): they stay NA
            if (valid) {
                put(1, NA);
                put(2, NA);
                put(7, NA);
                put(9, NA);
            }
            __syncthreads();
            if (valid && hr == 0) {
                double s = 0.0;
                for (int hh = 0; hh < 24; ++hh) s += s_dd[hh * CPB + cl];
                a.ddsum[c] += s;                                             // cpp:2309-2312
            }
            __syncthreads();
        }
        if (!BG) ring_day += a.out_day_stride;
    }
    if (F) {
        // a watched clamp met a NaN somewhere in this workgroup's tile: k_solve_fix redoes the tile's days of this launch.
        // ONE entry per tile: the waves' ballots meet in an LDS flag (every wave is past the last day's barrier; the flag's
        // own barrier below is reached by all), lane 0 pushes.
        if (__builtin_amdgcn_ballot_w64(cn.tripped()) != 0 && (tid & 63) == 0) s_trip = 1;
        __syncthreads();
        if (tid == 0 && s_trip) {
            const int i = atomicAdd(a.fix_count, 1);
            if (i < a.fix_cap) a.fix_list[i] = (int32_t)tile;
        }
    }
}

// blockIdx -> position in the launch's tile sequence.  Workgroups are dealt round-robin to the 8 XCDs (b and b+8
// share one, each XCD has its own L2).  The 168-B row segments a tile reads from the [field][cell] constant tables share
// their boundary cache lines with the neighbouring tiles, so consecutive tiles are given to the SAME XCD.  Speed only.
__device__ __forceinline__ int64_t tile_position(int64_t ntiles) {
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t pos = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (pos >= ntiles || (int64_t)(blockIdx.x >> 3) >= per_xcd) return -1;
    return pos;
}

// array forcing keeps ~17 more doubles live per lane (forcing values instead of an LDS table):
// it is built for 3 waves/SIMD (168 VGPRs, no scratch) and run with 32-cell workgroups
template <int CPB, int AF, bool BG, bool F, bool SSREQ>
__global__ __launch_bounds__(solve_threads(CPB), AF ? MCF_AF_WAVES : MCF_WAVES_PER_EU) void k_solve(SolveArgs a) {
    // every other resident workgroup shifts its wave-to-hour assignment by six hours
    const int rot = (int)((blockIdx.x >> 8) & 1);
    const int64_t pos = tile_position(a.ntiles_launch);
    if (pos < 0) return;
    const int64_t tile = a.tile_list ? (int64_t)a.tile_list[pos] : pos;
    solve_tile<CPB, AF, BG, F, SSREQ>(a, tile, a.day0, a.ndays, rot);
}

// Redoes, with the reference's compare-and-select clamps, the tiles in which a fast workgroup's canary tripped.  Launched
// behind every fast launch with a fixed small grid; with an empty list (the normal case) every workgroup leaves at once.
// An overflowing list means "everything": all tiles, all days of the launch.
// (built for TWO waves per SIMD where a workgroup's waves allow it — 8-wave tiles: the reference-form clamps' compare-and-select code
// needs more registers than the fast kernels, the list is normally empty, and occupancy is of no interest here; 12-wave tiles need
// three per SIMD to be resident at all)
template <int CPB, int AF>
__global__ __launch_bounds__(solve_threads(CPB), solve_threads(CPB) <= 512 ? 2 : solve_threads(CPB) <= 768 ? 3 : MCF_WAVES_PER_EU) void k_solve_fix(SolveArgs a) {
    const int n = *a.fix_count;
    if (n <= 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.fix_count + 1, n);     // running total, for mcf_plan_dispatch_stats
    if (n > a.fix_cap) {
        const int64_t ntiles = (a.N + CPB - 1) / CPB;
        for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
            __syncthreads();
            solve_tile<CPB, AF, false, false, false>(a, t, a.day0, a.ndays, 0);
        }
        return;
    }
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        __syncthreads();
        solve_tile<CPB, AF, false, false, false>(a, (int64_t)a.fix_list[i], a.day0, a.ndays, 0);
    }
}

// per tile of `cpb` cells: 1 if every valid cell is FL_REGULAR in every vegetation layer
__global__ void k_tile_regular(const double* __restrict__ cellc, int64_t N, int layers, int cpb, int64_t ntiles,
                               uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    bool ok = true;
    for (int l = 0; l < layers; ++l) {
        const double* fl = cellc + ((int64_t)l * ntiles + t) * tile_image_doubles_dev(cpb) + CF_FLAGS * cpb;
        for (int64_t c = t * cpb; c < (t + 1) * cpb && c < N; ++c) {
            const int f = (int)fl[c - t * cpb];
            if ((f & FL_VALID) && !(f & FL_REGULAR)) ok = false;
        }
    }
    out[t] = ok ? 1 : 0;
}

// ---- a launch over a SUBSET of the cells (mcf_kernels.h launch_cells_*; mcf_api.hip mcf_plan_run_days_cells) ---------------------
__global__ __launch_bounds__(256) void k_cells_class(const uint8_t* __restrict__ need, const uint8_t* __restrict__ tile_regular, int64_t N,
                                                     int cpb, uint8_t* __restrict__ cls, int32_t* __restrict__ blockcnt) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int k = 0;
    if (c < N && need[c]) k = (!tile_regular || tile_regular[c / cpb]) ? 1 : 2;
    if (c < N) cls[c] = (uint8_t)k;
    __shared__ int s_n[2];
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b1 = __ballot(k == 1), b2 = __ballot(k == 2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_n[0], __popcll(b1)); atomicAdd(&s_n[1], __popcll(b2)); }
    __syncthreads();
    if (threadIdx.x < 2) blockcnt[2 * (int64_t)blockIdx.x + threadIdx.x] = s_n[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_cells_place(const uint8_t* __restrict__ cls, int64_t N, const int32_t* __restrict__ blockoff,
                                                     int32_t* __restrict__ list) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = c < N ? cls[c] : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int s_w[4][2];
    const uint64_t b1 = __ballot(k == 1), b2 = __ballot(k == 2);
    if (lane == 0) { s_w[wave][0] = __popcll(b1); s_w[wave][1] = __popcll(b2); }
    __syncthreads();
    if (k) {
        int r = __popcll((k == 1 ? b1 : b2) & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) r += s_w[w][k - 1];
        list[(int64_t)blockoff[2 * (int64_t)blockIdx.x + k - 1] + r] = (int32_t)c;
    }
}
__global__ __launch_bounds__(256) void k_gather_image(const int32_t* __restrict__ list, int64_t ntiles_sub, const double* __restrict__ src,
                                                      int64_t ntiles_src, int layers, int cpb, double* __restrict__ dst) {
    const int64_t IMG = tile_image_doubles_dev(cpb);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)layers * ntiles_sub * IMG) return;
    const int64_t l = i / (ntiles_sub * IMG), rem = i - l * (ntiles_sub * IMG), t = rem / IMG;
    const int e = (int)(rem - t * IMG), r = e / cpb, j = e - r * cpb;
    double v = 0.0;
    if (r < CF_COUNT + kCellDirs) {
        const int64_t c = list[t * cpb + j];
        if (c >= 0) v = src[(l * ntiles_src + c / cpb) * IMG + (int64_t)r * cpb + c % cpb];
    }
    dst[i] = v;
}
// blockIdx.x = sub-tile, blockIdx.y = day: the day's values of every variable, read in the sub-ring's order
__global__ __launch_bounds__(256) void k_scatter_cells(const int32_t* __restrict__ list, const double* __restrict__ sub,
                                                       int64_t sub_tile_stride, double* __restrict__ ring, int64_t tile_stride,
                                                       int64_t day_doubles, int cpb) {
    const int64_t t = blockIdx.x;
    const int d = blockIdx.y, blk = ring_block_doubles(cpb);
    const double* s = sub + t * sub_tile_stride + (int64_t)d * day_doubles;
    for (int e = threadIdx.x; e < (int)day_doubles; e += 256) {
        const int v = e / blk, pos = e - v * blk;
        int j, h;
        if (!ring_unpos(cpb, pos, &j, &h)) continue;
        const int64_t c = list[t * cpb + j];
        if (c < 0) continue;
        ring[(c / cpb) * tile_stride + (int64_t)d * day_doubles + (int64_t)v * blk + ring_pos(cpb, (int)(c % cpb), h)] = s[e];
    }
}

// ------------------------------------------------------------------------------------
// Tbelowgroundv, cpp:1474-1539; maCpp cpp:561-572; manCpp cpp:597-627; one lane per cell.
// tg / tz are [N, tsteps] (cell fastest).  scratch is [N, 2*ndays].
__device__ inline double series(const double* p, int64_t N, int64_t c, int k, bool per_cell) {
    return per_cell ? p[c + N * k] : p[k];
}
__global__ __launch_bounds__(64) void k_belowground(BelowArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = a.N;
    if (c >= N) return;
    const int m = a.tsteps;
    const double* x = a.tg + c;
    double* z = a.tz + c;
    if (isnan(a.cellflag_hgt[c])) {
        for (int i = 0; i < m; ++i) z[N * i] = na_real();
        return;
    }
    const double meanD = a.ddsum[c] / (double)m;                           // cpp:2313
    const double nb = -118.35 * a.reqhgt / meanD;
    const int n = (int)round(nb);
    if (a.complete) {
        if (n < m) {
            if (n <= 48) {                                                   // cpp:600-602
                for (int i = 0; i < m; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += x[N * ((i - j + m) % m)];
                    z[N * i] = s / n;
                }
            } else {                                                         // cpp:604-625
                const int nd = m / 24;
                double* d = a.scratch + c;             // [nd]
                double* y = a.scratch + c + N * nd;    // [nd]
                for (int i = 0; i < nd; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < 24; ++j) s += x[N * (i * 24 + j)];
                    d[N * i] = s / 24.0;
                }
                const int n2 = n / 24;
                for (int i = 0; i < nd; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < n2; ++j) s += d[N * ((i - j + nd) % nd)];
                    y[N * i] = s / n2;
                }
                for (int i = 0; i < m; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < 24; ++j) {
                        int q = (i - j + m) % m;
                        s += (q < nd * 24) ? y[N * (q / 24)] : 0.0;
                    }
                    z[N * i] = s / 24;
                }
            }
        } else {                                                             // cpp:1487-1492
            double s = 0;
            for (int i = 0; i < m; ++i) s = s + x[N * i];
            double meanT = s / m;
            for (int i = 0; i < m; ++i) z[N * i] = meanT;
        }
        return;
    }
    // incomplete time sequence, cpp:1495-1536
    for (int i = 0; i < m; ++i) z[N * i] = x[N * i];
    const int nd = m / 24;
    const bool pc = a.per_cell_pointm != 0;
    const bool blend_day = (nb > 1.0 && nb <= 24.0);
    const bool blend_year = nb > 24.0;
    if (!blend_day && !blend_year) return;
    for (int dI = 0; dI < nd; ++dI) {
        double gmx = x[N * (dI * 24)], gmn = gmx, gsum = 0.0;
        double pmx = series(a.Tgp, N, c, dI * 24, pc), pmn = pmx, psum = 0.0, bsum = 0.0;
        for (int j = 0; j < 24; ++j) {
            double gv = x[N * (dI * 24 + j)], pv = series(a.Tgp, N, c, dI * 24 + j, pc);
            if (j > 0) {
                gmx = fmax(gmx, gv); gmn = fmin(gmn, gv);
                pmx = fmax(pmx, pv); pmn = fmin(pmn, pv);
            }
            gsum += gv; psum += pv;
            bsum += series(a.Tbp, N, c, dI * 24 + j, pc);
        }
        double gme = gsum / 24, pme = psum / 24, Tbpd = bsum / 24;
        double rat = (gmx - gmn) / (pmx - pmn);
        double dif = gme - pme;
        for (int j = 0; j < 24; ++j) {
            int i = dI * 24 + j;
            double Tzd = rat * (series(a.Tbp, N, c, i, pc) - Tbpd) + Tbpd + dif;
            if (blend_day) {
                double w1 = 1.0 / nb, w2 = nb / 24.0;
                double wgt = w1 / (w1 + w2);
                z[N * i] = wgt * x[N * i] + (1 - wgt) * Tzd;
            }
            if (blend_year) {
                if (nb < a.hiy) {
                    double w1 = 24.0 / nb, w2 = nb / a.hiy;
                    double wgt = w1 / (w1 + w2);
                    z[N * i] = wgt * Tzd + (1 - wgt) * a.mat;
                } else {
                    z[N * i] = a.mat;
                }
            }
        }
    }
    // Steps past the last whole day: the reference indexes its day statistics (vectors of nd*24 elements,
    // cpp:517-519) with them, a read past the end.  With zeros there (what the oracle's zero-filled arrays give)
    // rat = 0/0, so every blend that uses Tzd is NaN; only the `mat` branch has a defined value.
    for (int i = nd * 24; i < m; ++i)
        z[N * i] = (blend_year && !(nb < a.hiy)) ? a.mat : __longlong_as_double(0x7FF8000000000000LL);
}

// ------------------------------------------------------------------------------------
// runbioclimCpp, cpp:3245-3560: per-cell reductions of Tz (or tleaf) and soilm over time; one
// lane per cell, lanes along raster rows (coalesced).  All 19 values are produced; bio7 / bio3
// come from the values, not from possibly unrequested matrices.
// a cell's series in a ring slot: tile and cell resolved once per lane
struct RingCell {
    const double* p;      // linear: the cell's first step; tiled: the tile's first block
    int64_t N, day_stride;
    int cpb, cell;
    __device__ RingCell(const RingView& v, int64_t c) : N(v.N), day_stride(v.day_stride), cpb(v.cpb) {
        if (cpb == 0) { p = v.base + c; cell = 0; }
        else { const uint32_t t = (uint32_t)c / (uint32_t)cpb; cell = (int)((uint32_t)c - t * (uint32_t)cpb); p = v.base + (int64_t)t * v.tile_stride; }
    }
    __device__ __forceinline__ double operator[](int k) const {
        if (cpb == 0) return p[N * k];
        const int d = k / 24;
        return p[(int64_t)d * day_stride + ring_pos(cpb, cell, k - 24 * d)];
    }
};
__device__ inline double quarter_mean(const RingCell& x, const int32_t* q, int nq) {
    double s = 0.0;
    for (int i = 0; i < nq; ++i) s = s + x[q[i]];
    return s / 72.0;                                                    // cpp:3325 (fixed divisor)
}
__global__ __launch_bounds__(64) void k_bioclim(BioclimArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = a.N;
    if (c >= N) return;
    const RingCell tz(a.tz, c), sm(a.soilm, c);
    double* out = a.bio + c;
    const double NA = na_real();
    if (isnan(tz[0])) {                                                 // cpp:3505-3506
        for (int b = 0; b < 19; ++b) out[N * b] = NA;
        return;
    }
    const int T = a.tsteps;
    double bio[19];
    {   // bio1 cpp:3245, bio2 cpp:3256, bio4 cpp:3279
        double s = 0.0, dsum = 0.0, mon[12];
        for (int d = 0; d < 12; ++d) {
            double tmx = -273.15, tmn = 273.15, ms = 0.0;
            for (int h = 0; h < 24; ++h) {
                double v = tz[d * 24 + h];
                s = s + v;
                if (v > tmx) tmx = v;
                if (v < tmn) tmn = v;
                ms = ms + v;
            }
            dsum = dsum + (tmx - tmn);
            mon[d] = ms / 24;
        }
        bio[0] = s / 288.0;
        bio[1] = dsum / 12;
        double mean = 0.0;                                              // calc_std_dev cpp:3227
        for (int d = 0; d < 12; ++d) mean += mon[d];
        mean /= 12;
        double ss = 0.0;
        for (int d = 0; d < 12; ++d) ss += (mon[d] - mean) * (mon[d] - mean);
        bio[3] = sqrt(ss / 11) * 100.0;
    }
    {   // bio5 cpp:3297, bio6 cpp:3307
        double tmx = -273.15, tmn = 273.15;
        for (int i = 288; i < 312; ++i) { double v = tz[i]; if (v > tmx) tmx = v; }
        for (int i = 312; i < 336; ++i) { double v = tz[i]; if (v < tmn) tmn = v; }
        bio[4] = tmx;
        bio[5] = tmn;
    }
    bio[7] = quarter_mean(tz, a.wetq, a.nwet);
    bio[8] = quarter_mean(tz, a.dryq, a.ndry);
    bio[9] = quarter_mean(tz, a.hotq, a.nhot);
    bio[10] = quarter_mean(tz, a.colq, a.ncol);
    {   // bio12..bio15 cpp:3361-3404
        double me = 0.0;
        for (int i = 0; i < 288; ++i) me = me + sm[i];
        me = me / 288.0;
        double mx = 0.0, mn = 1.0, all = 0.0;
        for (int i = 0; i < T; ++i) {
            double v = sm[i];
            if (v > mx) mx = v;
            if (v < mn) mn = v;
            all += v;
        }
        double mean = all / T, ss = 0.0;
        for (int i = 0; i < T; ++i) { double dlt = sm[i] - mean; ss += dlt * dlt; }
        double sd = T <= 1 ? NA : sqrt(ss / (T - 1));
        bio[11] = me;
        bio[12] = mx;
        bio[13] = mn;
        bio[14] = me / sd;                                              // cpp:3402 (sic: mean / sd)
    }
    bio[15] = quarter_mean(sm, a.wetq, a.nwet);
    bio[16] = quarter_mean(sm, a.dryq, a.ndry);
    bio[17] = quarter_mean(sm, a.hotq, a.nhot);
    bio[18] = quarter_mean(sm, a.colq, a.ncol);
    bio[6] = bio[4] - bio[5];                                           // cpp:3533
    bio[2] = bio[1] / bio[6];                                           // cpp:3534
    for (int b = 0; b < 19; ++b) out[N * b] = bio[b];
}
void launch_bioclim(const BioclimArgs& a, hipStream_t s) {
    if (a.N <= 0) return;
    hipLaunchKernelGGL(k_bioclim, dim3((unsigned)((a.N + 63) / 64)), dim3(64), 0, s, a);
}

// ---- the streamed form (mcf_kernels.h BioAccArgs): k_bioclim's loops cut at chunk boundaries, every accumulation in the same
// order on the same operands — the nineteen matrices are bit for bit k_bioclim's (tests/test_bioclim_gpu.py).
__global__ __launch_bounds__(64) void k_bioclim_acc(BioAccArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = a.N;
    if (c >= N) return;
    const RingCell tz(a.tz, c), sm(a.soilm, c);
    double* st = a.state + c;
    auto S = [&](int row) -> double& { return st[N * row]; };
    if (a.day0 == 0) {
        S(0) = tz[0];
        S(1) = 0.0; S(2) = 0.0;
        S(15) = -273.15; S(16) = 273.15;
        for (int r = 17; r <= 21; ++r) S(r) = 0.0;
        S(22) = 0.0; S(23) = 1.0; S(24) = 0.0;
        for (int r = 25; r <= 28; ++r) S(r) = 0.0;
    }
    if (isnan(S(0))) return;                                            // cpp:3505-3506: every value NA (k_bioclim_fin)
    double s1 = S(1), dsum = S(2), b5 = S(15), b6 = S(16), me = S(21), mx = S(22), mn = S(23), all = S(24);
    for (int dl = 0; dl < a.ndays; ++dl) {
        const int d = a.day0 + dl, k0 = dl * 24;
        if (d < 12) {                                                   // bio1 cpp:3245, bio2 cpp:3256, bio4 cpp:3279
            double tmx = -273.15, tmn = 273.15, ms = 0.0;
            for (int h = 0; h < 24; ++h) {
                const double v = tz[k0 + h];
                s1 = s1 + v;
                if (v > tmx) tmx = v;
                if (v < tmn) tmn = v;
                ms = ms + v;
            }
            dsum = dsum + (tmx - tmn);
            S(3 + d) = ms / 24;
            for (int h = 0; h < 24; ++h) me = me + sm[k0 + h];          // bio12 cpp:3361
        } else if (d == 12) {                                           // bio5 cpp:3297
            for (int h = 0; h < 24; ++h) { const double v = tz[k0 + h]; if (v > b5) b5 = v; }
        } else if (d == 13) {                                           // bio6 cpp:3307
            for (int h = 0; h < 24; ++h) { const double v = tz[k0 + h]; if (v < b6) b6 = v; }
        }
        for (int h = 0; h < 24; ++h) {                                  // bio13, bio14, bio15's mean cpp:3371-3398
            const double v = sm[k0 + h];
            if (v > mx) mx = v;
            if (v < mn) mn = v;
            all += v;
        }
    }
    S(1) = s1; S(2) = dsum; S(15) = b5; S(16) = b6; S(21) = me; S(22) = mx; S(23) = mn; S(24) = all;
    const int kbase = a.day0 * 24;
    for (int qi = 0; qi < 4; ++qi) {                                    // cpp:3316-3358, 3406-3448
        if (a.qlo[qi] >= a.qhi[qi]) continue;
        double st_ = S(17 + qi), ss_ = S(25 + qi);
        for (int i = a.qlo[qi]; i < a.qhi[qi]; ++i) {
            const int k = a.q[qi][i] - kbase;
            st_ = st_ + tz[k];
            ss_ = ss_ + sm[k];
        }
        S(17 + qi) = st_; S(25 + qi) = ss_;
    }
}
void launch_bioclim_acc(const BioAccArgs& a, hipStream_t s) {
    if (a.N <= 0) return;
    hipLaunchKernelGGL(k_bioclim_acc, dim3((unsigned)((a.N + 63) / 64)), dim3(64), 0, s, a);
}
__global__ __launch_bounds__(64) void k_bioclim_fin(BioFinArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = a.N;
    if (c >= N) return;
    const double* st = a.state + c;
    auto S = [&](int row) { return st[N * row]; };
    double* out = a.bio + c;
    const double NA = na_real();
    if (isnan(S(0))) {
        for (int b = 0; b < 19; ++b) out[N * b] = NA;
        return;
    }
    const int T = a.tsteps;
    double bio[19];
    bio[0] = S(1) / 288.0;
    bio[1] = S(2) / 12;
    {
        double mean = 0.0;                                              // calc_std_dev cpp:3227
        for (int d = 0; d < 12; ++d) mean += S(3 + d);
        mean /= 12;
        double ss = 0.0;
        for (int d = 0; d < 12; ++d) ss += (S(3 + d) - mean) * (S(3 + d) - mean);
        bio[3] = sqrt(ss / 11) * 100.0;
    }
    bio[4] = S(15);
    bio[5] = S(16);
    for (int qi = 0; qi < 4; ++qi) { bio[7 + qi] = S(17 + qi) / 72.0; bio[15 + qi] = S(25 + qi) / 72.0; }     // cpp:3325
    {
        const double me = S(21) / 288.0;
        const double mean = S(24) / T;
        // second pass over the soil moisture series: the solver's value of every step, made again (BioFinArgs)
        const uint32_t cc = (uint32_t)c, tile = cc / (uint32_t)a.cpb, cl = cc - tile * (uint32_t)a.cpb;
        const int64_t IMG = tile_image_doubles_dev(a.cpb);
        double ss = 0.0;
        int layer = -2;
        double smin = 0.0, invrge = 0.0, eta = 0.0, rge = 0.0;
        bool valid = false;
        Canary cn;
        for (int d = 0; d < T / 24; ++d) {
            const int l = a.daylayer ? a.daylayer[d] : 0;
            if (l != layer) {
                layer = l;
                if (l >= 0) {
                    const double* img = a.cellc + ((int64_t)l * a.ntiles_total + tile) * IMG + cl;
                    smin = img[CF_SMIN * a.cpb]; invrge = img[CF_INVRGE * a.cpb]; eta = img[CF_ETA * a.cpb]; rge = img[CF_RGE * a.cpb];
                    valid = ((int)img[CF_FLAGS * a.cpb] & FL_VALID) != 0;
                }
            }
            const double* sp = a.tt + ((int64_t)d * TF_COUNT + TF_SOILMP) * 24;
            for (int h = 0; h < 24; ++h) {
                const double v = (l >= 0 && valid) ? soil_spread<false>(sp[h], smin, invrge, eta, rge, cn) : NA;
                const double dlt = v - mean;
                ss += dlt * dlt;
            }
        }
        const double sd = T <= 1 ? NA : sqrt(ss / (T - 1));
        bio[11] = me;
        bio[12] = S(22);
        bio[13] = S(23);
        bio[14] = me / sd;                                              // cpp:3402 (sic: mean / sd)
    }
    bio[6] = bio[4] - bio[5];                                           // cpp:3533
    bio[2] = bio[1] / bio[6];                                           // cpp:3534
    for (int b = 0; b < 19; ++b) out[N * b] = bio[b];
}
void launch_bioclim_fin(const BioFinArgs& a, hipStream_t s) {
    if (a.N <= 0) return;
    hipLaunchKernelGGL(k_bioclim_fin, dim3((unsigned)((a.N + 63) / 64)), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------
// Diagnostics: evaluates the lean elementary functions of mcf_device.hpp elementwise.
__global__ void k_selftest_math(int kind, const double* __restrict__ x, const double* __restrict__ y,
                                double* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    if (!live) i = 0;
    double a = x[i], b = y ? y[i] : 0.0, r;
    MathK K;
    K.set();
    __shared__ double s_exptab[256];      // the route k_solve takes
    K.use_table(s_exptab, (int)threadIdx.x);
    __shared__ __attribute__((aligned(16))) double s_logtab[512];
    K.use_log_table(s_logtab, (int)threadIdx.x, (int)blockDim.x);
    K.pin(true, true);     // as k_solve's vector-forcing kernels: the VGPR-resident constants the bounded exp needs
    __syncthreads();
    switch (kind) {
        case 0: r = fexp(a, K); break;
        case 1: r = flog(a, K); break;
        case 2: r = fdiv(a, b); break;
        case 3: r = fsqrt(a); break;
        case 4: r = frcp(a); break;
        case 5: r = satvap(a, K); break;
        case 7: r = fexp_b<true>(a, K); break;        // |x| < 5e6
        case 8: r = satvap_f<true>(a, K); break;      // the fast-clamp kernels' satvap (wave-uniform constants, bounded exp)
        case 9: r = fexp_s<true>(a, K); break;        // one-fma reduction: relative error grows with |x| 2^-54
        case 10: r = frcp_m(a); break;                // 46 bits
        case 11: r = fsqrt_m(a); break;               // 46 bits
        case 12: { double ra, rb; frcp2_m(a, b, ra, rb); r = ra + rb; break; }
        case 13: { double ra, rb; frcp2(a, b, ra, rb); r = ra + rb; break; }
        default: r = powxy(a, b, K); break;
    }
    if (live) out[i] = r;
}
void launch_selftest_math(int kind, const double* x, const double* y, double* out, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kind, x, y, out, n);
}

// ------------------------------------------------------------------------------------
// writetonc's array conversion `atonc` (R/dataprep.R:1064-1069) as an on-device sink: per time step
// transpose [rows, cols] -> [cols, rows] (aperm(a, c(2,1,3)): east becomes the fastest index),
// round(a * rd) half-to-even, as.integer (NA -> NA_integer_ = INT_MIN, which ncvar_put writes as the
// file's missval).  32 x 32 tiles through LDS so that both the fp64 loads (along rows) and the int32
// stores (along cols) are coalesced; 12 B of HBM traffic per element.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_transpose(RingView src, int64_t step0, int64_t rows, int64_t cols,
                                                        double scale, int32_t* __restrict__ dst) {
    __shared__ int32_t tile[32][33];
    const int64_t N = rows * cols;
    const int64_t step = step0 + blockIdx.z;
    int32_t* out = dst + (int64_t)blockIdx.z * N;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + tx, c = c0 + j;
        int32_t v = INT32_MIN;
        if (r < rows && c < cols) {
            const double x = rint(src.at(r + rows * c, step) * scale);
            if (x > -2147483648.0 && x < 2147483648.0) v = (int32_t)x;   // NaN and out-of-range -> NA_integer_
        }
        tile[j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + tx, r = r0 + j;
        if (r < rows && c < cols) out[c + cols * r] = tile[tx][j];
    }
}

// ------------------------------------------------------------------------------------
// NetCDF record sink (mcf_ncfile.hpp): the same transpose + round as k_pack_transpose for up to MCF_NOUT variables
// at once, written where the classic-format file wants them — record `step` = [8 B time | var0[rows][cols] | var1 …],
// big-endian int32, NA / out-of-range -> the file's missval.  blockIdx.z = step * nv + v.  `fill_only[v]`: the
// variable is defined in the file but never put by writetonc (R/dataprep.R:1163-1167), so it holds missval.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_nc(PackNcArgs a) {
    __shared__ int32_t tile[32][33];
    const int step = blockIdx.z / a.nv, v = blockIdx.z - step * a.nv;
    const int64_t N = a.rows * a.cols;
    int32_t* out = a.dst + (int64_t)step * a.rec_words + 2 + (int64_t)v * N;
    const double scale = a.scale[v];
    const bool fill = a.fill_only[v] != 0;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + tx, c = c0 + j;
        int32_t q = a.missval;
        if (!fill && r < a.rows && c < a.cols) {
            const double x = rint(a.src[v].at(r + a.rows * c, a.step0 + step) * scale);
            if (x > -2147483648.0 && x < 2147483648.0) q = (int32_t)x;
        }
        tile[j][tx] = (int32_t)__builtin_bswap32((uint32_t)q);
    }
    __syncthreads();
#pragma unroll
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + tx, r = r0 + j;
        if (r < a.rows && c < a.cols) out[c + a.cols * r] = tile[tx][j];
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
void launch_pack_nc(const PackNcArgs& a, int64_t nsteps, hipStream_t s) {
    if (nsteps <= 0 || a.nv <= 0) return;
    dim3 grid((unsigned)((a.rows + 31) / 32), (unsigned)((a.cols + 31) / 32), (unsigned)(nsteps * a.nv));
    hipLaunchKernelGGL(k_pack_nc, grid, dim3(256), 0, s, a);
}
void launch_pack_transpose(const RingView& src, int64_t step0, int64_t rows, int64_t cols, int64_t nsteps, double scale,
                           int32_t* dst, hipStream_t s) {
    if (nsteps <= 0) return;
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32), (unsigned)nsteps);
    hipLaunchKernelGGL(k_pack_transpose, grid, dim3(256), 0, s, src, step0, rows, cols, scale, dst);
}
void launch_tile_series(const double* src, int64_t nsteps, const RingView& dst, hipStream_t s) {
    if (nsteps <= 0 || dst.N <= 0) return;
    // gridDim.y limit; a piece is a whole number of days (65520 = 2730 x 24), so that a second piece lands on a day boundary
    // of the tiled ring (mcf_plan_create refuses ring slots of more than 2730 days, so one piece is all there ever is)
    for (int64_t k0 = 0; k0 < nsteps; k0 += 65520) {
        const int64_t n = std::min<int64_t>(65520, nsteps - k0);
        RingView v = dst;
        v.base = dst.base + (k0 / 24) * dst.day_stride;
        dim3 grid((unsigned)((dst.N + 255) / 256), (unsigned)n);
        hipLaunchKernelGGL(k_tile_series, grid, dim3(256), 0, s, src + dst.N * k0, v);
    }
}
void launch_untile(const RingView& src, int64_t step0, int64_t nsteps, double* dst, hipStream_t s) {
    if (nsteps <= 0 || src.N <= 0) return;
    for (int64_t k0 = 0; k0 < nsteps; k0 += 65535) {      // gridDim.y limit
        const int64_t n = std::min<int64_t>(65535, nsteps - k0);
        dim3 grid((unsigned)((src.N + 255) / 256), (unsigned)n);
        hipLaunchKernelGGL(k_untile, grid, dim3(256), 0, s, src, step0 + k0, dst + src.N * k0);
    }
}
void launch_mxtc_coarse(const double* force, int64_t stride, int crows, int ccols, int tsteps, const double* rowpos,
                        const double* colpos, int64_t rows, int64_t N, int altcorrect, const double* elevd,
                        const double* pkfac, double* mx, hipStream_t s) {
    hipLaunchKernelGGL(k_mxtc_coarse, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, force, stride, crows, ccols, tsteps,
                       rowpos, colpos, rows, N, altcorrect, elevd, pkfac, mx);
}
void launch_gather_cells(const RingView& src, int64_t step0, int64_t nsteps, const int64_t* cells, int64_t ncells, double* dst,
                         hipStream_t s) {
    const int64_t n = ncells * nsteps;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_cells, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, step0, nsteps, cells, ncells, dst);
}
void launch_fill(double* p, int64_t n, double v, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}
void launch_twi_partial(const double* twi, int64_t n, double tfact, double* out2, hipStream_t s) {
    hipLaunchKernelGGL(k_twi_partial, dim3(kTwiParts), dim3(256), 0, s, twi, n, tfact, out2);
    hipLaunchKernelGGL(k_twi_finish, dim3(1), dim3(64), 0, s, out2);
}
void launch_cell_setup(const CellSetupArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_cell_setup, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, s, a);
}
void launch_time_setup(const TimeSetupArgs& a, hipStream_t s) {
    if (a.nsteps <= 0) return;
    hipLaunchKernelGGL(k_time_setup, dim3((unsigned)((a.nsteps + 255) / 256)), dim3(256), 0, s, a);
}
void launch_date_setup(const DateSetupArgs& a, hipStream_t s) {
    if (a.nsteps <= 0) return;
    hipLaunchKernelGGL(k_date_setup, dim3((unsigned)((a.nsteps + 255) / 256)), dim3(256), 0, s, a);
}
void launch_mxtc(const double* tc, int64_t N, int nsteps, double* mx, hipStream_t s) {
    hipLaunchKernelGGL(k_mxtc, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, tc, N, nsteps, mx);
}
void launch_belowground(const BelowArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_belowground, dim3((unsigned)((a.N + 63) / 64)), dim3(64), 0, s, a);
}

static dim3 solve_grid(int64_t ntiles) { return dim3((unsigned)(8 * ((ntiles + 7) / 8))); }     // tile_position()
template <int CPB>
static void launch_solve_cpb(SolveArgs a, bool af, bool bg, bool fast, bool ss, hipStream_t s) {
    if (a.ntiles_launch <= 0) {                      // no list: every tile of the raster
        a.ntiles_launch = (a.N + CPB - 1) / CPB;
        a.tile_list = nullptr;
    }
    const dim3 grid = solve_grid(a.ntiles_launch), block(solve_threads(CPB));
    ss = ss && 2 * CPB <= 64;
    if (af) {
        // (42-cell tiles are a vector-forcing geometry: with the forcing values in registers a 16-wave workgroup spills
        // 52-148 bytes per lane; mcf_plan_create gives array forcing its 32-cell tiles instead and nothing is built here)
        if constexpr (CPB != 42) {
            if (bg) hipLaunchKernelGGL((k_solve<CPB, 1, true, false, false>), grid, block, 0, s, a);
            else if (fast) {
                hipLaunchKernelGGL((k_solve<CPB, 1, false, true, false>), grid, block, 0, s, a);
                hipLaunchKernelGGL((k_solve_fix<CPB, 1>), dim3(512), block, 0, s, a);
            } else hipLaunchKernelGGL((k_solve<CPB, 1, false, false, false>), grid, block, 0, s, a);
        }
    } else if (bg) {
        hipLaunchKernelGGL((k_solve<CPB, 0, true, false, false>), grid, block, 0, s, a);
    } else if (fast) {
        if (ss) hipLaunchKernelGGL((k_solve<CPB, 0, false, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_solve<CPB, 0, false, true, false>), grid, block, 0, s, a);
        hipLaunchKernelGGL((k_solve_fix<CPB, 0>), dim3(512), block, 0, s, a);
    } else {
        if (ss) hipLaunchKernelGGL((k_solve<CPB, 0, false, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_solve<CPB, 0, false, false, false>), grid, block, 0, s, a);
    }
}
void launch_cells_class(const uint8_t* need, const uint8_t* tile_regular, int64_t N, int cpb, uint8_t* cls, int32_t* blockcnt,
                        hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(k_cells_class, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, need, tile_regular, N, cpb, cls, blockcnt);
}
void launch_cells_place(const uint8_t* cls, int64_t N, const int32_t* blockoff, int32_t* list, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(k_cells_place, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cls, N, blockoff, list);
}
void launch_gather_image(const int32_t* list, int64_t ntiles_sub, const double* src, int64_t ntiles_src, int layers, int cpb,
                         double* dst, hipStream_t s) {
    const int64_t n = (int64_t)layers * ntiles_sub * tile_image_doubles_dev(cpb);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_image, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, list, ntiles_sub, src, ntiles_src, layers, cpb, dst);
}
void launch_scatter_cells(const int32_t* list, int64_t ntiles_sub, const double* sub, int64_t sub_tile_stride, double* ring,
                          int64_t tile_stride, int64_t day_doubles, int cpb, int ndays, hipStream_t s) {
    if (ntiles_sub <= 0 || ndays <= 0) return;
    hipLaunchKernelGGL(k_scatter_cells, dim3((unsigned)ntiles_sub, (unsigned)ndays), dim3(256), 0, s, list, sub, sub_tile_stride, ring,
                       tile_stride, day_doubles, cpb);
}
void launch_tile_regular(const double* cellc, int64_t N, int layers, int cpb, uint8_t* out, hipStream_t s) {
    const int64_t ntiles = (N + cpb - 1) / cpb;
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_tile_regular, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, cellc, N, layers, cpb, ntiles, out);
}
// coarse array forcing is built for the array-forcing geometry (32 cells per workgroup) only
// lds: every tile touches at most 4 coarse rows per raster column (mcf_plan_create): the taps are staged in LDS
static void launch_solve_coarse(SolveArgs a, bool bg, bool fast, bool lds, hipStream_t s) {
    constexpr int CPB = 32;
    if (a.ntiles_launch <= 0) {
        a.ntiles_launch = (a.N + CPB - 1) / CPB;
        a.tile_list = nullptr;
    }
    const dim3 grid = solve_grid(a.ntiles_launch), block(solve_threads(CPB));
    if (bg) hipLaunchKernelGGL((k_solve<CPB, 2, true, false, false>), grid, block, 0, s, a);
    else if (fast) {
        if (lds) hipLaunchKernelGGL((k_solve<CPB, 2, false, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_solve<CPB, 2, false, true, false>), grid, block, 0, s, a);
        hipLaunchKernelGGL((k_solve_fix<CPB, 2>), dim3(512), block, 0, s, a);
    } else hipLaunchKernelGGL((k_solve<CPB, 2, false, false, false>), grid, block, 0, s, a);
}
int twi_scratch_doubles() { return 2 + 2 * kTwiParts; }
int cell_field_count() { return CF_COUNT; }
int step_irregular_bit() { return kStepIrregular; }
int time_field_count() { return TF_COUNT; }
// mincondCpp (cpp:1321-1328): for a fixed stomatal resistance rs, Hlf and Hf are constants and
// gmin = 0.0463*|Hf|^0.2 * (|Rnet|/leafd)^0.2; returns |Hf|^0.2
double hf_pow02(double rs) {
    double Hlf = 1.09767 * pow(rs, 0.2672778);
    double Hf = -1.0 / (1.0 + exp(2.0 - Hlf));
    return pow(fabs(Hf), 0.2);
}

void launch_solve(const SolveArgs& a, int cells_per_block, bool af, bool bg, bool fast, bool soil_daily, hipStream_t s) {
    if (a.N <= 0 || a.ndays <= 0) return;
    if (a.crows > 0) { launch_solve_coarse(a, bg, fast, soil_daily, s); return; }     // (coarse: the flag says "taps through LDS")
    if (cells_per_block == 32) launch_solve_cpb<32>(a, af, bg, fast, soil_daily, s);
    else if (cells_per_block == 21) launch_solve_cpb<21>(a, af, bg, fast, soil_daily, s);
    else if (cells_per_block == 42) launch_solve_cpb<42>(a, af, bg, fast, soil_daily, s);
    else launch_solve_cpb<16>(a, af, bg, fast, soil_daily, s);
}
int soil_daily_bit() { return kSoilDaily; }
// timing variants (tools/variants/*.patch) report through this hook at plan destruction; the shipped library has nothing to say
void print_variant_stats() {}

}  // namespace mcf
