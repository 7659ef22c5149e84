// mcf_terrain.hip — on-device terrain pre-compute for the grid solver's inputs.
//
// Restates the R-side array arithmetic of the reference's marshaller (R/internal.R, "int:"):
//   hor[,,d]   .horizon(dtm, 15*d)                 int:909-925, called int:1144
//   svfa       0.5*cos(2*tan(mean(atan(hor))))+0.5 int:1147-1148
//   wsa[,,w]   .windsheltera(dtm, zref, s)         int:970-991  (.windcoef int:949-968)
//   slope, aspect  terra::terrain (Horn), NA -> 0  int:1124-1136
// A row block of a larger raster is handled with halo rows above/below (exchanged between
// ranks over RCCL by microclimf_amd/terrain.py); outside the RASTER the reference's zero
// padding applies, exactly as in the single-block case.
//
// terra's aggregate/resample/terrain are not part of the reference repo; their restatement
// here (block means anchored top-left, bilinear between block centres with clamping, Horn
// 8-neighbour) is documented in oracle/terrain_oracle.py — parity unpinned for those.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcf.h"
#include "mcf_terrain.h"

namespace mcf {
int api_fail(int code, const std::string& msg);   // mcf_api.hip
}

namespace {

struct Shift { int dr, dc; double inv_s2, s2; };
struct ShiftTable { Shift s[24][10]; };

struct Geo {
    int64_t rows, cols;        // own block
    int64_t RB;                // rows of the supplied array (with halos)
    int64_t hn;                // halo rows above
    int64_t row0, rows_total;  // global placement
};

// z/res with NA -> 0 inside the supplied array, 0 outside the RASTER (the reference's padding)
__device__ __forceinline__ double zpad(const double* __restrict__ Z, const Geo& g, int64_t grow, int64_t c) {
    if (grow < 0 || grow >= g.rows_total || c < 0 || c >= g.cols) return 0.0;
    int64_t b = grow - (g.row0 - g.hn);
    if (b < 0 || b >= g.RB) return 0.0;   // not supplied (rejected on the host for rows that matter)
    return Z[b + g.RB * c];
}

__global__ void k_prep(const double* __restrict__ dtm, double* __restrict__ Z, int64_t n, double inv_res) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = dtm[i];
    Z[i] = isnan(v) ? 0.0 : v * inv_res;
}

// tan(horizon) in 24 directions + sky view; one lane per own cell (lanes along raster rows)
__global__ __launch_bounds__(256) void k_horizon(const double* __restrict__ Z, Geo g, ShiftTable tab,
                                                 double* __restrict__ hor, double* __restrict__ svf) {
    int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t N = g.rows * g.cols;
    if (cell >= N) return;
    int64_t r = cell % g.rows, c = cell / g.rows;
    int64_t grow = g.row0 + r;
    double z0 = zpad(Z, g, grow, c);
    double satan = 0.0;
    for (int d = 0; d < 24; ++d) {
        double h = 0.0;
        for (int s = 0; s < 10; ++s) {
            const Shift sh = tab.s[d][s];
            double v = (zpad(Z, g, grow + sh.dr, c + sh.dc) - z0) / sh.s2;
            h = fmax(h, v);
        }
        if (hor) hor[(int64_t)d * N + cell] = h;
        satan += atan(h);
    }
    if (svf) {
        double msl = tan(satan / 24.0);
        svf[cell] = 0.5 * cos(2 * msl) + 0.5;
    }
}

// wind-shelter coefficient in 16 directions on the rows [e0, e0+ne) of the raster
__global__ __launch_bounds__(256) void k_windcoef(const double* __restrict__ Z, Geo g, ShiftTable tab16,
                                                  double hgt_over_res, int64_t e0, int64_t ne,
                                                  double* __restrict__ W /* [16][ne*cols] */) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t M = ne * g.cols;
    if (idx >= M) return;
    int64_t r = idx % ne, c = idx / ne;
    int64_t grow = e0 + r;
    double z0 = zpad(Z, g, grow, c);
    for (int d = 0; d < 16; ++d) {
        double h = 0.0;
        for (int s = 0; s < 10; ++s) {
            const Shift sh = tab16.s[d][s];
            double v = (zpad(Z, g, grow + sh.dr, c + sh.dc) - z0) / sh.s2;
            h = fmax(h, v);
            if (h < hgt_over_res / sh.s2) h = 0.0;               // int:964
        }
        W[(int64_t)d * M + idx] = 1 - atan(0.17 * 100 * h) / 1.65;   // int:966
    }
}

// aggregate(fact = s, fun = "mean"): coarse rows [I0, I0+nI), all coarse columns
__global__ __launch_bounds__(256) void k_block_mean(const double* __restrict__ W, Geo g, int s, int64_t e0,
                                                    int64_t ne, int64_t I0, int64_t nI, int64_t nJ,
                                                    double* __restrict__ Cm /* [16][nI*nJ] */) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t K = nI * nJ;
    if (idx >= K * 16) return;
    int d = (int)(idx / K);
    int64_t q = idx % K, Il = q % nI, J = q / nI;
    int64_t ra = (I0 + Il) * s, rb = std::min<int64_t>(ra + s, g.rows_total);
    int64_t ca = J * s, cb = std::min<int64_t>(ca + s, g.cols);
    double sum = 0.0;
    for (int64_t c = ca; c < cb; ++c)
        for (int64_t r = ra; r < rb; ++r) sum += W[(int64_t)d * (ne * g.cols) + (r - e0) + ne * c];
    Cm[idx] = sum / (double)((rb - ra) * (cb - ca));
}

// resample (bilinear between block centres, clamped) + the 16 -> 8 direction blend, int:980-990
__global__ __launch_bounds__(256) void k_resample_blend(const double* __restrict__ Cm, Geo g, int s, int64_t I0,
                                                        int64_t nI, int64_t nJ, int64_t NItot,
                                                        double* __restrict__ wsa /* [8][N] */) {
    int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t N = g.rows * g.cols;
    if (cell >= N) return;
    int64_t r = cell % g.rows, c = cell / g.rows;
    double tr = ((double)(g.row0 + r) - (s - 1) / 2.0) / s, tc = ((double)c - (s - 1) / 2.0) / s;
    int64_t i0 = (int64_t)floor(tr), j0 = (int64_t)floor(tc);
    double wr = tr - (double)i0, wc = tc - (double)j0;
    int64_t ia = std::min<int64_t>(std::max<int64_t>(i0, 0), NItot - 1) - I0;
    int64_t ib = std::min<int64_t>(std::max<int64_t>(i0 + 1, 0), NItot - 1) - I0;
    int64_t ja = std::min<int64_t>(std::max<int64_t>(j0, 0), nJ - 1);
    int64_t jb = std::min<int64_t>(std::max<int64_t>(j0 + 1, 0), nJ - 1);
    double a[16];
    for (int d = 0; d < 16; ++d) {
        const double* C = Cm + (int64_t)d * nI * nJ;
        double top = C[ia + nI * ja] * (1 - wc) + C[ia + nI * jb] * wc;
        double bot = C[ib + nI * ja] * (1 - wc) + C[ib + nI * jb] * wc;
        a[d] = top * (1 - wr) + bot * wr;
    }
    for (int w = 0; w < 8; ++w) {
        double m = 0.5 * a[2 * w] + 0.25 * a[2 * w + 1] + 0.25 * a[(2 * w + 15) & 15];
        wsa[(int64_t)w * N + cell] = m;
    }
}

// Horn (1981) 8-neighbour slope / aspect in degrees on the RAW elevations
__global__ __launch_bounds__(256) void k_slope_aspect(const double* __restrict__ dtm, Geo g, double res,
                                                      double aspect_na, double* __restrict__ slope,
                                                      double* __restrict__ aspect) {
    int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t N = g.rows * g.cols;
    if (cell >= N) return;
    int64_t r = cell % g.rows, c = cell / g.rows;
    int64_t grow = g.row0 + r;
    double sl = 0.0, as = aspect_na;   // terra's NA (raster edge, NA neighbour) -> 0 / aspect_na
    if (grow > 0 && grow < g.rows_total - 1 && c > 0 && c < g.cols - 1) {
        auto z = [&](int64_t dr, int64_t dc) {
            int64_t b = grow + dr - (g.row0 - g.hn);
            return dtm[b + g.RB * (c + dc)];
        };
        double nw = z(-1, -1), n_ = z(-1, 0), ne = z(-1, 1), w_ = z(0, -1), e_ = z(0, 1), sw = z(1, -1),
               s_ = z(1, 0), se = z(1, 1);
        double dzdx = ((ne + 2 * e_ + se) - (nw + 2 * w_ + sw)) / (8 * res);
        double dzdy = ((nw + 2 * n_ + ne) - (sw + 2 * s_ + se)) / (8 * res);
        double v = atan(sqrt(dzdx * dzdx + dzdy * dzdy)) * (180.0 / 3.14159265358979323846);
        double a = atan2(-dzdx, -dzdy) * (180.0 / 3.14159265358979323846);
        a = fmod(a + 360.0, 360.0);
        if (dzdx == 0.0 && dzdy == 0.0) a = 90.0;
        if (!isnan(v)) { sl = v; as = a; }
    }
    if (slope) slope[cell] = sl;
    if (aspect) aspect[cell] = as;
}

// ---- the same two stencils with 32-bit indices and a three-instruction exact quotient (round 4) -----------------------
// k_horizon / k_windcoef above spend ~40 vector instructions per sample: 64-bit bounds tests and index arithmetic, and an
// IEEE division by s^2.  Here a lane keeps (row in the supplied array, global row, column, element index) as 32-bit values
// (terrain_device takes this route when every extent fits), a sample's three range tests are unsigned compares, the load
// address is always a valid one (element 0 when the sample lies outside) and x / s^2 is
//     q = x * y;  r = fma(-q, s^2, x);  q = fma(r, y, q),   y = RN(1 / s^2)
// which is the correctly rounded quotient for these divisors (Markstein's final step; s^2 = 1 .. 100 has no all-ones
// significand) — the bits of the division it replaces.  The s loop is unrolled: s^2 and y are literals.
struct Geo32 { int rows, cols, RB, hn, row0, rows_total; };
struct Shift32 { int dr, dc; };
struct ShiftTable32 { Shift32 s[24][10]; };
struct WindLim { double v[10]; };      // hgt / res / s^2, the windcoef threshold of each step (int:964)

template <bool WIND>
__device__ __forceinline__ double stencil_max(const double* __restrict__ Z, const Geo32& g, const Shift32* __restrict__ sh, int b, int gr,
                                              int c, int idx, double z0, const WindLim& lim) {
    double h = 0.0;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        const int dr = sh[s].dr, dc = sh[s].dc;
        const bool in = (unsigned)(gr + dr) < (unsigned)g.rows_total && (unsigned)(b + dr) < (unsigned)g.RB &&
                        (unsigned)(c + dc) < (unsigned)g.cols;
        const int off = in ? idx + dr + g.RB * dc : 0;       // (element 0 is always there)
        const double z = Z[off];
        const double x = (in ? z : 0.0) - z0;
        const double s2 = (double)((s + 1) * (s + 1)), y = 1.0 / s2;
        double q = x * y;
        const double r = fma(-q, s2, x);
        q = fma(r, y, q);
        h = fmax(h, q);
        if (WIND && h < lim.v[s]) h = 0.0;
    }
    return h;
}
__global__ __launch_bounds__(256) void k_horizon32(const double* __restrict__ Z, Geo32 g, ShiftTable32 tab, WindLim lim,
                                                   double* __restrict__ hor, double* __restrict__ svf) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = g.rows * g.cols;
    if (cell >= N) return;
    const int r = cell % g.rows, c = cell / g.rows;
    const int gr = g.row0 + r, b = r + g.hn, idx = b + g.RB * c;
    const double z0 = Z[idx];
    double satan = 0.0;
    for (int d = 0; d < 24; ++d) {
        const double h = stencil_max<false>(Z, g, tab.s[d], b, gr, c, idx, z0, lim);
        if (hor) hor[(int64_t)d * N + cell] = h;
        satan += atan(h);
    }
    if (svf) {
        const double msl = tan(satan / 24.0);
        svf[cell] = 0.5 * cos(2 * msl) + 0.5;
    }
}
__global__ __launch_bounds__(256) void k_windcoef32(const double* __restrict__ Z, Geo32 g, ShiftTable32 tab16, WindLim lim, int e0, int ne,
                                                    double* __restrict__ W /* [16][ne*cols] */) {
    const int idx0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = ne * g.cols;
    if (idx0 >= M) return;
    const int r = idx0 % ne, c = idx0 / ne;
    const int gr = e0 + r, b = gr - (g.row0 - g.hn), idx = b + g.RB * c;
    // (rows of the aggregation blocks outside the supplied array: zpad's 0, as in k_windcoef)
    const bool own = (unsigned)b < (unsigned)g.RB;
    const double z0 = own ? Z[own ? idx : 0] : 0.0;
    for (int d = 0; d < 16; ++d) {
        const double h = stencil_max<true>(Z, g, tab16.s[d], b, gr, c, idx, z0, lim);
        W[(int64_t)d * M + idx0] = 1 - atan(0.17 * 100 * h) / 1.65;   // int:966
    }
}

void fill_shifts(ShiftTable& t, int ndir) {
    for (int d = 0; d < ndir; ++d) {
        double azi = (d * 360.0 / ndir) * (3.14159265358979323846 / 180);   // .ar(), int:113-115
        for (int s = 1; s <= 10; ++s) {
            double s2 = (double)(s * s);
            // R: rows (101 - cos(azi)*s^2):(...), cols (101 + sin(azi)*s^2):(...); fractional
            // indices are truncated toward zero when subsetting
            t.s[d][s - 1].dr = (int)trunc(101 - cos(azi) * s2) - 101;
            t.s[d][s - 1].dc = (int)trunc(101 + sin(azi) * s2) - 101;
            t.s[d][s - 1].s2 = s2;
            t.s[d][s - 1].inv_s2 = 1.0 / s2;
        }
    }
}

#define T_TRY(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                     __FILE__, __LINE__);                                                    \
            return mcf::api_fail(e_ == hipErrorOutOfMemory ? MCF_ERR_NOMEM : MCF_ERR_HIP, b_); \
        }                                                                                    \
    } while (0)

struct DevBufs {
    std::vector<void*> p;
    ~DevBufs() { for (void* q : p) (void)hipFree(q); }
    int alloc(void** out, int64_t bytes) {
        if (bytes <= 0) bytes = 8;
        hipError_t e = hipMalloc(out, (size_t)bytes);
        if (e != hipSuccess) return mcf::api_fail(MCF_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        p.push_back(*out);
        return MCF_OK;
    }
};

}  // namespace

// Device-level entry: `t.d_dtm` and every output pointer are device memory.  Used by
// mcf_precompute_terrain (host arrays) and by the snow driver's 5-day terrain refresh (mcf_snow.hip).
namespace mcf {
void TerrainWork::release() {
    for (int i = 0; i < 3; ++i) {
        if (p[i]) (void)hipFree(p[i]);
        p[i] = nullptr;
        cap[i] = 0;
    }
}
int terrain_device(const TerrainDev& t, TerrainWork* work) {
    const int64_t rows_total = t.rows_total > 0 ? t.rows_total : t.rows;
    const int64_t row0 = t.rows_total > 0 ? t.row0 : 0;
    const int s = t.agg > 0 ? t.agg : 10;
    const bool want_wsa = t.d_wsa != nullptr;
    Geo g;
    g.rows = t.rows; g.cols = t.cols; g.hn = t.halo_north;
    g.RB = t.halo_north + t.rows + t.halo_south;
    g.row0 = row0; g.rows_total = rows_total;
    const int64_t N = g.rows * g.cols, NB = g.RB * g.cols;
    DevBufs db;
    int rc;
    // scratch buffer `slot` of at least `bytes`: the caller's workspace if there is one, else released on return
    auto scratch = [&](int slot, void** out, int64_t bytes) -> int {
        if (!work) return db.alloc(out, bytes);
        if (work->cap[slot] < bytes) {
            if (work->p[slot]) (void)hipFree(work->p[slot]);
            work->p[slot] = nullptr;
            work->cap[slot] = 0;
            hipError_t e = hipMalloc(&work->p[slot], (size_t)bytes);
            if (e != hipSuccess) return mcf::api_fail(MCF_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
            work->cap[slot] = bytes;
        }
        *out = work->p[slot];
        return MCF_OK;
    };
    double* d_Z;
    if ((rc = scratch(0, (void**)&d_Z, NB * 8))) return rc;
    hipLaunchKernelGGL(k_prep, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, nullptr, t.d_dtm, d_Z, NB, 1.0 / t.res);
    const unsigned gridN = (unsigned)((N + 255) / 256);
    // every extent in 32 bits (a 16 384 x 16 384 block with its halos still is): the lean stencils
    static const bool no32 = getenv("MCF_TERRAIN_64") != nullptr;
    const bool fits32 = !no32 && NB < ((int64_t)1 << 30) && rows_total < ((int64_t)1 << 30);
    Geo32 g32;
    g32.rows = (int)g.rows; g32.cols = (int)g.cols; g32.RB = (int)g.RB; g32.hn = (int)g.hn; g32.row0 = (int)g.row0; g32.rows_total = (int)g.rows_total;
    auto narrow = [](const ShiftTable& a, ShiftTable32& b) {
        for (int d = 0; d < 24; ++d)
            for (int q = 0; q < 10; ++q) { b.s[d][q].dr = a.s[d][q].dr; b.s[d][q].dc = a.s[d][q].dc; }
    };
    if (t.d_hor || t.d_svfa) {
        ShiftTable t24;
        fill_shifts(t24, 24);
        if (fits32) {
            ShiftTable32 s24;
            narrow(t24, s24);
            WindLim lim{};
            hipLaunchKernelGGL(k_horizon32, dim3(gridN), dim3(256), 0, nullptr, d_Z, g32, s24, lim, t.d_hor, t.d_svfa);
        } else {
            hipLaunchKernelGGL(k_horizon, dim3(gridN), dim3(256), 0, nullptr, d_Z, g, t24, t.d_hor, t.d_svfa);
        }
        T_TRY(hipGetLastError());
    }
    if (want_wsa) {
        ShiftTable t16;
        memset(&t16, 0, sizeof t16);
        fill_shifts(t16, 16);
        const int64_t NItot = (rows_total + s - 1) / s, nJ = (g.cols + s - 1) / s;
        auto clampI = [&](int64_t i) { return std::min<int64_t>(std::max<int64_t>(i, 0), NItot - 1); };
        const int64_t I0 = clampI((int64_t)floor(((double)row0 - (s - 1) / 2.0) / s));
        const int64_t I1 = clampI((int64_t)floor(((double)(row0 + g.rows - 1) - (s - 1) / 2.0) / s) + 1);
        const int64_t nI = I1 - I0 + 1;
        const int64_t e0 = I0 * s, e1 = std::min<int64_t>((I1 + 1) * s, rows_total), ne = e1 - e0;
        double *d_W, *d_C;
        if ((rc = scratch(1, (void**)&d_W, 16 * ne * g.cols * 8))) return rc;
        if ((rc = scratch(2, (void**)&d_C, 16 * nI * nJ * 8))) return rc;
        int64_t M = ne * g.cols;
        if (fits32 && M < ((int64_t)1 << 30)) {
            ShiftTable32 s16;
            memset(&s16, 0, sizeof s16);
            narrow(t16, s16);
            WindLim lim;
            for (int q = 0; q < 10; ++q) lim.v[q] = (t.zref / t.res) / (double)((q + 1) * (q + 1));
            hipLaunchKernelGGL(k_windcoef32, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, d_Z, g32, s16, lim, (int)e0, (int)ne, d_W);
        } else {
            hipLaunchKernelGGL(k_windcoef, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, d_Z, g, t16,
                               t.zref / t.res, e0, ne, d_W);
        }
        hipLaunchKernelGGL(k_block_mean, dim3((unsigned)((16 * nI * nJ + 255) / 256)), dim3(256), 0, nullptr, d_W, g,
                           s, e0, ne, I0, nI, nJ, d_C);
        hipLaunchKernelGGL(k_resample_blend, dim3(gridN), dim3(256), 0, nullptr, d_C, g, s, I0, nI, nJ, NItot, t.d_wsa);
        T_TRY(hipGetLastError());
    }
    if (t.d_slope || t.d_aspect) {
        hipLaunchKernelGGL(k_slope_aspect, dim3(gridN), dim3(256), 0, nullptr, t.d_dtm, g, t.res, t.aspect_na,
                           t.d_slope, t.d_aspect);
        T_TRY(hipGetLastError());
    }
    T_TRY(hipDeviceSynchronize());   // temporaries are released on return
    return MCF_OK;
}
}  // namespace mcf

extern "C" int mcf_precompute_terrain(const mcf_terrain_in* in, const mcf_terrain_out* out, int32_t device) {
    if (!in || !out || !in->dtm) return mcf::api_fail(MCF_ERR_ARG, "null terrain argument");
    if (in->rows <= 0 || in->cols <= 0 || in->halo_north < 0 || in->halo_south < 0 || !(in->res > 0))
        return mcf::api_fail(MCF_ERR_ARG, "bad terrain geometry");
    const int64_t rows_total = in->rows_total > 0 ? in->rows_total : in->rows;
    const int64_t row0 = in->rows_total > 0 ? in->row0 : 0;
    if (row0 < 0 || row0 + in->rows > rows_total) return mcf::api_fail(MCF_ERR_ARG, "block outside the raster");
    const int s = in->agg > 0 ? in->agg : 10;
    // halo actually required: 100 rows for the +-100-cell stencil; the wind-shelter block means and
    // their bilinear blend reach (s-1)/2 + s rows further
    const bool want_wsa = out->wsa != nullptr;
    const int64_t need = want_wsa ? 100 + 2 * s + s / 2 : ((out->hor || out->svfa) ? 100 : 1);
    const int64_t avail_n = row0, avail_s = rows_total - row0 - in->rows;
    if (in->halo_north < std::min(need, avail_n) || in->halo_south < std::min(need, avail_s)) {
        char b[200];
        snprintf(b, sizeof b, "terrain block needs %lld halo rows (or all rows up to the raster edge)", (long long)need);
        return mcf::api_fail(MCF_ERR_ARG, b);
    }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return mcf::api_fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
    if (device < 0 || device >= nd) return mcf::api_fail(MCF_ERR_ARG, "device ordinal out of range");
    T_TRY(hipSetDevice(device));

    const int64_t N = in->rows * in->cols, NB = (in->halo_north + in->rows + in->halo_south) * in->cols;
    DevBufs db;
    int rc;
    double* d_dtm;
    if ((rc = db.alloc((void**)&d_dtm, NB * 8))) return rc;
    T_TRY(hipMemcpy(d_dtm, in->dtm, (size_t)NB * 8, hipMemcpyHostToDevice));
    mcf::TerrainDev t;
    memset(&t, 0, sizeof t);
    t.rows = in->rows; t.cols = in->cols; t.halo_north = in->halo_north; t.halo_south = in->halo_south;
    t.row0 = in->row0; t.rows_total = in->rows_total;
    t.d_dtm = d_dtm; t.res = in->res; t.zref = in->zref; t.agg = in->agg; t.aspect_na = 0.0;   // int:1132-1136
    if (out->slope && (rc = db.alloc((void**)&t.d_slope, N * 8))) return rc;
    if (out->aspect && (rc = db.alloc((void**)&t.d_aspect, N * 8))) return rc;
    if (out->hor && (rc = db.alloc((void**)&t.d_hor, N * 24 * 8))) return rc;
    if (out->svfa && (rc = db.alloc((void**)&t.d_svfa, N * 8))) return rc;
    if (out->wsa && (rc = db.alloc((void**)&t.d_wsa, N * 8 * 8))) return rc;
    if ((rc = mcf::terrain_device(t))) return rc;
    if (out->slope) T_TRY(hipMemcpy(out->slope, t.d_slope, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (out->aspect) T_TRY(hipMemcpy(out->aspect, t.d_aspect, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (out->hor) T_TRY(hipMemcpy(out->hor, t.d_hor, (size_t)N * 24 * 8, hipMemcpyDeviceToHost));
    if (out->svfa) T_TRY(hipMemcpy(out->svfa, t.d_svfa, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (out->wsa) T_TRY(hipMemcpy(out->wsa, t.d_wsa, (size_t)N * 8 * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}

// ---- one process, several devices (include/mcf.h mcf_precompute_terrain_multi) ------------------------------------------------
// The raster in contiguous row blocks, block b on devices[b % n_devices] by that device's host thread.  What couples the
// blocks are the stencils' halo rows, and the whole elevation raster is in THIS process's memory: each block gathers its rows
// plus the halo it needs (the very rows route 1 exchanges between ranks, terrain.py exchange_halo) out of the caller's array,
// runs mcf_precompute_terrain with its placement (row0 / rows_total), and scatters its rows of the results into the caller's
// arrays — no peer copies, no collective.
extern "C" int mcf_precompute_terrain_multi(const mcf_terrain_in* in, const mcf_terrain_out* out, const mcf_multi* mu) {
    if (!in || !out || !mu || !in->dtm) return mcf::api_fail(MCF_ERR_ARG, "null terrain argument");
    if (in->rows <= 0 || in->cols <= 0 || !(in->res > 0)) return mcf::api_fail(MCF_ERR_ARG, "bad terrain geometry");
    if (in->halo_north != 0 || in->halo_south != 0 || (in->rows_total > 0 && (in->row0 != 0 || in->rows_total != in->rows)))
        return mcf::api_fail(MCF_ERR_ARG, "mcf_precompute_terrain_multi takes the whole raster (no halos, no placement)");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return mcf::api_fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
    std::vector<int> devs;
    if (mu->n_devices <= 0) for (int d = 0; d < nd; ++d) devs.push_back(d);
    else {
        if (!mu->devices) return mcf::api_fail(MCF_ERR_ARG, "n_devices > 0 with a null device list");
        for (int i = 0; i < mu->n_devices; ++i) {
            if (mu->devices[i] < 0 || mu->devices[i] >= nd) return mcf::api_fail(MCF_ERR_ARG, "device ordinal out of range");
            devs.push_back(mu->devices[i]);
        }
    }
    const int64_t R = in->rows, C = in->cols;
    const int nb = (int)std::min<int64_t>(mu->n_blocks > 0 ? mu->n_blocks : (int)devs.size(), R);
    const int s = in->agg > 0 ? in->agg : 10;
    const int64_t need = out->wsa ? 100 + 2 * s + s / 2 : ((out->hor || out->svfa) ? 100 : 1);     // as mcf_precompute_terrain asks
    std::vector<int> rcs(devs.size(), MCF_OK);
    std::vector<std::string> errs(devs.size());
    std::vector<std::thread> threads;
    for (size_t t = 0; t < devs.size(); ++t) {
        threads.emplace_back([&, t] {
            std::vector<double> ext, part[5];
            for (int b = (int)t; b < nb; b += (int)devs.size()) {
                const int64_t r0 = R * b / nb, r1 = R * (b + 1) / nb, nr = r1 - r0;
                if (nr <= 0) continue;
                const int64_t hn = std::min(need, r0), hs = std::min(need, R - r1), RB = hn + nr + hs;
                ext.resize((size_t)(RB * C));
                for (int64_t c = 0; c < C; ++c) memcpy(&ext[(size_t)(RB * c)], in->dtm + (r0 - hn) + R * c, (size_t)RB * 8);
                mcf_terrain_in bi = *in;
                bi.rows = nr; bi.halo_north = (int32_t)hn; bi.halo_south = (int32_t)hs; bi.dtm = ext.data();
                bi.row0 = r0; bi.rows_total = R;
                double* const dst[5] = {out->slope, out->aspect, out->hor, out->svfa, out->wsa};
                const int layers[5] = {1, 1, 24, 1, 8};
                mcf_terrain_out bo;
                double** const bop[5] = {&bo.slope, &bo.aspect, &bo.hor, &bo.svfa, &bo.wsa};
                for (int k = 0; k < 5; ++k) {
                    if (dst[k]) part[k].resize((size_t)(nr * C * layers[k]));
                    *bop[k] = dst[k] ? part[k].data() : nullptr;
                }
                const int rc = mcf_precompute_terrain(&bi, &bo, devs[t]);
                if (rc != MCF_OK) { rcs[t] = rc; errs[t] = mcf_last_error(); return; }
                for (int k = 0; k < 5; ++k)
                    if (dst[k])
                        for (int64_t lc = 0; lc < C * layers[k]; ++lc)      // layer-column lc of the block -> the same one of the raster
                            memcpy(dst[k] + r0 + R * lc, &part[k][(size_t)(nr * lc)], (size_t)nr * 8);
            }
        });
    }
    for (auto& th : threads) th.join();
    for (size_t t = 0; t < devs.size(); ++t)
        if (rcs[t] != MCF_OK) return mcf::api_fail(rcs[t], errs[t]);
    return MCF_OK;
}
