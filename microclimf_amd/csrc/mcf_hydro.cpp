// mcf_hydro.cpp — the topographic wetness index input of the grid solver (soilc$twi): flowaccCpp
// (reference src/microclimfCpp.cpp:5326-5408, with flowdirCpp) and `.topidx` (R/internal.R:861-874).
//
// Host code, as in the reference: flow accumulation is one elevation-ordered sweep over the WHOLE raster (a cell
// hands its count to its lowest neighbour), the one pre-compute that neither tiles nor parallelises;
// it runs once per raster, 1024 x 1024 cells in ~0.15 s, next to a year of solver time.
//
// Kept from the reference on purpose:
//  * the lowest of the 3 x 3 neighbourhood INCLUDING the cell itself, first minimum in column-major order, and only
//    values below 9999.99 (cpp:5346): a pit points at itself and doubles its own count (cpp:5399-5401); cells whose
//    whole neighbourhood lies at or above 9999.99 m get no direction and pass nothing on;
//  * ties in the elevation order are processed larger (row-major) index first (std::greater on (value, index) pairs);
//  * the last cell of the order is never processed (cpp:5387 `size() - 1`);
//  * NA cells hold (double)NA_INTEGER = -2147483648 in the result (cpp:5374), not NA_real_.
// Guarded: an all-NA raster (the reference's `size() - 1` underflows).  Any NaN counts as NA here.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mcf.h"

namespace mcf {
int api_fail(int code, const std::string& msg);   // mcf_api.hip
}

namespace {

constexpr double kNaInt = -2147483648.0;

void flow_direction(int64_t R, int64_t C, const double* dm, std::vector<int8_t>& fd) {
    fd.assign((size_t)(R * C), 0);
    for (int64_t i = 0; i < R; ++i)
        for (int64_t j = 0; j < C; ++j) {
            if (isnan(dm[i + R * j])) continue;
            double minval = 9999.99;
            int indx = 1, best = 0;
            for (int jj = -1; jj <= 1; ++jj)
                for (int ii = -1; ii <= 1; ++ii, ++indx) {
                    const int64_t y = i + ii, x = j + jj;
                    if (y < 0 || y >= R || x < 0 || x >= C) continue;
                    const double v = dm[y + R * x];
                    if (!isnan(v) && v < minval) { minval = v; best = indx; }
                }
            fd[(size_t)(i + R * j)] = (int8_t)best;
        }
}

void flow_accumulation(int64_t R, int64_t C, const double* dm, double* fa) {
    std::vector<int8_t> fd;
    flow_direction(R, C, dm, fd);
    std::vector<std::pair<double, int64_t>> order;
    order.reserve((size_t)(R * C));
    for (int64_t i = 0; i < R; ++i)
        for (int64_t j = 0; j < C; ++j) {
            const double v = dm[i + R * j];
            fa[i + R * j] = isnan(v) ? kNaInt : 1.0;
            if (!isnan(v)) order.push_back({v, i * C + j});
        }
    if (order.empty()) return;
    std::sort(order.begin(), order.end(), std::greater<std::pair<double, int64_t>>());
    for (size_t k = 0; k + 1 < order.size(); ++k) {
        const int64_t y = order[k].second / C, x = order[k].second % C;
        const int f = fd[(size_t)(y + R * x)];
        if (f < 1 || f > 9) continue;
        const int64_t y2 = y + (f - 1) % 3 - 1, x2 = x + (f - 1) / 3 - 1;
        if (x2 >= 0 && x2 < C && y2 >= 0 && y2 < R && fa[y2 + R * x2] != kNaInt) fa[y2 + R * x2] += fa[y + R * x];
    }
}

}  // namespace

extern "C" int mcf_flowacc(int64_t rows, int64_t cols, const double* dtm, double* fa) {
    if (rows <= 0 || cols <= 0 || !dtm || !fa) return mcf::api_fail(MCF_ERR_ARG, "mcf_flowacc: bad dimensions or null argument");
    flow_accumulation(rows, cols, dtm, fa);
    return MCF_OK;
}

// .topidx, R/internal.R:861-874: a / tan(B) with a = (flowacc + 1) * xres * yres floored at 1 and B = Horn slope in
// radians (terra::terrain(dtm, unit = "radians"): NA on the raster edge and beside NA cells) floored at
// atan(0.02 / mean(res)), NA slopes replaced by the median of the others; masked by the dtm.
extern "C" int mcf_topidx(int64_t rows, int64_t cols, const double* dtm, double xres, double yres, double* twi) {
    if (rows <= 0 || cols <= 0 || !dtm || !twi || !(xres > 0) || !(yres > 0))
        return mcf::api_fail(MCF_ERR_ARG, "mcf_topidx: bad dimensions, resolution or null argument");
    const int64_t R = rows, C = cols, N = R * C;
    const double na = nan("");
    std::vector<double> B((size_t)N, na);
    for (int64_t j = 1; j + 1 < C; ++j)
        for (int64_t i = 1; i + 1 < R; ++i) {
            auto z = [&](int di, int dj) { return dtm[(i + di) + R * (j + dj)]; };
            // Horn (terra's 8-neighbour slope): north = row - 1, east = col + 1
            const double dzdx = ((z(-1, 1) + 2 * z(0, 1) + z(1, 1)) - (z(-1, -1) + 2 * z(0, -1) + z(1, -1))) / (8 * xres);
            const double dzdy = ((z(-1, -1) + 2 * z(-1, 0) + z(-1, 1)) - (z(1, -1) + 2 * z(1, 0) + z(1, 1))) / (8 * yres);
            B[(size_t)(i + R * j)] = isnan(z(0, 0)) ? na : atan(sqrt(dzdx * dzdx + dzdy * dzdy));   // NaN neighbour -> NaN
        }
    const double minslope = atan(0.02 / (0.5 * (xres + yres)));
    std::vector<double> ok;
    ok.reserve((size_t)N);
    for (double& b : B) {
        if (isnan(b)) continue;
        if (b < minslope) b = minslope;
        ok.push_back(b);
    }
    double med = na;
    if (!ok.empty()) {                                    // R's median: mean of the two middle values for an even count
        const size_t h = ok.size() / 2;
        std::nth_element(ok.begin(), ok.begin() + h, ok.end());
        med = ok[h];
        if (ok.size() % 2 == 0) med = 0.5 * (med + *std::max_element(ok.begin(), ok.begin() + h));
    }
    std::vector<double> fa((size_t)N);
    flow_accumulation(R, C, dtm, fa.data());
    const double NA_REAL = [] { uint64_t u = 0x7FF00000000007A2ULL; double d; memcpy(&d, &u, 8); return d; }();
    for (int64_t c = 0; c < N; ++c) {
        if (isnan(dtm[c])) { twi[c] = NA_REAL; continue; }
        double a = (fa[(size_t)c] + 1.0) * xres * yres;
        if (a < 1.0) a = 1.0;
        const double b = isnan(B[(size_t)c]) ? med : B[(size_t)c];
        twi[c] = a / tan(b);
    }
    return MCF_OK;
}
