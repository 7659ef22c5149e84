// mcf_snow.hip — the snow branch on the device: kernels and C-ABI entry points
//   mcf_gridmodelsnow1/2   per-cell, sequential-in-time snowpack model   cpp:4172-4673
//   mcf_gridmicrosnow1/2   microclimate of snow-covered cell-steps       cpp:4894-5214
// ("cpp:" = the reference's src/microclimfCpp.cpp).  Physics: mcf_snow_device.hpp.
//
// Parallel shape.  The snowpack is a recurrence in time (depth, density and age of two layers
// carried from step to step), so gridmodelsnow runs ONE LANE PER CELL with the lanes of a wave
// along the raster rows: every [rows,cols,tsteps] store is a coalesced 512-B line per wave and
// step, the state lives in registers, and — with data.frame climate — the per-step table row is
// wave-uniform (scalar loads).  gridmicrosnow has no recurrence: one lane per (cell, day) walks
// 24 hours after forming the day's mean snow temperature (snowdayan, cpp:4679-4712).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcf.h"
#include "mcf_hostpipe.hpp"
#include "mcf_snow_device.hpp"

// the ring kernel (lane per cell-hour, cell table in LDS, step record through scalar loads): 141 VGPRs without scratch at three
// waves per SIMD, 128 + 44 B (eight spill stores and loads per hour) at four — measured 0.515 against 0.55 s per simulated year
// of configs[4]'s share (the kernel moves ~144 B per snow cell-step: it is closer to the memory system's limit than to the
// VALU's, and the fourth wave hides more of the loads)
// waves per SIMD k_snowmodel is built for.  Round 4: with the cell's constants read from the workgroup's LDS table at their uses
// and the five output streams addressed as uniform base + 32-bit lane offset, the data.frame kernel fits 128 VGPRs without
// scratch (156 in round 3: three waves) — 105 once the step table comes through scalar loads (see the kernel's head); a fifth
// wave is out of reach through LDS, not registers (36 KB per 4-wave workgroup: the cell table + the exp / log tables).  The
// array-climate kernel derives the step's weather terms per lane and stays at three (168 VGPRs + 116 B of scratch; 232 B in round 3).
#ifndef MCF_MICRORING_WAVES
#define MCF_MICRORING_WAVES 4
#endif
#ifndef MCF_SNOW_WAVES
#define MCF_SNOW_WAVES 4
#endif
#ifndef MCF_SNOW_WAVES_AF
#define MCF_SNOW_WAVES_AF 3
#endif
#include "mcf_terrain.h"

namespace mcf {
int api_fail(int code, const std::string& msg);   // mcf_api.hip
}

namespace {
using namespace mcf;
using namespace mcf::snow;

// One time step with data.frame climate: everything that does not depend on the cell.
struct StepRow {
    MetT m;
    DayT d;
    SunT s;
    double rnet;     // RswabsG + RlwabsG - Rem of the point model, cpp:4229
    int32_t sindex, windex;
};
// Date part of a step for array climate (site part applied per cell).
struct DateRow2 {
    double sindec, cosdec, cosA, sinA;     // A = 0.261799 (hour + eot / 60 - 12): the hour angle without the cell's longitude
    int32_t windex, pad;
};

struct StepArgs {
    int tsteps;
    const int32_t *year, *month, *day;
    const double* hour;
    const double *temp, *relhum, *pres, *swdown, *difrad, *lwdown, *windspeed, *winddir, *precip;
    const double *Gp, *Tcp, *RswabsG, *RlwabsG, *umu;   // null for gridmicrosnow
    double lat, lon;
    int32_t degrees;   // horizon test in degrees (gridmodelsnow1) or radians
    StepRow* rows;
    DateRow2* dates;   // array climate
    double* mxtc;      // [1]
};

// ---- data.frame climate: per-step table -------------------------------------------------------
__global__ __launch_bounds__(256) void k_snow_steps(StepArgs a) {
    snow::snow_tables_init();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.tsteps) return;
    // (the record is assembled where it lives: a local StepRow — 44 doubles whose members are then copied as a block — went through
    // 352 B of scratch per lane)
    StepRow& r = a.rows[k];
    r.d = DayT{};
    r.rnet = 0.0;
    const SolDate sd = sol_date(a.year[k], a.month[k], a.day[k]);
    const double latr = a.lat * kPi / 180.0;
    const SolPos sp = sol_site(sd, a.hour[k], sin(latr), cos(latr), a.lon);
    r.s = sun_derive(sp, a.degrees != 0);
    r.sindex = dir_index(sp.azid, 15.0, 24);
    r.windex = dir_index(a.winddir[k], 45.0, 8);
    if (a.Gp) {
        met_derive(r.m, a.temp[k], a.relhum[k], a.pres[k], a.Tcp[k]);
        r.m.prec = a.precip[k];
        r.m.rsw = a.swdown[k]; r.m.rdif = a.difrad[k]; r.m.rlw = a.lwdown[k];
        r.m.umu = a.umu[k]; r.m.u2 = a.windspeed[k]; r.m.gp = a.Gp[k];
        r.rnet = a.RswabsG[k] + a.RlwabsG[k] - r.m.rem;
    } else {
        r.m = MetT{};
    }
}
// daily extremes of the point model's net radiation (cpp:4231-4282): one lane per day
__global__ __launch_bounds__(64) void k_snow_days(StepRow* rows, int tsteps) {
    snow::snow_tables_init();
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= tsteps / 24) return;
    DayT dy;
    day_init(dy);
    for (int h = 0; h < 24; ++h) {
        const StepRow& r = rows[d * 24 + h];
        day_accum(dy, r.rnet, r.m.rsw, r.m.rlw);
    }
    for (int h = 0; h < 24; ++h) rows[d * 24 + h].d = dy;
}
// snowalbCpp is a scan over time (hours since snowfall): a single lane walks the series once;
// the same walk yields the series maximum of temperature that gridmicrosnow1 needs (cpp:4973-4974)
__global__ void k_snow_alb(StepRow* rows, const double* precip, const double* temp, int tsteps, double* mxtc) {
    snow::snow_tables_init();
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int hs = 0;
    double mx = -273.15;
    for (int k = 0; k < tsteps; ++k) {
        if (k > 0) hs = precip[k] > 0 ? 0 : hs + 1;
        rows[k].m.alb = snow_albedo(hs);
        rows[k].m.ialb = gdiv(1.0, rows[k].m.alb);
        if (temp[k] > mx) mx = temp[k];
    }
    if (mxtc) *mxtc = mx;
}
// array climate: date-only part of the sun position
__global__ __launch_bounds__(256) void k_snow_dates(StepArgs a) {
    snow::snow_tables_init();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.tsteps) return;
    const SolDate sd = sol_date(a.year[k], a.month[k], a.day[k]);
    DateRow2 r;
    const double A = 0.261799 * (a.hour[k] + sd.eot / 60.0 - 12.0);             // cpp:44, 54 without the longitude term
    r.sindec = sd.sindec; r.cosdec = sd.cosdec; r.cosA = cos(A); r.sinA = sin(A);
    r.windex = dir_index(a.winddir[k], 45.0, 8);
    r.pad = 0;
    a.dates[k] = r;
}

// ---- gridmodelsnow ---------------------------------------------------------------------------------
struct ModelArgs {
    int64_t N;
    int tsteps;
    const double *pai, *hgt, *leaft, *clump, *slope, *aspect, *skyview, *wsa, *hor, *lats, *lons;
    const double *isnowdc, *isnowdg;
    const int32_t *isnowac, *isnowag;
    double zref;
    double sdp[4];
    const StepRow* rows;     // data.frame climate
    const DateRow2* dates;   // array climate
    const double *temp, *relhum, *pres, *swdown, *difrad, *lwdown, *windspeed, *precip;   // [N][T]
    const double *Gp, *Tcp, *RswabsG, *RlwabsG, *umu;                                     // [N][T]
    double *Tc, *Tg, *sdepc, *sdepg, *sden;   // [N][T] or null
    double *agec, *ageg, *meltc, *meltg;      // [N] or null
    // `.snowmodel1`'s topographic redistribution and hand-over (R/internal.R:2589-2612) fused into the chunk's model run
    // (tpic != null; the snow plan's chunk loop): the depths leave the kernel redistributed — sdepc holds totalSWE, sdepg the
    // ground snow depth — and the state of the next chunk (pack depth, snow surface, the two ages) is written by the same lane.
    // Until round 4 a second kernel re-read and re-wrote the chunk's three depth / density series: 10 GB per chunk of a
    // 512 x 4096 block, 0.21 s of a simulated year.
    const double *tpic, *dtm;
    double tpimean;
    double *isnowdc_out, *dtms;
    int32_t *isnowac_out, *isnowag_out;
    // [N][T / 24] or null (data.frame climate): the day's mean of the Tg series, as the snow-day microclimate takes it (cpp:5010-5016:
    // the 24 values added in their order, / 24; NA where the day's first value is) — made here, where the values are, its reader
    // does not read the series a second time for it
    double* tzd;
};

template <bool AF>
__global__ __launch_bounds__(256, AF ? MCF_SNOW_WAVES_AF : MCF_SNOW_WAVES) void k_snowmodel(ModelArgs a, const StepRow* __restrict__ rows,
                                                                                             const DateRow2* __restrict__ dates) {
    // (rows / dates = a.rows / a.dates as `__restrict__` kernel arguments of their own: a pointer read out of the by-value struct
    // carries no noalias, the series' stores might clobber the table for all the compiler knows, and every field of a step's
    // row came through a VECTOR load of a uniform address — 40 loads and as many VGPR pairs per step instead of scalar loads)
    snow::snow_tables_init();
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.N) return;
    const int64_t N = a.N;
    const double NA = na_real();
    const double hgt0 = a.hgt[c];
    // fused redistribution (see ModelArgs): the cell's tpi weight and the depths the chunk starts from
    const bool redist = a.tpic != nullptr;
    // (the cell's constants and the redistribution's state live in the workgroup's LDS table, lane = column: read — and the
    // two running values written — inside the step where they are used, not carried in registers)
    __shared__ double s_cv[CV_COUNT + 5 + (AF ? 4 : 0)][256];
    enum { RV_TPI = CV_COUNT, RV_ASD, RV_ASC, RV_TOT, RV_GD };
    {
        const int l = threadIdx.x;
        s_cv[RV_TPI][l] = redist ? a.tpic[c] / a.tpimean : 0.0;      // tpic / mean(tpic, na.rm = TRUE)
        s_cv[RV_ASD][l] = redist ? a.isnowdg[c] : 0.0;
        s_cv[RV_ASC][l] = redist ? a.isnowdc[c] : 0.0;
        s_cv[RV_TOT][l] = 0.0; s_cv[RV_GD][l] = 0.0;
    }
    // (dc, dg, den) of a step -> what is stored: as they are, or redistributed (int:2593-2606; the operations of the former
    // k_snow_redistribute in its order, also on an NA cell's NAs)
    auto finish = [&](double& dc, double& dg, double den) {
        if (!redist) return;
        int l = threadIdx.x;
        asm volatile("" : "+v"(l));
        const double r_asd = s_cv[RV_ASD][l], r_asc = s_cv[RV_ASC][l];
        const double dsnow = dg - r_asd;
        double dsnow2 = dsnow * s_cv[RV_TPI][l];
        if (dsnow < 0) dsnow2 = dsnow;
        const double cdsnow = dc - r_asc - dsnow;
        const double r_tot = r_asc + cdsnow + dsnow2, r_gd = r_asd + dsnow2;
        s_cv[RV_TOT][l] = r_tot; s_cv[RV_GD][l] = r_gd;
        dc = r_tot * den;
        dg = r_gd;
    };
    if (isnan(hgt0)) {   // cpp:4320-4321
        for (int k = 0; k < a.tsteps; ++k) {
            const int64_t o = c + N * k;
            double dc = NA, dg = NA;
            finish(dc, dg, NA);
            if (a.Tc) a.Tc[o] = NA;
            if (a.Tg) a.Tg[o] = NA;
            if (a.sdepc) a.sdepc[o] = dc;
            if (a.sdepg) a.sdepg[o] = dg;
            if (a.sden) a.sden[o] = NA;
        }
        if constexpr (!AF) {
            if (a.tzd && a.Tg)
                for (int d = 0; d < a.tsteps / 24; ++d) a.tzd[c + N * d] = NA;
        }
        if (redist && a.tsteps > 0) { a.isnowdc_out[c] = s_cv[RV_TOT][threadIdx.x]; a.dtms[c] = a.dtm[c] + s_cv[RV_GD][threadIdx.x]; }
        if (a.agec) a.agec[c] = NA;
        if (a.ageg) a.ageg[c] = NA;
        if (a.meltc) a.meltc[c] = NA;
        if (a.meltg) a.meltg[c] = NA;
        return;
    }
    {
        const int l = threadIdx.x;
        const double slope = a.slope[c];
        const SiteK sk = site_derive(slope, a.aspect[c]);
        s_cv[CV_PAI][l] = a.pai[c]; s_cv[CV_HGT][l] = hgt0; s_cv[CV_CLUMP][l] = a.clump[c]; s_cv[CV_LTRA][l] = a.leaft[c];
        s_cv[CV_SKYVIEW][l] = a.skyview[c];
        s_cv[CV_CS][l] = sk.cS; s_cv[CV_SS][l] = sk.sS; s_cv[CV_CA][l] = sk.cA; s_cv[CV_SA][l] = sk.sA; s_cv[CV_SLOPE][l] = slope;
    }
    if (AF) {     // the cell's part of the sun position: four more rows of the LDS table, read inside the step
        const SunCell sc = sun_cell(a.lats[c], a.lons[c]);
        const int l = threadIdx.x;
        s_cv[RV_GD + 1][l] = sc.sinlat; s_cv[RV_GD + 2][l] = sc.coslat; s_cv[RV_GD + 3][l] = sc.cosB; s_cv[RV_GD + 4][l] = sc.sinB;
    }
    Pack s;   // cpp:4325-4332
    s.agec = a.isnowac[c];
    s.ageg = a.isnowag[c];
    s.sdenc = snow_density(a.sdp, a.isnowdc[c], (double)s.agec);
    s.sdeng = ((a.sdp[0] - a.sdp[1]) * (1 - exp(-a.sdp[2] * a.isnowdg[c] * 0.5 / 100.0 - a.sdp[3] * s.ageg / 24.0)) +
               a.sdp[1]) * 1000.0;
    s.sdepc = a.isnowdc[c];
    s.sdepg = a.isnowdg[c];
    double meltc = AF ? NA : 0.0;   // only gridmodelsnow1 zeroes it (cpp:4333); gridmodelsnow2 adds to NA
    double meltg = NA;              // never zeroed in either (cpp:4308, 4396)
    const double nosnow_den = a.sdp[1] * 1000.0;
    const int ndays = a.tsteps / 24;
    int hs = 0;
    DayT dy;
    [[maybe_unused]] double tzsum = 0.0;         // the day's Tg values added up (tzd)
    [[maybe_unused]] bool tz_na = false;
    for (int k = 0; k < a.tsteps; ++k) {
        const int64_t o = c + N * k;
        // (an opaque lane index per step: the table's values are read at their uses, not hoisted in front of the loop)
        int li = (int)threadIdx.x;
        asm volatile("" : "+v"(li));
        CellV cv;
        cv.p = &s_cv[0][li]; cv.cs = 256;
        double tc, prec;
        if (AF) {
            tc = a.temp[o]; prec = a.precip[o];
            if (k > 0) hs = prec > 0 ? 0 : hs + 1;
            if (k % 24 == 0) {   // the day's extremes of the point model's net radiation, cpp:4514-4566
                if (k / 24 < ndays) {
                    day_init(dy);
                    for (int h = 0; h < 24; ++h) {
                        const int64_t q = c + N * (k + h);
                        const double rnet = a.RswabsG[q] + a.RlwabsG[q] - 0.97 * kSb * rad4(a.temp[q]);
                        day_accum(dy, rnet, a.swdown[q], a.lwdown[q]);
                    }
                } else {
                    dy.rmx = dy.rmn = dy.rswmx = dy.rlwmx = dy.rswmn = dy.rlwmn = dy.gmx = 0.0;
                }
            }
        } else {
            tc = rows[k].m.tc; prec = rows[k].m.prec;
        }
        bool snowtest = s.sdepc > 0.0;                       // cpp:4336-4338
        if (tc < 2.0 && prec > 0.0) snowtest = true;
        double vTc = 0.0, vTg = 0.0, vdc = 0.0, vdg = 0.0, vden = nosnow_den;
        if (snowtest) {
            PackOut po;
            if (AF) {
                MetT m;
                met_derive(m, tc, a.relhum[o], a.pres[o], a.Tcp[o]);
                m.prec = prec;
                m.rsw = a.swdown[o]; m.rdif = a.difrad[o]; m.rlw = a.lwdown[o];
                m.umu = a.umu[o]; m.u2 = a.windspeed[o]; m.gp = a.Gp[o];
                m.alb = snow_albedo(hs);
                m.ialb = gdiv(1.0, m.alb);
                const DateRow2 dr = dates[k];
                SunCell sc;
                sc.sinlat = s_cv[RV_GD + 1][li]; sc.coslat = s_cv[RV_GD + 2][li]; sc.cosB = s_cv[RV_GD + 3][li]; sc.sinB = s_cv[RV_GD + 4][li];
                int sindex;
                const SunT sun = sun_at_cell(dr.sindec, dr.cosdec, dr.cosA, dr.sinA, sc, sindex);
                const double ha = a.hor[(int64_t)sindex * N + c];
                const double ws = a.wsa[(int64_t)dr.windex * N + c];
                pack_step(m, dy, sun, cv, ha, ws, a.sdp, a.zref, s, po);
            } else {
                const StepRow& r = rows[k];
                const double ha = a.hor[(int64_t)r.sindex * N + c];
                const double ws = a.wsa[(int64_t)r.windex * N + c];
                pack_step(r.m, r.d, r.s, cv, ha, ws, a.sdp, a.zref, s, po);
            }
            vTc = po.Tc; vTg = po.Tg; vdc = s.sdepc; vdg = s.sdepg; vden = s.sdenc;
            meltc = meltc + po.melc;                          // cpp:4394-4396 (meltc is added twice)
            meltc = meltc + gdiv(po.melc * 1000.0, s.sdenc);   // densities are >= 1000 sdp[1] > 0 (or NaN)
            meltg = meltg + gdiv(po.melg * 1000.0, s.sdeng);
        }
        finish(vdc, vdg, vden);
        // the step's slab of an output is a uniform base (scalar registers) + the lane's 32-bit byte offset: no 64-bit vector
        // address per series kept across the loop (N * 8 < 4 GB: checked by the host)
        {
            uint32_t cb = (uint32_t)c * 8u;
            asm volatile("" : "+v"(cb));
            const int64_t so = N * k;
            auto st = [&](double* base, double v) { *(double*)((char*)(base + so) + cb) = v; };
            if (a.Tc) st(a.Tc, vTc);
            if (a.Tg) st(a.Tg, vTg);
            if (a.sdepc) st(a.sdepc, vdc);
            if (a.sdepg) st(a.sdepg, vdg);
            if (a.sden) st(a.sden, vden);
        }
        if constexpr (!AF) {
            if (a.tzd && a.Tg) {
                const int h = k % 24;
                if (h == 0) { tzsum = 0.0; tz_na = isnan(vTg); }
                tzsum += vTg;
                if (h == 23) a.tzd[c + N * (k / 24)] = tz_na ? NA : tzsum / 24.0;
            }
        }
    }
    if (a.agec) a.agec[c] = (double)s.agec;
    if (a.ageg) a.ageg[c] = (double)s.ageg;
    if (a.meltc) a.meltc[c] = meltc;
    if (a.meltg) a.meltg[c] = meltg;
    if (redist && a.tsteps > 0) {      // hand-over to the next chunk, int:2607-2612
        a.isnowdc_out[c] = s_cv[RV_TOT][threadIdx.x];
        a.dtms[c] = a.dtm[c] + s_cv[RV_GD][threadIdx.x];
        a.isnowac_out[c] = s.agec;
        a.isnowag_out[c] = s.ageg;
    }
}

// ---- gridmicrosnow ---------------------------------------------------------------------------------
struct MicroArgs {
    int64_t N;
    int tsteps;
    double reqhgt, mat, zref, hiy;
    const double *pai, *hgt, *leaft, *clump, *paia, *leafd, *leafden, *slope, *aspect, *skyview, *wsa, *hor;
    const double *lats, *lons, *Smax;
    const StepRow* rows;
    const DateRow2* dates;
    const double* mxtc1;   // [1] data.frame climate
    const MicroMet* mmet;  // [T] data.frame climate: weather-only terms of every step (k_micro_steps)
    const void* mstep;     // [T] MicroStep (the snow plan's ring kernel): rows' sun / albedo, mmet and the step's weather in one record
    int32_t day0;          // first day of this launch (blockIdx.y counts from it)
    const double *temp, *relhum, *pres, *swdown, *difrad, *lwdown, *windspeed, *precip, *umu;   // [T] or [N][T]
    const double *sTc, *sTg, *swe, *sdepg, *sden;   // snowm, [N][T]
    const double* tzd;  // [N][T / 24] the days' means of sTg made by the snow model's kernel (ModelArgs::tzd), or null: made here
    double* meanD;      // [N]
    double* mxtc;       // [N]        array climate
    int32_t* hs0;       // [N][nchunks] hours since snowfall at the start of each day (array climate)
    double* out[MCF_NOUT];   // [N][T] or null
};

// data.frame climate: the weather-only terms of snowabovepoint, once per step instead of once per cell-step
__global__ __launch_bounds__(256) void k_micro_steps(const double* __restrict__ temp, const double* __restrict__ relhum,
                                                     const double* __restrict__ pres,
                                                     const double* __restrict__ mxtc1, int tsteps, MicroMet* __restrict__ out) {
    snow::snow_tables_init();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= tsteps) return;
    out[k] = micro_met(temp[k], relhum[k], pres[k], *mxtc1);
}

// per-cell reductions over the whole series: meanDsnow (cpp:4713-4737) and, with array climate, the
// cell's maximum temperature (cpp:5139-5145) and the albedo clock at every day start
// Everything k_microsnow_ring reads of a step, in one record: ONE table pointer in scalar registers instead of eleven (the
// kernel keeps ~100 uniform values alive; what does not fit the SGPR file is spilled to VGPR lanes, a VALU instruction per access)
struct MicroStep {
    SunT s;
    MicroMet mm;
    double tc, pk, u2, rsw, rdif, rlw, umu, alb, ialb;
    int32_t sindex, windex;
};
__global__ __launch_bounds__(256) void k_micro_pack(const StepRow* __restrict__ rows, const MicroMet* __restrict__ mmet,
                                                    const double* __restrict__ temp, const double* __restrict__ pres,
                                                    const double* __restrict__ wind, const double* __restrict__ sw,
                                                    const double* __restrict__ dif, const double* __restrict__ lw,
                                                    const double* __restrict__ umu, int T, MicroStep* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    MicroStep& m = out[k];          // (in place: a local record went through 176 B of scratch)
    m.s = rows[k].s; m.mm = mmet[k];
    m.tc = temp[k]; m.pk = pres[k]; m.u2 = wind[k]; m.rsw = sw[k]; m.rdif = dif[k]; m.rlw = lw[k]; m.umu = umu[k];
    m.alb = rows[k].m.alb; m.ialb = rows[k].m.ialb;
    m.sindex = rows[k].sindex; m.windex = rows[k].windex;
}
// a step's term of meanDsnow (cpp:4713-4737): sqrt(2 kappa / omega), kappa from the pack's density (> 0, or NaN) — with the lean
// exponential, quotient and root (the device libm's made k_meand_accumulate a compute-bound kernel: 11 000 instructions per wave)
__device__ __forceinline__ double snow_damping_term(double den) {
    const double co = 0.0442 * gexp(5.181 * den * (1.0 / 1000.0));
    const double kap = gdiv(co, den * 2090.0);
    return gsqrt(kap * (2.0 / kOmdy));
}
template <bool AF>
__global__ __launch_bounds__(256) void k_microsnow_cell(MicroArgs a) {
    snow::snow_tables_init();
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.N) return;
    if (isnan(a.hgt[c])) return;
    const int64_t N = a.N;
    double meanD = na_real();
    if (!isnan(a.sden[c])) {
        double sumD = 0.0;
        for (int k = 0; k < a.tsteps; ++k) {
            sumD += snow_damping_term(a.sden[c + N * k]);
        }
        meanD = sumD / (double)a.tsteps;
    }
    a.meanD[c] = meanD;
    if (AF) {
        const int nch = (a.tsteps + 23) / 24;
        double mx = -273.15;
        int hs = 0;
        for (int k = 0; k < a.tsteps; ++k) {
            const double t = a.temp[c + N * k];
            if (t > mx) mx = t;
            if (k > 0) hs = a.precip[c + N * k] > 0 ? 0 : hs + 1;
            if (k % 24 == 0) a.hs0[c + N * (k / 24)] = hs;
        }
        (void)nch;
        a.mxtc[c] = mx;
    }
}

// Array climate, streamed (mcf_snowplan_micro_setup with array weather): the two per-cell scans of k_microsnow_cell<true> — the
// maximum air temperature and the albedo clock at every day start over the snow-day SUBSET series (cpp:5139-5145, 3733-3739) —
// a run of consecutive subset days at a time, the state (mx, hs) carried in memory between the runs; same walk, same values.
__global__ __launch_bounds__(256) void k_micro_scan(const double* __restrict__ temp, const double* __restrict__ precip, int64_t N, int sub0,
                                                    int ndays, const double* __restrict__ hgt, double* __restrict__ mxtc,
                                                    int32_t* __restrict__ hs_state, int32_t* __restrict__ hs0) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    if (isnan(hgt[c])) return;
    double mx = sub0 == 0 ? -273.15 : mxtc[c];
    int hs = sub0 == 0 ? 0 : hs_state[c];
    for (int d = 0; d < ndays; ++d)
        for (int h = 0; h < 24; ++h) {
            const int64_t q = c + N * (int64_t)(d * 24 + h);
            const double t = temp[q];
            if (t > mx) mx = t;
            if (sub0 + d > 0 || h > 0) hs = precip[q] > 0 ? 0 : hs + 1;
            if (h == 0) hs0[c + N * (int64_t)(sub0 + d)] = hs;
        }
    mxtc[c] = mx;
    hs_state[c] = hs;
}

// ---- gridmicrosnow1 inside the chunk loop, device-resident (mcf_snowplan_micro_*) ---------------------------------
// meanDsnow (cpp:4713-4737) is the mean over the WHOLE snow-day series of sqrt(2 kappa / omega); the series lives on the
// device one chunk at a time, so a first pass over the year adds the chunk's snow days to a per-cell running sum — the
// same terms in the same order as k_microsnow_cell's loop.
__global__ __launch_bounds__(256) void k_meand_accumulate(const double* __restrict__ sden, const double* __restrict__ hgt,
                                                          int64_t N, int ndays, const int32_t* __restrict__ snowday, int first,
                                                          double* __restrict__ sumD, int32_t* __restrict__ sden_na) {
    snow::snow_tables_init();
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || isnan(hgt[c])) return;
    double s = sumD[c];
    bool seen = !first;
    for (int d = 0; d < ndays; ++d) {
        if (!snowday[d]) continue;
        if (!seen) { sden_na[c] = isnan(sden[c + N * (d * 24)]) ? 1 : 0; seen = true; }
        for (int h = 0; h < 24; ++h) {
            s += snow_damping_term(sden[c + N * (d * 24 + h)]);
        }
    }
    sumD[c] = s;
}
__global__ __launch_bounds__(256) void k_meand_finish(const double* __restrict__ sumD, const int32_t* __restrict__ sden_na,
                                                      const double* __restrict__ hgt, int64_t N, double nsteps,
                                                      double* __restrict__ meanD) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    meanD[c] = (isnan(hgt[c]) || sden_na[c]) ? na_real() : sumD[c] / nsteps;
}

// k_microsnow on one chunk's snow days, reading the chunk's snow series where mcf_snowplan_run_chunk left them and
// writing into a ring slot of the grid solver's plan (tiled layout, RingView) — the merge of `.runmicrosnow1`
// (R/internal.R:3565-3578, 3633-3656) per cell-step:
//   snow day, SWE > 0              the snow microclimate (variables of `outsel`; the others as in the last line)
//   snow day, no snow on the cell  the no-snow solver's value if the day is a no-snow day as well, else NA
//   other days                     untouched (the solver's value)
// One lane per (cell, day of the chunk).  m.rows / m.temp ... are the SUBSET series of the snow days (daymap[d]: the
// chunk day's place in it, -1: not a snow day); m.sTc ... m.sden the chunk's series, chunk-local steps.
struct MicroRingArgs {
    MicroArgs m;
    mcf::RingView ring;          // the slot's geometry (base unused)
    // the slot's held variables lie `vstride` doubles apart in the reference's order from `base0` (the tiled ring's
    // [tile][day][variable][block]): bit i of `held` = the plan holds output i, its base = base0 + popcount(held below i) * vstride
    double* base0;
    int64_t vstride;
    uint32_t held, sel;          // sel: gridmicrosnow1's `out` mask
    const int32_t *daymap, *nosnow;
    int32_t ndays;
};
// Shape (round 4): one lane per (cell, hour).  A workgroup is 64 consecutive cells x one day of the chunk (blockIdx.y: the
// step rows are wave-uniform scalar loads); wave w takes the hours w, w + 4, ..., w + 20.  What depends on the cell only —
// the raw rasters and what is derived from them (the slope / aspect sines and cosines, log(clump), three reciprocals), the
// day's mean ground-snow temperature — is made ONCE per workgroup-day, a quarter by each wave, into LDS, and read from there
// inside the hour loop: until round 3 a lane walked the 24 hours of its cell with all of it in registers (230 VGPRs, two
// waves per SIMD).
enum MicroCell : int { MC_HGT, MC_PAI, MC_PAIA, MC_LEAFD, MC_CLUMP, MC_LTRA, MC_LEAFDEN, MC_SVFA, MC_LNCLUMP, MC_IHGT, MC_ILEAFD,
                       MC_IPAI, MC_CS, MC_SS, MC_CA, MC_SA, MC_SLOPE, MC_MEAND, MC_SMAX, MC_TZD, MC_COUNT,
                       // array climate (k_microsnow_ring<true>): the cell's part of the sun position, its maximum temperature
                       MC_SINLAT = MC_COUNT, MC_COSLAT, MC_COSB, MC_SINB, MC_MXTC, MC_COUNT_AF };
static_assert((int)MC_HGT == (int)MQ_HGT && (int)MC_PAI == (int)MQ_PAI && (int)MC_PAIA == (int)MQ_PAIA && (int)MC_LEAFD == (int)MQ_LEAFD && (int)MC_CLUMP == (int)MQ_CLUMP &&
              (int)MC_LTRA == (int)MQ_LTRA && (int)MC_LEAFDEN == (int)MQ_LEAFDEN && (int)MC_SVFA == (int)MQ_SVFA && (int)MC_LNCLUMP == (int)MQ_LNCLUMP &&
              (int)MC_IHGT == (int)MQ_IHGT && (int)MC_ILEAFD == (int)MQ_ILEAFD && (int)MC_IPAI == (int)MQ_IPAI, "micro_above reads the table's first rows");
// The per-step table and the two day lists come in as `__restrict__` kernel arguments of their own (tb = q.m.mstep): read out of
// the by-value struct a pointer carries no noalias, the ring's stores might clobber the table for all the compiler knows, and
// every field of an hour's record — ~30 doubles — came through VECTOR loads of a uniform address into VGPR pairs.  With that
// and the wave index declared uniform (readfirstlane) they are scalar loads into SGPRs.
// AF (round 5): array climate — gridmicrosnow2 (cpp:5058-5214).  The step's weather is the lane's own (nine [N][T] series), the
// sun position the cell's (sun_at_cell), the albedo clock the cell's: `hs0` holds the hours since snowfall at every day start
// (k_microsnow_cell<true>) and the day's precipitation is staged in LDS, so that a lane finds its hour's clock by walking back
// through at most 23 LDS values instead of re-reading the series.  The table argument then carries the date rows (DateRow2).
// The one-shot entries mcf_gridmicrosnow1 / 2 run this kernel too, over a linear [steps][cells] view of their output arrays
// (RingView with cpb = N: ring_pos(N, cell, hour) = hour N + cell) — the lane-per-(cell, day) kernel k_microsnow<AF> (197-201
// VGPRs, two waves per SIMD) is gone.
template <bool AF>
__global__ __launch_bounds__(256, AF ? 3 : MCF_MICRORING_WAVES) void k_microsnow_ring(MicroRingArgs q, const void* __restrict__ tbv,
                                                                                      const int32_t* __restrict__ tb_daymap,
                                                                                      const int32_t* __restrict__ tb_nosnow) {
    snow::snow_tables_init();
    const MicroStep* __restrict__ tb = (const MicroStep*)tbv;
    const DateRow2* __restrict__ td = (const DateRow2*)tbv;
    __shared__ double s_mc[AF ? MC_COUNT_AF : MC_COUNT][64];
    __shared__ double s_prec[AF ? 24 : 1][64];
    const MicroArgs& a = q.m;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t N = a.N;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int day = a.day0 + (int)blockIdx.y;    // uniform: the day's rows of the step tables are scalar loads
    const int sub = tb_daymap[day];
    if (sub < 0) return;
    const bool keep = tb_nosnow[day] != 0;        // the solver ran this day: snow-free cell-steps keep its values
    const double NA = na_real();
    const bool inr = c < N;
    const int64_t cc = inr ? c : N - 1;          // lanes past the raster read the last cell and write nothing
    const int k0 = day * 24;
    const int nh = min(24, a.tsteps - k0);       // (the one-shot entries: a last day may be short; its Tzd is NA, cpp:4700-4711)
    // ---- the workgroup-day's cell table
    if (AF) {
        if (wv == 0) {
            const SunCell sc = sun_cell(a.lats[cc], a.lons[cc]);
            s_mc[MC_SINLAT][lane] = sc.sinlat; s_mc[MC_COSLAT][lane] = sc.coslat; s_mc[MC_COSB][lane] = sc.cosB; s_mc[MC_SINB][lane] = sc.sinB;
            s_mc[MC_MXTC][lane] = a.mxtc[cc];
        }
        for (int h = wv; h < nh; h += 4) s_prec[h][lane] = a.precip[cc + N * (int64_t)(sub * 24 + h)];
    }
    if (wv == 0) {
        const double hgt = a.hgt[cc], pai = a.pai[cc], leafd = a.leafd[cc];
        s_mc[MC_HGT][lane] = hgt; s_mc[MC_PAI][lane] = pai; s_mc[MC_LEAFD][lane] = leafd;
        s_mc[MC_IHGT][lane] = gdiv(1.0, hgt); s_mc[MC_ILEAFD][lane] = gdiv(1.0, leafd); s_mc[MC_IPAI][lane] = gdiv(1.0, pai);
        s_mc[MC_PAIA][lane] = a.paia[cc]; s_mc[MC_LTRA][lane] = a.leaft[cc];
    } else if (wv == 1) {
        const double slope = a.slope[cc];
        s_mc[MC_SLOPE][lane] = slope;
        s_mc[MC_CS][lane] = cos(slope * kToRad); s_mc[MC_SS][lane] = sin(slope * kToRad);
        s_mc[MC_LEAFDEN][lane] = a.leafden[cc]; s_mc[MC_SVFA][lane] = a.skyview[cc];
    } else if (wv == 2) {
        const double aspect = a.aspect[cc], clump = a.clump[cc];
        s_mc[MC_CA][lane] = cos(aspect * kToRad); s_mc[MC_SA][lane] = sin(aspect * kToRad);
        s_mc[MC_CLUMP][lane] = clump;
        s_mc[MC_LNCLUMP][lane] = clump > 0.0 ? glog(clump) : 0.0;
    } else {
        // snowdayan's daily mean of the ground-snow temperature (its NA test looks at the FIRST step of the whole series,
        // cpp:4700; a chunk sees its own days only: the first step of the day — equal unless a cell's ground-snow
        // temperature turns NA part way, which gridmodelsnow never does)
        double Tzd = NA;
        if (nh == 24 && !isnan(a.sTg[cc + N * k0])) {
            double sumd = 0.0;
            for (int h = 0; h < 24; ++h) sumd += a.sTg[cc + N * (k0 + h)];
            Tzd = sumd / 24.0;
        }
        s_mc[MC_TZD][lane] = Tzd;
        s_mc[MC_MEAND][lane] = a.meanD[cc];
        s_mc[MC_SMAX][lane] = a.Smax ? a.Smax[cc] : 0.0;
    }
    __syncthreads();
    if (!inr) return;
    // the cell's place in the ring: its tile's block of this day, then the solver's lane position of (cell, hour)
    // (cpb == 0: the linear [step][cell] view of the one-shot entries' output arrays, RingView's convention)
    const int cpb = q.ring.cpb;
    const uint32_t tile = cpb ? (uint32_t)c / (uint32_t)cpb : 0u;
    const int cell = cpb ? (int)((uint32_t)c - tile * (uint32_t)cpb) : 0;
    const int64_t blk = cpb ? (int64_t)tile * q.ring.tile_stride + (int64_t)day * q.ring.day_stride : c + N * (int64_t)k0;
    for (int h = wv; h < nh; h += 4) {
        // (an opaque lane index per hour: the table is read where a value is used, not hoisted into registers in front of the loop;
        // an opaque block offset: or the ten variables' per-lane store addresses are kept across the loop — twenty registers)
        int li = lane;
        asm volatile("" : "+v"(li));
        int64_t bo = blk;
        asm volatile("" : "+v"(bo));
        // (opaque scalars too: or each variable's presence / selection masks, rank and base — 80 SGPRs — are hoisted in front of
        // the loop as invariants and spilled to VGPR lanes, a v_readlane per use; made at each store they are a few scalar
        // instructions beside the vector work)
        uint32_t held = q.held, selm = q.sel;
        int64_t vs = q.vstride;
        asm volatile("" : "+s"(held), "+s"(selm), "+s"(vs));
        auto has = [&](int i) { return (held >> i) & 1u; };
        auto put = [&](int i, int hh, double v) {
            const int rank = __builtin_popcount(held & ((1u << i) - 1u));
            q.base0[bo + rank * vs + (cpb ? (int64_t)mcf::ring_pos(cpb, cell, hh) : N * (int64_t)hh)] = v;
        };
        auto MC = [&](int f) { return s_mc[f][li]; };
        const double hgt = MC(MC_HGT);
        if (isnan(hgt)) {            // cpp:4988-4989: the cell is skipped — the blank template's NA unless the solver wrote it
            if (!keep) {
#pragma unroll
                for (int i = 0; i < MCF_NOUT; ++i) if (has(i)) put(i, h, NA);
            }
            continue;
        }
        const int64_t o = c + N * (k0 + h);          // chunk-local
        const int f = sub * 24 + h;                   // step of the snow-day subset series
        const double swe = a.swe[o];
        if (!(swe > 0.0)) {                           // cpp:4993
            if (!keep) {
#pragma unroll
                for (int i = 0; i < MCF_NOUT; ++i) if (has(i)) put(i, h, NA);
            }
            continue;
        }
        const double sdepg = a.sdepg[o], sTg = a.sTg[o];
        const double reqhgts = a.reqhgt - sdepg;
        // an output of this cell-step into the ring: gridmicrosnow1's `out` mask, the blank template's NA otherwise
        auto emit = [&](int i, double val) {
            if (!has(i)) return;
            if ((selm >> i) & 1u) put(i, h, val);
            else if (!keep) put(i, h, NA);
        };
        double Tz, tleaf, rh;
        if (reqhgts >= 0.0) {
            SiteK site;
            site.cS = MC(MC_CS); site.sS = MC(MC_SS); site.cA = MC(MC_CA); site.sA = MC(MC_SA); site.flat = MC(MC_SLOPE) == 0.0;
            MicroIn mi;
            mi.reqhgt = reqhgts; mi.zref = a.zref;
            mi.cell = &s_mc[0][li]; mi.cs = 64;           // (MC_HGT .. MC_IPAI are MQ_HGT .. MQ_IPAI)
            mi.Tg = sTg; mi.Tc = a.sTc[o]; mi.sden = a.sden[o]; mi.sdepg = sdepg;
            mi.sdepc = swe / mi.sden;
            if (AF) {
                const DateRow2 dr = td[f];
                SunCell sc;
                sc.sinlat = MC(MC_SINLAT); sc.coslat = MC(MC_COSLAT); sc.cosB = MC(MC_COSB); sc.sinB = MC(MC_SINB);
                int sindex;
                const SunT sun = sun_at_cell(dr.sindec, dr.cosdec, dr.cosA, dr.sinA, sc, sindex);
                mi.si = solar_index(sun, site, true);
                if (isnan(mi.si)) mi.si = sun.cosz;                        // cpp:5161
                mi.shadowmask = a.hor[(int64_t)sindex * N + c] > sun.tansa ? 0 : 1;
                mi.ws = a.wsa[(int64_t)dr.windex * N + c];
                const int64_t fo = c + N * (int64_t)f;                      // the lane's step of the (subset) weather series
                mi.tc = a.temp[fo]; mi.pk = a.pres[fo]; mi.u2 = a.windspeed[fo];
                mi.Rsw = a.swdown[fo]; mi.Rdif = a.difrad[fo]; mi.Rlw = a.lwdown[fo]; mi.umu = a.umu[fo];
                // the albedo clock (snowalbCpp, cpp:3733-3739): hours since the last snowfall — the day's start value, then back
                // through the day's precipitation to the lane's hour
                int hs = a.hs0[c + N * (int64_t)sub] + h;
                for (int j = h; j >= 1; --j)
                    if (s_prec[j][li] > 0) { hs = h - j; break; }
                mi.alb = snow_albedo(hs);
                mi.ialb = gdiv(1.0, mi.alb);
                const MicroMet mm = micro_met(mi.tc, a.relhum[fo], mi.pk, MC(MC_MXTC));
                const MicroOut mo = micro_above(mi, mm, sun, [&](int i, double val) { emit(i, val); return true; });
                Tz = mo.Tz; tleaf = mo.tleaf; rh = mo.rh;
            } else {
                const MicroStep& r = tb[f];
                const SunT sun = r.s;
                mi.si = solar_index(sun, site, true);
                if (isnan(mi.si)) mi.si = sun.cz;                          // cpp:5002
                mi.shadowmask = a.hor[(int64_t)r.sindex * N + c] > sun.tansa ? 0 : 1;
                mi.ws = a.wsa[(int64_t)r.windex * N + c];
                mi.tc = r.tc; mi.pk = r.pk; mi.u2 = r.u2;
                mi.Rsw = r.rsw; mi.Rdif = r.rdif; mi.Rlw = r.rlw; mi.umu = r.umu;
                mi.alb = r.alb; mi.ialb = r.ialb;
                // (wind speed and the five radiation streams go into the ring where micro_above has them final)
                const MicroOut mo = micro_above(mi, r.mm, sun, [&](int i, double val) { emit(i, val); return true; });
                Tz = mo.Tz; tleaf = mo.tleaf; rh = mo.rh;
            }
        } else {
            const double b = micro_below(reqhgts, MC(MC_MEAND), sTg, MC(MC_TZD), a.mat, a.hiy);
            Tz = b; tleaf = b; rh = 100.0;
            for (int i = 4; i < MCF_NOUT; ++i) emit(i, 0.0);
        }
        emit(0, Tz); emit(1, tleaf); emit(2, rh);
        emit(3, MC(MC_SMAX));
    }
}

// Round 5 shape, for the solver's 21-cell tiles: a workgroup is THREE tiles (63 consecutive cells, lane 63 idle) x EVERY snow
// day of the chunk; eight waves, wave w takes the hours w, w + 8, w + 16 of each day (the hour stays wave-uniform: the step
// record comes through scalar loads as before).  What that buys (profiles/r04_c4_aux_pmc_summary.json had 1.40 x the values
// written and 1.47 x the series read):
//   * the horizon and wind-shelter planes (24 + 8 values per cell; one of each was fetched per cell-STEP: 16 B beside the
//     40 B of snow series) and the cell table are staged in LDS ONCE per workgroup = once per chunk;
//   * every store is a whole 128-byte line.  Of a tile-day block's four lines per hour group (mcf_kernels.h ring_pos: three
//     hours x cells 0-15, then cells 16-20 of the three hours) the first three are one wave's sixteen consecutive lanes; the
//     fourth was written 40 B at a time by three different waves.  Its values now meet in LDS (s_tail) and are flushed behind
//     the day's barrier, a line per sixteen lanes.  A slot nobody produced (a snow-free cell-step of a day the solver ran as
//     well: its value stays) holds a sentinel and is not stored.
constexpr int kRtCells = 63, kRtWaves = MCF_MICRORING_WAVES == 3 ? 6 : 8;      // two workgroups per CU (56 KB of LDS each): 4 or 3 waves per SIMD
constexpr unsigned long long kTailEmpty = 0x7FF8A5A5DEAD0001ULL;      // (a NaN payload no arithmetic produces)
__global__ __launch_bounds__(64 * kRtWaves, MCF_MICRORING_WAVES) void k_microsnow_tiles(MicroRingArgs q, const MicroStep* __restrict__ tb,
                                                                                       const int32_t* __restrict__ tb_daymap,
                                                                                       const int32_t* __restrict__ tb_nosnow) {
    snow::snow_tables_init();
    __shared__ double s_mc[MC_COUNT][64];
    __shared__ double s_hw[32][64];                   // 24 horizon + 8 wind-shelter planes of the workgroup's cells
    __shared__ double s_tail[MCF_NOUT * 3 * 8 * 16];   // [held variable][tile][hour group][15 values + padding]
    constexpr int kTzdDays = 8;
    __shared__ double s_tzd[kTzdDays][64];            // the days' mean ground-snow temperatures (chunks of up to eight days)
    const MicroArgs& a = q.m;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t N = a.N;
    const int64_t c0 = (int64_t)blockIdx.x * kRtCells;     // uniform
    const bool inr = lane < kRtCells && c0 + lane < N;
    const double NA = na_real();
    {
        const int64_t cc = c0 + lane < N ? c0 + lane : N - 1;             // idle lanes read the last cell and write nothing
        // ---- once per workgroup: the cell table (a part by each of the first three waves) and the 32 direction planes
        if (wv == 0) {
            const double hgt = a.hgt[cc], pai = a.pai[cc], leafd = a.leafd[cc];
            s_mc[MC_HGT][lane] = hgt; s_mc[MC_PAI][lane] = pai; s_mc[MC_LEAFD][lane] = leafd;
            s_mc[MC_IHGT][lane] = gdiv(1.0, hgt); s_mc[MC_ILEAFD][lane] = gdiv(1.0, leafd); s_mc[MC_IPAI][lane] = gdiv(1.0, pai);
            s_mc[MC_PAIA][lane] = a.paia[cc]; s_mc[MC_LTRA][lane] = a.leaft[cc];
        } else if (wv == 1) {
            const double slope = a.slope[cc];
            s_mc[MC_SLOPE][lane] = slope;
            s_mc[MC_CS][lane] = cos(slope * kToRad); s_mc[MC_SS][lane] = sin(slope * kToRad);
            s_mc[MC_LEAFDEN][lane] = a.leafden[cc]; s_mc[MC_SVFA][lane] = a.skyview[cc];
        } else if (wv == 2) {
            const double aspect = a.aspect[cc], clump = a.clump[cc];
            s_mc[MC_CA][lane] = cos(aspect * kToRad); s_mc[MC_SA][lane] = sin(aspect * kToRad);
            s_mc[MC_CLUMP][lane] = clump;
            s_mc[MC_LNCLUMP][lane] = clump > 0.0 ? glog(clump) : 0.0;
            s_mc[MC_MEAND][lane] = a.meanD[cc];
            s_mc[MC_SMAX][lane] = a.Smax ? a.Smax[cc] : 0.0;
        } else {
            for (int p = wv - 3; p < 32; p += kRtWaves - 3) s_hw[p][lane] = p < 24 ? a.hor[(int64_t)p * N + cc] : a.wsa[(int64_t)(p - 24) * N + cc];
        }
        // ... and every snow day's mean ground-snow temperature (see k_microsnow_ring), a day per wave from the last wave down:
        // made per day by one wave in front of the day's barrier, its 24 loads were what the other seven waited for (round 5:
        // -1.9 % on configs[4]'s snow-day stage)
        if (q.ndays <= kTzdDays) {
            for (int day = kRtWaves - 1 - wv; day < q.ndays; day += kRtWaves) {
                if (tb_daymap[day] < 0) continue;
                const int64_t ci = c0 + lane < N ? lane : N - 1 - c0;
                if (a.tzd) {        // the snow model's kernel made them as it wrote the series (ModelArgs::tzd): one value, not 24
                    s_tzd[day][lane] = a.tzd[N * (int64_t)day + c0 + ci];
                    continue;
                }
                const double* tg = a.sTg + (N * (int64_t)(day * 24) + c0);
                double v[24];
#pragma unroll
                for (int h = 0; h < 24; ++h) v[h] = tg[ci + N * h];
                double Tzd = NA;
                if (!isnan(v[0])) {
                    double sumd = 0.0;
#pragma unroll
                    for (int h = 0; h < 24; ++h) sumd += v[h];
                    Tzd = sumd / 24.0;
                }
                s_tzd[day][lane] = Tzd;
            }
        }
    }
    // The lane's place: in the raster a uniform base + its lane number; in the tiled ring the block of its tile (uniform base of
    // the workgroup's first tile + tl tile strides) and, for cells 0-15, its column of the hour's line; cells 16-20 go to s_tail.
    // All as 32-bit byte offsets against uniform bases: no 64-bit vector address arithmetic per load or store.
    const int tl = lane / 21, cl = lane - 21 * tl;
    const bool direct = cl < 16;
    uint32_t rofs = (uint32_t)(tl * (int)q.ring.tile_stride + cl) * 8u;            // (a tile's days x variables x 512 doubles: < 2^29)
    const int tofs = tl * 128 + (cl - 16);
    uint32_t lofs = (uint32_t)lane * 8u;
    const int nheld = __builtin_popcount(q.held);
    for (int day = 0; day < q.ndays; ++day) {
        const int sub = tb_daymap[day];               // uniform: scalar loads
        if (sub < 0) continue;
        const bool keep = tb_nosnow[day] != 0;        // the solver ran this day: snow-free cell-steps keep its values
        const int k0 = day * 24;
        // the day's mean ground-snow temperature (see k_microsnow_ring) by the last wave; the tails' slots emptied by the threads
        // that flushed them (same thread, same slots: ordered without a barrier)
        if (q.ndays > kTzdDays && wv == kRtWaves - 1) {
            const double* tg = a.sTg + (N * k0 + c0);
            const int64_t cc = c0 + lane < N ? lane : N - 1 - c0;
            double Tzd = NA;
            if (!isnan(tg[cc])) {
                double sumd = 0.0;
                for (int h = 0; h < 24; ++h) sumd += tg[cc + N * h];
                Tzd = sumd / 24.0;
            }
            s_mc[MC_TZD][lane] = Tzd;
        }
        const double* const tzd_row = q.ndays <= kTzdDays ? &s_tzd[day][0] : &s_mc[MC_TZD][0];
        for (int e = tid; e < nheld * 384; e += 64 * kRtWaves) reinterpret_cast<unsigned long long*>(s_tail)[e] = kTailEmpty;
        __syncthreads();
        // the workgroup's first tile's block of this day (uniform)
        double* const dayblk = q.base0 + ((int64_t)blockIdx.x * 3) * q.ring.tile_stride + (int64_t)day * q.ring.day_stride;
        if (inr) {
            for (int h = wv; h < 24; h += kRtWaves) {
                int li = lane;
                asm volatile("" : "+v"(li));
                uint32_t held = q.held, selm = q.sel;
                int64_t vs = q.vstride;
                asm volatile("" : "+s"(held), "+s"(selm), "+s"(vs));
                auto has = [&](int i) { return (held >> i) & 1u; };
                const int hg = h / 3, hm = h - 3 * hg;               // uniform
                auto put = [&](int i, double v) {
                    const int rank = __builtin_popcount(held & ((1u << i) - 1u));
                    if (direct) {
                        asm("" : "+v"(rofs));       // (keeps the offset's zero-extension in the store's own block: mcf_kernels.hip `put`)
                        *(double*)((char*)(dayblk + (rank * vs + (64 * hg + 16 * hm))) + rofs) = v;
                    } else {
                        s_tail[(rank * 384 + hg * 16 + 5 * hm) + tofs] = v;
                    }
                };
                auto MC = [&](int f) { return s_mc[f][li]; };
                const double hgt = MC(MC_HGT);
                if (isnan(hgt)) {            // cpp:4988-4989
                    if (!keep) {
#pragma unroll
                        for (int i = 0; i < MCF_NOUT; ++i) if (has(i)) put(i, NA);
                    }
                    continue;
                }
                // the step's plane of each snow series (uniform) + the lane
                const int64_t po = N * (k0 + h) + c0;
                auto ser = [&](const double* p) {
                    asm("" : "+v"(lofs));
                    return *(const double*)((const char*)(p + po) + lofs);
                };
                const int f = sub * 24 + h;                   // step of the snow-day subset series
                const double swe = ser(a.swe);
                if (!(swe > 0.0)) {                           // cpp:4993
                    if (!keep) {
#pragma unroll
                        for (int i = 0; i < MCF_NOUT; ++i) if (has(i)) put(i, NA);
                    }
                    continue;
                }
                const double sdepg = ser(a.sdepg), sTg = ser(a.sTg);
                const double reqhgts = a.reqhgt - sdepg;
                auto emit = [&](int i, double val) {
                    if (!has(i)) return;
                    if ((selm >> i) & 1u) put(i, val);
                    else if (!keep) put(i, NA);
                };
                double Tz, tleaf, rh;
                if (reqhgts >= 0.0) {
                    const MicroStep& r = tb[f];
                    const SunT sun = r.s;
                    SiteK site;
                    site.cS = MC(MC_CS); site.sS = MC(MC_SS); site.cA = MC(MC_CA); site.sA = MC(MC_SA); site.flat = MC(MC_SLOPE) == 0.0;
                    MicroIn mi;
                    mi.si = solar_index(sun, site, true);
                    if (isnan(mi.si)) mi.si = sun.cz;                          // cpp:5002
                    mi.shadowmask = s_hw[r.sindex][li] > sun.tansa ? 0 : 1;
                    mi.ws = s_hw[24 + r.windex][li];
                    mi.reqhgt = reqhgts; mi.zref = a.zref;
                    mi.tc = r.tc; mi.pk = r.pk; mi.u2 = r.u2;
                    mi.Rsw = r.rsw; mi.Rdif = r.rdif; mi.Rlw = r.rlw; mi.umu = r.umu;
                    mi.cell = &s_mc[0][li]; mi.cs = 64;
                    mi.Tg = sTg; mi.Tc = ser(a.sTc); mi.sden = ser(a.sden); mi.sdepg = sdepg;
                    mi.sdepc = swe / mi.sden;
                    mi.alb = r.alb; mi.ialb = r.ialb;
                    const MicroOut mo = micro_above(mi, r.mm, sun, [&](int i, double val) { emit(i, val); return true; });
                    Tz = mo.Tz; tleaf = mo.tleaf; rh = mo.rh;
                } else {
                    const double b = micro_below(reqhgts, MC(MC_MEAND), sTg, tzd_row[li], a.mat, a.hiy);
                    Tz = b; tleaf = b; rh = 100.0;
                    for (int i = 4; i < MCF_NOUT; ++i) emit(i, 0.0);
                }
                emit(0, Tz); emit(1, tleaf); emit(2, rh);
                emit(3, MC(MC_SMAX));
            }
        }
        __syncthreads();
        // flush: element e = (rank * 3 + tile) * 128 + hour group * 16 + slot -> the fourth line of that block's hour group
        for (int e = tid; e < nheld * 384; e += 64 * kRtWaves) {
            const unsigned long long bits = reinterpret_cast<const unsigned long long*>(s_tail)[e];
            if (bits == kTailEmpty) continue;
            const int slot = e & 15, g = (e >> 4) & 7, rt = e >> 7, rank = rt / 3, t3 = rt - 3 * rank;
            dayblk[(int64_t)t3 * q.ring.tile_stride + rank * q.vstride + 64 * g + 48 + slot] = __longlong_as_double((long long)bits);
        }
    }
}

// ---- .snowmodel1's chunk loop (R/internal.R "int:" 2553-2617) ---------------------------------------
// albedo clock restarted at every chunk start: each gridmodelsnow1 call runs snowalbCpp on its own slice
__global__ void k_snow_alb_chunks(StepRow* rows, const double* precip, int tsteps, int chunk, int nchunks) {
    snow::snow_tables_init();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int k0 = c * chunk, k1 = min(tsteps, k0 + chunk);
    int hs = 0;
    for (int k = k0; k < k1; ++k) {
        if (k > k0) hs = precip[k] > 0 ? 0 : hs + 1;
        rows[k].m.alb = snow_albedo(hs);
        rows[k].m.ialb = gdiv(1.0, rows[k].m.alb);
    }
}
// dtms = dtm + ground snow depth (int:2562, 2614); NaN where the dtm is NA
__global__ __launch_bounds__(256) void k_add_snow(const double* __restrict__ dtm, const double* __restrict__ dep,
                                                  double scale, int64_t N, double* __restrict__ dtms) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < N) dtms[c] = dtm[c] + dep[c] * scale;
}
// mask(x, dtm): NA where the dtm is NA (int:2568, 2571)
__global__ __launch_bounds__(256) void k_mask2(const double* __restrict__ dtm, int64_t N, double* __restrict__ a,
                                               double* __restrict__ b) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < N && isnan(dtm[c])) { a[c] = na_real(); b[c] = na_real(); }
}
// deterministic (sum, count) of the non-NaN entries of x: kSumParts workgroups reduce fixed strided subsets with a
// fixed-shape tree, workgroup 0 of a second launch adds the partials in order (same result on every run)
constexpr int kSumParts = 128;
__global__ __launch_bounds__(256) void k_sumcount_part(const double* __restrict__ x, int64_t N, double* __restrict__ ws) {
    __shared__ double ss[256];
    __shared__ double sc[256];
    double s = 0.0, n = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)kSumParts * 256) {
        const double v = x[i];
        if (!isnan(v)) { s += v; n += 1.0; }
    }
    ss[threadIdx.x] = s; sc[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { ss[threadIdx.x] += ss[threadIdx.x + w]; sc[threadIdx.x] += sc[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ws[2 * blockIdx.x] = ss[0]; ws[2 * blockIdx.x + 1] = sc[0]; }
}
__global__ void k_sumcount_fin(const double* __restrict__ ws, double* __restrict__ out2) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double s = 0.0, n = 0.0;
    for (int p = 0; p < kSumParts; ++p) { s += ws[2 * p]; n += ws[2 * p + 1]; }
    out2[0] = s; out2[1] = n;
}
// ws: 2 * kSumParts doubles of scratch
void launch_sumcount(const double* x, int64_t N, double* ws, double* out2) {
    hipLaunchKernelGGL(k_sumcount_part, dim3(kSumParts), dim3(256), 0, nullptr, x, N, ws);
    hipLaunchKernelGGL(k_sumcount_fin, dim3(1), dim3(64), 0, nullptr, ws, out2);
}
// .tpicalc (int:2471-2485) on a row block of the raster.  `z` is the block's surface (dtm + ground snow) with
// `hn` halo rows above: row b of z is global row row0 - hn + b; RB rows in all.
struct TpiGeo {
    int64_t rows, cols, RB, hn, row0, rows_total;
    int af;
    int64_t I0, nI, nJ, NItot;   // coarse rows [I0, I0 + nI) are held, NItot in the whole raster
};
// coarse part: aggregate(dtm, af, na.rm = TRUE) — af x af block means from the raster's top-left corner
// over the non-NA cells
__global__ __launch_bounds__(256) void k_tpi_coarse(const double* __restrict__ z, TpiGeo g, double* __restrict__ cm) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= g.nI * g.nJ) return;
    const int64_t Il = q % g.nI, J = q / g.nI;
    const int64_t ra = (g.I0 + Il) * g.af, rb = min(ra + g.af, g.rows_total), ca = J * g.af, cb = min(ca + g.af, g.cols);
    double s = 0.0, n = 0.0;
    for (int64_t c = ca; c < cb; ++c)
        for (int64_t r = ra; r < rb; ++r) {
            const double v = z[(r - (g.row0 - g.hn)) + g.RB * c];
            if (!isnan(v)) { s += v; n += 1.0; }
        }
    cm[q] = s / n;   // 0/0 = NaN for an all-NA block
}
// fine part: resample (bilinear between block centres, clamped) or the raster mean, then
// tpic = exp((dtmc - dtm) * tfact) with its two clamps (`tpic[tpic < 0.05] <- 0.1` sic)
__global__ __launch_bounds__(256) void k_tpi_fine(const double* __restrict__ z, TpiGeo g, const double* __restrict__ cm,
                                                  double surface_mean, double tfact, double* __restrict__ tpic) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.rows * g.cols) return;
    const int64_t r = cell % g.rows, c = cell / g.rows;
    const double z0 = z[(g.hn + r) + g.RB * c];
    double zc;
    if (cm) {
        const int af = g.af;
        const double tr = ((double)(g.row0 + r) - (af - 1) / 2.0) / af, tcc = ((double)c - (af - 1) / 2.0) / af;
        const int64_t i0 = (int64_t)floor(tr), j0 = (int64_t)floor(tcc);
        const double wr = tr - (double)i0, wc = tcc - (double)j0;
        const int64_t ia = min(max(i0, (int64_t)0), g.NItot - 1) - g.I0, ib = min(max(i0 + 1, (int64_t)0), g.NItot - 1) - g.I0;
        const int64_t ja = min(max(j0, (int64_t)0), g.nJ - 1), jb = min(max(j0 + 1, (int64_t)0), g.nJ - 1);
        const double top = cm[ia + g.nI * ja] * (1 - wc) + cm[ia + g.nI * jb] * wc;
        const double bot = cm[ib + g.nI * ja] * (1 - wc) + cm[ib + g.nI * jb] * wc;
        zc = top * (1 - wr) + bot * wr;
    } else {
        zc = z0 * 0 + surface_mean;   // dtm * 0 + mean(dtm, na.rm = TRUE)
    }
    double t = exp((zc - z0) * tfact);
    if (t < 0.05) t = 0.1;
    if (t > 10) t = 10;
    tpic[cell] = t;
}
// redistribution of the chunk's snow-depth changes and hand-over to the next chunk (int:2589-2614);
// sdepc / sdepg are overwritten by totalSWE / groundsnowdepth
// applycpp3 (cpp:5553-5588): `parts` workgroups per time step stream fixed strided subsets of the cells (coalesced),
// fixed-shape tree in LDS, partials combined in order by a second kernel -> deterministic; NaN cells are skipped
__device__ __forceinline__ double apply3_combine(int fun, double a, double b) {
    if (fun < 2) return a + b;
    if (fun == 2) return b > a ? b : a;
    return b < a ? b : a;
}
__global__ __launch_bounds__(256) void k_apply3_part(const double* __restrict__ a, int64_t N, int fun, int parts,
                                                     double* __restrict__ ws /* [tsteps][parts][2] */) {
    __shared__ double sv[256];
    __shared__ double sn[256];
    const int64_t k = blockIdx.y;
    const double* x = a + k * N;
    double v = fun == 2 ? -INFINITY : (fun == 3 ? INFINITY : 0.0), n = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)parts * 256) {
        const double q = x[i];
        if (isnan(q)) continue;
        n += 1.0;
        v = apply3_combine(fun, v, q);
    }
    sv[threadIdx.x] = v; sn[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sv[threadIdx.x] = apply3_combine(fun, sv[threadIdx.x], sv[threadIdx.x + w]);
            sn[threadIdx.x] += sn[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[2 * (k * parts + blockIdx.x)] = sv[0];
        ws[2 * (k * parts + blockIdx.x) + 1] = sn[0];
    }
}
// max AND min of every step in ONE pass over the series (the snow plan asks for both of every chunk: snowdaysfun's operands).
// Each of the two is folded over the same cells in the same order as k_apply3_part folds it alone — same bits.
__global__ __launch_bounds__(256) void k_apply3_minmax_part(const double* __restrict__ a, int64_t N, int parts,
                                                            double* __restrict__ ws_max, double* __restrict__ ws_min) {
    __shared__ double smx[256];
    __shared__ double smn[256];
    __shared__ double sn[256];
    const int64_t k = blockIdx.y;
    const double* x = a + k * N;
    double mx = -INFINITY, mn = INFINITY, n = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)parts * 256) {
        const double q = x[i];
        if (isnan(q)) continue;
        n += 1.0;
        mx = apply3_combine(2, mx, q);
        mn = apply3_combine(3, mn, q);
    }
    smx[threadIdx.x] = mx; smn[threadIdx.x] = mn; sn[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            smx[threadIdx.x] = apply3_combine(2, smx[threadIdx.x], smx[threadIdx.x + w]);
            smn[threadIdx.x] = apply3_combine(3, smn[threadIdx.x], smn[threadIdx.x + w]);
            sn[threadIdx.x] += sn[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t o = 2 * (k * parts + blockIdx.x);
        ws_max[o] = smx[0]; ws_max[o + 1] = sn[0];
        ws_min[o] = smn[0]; ws_min[o + 1] = sn[0];
    }
}
__global__ __launch_bounds__(256) void k_apply3_fin(const double* __restrict__ ws, int64_t tsteps, int fun, int parts,
                                                    double* __restrict__ result, double* __restrict__ count) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= tsteps) return;
    double v = fun == 2 ? -INFINITY : (fun == 3 ? INFINITY : 0.0), n = 0.0;
    for (int p = 0; p < parts; ++p) {
        v = apply3_combine(fun, v, ws[2 * (k * parts + p)]);
        n += ws[2 * (k * parts + p) + 1];
    }
    if (fun == 0) v = n > 0 ? v / n : NAN;
    result[k] = v;
    if (count) count[k] = n;
}

// ---- host side -------------------------------------------------------------------------------------
#define S_TRY(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                     __FILE__, __LINE__);                                                    \
            return mcf::api_fail(e_ == hipErrorOutOfMemory ? MCF_ERR_NOMEM : MCF_ERR_HIP, b_); \
        }                                                                                    \
    } while (0)

struct Events {   // timing events released on every exit path
    std::vector<hipEvent_t> e;
    ~Events() { for (hipEvent_t x : e) (void)hipEventDestroy(x); }
    hipError_t make(int n) {
        for (int i = 0; i < n; ++i) {
            hipEvent_t x;
            hipError_t r = hipEventCreate(&x);
            if (r != hipSuccess) return r;
            e.push_back(x);
        }
        return hipSuccess;
    }
};

struct Bufs {
    std::vector<void*> p;
    int64_t bytes = 0;
    ~Bufs() { release_all(); }
    void release_all() { for (void* q : p) (void)hipFree(q); p.clear(); bytes = 0; }
    int alloc(void** out, int64_t n) {
        if (n <= 0) n = 8;
        hipError_t e = hipMalloc(out, (size_t)n);
        if (e != hipSuccess)
            return mcf::api_fail(MCF_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        p.push_back(*out);
        bytes += n;
        return MCF_OK;
    }
    // device copy of a host array of `n` elements of size `esz`
    template <class T>
    int up(const T** dev, const T* host, int64_t n, const char* what) {
        if (!host) return mcf::api_fail(MCF_ERR_ARG, std::string("null input: ") + what);
        void* d;
        int rc = alloc(&d, n * (int64_t)sizeof(T));
        if (rc) return rc;
        hipError_t e = hipMemcpy(d, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return mcf::api_fail(MCF_ERR_HIP, std::string("upload failed: ") + what);
        *dev = (const T*)d;
        return MCF_OK;
    }
};
// device -> caller-owned pageable memory; large results through the pinned ring + copy threads (mcf_hostpipe.hpp)
struct Downloader {
    mcf::HostPipe pipe;
    bool tried = false, ok = false;
    // rows of `width` bytes, contiguous on the device, `dpitch` bytes apart on the host (a row block's series)
    hipError_t get_pitched(void* dst, size_t dpitch, const void* dev, size_t width, size_t height) {
        static const bool no_pipe = getenv("MCF_NO_HOSTPIPE") != nullptr;
        if (width * height >= ((size_t)64 << 20) && width <= mcf::HostPipe::kPiece && !no_pipe) {
            if (!tried) { tried = true; ok = pipe.init(); }
            if (ok) {
                hipError_t e = hipDeviceSynchronize();      // the producers ran on the null stream
                if (e != hipSuccess) return e;
                return pipe.copy_pitched(dst, dpitch, dev, width, height, nullptr);
            }
        }
        return hipMemcpy2D(dst, dpitch, dev, width, width, height, hipMemcpyDeviceToHost);
    }
    hipError_t get(void* dst, const void* dev, size_t bytes) {
        static const bool no_pipe = getenv("MCF_NO_HOSTPIPE") != nullptr;
        if (bytes >= ((size_t)64 << 20) && !no_pipe) {
            if (!tried) { tried = true; ok = pipe.init(); }
            if (ok) {
                hipError_t e = hipDeviceSynchronize();      // the producers ran on the null stream
                if (e != hipSuccess) return e;
                return pipe.copy(dst, dev, bytes, nullptr);
            }
        }
        return hipMemcpy(dst, dev, bytes, hipMemcpyDeviceToHost);
    }
};

#define UP(dst, src, n) do { if ((rc = b.up(&(dst), (src), (n), #src))) return rc; } while (0)

int pick_device(int32_t device) {
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return mcf::api_fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
    if (device < 0 || device >= nd) return mcf::api_fail(MCF_ERR_ARG, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return mcf::api_fail(MCF_ERR_HIP, "hipSetDevice failed");
    return MCF_OK;
}
int check_room(int64_t need) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return MCF_OK;
    if ((double)need > 0.95 * (double)fr) {
        char m[200];
        snprintf(m, sizeof m, "snow call needs %.2f GB of device memory, %.2f GB free; split the time range",
                 need / 1e9, fr / 1e9);
        return mcf::api_fail(MCF_ERR_NOMEM, m);
    }
    return MCF_OK;
}

void snow_density_params(int snowenv, double sdp[4]) {   // snowdenp, cpp:3741-3749
    static const double tab[5][4] = {{0.5975, 0.2237, 0.0012, 0.0038},
                                     {0.5979, 0.2578, 0.001, 0.0038},
                                     {0.594, 0.2332, 0.0016, 0.0031},
                                     {0.363, 0.2425, 0.0029, 0.0049},
                                     {0.217, 0.217, 0.0, 0.0}};
    if (snowenv < 0 || snowenv > 4) snowenv = 0;
    for (int i = 0; i < 4; ++i) sdp[i] = tab[snowenv][i];
}

int common_checks(const mcf_snow_inputs* in) {
    if (!in) return mcf::api_fail(MCF_ERR_ARG, "null snow inputs");
    if (in->rows <= 0 || in->cols <= 0 || in->tsteps <= 0) return mcf::api_fail(MCF_ERR_ARG, "bad snow dimensions");
    if (in->tsteps > (1 << 30)) return mcf::api_fail(MCF_ERR_ARG, "tsteps too large");
    // (the snow kernels address a step's slab as uniform base + 32-bit lane byte offset)
    if (in->rows * in->cols >= ((int64_t)1 << 29)) return mcf::api_fail(MCF_ERR_ARG, "at most 2^29 - 1 cells per snow call (row-tile larger rasters)");
    if (!in->obstime.year || !in->obstime.month || !in->obstime.day || !in->obstime.hour)
        return mcf::api_fail(MCF_ERR_ARG, "null obstime");
    return MCF_OK;
}

// fills the time-class device tables shared by both entry points
int build_step_tables(Bufs& b, const mcf_snow_inputs* in, bool af, bool model, bool degrees, const StepRow** rows,
                      const DateRow2** dates, const double** mxtc1) {
    const int T = (int)in->tsteps;
    int rc;
    StepArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.tsteps = T;
    UP(sa.year, in->obstime.year, T);
    UP(sa.month, in->obstime.month, T);
    UP(sa.day, in->obstime.day, T);
    UP(sa.hour, in->obstime.hour, T);
    UP(sa.winddir, in->clim.winddir, T);
    sa.lat = in->other.lat; sa.lon = in->other.lon;
    sa.degrees = degrees ? 1 : 0;
    const unsigned grid = (unsigned)((T + 255) / 256);
    if (af) {
        if ((rc = b.alloc((void**)&sa.dates, (int64_t)T * sizeof(DateRow2)))) return rc;
        hipLaunchKernelGGL(k_snow_dates, dim3(grid), dim3(256), 0, nullptr, sa);
        *dates = sa.dates;
        *rows = nullptr;
        *mxtc1 = nullptr;
    } else {
        UP(sa.temp, in->clim.temp, T);
        UP(sa.precip, in->clim.precip, T);
        if (model) {
            UP(sa.relhum, in->clim.relhum, T);
            UP(sa.pres, in->clim.pres, T);
            UP(sa.swdown, in->clim.swdown, T);
            UP(sa.difrad, in->clim.difrad, T);
            UP(sa.lwdown, in->clim.lwdown, T);
            UP(sa.windspeed, in->clim.windspeed, T);
            UP(sa.Gp, in->pointm.Gp, T);
            UP(sa.Tcp, in->pointm.Tc, T);
            UP(sa.RswabsG, in->pointm.RswabsG, T);
            UP(sa.RlwabsG, in->pointm.RlwabsG, T);
            UP(sa.umu, in->pointm.umu, T);
        }
        if ((rc = b.alloc((void**)&sa.rows, (int64_t)T * sizeof(StepRow)))) return rc;
        if ((rc = b.alloc((void**)&sa.mxtc, 8))) return rc;
        hipLaunchKernelGGL(k_snow_steps, dim3(grid), dim3(256), 0, nullptr, sa);
        if (model && T / 24 > 0)
            hipLaunchKernelGGL(k_snow_days, dim3((unsigned)((T / 24 + 63) / 64)), dim3(64), 0, nullptr, sa.rows, T);
        hipLaunchKernelGGL(k_snow_alb, dim3(1), dim3(64), 0, nullptr, sa.rows, sa.precip, sa.temp, T, sa.mxtc);
        *rows = sa.rows;
        *dates = nullptr;
        *mxtc1 = sa.mxtc;
    }
    if (hipGetLastError() != hipSuccess) return mcf::api_fail(MCF_ERR_HIP, "snow table kernels failed to launch");
    return MCF_OK;
}

int run_snowmodel(const mcf_snow_inputs* in, mcf_snowmodel_out* out, int32_t device, bool af) {
    int rc;
    if ((rc = common_checks(in))) return rc;
    if (!out) return mcf::api_fail(MCF_ERR_ARG, "null snow outputs");
    if ((in->array_forcing != 0) != af) return mcf::api_fail(MCF_ERR_ARG, "array_forcing does not match the entry point");
    if ((rc = pick_device(device))) return rc;
    const int64_t N = in->rows * in->cols;
    const int T = (int)in->tsteps;
    const int64_t NT = N * T;
    int nout3 = 0;
    double* host3[5] = {out->Tc, out->Tg, out->sdepc, out->sdepg, out->sden};
    for (double* p : host3) nout3 += p != nullptr;
    if ((rc = check_room((af ? 13 * NT : 0) * 8 + (int64_t)nout3 * NT * 8 + 60 * N * 8))) return rc;
    Bufs b;
    ModelArgs a;
    memset(&a, 0, sizeof a);
    a.N = N; a.tsteps = T; a.zref = in->other.zref;
    snow_density_params(in->snowenv, a.sdp);
    UP(a.pai, in->vegp.pai, N);
    UP(a.hgt, in->vegp.hgt, N);
    UP(a.leaft, in->vegp.leaft, N);
    UP(a.clump, in->vegp.clump, N);
    UP(a.slope, in->other.slope, N);
    UP(a.aspect, in->other.aspect, N);
    UP(a.skyview, in->other.skyview, N);
    UP(a.wsa, in->other.wsa, 8 * N);
    UP(a.hor, in->other.hor, 24 * N);
    UP(a.isnowdc, in->other.isnowdc, N);
    UP(a.isnowdg, in->other.isnowdg, N);
    UP(a.isnowac, in->other.isnowac, N);
    UP(a.isnowag, in->other.isnowag, N);
    const double* unused = nullptr;
    if ((rc = build_step_tables(b, in, af, true, !af, &a.rows, &a.dates, &unused))) return rc;
    if (af) {
        UP(a.lats, in->other.lats, N);
        UP(a.lons, in->other.lons, N);
        UP(a.temp, in->clim.temp, NT);
        UP(a.relhum, in->clim.relhum, NT);
        UP(a.pres, in->clim.pres, NT);
        UP(a.swdown, in->clim.swdown, NT);
        UP(a.difrad, in->clim.difrad, NT);
        UP(a.lwdown, in->clim.lwdown, NT);
        UP(a.windspeed, in->clim.windspeed, NT);
        UP(a.precip, in->clim.precip, NT);
        UP(a.Gp, in->pointm.Gp, NT);
        UP(a.Tcp, in->pointm.Tc, NT);
        UP(a.RswabsG, in->pointm.RswabsG, NT);
        UP(a.RlwabsG, in->pointm.RlwabsG, NT);
        UP(a.umu, in->pointm.umu, NT);
    }
    double** dev3[5] = {&a.Tc, &a.Tg, &a.sdepc, &a.sdepg, &a.sden};
    for (int v = 0; v < 5; ++v)
        if (host3[v] && (rc = b.alloc((void**)dev3[v], NT * 8))) return rc;
    double* host2[4] = {out->agec, out->ageg, out->meltc, out->meltg};
    double** dev2[4] = {&a.agec, &a.ageg, &a.meltc, &a.meltg};
    for (int v = 0; v < 4; ++v)
        if (host2[v] && (rc = b.alloc((void**)dev2[v], N * 8))) return rc;
    const unsigned grid = (unsigned)((N + 255) / 256);
    Events evs;
    const bool timing = getenv("MCF_TIMING") != nullptr;
    if (timing) { S_TRY(evs.make(2)); S_TRY(hipEventRecord(evs.e[0], nullptr)); }
    if (af) hipLaunchKernelGGL(k_snowmodel<true>, dim3(grid), dim3(256), 0, nullptr, a, a.rows, a.dates);
    else hipLaunchKernelGGL(k_snowmodel<false>, dim3(grid), dim3(256), 0, nullptr, a, a.rows, a.dates);
    S_TRY(hipGetLastError());
    if (timing) {
        S_TRY(hipEventRecord(evs.e[1], nullptr));
        S_TRY(hipEventSynchronize(evs.e[1]));
        float ms = 0;
        S_TRY(hipEventElapsedTime(&ms, evs.e[0], evs.e[1]));
        fprintf(stderr, "[mcf] k_snowmodel<%d>: %lld cells x %d steps in %.3f ms (%.3e cell-steps/s)\n", (int)af,
                (long long)N, T, ms, (double)NT / (ms * 1e-3));
    }
    Downloader dl;
    for (int v = 0; v < 5; ++v)
        if (host3[v]) S_TRY(dl.get(host3[v], *dev3[v], (size_t)NT * 8));
    for (int v = 0; v < 4; ++v)
        if (host2[v]) S_TRY(hipMemcpy(host2[v], *dev2[v], (size_t)N * 8, hipMemcpyDeviceToHost));
    S_TRY(hipDeviceSynchronize());
    return MCF_OK;
}

int run_microsnow(const mcf_snow_inputs* in, const mcf_snowm* sm, double reqhgt, double mat, const int32_t* outsel,
                  mcf_outputs* micro, int32_t device, bool af) {
    int rc;
    if ((rc = common_checks(in))) return rc;
    if (!sm || !outsel || !micro) return mcf::api_fail(MCF_ERR_ARG, "null gridmicrosnow argument");
    if ((in->array_forcing != 0) != af) return mcf::api_fail(MCF_ERR_ARG, "array_forcing does not match the entry point");
    for (int v = 0; v < MCF_NOUT; ++v)
        if (outsel[v] && !micro->var[v]) return mcf::api_fail(MCF_ERR_ARG, "requested micro variable has a null buffer");
    if (outsel[MCF_OUT_SOILM] && !in->other.Smax) return mcf::api_fail(MCF_ERR_ARG, "soilm requested but other$Smax is null");
    if ((rc = pick_device(device))) return rc;
    const int64_t N = in->rows * in->cols;
    const int T = (int)in->tsteps;
    const int64_t NT = N * T;
    const int nch = (T + 23) / 24;
    int nsel = 0;
    for (int v = 0; v < MCF_NOUT; ++v) nsel += outsel[v] != 0;
    if ((rc = check_room(((af ? 9 : 0) + 5 + nsel) * NT * 8 + 60 * N * 8))) return rc;
    Bufs b;
    MicroArgs a;
    memset(&a, 0, sizeof a);
    a.N = N; a.tsteps = T; a.reqhgt = reqhgt; a.mat = mat; a.zref = in->other.zref;
    const int y0 = in->obstime.year[0];
    a.hiy = (y0 % 4 == 0 && (y0 % 100 != 0 || y0 % 400 == 0)) ? 366 * 24 : 365 * 24;   // cpp:4984
    UP(a.pai, in->vegp.pai, N);
    UP(a.hgt, in->vegp.hgt, N);
    UP(a.leaft, in->vegp.leaft, N);
    UP(a.clump, in->vegp.clump, N);
    UP(a.paia, in->vegp.paia, N);
    UP(a.leafd, in->vegp.leafd, N);
    UP(a.leafden, in->vegp.leafden, N);
    UP(a.slope, in->other.slope, N);
    UP(a.aspect, in->other.aspect, N);
    UP(a.skyview, in->other.skyview, N);
    UP(a.wsa, in->other.wsa, 8 * N);
    UP(a.hor, in->other.hor, 24 * N);
    if (outsel[MCF_OUT_SOILM]) UP(a.Smax, in->other.Smax, N);
    if ((rc = build_step_tables(b, in, af, false, false, &a.rows, &a.dates, &a.mxtc1))) return rc;
    const int64_t F = af ? NT : T;
    if (af) {
        UP(a.lats, in->other.lats, N);
        UP(a.lons, in->other.lons, N);
    }
    UP(a.temp, in->clim.temp, F);
    UP(a.relhum, in->clim.relhum, F);
    UP(a.pres, in->clim.pres, F);
    UP(a.swdown, in->clim.swdown, F);
    UP(a.difrad, in->clim.difrad, F);
    UP(a.lwdown, in->clim.lwdown, F);
    UP(a.windspeed, in->clim.windspeed, F);
    UP(a.precip, in->clim.precip, F);
    UP(a.umu, in->clim.umu, F);
    UP(a.sTc, sm->Tc, NT);
    UP(a.sTg, sm->Tg, NT);
    UP(a.swe, sm->totalSWE, NT);
    UP(a.sdepg, sm->groundsnowdepth, NT);
    UP(a.sden, sm->snowden, NT);
    if ((rc = b.alloc((void**)&a.meanD, N * 8))) return rc;
    if (af) {
        if ((rc = b.alloc((void**)&a.mxtc, N * 8))) return rc;
        if ((rc = b.alloc((void**)&a.hs0, N * (int64_t)nch * 4))) return rc;
    }
    // The requested outputs in ONE buffer, NT apart (in/out: each starts as the no-snow solver's field): the linear view the
    // (cell, hour) kernel of the chunk loop writes through (k_microsnow_ring: RingView cpb = 0) — every day a snow day, every
    // snow-free cell-step kept.
    double* outbuf = nullptr;
    if ((rc = b.alloc((void**)&outbuf, (int64_t)std::max(nsel, 1) * NT * 8))) return rc;
    MicroRingArgs q;
    memset(&q, 0, sizeof q);
    {
        int rank = 0;
        for (int v = 0; v < MCF_NOUT; ++v) {
            if (!outsel[v]) continue;
            a.out[v] = outbuf + (int64_t)rank * NT;
            S_TRY(hipMemcpyAsync(a.out[v], micro->var[v], (size_t)NT * 8, hipMemcpyHostToDevice, nullptr));
            q.held |= 1u << v;
            ++rank;
        }
        q.sel = q.held;
        q.base0 = outbuf; q.vstride = NT;
        q.ring.N = N; q.ring.cpb = 0;
    }
    const unsigned gridN = (unsigned)((N + 255) / 256);
    if (!af) {
        MicroMet* mm;
        if ((rc = b.alloc((void**)&mm, (int64_t)T * sizeof(MicroMet)))) return rc;
        hipLaunchKernelGGL(k_micro_steps, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, nullptr, a.temp, a.relhum, a.pres, a.mxtc1, T, mm);
        a.mmet = mm;
        MicroStep* ms;
        if ((rc = b.alloc((void**)&ms, (int64_t)T * sizeof(MicroStep)))) return rc;
        hipLaunchKernelGGL(k_micro_pack, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, nullptr, a.rows, a.mmet, a.temp, a.pres, a.windspeed,
                           a.swdown, a.difrad, a.lwdown, a.umu, T, ms);
        a.mstep = ms;
    }
    if (af) hipLaunchKernelGGL(k_microsnow_cell<true>, dim3(gridN), dim3(256), 0, nullptr, a);
    else hipLaunchKernelGGL(k_microsnow_cell<false>, dim3(gridN), dim3(256), 0, nullptr, a);
    {
        std::vector<int32_t> ident((size_t)nch), ones((size_t)nch, 1);
        for (int d = 0; d < nch; ++d) ident[(size_t)d] = d;
        int32_t *d_map = nullptr, *d_one = nullptr;
        if ((rc = b.alloc((void**)&d_map, (int64_t)nch * 4))) return rc;
        if ((rc = b.alloc((void**)&d_one, (int64_t)nch * 4))) return rc;
        S_TRY(hipMemcpy(d_map, ident.data(), (size_t)nch * 4, hipMemcpyHostToDevice));
        S_TRY(hipMemcpy(d_one, ones.data(), (size_t)nch * 4, hipMemcpyHostToDevice));
        q.daymap = d_map; q.nosnow = d_one; q.ndays = nch;
        for (int d0 = 0; d0 < nch; d0 += 32768) {        // (cells, days): the grid's y extent is 16 bits
            a.day0 = d0;
            q.m = a;
            const dim3 grid((unsigned)((N + 63) / 64), (unsigned)std::min(32768, nch - d0));
            if (af) hipLaunchKernelGGL(k_microsnow_ring<true>, grid, dim3(256), 0, nullptr, q, (const void*)a.dates, q.daymap, q.nosnow);
            else hipLaunchKernelGGL(k_microsnow_ring<false>, grid, dim3(256), 0, nullptr, q, a.mstep, q.daymap, q.nosnow);
        }
    }
    S_TRY(hipGetLastError());
    Downloader dl;
    for (int v = 0; v < MCF_NOUT; ++v)
        if (outsel[v]) S_TRY(dl.get(micro->var[v], a.out[v], (size_t)NT * 8));
    S_TRY(hipDeviceSynchronize());
    return MCF_OK;
}

// (the chunk loop lives in mcf_snowplan below)

}  // namespace

// ---- .snowmodel1's chunk loop as a stepwise, device-resident plan (one per rank's row block) ---------------
struct mcf_snowplan {
    int device = 0;
    int64_t rows = 0, cols = 0, N = 0, row0 = 0, rows_total = 0;
    int T = 0, chunk = 120, nchunks = 1, ss = 10;
    double res = 1.0, tfact = 0.02, zref = 2.0;
    Bufs b;
    ModelArgs a;
    const StepRow* rows_tab = nullptr;
    // array weather (`.snowmodel2`'s loop): the caller's thirteen [N][T] series — a chunk's slices go up as the loop reaches it —
    // and the date rows of every step
    bool af = false;
    const double* h_series[13] = {};      // temp, relhum, pres, swdown, difrad, lwdown, windspeed, precip, Gp, Tc, RswabsG, RlwabsG, umu
    double* d_series[13] = {};            // [N][chunk]
    const DateRow2* dates_tab = nullptr;
    const double *d_dtm = nullptr, *d_isnowdg = nullptr;
    double *d_isnowdc = nullptr, *d_dtms = nullptr, *d_slope = nullptr, *d_aspect = nullptr, *d_svf = nullptr,
           *d_wsa = nullptr, *d_hor = nullptr, *d_tpic = nullptr, *d_mean2 = nullptr, *d_cm = nullptr, *d_ext = nullptr,
           *d_sumws = nullptr;
    int64_t ext_cap = 0, cm_cap = 0;
    // the surface (own rows + halos) the terrain arrays were last derived from, and its geometry: a chunk that starts from the
    // very same surface — every snow-free stretch of the year — keeps them (bit patterns compared on the device)
    double* d_zlast = nullptr;
    int64_t zlast_n = 0, zlast_cap = 0;
    int32_t zlast_hn = -1, zlast_hs = -1;
    int32_t* d_zdiff = nullptr;
    int terrain_reused = 0, terrain_refreshed = 0;
    // mcf_snowplan_apply3: the chunk's per-step max and min of totalSWE come out of one pass (k_apply3_minmax_part) into a
    // workspace the plan owns; whichever of the two is asked for first computes both, the other is answered from here
    double* d_mm = nullptr;              // [2 x chunk x parts x 2] partials, [4 x chunk] results
    int mm_parts = 0, mm_chunk = -1;     // mm_chunk: the chunk whose run the cached values belong to (-1: none)
    std::vector<double> mm_host;         // max[chunk], count[chunk], min[chunk], count[chunk]
    int32_t *d_ac = nullptr, *d_ag = nullptr;
    std::vector<double> wind;
    Downloader dl;
    int prepared = -1;
    double t_terrain = 0, t_model = 0;   // ms, MCF_TIMING
    mcf::TerrainWork twork;              // terrain_device's scratch, kept across the chunks
    // initial hand-over state, for mcf_snowplan_reset (the snow-day microclimate needs a second pass over the series)
    double* d_isnowdc0 = nullptr;
    int32_t *d_ac0 = nullptr, *d_ag0 = nullptr;
    // gridmicrosnow1 inside the chunk loop (mcf_snowplan_micro_*)
    double *d_sumD = nullptr, *d_meanD = nullptr;
    int32_t* d_sden_na = nullptr;        // the subset series' first snow density is NA (cpp:4716)
    int64_t sumD_steps = 0;
    bool micro_ready = false;
    Bufs mb, mbs;                        // the micro set-up's buffers: per-call (series) and static (vegetation, terrain)
    bool micro_static = false;
    MicroArgs ma;
    // array weather: the caller's whole-series [N][T] arrays of gridmicrosnow2's weather (temp, relhum, pres, swdown, difrad, lwdown,
    // windspeed, precip, umu) — a chunk's snow days go up when the chunk's microclimate runs — and their device slabs [N][chunk]
    bool micro_af = false;
    const double* h_micro[9] = {};
    double* d_micro[9] = {};
    int32_t outsel[MCF_NOUT] = {};
    std::vector<int32_t> sub_of_day;     // absolute day -> day of the snow-day subset series, or -1
    int32_t *d_daymap = nullptr, *d_nosnow = nullptr;     // [chunk days]
    // mcf_snowplan_set_series: which of the five device series (bit 0 Tc, 1 Tg, 2 totalSWE, 3 ground snow depth, 4 density) the
    // next run_chunk writes; series_valid: what the plan's working buffers hold of the chunk run last
    uint32_t series_mask = 31, series_valid = 0;
    uint8_t* d_tflag = nullptr;          // mcf_snowplan_covered_tiles: one flag per tile of the solver plan
    int64_t tflag_cap = 0;
    uint8_t* d_need = nullptr;           // mcf_snowplan_free_cells: one flag per cell + an 8-byte count behind them
    int64_t need_cap = 0;
    // hand-over state at the start of a chunk (mcf_snowplan_checkpoint): isnowdc, the snow surface, the two age matrices
    std::vector<char*> ckpt;
    // series of chunks kept on the device between the two passes (mcf_snowplan_keep_chunk): a kept chunk's buffers are the ones
    // the model wrote — the plan goes on with fresh ones — so pass 2 neither re-runs the chunk nor copies anything
    struct Kept { double *Tc = nullptr, *Tg = nullptr, *sdepc = nullptr, *sdepg = nullptr, *sden = nullptr, *tzd = nullptr; };
    // (a kept set belongs to the run of its chunk that was current when it was handed over: running the chunk again — a new
    // pass 1 without mcf_snowplan_release_kept, a re-run after mcf_snowplan_reset — returns the stale set to the pool, so that
    // mcf_snowplan_microsnow can never read last year's series for it)
    std::vector<Kept> kept;
    std::vector<Kept> pool;              // released sets, reused by the next year's pass 1 (hipMalloc of 10 GB costs 0.25 s)
    int64_t keep_budget = -1;            // bytes of kept sets this plan may ALLOCATE in all (mcf_snowplan_set_keep_budget); -1: no limit of its own
    int64_t keep_allocated = 0;
    Bufs kb;
    ~mcf_snowplan() { twork.release(); }
};

namespace {

int chunk_af(const mcf_snowplan* sp, int ch, int* af) {   // int:2589-2590
    const int k0 = ch * sp->chunk, ns = std::min(sp->chunk, sp->T - k0);
    double wsum = 0.0;
    for (int k = 0; k < ns; ++k) wsum += sp->wind[k0 + k];
    const double tpr = 10 * sqrt(wsum / ns);
    double afd = nearbyint(tpr / sp->res);                // R's round(x, 0): half to even
    if (sp->af && !(afd >= 2.0)) afd = 2.0;               // `.snowmodel2`: `if (af < 2) af <- 2`, int:2984
    if (!(afd >= 1.0))
        return mcf::api_fail(MCF_ERR_ARG, "snow driver: aggregation factor round(10*sqrt(mean wind)/res) is 0 (terra::aggregate fails)");
    *af = (int)std::min(afd, 1e9);
    return MCF_OK;
}

}  // namespace

extern "C" int mcf_snowplan_chunk_af(const mcf_snowplan* sp, int32_t chunk, int32_t* af) {
    if (!sp || !af) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (chunk < 0 || chunk >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    int a = 1;
    const int rc = chunk_af(sp, chunk, &a);
    *af = a;
    return rc;
}
extern "C" int mcf_snowplan_create(const mcf_snowdriver_in* din, int64_t row0, int64_t rows_total, int32_t device,
                                   mcf_snowplan** out) {
    int rc;
    if (!din || !out) return mcf::api_fail(MCF_ERR_ARG, "null snow driver argument");
    const mcf_snow_inputs* in = &din->base;
    if ((rc = common_checks(in))) return rc;
    const bool af = in->array_forcing != 0;
    if (!din->dtm || !(din->res > 0)) return mcf::api_fail(MCF_ERR_ARG, "snow driver needs dtm and res > 0");
    if (!in->clim.windspeed) return mcf::api_fail(MCF_ERR_ARG, "null input: windspeed");
    if (rows_total <= 0) { rows_total = in->rows; row0 = 0; }
    if (row0 < 0 || row0 + in->rows > rows_total) return mcf::api_fail(MCF_ERR_ARG, "block outside the raster");
    if (af) {
        if (row0 != 0 || rows_total != in->rows) return mcf::api_fail(MCF_ERR_ARG, "snow driver, array weather: one block (the whole raster)");
        if (!din->af_wind) return mcf::api_fail(MCF_ERR_ARG, "snow driver, array weather: af_wind (the chunk wind series) is null");
        if (!in->other.lats || !in->other.lons || !in->clim.winddir) return mcf::api_fail(MCF_ERR_ARG, "snow driver, array weather: lats / lons / winddir");
        const double* need[13] = {in->clim.temp, in->clim.relhum, in->clim.pres, in->clim.swdown, in->clim.difrad, in->clim.lwdown,
                                  in->clim.windspeed, in->clim.precip, in->pointm.Gp, in->pointm.Tc, in->pointm.RswabsG,
                                  in->pointm.RlwabsG, in->pointm.umu};
        for (const double* q : need)
            if (!q) return mcf::api_fail(MCF_ERR_ARG, "snow driver, array weather: a climate / point-model array is null");
    }
    if ((rc = pick_device(device))) return rc;
    mcf_snowplan* sp = new mcf_snowplan();
    struct Guard { mcf_snowplan* p; ~Guard() { delete p; } } guard{sp};
    sp->device = device;
    sp->rows = in->rows; sp->cols = in->cols; sp->N = in->rows * in->cols; sp->row0 = row0; sp->rows_total = rows_total;
    sp->T = (int)in->tsteps;
    sp->chunk = din->chunk_steps > 0 ? din->chunk_steps : 120;
    if (sp->chunk % 24) return mcf::api_fail(MCF_ERR_ARG, "snow driver: chunk_steps must be whole days");
    sp->nchunks = std::max(1, sp->T / sp->chunk);   // `for (day in 1:n5days)`: 1:x truncates, and 1:0.4 still runs once
    sp->res = din->res; sp->tfact = din->tfact; sp->zref = in->other.zref;
    sp->ss = din->res <= 100 ? 10 : 1;              // int:2577-2578
    sp->af = af;
    if (af && din->af_wsa_s > 0) sp->ss = din->af_wsa_s;                        // int:2963-2964
    if (af) sp->wind.assign(din->af_wind, din->af_wind + sp->T);                // wss, int:2981
    else sp->wind.assign(in->clim.windspeed, in->clim.windspeed + sp->T);
    const int64_t N = sp->N;
    const int T = sp->T;
    Bufs& b = sp->b;
    ModelArgs& a = sp->a;
    memset(&a, 0, sizeof a);
    a.N = N; a.zref = sp->zref;
    snow_density_params(in->snowenv, a.sdp);
    UP(a.pai, in->vegp.pai, N);
    UP(a.hgt, in->vegp.hgt, N);
    UP(a.leaft, in->vegp.leaft, N);
    UP(a.clump, in->vegp.clump, N);
    UP(sp->d_dtm, din->dtm, N);
    UP(sp->d_isnowdg, in->other.isnowdg, N);
    { const double* t; UP(t, in->other.isnowdc, N); sp->d_isnowdc = const_cast<double*>(t); }
    { const int32_t* t; UP(t, in->other.isnowac, N); sp->d_ac = const_cast<int32_t*>(t); }
    { const int32_t* t; UP(t, in->other.isnowag, N); sp->d_ag = const_cast<int32_t*>(t); }
    { const double* t; UP(t, in->other.isnowdc, N); sp->d_isnowdc0 = const_cast<double*>(t); }
    { const int32_t* t; UP(t, in->other.isnowac, N); sp->d_ac0 = const_cast<int32_t*>(t); }
    { const int32_t* t; UP(t, in->other.isnowag, N); sp->d_ag0 = const_cast<int32_t*>(t); }
    if ((rc = b.alloc((void**)&sp->d_sumD, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_meanD, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_sden_na, N * 4))) return rc;
    if ((rc = b.alloc((void**)&sp->d_daymap, (sp->chunk / 24 + 1) * 4))) return rc;
    if ((rc = b.alloc((void**)&sp->d_nosnow, (sp->chunk / 24 + 1) * 4))) return rc;
    if ((rc = b.alloc((void**)&sp->d_dtms, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_slope, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_aspect, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_svf, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_wsa, 8 * N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_hor, 24 * N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_tpic, N * 8))) return rc;
    if ((rc = b.alloc((void**)&sp->d_mean2, 16))) return rc;
    if ((rc = b.alloc((void**)&sp->d_sumws, 2 * kSumParts * 8))) return rc;
    a.slope = sp->d_slope; a.aspect = sp->d_aspect; a.skyview = sp->d_svf; a.wsa = sp->d_wsa; a.hor = sp->d_hor;
    a.isnowdc = sp->d_isnowdc; a.isnowdg = sp->d_isnowdg; a.isnowac = sp->d_ac; a.isnowag = sp->d_ag;
    const int64_t CN = (int64_t)sp->chunk * N;
    const double* mx_unused;
    if (af) {
        // gridmodelsnow2 on each chunk: the date rows of every step; the cell's part of the sun position and its albedo clock
        // (restarted at the chunk's first step, as every gridmodelsnow2 call does) are the kernel's
        const StepRow* rows_unused;
        if ((rc = build_step_tables(b, in, true, true, false, &rows_unused, &sp->dates_tab, &mx_unused))) return rc;
        UP(a.lats, in->other.lats, N);
        UP(a.lons, in->other.lons, N);
        const double* hs[13] = {in->clim.temp, in->clim.relhum, in->clim.pres, in->clim.swdown, in->clim.difrad, in->clim.lwdown,
                                in->clim.windspeed, in->clim.precip, in->pointm.Gp, in->pointm.Tc, in->pointm.RswabsG,
                                in->pointm.RlwabsG, in->pointm.umu};
        for (int f = 0; f < 13; ++f) {
            sp->h_series[f] = hs[f];
            if ((rc = b.alloc((void**)&sp->d_series[f], CN * 8))) return rc;
        }
        a.temp = sp->d_series[0]; a.relhum = sp->d_series[1]; a.pres = sp->d_series[2]; a.swdown = sp->d_series[3];
        a.difrad = sp->d_series[4]; a.lwdown = sp->d_series[5]; a.windspeed = sp->d_series[6]; a.precip = sp->d_series[7];
        a.Gp = sp->d_series[8]; a.Tcp = sp->d_series[9]; a.RswabsG = sp->d_series[10]; a.RlwabsG = sp->d_series[11];
        a.umu = sp->d_series[12];
    } else {
        const DateRow2* dates_unused;
        if ((rc = build_step_tables(b, in, false, true, true, &sp->rows_tab, &dates_unused, &mx_unused))) return rc;
        // albedo per chunk (overrides the whole-series scan of build_step_tables)
        const double* d_prec;
        UP(d_prec, in->clim.precip, T);
        hipLaunchKernelGGL(k_snow_alb_chunks, dim3((unsigned)((sp->nchunks + 63) / 64)), dim3(64), 0, nullptr,
                           const_cast<StepRow*>(sp->rows_tab), d_prec, T, sp->chunk, sp->nchunks);
    }
    if ((rc = b.alloc((void**)&a.Tc, CN * 8))) return rc;
    if ((rc = b.alloc((void**)&a.Tg, CN * 8))) return rc;
    if ((rc = b.alloc((void**)&a.sdepc, CN * 8))) return rc;
    if ((rc = b.alloc((void**)&a.sdepg, CN * 8))) return rc;
    if ((rc = b.alloc((void**)&a.sden, CN * 8))) return rc;
    if (!sp->af && (rc = b.alloc((void**)&a.tzd, (int64_t)std::max(sp->chunk / 24, 1) * N * 8))) return rc;
    if ((rc = b.alloc((void**)&a.agec, N * 8))) return rc;
    if ((rc = b.alloc((void**)&a.ageg, N * 8))) return rc;
    hipLaunchKernelGGL(k_add_snow, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, sp->d_dtm, sp->d_isnowdg, 1.0,
                       N, sp->d_dtms);   // int:2562
    S_TRY(hipGetLastError());
    S_TRY(hipDeviceSynchronize());
    guard.p = nullptr;
    *out = sp;
    return MCF_OK;
}
extern "C" void mcf_snowplan_destroy(mcf_snowplan* sp) {
    if (!sp) return;
    (void)hipSetDevice(sp->device);
    delete sp;
}
extern "C" int32_t mcf_snowplan_chunks(const mcf_snowplan* sp) { return sp ? sp->nchunks : 0; }
extern "C" int mcf_snowplan_surface(mcf_snowplan* sp, double* host_own) {
    if (!sp || !host_own) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    S_TRY(hipSetDevice(sp->device));
    S_TRY(hipMemcpy(host_own, sp->d_dtms, (size_t)sp->N * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}
extern "C" int mcf_snowplan_handover(mcf_snowplan* sp, double* host_isnowdc) {
    if (!sp || !host_isnowdc) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    S_TRY(hipSetDevice(sp->device));
    S_TRY(hipMemcpy(host_isnowdc, sp->d_isnowdc, (size_t)sp->N * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}
extern "C" int mcf_snowplan_surface_partial(mcf_snowplan* sp, double* sum, double* count) {
    if (!sp || !sum || !count) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    S_TRY(hipSetDevice(sp->device));
    launch_sumcount(sp->d_dtms, sp->N, sp->d_sumws, sp->d_mean2);
    double h[2];
    S_TRY(hipMemcpy(h, sp->d_mean2, 16, hipMemcpyDeviceToHost));
    *sum = h[0]; *count = h[1];
    return MCF_OK;
}
// 1 into *diff if the two surfaces differ in any bit (NaN cells compare by pattern)
__global__ __launch_bounds__(256) void k_surface_differs(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                         int32_t* __restrict__ diff) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool d = t < n && __double_as_longlong(a[t]) != __double_as_longlong(b[t]);
    // (a wave that sees the flag up already leaves it alone: with surfaces that differ everywhere — every chunk of the snow season —
    // 40 000 atomics on one address took 0.2 ms of a launch that reads 33 MB)
    if (__builtin_amdgcn_ballot_w64(d) != 0 && (threadIdx.x & 63) == 0 && *(volatile int32_t*)diff == 0) atomicOr(diff, 1);
}
// own block + halo rows, column-major [hn + rows + hs, cols], put together on the device from three column-major pieces
__global__ void k_ext_assemble(double* __restrict__ ext, const double* __restrict__ own, const double* __restrict__ north,
                               const double* __restrict__ south, int64_t rows, int64_t cols, int hn, int hs) {
    const int64_t RB = hn + rows + hs, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= RB * cols) return;
    const int64_t r = t % RB, c = t / RB;
    ext[t] = r < hn ? north[r + (int64_t)hn * c] : r < hn + rows ? own[(r - hn) + rows * c] : south[(r - hn - rows) + (int64_t)hs * c];
}
// the own block's first hn / last hs rows as column-major [h, cols] pieces (what a neighbouring rank receives as its halo)
__global__ void k_halo_pack(const double* __restrict__ own, int64_t rows, int64_t cols, int hn, int hs, double* __restrict__ north,
                            double* __restrict__ south) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, h = hn + hs;
    if (t >= h * cols) return;
    const int64_t i = t % h, c = t / h;
    if (i < hn) north[i + (int64_t)hn * c] = own[i + rows * c];
    else south[(i - hn) + (int64_t)hs * c] = own[(rows - hs + (i - hn)) + rows * c];
}
static int ext_room(mcf_snowplan* sp, int64_t n) {
    if (sp->ext_cap < n) {
        int rc;
        if ((rc = sp->b.alloc((void**)&sp->d_ext, n * 8))) return rc;
        sp->ext_cap = n;
    }
    return MCF_OK;
}
static int prepare_chunk_on(mcf_snowplan* sp, int32_t ch, const double* d_z, int32_t hn, int32_t hs, double surface_mean,
                            double* tpic_sum, double* tpic_count);

extern "C" int mcf_snowplan_prepare_chunk(mcf_snowplan* sp, int32_t ch, const double* ext, int32_t hn, int32_t hs,
                                          double surface_mean, double* tpic_sum, double* tpic_count) {
    if (!sp || !tpic_sum || !tpic_count) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    if (hn < 0 || hs < 0 || (!ext && (hn || hs))) return mcf::api_fail(MCF_ERR_ARG, "bad halo");
    S_TRY(hipSetDevice(sp->device));
    const double* d_z = sp->d_dtms;
    if (ext) {
        const int64_t n = (hn + sp->rows + hs) * sp->cols;
        int rc;
        if ((rc = ext_room(sp, n))) return rc;
        S_TRY(hipMemcpy(sp->d_ext, ext, (size_t)n * 8, hipMemcpyHostToDevice));
        d_z = sp->d_ext;
    }
    return prepare_chunk_on(sp, ch, d_z, hn, hs, surface_mean, tpic_sum, tpic_count);
}
extern "C" int mcf_snowplan_pack_halo(mcf_snowplan* sp, int32_t hn, double* d_north, int32_t hs, double* d_south) {
    if (!sp || hn < 0 || hs < 0 || hn > sp->rows || hs > sp->rows || (hn && !d_north) || (hs && !d_south))
        return mcf::api_fail(MCF_ERR_ARG, "bad mcf_snowplan_pack_halo argument");
    S_TRY(hipSetDevice(sp->device));
    const int64_t n = (int64_t)(hn + hs) * sp->cols;
    if (n > 0) hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, sp->d_dtms, sp->rows, sp->cols, hn, hs, d_north, d_south);
    S_TRY(hipGetLastError());
    S_TRY(hipStreamSynchronize(nullptr));      // the pieces are the caller's to send from here on
    return MCF_OK;
}
extern "C" int mcf_snowplan_prepare_chunk_dev(mcf_snowplan* sp, int32_t ch, const double* d_north, int32_t hn, const double* d_south,
                                              int32_t hs, double surface_mean, double* tpic_sum, double* tpic_count) {
    if (!sp || !tpic_sum || !tpic_count) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    if (hn < 0 || hs < 0 || (hn && !d_north) || (hs && !d_south)) return mcf::api_fail(MCF_ERR_ARG, "bad halo");
    S_TRY(hipSetDevice(sp->device));
    const double* d_z = sp->d_dtms;
    if (hn || hs) {
        const int64_t n = (hn + sp->rows + hs) * sp->cols;
        int rc;
        if ((rc = ext_room(sp, n))) return rc;
        hipLaunchKernelGGL(k_ext_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, sp->d_ext, sp->d_dtms, d_north, d_south,
                           sp->rows, sp->cols, hn, hs);
        S_TRY(hipGetLastError());
        d_z = sp->d_ext;
    }
    return prepare_chunk_on(sp, ch, d_z, hn, hs, surface_mean, tpic_sum, tpic_count);
}
static int prepare_chunk_on(mcf_snowplan* sp, int32_t ch, const double* d_z, int32_t hn, int32_t hs, double surface_mean,
                            double* tpic_sum, double* tpic_count) {
    int rc, af;
    if ((rc = chunk_af(sp, ch, &af))) return rc;
    const int64_t rows = sp->rows, cols = sp->cols, N = sp->N, RB = hn + rows + hs;
    const int64_t me = std::min(sp->rows_total, cols);
    const bool coarse = (double)af < me / 2.0;
    TpiGeo g;
    g.rows = rows; g.cols = cols; g.RB = RB; g.hn = hn; g.row0 = sp->row0; g.rows_total = sp->rows_total; g.af = af;
    g.NItot = (sp->rows_total + af - 1) / af; g.nJ = (cols + af - 1) / af;
    auto clampI = [&](int64_t i) { return std::min<int64_t>(std::max<int64_t>(i, 0), g.NItot - 1); };
    g.I0 = clampI((int64_t)floor(((double)sp->row0 - (af - 1) / 2.0) / af));
    const int64_t I1 = clampI((int64_t)floor(((double)(sp->row0 + rows - 1) - (af - 1) / 2.0) / af) + 1);
    g.nI = I1 - g.I0 + 1;
    // halo rows this chunk needs (or every row up to the raster edge): the terrain stencil and, for the tpi,
    // whole af x af blocks around the own rows
    const int64_t avail_n = sp->row0, avail_s = sp->rows_total - sp->row0 - rows;
    int64_t need_n = 100 + 2 * sp->ss + sp->ss / 2, need_s = need_n;
    if (coarse) {
        need_n = std::max(need_n, sp->row0 - g.I0 * af);
        need_s = std::max(need_s, std::min<int64_t>((I1 + 1) * af, sp->rows_total) - (sp->row0 + rows));
    }
    if (hn < std::min(need_n, avail_n) || hs < std::min(need_s, avail_s)) {
        char m[200];
        snprintf(m, sizeof m, "snow plan: chunk %d needs %lld / %lld halo rows north / south (or all rows up to the raster edge)",
                 ch, (long long)need_n, (long long)need_s);
        return mcf::api_fail(MCF_ERR_ARG, m);
    }
    const bool timing = getenv("MCF_TIMING") != nullptr;
    Events evs;
    if (timing) { S_TRY(evs.make(2)); S_TRY(hipEventRecord(evs.e[0], nullptr)); }
    // terrain of dtm + snow (int:2566-2580)
    mcf::TerrainDev td;
    memset(&td, 0, sizeof td);
    td.rows = rows; td.cols = cols; td.halo_north = hn; td.halo_south = hs; td.row0 = sp->row0; td.rows_total = sp->rows_total;
    td.d_dtm = d_z; td.res = sp->res; td.zref = sp->zref; td.agg = sp->ss; td.aspect_na = 180.0;
    td.d_slope = sp->d_slope; td.d_aspect = sp->d_aspect; td.d_hor = sp->d_hor; td.d_svfa = sp->d_svf; td.d_wsa = sp->d_wsa;
    const unsigned gridN = (unsigned)((N + 255) / 256);
    // The terrain arrays are functions of the surface alone.  If this chunk starts from the surface they were last derived
    // from, bit for bit — no snow has lain anywhere since — they are kept (MCF_SNOW_TERRAIN_ALWAYS=1: never).
    const int64_t zn = RB * cols;
    bool same = false;
    static const bool always = getenv("MCF_SNOW_TERRAIN_ALWAYS") != nullptr;
    if (!always) {
        if (!sp->d_zdiff && (rc = sp->b.alloc((void**)&sp->d_zdiff, 4))) return rc;
        if (sp->d_zlast && sp->zlast_n == zn && sp->zlast_hn == hn && sp->zlast_hs == hs) {
            S_TRY(hipMemsetAsync(sp->d_zdiff, 0, 4, nullptr));
            hipLaunchKernelGGL(k_surface_differs, dim3((unsigned)((zn + 255) / 256)), dim3(256), 0, nullptr, d_z, (const double*)sp->d_zlast, zn,
                               sp->d_zdiff);
            int32_t diff = 1;
            S_TRY(hipMemcpy(&diff, sp->d_zdiff, 4, hipMemcpyDeviceToHost));
            same = diff == 0;
        }
    }
    if (same) {
        ++sp->terrain_reused;
    } else {
        if ((rc = mcf::terrain_device(td, &sp->twork))) return rc;
        hipLaunchKernelGGL(k_mask2, dim3(gridN), dim3(256), 0, nullptr, sp->d_dtm, N, sp->d_slope, sp->d_aspect);
        ++sp->terrain_refreshed;
        if (!always) {
            if (sp->zlast_cap < zn) {
                if ((rc = sp->b.alloc((void**)&sp->d_zlast, zn * 8))) return rc;
                sp->zlast_cap = zn;
            }
            S_TRY(hipMemcpyAsync(sp->d_zlast, d_z, (size_t)zn * 8, hipMemcpyDeviceToDevice, nullptr));
            sp->zlast_n = zn; sp->zlast_hn = hn; sp->zlast_hs = hs;
        }
    }
    // topographic positioning index (int:2589-2592, 2471-2485)
    if (coarse) {
        if (sp->cm_cap < g.nI * g.nJ) {
            if ((rc = sp->b.alloc((void**)&sp->d_cm, g.nI * g.nJ * 8))) return rc;
            sp->cm_cap = g.nI * g.nJ;
        }
        hipLaunchKernelGGL(k_tpi_coarse, dim3((unsigned)((g.nI * g.nJ + 255) / 256)), dim3(256), 0, nullptr, d_z, g, sp->d_cm);
        hipLaunchKernelGGL(k_tpi_fine, dim3(gridN), dim3(256), 0, nullptr, d_z, g, (const double*)sp->d_cm, 0.0, sp->tfact,
                           sp->d_tpic);
    } else {
        hipLaunchKernelGGL(k_tpi_fine, dim3(gridN), dim3(256), 0, nullptr, d_z, g, (const double*)nullptr, surface_mean,
                           sp->tfact, sp->d_tpic);
    }
    launch_sumcount(sp->d_tpic, N, sp->d_sumws, sp->d_mean2);
    S_TRY(hipGetLastError());
    double h[2];
    S_TRY(hipMemcpy(h, sp->d_mean2, 16, hipMemcpyDeviceToHost));
    *tpic_sum = h[0]; *tpic_count = h[1];
    if (timing) {
        S_TRY(hipEventRecord(evs.e[1], nullptr));
        S_TRY(hipEventSynchronize(evs.e[1]));
        float ms = 0;
        S_TRY(hipEventElapsedTime(&ms, evs.e[0], evs.e[1]));
        sp->t_terrain += ms;
    }
    sp->prepared = ch;
    return MCF_OK;
}
// host_step0: the step of the host arrays the chunk's first step goes to (the chunk's own place in whole-series arrays, or 0
// for a caller that takes the chunk into a chunk-sized buffer); fill_tail: whole-series arrays — the steps no chunk covers
// become NA behind the last chunk
// row_pitch: 0 / rows = dense [rows, cols, steps] host arrays; > rows: the host arrays are row blocks of a taller column-major
// raster with that many rows per column, written in place (one strided DMA per series: no block-sized host buffer, no scatter)
static int run_chunk_to(mcf_snowplan* sp, int32_t ch, double tpic_mean, const mcf_snowdriver_out* out, int64_t host_step0, bool fill_tail,
                        int64_t row_pitch);
extern "C" int mcf_snowplan_run_chunk(mcf_snowplan* sp, int32_t ch, double tpic_mean, mcf_snowdriver_out* out) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    return run_chunk_to(sp, ch, tpic_mean, out, (int64_t)ch * sp->chunk, true, 0);
}
extern "C" int mcf_snowplan_run_chunk_pitched(mcf_snowplan* sp, int32_t ch, double tpic_mean, const mcf_snowdriver_out* out,
                                              int64_t row_pitch) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (row_pitch != 0 && row_pitch < sp->rows) return mcf::api_fail(MCF_ERR_ARG, "row_pitch smaller than rows");
    return run_chunk_to(sp, ch, tpic_mean, out, (int64_t)ch * sp->chunk, true, row_pitch);
}
static int run_chunk_to(mcf_snowplan* sp, int32_t ch, double tpic_mean, const mcf_snowdriver_out* out, int64_t host_step0, bool fill_tail,
                        int64_t row_pitch) {
    if (!sp || !out) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (row_pitch <= 0) row_pitch = sp->rows;
    if (ch != sp->prepared) return mcf::api_fail(MCF_ERR_STATE, "snow plan: run_chunk needs prepare_chunk of the same chunk first");
    S_TRY(hipSetDevice(sp->device));
    sp->mm_chunk = -1;
    if ((size_t)ch < sp->kept.size() && sp->kept[(size_t)ch].Tc) {     // a set kept from an earlier run of this chunk is stale now
        sp->pool.push_back(sp->kept[(size_t)ch]);
        sp->kept[(size_t)ch] = mcf_snowplan::Kept();
    }
    const int64_t N = sp->N;
    const int k0 = ch * sp->chunk, ns = std::min(sp->chunk, sp->T - k0);
    const unsigned gridN = (unsigned)((N + 255) / 256);
    const bool timing = getenv("MCF_TIMING") != nullptr;
    Events evs;
    if (timing) { S_TRY(evs.make(2)); S_TRY(hipEventRecord(evs.e[0], nullptr)); }
    ModelArgs a = sp->a;
    a.rows = sp->af ? nullptr : sp->rows_tab + k0;            // gridmodelsnow1 on the chunk (int:2587)
    a.dates = sp->af ? sp->dates_tab + k0 : nullptr;          // gridmodelsnow2 (int:2979)
    a.tsteps = ns;
    if (sp->af)                                               // the chunk's slices of the caller's series: [N][T], a step's raster contiguous
        for (int f = 0; f < 13; ++f)
            S_TRY(hipMemcpyAsync(sp->d_series[f], sp->h_series[f] + (int64_t)k0 * N, (size_t)ns * N * 8, hipMemcpyHostToDevice, nullptr));
    // ... with the redistribution by the topographic position index and the hand-over fused in (ModelArgs)
    a.tpic = sp->d_tpic; a.tpimean = tpic_mean; a.dtm = sp->d_dtm;
    a.isnowdc_out = sp->d_isnowdc; a.dtms = sp->d_dtms; a.isnowac_out = sp->d_ac; a.isnowag_out = sp->d_ag;
    {   // series nobody will read are not written (mcf_snowplan_set_series): the kernel's stores are what a snow-free chunk costs
        const uint32_t m = sp->series_mask;
        double* const hostp[5] = {out->Tc, out->Tg, out->totalSWE, out->groundsnowdepth, out->snowden};
        for (int v = 0; v < 5; ++v)
            if (!((m >> v) & 1u) && hostp[v]) return mcf::api_fail(MCF_ERR_STATE, "snow plan: a series the caller asks for is switched off (mcf_snowplan_set_series)");
        if (!(m & 1u)) a.Tc = nullptr;
        if (!(m & 2u)) a.Tg = nullptr;
        if (!(m & 4u)) a.sdepc = nullptr;
        if (!(m & 8u)) a.sdepg = nullptr;
        if (!(m & 16u)) a.sden = nullptr;
        sp->series_valid = m;
    }
    if (sp->af) hipLaunchKernelGGL(k_snowmodel<true>, dim3(gridN), dim3(256), 0, nullptr, a, a.rows, a.dates);
    else hipLaunchKernelGGL(k_snowmodel<false>, dim3(gridN), dim3(256), 0, nullptr, a, a.rows, a.dates);
    S_TRY(hipGetLastError());
    if (timing) {
        S_TRY(hipEventRecord(evs.e[1], nullptr));
        S_TRY(hipEventSynchronize(evs.e[1]));
        float ms = 0;
        S_TRY(hipEventElapsedTime(&ms, evs.e[0], evs.e[1]));
        sp->t_model += ms;
    }
    double* hostv[5] = {out->Tc, out->Tg, out->groundsnowdepth, out->totalSWE, out->snowden};
    double* devv[5] = {a.Tc, a.Tg, a.sdepg, a.sdepc, a.sden};
    const int64_t HS = row_pitch * sp->cols;        // host doubles per step
    for (int v = 0; v < 5; ++v) {
        if (!hostv[v]) continue;
        if (row_pitch == sp->rows) S_TRY(sp->dl.get(hostv[v] + host_step0 * N, devv[v], (size_t)ns * N * 8));
        else S_TRY(sp->dl.get_pitched(hostv[v] + host_step0 * HS, (size_t)row_pitch * 8, devv[v], (size_t)sp->rows * 8, (size_t)(sp->cols * ns)));
    }
    if (fill_tail && ch == sp->nchunks - 1) {   // steps that no chunk covers stay NA (R pre-fills its arrays, int:2554-2558)
        union { uint64_t u; double d; } na; na.u = kNaRealBits;
        const int covered = std::min(sp->T, sp->nchunks * sp->chunk);
        for (double* h : hostv)
            if (h)
                for (int64_t lc = (int64_t)covered * sp->cols; lc < (int64_t)sp->T * sp->cols; ++lc)
                    for (int64_t r = 0; r < sp->rows; ++r) h[r + row_pitch * lc] = na.d;
    }
    sp->prepared = -1;
    return MCF_OK;
}

static int snowmodel_loop(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, int32_t device);
extern "C" int mcf_snowmodel1(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, int32_t device) {
    if (in && in->base.array_forcing) return mcf::api_fail(MCF_ERR_ARG, "mcf_snowmodel1 takes data.frame (vector) climate; array weather: mcf_snowmodel2");
    return snowmodel_loop(in, out, device);
}
extern "C" int mcf_snowmodel2(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, int32_t device) {
    if (in && !in->base.array_forcing) return mcf::api_fail(MCF_ERR_ARG, "mcf_snowmodel2 takes array weather; data.frame climate: mcf_snowmodel1");
    return snowmodel_loop(in, out, device);
}
static int snowmodel_loop(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, int32_t device) {
    if (!out) return mcf::api_fail(MCF_ERR_ARG, "null snow driver argument");
    mcf_snowplan* sp = nullptr;
    int rc = mcf_snowplan_create(in, 0, 0, device, &sp);
    if (rc) return rc;
    struct Guard { mcf_snowplan* p; ~Guard() { mcf_snowplan_destroy(p); } } guard{sp};
    const int64_t me = std::min(sp->rows_total, sp->cols);
    for (int ch = 0; ch < sp->nchunks; ++ch) {
        int af;
        if ((rc = chunk_af(sp, ch, &af))) return rc;
        double mean = 0.0, s = 0.0, n = 0.0;
        if (!((double)af < me / 2.0)) {            // .tpicalc's raster-mean branch
            if ((rc = mcf_snowplan_surface_partial(sp, &s, &n))) return rc;
            mean = s / n;
        }
        if ((rc = mcf_snowplan_prepare_chunk(sp, ch, nullptr, 0, 0, mean, &s, &n))) return rc;
        if ((rc = mcf_snowplan_run_chunk(sp, ch, s / n, out))) return rc;
    }
    if (getenv("MCF_TIMING"))
        fprintf(stderr, "[mcf] snowmodel1: %d chunks of %d steps, %lld cells: terrain + tpi %.2f ms, gridmodelsnow + "
                "redistribute %.2f ms\n", sp->nchunks, sp->chunk, (long long)sp->N, sp->t_terrain, sp->t_model);
    return MCF_OK;
}

// ---- one process, several devices (include/mcf.h mcf_snowmodel1_multi) ---------------------------------------------------------
// The chunk loop over row blocks of ONE raster held by this process: block b is a snow plan on devices[b % n_devices], driven
// by that device's host thread.  What couples the blocks per chunk — the snow surface's halo rows for the terrain stencil and
// .tpicalc's block means, and the two raster-wide means as (sum, count) — goes through host memory between three phases
// (the phases of snow.py snowmodel1_chunks_tiled, where ranks exchange the same things over RCCL):
//   1  every block writes its rows of the surface into one whole-raster array and reports its (sum, count)
//   2  every block takes its rows plus halo out of that array, refreshes terrain + tpi, reports tpic's (sum, count)
//   3  every block runs the chunk with the raster-wide tpic mean and copies its series out
// Partial sums are added in block order, so a run is reproducible for a given n_blocks; against the single-plan run the two
// means differ in their last bits (another summation tree), like route 1's.
namespace {
struct PhaseBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0, generation = 0;
    explicit PhaseBarrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const int g = generation;
        if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return g != generation; });
    }
};
template <class T>
void gather_rows(std::vector<T>& dst, const T* src, int64_t R, int64_t C, int64_t r0, int64_t nr, int64_t layers = 1) {
    dst.resize((size_t)(nr * C * layers));
    for (int64_t lc = 0; lc < C * layers; ++lc) memcpy(&dst[(size_t)(nr * lc)], src + r0 + R * lc, (size_t)nr * sizeof(T));
}
}  // namespace

static int snowmodel1_multi_impl(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, const mcf_multi* mu);
extern "C" int mcf_snowmodel1_multi(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, const mcf_multi* mu) {
    try { return snowmodel1_multi_impl(in, out, mu); }
    catch (const std::exception& e) { return mcf::api_fail(MCF_ERR_NOMEM, std::string("mcf_snowmodel1_multi: ") + e.what()); }
}
static int snowmodel1_multi_impl(const mcf_snowdriver_in* in, mcf_snowdriver_out* out, const mcf_multi* mu) {
    if (!in || !out || !mu) return mcf::api_fail(MCF_ERR_ARG, "null snow driver argument");
    const mcf_snow_inputs& base = in->base;
    if (base.rows <= 0 || base.cols <= 0 || !in->dtm) return mcf::api_fail(MCF_ERR_ARG, "snow driver needs the raster and its dtm");
    const mcf_snow_vegp& vg = base.vegp;
    const mcf_snow_other& ot = base.other;
    if (!vg.pai || !vg.hgt || !vg.leaft || !vg.clump || !ot.isnowdc || !ot.isnowdg || !ot.isnowac || !ot.isnowag)
        return mcf::api_fail(MCF_ERR_ARG, "null input: a vegetation or initial-snow raster");
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
    const int64_t R = base.rows, C = base.cols;
    const int nb = (int)std::min<int64_t>(mu->n_blocks > 0 ? mu->n_blocks : (int)devs.size(), R);
    const int nt = (int)std::min<size_t>(devs.size(), (size_t)nb);
    struct Block {
        int64_t r0 = 0, nr = 0;
        std::vector<double> pai, hgt, leaft, clump, dc, dg, dtm, ext;
        std::vector<int32_t> ac, ag;
        mcf_snowplan* sp = nullptr;
        double s = 0, n = 0, ts = 0, tn = 0;
        ~Block() { if (sp) mcf_snowplan_destroy(sp); }
    };
    std::vector<Block> blocks((size_t)nb);
    std::vector<double> surface((size_t)(R * C));        // the whole raster's snow surface of the current chunk
    std::vector<int> rcs((size_t)nt, MCF_OK);
    std::vector<std::string> errs((size_t)nt);
    std::atomic<bool> failed{false};
    PhaseBarrier bar(nt);
    int nchunks = 0;
    double smean = 0.0, tmean = 0.0;
    double* const dst[5] = {out->Tc, out->Tg, out->groundsnowdepth, out->totalSWE, out->snowden};
    // every thread runs every phase of every chunk (also after a failure: the barrier counts heads), doing nothing once failed
    auto worker = [&](int t) {
        auto fail_here = [&](int rc) { rcs[(size_t)t] = rc; errs[(size_t)t] = mcf_last_error(); failed = true; };
        // (a std::bad_alloc from a block's host vectors must neither leave the thread — std::terminate would take the host R /
        // Python process down — nor skip a barrier the other threads wait at: each phase body runs under this guard)
        auto guarded = [&](auto&& body) {
            try { body(); }
            catch (const std::exception& e) { rcs[(size_t)t] = MCF_ERR_NOMEM; errs[(size_t)t] = std::string("snow driver: ") + e.what(); failed = true; }
        };
        guarded([&] {
        for (int b = t; b < nb && !failed; b += nt) {        // ---- plans
            Block& k = blocks[(size_t)b];
            k.r0 = R * b / nb; k.nr = R * (b + 1) / nb - k.r0;
            gather_rows(k.pai, vg.pai, R, C, k.r0, k.nr); gather_rows(k.hgt, vg.hgt, R, C, k.r0, k.nr);
            gather_rows(k.leaft, vg.leaft, R, C, k.r0, k.nr); gather_rows(k.clump, vg.clump, R, C, k.r0, k.nr);
            gather_rows(k.dc, ot.isnowdc, R, C, k.r0, k.nr); gather_rows(k.dg, ot.isnowdg, R, C, k.r0, k.nr);
            gather_rows(k.ac, ot.isnowac, R, C, k.r0, k.nr); gather_rows(k.ag, ot.isnowag, R, C, k.r0, k.nr);
            gather_rows(k.dtm, in->dtm, R, C, k.r0, k.nr);
            mcf_snowdriver_in bi = *in;
            bi.base.rows = k.nr;
            bi.base.vegp.pai = k.pai.data(); bi.base.vegp.hgt = k.hgt.data(); bi.base.vegp.leaft = k.leaft.data();
            bi.base.vegp.clump = k.clump.data();
            bi.base.other.isnowdc = k.dc.data(); bi.base.other.isnowdg = k.dg.data();
            bi.base.other.isnowac = k.ac.data(); bi.base.other.isnowag = k.ag.data();
            bi.base.other.slope = bi.base.other.aspect = bi.base.other.skyview = bi.base.other.wsa = bi.base.other.hor = nullptr;
            bi.dtm = k.dtm.data();
            const int rc = mcf_snowplan_create(&bi, k.r0, R, devs[(size_t)t], &k.sp);
            if (rc) { fail_here(rc); break; }
        }
        });
        bar.wait();
        if (t == 0 && !failed) nchunks = blocks[0].sp->nchunks;
        bar.wait();
        for (int ch = 0; ch < nchunks; ++ch) {
            guarded([&] {
            for (int b = t; b < nb && !failed; b += nt) {    // ---- phase 1: the surface
                Block& k = blocks[(size_t)b];
                k.ext.resize((size_t)(k.nr * C));
                int rc = mcf_snowplan_surface(k.sp, k.ext.data());
                if (!rc) rc = mcf_snowplan_surface_partial(k.sp, &k.s, &k.n);
                if (rc) { fail_here(rc); break; }
                for (int64_t c = 0; c < C; ++c) memcpy(&surface[(size_t)(k.r0 + R * c)], &k.ext[(size_t)(k.nr * c)], (size_t)k.nr * 8);
            }
            });
            bar.wait();
            if (t == 0 && !failed) {
                double s = 0, n = 0;
                for (const Block& k : blocks) { s += k.s; n += k.n; }
                smean = s / n;
            }
            bar.wait();
            guarded([&] {
            for (int b = t; b < nb && !failed; b += nt) {    // ---- phase 2: halos, terrain, tpi
                Block& k = blocks[(size_t)b];
                int af = 1;
                int rc = chunk_af(k.sp, ch, &af);
                if (rc) { fail_here(rc); break; }
                // what prepare_chunk asks for at most (the stencil's reach, whole af x af blocks), or every row up to the edge
                const int64_t want = 100 + 3 * (int64_t)k.sp->ss + 2 * (int64_t)af;
                const int64_t hn = std::min(want, k.r0), hs = std::min(want, R - k.r0 - k.nr), RB = hn + k.nr + hs;
                gather_rows(k.ext, surface.data(), R, C, k.r0 - hn, RB);
                rc = mcf_snowplan_prepare_chunk(k.sp, ch, (hn || hs) ? k.ext.data() : nullptr, (int32_t)hn, (int32_t)hs, smean, &k.ts, &k.tn);
                if (rc) { fail_here(rc); break; }
            }
            });
            bar.wait();
            if (t == 0 && !failed) {
                double s = 0, n = 0;
                for (const Block& k : blocks) { s += k.ts; n += k.tn; }
                tmean = s / n;
            }
            bar.wait();
            for (int b = t; b < nb && !failed; b += nt) {    // ---- phase 3: the chunk
                Block& k = blocks[(size_t)b];
                // the chunk's series go straight into the block's rows of the caller's arrays (strided DMA through the row
                // pitch); the steps no chunk covers become NA behind the last chunk, as in the single-plan run
                mcf_snowdriver_out bo;
                double** const bop[5] = {&bo.Tc, &bo.Tg, &bo.groundsnowdepth, &bo.totalSWE, &bo.snowden};
                for (int v = 0; v < 5; ++v) *bop[v] = dst[v] ? dst[v] + k.r0 : nullptr;
                const int rc = run_chunk_to(k.sp, ch, tmean, &bo, (int64_t)ch * k.sp->chunk, true, R);
                if (rc) { fail_here(rc); break; }
            }
            bar.wait();
        }
    };
    std::vector<std::thread> threads;
    for (int t = 0; t < nt; ++t) threads.emplace_back(worker, t);
    for (auto& th : threads) th.join();
    for (int t = 0; t < nt; ++t)
        if (rcs[(size_t)t] != MCF_OK) return mcf::api_fail(rcs[(size_t)t], errs[(size_t)t]);
    return MCF_OK;
}

// applycpp3 over device-resident data: [N][tsteps] -> result / count [tsteps] on the host
static int apply3_device(const double* d_a, int64_t N, int64_t tsteps, int fun, double* result, double* count) {
    int rc;
    Bufs b;
    double *d_r, *d_c = nullptr;
    if ((rc = b.alloc((void**)&d_r, tsteps * 8))) return rc;
    if (count && (rc = b.alloc((void**)&d_c, tsteps * 8))) return rc;
    // enough workgroups to keep the memory system busy whatever the ratio of cells to steps
    int parts = (int)std::min<int64_t>(64, std::max<int64_t>(1, N / 16384));
    if (tsteps * parts < 2048) parts = (int)std::min<int64_t>(64, std::max<int64_t>(parts, (2048 + tsteps - 1) / tsteps));
    double* d_ws;
    if ((rc = b.alloc((void**)&d_ws, tsteps * parts * 16))) return rc;
    hipLaunchKernelGGL(k_apply3_part, dim3((unsigned)parts, (unsigned)tsteps), dim3(256), 0, nullptr, d_a, N, fun, parts, d_ws);
    hipLaunchKernelGGL(k_apply3_fin, dim3((unsigned)((tsteps + 255) / 256)), dim3(256), 0, nullptr, d_ws, tsteps, fun, parts, d_r,
                       d_c);
    S_TRY(hipGetLastError());
    S_TRY(hipMemcpy(result, d_r, (size_t)tsteps * 8, hipMemcpyDeviceToHost));
    if (count) S_TRY(hipMemcpy(count, d_c, (size_t)tsteps * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}
extern "C" int mcf_snowplan_apply3(mcf_snowplan* sp, int32_t chunk, int32_t fun, double* result, double* count) {
    if (!sp || !result || fun < 0 || fun > 3) return mcf::api_fail(MCF_ERR_ARG, "bad mcf_snowplan_apply3 argument");
    if (chunk < 0 || chunk >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    S_TRY(hipSetDevice(sp->device));
    const int ns = std::min(sp->chunk, sp->T - chunk * sp->chunk);
    // (sdepc holds totalSWE after the redistribution; the series are the ones of the chunk run last)
    if (!(sp->series_valid & 4u)) return mcf::api_fail(MCF_ERR_STATE, "snow plan: the chunk's totalSWE series was switched off (mcf_snowplan_set_series)");
    if (fun < 2) return apply3_device(sp->a.sdepc, sp->N, ns, fun, result, count);
    if (sp->mm_chunk != chunk) {
        const int64_t N = sp->N, C = sp->chunk;
        int parts = (int)std::min<int64_t>(64, std::max<int64_t>(1, N / 16384));      // (apply3_device's choice: same partials, same bits)
        if ((int64_t)ns * parts < 2048) parts = (int)std::min<int64_t>(64, std::max<int64_t>(parts, (2048 + ns - 1) / ns));
        const int pmax = 64;
        if (!sp->d_mm) {
            int rc;
            if ((rc = sp->b.alloc((void**)&sp->d_mm, (2 * C * pmax * 2 + 4 * C) * 8))) return rc;
            sp->mm_host.assign((size_t)(4 * C), 0.0);
        }
        double *ws_max = sp->d_mm, *ws_min = sp->d_mm + C * pmax * 2, *d_r = sp->d_mm + 2 * C * pmax * 2;
        hipLaunchKernelGGL(k_apply3_minmax_part, dim3((unsigned)parts, (unsigned)ns), dim3(256), 0, nullptr, (const double*)sp->a.sdepc, N, parts,
                           ws_max, ws_min);
        hipLaunchKernelGGL(k_apply3_fin, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nullptr, (const double*)ws_max, (int64_t)ns, 2, parts, d_r, d_r + C);
        hipLaunchKernelGGL(k_apply3_fin, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nullptr, (const double*)ws_min, (int64_t)ns, 3, parts, d_r + 2 * C,
                           d_r + 3 * C);
        S_TRY(hipGetLastError());
        S_TRY(hipMemcpy(sp->mm_host.data(), d_r, (size_t)(4 * C) * 8, hipMemcpyDeviceToHost));
        sp->mm_chunk = chunk;
    }
    const double* h = sp->mm_host.data() + (fun == 2 ? 0 : 2 * sp->chunk);
    memcpy(result, h, (size_t)ns * 8);
    if (count) memcpy(count, h + sp->chunk, (size_t)ns * 8);
    return MCF_OK;
}
extern "C" int mcf_applycpp3(const double* a, int64_t rows, int64_t cols, int64_t tsteps, int32_t fun, double* result,
                             double* count, int32_t device) {
    if (!a || !result || rows <= 0 || cols <= 0 || tsteps <= 0 || tsteps > 65535 || fun < 0 || fun > 3)
        return mcf::api_fail(MCF_ERR_ARG, "bad applycpp3 argument");
    int rc;
    if ((rc = pick_device(device))) return rc;
    const int64_t N = rows * cols;
    if ((rc = check_room(N * tsteps * 8))) return rc;
    Bufs b;
    const double* d_a;
    UP(d_a, a, N * tsteps);
    return apply3_device(d_a, N, tsteps, (int)fun, result, count);
}

// k_tpi_fine's result divided by its raster mean
__global__ __launch_bounds__(256) void k_scale_by_mean(double* __restrict__ x, int64_t N, const double* __restrict__ sumcount) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) x[i] /= sumcount[0] / sumcount[1];
}
extern "C" int mcf_tpicalc(int64_t rows, int64_t cols, const double* dtm, int32_t af, double tfact, double* tpic, int32_t device) {
    if (!dtm || !tpic || rows <= 0 || cols <= 0) return mcf::api_fail(MCF_ERR_ARG, "mcf_tpicalc: null argument or empty raster");
    if (af < 1) return mcf::api_fail(MCF_ERR_ARG, "mcf_tpicalc: aggregation factor below 1 (terra::aggregate fails)");
    int rc;
    if ((rc = pick_device(device))) return rc;
    const int64_t N = rows * cols;
    if ((rc = check_room(N * 24))) return rc;
    Bufs b;
    const double* d_z;
    double *d_t, *d_ws, *d_m2, *d_cm = nullptr;
    UP(d_z, dtm, N);
    if ((rc = b.alloc((void**)&d_t, N * 8))) return rc;
    if ((rc = b.alloc((void**)&d_ws, 2 * kSumParts * 8))) return rc;
    if ((rc = b.alloc((void**)&d_m2, 16))) return rc;
    TpiGeo g;
    g.rows = rows; g.cols = cols; g.RB = rows; g.hn = 0; g.row0 = 0; g.rows_total = rows; g.af = af;
    g.NItot = (rows + af - 1) / af; g.nJ = (cols + af - 1) / af; g.I0 = 0; g.nI = g.NItot;
    const unsigned gridN = (unsigned)((N + 255) / 256);
    if ((double)af < std::min(rows, cols) / 2.0) {
        if ((rc = b.alloc((void**)&d_cm, g.nI * g.nJ * 8))) return rc;
        hipLaunchKernelGGL(k_tpi_coarse, dim3((unsigned)((g.nI * g.nJ + 255) / 256)), dim3(256), 0, nullptr, d_z, g, d_cm);
        hipLaunchKernelGGL(k_tpi_fine, dim3(gridN), dim3(256), 0, nullptr, d_z, g, (const double*)d_cm, 0.0, tfact, d_t);
    } else {
        launch_sumcount(d_z, N, d_ws, d_m2);
        double h[2];
        S_TRY(hipMemcpy(h, d_m2, 16, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_tpi_fine, dim3(gridN), dim3(256), 0, nullptr, d_z, g, (const double*)nullptr, h[0] / h[1], tfact, d_t);
    }
    launch_sumcount(d_t, N, d_ws, d_m2);
    hipLaunchKernelGGL(k_scale_by_mean, dim3(gridN), dim3(256), 0, nullptr, d_t, N, (const double*)d_m2);
    S_TRY(hipGetLastError());
    S_TRY(hipMemcpy(tpic, d_t, (size_t)N * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}

// ---- the snow-day microclimate inside the chunk loop --------------------------------------------------------------------
extern "C" int mcf_snowplan_reset(mcf_snowplan* sp) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    S_TRY(hipSetDevice(sp->device));
    const int64_t N = sp->N;
    S_TRY(hipMemcpy(sp->d_isnowdc, sp->d_isnowdc0, (size_t)N * 8, hipMemcpyDeviceToDevice));
    S_TRY(hipMemcpy(sp->d_ac, sp->d_ac0, (size_t)N * 4, hipMemcpyDeviceToDevice));
    S_TRY(hipMemcpy(sp->d_ag, sp->d_ag0, (size_t)N * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(k_add_snow, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, sp->d_dtm, sp->d_isnowdg, 1.0, N,
                       sp->d_dtms);
    S_TRY(hipGetLastError());
    sp->prepared = -1;
    return MCF_OK;
}
// Sparse read-back of the plan's device state: `n` cells x the array's depth (planes N apart), for in-run checks of a sample
// of cells against the oracle (tools/bench_snow.py) — the analogue of mcf_plan_fetch_cells for the snow plan.
__global__ __launch_bounds__(256) void k_gather_planes(const double* __restrict__ src, const int32_t* __restrict__ isrc, int64_t N,
                                                       const int64_t* __restrict__ cells, int n, int depth,
                                                       double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * depth) return;
    const int j = t % n, k = t / n;
    const int64_t q = cells[j] + N * (int64_t)k;
    out[t] = src ? src[q] : (double)isrc[q];
}
extern "C" int mcf_snowplan_fetch_cells(mcf_snowplan* sp, int32_t what, const int64_t* cells, int32_t n, double* out,
                                        int32_t* depth_out) {
    if (!sp || !cells || !out || n <= 0) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    const int ns = sp->chunk;
    const double* src = nullptr;
    const int32_t* isrc = nullptr;
    int depth = 1;
    switch (what) {
        case MCF_SNOWPLAN_ISNOWDC: src = sp->d_isnowdc; break;
        case MCF_SNOWPLAN_ISNOWAC: isrc = sp->d_ac; break;
        case MCF_SNOWPLAN_ISNOWAG: isrc = sp->d_ag; break;
        case MCF_SNOWPLAN_SLOPE: src = sp->d_slope; break;
        case MCF_SNOWPLAN_ASPECT: src = sp->d_aspect; break;
        case MCF_SNOWPLAN_SKYVIEW: src = sp->d_svf; break;
        case MCF_SNOWPLAN_WSA: src = sp->d_wsa; depth = 8; break;
        case MCF_SNOWPLAN_HOR: src = sp->d_hor; depth = 24; break;
        case MCF_SNOWPLAN_TC: src = sp->a.Tc; depth = ns; break;
        case MCF_SNOWPLAN_TG: src = sp->a.Tg; depth = ns; break;
        case MCF_SNOWPLAN_SDEPG: src = sp->a.sdepg; depth = ns; break;
        case MCF_SNOWPLAN_SDEN: src = sp->a.sden; depth = ns; break;
        case MCF_SNOWPLAN_TOTALSWE: src = sp->a.sdepc; depth = ns; break;
        default: return mcf::api_fail(MCF_ERR_ARG, "snow plan: unknown array");
    }
    for (int j = 0; j < n; ++j)
        if (cells[j] < 0 || cells[j] >= sp->N) return mcf::api_fail(MCF_ERR_ARG, "snow plan: cell index outside the block");
    S_TRY(hipSetDevice(sp->device));
    Bufs tmp;
    int rc;
    int64_t* d_cells;
    double* d_out;
    if ((rc = tmp.alloc((void**)&d_cells, (int64_t)n * 8))) return rc;
    if ((rc = tmp.alloc((void**)&d_out, (int64_t)n * depth * 8))) return rc;
    S_TRY(hipMemcpy(d_cells, cells, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather_planes, dim3((unsigned)((n * depth + 255) / 256)), dim3(256), 0, nullptr, src, isrc, sp->N,
                       (const int64_t*)d_cells, n, depth, d_out);
    S_TRY(hipGetLastError());
    S_TRY(hipMemcpy(out, d_out, (size_t)n * depth * 8, hipMemcpyDeviceToHost));
    if (depth_out) *depth_out = depth;
    return MCF_OK;
}

// HBM is 288 GB and a chunk's five series of one rank's block are 10 GB: what pass 1 wrote for a chunk with a snow day can
// simply stay.  keep_chunk (after run_chunk and whatever reads the series) hands the chunk's buffers over to the cache and
// gives the plan fresh ones, as long as `reserve_bytes` of device memory stay free; mcf_snowplan_microsnow then reads a kept
// chunk where it lies, and the caller neither restores nor re-runs it.  release_kept hands the sets to a pool for the next
// year (allocation is the expensive part: the cache pays from the second year of a plan on).
extern "C" int mcf_snowplan_keep_chunk(mcf_snowplan* sp, int32_t ch, int64_t reserve_bytes, int32_t* kept) {
    if (!sp || !kept) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    *kept = 0;
    S_TRY(hipSetDevice(sp->device));
    if (sp->kept.size() < (size_t)sp->nchunks) sp->kept.resize((size_t)sp->nchunks);
    if (sp->kept[ch].Tc) { *kept = 1; return MCF_OK; }
    if (sp->series_valid != 31) return MCF_OK;      // (a chunk run with some series switched off cannot be kept)
    const int64_t one = (int64_t)sp->chunk * sp->N * 8;
    const int64_t small = sp->a.tzd ? (int64_t)std::max(sp->chunk / 24, 1) * sp->N * 8 : 0;      // the days' means of Tg beside them
    double* fresh[6] = {};
    if (!sp->pool.empty()) {               // a set an earlier year released
        const mcf_snowplan::Kept f = sp->pool.back();
        sp->pool.pop_back();
        fresh[0] = f.Tc; fresh[1] = f.Tg; fresh[2] = f.sdepc; fresh[3] = f.sdepg; fresh[4] = f.sden; fresh[5] = f.tzd;
    } else {
        size_t free_b = 0, total_b = 0;
        S_TRY(hipMemGetInfo(&free_b, &total_b));
        if ((int64_t)free_b < 5 * one + small + std::max<int64_t>(reserve_bytes, 0)) return MCF_OK;
        if (sp->keep_budget >= 0 && sp->keep_allocated + 5 * one > sp->keep_budget) return MCF_OK;
        for (int v = 0; v < 6; ++v) {
            if (v == 5 && !small) break;
            if (hipMalloc((void**)&fresh[v], (size_t)(v == 5 ? small : one)) != hipSuccess) {      // (another process took the room: not an error)
                (void)hipGetLastError();
                for (int u = 0; u < v; ++u) { (void)hipFree(fresh[u]); sp->kb.p.pop_back(); }
                return MCF_OK;
            }
            sp->kb.p.push_back(fresh[v]);
        }
        sp->keep_allocated += 5 * one;
    }
    S_TRY(hipDeviceSynchronize());         // (the chunk's kernels are done before its buffers change hands)
    ModelArgs& a = sp->a;
    mcf_snowplan::Kept k;
    k.Tc = a.Tc; k.Tg = a.Tg; k.sdepc = a.sdepc; k.sdepg = a.sdepg; k.sden = a.sden; k.tzd = a.tzd;
    a.Tc = fresh[0]; a.Tg = fresh[1]; a.sdepc = fresh[2]; a.sdepg = fresh[3]; a.sden = fresh[4]; a.tzd = fresh[5];
    sp->kept[ch] = k;
    sp->mm_chunk = -1;                     // (the plan's buffers are fresh ones now)
    *kept = 1;
    return MCF_OK;
}
extern "C" int mcf_snowplan_set_series(mcf_snowplan* sp, uint32_t mask) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    if (mask > 31u) return mcf::api_fail(MCF_ERR_ARG, "mcf_snowplan_set_series: five series, mask <= 31");
    sp->series_mask = mask;
    return MCF_OK;
}
// would mcf_snowplan_keep_chunk keep a chunk now? (a pooled set, or room for a new one beside `reserve_bytes`)
extern "C" int mcf_snowplan_can_keep(mcf_snowplan* sp, int64_t reserve_bytes, int32_t* yes) {
    if (!sp || !yes) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    *yes = 0;
    if (!sp->pool.empty()) { *yes = 1; return MCF_OK; }
    S_TRY(hipSetDevice(sp->device));
    size_t free_b = 0, total_b = 0;
    S_TRY(hipMemGetInfo(&free_b, &total_b));
    const int64_t one = (int64_t)sp->chunk * sp->N * 8;
    *yes = (int64_t)free_b >= 5 * one + std::max<int64_t>(reserve_bytes, 0) ? 1 : 0;
    if (sp->keep_budget >= 0 && sp->keep_allocated + 5 * one > sp->keep_budget) *yes = 0;
    return MCF_OK;
}
extern "C" int mcf_snowplan_set_keep_budget(mcf_snowplan* sp, int64_t bytes) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    sp->keep_budget = bytes;
    return MCF_OK;
}
extern "C" int mcf_snowplan_release_kept(mcf_snowplan* sp) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    S_TRY(hipSetDevice(sp->device));
    S_TRY(hipDeviceSynchronize());
    // the sets go to a pool for the next year's pass 1 (allocating 10 GB takes a quarter of a second: a year's cache costs more
    // to allocate than it saves, so it is allocated once per plan); the plan's allocation lists still own every buffer
    for (auto& k : sp->kept)
        if (k.Tc) sp->pool.push_back(k);
    sp->kept.clear();
    return MCF_OK;
}

// The state a chunk starts from — the pack depth handed over, the snow surface the terrain refresh reads, the two ages:
// 24 bytes per cell.  Pass 1 of the snow-day microclimate checkpoints every chunk; pass 2 then restores and re-runs only the
// chunks that hold a snow day (the others contribute nothing but the no-snow solver's days).
extern "C" int mcf_snowplan_checkpoint(mcf_snowplan* sp, int32_t ch) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    S_TRY(hipSetDevice(sp->device));
    const size_t N = (size_t)sp->N;
    if (sp->ckpt.size() < (size_t)sp->nchunks) sp->ckpt.resize((size_t)sp->nchunks, nullptr);
    if (!sp->ckpt[ch]) {
        int rc;
        if ((rc = sp->b.alloc((void**)&sp->ckpt[ch], (int64_t)N * 24))) return rc;
    }
    char* c = sp->ckpt[ch];
    S_TRY(hipMemcpyAsync(c, sp->d_isnowdc, N * 8, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(c + N * 8, sp->d_dtms, N * 8, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(c + N * 16, sp->d_ac, N * 4, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(c + N * 20, sp->d_ag, N * 4, hipMemcpyDeviceToDevice, nullptr));
    return MCF_OK;
}
extern "C" int mcf_snowplan_restore(mcf_snowplan* sp, int32_t ch) {
    if (!sp) return mcf::api_fail(MCF_ERR_ARG, "null snow plan");
    if (ch < 0 || (size_t)ch >= sp->ckpt.size() || !sp->ckpt[ch])
        return mcf::api_fail(MCF_ERR_STATE, "snow plan: no checkpoint of this chunk (mcf_snowplan_checkpoint in the first pass)");
    S_TRY(hipSetDevice(sp->device));
    const size_t N = (size_t)sp->N;
    const char* c = sp->ckpt[ch];
    S_TRY(hipMemcpyAsync(sp->d_isnowdc, c, N * 8, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(sp->d_dtms, c + N * 8, N * 8, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(sp->d_ac, c + N * 16, N * 4, hipMemcpyDeviceToDevice, nullptr));
    S_TRY(hipMemcpyAsync(sp->d_ag, c + N * 20, N * 4, hipMemcpyDeviceToDevice, nullptr));
    sp->prepared = -1;
    return MCF_OK;
}
extern "C" int mcf_snowplan_meand_accumulate(mcf_snowplan* sp, int32_t ch, const int32_t* snowday) {
    if (!sp || !snowday) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    S_TRY(hipSetDevice(sp->device));
    const int ns = std::min(sp->chunk, sp->T - ch * sp->chunk), nd = ns / 24;
    int nsnow = 0;
    for (int d = 0; d < nd; ++d) nsnow += snowday[d] != 0;
    if (ch == 0 || sp->sumD_steps < 0) sp->sumD_steps = 0;
    if (ch == 0) S_TRY(hipMemset(sp->d_sumD, 0, (size_t)sp->N * 8));
    if (ch == 0) S_TRY(hipMemset(sp->d_sden_na, 0, (size_t)sp->N * 4));
    if (nsnow == 0) return MCF_OK;
    if (!(sp->series_valid & 16u)) return mcf::api_fail(MCF_ERR_STATE, "snow plan: the chunk's snow density series was switched off (mcf_snowplan_set_series)");
    S_TRY(hipMemcpy(sp->d_daymap, snowday, (size_t)nd * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_meand_accumulate, dim3((unsigned)((sp->N + 255) / 256)), dim3(256), 0, nullptr, sp->a.sden, sp->a.hgt,
                       sp->N, nd, (const int32_t*)sp->d_daymap, sp->sumD_steps == 0 ? 1 : 0, sp->d_sumD, sp->d_sden_na);
    S_TRY(hipGetLastError());
    sp->sumD_steps += (int64_t)nsnow * 24;
    return MCF_OK;
}
extern "C" int mcf_snowplan_micro_setup(mcf_snowplan* sp, const mcf_snow_inputs* sub, const int32_t* sub_of_day, int32_t ndays,
                                        double reqhgt, double mat, const int32_t outsel[MCF_NOUT], int32_t reuse_static) {
    if (!sp || !sub || !sub_of_day || !outsel) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    int rc;
    if ((rc = common_checks(sub))) return rc;
    // Array weather (gridmicrosnow2, cpp:5058-5214): `sub` then carries the WHOLE series — obstime [T], clim.* as [rows,cols,T],
    // clim.winddir [T], other.lats / lons — and sub_of_day says which of its days form the snow-day subset series (subsetting
    // nine arrays on the host would copy them; here a chunk's snow days are uploaded when the chunk's microclimate runs).
    const bool maf = sub->array_forcing != 0;
    if (maf != sp->af) return mcf::api_fail(MCF_ERR_ARG, "micro set-up: the weather's geometry (data.frame / array) is not the snow plan's");
    if (sub->rows != sp->rows || sub->cols != sp->cols) return mcf::api_fail(MCF_ERR_ARG, "micro set-up: the raster is not the plan's");
    if (ndays * 24 > sp->T) return mcf::api_fail(MCF_ERR_ARG, "micro set-up: more days than the series");
    if (maf && sub->tsteps != sp->T) return mcf::api_fail(MCF_ERR_ARG, "micro set-up, array weather: the whole series is expected");
    if (outsel[MCF_OUT_SOILM] && !sub->other.Smax) return mcf::api_fail(MCF_ERR_ARG, "soilm requested but other$Smax is null");
    S_TRY(hipSetDevice(sp->device));
    sp->mb.release_all();
    sp->micro_ready = false;
    // reuse_static: vegetation, terrain and Smax are those of the previous set-up (a year's day list changes, the raster does
    // not) — 45 matrices that are not uploaded again
    const bool keep_static = reuse_static && sp->micro_static;
    const MicroArgs prev = sp->ma;
    if (!keep_static) { sp->mbs.release_all(); sp->micro_static = false; }
    const int64_t N = sp->N;
    int nsub = 0;
    for (int d = 0; d < ndays; ++d) nsub += sub_of_day[d] >= 0;
    const int T = maf ? nsub * 24 : (int)sub->tsteps;          // steps of the subset series
    for (int d = 0; d < ndays; ++d)
        if (sub_of_day[d] >= 0 && (int64_t)sub_of_day[d] * 24 + 24 > T) return mcf::api_fail(MCF_ERR_ARG, "micro set-up: day map points past the subset series");
    if (maf) {       // the subset's days in the order of the series (the albedo clock and the kernel's addressing walk them so)
        int next = 0;
        for (int d = 0; d < ndays; ++d)
            if (sub_of_day[d] >= 0 && sub_of_day[d] != next++) return mcf::api_fail(MCF_ERR_ARG, "micro set-up, array weather: the day map must number the snow days 0, 1, 2, ... in order");
    }
    sp->sub_of_day.assign(sub_of_day, sub_of_day + ndays);
    sp->micro_af = maf;
    Bufs& b = sp->mb;
    MicroArgs& a = sp->ma;
    memset(&a, 0, sizeof a);
    a.N = N; a.tsteps = T; a.reqhgt = reqhgt; a.mat = mat; a.zref = sub->other.zref;
    int y0 = sub->obstime.year[0];          // the SUBSET series' first year (array weather: `sub` is the whole series)
    if (maf)
        for (int d = 0; d < ndays; ++d)
            if (sub_of_day[d] == 0) { y0 = sub->obstime.year[(size_t)d * 24]; break; }
    a.hiy = (y0 % 4 == 0 && (y0 % 100 != 0 || y0 % 400 == 0)) ? 366 * 24 : 365 * 24;   // cpp:4984
    if (keep_static) {
        a.pai = prev.pai; a.hgt = prev.hgt; a.leaft = prev.leaft; a.clump = prev.clump; a.paia = prev.paia; a.leafd = prev.leafd;
        a.leafden = prev.leafden; a.slope = prev.slope; a.aspect = prev.aspect; a.skyview = prev.skyview; a.wsa = prev.wsa;
        a.hor = prev.hor; a.Smax = prev.Smax;
        if (outsel[MCF_OUT_SOILM] && !a.Smax) return mcf::api_fail(MCF_ERR_ARG, "micro set-up: soilm was not part of the static set-up being reused");
    } else {
        Bufs& b = sp->mbs;       // (shadows the series' buffer set: these uploads outlive the next set-up)
        UP(a.pai, sub->vegp.pai, N);
        UP(a.hgt, sub->vegp.hgt, N);
        UP(a.leaft, sub->vegp.leaft, N);
        UP(a.clump, sub->vegp.clump, N);
        UP(a.paia, sub->vegp.paia, N);
        UP(a.leafd, sub->vegp.leafd, N);
        UP(a.leafden, sub->vegp.leafden, N);
        UP(a.slope, sub->other.slope, N);
        UP(a.aspect, sub->other.aspect, N);
        UP(a.skyview, sub->other.skyview, N);
        UP(a.wsa, sub->other.wsa, 8 * N);
        UP(a.hor, sub->other.hor, 24 * N);
        if (sub->other.Smax) UP(a.Smax, sub->other.Smax, N);
        sp->micro_static = true;
    }
    if (maf) {
        if (T == 0) return mcf::api_fail(MCF_ERR_ARG, "micro set-up, array weather: no snow day");
        const double* hm[9] = {sub->clim.temp, sub->clim.relhum, sub->clim.pres, sub->clim.swdown, sub->clim.difrad, sub->clim.lwdown,
                               sub->clim.windspeed, sub->clim.precip, sub->clim.umu};
        for (const double* q : hm)
            if (!q) return mcf::api_fail(MCF_ERR_ARG, "micro set-up, array weather: a weather array is null");
        if (!sub->other.lats || !sub->other.lons || !sub->clim.winddir) return mcf::api_fail(MCF_ERR_ARG, "micro set-up, array weather: lats / lons / winddir");
        UP(a.lats, sub->other.lats, N);
        UP(a.lons, sub->other.lons, N);
        // the subset series' date rows (obstime and wind direction of the snow days)
        std::vector<int32_t> yr((size_t)T), mo((size_t)T), dy((size_t)T);
        std::vector<double> hr((size_t)T), wd((size_t)T);
        for (int d = 0; d < ndays; ++d) {
            if (sub_of_day[d] < 0) continue;
            for (int hh = 0; hh < 24; ++hh) {
                const size_t q = (size_t)sub_of_day[d] * 24 + hh, w = (size_t)d * 24 + hh;
                yr[q] = sub->obstime.year[w]; mo[q] = sub->obstime.month[w]; dy[q] = sub->obstime.day[w]; hr[q] = sub->obstime.hour[w];
                wd[q] = sub->clim.winddir[w];
            }
        }
        mcf_snow_inputs ds = *sub;
        ds.tsteps = T;
        ds.obstime.year = yr.data(); ds.obstime.month = mo.data(); ds.obstime.day = dy.data(); ds.obstime.hour = hr.data();
        ds.clim.winddir = wd.data();
        if ((rc = build_step_tables(b, &ds, true, false, false, &a.rows, &a.dates, &a.mxtc1))) return rc;
        S_TRY(hipDeviceSynchronize());      // (the table kernels have read the host vectors' uploads)
        // the slabs a chunk's snow days are uploaded into, addressed with the subset series' step numbers (mcf_snowplan_microsnow
        // shifts the bases by the chunk's first subset day)
        const int64_t CN = (int64_t)sp->chunk * N;
        for (int f = 0; f < 9; ++f) {
            sp->h_micro[f] = hm[f];
            if ((rc = b.alloc((void**)&sp->d_micro[f], CN * 8))) return rc;
        }
        // mxtc and the albedo clock at every subset day's start: the series' temperature and precipitation streamed once, a run of
        // consecutive snow days at a time through the first two slabs
        if ((rc = b.alloc((void**)&a.mxtc, N * 8))) return rc;
        if ((rc = b.alloc((void**)&a.hs0, N * (int64_t)std::max(nsub, 1) * 4))) return rc;
        int32_t* d_hs;
        if ((rc = b.alloc((void**)&d_hs, N * 4))) return rc;
        S_TRY(hipMemset(a.mxtc, 0, (size_t)N * 8));
        const int cd = sp->chunk / 24;
        for (int d = 0; d < ndays;) {
            if (sub_of_day[d] < 0) { ++d; continue; }
            int e = d;
            while (e < ndays && sub_of_day[e] >= 0 && e - d < cd) ++e;
            const int64_t off = (int64_t)d * 24 * N, n = (int64_t)(e - d) * 24 * N;
            S_TRY(hipMemcpyAsync(sp->d_micro[0], hm[0] + off, (size_t)n * 8, hipMemcpyHostToDevice, nullptr));
            S_TRY(hipMemcpyAsync(sp->d_micro[1], hm[7] + off, (size_t)n * 8, hipMemcpyHostToDevice, nullptr));
            hipLaunchKernelGGL(k_micro_scan, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, (const double*)sp->d_micro[0],
                               (const double*)sp->d_micro[1], N, sub_of_day[d], e - d, a.hgt, a.mxtc, d_hs, a.hs0);
            S_TRY(hipGetLastError());
            d = e;
        }
    } else {
    if ((rc = build_step_tables(b, sub, false, false, false, &a.rows, &a.dates, &a.mxtc1))) return rc;
    UP(a.temp, sub->clim.temp, T);
    UP(a.relhum, sub->clim.relhum, T);
    UP(a.pres, sub->clim.pres, T);
    UP(a.swdown, sub->clim.swdown, T);
    UP(a.difrad, sub->clim.difrad, T);
    UP(a.lwdown, sub->clim.lwdown, T);
    UP(a.windspeed, sub->clim.windspeed, T);
    UP(a.precip, sub->clim.precip, T);
    UP(a.umu, sub->clim.umu, T);
    }
    if (!maf) {
        MicroMet* mm;
        if ((rc = b.alloc((void**)&mm, (int64_t)T * sizeof(MicroMet)))) return rc;
        hipLaunchKernelGGL(k_micro_steps, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, nullptr, a.temp, a.relhum, a.pres, a.mxtc1, T, mm);
        a.mmet = mm;
        MicroStep* ms;
        if ((rc = b.alloc((void**)&ms, (int64_t)T * sizeof(MicroStep)))) return rc;
        hipLaunchKernelGGL(k_micro_pack, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, nullptr, a.rows, a.mmet, a.temp, a.pres, a.windspeed,
                           a.swdown, a.difrad, a.lwdown, a.umu, T, ms);
        a.mstep = ms;
    }
    // the chunk's snow series, where mcf_snowplan_run_chunk leaves them (sdepc holds totalSWE after the redistribution)
    a.sTc = sp->a.Tc; a.sTg = sp->a.Tg; a.swe = sp->a.sdepc; a.sdepg = sp->a.sdepg; a.sden = sp->a.sden;
    // meanDsnow of the first pass (mcf_snowplan_meand_accumulate over every chunk)
    hipLaunchKernelGGL(k_meand_finish, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, (const double*)sp->d_sumD,
                       (const int32_t*)sp->d_sden_na, a.hgt, N, (double)std::max<int64_t>(sp->sumD_steps, 1), sp->d_meanD);
    S_TRY(hipGetLastError());
    a.meanD = sp->d_meanD;
    memcpy(sp->outsel, outsel, sizeof sp->outsel);
    S_TRY(hipDeviceSynchronize());
    sp->micro_ready = true;
    return MCF_OK;
}
// one lane per cell: a cell that is not under snow at every step of the days (or has no vegetation height: gridmicrosnow1 skips
// it) clears its tile's flag
__global__ __launch_bounds__(256) void k_tiles_covered(const double* __restrict__ swe, const double* __restrict__ hgt, int64_t N, int k0,
                                                       int nsteps, int cpb, uint8_t* __restrict__ flag) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    bool cov = !isnan(hgt[c]);
    for (int k = 0; cov && k < nsteps; ++k) cov = swe[c + N * (int64_t)(k0 + k)] > 0.0;
    if (!cov) flag[c / cpb] = 0;
}
extern "C" int mcf_snowplan_covered_tiles(mcf_snowplan* sp, mcf_plan* plan, int32_t ch, int32_t day, int32_t ndays, uint8_t* skip_tile,
                                          int64_t n_tiles, int64_t* n_covered) {
    if (!sp || !plan || !skip_tile || !n_covered) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (!sp->micro_ready) return mcf::api_fail(MCF_ERR_STATE, "snow plan: mcf_snowplan_micro_setup first");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    const int ns = std::min(sp->chunk, sp->T - ch * sp->chunk);
    if (day < 0 || ndays < 1 || (day + ndays) * 24 > ns) return mcf::api_fail(MCF_ERR_ARG, "days outside the chunk");
    hipStream_t stream;
    int64_t N;
    int device, slot_days, rc;
    mcf::RingView views[MCF_NOUT];
    int32_t has[MCF_NOUT];
    if ((rc = mcf::plan_ring_views(plan, 0, views, has, &stream, &N, &device, &slot_days))) return rc;
    int cpb = 0;
    bool all_sel = true;
    for (int v = 0; v < MCF_NOUT; ++v)
        if (has[v]) { cpb = views[v].cpb; all_sel = all_sel && sp->outsel[v] != 0; }
    if (cpb <= 0) return mcf::api_fail(MCF_ERR_STATE, "the snow-day microclimate needs a plan with the tiled ring (reqhgt >= 0)");
    if (N != sp->N || device != sp->device) return mcf::api_fail(MCF_ERR_ARG, "snow plan and solver plan differ in raster or device");
    if (n_tiles != (N + cpb - 1) / cpb) return mcf::api_fail(MCF_ERR_ARG, "n_tiles is not the solver plan's number of tiles");
    *n_covered = 0;
    memset(skip_tile, 0, (size_t)n_tiles);
    if (!all_sel) return MCF_OK;           // an output the snow microclimate does not produce stays the solver's everywhere
    S_TRY(hipSetDevice(sp->device));
    if (!sp->d_tflag || sp->tflag_cap < n_tiles) {
        if ((rc = sp->b.alloc((void**)&sp->d_tflag, n_tiles))) return rc;
        sp->tflag_cap = n_tiles;
    }
    const bool k = (size_t)ch < sp->kept.size() && sp->kept[ch].Tc;
    if (!k && !(sp->series_valid & 4u)) return mcf::api_fail(MCF_ERR_STATE, "snow plan: the chunk's totalSWE series was switched off (mcf_snowplan_set_series)");
    const double* swe = k ? sp->kept[ch].sdepc : sp->a.sdepc;
    S_TRY(hipMemset(sp->d_tflag, 1, (size_t)n_tiles));
    hipLaunchKernelGGL(k_tiles_covered, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, swe, sp->ma.hgt, N, day * 24, ndays * 24, cpb,
                       sp->d_tflag);
    S_TRY(hipGetLastError());
    S_TRY(hipMemcpy(skip_tile, sp->d_tflag, (size_t)n_tiles, hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (int64_t t = 0; t < n_tiles; ++t) n += skip_tile[t] != 0;
    *n_covered = n;
    return MCF_OK;
}
// one lane per cell: 1 where the cell's solver values of the days survive the merge (k_tiles_covered's test, negated)
__global__ __launch_bounds__(256) void k_cells_free(const double* __restrict__ swe, const double* __restrict__ hgt, int64_t N, int k0,
                                                    int nsteps, uint8_t* __restrict__ need, unsigned long long* __restrict__ count) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool cov = false;
    if (c < N) {
        cov = !isnan(hgt[c]);
        for (int k = 0; cov && k < nsteps; ++k) cov = swe[c + N * (int64_t)(k0 + k)] > 0.0;
        need[c] = cov ? 0 : 1;
    }
    __shared__ int s_n;                     // (one atomic per workgroup: a wave-level one on a single address serialises the launch)
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t b = __ballot(c < N && !cov);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_n, (int)__popcll(b));
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(count, (unsigned long long)s_n);
}
extern "C" int mcf_snowplan_free_cells(mcf_snowplan* sp, mcf_plan* plan, int32_t ch, int32_t day, int32_t ndays, const uint8_t** need_cell,
                                       int64_t* n_need) {
    if (!sp || !plan || !need_cell || !n_need) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (!sp->micro_ready) return mcf::api_fail(MCF_ERR_STATE, "snow plan: mcf_snowplan_micro_setup first");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    const int ns = std::min(sp->chunk, sp->T - ch * sp->chunk);
    if (day < 0 || ndays < 1 || (day + ndays) * 24 > ns) return mcf::api_fail(MCF_ERR_ARG, "days outside the chunk");
    hipStream_t stream;
    int64_t N;
    int device, slot_days, rc;
    mcf::RingView views[MCF_NOUT];
    int32_t has[MCF_NOUT];
    if ((rc = mcf::plan_ring_views(plan, 0, views, has, &stream, &N, &device, &slot_days))) return rc;
    bool all_sel = true, any = false;
    for (int v = 0; v < MCF_NOUT; ++v)
        if (has[v]) { any = true; all_sel = all_sel && sp->outsel[v] != 0; }
    if (!any) return mcf::api_fail(MCF_ERR_STATE, "the solver plan holds no output");
    if (N != sp->N || device != sp->device) return mcf::api_fail(MCF_ERR_ARG, "snow plan and solver plan differ in raster or device");
    S_TRY(hipSetDevice(sp->device));
    if (!sp->d_need || sp->need_cap < N) {
        if ((rc = sp->b.alloc((void**)&sp->d_need, N + 16))) return rc;
        sp->need_cap = N;
    }
    *need_cell = sp->d_need;
    unsigned long long* d_count = (unsigned long long*)(sp->d_need + (N + 7) / 8 * 8);
    if (!all_sel) {                        // an output the snow microclimate does not produce stays the solver's everywhere
        S_TRY(hipMemset(sp->d_need, 1, (size_t)N));
        *n_need = N;
        return MCF_OK;
    }
    const bool k = (size_t)ch < sp->kept.size() && sp->kept[ch].Tc;
    if (!k && !(sp->series_valid & 4u)) return mcf::api_fail(MCF_ERR_STATE, "snow plan: the chunk's totalSWE series was switched off (mcf_snowplan_set_series)");
    const double* swe = k ? sp->kept[ch].sdepc : sp->a.sdepc;
    S_TRY(hipMemset(d_count, 0, 8));
    hipLaunchKernelGGL(k_cells_free, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, swe, sp->ma.hgt, N, day * 24, ndays * 24,
                       sp->d_need, d_count);
    S_TRY(hipGetLastError());
    unsigned long long n = 0;
    S_TRY(hipMemcpy(&n, d_count, 8, hipMemcpyDeviceToHost));       // (also: the flags are complete)
    *n_need = (int64_t)n;
    return MCF_OK;
}
extern "C" int mcf_snowplan_microsnow(mcf_snowplan* sp, mcf_plan* plan, int32_t ch, int32_t slot, const int32_t* nosnowday) {
    if (!sp || !plan || !nosnowday) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    if (!sp->micro_ready) return mcf::api_fail(MCF_ERR_STATE, "snow plan: mcf_snowplan_micro_setup first");
    if (ch < 0 || ch >= sp->nchunks) return mcf::api_fail(MCF_ERR_ARG, "chunk out of range");
    MicroRingArgs q;
    memset(&q, 0, sizeof q);
    hipStream_t stream;
    int64_t N;
    int device, slot_days, rc;
    mcf::RingView views[MCF_NOUT];
    int32_t has[MCF_NOUT];
    if ((rc = mcf::plan_ring_views(plan, slot, views, has, &stream, &N, &device, &slot_days))) return rc;
    for (int v = 0; v < MCF_NOUT; ++v) {
        if (!has[v]) continue;
        if (!q.held) { q.base0 = const_cast<double*>(views[v].base); q.ring = views[v]; }
        q.held |= 1u << v;
    }
    if (!q.held || q.ring.cpb <= 0) return mcf::api_fail(MCF_ERR_STATE, "the snow-day microclimate needs a plan with the tiled ring (reqhgt >= 0)");
    {   // the held variables' blocks follow each other at one stride (mcf_kernels.h RingView: [tile][day][variable][block])
        int rank = 0;
        const double* prev = nullptr;
        for (int v = 0; v < MCF_NOUT; ++v) {
            if (!has[v]) continue;
            if (rank == 1) q.vstride = views[v].base - prev;
            if (rank >= 1 && views[v].base - prev != q.vstride) return mcf::api_fail(MCF_ERR_STATE, "snow-day microclimate: the ring's variables are not equally spaced");
            prev = views[v].base;
            ++rank;
        }
    }
    if (N != sp->N || device != sp->device) return mcf::api_fail(MCF_ERR_ARG, "snow plan and solver plan differ in raster or device");
    const int ns = std::min(sp->chunk, sp->T - ch * sp->chunk), nd = ns / 24, day0 = ch * (sp->chunk / 24);
    if (nd > slot_days) return mcf::api_fail(MCF_ERR_ARG, "the ring slot holds fewer days than a snow chunk");
    if (day0 + nd > (int)sp->sub_of_day.size()) return mcf::api_fail(MCF_ERR_ARG, "chunk past the day map of the micro set-up");
    S_TRY(hipSetDevice(sp->device));
    int any = 0;
    for (int d = 0; d < nd; ++d) any |= sp->sub_of_day[day0 + d] >= 0;
    if (!any) return MCF_OK;
    // (the snow kernels run on the null stream, which orders itself against the plan's non-blocking stream only through the
    // explicit waits here: the solver's launches of this chunk first, this kernel before the plan's next use of the slot)
    S_TRY(hipStreamSynchronize(stream));
    S_TRY(hipMemcpy(sp->d_daymap, sp->sub_of_day.data() + day0, (size_t)nd * 4, hipMemcpyHostToDevice));
    S_TRY(hipMemcpy(sp->d_nosnow, nosnowday, (size_t)nd * 4, hipMemcpyHostToDevice));
    q.m = sp->ma;
    q.m.day0 = 0;
    {   // the chunk's snow series: where pass 1 left them if the chunk was kept, the plan's working buffers otherwise
        const bool k = (size_t)ch < sp->kept.size() && sp->kept[ch].Tc;
        if (!k && sp->series_valid != 31) return mcf::api_fail(MCF_ERR_STATE, "snow plan: the chunk was run with series switched off (mcf_snowplan_set_series)");
        q.m.sTc = k ? sp->kept[ch].Tc : sp->a.Tc;
        q.m.sTg = k ? sp->kept[ch].Tg : sp->a.Tg;
        q.m.tzd = k ? sp->kept[ch].tzd : sp->a.tzd;          // (written with the Tg series: every series of the chunk is there, checked above)
        q.m.swe = k ? sp->kept[ch].sdepc : sp->a.sdepc;
        q.m.sdepg = k ? sp->kept[ch].sdepg : sp->a.sdepg;
        q.m.sden = k ? sp->kept[ch].sden : sp->a.sden;
    }
    for (int v = 0; v < MCF_NOUT; ++v) q.sel |= sp->outsel[v] ? 1u << v : 0u;
    q.daymap = sp->d_daymap; q.nosnow = sp->d_nosnow; q.ndays = nd;
    static const bool old_shape = getenv("MCF_MICRORING_OLD") != nullptr;      // A/B
    if (sp->micro_af) {
        // array weather: the chunk's snow days of the nine series into the slabs; the kernel addresses a series with the subset's
        // step number, so the bases are shifted back by the chunk's first subset day
        int sub_first = -1;
        for (int d = 0; d < nd; ++d) {
            const int sb = sp->sub_of_day[day0 + d];
            if (sb < 0) continue;
            if (sub_first < 0) sub_first = sb;
            for (int f = 0; f < 9; ++f)
                S_TRY(hipMemcpyAsync(sp->d_micro[f] + (int64_t)(sb - sub_first) * 24 * N, sp->h_micro[f] + (int64_t)(day0 + d) * 24 * N,
                                     (size_t)24 * N * 8, hipMemcpyHostToDevice, nullptr));
        }
        const int64_t back = (int64_t)sub_first * 24 * N;
        q.m.temp = sp->d_micro[0] - back; q.m.relhum = sp->d_micro[1] - back; q.m.pres = sp->d_micro[2] - back;
        q.m.swdown = sp->d_micro[3] - back; q.m.difrad = sp->d_micro[4] - back; q.m.lwdown = sp->d_micro[5] - back;
        q.m.windspeed = sp->d_micro[6] - back; q.m.precip = sp->d_micro[7] - back; q.m.umu = sp->d_micro[8] - back;
        hipLaunchKernelGGL(k_microsnow_ring<true>, dim3((unsigned)((N + 63) / 64), (unsigned)nd), dim3(256), 0, nullptr, q, (const void*)q.m.dates,
                           q.daymap, q.nosnow);
    } else if (q.ring.cpb == 21 && !old_shape)
        hipLaunchKernelGGL(k_microsnow_tiles, dim3((unsigned)((N + kRtCells - 1) / kRtCells)), dim3(64 * kRtWaves), 0, nullptr, q,
                           (const MicroStep*)q.m.mstep, q.daymap, q.nosnow);
    else
        hipLaunchKernelGGL(k_microsnow_ring<false>, dim3((unsigned)((N + 63) / 64), (unsigned)nd), dim3(256), 0, nullptr, q, q.m.mstep,
                           q.daymap, q.nosnow);
    S_TRY(hipGetLastError());
    S_TRY(hipDeviceSynchronize());
    return MCF_OK;
}

extern "C" int32_t mcf_snowenv_from_name(const char* name) {
    if (!name) return MCF_SNOWENV_ALPINE;
    if (!strcmp(name, "Maritime")) return MCF_SNOWENV_MARITIME;
    if (!strcmp(name, "Prairie")) return MCF_SNOWENV_PRAIRIE;
    if (!strcmp(name, "Tundra")) return MCF_SNOWENV_TUNDRA;
    if (!strcmp(name, "Taiga")) return MCF_SNOWENV_TAIGA;
    return MCF_SNOWENV_ALPINE;
}
extern "C" int mcf_gridmodelsnow1(const mcf_snow_inputs* in, mcf_snowmodel_out* out, int32_t device) {
    return run_snowmodel(in, out, device, false);
}
extern "C" int mcf_gridmodelsnow2(const mcf_snow_inputs* in, mcf_snowmodel_out* out, int32_t device) {
    return run_snowmodel(in, out, device, true);
}
extern "C" int mcf_gridmicrosnow1(const mcf_snow_inputs* in, const mcf_snowm* snowm, double reqhgt, double mat,
                                  const int32_t out[MCF_NOUT], mcf_outputs* micro, int32_t device) {
    return run_microsnow(in, snowm, reqhgt, mat, out, micro, device, false);
}
extern "C" int mcf_gridmicrosnow2(const mcf_snow_inputs* in, const mcf_snowm* snowm, double reqhgt, double mat,
                                  const int32_t out[MCF_NOUT], mcf_outputs* micro, int32_t device) {
    return run_microsnow(in, snowm, reqhgt, mat, out, micro, device, true);
}
