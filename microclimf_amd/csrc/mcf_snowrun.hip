// mcf_snowrun.hip — `runmicro(..., snow = TRUE)` with data.frame weather, device-resident, behind ONE C entry.
//
// The reference's `.runmicrosnow1` (R/internal.R:3581-3659) takes the whole year's snow series (`smod`, the arrays
// `.snowmodel1` returned, R/internal.R:2498-2619) from host memory, solves the days with a snow-free cell somewhere with the
// grid solver, the days with snow somewhere with gridmicrosnow1 (src/microclimfCpp.cpp:4894-5056), and merges by day.  Here
// the snow series never leave the device: the entry drives `.snowmodel1`'s chunk loop (include/mcf.h mcf_snowplan_*) and
// `.runmicrosnow1`'s two models in the solver's output ring — the sequence tools/bench_snow.py times for BASELINE configs[4]
// — and only the merged output (and, if asked for, the snow series) crosses PCIe.  Host orchestration only: every kernel is
// launched through the plan entry points of mcf_api.hip / mcf_snow.hip.
//
//   pass 1   per 5-day chunk: [checkpoint] -> snow surface (halo rows of neighbouring row blocks through host memory) ->
//            terrain refresh + tpi -> gridmodelsnow1 + redistribution -> applycpp3 max / min of totalSWE -> snowdaysfun
//            (src/microclimfCpp.cpp:5531-5550) -> running sum of the snow damping depth; a snow chunk's series stay in HBM
//            while room remains
//   between  gridmicrosnow1's set-up on the snow-day SUBSET of the caller's whole-series inputs (day subsetting here, what
//            `subsetpointmodel(micropoint, days = snowdays)` does in R), the solver's maximum temperature over the no-snow subset
//   pass 2   per chunk: restore + re-run the snow chunk unless its series were kept; the solver on the chunk's no-snow days at
//            their own place in the ring slot; k_microsnow_ring over it; the slot's merged days to the caller's arrays
//
// Row blocks: the raster is cut into contiguous row blocks, block b on devices[b % n_devices], one host thread per device
// (as mcf_snowmodel1_multi): per chunk the blocks' snow surfaces meet in one whole-raster host array, the two raster-wide
// means and the per-step extremes of totalSWE are combined in block order.  One block = the single-device sequence, bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcf.h"

namespace mcf {
int api_fail(int code, const std::string& msg);   // mcf_api.hip
}

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

double na_real_host() {
    union { uint64_t u; double d; } na;
    na.u = 0x7FF00000000007A2ULL;
    return na.d;
}

struct Block {
    int64_t r0 = 0, nr = 0;
    int device = 0;
    // the block's rows of the snow model's rasters (the snow plan uploads from dense arrays)
    std::vector<double> pai, hgt, leaft, clump, dc, dg, dtm, ext;
    std::vector<int32_t> ac, ag;
    mcf_snowplan* sp = nullptr;
    mcf_plan* plan = nullptr;
    double s = 0, n = 0, ts = 0, tn = 0, twi_s = 0;
    int64_t twi_n = 0;
    std::vector<double> mx, cmx, mn, cmn;      // applycpp3 of the chunk just run
    std::vector<char> kept;                    // per chunk: its series stayed on the device
    mcf_grid_inputs gsub{};                    // the block's view of the solver's inputs (array weather: a chunk's forcing is uploaded from it)
    // pass 2: the block's rows of gridmicrosnow1's static rasters
    std::vector<double> m_pai, m_hgt, m_leaft, m_clump, m_paia, m_leafd, m_leafden, m_slope, m_aspect, m_svf, m_wsa, m_hor, m_smax;
};

}  // namespace

struct mcf_snowrun {
    int64_t R = 0, C = 0, T = 0;
    int ndays = 0, chunk_days = 5, nchunks = 0;
    int nb = 1, nt = 1;
    std::vector<int> devs;
    std::vector<Block> blocks;
    std::vector<double> surface;               // the whole raster's snow surface of the current chunk (nb > 1)
    mcf_grid_inputs grid{};
    mcf_options opt{};
    mcf_snowdriver_in snow{};
    std::vector<int32_t> snowday, nosnowday;   // [ndays]
    bool pass1_done = false;
    bool af = false;                           // array weather: `.snowmodel2` + `.runmicrosnow2` (mcf_runmicrosnow2)
    int64_t keep_reserve = (int64_t)8 << 30;
    // A run is one simulated period on fresh plans: the device memory a kept chunk needs would have to be ALLOCATED for it (5 GB per
    // chunk of a 1024 x 1024 raster: 0.1 s, measured 3 s of a 6 s call) where re-running the chunk in pass 2 takes 3.5 ms — chunks are
    // kept only on request (MCF_SNOWRUN_KEEP=1); the stepwise API's pool across years (tools/bench_snow.py) is where keeping pays
    bool keep = false;
    // what pass 2 did not have to do (mcf_snowrun_stats)
    std::atomic<int64_t> st_tile_days{0}, st_tile_days_left_out{0}, st_chunks_kept{0}, st_chunks_rerun{0};
    ~mcf_snowrun() {
        for (Block& k : blocks) {
            if (k.plan || k.sp) (void)hipSetDevice(k.device);
            if (k.plan) mcf_plan_destroy(k.plan);
            if (k.sp) mcf_snowplan_destroy(k.sp);
        }
    }
};

namespace {

// runs fn(t, guarded, fail_here) on every worker thread; a phase body run under `guarded` can neither let an exception leave
// its thread nor skip a barrier
template <class F>
int run_workers(mcf_snowrun* h, F&& fn) {
    const int nt = h->nt;
    std::vector<int> rcs((size_t)nt, MCF_OK);
    std::vector<std::string> errs((size_t)nt);
    std::atomic<bool> failed{false};
    PhaseBarrier bar(nt);
    auto worker = [&](int t) {
        auto fail_here = [&, t](int rc) {
            if (rcs[(size_t)t] == MCF_OK) { rcs[(size_t)t] = rc; errs[(size_t)t] = mcf_last_error(); }
            failed = true;
        };
        auto guarded = [&, t](auto&& body) {
            if (failed) return;
            try { body(); }
            catch (const std::exception& e) {
                if (rcs[(size_t)t] == MCF_OK) { rcs[(size_t)t] = MCF_ERR_NOMEM; errs[(size_t)t] = std::string("snow run: ") + e.what(); }
                failed = true;
            }
        };
        fn(t, bar, failed, guarded, fail_here);
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < nt; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto& th : threads) th.join();
    for (int t = 0; t < nt; ++t)
        if (rcs[(size_t)t] != MCF_OK) return mcf::api_fail(rcs[(size_t)t], errs[(size_t)t]);
    return MCF_OK;
}

// snowdaysfun, src/microclimfCpp.cpp:5531-5550: a snow day has snow somewhere in some hour (max > 0), a no-snow day a
// snow-free cell in some hour (min == 0); NaN compares false both ways
void snowdays_of(const double* mx, const double* mn, int nd, int32_t* snow, int32_t* nosnow) {
    for (int d = 0; d < nd; ++d) {
        int s = 0, n = 0;
        for (int hh = 0; hh < 24; ++hh) {
            s |= mx[d * 24 + hh] > 0.0;
            n |= mn[d * 24 + hh] == 0.0;
        }
        snow[d] = s; nosnow[d] = n;
    }
}

// one chunk of the snow model over every block: phases separated by the workers' barrier.  `with_apply3`: pass 1 also takes
// the per-step extremes of totalSWE.  Collective: every worker calls it for the same chunk.
// smod (pass 1, optional): the caller's whole-series snow arrays — a block's chunk goes straight into its rows.
template <class G, class FH>
void snow_chunk(mcf_snowrun* h, int t, int ch, PhaseBarrier& bar, std::atomic<bool>& failed, G& guarded, FH& fail_here,
                bool with_apply3, const mcf_snowdriver_out* smod, double* smean, double* tmean) {
    const int nb = h->nb, nt = h->nt;
    const int64_t R = h->R, C = h->C;
    auto block_out = [&](const Block& k) {
        mcf_snowdriver_out bo{};
        if (smod) {
            bo = *smod;
            double** const bop[5] = {&bo.Tc, &bo.Tg, &bo.groundsnowdepth, &bo.totalSWE, &bo.snowden};
            for (double** q : bop) if (*q) *q += k.r0;
        }
        return bo;
    };
    if (nb == 1) {
        // one block: the sequence of mcf_snowmodel1 (no host copy of the surface; the raster-wide mean only where .tpicalc
        // falls back on it — mcf_snowplan_prepare_chunk ignores it otherwise)
        if (t == 0) guarded([&] {
            Block& k = h->blocks[0];
            double s = 0, n = 1, ts = 0, tn = 1;
            int rc = mcf_snowplan_surface_partial(k.sp, &s, &n);
            if (!rc) rc = mcf_snowplan_prepare_chunk(k.sp, ch, nullptr, 0, 0, s / n, &ts, &tn);
            if (!rc) { const mcf_snowdriver_out bo = block_out(k); rc = mcf_snowplan_run_chunk_pitched(k.sp, ch, ts / tn, &bo, R); }
            if (rc) fail_here(rc);
        });
    } else {
        guarded([&] {                                         // ---- phase 1: the surface
            for (int b = t; b < nb && !failed; b += nt) {
                Block& k = h->blocks[(size_t)b];
                k.ext.resize((size_t)(k.nr * C));
                int rc = mcf_snowplan_surface(k.sp, k.ext.data());
                if (!rc) rc = mcf_snowplan_surface_partial(k.sp, &k.s, &k.n);
                if (rc) { fail_here(rc); break; }
                for (int64_t c = 0; c < C; ++c) memcpy(&h->surface[(size_t)(k.r0 + R * c)], &k.ext[(size_t)(k.nr * c)], (size_t)k.nr * 8);
            }
        });
        bar.wait();
        if (t == 0 && !failed) {
            double s = 0, n = 0;
            for (const Block& k : h->blocks) { s += k.s; n += k.n; }
            *smean = s / n;
        }
        bar.wait();
        guarded([&] {                                         // ---- phase 2: halos, terrain, tpi
            for (int b = t; b < nb && !failed; b += nt) {
                Block& k = h->blocks[(size_t)b];
                // what prepare_chunk asks for at most (the terrain stencil's reach, whole af x af blocks of the tpi), or every row up
                // to the raster edge
                int32_t af = 1;
                int rc = mcf_snowplan_chunk_af(k.sp, ch, &af);
                if (rc) { fail_here(rc); break; }
                const int64_t ss = h->snow.res <= 100 ? 10 : 1;
                const int64_t want = 100 + 3 * ss + 2 * (int64_t)af;
                const int64_t hn = std::min(want, k.r0), hs = std::min(want, R - k.r0 - k.nr), RB = hn + k.nr + hs;
                gather_rows(k.ext, h->surface.data(), R, C, k.r0 - hn, RB);
                rc = mcf_snowplan_prepare_chunk(k.sp, ch, (hn || hs) ? k.ext.data() : nullptr, (int32_t)hn, (int32_t)hs, *smean, &k.ts, &k.tn);
                if (rc) { fail_here(rc); break; }
            }
        });
        bar.wait();
        if (t == 0 && !failed) {
            double s = 0, n = 0;
            for (const Block& k : h->blocks) { s += k.ts; n += k.tn; }
            *tmean = s / n;
        }
        bar.wait();
        guarded([&] {                                         // ---- phase 3: the chunk
            for (int b = t; b < nb && !failed; b += nt) {
                Block& k = h->blocks[(size_t)b];
                const mcf_snowdriver_out bo = block_out(k);
                const int rc = mcf_snowplan_run_chunk_pitched(k.sp, ch, *tmean, &bo, R);
                if (rc) { fail_here(rc); break; }
            }
        });
    }
    if (with_apply3) guarded([&] {
        const int ns = h->chunk_days * 24;
        for (int b = t; b < nb && !failed; b += nt) {
            Block& k = h->blocks[(size_t)b];
            k.mx.assign((size_t)ns, 0.0); k.cmx.assign((size_t)ns, 0.0); k.mn.assign((size_t)ns, 0.0); k.cmn.assign((size_t)ns, 0.0);
            int rc = mcf_snowplan_apply3(k.sp, ch, MCF_APPLY_MAX, k.mx.data(), k.cmx.data());
            if (!rc) rc = mcf_snowplan_apply3(k.sp, ch, MCF_APPLY_MIN, k.mn.data(), k.cmn.data());
            if (rc) { fail_here(rc); break; }
        }
    });
    bar.wait();
}

int check_create(const mcf_microsnow_in* in, const mcf_options* opt, const mcf_multi* mu) {
    if (!in || !opt || !in->grid || !in->snow) return mcf::api_fail(MCF_ERR_ARG, "null snow-run argument");
    const mcf_grid_inputs& g = *in->grid;
    const mcf_snow_inputs& sb = in->snow->base;
    // Array weather (round 5): `.snowmodel2`'s loop + `.runmicrosnow2` (R/internal.R:2950-3008, 3661-3745) — the solver's and the snow
    // model's weather as arrays at the raster's resolution, what runmicro2Cpp / gridmodelsnow2 / gridmicrosnow2 take.  Both sides in
    // the same geometry; one block (a chunk's slices are uploaded from the caller's whole-raster arrays as the loop reaches them).
    if ((g.array_forcing != 0) != (sb.array_forcing != 0))
        return mcf::api_fail(MCF_ERR_ARG, "snow run: the solver's and the snow model's weather differ in geometry (data.frame / array)");
    if (g.array_forcing == 2)
        return mcf::api_fail(MCF_ERR_ARG, "snow run: coarse array forcing is not supported here (resample to the raster first, as `.snowmodel2` does)");
    if (g.array_forcing && mu && (mu->n_blocks > 1 || mu->n_devices > 1))
        return mcf::api_fail(MCF_ERR_ARG, "snow run, array weather: one block on one device");
    // Time-varying vegetation (round 5).  `.runmicronosnow` sends a layered `vegp` to `.runmodel3Cpp` on the no-snow-day SUBSET
    // (R/internal.R:3333-3342), which deals the subset's days to layers by `.sortvegp(vegp, "C", n, subs)` — the layer a step has
    // in the WHOLE series (round(seq(0.50001, dmx + 0.5, length.out = n))[subs], the day's mode, R/internal.R:252-270) — and
    // renumbers the layers it uses (:1391-1399).  A no-snow day therefore runs with the layer the whole-series table gives it:
    // the solver plan takes the caller's whole-series layer table (lyr_st / lyr_ed in whole-series steps, as mcf_runmicro3) and
    // the days run at their own place in it.  The snow model's and gridmicrosnow1's rasters are `.sortl` / `.sortl2` means —
    // single-layer — either way.
    if (g.veg_layers > 1 && (!g.lyr_st || !g.lyr_ed))
        return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow1: layered vegetation needs lyr_st / lyr_ed (whole-series steps)");
    if (opt->reqhgt < 0)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow1: reqhgt < 0 needs the whole series at once (Tbelowgroundv); use mcf_runmicro1 + "
                                          "mcf_gridmicrosnow1 on host arrays");
    if (g.rows <= 0 || g.cols <= 0 || g.tsteps < 24) return mcf::api_fail(MCF_ERR_ARG, "bad dimensions");
    if (sb.rows != g.rows || sb.cols != g.cols || sb.tsteps != g.tsteps)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow1: the solver's and the snow model's inputs differ in shape");
    if (g.row_pitch > 0 && g.row_pitch != g.rows) return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow1 takes dense rasters");
    if (!in->snow->dtm || !sb.vegp.pai || !sb.vegp.hgt || !sb.vegp.leaft || !sb.vegp.clump || !sb.other.isnowdc || !sb.other.isnowdg ||
        !sb.other.isnowac || !sb.other.isnowag)
        return mcf::api_fail(MCF_ERR_ARG, "null input: a snow-model raster");
    if (!g.clim.tc) return mcf::api_fail(MCF_ERR_ARG, "null input: climdata$temp");
    if (in->snow->chunk_steps != 0 && in->snow->chunk_steps % 24)
        return mcf::api_fail(MCF_ERR_ARG, "snow driver: chunk_steps must be whole days");
    return MCF_OK;
}

}  // namespace

extern "C" int mcf_snowrun_create(const mcf_microsnow_in* in, const mcf_options* opt, const mcf_multi* mu, mcf_snowrun** out) {
    try {
        if (!out) return mcf::api_fail(MCF_ERR_ARG, "null snow-run argument");
        int rc = check_create(in, opt, mu);
        if (rc) return rc;
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
            return mcf::api_fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
        mcf_snowrun* h = new mcf_snowrun();
        struct Guard { mcf_snowrun* p; ~Guard() { delete p; } } guard{h};
        if (!mu) {
            if (opt->device < 0 || opt->device >= nd) return mcf::api_fail(MCF_ERR_ARG, "device ordinal out of range");
            h->devs.push_back(opt->device);
        } else if (mu->n_devices <= 0) {
            for (int d = 0; d < nd; ++d) h->devs.push_back(d);
        } else {
            if (!mu->devices) return mcf::api_fail(MCF_ERR_ARG, "n_devices > 0 with a null device list");
            for (int i = 0; i < mu->n_devices; ++i) {
                if (mu->devices[i] < 0 || mu->devices[i] >= nd) return mcf::api_fail(MCF_ERR_ARG, "device ordinal out of range");
                h->devs.push_back(mu->devices[i]);
            }
        }
        h->grid = *in->grid; h->opt = *opt; h->snow = *in->snow;
        h->af = h->grid.array_forcing != 0;
        const int64_t R = h->R = h->grid.rows, C = h->C = h->grid.cols;
        h->T = h->grid.tsteps;
        h->ndays = (int)(h->T / 24);
        const int chunk = h->snow.chunk_steps > 0 ? h->snow.chunk_steps : 120;
        h->chunk_days = chunk / 24;
        h->nchunks = std::max(1, (int)(h->T / chunk));          // `for (day in 1:n5days)`, R/internal.R:2553-2565
        h->nb = (int)std::min<int64_t>(mu && mu->n_blocks > 0 ? mu->n_blocks : (int)h->devs.size(), R);
        if (h->af) h->nb = 1;
        h->nt = (int)std::min<size_t>(h->devs.size(), (size_t)h->nb);
        h->blocks.resize((size_t)h->nb);
        if (h->nb > 1) h->surface.assign((size_t)(R * C), 0.0);
        h->snowday.assign((size_t)std::max(h->ndays, h->nchunks * h->chunk_days), 0);
        h->nosnowday.assign(h->snowday.size(), 0);
        if (const char* e = getenv("MCF_SNOW_KEEP_RESERVE_GB")) h->keep_reserve = (int64_t)(atof(e) * 1073741824.0);
        h->keep = getenv("MCF_SNOWRUN_KEEP") != nullptr;
        const mcf_snow_inputs& base = h->snow.base;
        rc = run_workers(h, [&](int t, PhaseBarrier& bar, std::atomic<bool>& failed, auto& guarded, auto& fail_here) {
            guarded([&] {
                for (int b = t; b < h->nb && !failed; b += h->nt) {
                    Block& k = h->blocks[(size_t)b];
                    k.device = h->devs[(size_t)t];
                    k.r0 = R * b / h->nb; k.nr = R * (b + 1) / h->nb - k.r0;
                    // ---- the block's snow plan
                    mcf_snowdriver_in bi = h->snow;
                    if (h->nb > 1) {
                        gather_rows(k.pai, base.vegp.pai, R, C, k.r0, k.nr); gather_rows(k.hgt, base.vegp.hgt, R, C, k.r0, k.nr);
                        gather_rows(k.leaft, base.vegp.leaft, R, C, k.r0, k.nr); gather_rows(k.clump, base.vegp.clump, R, C, k.r0, k.nr);
                        gather_rows(k.dc, base.other.isnowdc, R, C, k.r0, k.nr); gather_rows(k.dg, base.other.isnowdg, R, C, k.r0, k.nr);
                        gather_rows(k.ac, base.other.isnowac, R, C, k.r0, k.nr); gather_rows(k.ag, base.other.isnowag, R, C, k.r0, k.nr);
                        gather_rows(k.dtm, h->snow.dtm, R, C, k.r0, k.nr);
                        bi.base.rows = k.nr;
                        bi.base.vegp.pai = k.pai.data(); bi.base.vegp.hgt = k.hgt.data(); bi.base.vegp.leaft = k.leaft.data();
                        bi.base.vegp.clump = k.clump.data();
                        bi.base.other.isnowdc = k.dc.data(); bi.base.other.isnowdg = k.dg.data();
                        bi.base.other.isnowac = k.ac.data(); bi.base.other.isnowag = k.ag.data();
                        bi.dtm = k.dtm.data();
                    }
                    bi.base.other.slope = bi.base.other.aspect = bi.base.other.skyview = bi.base.other.wsa = bi.base.other.hor = nullptr;
                    int rc2 = mcf_snowplan_create(&bi, k.r0, R, k.device, &k.sp);
                    if (rc2) { fail_here(rc2); break; }
                    // ---- ... and its solver plan: the caller's arrays read in place through the row pitch
                    mcf_grid_inputs sub = h->grid;
                    sub.rows = k.nr;
                    sub.row_pitch = R;
                    auto off = [&](const double*& q) { if (q) q += k.r0; };
                    off(sub.vegp.hgt); off(sub.vegp.pai); off(sub.vegp.x); off(sub.vegp.gsmax); off(sub.vegp.leafr); off(sub.vegp.leaft);
                    off(sub.vegp.clump); off(sub.vegp.leafd); off(sub.vegp.paia); off(sub.vegp.leafden);
                    off(sub.soilc.Smin); off(sub.soilc.Smax); off(sub.soilc.gref); off(sub.soilc.soilb); off(sub.soilc.Psie);
                    off(sub.soilc.Vq); off(sub.soilc.Vm); off(sub.soilc.Mc); off(sub.soilc.rho); off(sub.soilc.slope);
                    off(sub.soilc.aspect); off(sub.soilc.twi); off(sub.soilc.svfa); off(sub.soilc.wsa); off(sub.soilc.hor);
                    if (h->af) {                 // (layered / lat-lon arrays of the array-forcing solver, read through the same pitch)
                        off(sub.clim.tc); off(sub.clim.es); off(sub.clim.ea); off(sub.clim.tdew); off(sub.clim.pk); off(sub.clim.swdown);
                        off(sub.clim.difrad); off(sub.clim.lwdown); off(sub.clim.windspeed);
                        off(sub.pointm.soilm); off(sub.pointm.Tg); off(sub.pointm.Tbp); off(sub.pointm.G); off(sub.pointm.umu);
                        off(sub.pointm.kp); off(sub.pointm.muGp); off(sub.pointm.dtrp);
                        off(sub.lats); off(sub.lons);
                    }
                    k.gsub = sub;
                    mcf_options o = h->opt;
                    o.device = k.device;
                    rc2 = mcf_plan_create(&sub, &o, h->chunk_days, 2, &k.plan);
                    if (rc2) { fail_here(rc2); break; }
                    if (h->nb > 1 && (rc2 = mcf_plan_twi_partial(k.plan, &k.twi_s, &k.twi_n))) { fail_here(rc2); break; }
                    k.kept.assign((size_t)h->nchunks, 0);
                }
            });
            bar.wait();
            // the solver's one global reduction (src/microclimfCpp.cpp:993-1004): partial sums in block order
            if (h->nb > 1) guarded([&] {
                double s = 0; int64_t n = 0;
                for (const Block& k : h->blocks) { s += k.twi_s; n += k.twi_n; }
                for (int b = t; b < h->nb && !failed; b += h->nt) {
                    const int rc2 = mcf_plan_set_twi_mean(h->blocks[(size_t)b].plan, s / (double)n);
                    if (rc2) { fail_here(rc2); break; }
                }
            });
        });
        if (rc) return rc;
        guard.p = nullptr;
        *out = h;
        return MCF_OK;
    } catch (const std::exception& e) {
        return mcf::api_fail(MCF_ERR_NOMEM, std::string("mcf_snowrun_create: ") + e.what());
    }
}

extern "C" void mcf_snowrun_destroy(mcf_snowrun* h) { delete h; }

// Pass 1's snow chunks stay in device memory for pass 2, up to `bytes` in all (shared out among the row blocks by their rows);
// 0 switches keeping off.  The sets are allocated as pass 1 first needs them and POOLED in the handle's snow plans: a handle that
// runs a second period (mcf_snowrun_pass1 again) allocates nothing — where allocating 10 GB costs more than re-running the chunk
// (the struct's comment), keeping pays from the second period on, or on the first when the caller wants `smod` anyway.
extern "C" int mcf_snowrun_keep(mcf_snowrun* h, int64_t bytes) {
    if (!h) return mcf::api_fail(MCF_ERR_ARG, "null snow run");
    if (bytes < 0) return mcf::api_fail(MCF_ERR_ARG, "mcf_snowrun_keep: bytes >= 0");
    h->keep = bytes > 0;
    for (Block& k : h->blocks) {
        const int rc = mcf_snowplan_set_keep_budget(k.sp, bytes > 0 ? (int64_t)((double)bytes * (double)k.nr / (double)h->R) : 0);
        if (rc) return rc;
    }
    return MCF_OK;
}

extern "C" int32_t mcf_snowrun_days(const mcf_snowrun* h) { return h ? h->ndays : 0; }
extern "C" int mcf_snowrun_stats(const mcf_snowrun* h, int64_t stats[4]) {
    if (!h || !stats) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    stats[0] = h->st_tile_days; stats[1] = h->st_tile_days_left_out; stats[2] = h->st_chunks_kept; stats[3] = h->st_chunks_rerun;
    return MCF_OK;
}

extern "C" int mcf_snowrun_pass1(mcf_snowrun* h, const mcf_snowdriver_out* smod, int32_t* snowday, int32_t* nosnowday) {
    if (!h) return mcf::api_fail(MCF_ERR_ARG, "null snow run");
    try {
        const int cd = h->chunk_days, ns = cd * 24;
        std::fill(h->snowday.begin(), h->snowday.end(), 0);
        std::fill(h->nosnowday.begin(), h->nosnowday.end(), 0);
        h->pass1_done = false;
        double smean = 0, tmean = 0;
        const int rc = run_workers(h, [&](int t, PhaseBarrier& bar, std::atomic<bool>& failed, auto& guarded, auto& fail_here) {
            guarded([&] {
                for (int b = t; b < h->nb && !failed; b += h->nt) {
                    Block& k = h->blocks[(size_t)b];
                    int rc2 = mcf_snowplan_reset(k.sp);
                    if (!rc2) rc2 = mcf_snowplan_release_kept(k.sp);
                    if (rc2) { fail_here(rc2); break; }
                    std::fill(k.kept.begin(), k.kept.end(), 0);
                }
            });
            bar.wait();
            for (int ch = 0; ch < h->nchunks; ++ch) {
                guarded([&] {
                    for (int b = t; b < h->nb && !failed; b += h->nt) {
                        Block& k = h->blocks[(size_t)b];
                        int rc2 = mcf_snowplan_checkpoint(k.sp, ch);      // pass 2 starts any chunk from here
                        // a chunk that could not stay in HBM is re-run by pass 2 if it holds a snow day: unless the caller wants the snow
                        // series, pass 1 writes only what it reads itself of such a chunk (totalSWE, density: mcf_snowplan_set_series)
                        int32_t room = 0;
                        if (!rc2 && h->keep) rc2 = mcf_snowplan_can_keep(k.sp, h->keep_reserve, &room);
                        if (!rc2) rc2 = mcf_snowplan_set_series(k.sp, (room || smod) ? 31u : (4u | 16u));
                        if (rc2) { fail_here(rc2); break; }
                    }
                });
                snow_chunk(h, t, ch, bar, failed, guarded, fail_here, true, smod, &smean, &tmean);
                // (under `guarded` like every phase body: an exception here — the vectors allocate — must neither leave the thread
                // while the others are joinable nor skip the barrier below)
                if (t == 0) guarded([&] {
                    // extremes over the blocks (max / min skip blocks whose step held no value), then the chunk's day classes
                    std::vector<double> mx((size_t)ns, -INFINITY), mn((size_t)ns, INFINITY);
                    for (const Block& k : h->blocks)
                        for (int q = 0; q < ns; ++q) {
                            if (k.cmx[(size_t)q] > 0 && k.mx[(size_t)q] > mx[(size_t)q]) mx[(size_t)q] = k.mx[(size_t)q];
                            if (k.cmn[(size_t)q] > 0 && k.mn[(size_t)q] < mn[(size_t)q]) mn[(size_t)q] = k.mn[(size_t)q];
                        }
                    snowdays_of(mx.data(), mn.data(), cd, &h->snowday[(size_t)(ch * cd)], &h->nosnowday[(size_t)(ch * cd)]);
                });
                bar.wait();
                guarded([&] {
                    bool any = false;
                    for (int d = 0; d < cd; ++d) any |= h->snowday[(size_t)(ch * cd + d)] != 0;
                    for (int b = t; b < h->nb && !failed; b += h->nt) {
                        Block& k = h->blocks[(size_t)b];
                        int rc2 = mcf_snowplan_meand_accumulate(k.sp, ch, &h->snowday[(size_t)(ch * cd)]);
                        int32_t kept = 0;
                        if (!rc2 && any && h->keep) rc2 = mcf_snowplan_keep_chunk(k.sp, ch, h->keep_reserve, &kept);
                        if (rc2) { fail_here(rc2); break; }
                        k.kept[(size_t)ch] = (char)kept;
                    }
                });
                bar.wait();
            }
        });
        if (rc) return rc;
        // steps past the last whole chunk: the snow model leaves them NA (R/internal.R:2554-2558), `.runmicrosnow1` turns NA
        // into 0 (:3586) — days without snow anywhere, solved by the grid solver
        for (int d = h->nchunks * cd; d < h->ndays; ++d) { h->snowday[(size_t)d] = 0; h->nosnowday[(size_t)d] = 1; }
        h->pass1_done = true;
        if (snowday) memcpy(snowday, h->snowday.data(), (size_t)h->ndays * 4);
        if (nosnowday) memcpy(nosnowday, h->nosnowday.data(), (size_t)h->ndays * 4);
        return MCF_OK;
    } catch (const std::exception& e) {
        return mcf::api_fail(MCF_ERR_NOMEM, std::string("mcf_snowrun_pass1: ") + e.what());
    }
}

extern "C" int mcf_snowrun_pass2(mcf_snowrun* h, const mcf_snow_inputs* micro, double mat, mcf_outputs* out) {
    if (!h || !out) return mcf::api_fail(MCF_ERR_ARG, "null snow-run argument");
    if (!h->pass1_done) return mcf::api_fail(MCF_ERR_STATE, "snow run: mcf_snowrun_pass1 first");
    try {
        const int cd = h->chunk_days, ndays = h->ndays;
        const int64_t R = h->R, C = h->C, T = h->T, HS = R * C;
        for (int v = 0; v < MCF_NOUT; ++v)
            if (h->opt.out[v] && !out->var[v]) return mcf::api_fail(MCF_ERR_ARG, "null output array for a requested variable");
        // ---- day lists
        std::vector<int> sdays, ndays_;
        for (int d = 0; d < ndays; ++d) {
            if (h->snowday[(size_t)d]) sdays.push_back(d);
            if (h->nosnowday[(size_t)d]) ndays_.push_back(d);
        }
        std::vector<int32_t> sub_of_day(h->snowday.size(), -1);
        for (size_t i = 0; i < sdays.size(); ++i) sub_of_day[(size_t)sdays[i]] = (int32_t)i;
        // gridmicrosnow1's `out` (R/internal.R:3616-3622)
        int32_t outm[MCF_NOUT];
        for (int v = 0; v < MCF_NOUT; ++v) outm[v] = h->opt.out[v] ? 1 : 0;
        if (h->opt.reqhgt == 0.0) {
            static const int32_t ground[MCF_NOUT] = {1, 0, 0, 1, 0, 1, 1, 1, 1, 1};
            memcpy(outm, ground, sizeof outm);
        }
        // ---- the snow-day subset of the whole-series inputs (subsetpointmodel(micropoint, days = snowdays), R/internal.R:3599)
        const int64_t TS = (int64_t)sdays.size() * 24;
        std::vector<int32_t> yr, mo, dy;
        std::vector<double> hr, ser[10];
        mcf_snow_inputs sub{};
        if (!sdays.empty()) {
            if (!micro) return mcf::api_fail(MCF_ERR_ARG, "snow run: the year has snow days, gridmicrosnow1's inputs are needed");
            if (micro->rows != R || micro->cols != C || micro->tsteps != T || (micro->array_forcing != 0) != h->af)
                return mcf::api_fail(MCF_ERR_ARG, "snow run: gridmicrosnow's inputs must be the whole series on the whole raster, in the run's weather geometry");
            const mcf_snow_climate& cl = micro->clim;
            const double* src[10] = {cl.temp, cl.relhum, cl.pres, cl.swdown, cl.difrad, cl.lwdown, cl.windspeed, cl.winddir, cl.precip, cl.umu};
            static const char* nm[10] = {"temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir", "precip", "umu"};
            for (int f = 0; f < 10; ++f)
                if (!src[f]) return mcf::api_fail(MCF_ERR_ARG, std::string("null input: gridmicrosnow1 weather$") + nm[f]);
            const mcf_obstime& ob = micro->obstime;
            if (!ob.year || !ob.month || !ob.day || !ob.hour) return mcf::api_fail(MCF_ERR_ARG, "null obstime");
            const mcf_snow_vegp& vg = micro->vegp;
            const mcf_snow_other& ot = micro->other;
            if (!vg.pai || !vg.hgt || !vg.leaft || !vg.clump || !vg.paia || !vg.leafd || !vg.leafden || !ot.slope || !ot.aspect ||
                !ot.skyview || !ot.wsa || !ot.hor)
                return mcf::api_fail(MCF_ERR_ARG, "null input: a gridmicrosnow1 raster");
            if (outm[MCF_OUT_SOILM] && h->opt.out[MCF_OUT_SOILM] && !ot.Smax) return mcf::api_fail(MCF_ERR_ARG, "soilm requested but other$Smax is null");
            sub = *micro;
            if (!h->af) {      // (array weather: the snow plan takes the whole series and the day map — nine arrays are not copied)
            yr.resize((size_t)TS); mo.resize((size_t)TS); dy.resize((size_t)TS); hr.resize((size_t)TS);
            for (auto& s : ser) s.resize((size_t)TS);
            for (size_t i = 0; i < sdays.size(); ++i)
                for (int hh = 0; hh < 24; ++hh) {
                    const int64_t a = (int64_t)sdays[i] * 24 + hh, q = (int64_t)i * 24 + hh;
                    yr[(size_t)q] = ob.year[a]; mo[(size_t)q] = ob.month[a]; dy[(size_t)q] = ob.day[a]; hr[(size_t)q] = ob.hour[a];
                    for (int f = 0; f < 10; ++f) ser[f][(size_t)q] = src[f][a];
                }
            sub.tsteps = TS;
            sub.obstime.year = yr.data(); sub.obstime.month = mo.data(); sub.obstime.day = dy.data(); sub.obstime.hour = hr.data();
            sub.clim.temp = ser[0].data(); sub.clim.relhum = ser[1].data(); sub.clim.pres = ser[2].data(); sub.clim.swdown = ser[3].data();
            sub.clim.difrad = ser[4].data(); sub.clim.lwdown = ser[5].data(); sub.clim.windspeed = ser[6].data();
            sub.clim.winddir = ser[7].data(); sub.clim.precip = ser[8].data(); sub.clim.umu = ser[9].data();
            }
        }
        // the solver's maximum air temperature over the NO-snow subset (src/microclimfCpp.cpp:2159-2168 on what `.runmicronosnow`
        // hands it, R/internal.R:3605)
        double mxtc = -INFINITY;
        if (!h->af)
            for (int d : ndays_)
                for (int hh = 0; hh < 24; ++hh) { const double v = h->grid.clim.tc[(int64_t)d * 24 + hh]; if (v > mxtc) mxtc = v; }
        const double NA = na_real_host();
        double smean = 0, tmean = 0;
        const int rc = run_workers(h, [&](int t, PhaseBarrier& bar, std::atomic<bool>& failed, auto& guarded, auto& fail_here) {
            guarded([&] {
                for (int b = t; b < h->nb && !failed; b += h->nt) {
                    Block& k = h->blocks[(size_t)b];
                    int rc2 = MCF_OK;
                    if (!sdays.empty()) {
                        mcf_snow_inputs bs = sub;
                        if (h->nb > 1) {
                            const mcf_snow_vegp& vg = micro->vegp;
                            const mcf_snow_other& ot = micro->other;
                            gather_rows(k.m_pai, vg.pai, R, C, k.r0, k.nr); gather_rows(k.m_hgt, vg.hgt, R, C, k.r0, k.nr);
                            gather_rows(k.m_leaft, vg.leaft, R, C, k.r0, k.nr); gather_rows(k.m_clump, vg.clump, R, C, k.r0, k.nr);
                            gather_rows(k.m_paia, vg.paia, R, C, k.r0, k.nr); gather_rows(k.m_leafd, vg.leafd, R, C, k.r0, k.nr);
                            gather_rows(k.m_leafden, vg.leafden, R, C, k.r0, k.nr);
                            gather_rows(k.m_slope, ot.slope, R, C, k.r0, k.nr); gather_rows(k.m_aspect, ot.aspect, R, C, k.r0, k.nr);
                            gather_rows(k.m_svf, ot.skyview, R, C, k.r0, k.nr);
                            gather_rows(k.m_wsa, ot.wsa, R, C, k.r0, k.nr, 8); gather_rows(k.m_hor, ot.hor, R, C, k.r0, k.nr, 24);
                            if (ot.Smax) gather_rows(k.m_smax, ot.Smax, R, C, k.r0, k.nr);
                            bs.rows = k.nr;
                            bs.vegp.pai = k.m_pai.data(); bs.vegp.hgt = k.m_hgt.data(); bs.vegp.leaft = k.m_leaft.data();
                            bs.vegp.clump = k.m_clump.data(); bs.vegp.paia = k.m_paia.data(); bs.vegp.leafd = k.m_leafd.data();
                            bs.vegp.leafden = k.m_leafden.data();
                            bs.other.slope = k.m_slope.data(); bs.other.aspect = k.m_aspect.data(); bs.other.skyview = k.m_svf.data();
                            bs.other.wsa = k.m_wsa.data(); bs.other.hor = k.m_hor.data();
                            bs.other.Smax = ot.Smax ? k.m_smax.data() : nullptr;
                        }
                        rc2 = mcf_snowplan_micro_setup(k.sp, &bs, sub_of_day.data(), (int32_t)sub_of_day.size(), h->opt.reqhgt, mat, outm, 0);
                    }
                    if (!rc2 && !ndays_.empty())         // (array weather: per cell, cpp:2467-2471, over the no-snow days)
                        rc2 = h->af ? mcf_plan_set_mxtc_days(k.plan, &k.gsub, h->nosnowday.data(), ndays) : mcf_plan_set_mxtc(k.plan, mxtc);
                    if (!rc2) rc2 = mcf_snowplan_set_series(k.sp, 31u);         // (pass 2's re-runs feed the snow microclimate)
                    if (rc2) { fail_here(rc2); break; }
                }
            });
            bar.wait();
            // the merged days of a ring slot to the caller: the block's rows in place, through the row pitch
            auto fetch_days = [&](Block& k, int slot, int d0, int nd) -> int {
                for (int v = 0; v < MCF_NOUT; ++v) {
                    if (!h->opt.out[v]) continue;
                    double* dst = out->var[v] + k.r0 + HS * (int64_t)d0 * 24;
                    int rc2 = mcf_plan_fetch_pitched(k.plan, slot, v, 0, (int64_t)nd * 24, dst, R);
                    if (rc2) return rc2;
                    // a day in NEITHER class (a melted pack's negative rounding residue: max <= 0 and min != 0) is no day of either
                    // model; the reference's merge indexes past its arrays there (R/internal.R:3650-3655) — NA here
                    for (int d = 0; d < nd; ++d)
                        if (!h->snowday[(size_t)(d0 + d)] && !h->nosnowday[(size_t)(d0 + d)])
                            for (int64_t lc = (int64_t)(d0 + d) * 24 * C; lc < (int64_t)(d0 + d + 1) * 24 * C; ++lc)
                                for (int64_t r = 0; r < k.nr; ++r) out->var[v][k.r0 + r + R * lc] = NA;
                }
                return MCF_OK;
            };
            // runs of consecutive no-snow days.  Where such a run lies in a chunk with snow, the tiles whose cells are all under snow
            // for the whole run are left out (include/mcf.h mcf_plan_run_days_masked): gridmicrosnow1 overwrites every one of their values
            const bool no_skip = getenv("MCF_SNOW_NO_TILE_SKIP") != nullptr;
            const bool no_cells = getenv("MCF_SNOW_NO_CELL_GATHER") != nullptr;      // (A/B: tiles as the unit, as in round 4)
            std::vector<uint8_t> skip;
            auto solver_days = [&](Block& k, int slot, int ch, int d0, int nd, bool has_snow) -> int {
                int q = 0;
                if (h->af) {      // array weather: the chunk's forcing into the slot once (all its days: the runs address them by day)
                    bool any = false;
                    for (int d = 0; d < nd; ++d) any |= h->nosnowday[(size_t)(d0 + d)] != 0;
                    if (!any) return MCF_OK;
                    const int rc3 = mcf_plan_upload_forcing_days(k.plan, &k.gsub, d0, nd, slot);
                    if (rc3) return rc3;
                }
                while (q < nd) {
                    if (!h->nosnowday[(size_t)(d0 + q)]) { ++q; continue; }
                    // (a run ends where the days' class changes: on a day without snow anywhere every cell is the solver's)
                    const bool both = h->snowday[(size_t)(d0 + q)] != 0;
                    int e = q;
                    while (e < nd && h->nosnowday[(size_t)(d0 + e)] && (h->snowday[(size_t)(d0 + e)] != 0) == both) ++e;
                    int rc2;
                    if (has_snow && both && !no_skip && ch >= 0 && !h->af) {
                        mcf_ring_layout lay;
                        if ((rc2 = mcf_plan_ring_layout(k.plan, &lay))) return rc2;
                        const int64_t nt = (lay.cells + lay.cells_per_tile - 1) / lay.cells_per_tile;
                        const uint8_t* need = nullptr;
                        int64_t n_need = lay.cells;
                        if (!no_cells && (rc2 = mcf_snowplan_free_cells(k.sp, k.plan, ch, q, e - q, &need, &n_need))) return rc2;
                        if (!no_cells && 16 * n_need <= lay.cells) {
                            // the few cells that are not under snow throughout, gathered into tiles of their own (scattered cells
                            // pay below ~ 8 % of the raster, profiles/r05_cells_rate.txt: their values reach the ring 8 bytes at a time)
                            rc2 = mcf_plan_run_days_cells(k.plan, d0 + q, e - q, slot, q, need, lay.cells, nullptr);
                            h->st_tile_days += nt * (e - q);
                            h->st_tile_days_left_out += (nt - (n_need + lay.cells_per_tile - 1) / lay.cells_per_tile) * (e - q);
                        } else {
                            int64_t ncov = 0;
                            skip.resize((size_t)nt);
                            if ((rc2 = mcf_snowplan_covered_tiles(k.sp, k.plan, ch, q, e - q, skip.data(), nt, &ncov))) return rc2;
                            rc2 = mcf_plan_run_days_masked(k.plan, d0 + q, e - q, slot, q, ncov ? skip.data() : nullptr, ncov ? nt : 0);
                            h->st_tile_days += nt * (e - q);
                            h->st_tile_days_left_out += ncov * (e - q);
                        }
                    } else {
                        rc2 = mcf_plan_run_days_at(k.plan, d0 + q, e - q, slot, q);
                    }
                    if (rc2) return rc2;
                    q = e;
                }
                return MCF_OK;
            };
            int slot = 0;
            for (int ch = 0; ch < h->nchunks; ++ch, slot ^= 1) {
                const int d0 = ch * cd, nd = std::min(cd, ndays - d0);
                bool has_snow = false, kept_all = true;
                for (int d = 0; d < cd; ++d) has_snow |= h->snowday[(size_t)(d0 + d)] != 0;
                for (const Block& k : h->blocks) kept_all = kept_all && k.kept[(size_t)ch];
                if (t == 0 && has_snow) ++(kept_all ? h->st_chunks_kept : h->st_chunks_rerun);
                if (has_snow && !kept_all) {          // collective: the blocks' surfaces couple through their halos
                    guarded([&] {
                        for (int b = t; b < h->nb && !failed; b += h->nt) {
                            const int rc2 = mcf_snowplan_restore(h->blocks[(size_t)b].sp, ch);
                            if (rc2) { fail_here(rc2); break; }
                        }
                    });
                    bar.wait();
                    snow_chunk(h, t, ch, bar, failed, guarded, fail_here, false, nullptr, &smean, &tmean);
                }
                guarded([&] {
                    for (int b = t; b < h->nb && !failed; b += h->nt) {
                        Block& k = h->blocks[(size_t)b];
                        int rc2 = solver_days(k, slot, ch, d0, nd, has_snow);
                        if (!rc2 && has_snow) rc2 = mcf_snowplan_microsnow(k.sp, k.plan, ch, slot, &h->nosnowday[(size_t)d0]);
                        if (!rc2 && nd > 0) rc2 = fetch_days(k, slot, d0, nd);
                        if (rc2) { fail_here(rc2); break; }
                    }
                });
                bar.wait();
            }
            // days past the last whole chunk: the solver's alone
            guarded([&] {
                for (int b = t; b < h->nb && !failed; b += h->nt) {
                    Block& k = h->blocks[(size_t)b];
                    for (int d0 = h->nchunks * cd; d0 < ndays && !failed; d0 += cd) {
                        const int nd = std::min(cd, ndays - d0);
                        int rc2 = solver_days(k, 0, -1, d0, nd, false);
                        if (!rc2) rc2 = fetch_days(k, 0, d0, nd);
                        if (rc2) { fail_here(rc2); break; }
                    }
                    // steps past the last whole day stay NA (src/microclimfCpp.cpp:2116)
                    for (int v = 0; v < MCF_NOUT; ++v)
                        if (h->opt.out[v])
                            for (int64_t lc = (int64_t)ndays * 24 * C; lc < T * C; ++lc)
                                for (int64_t r = 0; r < k.nr; ++r) out->var[v][k.r0 + r + R * lc] = NA;
                }
            });
        });
        return rc;
    } catch (const std::exception& e) {
        return mcf::api_fail(MCF_ERR_NOMEM, std::string("mcf_snowrun_pass2: ") + e.what());
    }
}

static int runmicrosnow1_impl(const mcf_microsnow_in* in, const mcf_options* opt, const mcf_multi* mu, mcf_outputs* out,
                              const mcf_snowdriver_out* smod) {
    mcf_snowrun* h = nullptr;
    int rc = mcf_snowrun_create(in, opt, mu, &h);
    if (rc) return rc;
    struct Guard { mcf_snowrun* p; ~Guard() { mcf_snowrun_destroy(p); } } guard{h};
    if ((rc = mcf_snowrun_pass1(h, smod, nullptr, nullptr))) return rc;
    return mcf_snowrun_pass2(h, in->micro, in->mat, out);
}
extern "C" int mcf_runmicrosnow1(const mcf_microsnow_in* in, const mcf_options* opt, mcf_outputs* out, const mcf_snowdriver_out* smod) {
    if (in && in->grid && in->grid->array_forcing) return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow1 takes data.frame (vector) weather; array weather: mcf_runmicrosnow2");
    return runmicrosnow1_impl(in, opt, nullptr, out, smod);
}
// `.snowmodel2`'s loop + `.runmicrosnow2` (R/internal.R:2950-3008, 3661-3745): the same run with array weather
extern "C" int mcf_runmicrosnow2(const mcf_microsnow_in* in, const mcf_options* opt, mcf_outputs* out, const mcf_snowdriver_out* smod) {
    if (in && in->grid && !in->grid->array_forcing) return mcf::api_fail(MCF_ERR_ARG, "mcf_runmicrosnow2 takes array weather; data.frame weather: mcf_runmicrosnow1");
    return runmicrosnow1_impl(in, opt, nullptr, out, smod);
}
extern "C" int mcf_runmicrosnow1_multi(const mcf_microsnow_in* in, const mcf_options* opt, const mcf_multi* multi, mcf_outputs* out,
                                       const mcf_snowdriver_out* smod) {
    if (!multi) return mcf::api_fail(MCF_ERR_ARG, "null argument");
    return runmicrosnow1_impl(in, opt, multi, out, smod);
}
