// mcf_api.hip — the C ABI of include/mcf.h: plan management, day-chunk streaming
// through HBM, and the one-shot host-to-host entry points that stand in for the
// reference's _microclimf_runmicro1Cpp / _microclimf_runmicro2Cpp
// (src/RcppExports.cpp:250-297).
//
// No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <future>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcf.h"
#include "mcf_kernels.h"
#include "mcf_hostpipe.hpp"
#include "mcf_ncfile.hpp"
#include "mcf_nc4file.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                     __FILE__, __LINE__);                                                    \
            return fail(e_ == hipErrorOutOfMemory ? MCF_ERR_NOMEM : MCF_ERR_HIP, b_);        \
        }                                                                                    \
    } while (0)

const uint64_t kNaBits = 0x7FF00000000007A2ULL;
double na_real_host() {
    double d;
    memcpy(&d, &kNaBits, 8);
    return d;
}

}  // namespace

namespace mcf {
int api_fail(int code, const std::string& msg) { return fail(code, msg); }
}

struct mcf_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t rows = 0, cols = 0, N = 0, tsteps = 0;
    int64_t pitch = 0;      // rows of the (taller) column-major raster the host arrays are blocks of; = rows: dense
    int ndays = 0;
    bool af = false, bg = false;
    bool coarse = false;                 // array_forcing == 2: coarse arrays interpolated in the solver
    bool coarse_lds = false;             // ... with the taps staged in LDS: every 32-cell tile touches <= 4 coarse rows per column
    int crows = 0, ccols = 0;
    const double *d_crowpos = nullptr, *d_ccolpos = nullptr;
    int altcorrect = 0;
    const double *d_elevd = nullptr, *d_pkfac = nullptr;
    int cpb = 16;
    int layers = 1;
    int32_t* d_daylayer = nullptr;
    mcf_options opt{};
    mcf::Globals g{};
    int hiy = 8760;
    double lat = 0, lon = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    // static inputs
    const double* d_veg[10] = {};
    const double* d_soil[13] = {};
    const double *d_wsa = nullptr, *d_hor = nullptr, *d_lats = nullptr, *d_lons = nullptr;
    double* d_cellc = nullptr;
    double* d_tt = nullptr;
    double* d_twi2 = nullptr;
    double twi_sum = 0;
    int64_t twi_count = 0;
    double twi_mean = 0;
    bool cells_ready = false;
    // fast-clamp dispatch (vector forcing, reqhgt >= 0): tiles whose valid cells are all FL_REGULAR run the min / max
    // variant of the solver, the others the reference's compare-and-select form; days with an irregular forcing step
    // send the whole launch through the latter (mcf_device.hpp `cap`, mcf_kernels.hip k_solve / k_solve_fix)
    bool fast_enabled = false;
    std::vector<char> day_irregular, day_soil_daily;
    int32_t *d_tiles_fast = nullptr, *d_tiles_slow = nullptr;
    // mcf_plan_run_days_masked: the tile classes on the host, and the launch's own lists on the device
    std::vector<int32_t> h_tiles_fast, h_tiles_slow, h_tiles_sub;
    int32_t* d_tiles_sub = nullptr;
    int64_t tiles_sub_cap = 0, masked_tiles_skipped = 0;
    // mcf_plan_run_days_cells: the cells' classes, the per-256-cell counts / list offsets, the list of gathered cells, the
    // gathered tiles' images and their ring (grown as needed)
    uint8_t* d_cls = nullptr;
    uint8_t* d_tile_regular = nullptr;      // [ntiles] 1: a tile of the fast list (null: no fast list)
    int32_t *d_blockcnt = nullptr, *d_celllist = nullptr;
    double *d_subimg = nullptr, *d_subring = nullptr;
    int64_t cls_cap = 0, blockcnt_cap = 0, celllist_cap = 0, subimg_cap = 0, subring_cap = 0;      // bytes
    std::vector<int32_t> h_blockcnt, h_cells_slow;
    int32_t* d_cells_slow = nullptr;
    int64_t cells_slow_cap = 0;
    hipStream_t prep = nullptr;              // the list is made beside the plan's stream, which is not drained for it
    hipEvent_t ev_prep = nullptr, ev_cells_done = nullptr;
    bool cells_inflight = false;
    int64_t cells_runs = 0, cells_gathered = 0;
    int64_t n_fast = 0, n_slow = 0, tiles_cap = 0;
    int32_t *d_fix_count = nullptr, *d_fix_list = nullptr;
    int fix_cap = 8192;
    int64_t fast_launches = 0, slow_launches = 0;
    // array forcing
    double* d_dt = nullptr;
    int32_t* d_windex = nullptr;
    double* d_mxtc = nullptr;
    double* d_force = nullptr;  // array forcing: tiled ring (see mcf_plan_create); coarse: [15][crows*ccols][T]
    double* d_force_stage = nullptr;
    int64_t force_tile_stride = 0, force_day_stride = 0, force_slot_elems = 0;
    std::vector<int> force_day0, force_ndays;
    // outputs
    int ring_days = 0, ring_slots = 0;
    // reqhgt >= 0: TILED, [slots][tile][day][var][ring_block_doubles(cpb)] in the solver's lane order (mcf_kernels.h
    // RingView); reqhgt < 0: linear, [slots][nvars][N*ring_days*24]
    double* d_ring = nullptr;
    bool tiled = false;
    int64_t ntiles = 0, slot_elems = 0;                        // elements per slot (all variables)
    int64_t ring_tile_stride = 0, ring_day_stride = 0, ring_var_stride = 0;
    double* d_stage = nullptr;   // untiled [N][steps] pieces on their way to the host (mcf_plan_fetch)
    int64_t stage_elems = 0;
    int var_slot[MCF_NOUT];     // index among enabled vars or -1
    int nvars = 0;
    // reqhgt < 0
    double *d_tgser = nullptr, *d_ddsum = nullptr, *d_scratch = nullptr;
    const double *d_Tgp = nullptr, *d_Tbp = nullptr;
    // timing
    bool ktiming = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
    double ktotal_ms = 0;
    int64_t klaunches = 0;
    int64_t valid_cells = 0;
    // packed sink staging
    int32_t* d_pack = nullptr;
    int64_t pack_elems = 0;
    // device -> pageable host copies of large results (mcf_hostpipe.hpp)
    mcf::HostPipe* pipe = nullptr;
    hipEvent_t ev_pipe = nullptr;
    bool pipe_failed = false;
};

namespace {

// lazily sets up the pinned ring + host copy threads of a plan; false: unavailable (callers fall back to hipMemcpy)
bool ensure_pipe(mcf_plan* p) {
    if (p->pipe) return true;
    if (p->pipe_failed) return false;
    p->pipe = new mcf::HostPipe();
    if (!p->pipe->init() || hipEventCreateWithFlags(&p->ev_pipe, hipEventDisableTiming) != hipSuccess) {
        delete p->pipe;
        p->pipe = nullptr;
        p->pipe_failed = true;
        return false;
    }
    return true;
}

int dalloc(mcf_plan* p, void** ptr, int64_t nbytes) {
    if (nbytes <= 0) nbytes = 8;
    hipError_t e = hipMalloc(ptr, (size_t)nbytes);
    if (e != hipSuccess) {
        char b[256];
        snprintf(b, sizeof b, "hipMalloc(%lld bytes) failed: %s", (long long)nbytes, hipGetErrorString(e));
        return fail(MCF_ERR_NOMEM, b);
    }
    p->allocs.push_back(*ptr);
    p->bytes += nbytes;
    return MCF_OK;
}

// a buffer that grows: the old one is released first.  A buffer that has to grow gets a quarter more than is asked for — the
// sets of cells of successive calls differ by little, and a release + allocation of a gigabyte costs 50 ms
int dregrow(mcf_plan* p, void** ptr, int64_t* cap, int64_t nbytes) {
    if (*ptr && *cap >= nbytes) return MCF_OK;
    if (*ptr) nbytes += nbytes / 4;
    if (*ptr) {
        for (size_t i = 0; i < p->allocs.size(); ++i)
            if (p->allocs[i] == *ptr) { p->allocs.erase(p->allocs.begin() + (long)i); break; }
        (void)hipFree(*ptr);
        p->bytes -= *cap;
        *ptr = nullptr;
        *cap = 0;
    }
    const int rc = dalloc(p, ptr, nbytes);
    if (!rc) *cap = std::max<int64_t>(nbytes, 8);
    return rc;
}

// variable `var` (requested) of ring slot `slot` as its consumers address it
mcf::RingView ring_view(const mcf_plan* p, int slot, int var) {
    mcf::RingView v{};
    v.N = p->N;
    const double* sb = p->d_ring + (int64_t)slot * p->slot_elems;
    if (p->tiled) {
        v.base = sb + (int64_t)p->var_slot[var] * p->ring_var_stride;
        v.tile_stride = p->ring_tile_stride; v.day_stride = p->ring_day_stride; v.cpb = p->cpb;
    } else {
        v.base = sb + (int64_t)p->var_slot[var] * (p->N * (int64_t)p->ring_days * 24);
        v.cpb = 0;
    }
    return v;
}

// host [rows x ncols] with leading dimension p->pitch -> dense device memory (and the other way): plain copies for dense hosts
hipError_t copy_in(mcf_plan* p, void* dev, const double* host, int64_t ncols) {
    if (p->pitch == p->rows) return hipMemcpyAsync(dev, host, (size_t)(p->rows * ncols) * 8, hipMemcpyHostToDevice, p->stream);
    return hipMemcpy2DAsync(dev, (size_t)p->rows * 8, host, (size_t)p->pitch * 8, (size_t)p->rows * 8, (size_t)ncols,
                            hipMemcpyHostToDevice, p->stream);
}
// `spatial`: n = rows x (cols x layers) values of a raster array (read with the plan's row pitch); else a plain vector
int upload(mcf_plan* p, const double* host, int64_t n, const double** dev, const char* name, bool spatial = true) {
    if (!host) return fail(MCF_ERR_ARG, std::string("missing input array: ") + name);
    void* d = nullptr;
    int rc = dalloc(p, &d, n * 8);
    if (rc) return rc;
    if (spatial && p->pitch != p->rows) HIP_TRY(copy_in(p, d, host, n / p->rows));
    else HIP_TRY(hipMemcpyAsync(d, host, (size_t)n * 8, hipMemcpyHostToDevice, p->stream));
    *dev = (const double*)d;
    return MCF_OK;
}

int check_inputs(const mcf_grid_inputs* in, const mcf_options* opt) {
    if (!in || !opt) return fail(MCF_ERR_ARG, "null inputs/options");
    if (in->rows <= 0 || in->cols <= 0) return fail(MCF_ERR_ARG, "rows/cols must be positive");
    if (in->tsteps < 0) return fail(MCF_ERR_ARG, "negative tsteps");
    if (in->tsteps > 0 && (!in->obstime.year || !in->obstime.month || !in->obstime.day || !in->obstime.hour))
        return fail(MCF_ERR_ARG, "obstime columns missing");
    if (in->array_forcing && (!in->lats || !in->lons)) return fail(MCF_ERR_ARG, "lats/lons missing");
    if (in->veg_layers > 1) {
        if (!in->lyr_st || !in->lyr_ed) return fail(MCF_ERR_ARG, "dfsel (lyr_st / lyr_ed) missing");
        for (int l = 0; l < in->veg_layers; ++l) {
            int span = in->lyr_ed[l] - in->lyr_st[l] + 1;
            if (span < 24)      // the reference's Rcpp::stop, src/microclimfCpp.cpp:2636-2637
                return fail(MCF_ERR_ARG, "Too many layers in vegp. Max layers must be <= max days");
            if (in->lyr_st[l] < 0 || in->lyr_st[l] % 24 != 0 || in->lyr_st[l] + (span / 24) * 24 > in->tsteps)
                return fail(MCF_ERR_ARG, "dfsel: layer ranges must start on whole days inside the series");
        }
    }
    if (!(opt->cells_per_block == 0 || opt->cells_per_block == 16 || opt->cells_per_block == 21 ||
          opt->cells_per_block == 32 || opt->cells_per_block == 42))
        return fail(MCF_ERR_ARG, "cells_per_block must be 0, 16, 21, 32 or 42");
    return MCF_OK;
}

int ensure_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MCF_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return MCF_OK;
}

const double* const* clim_ptrs(const mcf_grid_inputs* in, const double* out[15]) {
    // TF_TC .. TF_DTRP order
    out[0] = in->clim.tc; out[1] = in->clim.es; out[2] = in->clim.ea; out[3] = in->clim.tdew;
    out[4] = in->clim.pk; out[5] = in->clim.swdown; out[6] = in->clim.difrad; out[7] = in->clim.lwdown;
    out[8] = in->clim.windspeed; out[9] = in->pointm.soilm; out[10] = in->pointm.G;
    out[11] = in->pointm.umu; out[12] = in->pointm.kp; out[13] = in->pointm.muGp; out[14] = in->pointm.dtrp;
    return out;
}
const char* kRawNames[15] = {"tc", "es", "ea", "tdew", "pk", "swdown", "difrad", "lwdown", "windspeed",
                             "pointm$soilm", "pointm$G", "pointm$umu", "pointm$kp", "pointm$muGp",
                             "pointm$dtrp"};

int ensure_cells(mcf_plan* p) {
    if (p->cells_ready) return MCF_OK;
  for (int l = 0; l < p->layers; ++l) {
    mcf::CellSetupArgs a{};
    a.N = p->N;
    const int64_t lo = (int64_t)l * p->N;      // layer l of the [rows,cols,layers] vegetation arrays
    a.hgt = p->d_veg[0] + lo; a.pai = p->d_veg[1] + lo; a.x = p->d_veg[2] + lo; a.gsmax = p->d_veg[3] + lo;
    a.leafr = p->d_veg[4] + lo; a.leaft = p->d_veg[5] + lo; a.clump = p->d_veg[6] + lo; a.leafd = p->d_veg[7] + lo;
    a.paia = p->d_veg[8] + lo; a.leafden = p->d_veg[9] + lo;
    a.hgt0 = p->d_veg[0];
    a.Smin = p->d_soil[0]; a.Smax = p->d_soil[1]; a.gref = p->d_soil[2]; a.soilb = p->d_soil[3];
    a.Psie = p->d_soil[4]; a.Vq = p->d_soil[5]; a.Vm = p->d_soil[6]; a.Mc = p->d_soil[7];
    a.rho = p->d_soil[8]; a.slope = p->d_soil[9]; a.aspect = p->d_soil[10]; a.twi = p->d_soil[11];
    a.svfa = p->d_soil[12];
    a.hor = p->d_hor; a.wsa = p->d_wsa;
    a.lats = p->d_lats; a.lons = p->d_lons; a.lat = p->lat; a.lon = p->lon;
    a.crowpos = p->d_crowpos; a.ccolpos = p->d_ccolpos; a.rows = p->rows;
    a.elevd = p->d_elevd; a.pkfac = p->d_pkfac;
    a.tfact = p->opt.tfact;
    a.twi_mean = p->twi_mean;
    a.g = p->g;
    a.cpb = p->cpb;
    a.cellc = p->d_cellc + (int64_t)l * p->ntiles * mcf::tile_image_doubles(p->cpb);
    mcf::launch_cell_setup(a, p->stream);
    HIP_TRY(hipGetLastError());
  }
    if (!p->af && p->day_irregular.empty() && p->ndays > 0) {
        // the per-day flags of the time table, once: kStepIrregular and kSoilDaily in TF_IDX (its last field)
        const int tfc = mcf::time_field_count();
        std::vector<double> idx((size_t)p->ndays * 24);
        HIP_TRY(hipMemcpy2DAsync(idx.data(), 24 * 8, p->d_tt + (int64_t)(tfc - 1) * 24, (size_t)tfc * 24 * 8, 24 * 8,
                                 (size_t)p->ndays, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        static const bool no_share = getenv("MCF_NO_SOIL_SHARE") != nullptr;     // A/B runs
        p->day_irregular.assign((size_t)p->ndays, 0);
        p->day_soil_daily.assign((size_t)p->ndays, 0);
        for (int d = 0; d < p->ndays; ++d) {
            for (int h = 0; h < 24; ++h)
                if ((int)idx[(size_t)d * 24 + h] & mcf::step_irregular_bit()) p->day_irregular[(size_t)d] = 1;
            p->day_soil_daily[(size_t)d] = (!no_share && ((int)idx[(size_t)d * 24] & mcf::soil_daily_bit())) ? 1 : 0;
        }
    }
    if (p->fast_enabled) {
        int rc;
        const int64_t ntiles = (p->N + p->cpb - 1) / p->cpb;
        if (p->tiles_cap < ntiles) {
            void* q;
            if ((rc = dalloc(p, &q, ntiles * 4))) return rc;
            p->d_tiles_fast = (int32_t*)q;
            if ((rc = dalloc(p, &q, ntiles * 4))) return rc;
            p->d_tiles_slow = (int32_t*)q;
            p->tiles_cap = ntiles;
        }
        if (!p->d_fix_count) {
            void* q;
            if ((rc = dalloc(p, &q, 64))) return rc;
            p->d_fix_count = (int32_t*)q;
            HIP_TRY(hipMemsetAsync(p->d_fix_count, 0, 64, p->stream));
            if ((rc = dalloc(p, &q, (int64_t)p->fix_cap * 8))) return rc;
            p->d_fix_list = (int32_t*)q;
        }
        // tile classes (the flags stay on the device for mcf_plan_run_days_cells)
        if (!p->d_tile_regular) {
            void* q;
            if ((rc = dalloc(p, &q, ntiles))) return rc;
            p->d_tile_regular = (uint8_t*)q;
        }
        uint8_t* d_flag = p->d_tile_regular;
        mcf::launch_tile_regular(p->d_cellc, p->N, p->layers, p->cpb, d_flag, p->stream);
        HIP_TRY(hipGetLastError());
        std::vector<uint8_t> flag((size_t)ntiles);
        HIP_TRY(hipMemcpyAsync(flag.data(), d_flag, (size_t)ntiles, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        std::vector<int32_t> fastl, slowl;
        fastl.reserve((size_t)ntiles);
        for (int64_t t = 0; t < ntiles; ++t) (flag[(size_t)t] ? fastl : slowl).push_back((int32_t)t);
        p->n_fast = (int64_t)fastl.size();
        p->n_slow = (int64_t)slowl.size();
        p->h_tiles_fast = fastl; p->h_tiles_slow = slowl;
        if (p->n_slow > 0) {     // with no irregular tile the fast launch needs no list (identity)
            HIP_TRY(hipMemcpy(p->d_tiles_fast, fastl.data(), fastl.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_tiles_slow, slowl.data(), slowl.size() * 4, hipMemcpyHostToDevice));
        }
    }
    p->cells_ready = true;     // only now: a failed allocation or copy above leaves the plan to try again, not half set up
    return MCF_OK;
}

}  // namespace

// ---- writetonc sink (R/dataprep.R:1063-1260) -------------------------------------------------------------------
struct mcf_ncfile {
    std::unique_ptr<mcf::NcFile> f;
    int var_of[MCF_NOUT];        // file variable index of solver output v, or -1
    int out_of[MCF_NOUT];        // solver output of file variable k
    double scale[MCF_NOUT];      // per file variable
    int fill_only[MCF_NOUT];
    // staging of records on their way to the file; plain arrays, not vectors: no zero-fill of memory about to be overwritten
    std::unique_ptr<uint8_t[]> stage[2];
    size_t stage_bytes[2] = {0, 0};
    uint8_t* staging(int i, size_t n) {
        if (stage_bytes[i] < n) { stage[i].reset(new uint8_t[n]); stage_bytes[i] = n; }
        return stage[i].get();
    }
};

namespace {

std::string r_number(double x) {   // as.character(<double>): 15 significant digits
    char b[64];
    snprintf(b, sizeof b, "%.15g", x);
    return b;
}

// variables writetonc defines for a height, with its long names and units (dataprep.R:1097-1157, 1180-1214, 1234-1244)
bool nc_var_def(int v, double reqhgt, mcf::NcVarDef* d, double* scale, bool* put_by_reference) {
    static const char* names[MCF_NOUT] = {"Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown",
                                          "Rlwdown", "Rswup", "Rlwup"};
    static const char* radlong[5] = {"Downward direct shortwave radiation", "Downward diffuse shortwave radiation",
                                     "Downward longwave radiation", "Upward shortwave radiation", "Upward longwave radiation"};
    d->name = names[v];
    *put_by_reference = true;
    const std::string h = r_number(fabs(reqhgt));
    if (v >= 5) {
        if (reqhgt < 0) return false;
        d->long_name = radlong[v - 5]; d->units = "W/m^2"; *scale = 1;
        *put_by_reference = false;   // `if ("raddir" %in% vars)` never holds for the documented names (dataprep.R:1163-1167)
        return true;
    }
    switch (v) {
    case 0:
        d->long_name = reqhgt > 0 ? "Air temperature at height " + h + " m"
                     : reqhgt == 0 ? std::string("Soil surface temperature") : "Soil temperature at depth " + h + " m";
        d->units = "deg C x 100"; *scale = 100; return true;
    case 1:
        if (!(reqhgt > 0)) return false;
        d->long_name = "Leaf temperature at height " + h + " m"; d->units = "deg C x 100"; *scale = 100; return true;
    case 2:
        if (!(reqhgt > 0)) return false;
        d->long_name = "Relative humidity at height " + h + " m"; d->units = "Percentage"; *scale = 1; return true;
    case 3:
        d->long_name = "Soil surface moisture";
        d->units = reqhgt < 0 ? "Percentage volume" : "Volume percentage soil moisture in top 10 cm of soil";
        *scale = 100;
        *put_by_reference = false;   // `ncvar_put(nccew, …)`: an undefined object, the put is an R error (dataprep.R:1161)
        return true;
    default:
        if (!(reqhgt > 0)) return false;
        d->long_name = "Wind speed at height " + h + " m"; d->units = "m/s x 100"; *scale = 100; return true;
    }
}

inline int32_t nc_pack(double x, double scale) {
    const double q = rint(x * scale);
    return (q > -2147483648.0 && q < 2147483648.0) ? (int32_t)q : mcf::NcFile::kMissval;
}

}  // namespace

extern "C" {

int mcf_abi_version(void) { return MCF_ABI_VERSION; }
const char* mcf_last_error(void) { return g_err.c_str(); }
int mcf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mcf_plan_destroy(mcf_plan* p) {
    if (p) mcf::print_variant_stats();
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    if (p->prep) { hipStreamSynchronize(p->prep); hipStreamDestroy(p->prep); }
    if (p->ev_prep) hipEventDestroy(p->ev_prep);
    if (p->ev_cells_done) hipEventDestroy(p->ev_cells_done);
    for (auto& e : p->kev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (void* a : p->allocs) hipFree(a);
    delete p->pipe;
    if (p->ev_pipe) hipEventDestroy(p->ev_pipe);
    if (p->ev0) hipEventDestroy(p->ev0);
    if (p->ev1) hipEventDestroy(p->ev1);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}

int mcf_plan_create(const mcf_grid_inputs* in, const mcf_options* opt, int32_t ring_days, int32_t ring_slots,
                    mcf_plan** out) {
    if (!out) return fail(MCF_ERR_ARG, "null plan pointer");
    *out = nullptr;
    int rc = check_inputs(in, opt);
    if (rc) return rc;
    rc = ensure_device(opt->device);
    if (rc) return rc;
    mcf_plan* p = new mcf_plan();
    struct Guard { mcf_plan* p; ~Guard() { if (p) mcf_plan_destroy(p); } } guard{p};
    p->device = opt->device;
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&p->ev0));
    HIP_TRY(hipEventCreate(&p->ev1));
    p->rows = in->rows; p->cols = in->cols; p->N = in->rows * in->cols; p->tsteps = in->tsteps;
    p->pitch = in->row_pitch > 0 ? in->row_pitch : in->rows;
    if (p->pitch < p->rows) return fail(MCF_ERR_ARG, "row_pitch smaller than rows");
    p->ndays = (int)(in->tsteps / 24);                       // cpp:2116 truncation
    p->af = in->array_forcing != 0;
    p->coarse = in->array_forcing == 2;
    p->bg = opt->reqhgt < 0.0;
    if (p->coarse) {
        if (in->coarse_rows < 1 || in->coarse_cols < 1 || !in->coarse_rowpos || !in->coarse_colpos || !in->coarse_relhum ||
            !in->coarse_winddir)
            return fail(MCF_ERR_ARG, "coarse array forcing needs coarse_rows/cols, coarse_rowpos/colpos, coarse_relhum and coarse_winddir");
        // (the solver addresses a day of a coarse field with 32-bit byte offsets: 24 x cells x 8 B < 2^32)
        if ((int64_t)in->coarse_rows * in->coarse_cols > 11000000) return fail(MCF_ERR_ARG, "coarse grid too large (more than 1.1e7 cells)");
        if (p->bg && !opt->complete) return fail(MCF_ERR_ARG, "coarse array forcing: reqhgt < 0 needs complete = 1");
        for (int64_t i = 0; i < in->rows; ++i)
            if (!(in->coarse_rowpos[i] >= 0.0 && in->coarse_rowpos[i] <= in->coarse_rows - 1))
                return fail(MCF_ERR_ARG, "coarse_rowpos must lie in [0, coarse_rows - 1]");
        for (int64_t j = 0; j < in->cols; ++j)
            if (!(in->coarse_colpos[j] >= 0.0 && in->coarse_colpos[j] <= in->coarse_cols - 1))
                return fail(MCF_ERR_ARG, "coarse_colpos must lie in [0, coarse_cols - 1]");
        p->crows = in->coarse_rows; p->ccols = in->coarse_cols;
        p->altcorrect = in->coarse_altcorrect;
        // the LDS-staged taps (k_solve, CLDS): a tile's 32 cells lie in at most two raster columns and, in each, reach at most
        // four coarse rows (the rows of its first and last cell and one more)
        p->coarse_lds = in->rows >= 32 && getenv("MCF_NO_COARSE_LDS") == nullptr;
        // ... with row positions that do not decrease down a raster column: the kernel finds a tile's wrap into the next column
        // where the position falls back, and the window test below looks at a window's two ends only.  A flipped or otherwise
        // non-monotone coarse grid takes the per-lane taps, which make no such assumption.
        for (int64_t i = 1; p->coarse_lds && i < in->rows; ++i)
            if (in->coarse_rowpos[i] < in->coarse_rowpos[i - 1]) p->coarse_lds = false;
        for (int64_t i = 0; p->coarse_lds && i < in->rows; ++i) {
            const int64_t l = std::min<int64_t>(i + 31, in->rows - 1);
            if (floor(in->coarse_rowpos[l]) - floor(in->coarse_rowpos[i]) + 2 > 4) p->coarse_lds = false;
        }
        if (p->altcorrect < 0 || p->altcorrect > 2) return fail(MCF_ERR_ARG, "coarse_altcorrect must be 0, 1 or 2");
        if (p->altcorrect && (!in->coarse_dtm || !in->fine_dtm))
            return fail(MCF_ERR_ARG, "altitude correction needs coarse_dtm and fine_dtm");
    }
    // vector forcing: two 8-wave workgroups per CU (21 cells); array forcing: one 12-wave workgroup
    p->cpb = opt->cells_per_block ? opt->cells_per_block : (in->array_forcing ? 32 : 21);
    // coarse array forcing is built for 32-cell tiles only: tile classes, tile lists and the ring's blocks must agree
    if (p->coarse) p->cpb = 32;
    if (in->array_forcing && p->cpb == 42) p->cpb = 32;      // 42-cell tiles are built for vector forcing only (they would spill)
    if (p->N >= ((int64_t)1 << 31)) return fail(MCF_ERR_ARG, "at most 2^31 - 1 cells per plan (row-tile larger rasters)");
    {
        static const bool no_fast = getenv("MCF_NO_FAST_CLAMPS") != nullptr;     // A/B runs
        p->fast_enabled = !no_fast && !(opt->reqhgt < 0.0);
    }
    p->layers = in->veg_layers > 1 ? in->veg_layers : 1;
    p->opt = *opt;
    p->lat = in->lat; p->lon = in->lon;
    const int64_t N = p->N, T = p->tsteps;
    if (ring_slots < 1) ring_slots = 1;
    if (ring_days < 1) ring_days = 1;
    if (ring_days > std::max(p->ndays, 1)) ring_days = std::max(p->ndays, 1);
    // reqhgt < 0 smooths the whole series (incl. steps past the last whole day, which read as 0)
    if (p->bg) { ring_slots = 1; ring_days = (int)std::max<int64_t>((in->tsteps + 23) / 24, 1); }
    // array forcing re-lays a slot's series with one launch row per step (gridDim.y <= 65535): 2730 whole days at most
    if (in->array_forcing == 1 && ring_days > 2730) ring_days = 2730;
    p->ring_days = ring_days; p->ring_slots = ring_slots;

    // ---- static rasters
    const double* veg[10] = {in->vegp.hgt, in->vegp.pai, in->vegp.x, in->vegp.gsmax, in->vegp.leafr,
                             in->vegp.leaft, in->vegp.clump, in->vegp.leafd, in->vegp.paia, in->vegp.leafden};
    const char* vegn[10] = {"hgt", "pai", "x", "gsmax", "leafr", "leaft", "clump", "leafd", "paia", "leafden"};
    for (int i = 0; i < 10; ++i)
        if ((rc = upload(p, veg[i], N * p->layers, &p->d_veg[i], vegn[i]))) return rc;
    const double* soil[13] = {in->soilc.Smin, in->soilc.Smax, in->soilc.gref, in->soilc.soilb, in->soilc.Psie,
                              in->soilc.Vq, in->soilc.Vm, in->soilc.Mc, in->soilc.rho, in->soilc.slope,
                              in->soilc.aspect, in->soilc.twi, in->soilc.svfa};
    const char* soiln[13] = {"Smin", "Smax", "gref", "soilb", "Psie", "Vq", "Vm", "Mc", "rho", "slope",
                             "aspect", "twi", "svfa"};
    for (int i = 0; i < 13; ++i)
        if ((rc = upload(p, soil[i], N, &p->d_soil[i], soiln[i]))) return rc;
    if ((rc = upload(p, in->soilc.wsa, N * 8, &p->d_wsa, "wsa"))) return rc;
    if ((rc = upload(p, in->soilc.hor, N * 24, &p->d_hor, "hor"))) return rc;
    if (p->af) {
        if ((rc = upload(p, in->lats, N, &p->d_lats, "lats"))) return rc;
        if ((rc = upload(p, in->lons, N, &p->d_lons, "lons"))) return rc;
    }
    std::vector<double> cpk_sea;     // coarse pressure reduced to sea level (altitude correction)
    if (p->coarse) {
        if ((rc = upload(p, in->coarse_rowpos, in->rows, &p->d_crowpos, "coarse_rowpos", false))) return rc;
        if ((rc = upload(p, in->coarse_colpos, in->cols, &p->d_ccolpos, "coarse_colpos", false))) return rc;
        if (p->altcorrect) {
            // R/internal.R:1236-1241: psl = pk / ((293 - 0.0065 zc) / 293)^5.26 on the coarse grid, back up with the
            // fine elevation after resampling; elevd = resample(dtmc) - dtm
            const int cr = p->crows, cc = p->ccols;
            std::vector<double> zc((size_t)cr * cc), elevd((size_t)N), pkfac((size_t)N);
            for (size_t q = 0; q < zc.size(); ++q) zc[q] = std::isnan(in->coarse_dtm[q]) ? 0.0 : in->coarse_dtm[q];
            for (int64_t j = 0; j < in->cols; ++j) {
                const double cp = in->coarse_colpos[j], fc = floor(cp), wx = cp - fc;
                const int c0 = (int)fc, c1 = c0 + 1 < cc ? c0 + 1 : c0;
                for (int64_t i = 0; i < in->rows; ++i) {
                    const double rp = in->coarse_rowpos[i], fr = floor(rp), wy = rp - fr;
                    const int r0 = (int)fr, r1 = r0 + 1 < cr ? r0 + 1 : r0;
                    const double top = (1.0 - wx) * zc[(size_t)(r0 + cr * c0)] + wx * zc[(size_t)(r0 + cr * c1)];
                    const double bot = (1.0 - wx) * zc[(size_t)(r1 + cr * c0)] + wx * zc[(size_t)(r1 + cr * c1)];
                    const double z = in->fine_dtm[i + p->pitch * j];
                    elevd[(size_t)(i + in->rows * j)] = ((1.0 - wy) * top + wy * bot) - z;
                    pkfac[(size_t)(i + in->rows * j)] = pow((293.0 - 0.0065 * z) / 293.0, 5.26);
                }
            }
            if ((rc = upload(p, elevd.data(), N, &p->d_elevd, "elevd", false))) return rc;
            if ((rc = upload(p, pkfac.data(), N, &p->d_pkfac, "pkfac", false))) return rc;
            HIP_TRY(hipStreamSynchronize(p->stream));
            if (in->tsteps > 0 && in->clim.pk) {
                const int64_t cN = (int64_t)cr * cc;
                cpk_sea.resize((size_t)(cN * in->tsteps));
                for (int64_t k = 0; k < in->tsteps; ++k)
                    for (int64_t q = 0; q < cN; ++q)
                        cpk_sea[(size_t)(q + cN * k)] = in->clim.pk[q + cN * k] / pow((293.0 - 0.0065 * zc[(size_t)q]) / 293.0, 5.26);
            }
        }
    }
    int64_t nvalid = 0;
    for (int64_t j = 0; j < in->cols; ++j)
        for (int64_t i = 0; i < in->rows; ++i) nvalid += !std::isnan(in->vegp.hgt[i + p->pitch * j]);
    p->valid_cells = nvalid;

    // ---- solver constants
    p->g.reqhgt = opt->reqhgt;
    p->g.reqhgt2 = opt->reqhgt < 0.00001 ? 0.00001 : opt->reqhgt;      // cpp:2246-2247
    p->g.zref = opt->zref;
    p->g.hf0p = mcf::hf_pow02(1 / 999.99);   // mincondCpp(leafabs, 999.99, ...) at cpp:1348
    p->g.hf500p = mcf::hf_pow02(500.0);      // rs capped at 500 (gs <= 0.002), cpp:1321-1323
    p->g.shadowmask = p->af ? 0 : 1;
    p->g.dTmx = 0.0;
    p->hiy = 365 * 24;
    if (T > 0 && in->obstime.year[0] % 4 == 0) p->hiy = 366 * 24;      // cpp:2171-2172

    // ---- the one global reduction: mean of log(twi)/tfact, cpp:993-1004
    void* tmp = nullptr;
    if ((rc = dalloc(p, &tmp, (int64_t)mcf::twi_scratch_doubles() * 8))) return rc;
    p->d_twi2 = (double*)tmp;
    mcf::launch_twi_partial(p->d_soil[11], N, opt->tfact, p->d_twi2, p->stream);
    double h2[2];
    HIP_TRY(hipMemcpyAsync(h2, p->d_twi2, 16, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->twi_sum = h2[0];
    p->twi_count = (int64_t)h2[1];
    p->twi_mean = h2[0] / h2[1];

    // per-cell constants, tile-major (mcf_kernels.h CellSetupArgs); zeroed once: the cells past the raster's end in the last
    // tile's image are read (not used) by the solver
    p->ntiles = (N + p->cpb - 1) / p->cpb;
    const int64_t cellc_bytes = (int64_t)p->layers * p->ntiles * mcf::tile_image_doubles(p->cpb) * 8;
    if ((rc = dalloc(p, &tmp, cellc_bytes))) return rc;
    p->d_cellc = (double*)tmp;
    HIP_TRY(hipMemsetAsync(p->d_cellc, 0, (size_t)cellc_bytes, p->stream));
    if (p->layers > 1) {
        // day -> layer map from dfsel (cpp:2629-2640, k = dy*24 + hr + st[lyr]); -1 = not covered
        std::vector<int32_t> dl((size_t)std::max(p->ndays, 1), -1);
        for (int l = 0; l < p->layers; ++l) {
            int nd = (in->lyr_ed[l] - in->lyr_st[l] + 1) / 24, d0 = in->lyr_st[l] / 24;
            for (int d = d0; d < d0 + nd && d < p->ndays; ++d) dl[d] = l;
        }
        if ((rc = dalloc(p, &tmp, (int64_t)dl.size() * 4))) return rc;
        p->d_daylayer = (int32_t*)tmp;
        HIP_TRY(hipMemcpyAsync(p->d_daylayer, dl.data(), dl.size() * 4, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }

    // ---- time tables
    const double* raw[15];
    clim_ptrs(in, raw);
    for (int f = 0; f < 15; ++f)
        if (T > 0 && !raw[f] && !(p->coarse && f >= 1 && f <= 3))     // es, ea, tdew are derived in coarse mode
            return fail(MCF_ERR_ARG, std::string("missing forcing array: ") + kRawNames[f]);
    if (T > 0 && !p->coarse && !in->clim.winddir) return fail(MCF_ERR_ARG, "missing forcing array: winddir");
    // coarse mode: wind components per coarse cell and the raster-mean direction per step (R/internal.R:1255-1264)
    std::vector<double> cwu, cwv, cwd;
    if (p->coarse && T > 0) {
        const int64_t cN = (int64_t)p->crows * p->ccols;
        cwu.resize((size_t)(cN * T)); cwv.resize((size_t)(cN * T)); cwd.resize((size_t)T);
        for (int64_t k = 0; k < T; ++k) {
            double su = 0, sv = 0;
            int64_t nu = 0, nv = 0;
            for (int64_t q = 0; q < cN; ++q) {
                const double u2 = in->clim.windspeed[q + cN * k], wd = in->coarse_winddir[q + cN * k] * M_PI / 180.0;
                const double u = u2 * cos(wd), v = u2 * sin(wd);
                cwu[(size_t)(q + cN * k)] = u; cwv[(size_t)(q + cN * k)] = v;
                if (!std::isnan(u)) { su += u; ++nu; }               // apply(wu, 3, mean, na.rm = TRUE)
                if (!std::isnan(v)) { sv += v; ++nv; }
            }
            double d = fmod(atan2(sv / (double)nv, su / (double)nu) * 180.0 / M_PI, 360.0);
            if (d < 0) d += 360.0;                                    // R's %% takes the sign of the divisor
            cwd[(size_t)k] = d;
        }
    }
    std::vector<void*> temps;   // freed after setup
    auto up_tmp = [&](const void* host, int64_t nbytes, void** dev) -> int {
        if (nbytes <= 0) nbytes = 8;
        HIP_TRY(hipMalloc(dev, (size_t)nbytes));
        temps.push_back(*dev);
        if (host) HIP_TRY(hipMemcpyAsync(*dev, host, (size_t)nbytes, hipMemcpyHostToDevice, p->stream));
        return MCF_OK;
    };
    struct TmpGuard { std::vector<void*>& t; ~TmpGuard() { for (void* q : t) hipFree(q); } } tguard{temps};
    void *dy = nullptr, *dm = nullptr, *dd = nullptr, *dh = nullptr, *dw = nullptr;
    if (T > 0) {
        if ((rc = up_tmp(in->obstime.year, T * 4, &dy))) return rc;
        if ((rc = up_tmp(in->obstime.month, T * 4, &dm))) return rc;
        if ((rc = up_tmp(in->obstime.day, T * 4, &dd))) return rc;
        if ((rc = up_tmp(in->obstime.hour, T * 8, &dh))) return rc;
        if ((rc = up_tmp(p->coarse ? cwd.data() : in->clim.winddir, T * 8, &dw))) return rc;
    }
    if (!p->af) {
        double mxtc = -273.15;                                           // cpp:2159-2168
        for (int64_t k = 0; k < T; ++k)
            if (in->clim.tc[k] > mxtc) mxtc = in->clim.tc[k];
        p->g.dTmx = -0.6273 * mxtc + 49.79;                              // cpp:1236
        if ((rc = dalloc(p, &tmp, (int64_t)std::max(p->ndays, 1) * mcf::time_field_count() * 24 * 8))) return rc;
        p->d_tt = (double*)tmp;
        if (p->ndays > 0) {
            mcf::TimeSetupArgs ta{};
            ta.nsteps = p->ndays * 24;
            ta.year = (const int32_t*)dy; ta.month = (const int32_t*)dm; ta.day = (const int32_t*)dd;
            ta.hour = (const double*)dh; ta.winddir = (const double*)dw;
            for (int f = 0; f < 15; ++f) {
                void* q = nullptr;
                if ((rc = up_tmp(raw[f], T * 8, &q))) return rc;
                ta.raw[f] = (const double*)q;
            }
            ta.lat = in->lat; ta.lon = in->lon;
            ta.tt = p->d_tt;
            mcf::launch_time_setup(ta, p->stream);
            HIP_TRY(hipGetLastError());
        }
    } else {
        if ((rc = dalloc(p, &tmp, std::max<int64_t>(T, 1) * 4 * 8))) return rc;
        p->d_dt = (double*)tmp;
        if ((rc = dalloc(p, &tmp, std::max<int64_t>(T, 1) * 4))) return rc;
        p->d_windex = (int32_t*)tmp;
        if ((rc = dalloc(p, &tmp, N * 8))) return rc;
        p->d_mxtc = (double*)tmp;
        if (T > 0) {
            mcf::DateSetupArgs da{};
            da.nsteps = (int)T;
            da.year = (const int32_t*)dy; da.month = (const int32_t*)dm; da.day = (const int32_t*)dd;
            da.hour = (const double*)dh; da.winddir = (const double*)dw;
            da.dt = p->d_dt; da.windex = p->d_windex;
            mcf::launch_date_setup(da, p->stream);
            HIP_TRY(hipGetLastError());
        }
        if (p->coarse) {
            // the whole series of the 15 coarse slabs stays resident: [15][crows*ccols][T]
            const int64_t cN = (int64_t)p->crows * p->ccols, slab = cN * std::max<int64_t>(T, 1);
            if ((rc = dalloc(p, &tmp, 15 * slab * 8))) return rc;
            p->d_force = (double*)tmp;
            const double* src[15];
            for (int f = 0; f < 15; ++f) src[f] = raw[f];
            src[1] = in->coarse_relhum;                  // slot TF_ES
            src[2] = cwv.data();                         // slot TF_EA: v component
            src[3] = nullptr;                            // slot TF_TDEW unused
            if (p->altcorrect) src[4] = cpk_sea.data();  // slot TF_PK: sea-level pressure
            src[8] = cwu.data();                         // slot TF_U2: u component
            for (int f = 0; f < 15 && T > 0; ++f)
                if (src[f]) HIP_TRY(hipMemcpyAsync(p->d_force + f * slab, src[f], (size_t)(cN * T * 8), hipMemcpyHostToDevice, p->stream));
            if (T > 0) {
                mcf::launch_mxtc_coarse(p->d_force, slab, p->crows, p->ccols, (int)T, p->d_crowpos, p->d_ccolpos, p->rows, N,
                                        p->altcorrect, p->d_elevd, p->d_pkfac, p->d_mxtc, p->stream);
                HIP_TRY(hipGetLastError());
            } else {
                mcf::launch_fill(p->d_mxtc, N, -273.15, p->stream);
            }
            HIP_TRY(hipStreamSynchronize(p->stream));    // cwu / cwv are about to go out of scope
            p->force_day0.assign(ring_slots, 0);
            p->force_ndays.assign(ring_slots, p->ndays);
        } else {
        // per-cell max air temperature over the WHOLE series (cpp:2467-2471), streamed in slabs
        mcf::launch_fill(p->d_mxtc, N, -273.15, p->stream);
        int64_t slab_steps = std::max<int64_t>(1, std::min<int64_t>(T, (int64_t)(256LL << 20) / (N * 8)));
        void* dslab = nullptr;
        if (T > 0) {
            if ((rc = up_tmp(nullptr, slab_steps * N * 8, &dslab))) return rc;
            for (int64_t k0 = 0; k0 < T; k0 += slab_steps) {
                int64_t ns = std::min(slab_steps, T - k0);
                HIP_TRY(copy_in(p, dslab, in->clim.tc + p->pitch * p->cols * k0, p->cols * ns));
                mcf::launch_mxtc((const double*)dslab, N, (int)ns, p->d_mxtc, p->stream);
            }
            HIP_TRY(hipGetLastError());
        }
        // forcing ring, TILED like the output ring: [slots][tile][day][15 series][ring_block_doubles(cpb)] in the solver's lane
        // order, so that a workgroup's 360 loads per day (15 series x 24 hours, each 8 B x N x hour apart in the caller's
        // arrays) become whole lines of one contiguous 90 KB run; mcf_plan_upload_forcing_days re-lays each series on the
        // device behind its host-to-device copy
        p->ntiles = (N + p->cpb - 1) / p->cpb;
        p->force_day_stride = 15 * (int64_t)mcf::ring_block_doubles(p->cpb);
        p->force_tile_stride = (int64_t)ring_days * p->force_day_stride;
        p->force_slot_elems = p->ntiles * p->force_tile_stride;
        if ((rc = dalloc(p, &tmp, (int64_t)ring_slots * p->force_slot_elems * 8))) return rc;
        p->d_force = (double*)tmp;
        if ((rc = dalloc(p, &tmp, N * (int64_t)ring_days * 24 * 8))) return rc;     // one series of one slot, as uploaded
        p->d_force_stage = (double*)tmp;
        p->force_day0.assign(ring_slots, -1);
        p->force_ndays.assign(ring_slots, 0);
        }
    }

    // ---- output ring
    p->nvars = 0;
    for (int v = 0; v < MCF_NOUT; ++v) p->var_slot[v] = opt->out[v] ? p->nvars++ : -1;
    p->tiled = !p->bg;
    p->ntiles = (N + p->cpb - 1) / p->cpb;
    if (p->tiled) {
        const int64_t blk = mcf::ring_block_doubles(p->cpb);
        p->ring_var_stride = blk;
        p->ring_day_stride = (int64_t)std::max(p->nvars, 1) * blk;
        p->ring_tile_stride = (int64_t)ring_days * p->ring_day_stride;
        p->slot_elems = p->ntiles * p->ring_tile_stride;
    } else {
        p->slot_elems = (int64_t)std::max(p->nvars, 1) * N * (int64_t)ring_days * 24;
    }
    if ((rc = dalloc(p, &tmp, (int64_t)ring_slots * p->slot_elems * 8))) return rc;
    p->d_ring = (double*)tmp;

    // ---- below-ground series
    if (p->bg) {
        if ((rc = dalloc(p, &tmp, N * std::max<int64_t>(T, 1) * 8))) return rc;
        p->d_tgser = (double*)tmp;
        // steps past the last whole day read as 0 in the reference (std::vector<double> Tg(tsteps))
        mcf::launch_fill(p->d_tgser, N * T, 0.0, p->stream);
        if ((rc = dalloc(p, &tmp, N * 8))) return rc;
        p->d_ddsum = (double*)tmp;
        mcf::launch_fill(p->d_ddsum, N, 0.0, p->stream);
        if ((rc = dalloc(p, &tmp, N * 2 * (int64_t)std::max(p->ndays, 1) * 8))) return rc;
        p->d_scratch = (double*)tmp;
        if (!opt->complete) {
            int64_t n = p->af ? N * T : T;
            if ((rc = upload(p, in->pointm.Tg, n, &p->d_Tgp, "pointm$Tg", p->af))) return rc;
            if ((rc = upload(p, in->pointm.Tbp, n, &p->d_Tbp, "pointm$Tbp", p->af))) return rc;
        }
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    guard.p = nullptr;
    *out = p;
    return MCF_OK;
}

}  // extern "C"
namespace mcf {
int plan_ring_views(mcf_plan* p, int slot, RingView views[10], int32_t has[10], hipStream_t* stream, int64_t* N, int* device,
                    int* slot_days) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    if (slot < 0 || slot >= p->ring_slots) return fail(MCF_ERR_ARG, "slot out of range");
    for (int v = 0; v < MCF_NOUT; ++v) {
        has[v] = p->var_slot[v] >= 0;
        if (has[v]) views[v] = ring_view(p, slot, v);
    }
    *stream = p->stream; *N = p->N; *device = p->device; *slot_days = p->ring_days;
    return MCF_OK;
}
}  // namespace mcf
extern "C" {

int mcf_plan_set_mxtc(mcf_plan* p, double mxtc) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    if (p->af) return fail(MCF_ERR_STATE, "array forcing takes the maximum per cell from its series");
    p->g.dTmx = -0.6273 * mxtc + 49.79;                              // cpp:1236 (every launch takes the plan's globals)
    return MCF_OK;
}

// array forcing: the per-cell maximum air temperature (src/microclimfCpp.cpp:2467-2471) over the days with dayflag != 0 only — what
// runmicro2Cpp computes when `.runmicronosnow` hands it the no-snow-day SUBSET (R/internal.R:3689); streamed through the plan's
// forcing stage, a run of consecutive flagged days at a time
int mcf_plan_set_mxtc_days(mcf_plan* p, const mcf_grid_inputs* in, const int32_t* dayflag, int32_t ndays) {
    if (!p || !in || !dayflag) return fail(MCF_ERR_ARG, "null argument");
    if (!p->af || p->coarse) return fail(MCF_ERR_STATE, "mcf_plan_set_mxtc_days: a plan with (fine) array forcing");
    if (ndays < 0 || ndays > p->ndays) return fail(MCF_ERR_ARG, "day range out of bounds");
    if (!in->clim.tc) return fail(MCF_ERR_ARG, "missing forcing array: tc");
    HIP_TRY(hipSetDevice(p->device));
    mcf::launch_fill(p->d_mxtc, p->N, -273.15, p->stream);
    for (int d = 0; d < ndays;) {
        if (!dayflag[d]) { ++d; continue; }
        int e = d;
        while (e < ndays && dayflag[e] && e - d < p->ring_days) ++e;
        HIP_TRY(copy_in(p, p->d_force_stage, in->clim.tc + p->pitch * p->cols * (int64_t)d * 24, p->cols * (int64_t)(e - d) * 24));
        mcf::launch_mxtc((const double*)p->d_force_stage, p->N, (e - d) * 24, p->d_mxtc, p->stream);
        HIP_TRY(hipGetLastError());
        d = e;
    }
    return MCF_OK;
}

int mcf_plan_twi_partial(mcf_plan* p, double* sum, int64_t* count) {
    if (!p || !sum || !count) return fail(MCF_ERR_ARG, "null argument");
    *sum = p->twi_sum;
    *count = p->twi_count;
    return MCF_OK;
}

int mcf_plan_set_twi_mean(mcf_plan* p, double mean) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    p->twi_mean = mean;
    p->cells_ready = false;
    return MCF_OK;
}

int mcf_plan_upload_forcing_days(mcf_plan* p, const mcf_grid_inputs* in, int32_t day0, int32_t ndays, int32_t slot) {
    if (!p || !in) return fail(MCF_ERR_ARG, "null argument");
    if (!p->af) return fail(MCF_ERR_STATE, "plan uses vector forcing");
    if (p->coarse) return MCF_OK;            // the coarse series is resident since mcf_plan_create
    if (slot < 0 || slot >= p->ring_slots) return fail(MCF_ERR_ARG, "slot out of range");
    if (day0 < 0 || ndays < 1 || ndays > p->ring_days || day0 + ndays > p->ndays)
        return fail(MCF_ERR_ARG, "day range out of bounds");
    HIP_TRY(hipSetDevice(p->device));
    const double* raw[15];
    clim_ptrs(in, raw);
    const int64_t N = p->N;
    for (int f = 0; f < 15; ++f)
        if (!raw[f]) return fail(MCF_ERR_ARG, std::string("missing forcing array: ") + kRawNames[f]);
    for (int f = 0; f < 15; ++f) {
        // the series as the caller holds it ([rows, cols, steps]) into the staging slab, then into its tiled place on the device
        // (stream order: the next copy into the slab waits for this series' kernel)
        const double* src = raw[f] + p->pitch * p->cols * (int64_t)day0 * 24;
        HIP_TRY(copy_in(p, p->d_force_stage, src, p->cols * (int64_t)ndays * 24));
        mcf::RingView v{};
        v.base = p->d_force + (int64_t)slot * p->force_slot_elems + (int64_t)f * mcf::ring_block_doubles(p->cpb);
        v.N = N; v.tile_stride = p->force_tile_stride; v.day_stride = p->force_day_stride; v.cpb = p->cpb;
        mcf::launch_tile_series(p->d_force_stage, (int64_t)ndays * 24, v, p->stream);
        HIP_TRY(hipGetLastError());
    }
    p->force_day0[slot] = day0;
    p->force_ndays[slot] = ndays;
    return MCF_OK;
}

int mcf_plan_run_days(mcf_plan* p, int32_t day0, int32_t ndays, int32_t slot) {
    return mcf_plan_run_days_at(p, day0, ndays, slot, 0);
}

int mcf_plan_run_days_at(mcf_plan* p, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0) {
    return mcf_plan_run_days_masked(p, day0, ndays, slot, slot_day0, nullptr, 0);
}

namespace {
// the launch description of days [day0, day0 + ndays) into `slot` from its day `slot_day0` on — everything but the tile list —,
// and which instantiation the days allow: `fast` (the min / max clamps: every day regular), `soil_daily` (the per cell-day soil
// state shared through LDS)
int solve_args(mcf_plan* p, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0, mcf::SolveArgs& a, bool& fast, bool& soil_daily) {
    if (slot < 0 || slot >= p->ring_slots) return fail(MCF_ERR_ARG, "slot out of range");
    if (day0 < 0 || ndays < 1 || day0 + ndays > p->ndays) return fail(MCF_ERR_ARG, "day range out of bounds");
    if (!p->bg && (slot_day0 < 0 || slot_day0 + ndays > p->ring_days)) return fail(MCF_ERR_ARG, "more days than the ring slot holds");
    if (slot_day0 != 0 && p->bg) return fail(MCF_ERR_ARG, "a day offset inside the slot needs reqhgt >= 0");
    HIP_TRY(hipSetDevice(p->device));
    int rc = ensure_cells(p);
    if (rc) return rc;
    a = mcf::SolveArgs{};
    a.N = p->N;
    a.cellc = p->d_cellc; a.ntiles_total = p->ntiles; a.tt = p->d_tt;
    a.daylayer = p->d_daylayer;
    const int64_t cap = p->N * (int64_t)p->ring_days * 24;
    if (p->af && p->coarse) {
        a.af_base = p->d_force;
        a.af_stride = (int64_t)p->crows * p->ccols * std::max<int64_t>(p->tsteps, 1);
        a.crows = p->crows; a.ccols = p->ccols;
        a.altcorrect = p->altcorrect;
        a.dt = p->d_dt; a.windex = p->d_windex; a.mxtc = p->d_mxtc;
    } else if (p->af) {
        // (the days may be a run INSIDE what the slot's forcing holds — the snow run solves a chunk's no-snow days at their own
        // place: the kernel counts forcing days from its first day, so the base is moved to that day's block)
        if (p->force_day0[slot] > day0 || p->force_day0[slot] + p->force_ndays[slot] < day0 + ndays)
            return fail(MCF_ERR_STATE, "forcing for these days has not been uploaded to this slot");
        a.af_base = p->d_force + (int64_t)slot * p->force_slot_elems + (int64_t)(day0 - p->force_day0[slot]) * p->force_day_stride;
        a.af_tile_stride = p->force_tile_stride; a.af_day_stride = p->force_day_stride;
        a.dt = p->d_dt; a.windex = p->d_windex; a.mxtc = p->d_mxtc;
    }
    a.out_base = p->d_ring + (int64_t)slot * p->slot_elems;
    a.out_stride = cap;
    a.out_tile_stride = p->ring_tile_stride; a.out_day_stride = p->ring_day_stride; a.out_var_stride = p->ring_var_stride;
    a.slot_day0 = slot_day0;
    a.out_sel = 0;
    for (int v = 0; v < MCF_NOUT; ++v)
        a.out_sel |= (uint64_t)(p->var_slot[v] < 0 ? 15 : p->var_slot[v]) << (4 * v);
    a.slot_step0 = p->bg ? (int64_t)day0 * 24 : 0;
    a.tgser = p->d_tgser; a.ddsum = p->d_ddsum;
    a.day0 = day0; a.ndays = ndays; a.total_days = p->ndays;
    {
        // pass 2 yields Tz (+ tleaf, relhum for reqhgt > 0) and the long-wave fluxes; requests for
        // soilm / windspeed / short-wave fluxes alone are served by pass 1
        const bool o0 = p->var_slot[MCF_OUT_TZ] >= 0, o1 = p->var_slot[MCF_OUT_TLEAF] >= 0,
                   o2 = p->var_slot[MCF_OUT_RELHUM] >= 0, o7 = p->var_slot[MCF_OUT_RLWDOWN] >= 0,
                   o9 = p->var_slot[MCF_OUT_RLWUP] >= 0;
        const double rq = p->opt.reqhgt;
        a.need_tv = (rq > 0.0 && (o0 || o1 || o2 || o7 || o9)) || (rq == 0.0 && (o7 || o9));
        a.need_pass2 = p->bg ? (o0 ? 1 : 0) : (a.need_tv || (rq == 0.0 && o0));
    }
    a.g = p->g;
    a.fix_count = p->d_fix_count; a.fix_list = p->d_fix_list; a.fix_cap = p->fix_cap;
    fast = p->fast_enabled && p->n_fast > 0;
    for (int d = day0; fast && !p->af && d < day0 + ndays; ++d)     // array forcing: every lane checks its own forcing values
        if (p->day_irregular[(size_t)d]) fast = false;
    soil_daily = !p->af && !p->day_soil_daily.empty();
    for (int d = day0; soil_daily && d < day0 + ndays; ++d)
        if (!p->day_soil_daily[(size_t)d]) soil_daily = false;
    if (p->coarse) soil_daily = p->coarse_lds;      // (coarse array forcing: the same launch flag selects the LDS-staged taps)
    return MCF_OK;
}
}  // namespace

int mcf_plan_run_days_masked(mcf_plan* p, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0, const uint8_t* skip_tile,
                             int64_t n_skip_tile) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    if (skip_tile && (p->bg || p->af)) return fail(MCF_ERR_ARG, "a tile mask needs vector forcing and reqhgt >= 0");
    if (skip_tile && n_skip_tile != (p->ntiles > 0 ? p->ntiles : (p->N + p->cpb - 1) / p->cpb))
        return fail(MCF_ERR_ARG, "the tile mask's length is not the plan's number of tiles");
    mcf::SolveArgs a{};
    bool fast = false, soil_daily = false;
    int rc = solve_args(p, day0, ndays, slot, slot_day0, a, fast, soil_daily);
    if (rc) return rc;
    // a tile mask: the launch's tile lists are the plan's minus the masked tiles (the kernel takes any list; a tile it is not
    // given is simply not touched in the slot)
    const int32_t *sub_fast = nullptr, *sub_slow = nullptr;
    int64_t n_sub_fast = 0, n_sub_slow = 0;
    if (skip_tile) {
        const int64_t ntiles = p->ntiles;
        if (p->tiles_sub_cap < ntiles) {
            void* q;
            if ((rc = dalloc(p, &q, ntiles * 4))) return rc;
            p->d_tiles_sub = (int32_t*)q;
            p->tiles_sub_cap = ntiles;
        }
        // (the lists live in the plan: they go up on the plan's own stream, in order with the launches that read them, and an
        // earlier masked launch may still be reading the device copy or an earlier copy the host one)
        HIP_TRY(hipStreamSynchronize(p->stream));
        std::vector<int32_t>& l = p->h_tiles_sub;
        l.clear();
        l.reserve((size_t)ntiles);
        if (fast) {
            for (int32_t t : p->h_tiles_fast) if (!skip_tile[t]) l.push_back(t);
            n_sub_fast = (int64_t)l.size();
            for (int32_t t : p->h_tiles_slow) if (!skip_tile[t]) l.push_back(t);
            n_sub_slow = (int64_t)l.size() - n_sub_fast;
        } else {
            for (int64_t t = 0; t < ntiles; ++t) if (!skip_tile[t]) l.push_back((int32_t)t);
            n_sub_fast = (int64_t)l.size();
        }
        p->masked_tiles_skipped += ntiles - n_sub_fast - n_sub_slow;
        if (!l.empty()) HIP_TRY(hipMemcpyAsync(p->d_tiles_sub, l.data(), l.size() * 4, hipMemcpyHostToDevice, p->stream));
        sub_fast = p->d_tiles_sub; sub_slow = p->d_tiles_sub + n_sub_fast;
    }
    auto launch = [&]() {
        if (skip_tile) {
            if (fast) (void)hipMemsetAsync(p->d_fix_count, 0, 4, p->stream);
            if (n_sub_fast) {
                a.tile_list = sub_fast; a.ntiles_launch = n_sub_fast;
                if (fast) { mcf::launch_solve(a, p->cpb, p->af, false, true, soil_daily, p->stream); ++p->fast_launches; }
                else { mcf::launch_solve(a, p->cpb, p->af, p->bg, false, soil_daily, p->stream); ++p->slow_launches; }
            }
            if (n_sub_slow) {
                a.tile_list = sub_slow; a.ntiles_launch = n_sub_slow;
                mcf::launch_solve(a, p->cpb, p->af, false, false, soil_daily, p->stream);
                ++p->slow_launches;
            }
        } else if (fast) {
            (void)hipMemsetAsync(p->d_fix_count, 0, 4, p->stream);
            a.tile_list = p->n_slow > 0 ? p->d_tiles_fast : nullptr;
            a.ntiles_launch = p->n_fast;
            mcf::launch_solve(a, p->cpb, p->af, false, true, soil_daily, p->stream);
            ++p->fast_launches;
            if (p->n_slow > 0) {
                ++p->slow_launches;
                a.tile_list = p->d_tiles_slow;
                a.ntiles_launch = p->n_slow;
                mcf::launch_solve(a, p->cpb, p->af, false, false, soil_daily, p->stream);
            }
        } else {
            a.tile_list = nullptr;
            a.ntiles_launch = 0;
            ++p->slow_launches;
            mcf::launch_solve(a, p->cpb, p->af, p->bg, false, soil_daily, p->stream);
        }
    };
    if (p->ktiming) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, p->stream));
        launch();
        HIP_TRY(hipEventRecord(e1, p->stream));
        p->kev.emplace_back(e0, e1);
    } else {
        launch();
    }
    HIP_TRY(hipGetLastError());
    return MCF_OK;
}

int mcf_plan_run_days_cells(mcf_plan* p, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0, const uint8_t* need_cell,
                            int64_t n_cells, int64_t* n_gathered) {
    if (!p || !need_cell) return fail(MCF_ERR_ARG, "null argument");
    if (p->bg || p->af || !p->tiled) return fail(MCF_ERR_ARG, "a cell subset needs vector forcing and reqhgt >= 0");
    if (n_cells != p->N) return fail(MCF_ERR_ARG, "the cell flags' length is not the plan's number of cells");
    mcf::SolveArgs a{};
    bool fast = false, soil_daily = false;
    int rc = solve_args(p, day0, ndays, slot, slot_day0, a, fast, soil_daily);
    if (rc) return rc;
    if (n_gathered) *n_gathered = 0;
    const int64_t N = p->N;
    const int cpb = p->cpb;
    const int64_t nb = (N + 255) / 256, IMG = mcf::tile_image_doubles(cpb), blk = mcf::ring_block_doubles(cpb);
    // The list of cells is made on a stream of its own: the plan's stream — the previous chunk's solver launches, the snow-day
    // microclimate over them — is not drained for it, only made to wait for the list.  An earlier call's launches and copies may
    // still read the buffers and host vectors: its last event first.
    if (!p->prep) {
        HIP_TRY(hipStreamCreateWithFlags(&p->prep, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&p->ev_prep, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->ev_cells_done, hipEventDisableTiming));
    }
    const bool trace = getenv("MCF_CELLS_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (p->cells_inflight) { HIP_TRY(hipEventSynchronize(p->ev_cells_done)); p->cells_inflight = false; }
    const double t1 = now();
    if ((rc = dregrow(p, (void**)&p->d_cls, &p->cls_cap, N))) return rc;
    if ((rc = dregrow(p, (void**)&p->d_blockcnt, &p->blockcnt_cap, nb * 8))) return rc;
    // 1. the wanted cells by class, counted per 256 cells; the list offsets are the host's prefix sums of those counts
    mcf::launch_cells_class(need_cell, p->fast_enabled ? p->d_tile_regular : nullptr, N, cpb, p->d_cls, p->d_blockcnt, p->prep);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t>& cnt = p->h_blockcnt;
    cnt.resize((size_t)nb * 2);
    HIP_TRY(hipMemcpyAsync(cnt.data(), p->d_blockcnt, (size_t)nb * 8, hipMemcpyDeviceToHost, p->prep));
    HIP_TRY(hipStreamSynchronize(p->prep));
    const double t2 = now();
    int64_t n1 = 0, n2 = 0;
    for (int64_t b = 0; b < nb; ++b) { n1 += cnt[(size_t)(2 * b)]; n2 += cnt[(size_t)(2 * b + 1)]; }
    if (n_gathered) *n_gathered = n1 + n2;
    if (n1 + n2 == 0) return MCF_OK;
    // the cells of the plan's fast tiles first, in whole tiles; those of its other tiles in tiles of their own behind them: every
    // cell meets the instantiation it meets in a launch of the plan's own tiles (the two differ in the last bits)
    const int64_t base2 = (n1 + cpb - 1) / cpb * cpb, total = base2 + (n2 + cpb - 1) / cpb * cpb;
    const int64_t nt_sub = total / cpb, nt_fast = base2 / cpb, nt_slow = nt_sub - nt_fast;
    if (total > INT32_MAX) return fail(MCF_ERR_ARG, "too many cells");
    {
        int64_t o1 = 0, o2 = base2;
        for (int64_t b = 0; b < nb; ++b) {
            const int32_t c1 = cnt[(size_t)(2 * b)], c2 = cnt[(size_t)(2 * b + 1)];
            cnt[(size_t)(2 * b)] = (int32_t)o1; cnt[(size_t)(2 * b + 1)] = (int32_t)o2;
            o1 += c1; o2 += c2;
        }
    }
    // (buffers that grow are released first: hipFree waits for the device)
    if ((rc = dregrow(p, (void**)&p->d_celllist, &p->celllist_cap, total * 4))) return rc;
    if ((rc = dregrow(p, (void**)&p->d_subimg, &p->subimg_cap, (int64_t)p->layers * nt_sub * IMG * 8))) return rc;
    const int64_t day_doubles = p->ring_day_stride;                  // variables x block
    const double budget_gb = getenv("MCF_CELLS_RING_GB") ? atof(getenv("MCF_CELLS_RING_GB")) : 0.0;
    const int64_t budget = budget_gb > 0.0 ? (int64_t)(budget_gb * 1e9)
                                           : std::max<int64_t>((int64_t)512 << 20, (int64_t)p->ring_slots * p->slot_elems);       // (bytes: an eighth of the ring)
    const int pass_days = (int)std::max<int64_t>(1, std::min<int64_t>(ndays, budget / (nt_sub * day_doubles * 8)));
    if ((rc = dregrow(p, (void**)&p->d_subring, &p->subring_cap, nt_sub * pass_days * day_doubles * 8))) return rc;
    if (nt_slow > 0 && fast && (rc = dregrow(p, (void**)&p->d_cells_slow, &p->cells_slow_cap, nt_slow * 4))) return rc;
    const double t3 = now();
    HIP_TRY(hipMemcpyAsync(p->d_blockcnt, cnt.data(), (size_t)nb * 8, hipMemcpyHostToDevice, p->prep));
    HIP_TRY(hipMemsetAsync(p->d_celllist, 0xFF, (size_t)total * 4, p->prep));         // -1: no cell (a class's last tile)
    mcf::launch_cells_place(p->d_cls, N, p->d_blockcnt, p->d_celllist, p->prep);
    HIP_TRY(hipGetLastError());
    if (nt_slow > 0 && fast) {          // the second class's tile numbers, for the launch that takes a list
        std::vector<int32_t>& l = p->h_cells_slow;
        l.resize((size_t)nt_slow);
        for (int64_t t = 0; t < nt_slow; ++t) l[(size_t)t] = (int32_t)(nt_fast + t);
        HIP_TRY(hipMemcpyAsync(p->d_cells_slow, l.data(), l.size() * 4, hipMemcpyHostToDevice, p->prep));
    }
    HIP_TRY(hipEventRecord(p->ev_prep, p->prep));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_prep, 0));
    // 2. their tiles' images
    mcf::launch_gather_image(p->d_celllist, nt_sub, p->d_cellc, p->ntiles, p->layers, cpb, p->d_subimg, p->stream);
    HIP_TRY(hipGetLastError());
    // 3. the solver on them, into a ring of their own — as many days at a time as the budget holds —, and from there to the
    // cells' places in the slot
    a.cellc = p->d_subimg; a.ntiles_total = nt_sub; a.N = total;
    a.out_base = p->d_subring;
    a.out_tile_stride = (int64_t)pass_days * day_doubles; a.out_day_stride = day_doubles; a.out_var_stride = blk;
    a.slot_day0 = 0;
    double* const slot_base = p->d_ring + (int64_t)slot * p->slot_elems + (int64_t)slot_day0 * p->ring_day_stride;
    for (int d = 0; d < ndays; d += pass_days) {
        const int nd = std::min(pass_days, ndays - d);
        a.day0 = day0 + d; a.ndays = nd;
        if (fast) {
            (void)hipMemsetAsync(p->d_fix_count, 0, 4, p->stream);
            if (nt_fast > 0) {
                a.tile_list = nullptr; a.ntiles_launch = nt_fast;
                mcf::launch_solve(a, cpb, false, false, true, soil_daily, p->stream);
                ++p->fast_launches;
            }
            if (nt_slow > 0) {
                a.tile_list = p->d_cells_slow; a.ntiles_launch = nt_slow;
                mcf::launch_solve(a, cpb, false, false, false, soil_daily, p->stream);
                ++p->slow_launches;
            }
        } else {
            a.tile_list = nullptr; a.ntiles_launch = nt_sub;
            mcf::launch_solve(a, cpb, false, false, false, soil_daily, p->stream);
            ++p->slow_launches;
        }
        HIP_TRY(hipGetLastError());
        mcf::launch_scatter_cells(p->d_celllist, nt_sub, p->d_subring, a.out_tile_stride, slot_base + (int64_t)d * p->ring_day_stride,
                                  p->ring_tile_stride, day_doubles, cpb, nd, p->stream);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(p->ev_cells_done, p->stream));
    if (trace)
        fprintf(stderr, "[mcf] run_days_cells: %lld cells x %d days; host ms: wait %.3f, class+counts %.3f, offsets+buffers %.3f, enqueue %.3f\n",
                (long long)(n1 + n2), (int)ndays, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    p->cells_inflight = true;
    ++p->cells_runs;
    p->cells_gathered += (n1 + n2) * ndays;
    return MCF_OK;
}

int mcf_plan_dispatch_stats(mcf_plan* p, mcf_dispatch_stats* st) {
    if (!p || !st) return fail(MCF_ERR_ARG, "null argument");
    memset(st, 0, sizeof *st);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    st->fast_tiles = p->fast_enabled ? p->n_fast : 0;
    st->slow_tiles = p->fast_enabled ? p->n_slow : (p->N + p->cpb - 1) / p->cpb;
    for (char c : p->day_irregular) st->irregular_days += c ? 1 : 0;
    st->fast_launches = p->fast_launches;
    st->slow_launches = p->slow_launches;
    if (p->d_fix_count) {
        int32_t v[2] = {0, 0};
        HIP_TRY(hipMemcpy(v, p->d_fix_count, 8, hipMemcpyDeviceToHost));
        st->canary_trips = v[1];
    }
    return MCF_OK;
}

int mcf_plan_belowground(mcf_plan* p) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    if (!p->bg) return fail(MCF_ERR_STATE, "reqhgt >= 0: nothing to smooth");
    if (p->var_slot[MCF_OUT_TZ] < 0) return MCF_OK;                      // cpp:2307 `&& out[0]`
    HIP_TRY(hipSetDevice(p->device));
    mcf::BelowArgs b{};
    b.N = p->N; b.tsteps = (int)p->tsteps; b.complete = p->opt.complete; b.hiy = p->hiy;
    b.per_cell_pointm = p->af ? 1 : 0;
    b.reqhgt = p->opt.reqhgt; b.mat = p->opt.mat;
    b.cellflag_hgt = p->d_veg[0];
    b.tg = p->d_tgser; b.ddsum = p->d_ddsum; b.Tgp = p->d_Tgp; b.Tbp = p->d_Tbp;
    b.scratch = p->d_scratch;
    b.tz = const_cast<double*>(ring_view(p, 0, MCF_OUT_TZ).base);      // reqhgt < 0: the linear ring
    if (p->tsteps > (int64_t)p->ring_days * 24)
        return fail(MCF_ERR_STATE, "ring slot smaller than the series");
    mcf::launch_belowground(b, p->stream);
    HIP_TRY(hipGetLastError());
    return MCF_OK;
}

int mcf_plan_sync(mcf_plan* p) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MCF_OK;
}

int mcf_plan_fetch(mcf_plan* p, int32_t slot, int32_t var, int64_t step0, int64_t nsteps, double* host_dst) {
    return mcf_plan_fetch_pitched(p, slot, var, step0, nsteps, host_dst, 0);
}

int mcf_plan_fetch_pitched(mcf_plan* p, int32_t slot, int32_t var, int64_t step0, int64_t nsteps, double* host_dst, int64_t row_pitch) {
    if (!p || !host_dst) return fail(MCF_ERR_ARG, "null argument");
    if (row_pitch == 0) row_pitch = p->rows;
    if (row_pitch < p->rows) return fail(MCF_ERR_ARG, "row_pitch smaller than rows");
    if (slot < 0 || slot >= p->ring_slots || var < 0 || var >= MCF_NOUT) return fail(MCF_ERR_ARG, "bad slot/var");
    if (p->var_slot[var] < 0) return fail(MCF_ERR_ARG, "variable was not requested in out[]");
    const int64_t cap_steps = (int64_t)p->ring_days * 24;
    if (step0 < 0 || nsteps < 0 || step0 + nsteps > cap_steps) return fail(MCF_ERR_ARG, "step range out of slot");
    HIP_TRY(hipSetDevice(p->device));
    if (nsteps == 0) return MCF_OK;
    const mcf::RingView view = ring_view(p, slot, var);
    // large results: pinned ring + host copy threads instead of hipMemcpy's single-threaded staging
    static const bool no_pipe = getenv("MCF_NO_HOSTPIPE") != nullptr;
    auto to_host = [&](double* dst, const double* src, size_t bytes) -> int {
        if (row_pitch != p->rows) {      // a block of a taller raster: column by column into its place
            const size_t width = (size_t)p->rows * 8, height = bytes / width;
            if (bytes >= ((size_t)64 << 20) && width <= mcf::HostPipe::kPiece && !no_pipe && ensure_pipe(p)) {
                // contiguous DMA into the pinned ring, the scatter by the host copy threads (mcf_hostpipe.hpp)
                HIP_TRY(hipEventRecord(p->ev_pipe, p->stream));
                HIP_TRY(p->pipe->copy_pitched(dst, (size_t)row_pitch * 8, src, width, height, p->ev_pipe));
                return MCF_OK;
            }
            HIP_TRY(hipMemcpy2DAsync(dst, (size_t)row_pitch * 8, src, width, width, height, hipMemcpyDeviceToHost, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
            return MCF_OK;
        }
        if (bytes >= ((size_t)64 << 20) && !no_pipe && ensure_pipe(p)) {
            HIP_TRY(hipEventRecord(p->ev_pipe, p->stream));
            HIP_TRY(p->pipe->copy(dst, src, bytes, p->ev_pipe));
            return MCF_OK;
        }
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return MCF_OK;
    };
    if (!p->tiled) return to_host(host_dst, view.base + p->N * step0, (size_t)(p->N * nsteps) * 8);
    // tiled ring: the reference's [rows, cols, steps] layout is made on the device (k_untile), in pieces of <= 1 GB
    const int64_t piece = std::max<int64_t>(1, std::min<int64_t>(nsteps, ((int64_t)1 << 27) / std::max<int64_t>(p->N, 1)));
    if (p->stage_elems < piece * p->N) {
        int rc;
        void* q;
        if ((rc = dalloc(p, &q, piece * p->N * 8))) return rc;      // an earlier (smaller) buffer is released with the plan
        p->d_stage = (double*)q;
        p->stage_elems = piece * p->N;
    }
    for (int64_t k0 = 0; k0 < nsteps; k0 += piece) {
        const int64_t n = std::min(piece, nsteps - k0);
        mcf::launch_untile(view, step0 + k0, n, p->d_stage, p->stream);
        HIP_TRY(hipGetLastError());
        int rc = to_host(host_dst + row_pitch * p->cols * k0, p->d_stage, (size_t)(p->N * n) * 8);
        if (rc) return rc;
    }
    return MCF_OK;
}

int mcf_plan_fetch_cells(mcf_plan* p, int32_t slot, int32_t var, int64_t step0, int64_t nsteps, const int64_t* cells,
                         int64_t ncells, double* host_dst) {
    if (!p || !host_dst || !cells) return fail(MCF_ERR_ARG, "null argument");
    if (slot < 0 || slot >= p->ring_slots || var < 0 || var >= MCF_NOUT) return fail(MCF_ERR_ARG, "bad slot/var");
    if (p->var_slot[var] < 0) return fail(MCF_ERR_ARG, "variable was not requested in out[]");
    const int64_t cap_steps = (int64_t)p->ring_days * 24;
    if (step0 < 0 || nsteps < 0 || step0 + nsteps > cap_steps) return fail(MCF_ERR_ARG, "step range out of slot");
    if (ncells < 0) return fail(MCF_ERR_ARG, "negative cell count");
    for (int64_t i = 0; i < ncells; ++i)
        if (cells[i] < 0 || cells[i] >= p->N) return fail(MCF_ERR_ARG, "cell index outside the raster");
    if (ncells == 0 || nsteps == 0) return MCF_OK;
    HIP_TRY(hipSetDevice(p->device));
    int64_t* d_cells = nullptr;
    double* d_dst = nullptr;
    HIP_TRY(hipMalloc((void**)&d_cells, (size_t)ncells * 8));
    hipError_t e = hipMalloc((void**)&d_dst, (size_t)(ncells * nsteps) * 8);
    if (e != hipSuccess) { (void)hipFree(d_cells); return fail(MCF_ERR_NOMEM, "hipMalloc failed for the gather buffer"); }
    e = hipMemcpyAsync(d_cells, cells, (size_t)ncells * 8, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        mcf::launch_gather_cells(ring_view(p, slot, var), step0, nsteps, d_cells, ncells, d_dst, p->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, d_dst, (size_t)(ncells * nsteps) * 8, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    (void)hipFree(d_cells);
    (void)hipFree(d_dst);
    if (e != hipSuccess) return fail(MCF_ERR_HIP, std::string("mcf_plan_fetch_cells: ") + hipGetErrorString(e));
    return MCF_OK;
}

int mcf_plan_fetch_packed(mcf_plan* p, int32_t slot, int32_t var, int64_t step0, int64_t nsteps, double scale,
                          int32_t* host_dst, float* kernel_ms) {
    if (!p || !host_dst) return fail(MCF_ERR_ARG, "null argument");
    if (slot < 0 || slot >= p->ring_slots || var < 0 || var >= MCF_NOUT) return fail(MCF_ERR_ARG, "bad slot/var");
    if (p->var_slot[var] < 0) return fail(MCF_ERR_ARG, "variable was not requested in out[]");
    const int64_t cap_steps = (int64_t)p->ring_days * 24;
    if (step0 < 0 || nsteps < 0 || step0 + nsteps > cap_steps) return fail(MCF_ERR_ARG, "step range out of slot");
    if (nsteps > 65535) return fail(MCF_ERR_ARG, "at most 65535 steps per packed fetch");
    HIP_TRY(hipSetDevice(p->device));
    if (p->pack_elems < p->N * nsteps) {
        int rc;
        void* q;
        if ((rc = dalloc(p, &q, p->N * nsteps * 4))) return rc;     // earlier (smaller) buffers are released with the plan
        p->d_pack = (int32_t*)q;
        p->pack_elems = p->N * nsteps;
    }
    if (kernel_ms) HIP_TRY(hipEventRecord(p->ev0, p->stream));
    mcf::launch_pack_transpose(ring_view(p, slot, var), step0, p->rows, p->cols, nsteps, scale, p->d_pack, p->stream);
    HIP_TRY(hipGetLastError());
    if (kernel_ms) HIP_TRY(hipEventRecord(p->ev1, p->stream));
    HIP_TRY(hipMemcpyAsync(host_dst, p->d_pack, (size_t)(p->N * nsteps) * 4, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, p->ev0, p->ev1));
    return MCF_OK;
}

int mcf_nc_create(const char* path, const mcf_nc_spec* sp, mcf_ncfile** out) {
    if (!path || !sp || !out) return fail(MCF_ERR_ARG, "null argument");
    if (sp->rows <= 0 || sp->cols <= 0 || sp->nsteps < 0 || !sp->east || !sp->north || (sp->nsteps > 0 && !sp->time_hours))
        return fail(MCF_ERR_ARG, "mcf_nc_create: bad dimensions or null coordinate vector");
    mcf_ncfile* nc = new mcf_ncfile();
    std::vector<mcf::NcVarDef> defs;
    for (int v = 0; v < MCF_NOUT; ++v) {
        nc->var_of[v] = -1;
        if (!sp->vars[v]) continue;
        mcf::NcVarDef d;
        double sc;
        bool put;
        if (!nc_var_def(v, sp->reqhgt, &d, &sc, &put)) {
            delete nc;
            return fail(MCF_ERR_ARG, "mcf_nc_create: writetonc defines no variable '" + d.name + "' at this reqhgt");
        }
        const int k = (int)defs.size();
        nc->var_of[v] = k; nc->out_of[k] = v; nc->scale[k] = sc;
        nc->fill_only[k] = (sp->reference_puts_only && !put) ? 1 : 0;
        defs.push_back(d);
    }
    if (defs.empty()) { delete nc; return fail(MCF_ERR_ARG, "mcf_nc_create: no variable selected"); }
    std::string e;
    if (sp->format == MCF_NC_NETCDF4) {
        // the reference's container (dataprep.R:1110: compression = 9); deflate_level 0 = the reference's 9, -1 = none
        const int level = sp->deflate_level == 0 ? 9 : sp->deflate_level < 0 ? 0 : sp->deflate_level;
        mcf::Nc4File* f4 = new mcf::Nc4File();
        nc->f.reset(f4);
        if (sp->nsteps == 0) e = "a netCDF-4 file needs at least one time step";
        else if (level > 9) e = "deflate_level above 9";
        else e = f4->create4(path, sp->rows, sp->cols, sp->nsteps, sp->east, sp->north, sp->time_hours, sp->crs_wkt, defs, level);
    } else if (sp->format == MCF_NC_CLASSIC) {
        nc->f.reset(new mcf::NcFile());
        e = nc->f->create(path, sp->rows, sp->cols, sp->nsteps, sp->east, sp->north, sp->time_hours, sp->crs_wkt, defs);
    } else {
        e = "unknown format";
    }
    if (!e.empty()) { delete nc; return fail(MCF_ERR_ARG, "mcf_nc_create: " + e); }
    *out = nc;
    return MCF_OK;
}

int mcf_nc_close(mcf_ncfile* nc) {
    if (!nc) return MCF_OK;
    const std::string e = nc->f->close();
    delete nc;
    return e.empty() ? MCF_OK : fail(MCF_ERR_ARG, "mcf_nc_close: " + e);
}

int mcf_nc_write_host(mcf_ncfile* nc, int64_t step0, int64_t nsteps, const double* const vars[MCF_NOUT]) {
    if (!nc || !vars) return fail(MCF_ERR_ARG, "null argument");
    if (step0 < 0 || nsteps < 0 || step0 + nsteps > nc->f->nsteps) return fail(MCF_ERR_ARG, "mcf_nc_write_host: step range outside the file");
    const int64_t R = nc->f->rows, C = nc->f->cols, N = R * C, rb = nc->f->rec_bytes;
    for (int k = 0; k < nc->f->nvars; ++k)
        if (!nc->fill_only[k] && !vars[nc->out_of[k]]) return fail(MCF_ERR_ARG, "mcf_nc_write_host: a variable of the file is missing");
    const int64_t piece = std::max<int64_t>(1, ((int64_t)64 << 20) / rb);
    for (int64_t s0 = 0; s0 < nsteps; s0 += piece) {
        const int64_t n = std::min(piece, nsteps - s0);
        uint8_t* st = nc->staging(0, (size_t)(n * rb));
        for (int64_t s = 0; s < n; ++s)
            for (int k = 0; k < nc->f->nvars; ++k) {
                uint8_t* dst = st + s * rb + 8 + (int64_t)k * N * 4;
                const double* src = nc->fill_only[k] ? nullptr : vars[nc->out_of[k]] + (s0 + s) * N;
                for (int64_t r = 0; r < R; ++r)
                    for (int64_t c = 0; c < C; ++c)
                        mcf::NcFile::store_i32(dst + 4 * (c + C * r), src ? nc_pack(src[r + R * c], nc->scale[k]) : mcf::NcFile::kMissval);
            }
        const std::string e = nc->f->write_records(step0 + s0, n, st);
        if (!e.empty()) return fail(MCF_ERR_ARG, "mcf_nc_write_host: " + e);
    }
    return MCF_OK;
}

int mcf_nc_write_plan(mcf_ncfile* nc, mcf_plan* p, int32_t slot, int64_t slot_step0, int64_t file_step0, int64_t nsteps,
                      float* kernel_ms) {
    if (!nc || !p) return fail(MCF_ERR_ARG, "null argument");
    if (nc->f->rows != p->rows || nc->f->cols != p->cols) return fail(MCF_ERR_ARG, "mcf_nc_write_plan: the file's grid is not the plan's");
    if (slot < 0 || slot >= p->ring_slots) return fail(MCF_ERR_ARG, "bad slot");
    const int64_t cap_steps = (int64_t)p->ring_days * 24;
    if (slot_step0 < 0 || nsteps < 0 || slot_step0 + nsteps > cap_steps) return fail(MCF_ERR_ARG, "step range out of slot");
    if (file_step0 < 0 || file_step0 + nsteps > nc->f->nsteps) return fail(MCF_ERR_ARG, "mcf_nc_write_plan: step range outside the file");
    mcf::PackNcArgs a{};
    a.nv = nc->f->nvars; a.missval = mcf::NcFile::kMissval; a.rows = p->rows; a.cols = p->cols;
    a.rec_words = nc->f->rec_bytes / 4;
    for (int k = 0; k < a.nv; ++k) {
        const int v = nc->out_of[k];
        a.scale[k] = nc->scale[k];
        a.fill_only[k] = nc->fill_only[k];
        if (a.fill_only[k]) { a.src[k] = mcf::RingView{p->d_ring, p->N, 0, 0, 0}; continue; }   // never read
        if (p->var_slot[v] < 0) return fail(MCF_ERR_ARG, "mcf_nc_write_plan: a variable of the file was not requested in out[]");
        a.src[k] = ring_view(p, slot, v);
    }
    HIP_TRY(hipSetDevice(p->device));
    const int64_t rb = nc->f->rec_bytes;
    int64_t piece = std::min<int64_t>(65535 / a.nv, std::max<int64_t>(1, ((int64_t)256 << 20) / rb));
    piece = std::min(piece, std::max<int64_t>(nsteps, 1));
    if (p->pack_elems < piece * (rb / 4)) {
        int rc;
        void* q;
        if ((rc = dalloc(p, &q, piece * rb))) return rc;
        p->d_pack = (int32_t*)q;
        p->pack_elems = piece * (rb / 4);
    }
    a.dst = p->d_pack;
    static const bool no_pipe = getenv("MCF_NO_HOSTPIPE") != nullptr;
    std::future<std::string> pending;            // the previous piece going to disk while this one is packed and copied
    float kms = 0;
    int rc = MCF_OK;
    std::string werr;
    for (int64_t s0 = 0, i = 0; s0 < nsteps && rc == MCF_OK; s0 += piece, ++i) {
        const int64_t n = std::min(piece, nsteps - s0);
        mcf::PackNcArgs b = a;
        b.step0 = slot_step0 + s0;
        hipError_t e = hipSuccess;
        if (kernel_ms) e = hipEventRecord(p->ev0, p->stream);
        mcf::launch_pack_nc(b, n, p->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && kernel_ms) e = hipEventRecord(p->ev1, p->stream);
        // stage[i & 1] was handed to the writer two pieces ago: that write has been joined (below) before piece i-1 began
        const size_t bytes = (size_t)(n * rb);
        uint8_t* st = nc->staging((int)(i & 1), bytes);
        if (e == hipSuccess) {
            if (bytes >= ((size_t)64 << 20) && !no_pipe && ensure_pipe(p)) {
                e = hipEventRecord(p->ev_pipe, p->stream);
                if (e == hipSuccess) e = p->pipe->copy(st, p->d_pack, bytes, p->ev_pipe);
            } else {
                e = hipMemcpyAsync(st, p->d_pack, bytes, hipMemcpyDeviceToHost, p->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
            }
        }
        if (e == hipSuccess && kernel_ms) {
            float ms = 0;
            e = hipEventElapsedTime(&ms, p->ev0, p->ev1);
            kms += ms;
        }
        if (pending.valid()) werr = pending.get();
        if (e != hipSuccess) { rc = fail(MCF_ERR_HIP, std::string("mcf_nc_write_plan: ") + hipGetErrorString(e)); break; }
        if (!werr.empty()) break;
        mcf::NcFile* f = nc->f.get();
        uint8_t* data = st;
        const int64_t fs = file_step0 + s0;
        pending = std::async(std::launch::async, [f, fs, n, data] { return f->write_records(fs, n, data); });
    }
    if (pending.valid()) {
        const std::string e2 = pending.get();
        if (werr.empty()) werr = e2;
    }
    if (rc != MCF_OK) return rc;
    if (!werr.empty()) return fail(MCF_ERR_ARG, "mcf_nc_write_plan: " + werr);
    if (kernel_ms) *kernel_ms = kms;
    return MCF_OK;
}

int mcf_plan_slot_ptr(mcf_plan* p, int32_t slot, int32_t var, void** dev_ptr) {
    if (!p || !dev_ptr) return fail(MCF_ERR_ARG, "null argument");
    if (slot < 0 || slot >= p->ring_slots || var < 0 || var >= MCF_NOUT || p->var_slot[var] < 0)
        return fail(MCF_ERR_ARG, "bad slot/var");
    *dev_ptr = const_cast<double*>(ring_view(p, slot, var).base);
    return MCF_OK;
}

int mcf_plan_ring_layout(mcf_plan* p, mcf_ring_layout* out) {
    if (!p || !out) return fail(MCF_ERR_ARG, "null argument");
    memset(out, 0, sizeof *out);
    out->tiled = p->tiled ? 1 : 0;
    out->cells = p->N;
    out->slot_days = p->ring_days;
    if (p->tiled) {
        out->cells_per_tile = p->cpb;
        out->block_doubles = mcf::ring_block_doubles(p->cpb);
        out->tile_stride = p->ring_tile_stride;
        out->day_stride = p->ring_day_stride;
    }
    return MCF_OK;
}

int64_t mcf_ring_index(const mcf_ring_layout* l, int64_t cell, int64_t step) {
    if (!l || cell < 0 || cell >= l->cells || step < 0 || step >= (int64_t)l->slot_days * 24) return -1;
    mcf::RingView v{nullptr, l->cells, l->tile_stride, l->day_stride, l->tiled ? l->cells_per_tile : 0};
    return v.index(cell, step);
}

int mcf_plan_timer_start(mcf_plan* p) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    return MCF_OK;
}

int mcf_plan_timer_stop(mcf_plan* p, float* ms) {
    if (!p || !ms) return fail(MCF_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return MCF_OK;
}

int mcf_plan_kernel_timing(mcf_plan* p, int32_t enable) {
    if (!p) return fail(MCF_ERR_ARG, "null plan");
    p->ktiming = enable != 0;
    if (!enable) return MCF_OK;
    for (auto& e : p->kev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    p->kev.clear();
    p->ktotal_ms = 0;
    p->klaunches = 0;
    return MCF_OK;
}

int mcf_plan_kernel_stats(mcf_plan* p, double* total_ms, int64_t* launches) {
    if (!p || !total_ms || !launches) return fail(MCF_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (auto& e : p->kev) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        p->ktotal_ms += ms;
        p->klaunches += 1;
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    p->kev.clear();
    *total_ms = p->ktotal_ms;
    *launches = p->klaunches;
    return MCF_OK;
}

int mcf_selftest_math(int32_t kind, const double* x, const double* y, double* out, int64_t n, int32_t device) {
    if (!x || !out || n < 0) return fail(MCF_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    struct G { double *&a, *&b, *&c; ~G() { hipFree(a); hipFree(b); hipFree(c); } } g{dx, dy, dout};
    size_t nb = (size_t)std::max<int64_t>(n, 1) * 8;
    HIP_TRY(hipMalloc((void**)&dx, nb));
    HIP_TRY(hipMalloc((void**)&dout, nb));
    HIP_TRY(hipMemcpy(dx, x, (size_t)n * 8, hipMemcpyHostToDevice));
    if (y) {
        HIP_TRY(hipMalloc((void**)&dy, nb));
        HIP_TRY(hipMemcpy(dy, y, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    mcf::launch_selftest_math(kind, dx, dy, dout, n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}

static const int32_t kBioSt[14] = {0, 24, 48, 72, 96, 120, 144, 168, 192, 216, 240, 264, 288, 312};
static const int32_t kBioEd[14] = {23, 47, 71, 95, 119, 143, 167, 191, 215, 239, 263, 287, 311, 335};
// the requested matrices of bio [19][N] into the caller's arrays (out_pitch > rows: a row block's rows of taller matrices)
static int bioclim_download(mcf_plan* p, const mcf_grid_inputs* in, const mcf_bioclim_sel* sel, mcf_bioclim_out* out, const double* bio,
                            int64_t out_pitch) {
    const int64_t N = in->rows * in->cols;
    for (int v = 0; v < MCF_NBIO; ++v)
        if (sel->out[v]) {
            if (!out->bio[v]) return fail(MCF_ERR_ARG, "requested bioclim variable has a null buffer");
            if (out_pitch > in->rows)
                HIP_TRY(hipMemcpy2DAsync(out->bio[v], (size_t)out_pitch * 8, bio + (int64_t)v * N, (size_t)in->rows * 8,
                                         (size_t)in->rows * 8, (size_t)in->cols, hipMemcpyDeviceToHost, p->stream));
            else
                HIP_TRY(hipMemcpyAsync(out->bio[v], bio + (int64_t)v * N, (size_t)N * 8, hipMemcpyDeviceToHost, p->stream));
        }
    return mcf_plan_sync(p);
}
// twi_mean / out_pitch: a row block of a taller raster (the bioclim `_multi` entries): the raster-wide twi mean to install, and
// the rows of the caller's [rows_total, cols] matrices the block's rows are written into in place
static int run_bioclim(const mcf_grid_inputs* in_caller, const mcf_options* opt_in, const mcf_bioclim_sel* sel,
                       mcf_bioclim_out* out, int want_af, int layered = 0, const double* twi_mean = nullptr, int64_t out_pitch = 0) {
    if (!sel || !out || !in_caller) return fail(MCF_ERR_ARG, "null bioclim argument");
    // runbioclim3Cpp / 4Cpp (cpp:3620-3658 / 3660-3700): vegetation arrays [rows, cols, >= 14] and a fixed dfsel of
    // fourteen one-day layers — the twelve monthly days, the hottest and the coldest day; later steps (the quarter
    // days) belong to no layer and stay NA, as in the reference
    mcf_grid_inputs in_l;
    if (layered) {
        in_l = *in_caller;
        in_l.veg_layers = 14;
        in_l.lyr_st = kBioSt;
        in_l.lyr_ed = kBioEd;
    }
    const mcf_grid_inputs* in = layered ? &in_l : in_caller;
    int rc = check_inputs(in, opt_in);
    if (rc) return rc;
    if ((in->array_forcing != 0) != (want_af != 0)) return fail(MCF_ERR_ARG, "forcing geometry does not match the entry point");
    const int64_t T = in->tsteps, N = in->rows * in->cols;
    if (T < 336 || T % 24 != 0) return fail(MCF_ERR_ARG, "runbioclim needs >= 336 hourly steps in whole days");
    const int32_t* q[4] = {sel->wetq, sel->dryq, sel->hotq, sel->colq};
    const int32_t nq[4] = {sel->nwet, sel->ndry, sel->nhot, sel->ncol};
    for (int i = 0; i < 4; ++i) {
        if (nq[i] < 0 || (nq[i] > 0 && !q[i])) return fail(MCF_ERR_ARG, "bad quarter index vector");
        for (int j = 0; j < nq[i]; ++j)
            if (q[i][j] < 0 || q[i][j] >= T) return fail(MCF_ERR_ARG, "quarter index outside the time series");
    }
    mcf_options opt = *opt_in;
    opt.complete = 1;                                                   // cpp:3576: complete = true
    for (int v = 0; v < MCF_NOUT; ++v) opt.out[v] = 0;
    const int tvar = sel->air ? MCF_OUT_TZ : MCF_OUT_TLEAF;              // cpp:3569-3575
    opt.out[tvar] = 1;
    opt.out[MCF_OUT_SOILM] = 1;
    if ((rc = ensure_device(opt.device))) return rc;
    const int ndays = (int)(T / 24);
    // Streamed (round 5; vector forcing above ground, quarter lists in ascending order — what runbioclim passes): the solver
    // runs in day chunks into a ring of a few GB and k_bioclim_acc folds each chunk into 29 doubles of running state per cell;
    // nothing of size cells x steps is allocated (the whole-series form below needs 2 x 8 B x cells x steps: 167 GB for a 4096^2
    // raster, most of the call's time).  Same accumulation order, same operands: the matrices are bit for bit the whole-series
    // form's (MCF_BIOCLIM_WHOLE=1 selects that one: the A/B of tests/test_bioclim_gpu.py).
    bool streamed = !in->array_forcing && !(opt.reqhgt < 0.0) && getenv("MCF_BIOCLIM_WHOLE") == nullptr;
    for (int i = 0; streamed && i < 4; ++i)
        for (int j = 1; j < nq[i]; ++j)
            if (q[i][j] < q[i][j - 1]) streamed = false;
    int chunk_days = ndays;
    if (streamed) {
        const int cpb = opt.cells_per_block ? opt.cells_per_block : 21;
        const double day_bytes = 2.0 * 8.0 * (double)((N + cpb - 1) / cpb) * (double)mcf::ring_block_doubles(cpb);
        const char* e = getenv("MCF_BIOCLIM_RING_GB");
        const double budget = (e && atof(e) > 0.0 ? atof(e) : 14.0) * 1e9;     // (4096^2: 0.80 s with 14 GB = two-day chunks, 0.85 s with 8, 2.3 s with 27 — the allocation; profiles/r05_sink_rates.txt)
        chunk_days = (int)std::max(1.0, std::min((double)ndays, floor(budget / day_bytes)));
    }
    mcf_plan* p = nullptr;
    if ((rc = mcf_plan_create(in, &opt, chunk_days, 1, &p))) return rc;
    struct Guard { mcf_plan* p; ~Guard() { mcf_plan_destroy(p); } } guard{p};
    if (twi_mean && (rc = mcf_plan_set_twi_mean(p, *twi_mean))) return rc;
    if (streamed && p->tiled) {
        void* tmp = nullptr;
        mcf::BioAccArgs acc{};
        acc.N = N;
        acc.tz = ring_view(p, 0, tvar); acc.soilm = ring_view(p, 0, MCF_OUT_SOILM);
        for (int i = 0; i < 4; ++i) {
            if ((rc = dalloc(p, &tmp, (int64_t)std::max(nq[i], 1) * 4))) return rc;
            if (nq[i] > 0) HIP_TRY(hipMemcpyAsync(tmp, q[i], (size_t)nq[i] * 4, hipMemcpyHostToDevice, p->stream));
            acc.q[i] = (const int32_t*)tmp;
        }
        if ((rc = dalloc(p, &tmp, (int64_t)mcf::kBioStateRows * N * 8))) return rc;
        acc.state = (double*)tmp;
        for (int d0 = 0; d0 < ndays; d0 += p->ring_days) {
            const int nd = std::min(p->ring_days, ndays - d0);
            if ((rc = mcf_plan_run_days(p, d0, nd, 0))) return rc;
            acc.day0 = d0; acc.ndays = nd;
            for (int i = 0; i < 4; ++i) {
                acc.qlo[i] = (int32_t)(std::lower_bound(q[i], q[i] + nq[i], d0 * 24) - q[i]);
                acc.qhi[i] = (int32_t)(std::lower_bound(q[i], q[i] + nq[i], (d0 + nd) * 24) - q[i]);
            }
            mcf::launch_bioclim_acc(acc, p->stream);
            HIP_TRY(hipGetLastError());
        }
        mcf::BioFinArgs fin{};
        fin.N = N; fin.tsteps = (int)T; fin.state = acc.state;
        fin.cellc = p->d_cellc; fin.ntiles_total = p->ntiles; fin.cpb = p->cpb; fin.daylayer = p->d_daylayer; fin.tt = p->d_tt;
        if ((rc = dalloc(p, &tmp, (int64_t)MCF_NBIO * N * 8))) return rc;
        fin.bio = (double*)tmp;
        mcf::launch_bioclim_fin(fin, p->stream);
        HIP_TRY(hipGetLastError());
        return bioclim_download(p, in, sel, out, fin.bio, out_pitch);
    }
    if (streamed) return fail(MCF_ERR_STATE, "bioclim: the streamed sink expects the tiled ring");
    if (in->array_forcing && (rc = mcf_plan_upload_forcing_days(p, in, 0, ndays, 0))) return rc;
    if ((rc = mcf_plan_run_days(p, 0, ndays, 0))) return rc;
    if (p->bg && (rc = mcf_plan_belowground(p))) return rc;
    void* tmp = nullptr;
    mcf::BioclimArgs b{};
    b.N = N; b.tsteps = (int)T;
    b.tz = ring_view(p, 0, tvar); b.soilm = ring_view(p, 0, MCF_OUT_SOILM);
    const int32_t** dq[4] = {&b.wetq, &b.dryq, &b.hotq, &b.colq};
    b.nwet = nq[0]; b.ndry = nq[1]; b.nhot = nq[2]; b.ncol = nq[3];
    for (int i = 0; i < 4; ++i) {
        if ((rc = dalloc(p, &tmp, (int64_t)std::max(nq[i], 1) * 4))) return rc;
        if (nq[i] > 0) HIP_TRY(hipMemcpyAsync(tmp, q[i], (size_t)nq[i] * 4, hipMemcpyHostToDevice, p->stream));
        *dq[i] = (const int32_t*)tmp;
    }
    if ((rc = dalloc(p, &tmp, (int64_t)MCF_NBIO * N * 8))) return rc;
    b.bio = (double*)tmp;
    mcf::launch_bioclim(b, p->stream);
    HIP_TRY(hipGetLastError());
    return bioclim_download(p, in, sel, out, b.bio, out_pitch);
}
int mcf_runbioclim1(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, mcf_bioclim_out* out) {
    return run_bioclim(in, opt, sel, out, 0);
}
int mcf_runbioclim3(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, mcf_bioclim_out* out) {
    return run_bioclim(in, opt, sel, out, 0, 1);
}
int mcf_runbioclim4(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, mcf_bioclim_out* out) {
    return run_bioclim(in, opt, sel, out, 1, 1);
}
int mcf_runbioclim2(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, mcf_bioclim_out* out) {
    return run_bioclim(in, opt, sel, out, 1);
}

int64_t mcf_plan_valid_cells(const mcf_plan* p) { return p ? p->valid_cells : 0; }
int64_t mcf_plan_bytes(const mcf_plan* p) { return p ? p->bytes : 0; }

// ---- one-shot host-to-host solve ---------------------------------------------------------
// twi_mean: null, or the raster-wide mean of log(twi)/tfact to install (a row block of a larger raster, run_multi)
// sharers: host threads that solve their blocks on this device at the same time (one-process multi-device route with a device
// listed more than once): each sizes its ring from its share of the free HBM
static int run_oneshot(const mcf_grid_inputs* in, const mcf_options* opt, mcf_outputs* out, int want_af,
                       const double* twi_mean = nullptr, int sharers = 1) {
    int rc = check_inputs(in, opt);
    if (rc) return rc;
    if (!out) return fail(MCF_ERR_ARG, "null outputs");
    if ((in->array_forcing != 0) != (want_af != 0))
        return fail(MCF_ERR_ARG, want_af ? "mcf_runmicro2 needs array_forcing = 1" : "mcf_runmicro1 needs array_forcing = 0");
    for (int v = 0; v < MCF_NOUT; ++v)
        if (opt->out[v] && !out->var[v]) return fail(MCF_ERR_ARG, "requested output has a null buffer");
    rc = ensure_device(opt->device);
    if (rc) return rc;
    const int64_t N = in->rows * in->cols, T = in->tsteps;
    // host arrays may be row blocks of a taller raster (row_pitch): a time step of an output is then pitch x cols values on
    const int64_t pitch = in->row_pitch > 0 ? in->row_pitch : in->rows, HS = pitch * in->cols;
    const int ndays = (int)(T / 24);
    int nvars = 0;
    for (int v = 0; v < MCF_NOUT; ++v) nvars += opt->out[v] ? 1 : 0;
    const bool bg = opt->reqhgt < 0.0;
    // ---- chunk size from free HBM
    int chunk = opt->days_per_chunk;
    if (bg) {
        chunk = std::max(ndays, 1);
    } else if (chunk <= 0) {
        size_t fr = 0, tot = 0;
        HIP_TRY(hipMemGetInfo(&fr, &tot));
        double per_day = (double)N * 24 * 8 * (nvars + (in->array_forcing ? 15 : 0));
        double budget = 0.6 * (double)fr / std::max(sharers, 1) - (double)N * 8 * 200;
        chunk = (int)std::max(1.0, std::min((double)std::max(ndays, 1), budget / std::max(per_day, 1.0)));
        chunk = std::min(chunk, 64);
    }
    chunk = std::max(1, std::min(chunk, std::max(ndays, 1)));
    const bool timing = getenv("MCF_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t_solve = 0, t_fetch = 0;
    mcf_plan* p = nullptr;
    rc = mcf_plan_create(in, opt, chunk, 1, &p);
    if (rc) return rc;
    struct Guard { mcf_plan* p; ~Guard() { mcf_plan_destroy(p); } } guard{p};
    if (twi_mean && (rc = mcf_plan_set_twi_mean(p, *twi_mean))) return rc;
    double t_create = now() - t0;
    for (int d0 = 0; d0 < ndays; d0 += chunk) {
        int nd = std::min(chunk, ndays - d0);
        if (in->array_forcing && (rc = mcf_plan_upload_forcing_days(p, in, d0, nd, 0))) return rc;
        double ta = now();
        if ((rc = mcf_plan_run_days(p, d0, nd, 0))) return rc;
        if (timing) { mcf_plan_sync(p); t_solve += now() - ta; ta = now(); }
        if (!bg) {
            for (int v = 0; v < MCF_NOUT; ++v)
                if (opt->out[v])
                    if ((rc = mcf_plan_fetch_pitched(p, 0, v, 0, (int64_t)nd * 24, out->var[v] + HS * (int64_t)d0 * 24, pitch)))
                        return rc;
        }
        if (timing) t_fetch += now() - ta;
    }
    if (timing)
        fprintf(stderr, "[mcf] one-shot: chunk %d days, plan create %.3f s, solve %.3f s, fetch %.3f s\n", chunk,
                t_create, t_solve, t_fetch);
    if (bg && ndays > 0) {
        for (int v = 0; v < MCF_NOUT; ++v)
            if (opt->out[v] && v != MCF_OUT_TZ)
                if ((rc = mcf_plan_fetch_pitched(p, 0, v, 0, (int64_t)ndays * 24, out->var[v], pitch))) return rc;
    }
    // steps past the last whole day are never computed by the reference and stay NA (cpp:2116)
    const double na = na_real_host();
    for (int v = 0; v < MCF_NOUT; ++v)
        if (opt->out[v])
            for (int64_t k = (int64_t)ndays * 24; k < T; ++k)
                for (int64_t j = 0; j < in->cols; ++j)
                    for (int64_t i = 0; i < in->rows; ++i) out->var[v][i + pitch * j + HS * k] = na;
    if (bg && opt->out[MCF_OUT_TZ] && T > 0) {
        // Tbelowgroundv runs over all tsteps (cpp:2314-2319)
        if ((rc = mcf_plan_belowground(p))) return rc;
        if ((rc = mcf_plan_fetch_pitched(p, 0, MCF_OUT_TZ, 0, T, out->var[MCF_OUT_TZ], pitch))) return rc;
    }
    return mcf_plan_sync(p);
}

// ---- one process, several devices --------------------------------------------------------------------------------------
// The raster is cut into contiguous row blocks holding about the same number of valid cells (boundaries on multiples of 10
// rows, as microclimf_amd/distributed.py balanced_row_blocks); block b is solved on devices[b % n_devices] by that device's
// host thread with its own plan and stream.  Cells are independent inside the solver; its one global reduction — the mean
// of log(twi)/tfact over the raster (src/microclimfCpp.cpp:993-1004) — is taken over the WHOLE raster first, with the
// kernels a single-device plan uses, and installed in every block's plan: the result is bit for bit the single-device one.
// Host arrays are not copied: a block reads and writes its rows in place through the row pitch.
static std::vector<std::pair<int64_t, int64_t>> row_blocks(const mcf_grid_inputs* in, int nb) {
    const int64_t R = in->rows, pitch = in->row_pitch > 0 ? in->row_pitch : in->rows;
    const int64_t mult = R / 10 >= nb ? 10 : 1, ng = (R + mult - 1) / mult;
    std::vector<double> cum((size_t)ng + 1, 0.0);
    for (int64_t j = 0; j < in->cols; ++j)
        for (int64_t i = 0; i < R; ++i)
            if (!std::isnan(in->vegp.hgt[i + pitch * j])) cum[(size_t)(i / mult) + 1] += 1.0;
    for (int64_t g = 0; g < ng; ++g) cum[(size_t)g + 1] += cum[(size_t)g];
    const double total = cum[(size_t)ng];
    std::vector<int64_t> cuts{0};
    for (int r = 1; r < nb; ++r) {
        int64_t g = ng * r / nb;
        if (total > 0) {
            const double want = total * r / nb;
            g = std::lower_bound(cum.begin(), cum.end(), want) - cum.begin();
            if (g > 0 && std::fabs(cum[(size_t)g - 1] - want) <= std::fabs(cum[(size_t)std::min(g, ng)] - want)) --g;
        }
        g = std::min(std::max(g, cuts.back() + 1), ng - (nb - r));
        cuts.push_back(g);
    }
    cuts.push_back(ng);
    std::vector<std::pair<int64_t, int64_t>> out;
    for (int r = 0; r < nb; ++r) {
        const int64_t r0 = cuts[(size_t)r] * mult, r1 = std::min(cuts[(size_t)r + 1] * mult, R);
        out.emplace_back(r0, r1 - r0);
    }
    return out;
}

extern "C++" {
// block_fn(sub, o, r0, twi_mean, sharers): one row block — `sub` is the caller's inputs narrowed to the block's rows (same arrays, offset,
// read through the row pitch), `o` the options with the block's device, r0 the block's first row
template <class F>
static int for_row_blocks(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* mu, F&& block_fn) {
    int rc = check_inputs(in, opt);
    if (rc) return rc;
    if (!mu) return fail(MCF_ERR_ARG, "null argument");
    // what row_blocks and the whole-raster twi mean read, before any plan has validated the inputs
    if (!in->vegp.hgt) return fail(MCF_ERR_ARG, "missing input array: vegp$hgt");
    if (!in->soilc.twi) return fail(MCF_ERR_ARG, "missing input array: twi");
    int ndev_avail = 0;
    if (hipGetDeviceCount(&ndev_avail) != hipSuccess || ndev_avail <= 0)
        return fail(MCF_ERR_NO_DEVICE, "no HIP device available (libmcfhip has no CPU fallback)");
    // the calling thread's current device is put back on every way out
    int caller_dev = 0;
    const bool have_caller_dev = hipGetDevice(&caller_dev) == hipSuccess;
    struct RestoreDev { bool on; int d; ~RestoreDev() { if (on) (void)hipSetDevice(d); } } restore_dev{have_caller_dev, caller_dev};
    std::vector<int> devs;
    if (mu->n_devices <= 0) for (int d = 0; d < ndev_avail; ++d) devs.push_back(d);      // every visible device
    else {
        if (!mu->devices) return fail(MCF_ERR_ARG, "n_devices > 0 with a null device list");
        for (int i = 0; i < mu->n_devices; ++i) {
            if (mu->devices[i] < 0 || mu->devices[i] >= ndev_avail) return fail(MCF_ERR_ARG, "device ordinal out of range");
            devs.push_back(mu->devices[i]);
        }
    }
    int nb = mu->n_blocks > 0 ? mu->n_blocks : (int)devs.size();
    nb = (int)std::min<int64_t>(nb, in->rows);
    const int64_t pitch = in->row_pitch > 0 ? in->row_pitch : in->rows;
    // ---- the one global reduction, over the whole raster, on the first device (the kernels of mcf_plan_create)
    double twi_mean;
    {
        HIP_TRY(hipSetDevice(devs[0]));
        const int64_t N = in->rows * in->cols;
        if (!in->soilc.twi) return fail(MCF_ERR_ARG, "missing input array: twi");
        double *d_twi = nullptr, *d_ws = nullptr;
        HIP_TRY(hipMalloc((void**)&d_twi, (size_t)N * 8));
        struct G { double *&a, *&b; ~G() { (void)hipFree(a); (void)hipFree(b); } } g{d_twi, d_ws};
        HIP_TRY(hipMalloc((void**)&d_ws, (size_t)mcf::twi_scratch_doubles() * 8));
        HIP_TRY(hipMemcpy2D(d_twi, (size_t)in->rows * 8, in->soilc.twi, (size_t)pitch * 8, (size_t)in->rows * 8, (size_t)in->cols,
                            hipMemcpyHostToDevice));
        mcf::launch_twi_partial(d_twi, N, opt->tfact, d_ws, nullptr);
        double h2[2];
        HIP_TRY(hipMemcpy(h2, d_ws, 16, hipMemcpyDeviceToHost));
        twi_mean = h2[0] / h2[1];
    }
    const auto blocks = row_blocks(in, nb);
    std::vector<int> rcs(devs.size(), MCF_OK);
    std::vector<std::string> errs(devs.size());
    std::vector<std::thread> threads;
    for (size_t t = 0; t < devs.size(); ++t) {
        threads.emplace_back([&, t] {
            // (an exception must not leave a worker thread: std::terminate would take the host R / Python process down)
            try {
            for (int b = (int)t; b < nb; b += (int)devs.size()) {
                const int64_t r0 = blocks[(size_t)b].first, nr = blocks[(size_t)b].second;
                if (nr <= 0) continue;
                mcf_grid_inputs sub = *in;
                sub.rows = nr;
                sub.row_pitch = pitch;
                auto off = [&](const double*& q) { if (q) q += r0; };
                off(sub.vegp.hgt); off(sub.vegp.pai); off(sub.vegp.x); off(sub.vegp.gsmax); off(sub.vegp.leafr); off(sub.vegp.leaft);
                off(sub.vegp.clump); off(sub.vegp.leafd); off(sub.vegp.paia); off(sub.vegp.leafden);
                off(sub.soilc.Smin); off(sub.soilc.Smax); off(sub.soilc.gref); off(sub.soilc.soilb); off(sub.soilc.Psie);
                off(sub.soilc.Vq); off(sub.soilc.Vm); off(sub.soilc.Mc); off(sub.soilc.rho); off(sub.soilc.slope);
                off(sub.soilc.aspect); off(sub.soilc.twi); off(sub.soilc.svfa); off(sub.soilc.wsa); off(sub.soilc.hor);
                off(sub.lats); off(sub.lons); off(sub.coarse_rowpos); off(sub.fine_dtm);
                if (in->array_forcing == 1) {
                    off(sub.clim.tc); off(sub.clim.es); off(sub.clim.ea); off(sub.clim.tdew); off(sub.clim.pk); off(sub.clim.swdown);
                    off(sub.clim.difrad); off(sub.clim.lwdown); off(sub.clim.windspeed);
                    off(sub.pointm.soilm); off(sub.pointm.G); off(sub.pointm.umu); off(sub.pointm.kp); off(sub.pointm.muGp);
                    off(sub.pointm.dtrp); off(sub.pointm.Tg); off(sub.pointm.Tbp);
                }
                mcf_options o = *opt;
                o.device = devs[t];
                int sharers = 0;
                for (int d : devs) sharers += d == devs[t];
                const int rcb = block_fn(sub, o, r0, &twi_mean, sharers);
                if (rcb != MCF_OK) { rcs[t] = rcb; errs[t] = g_err; return; }
            }
            } catch (const std::exception& e) {
                rcs[t] = MCF_ERR_NOMEM; errs[t] = std::string("row-block worker: ") + e.what();
            }
        });
    }
    for (auto& th : threads) th.join();
    for (size_t t = 0; t < devs.size(); ++t)
        if (rcs[t] != MCF_OK) return fail(rcs[t], errs[t]);
    return MCF_OK;
}

}  // extern "C++"
static int run_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* mu, mcf_outputs* out, int want_af) {
    if (!out) return fail(MCF_ERR_ARG, "null argument");
    return for_row_blocks(in, opt, mu, [&](const mcf_grid_inputs& sub, const mcf_options& o, int64_t r0, const double* twi_mean, int sharers) {
        mcf_outputs so = *out;
        for (int v = 0; v < MCF_NOUT; ++v) if (so.var[v]) so.var[v] += r0;
        return run_oneshot(&sub, &o, &so, want_af, twi_mean, sharers);
    });
}
// the fused bioclim sink over row blocks: a block's nineteen [rows, cols] matrices go into its rows of the caller's
static int run_bioclim_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, const mcf_multi* mu,
                             mcf_bioclim_out* out, int want_af, int layered) {
    if (!out || !sel || !in) return fail(MCF_ERR_ARG, "null bioclim argument");
    const int64_t pitch = in->row_pitch > 0 ? in->row_pitch : in->rows;
    mcf_grid_inputs in_l = *in;
    if (layered) { in_l.veg_layers = 14; in_l.lyr_st = kBioSt; in_l.lyr_ed = kBioEd; }     // as run_bioclim: the fixed dfsel
    return for_row_blocks(&in_l, opt, mu, [&](const mcf_grid_inputs& sub, const mcf_options& o, int64_t r0, const double* twi_mean, int) {
        mcf_bioclim_out bo = *out;
        for (int v = 0; v < MCF_NBIO; ++v) if (bo.bio[v]) bo.bio[v] += r0;
        return run_bioclim(&sub, &o, sel, &bo, want_af, layered, twi_mean, pitch);
    });
}
int mcf_runbioclim1_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, const mcf_multi* mu, mcf_bioclim_out* out) {
    return run_bioclim_multi(in, opt, sel, mu, out, 0, 0);
}
int mcf_runbioclim2_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, const mcf_multi* mu, mcf_bioclim_out* out) {
    return run_bioclim_multi(in, opt, sel, mu, out, 1, 0);
}
int mcf_runbioclim3_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, const mcf_multi* mu, mcf_bioclim_out* out) {
    return run_bioclim_multi(in, opt, sel, mu, out, 0, 1);
}
int mcf_runbioclim4_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_bioclim_sel* sel, const mcf_multi* mu, mcf_bioclim_out* out) {
    return run_bioclim_multi(in, opt, sel, mu, out, 1, 1);
}

int mcf_runmicro1(const mcf_grid_inputs* in, const mcf_options* opt, mcf_outputs* out) {
    return run_oneshot(in, opt, out, 0);
}
int mcf_runmicro1_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* multi, mcf_outputs* out) {
    return run_multi(in, opt, multi, out, 0);
}
int mcf_runmicro2_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* multi, mcf_outputs* out) {
    return run_multi(in, opt, multi, out, 1);
}
int mcf_runmicro2(const mcf_grid_inputs* in, const mcf_options* opt, mcf_outputs* out) {
    return run_oneshot(in, opt, out, 1);
}
int mcf_runmicro3(const mcf_grid_inputs* in, const mcf_options* opt, mcf_outputs* out) {
    if (in && in->veg_layers < 1) return fail(MCF_ERR_ARG, "mcf_runmicro3 needs veg_layers >= 1 and dfsel");
    return run_oneshot(in, opt, out, 0);
}
int mcf_runmicro4(const mcf_grid_inputs* in, const mcf_options* opt, mcf_outputs* out) {
    if (in && in->veg_layers < 1) return fail(MCF_ERR_ARG, "mcf_runmicro4 needs veg_layers >= 1 and dfsel");
    return run_oneshot(in, opt, out, 1);
}
// time-varying vegetation over row blocks: the layered arrays [rows, cols, layers] are read through the same row pitch
int mcf_runmicro3_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* multi, mcf_outputs* out) {
    if (in && in->veg_layers < 1) return fail(MCF_ERR_ARG, "mcf_runmicro3_multi needs veg_layers >= 1 and dfsel");
    return run_multi(in, opt, multi, out, 0);
}
int mcf_runmicro4_multi(const mcf_grid_inputs* in, const mcf_options* opt, const mcf_multi* multi, mcf_outputs* out) {
    if (in && in->veg_layers < 1) return fail(MCF_ERR_ARG, "mcf_runmicro4_multi needs veg_layers >= 1 and dfsel");
    return run_multi(in, opt, multi, out, 1);
}

}  // extern "C"
