// mcf_kernels.h — kernel argument blocks and launchers (host-visible part).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcf {

struct Globals {
    double reqhgt;   // as given
    double reqhgt2;  // max(reqhgt, 1e-5)                       cpp:2246-2247
    double zref;
    double dTmx;     // -0.6273*mxtc + 49.79 (vector forcing)   cpp:1236
    double hf0p;     // |Hf|^0.2 of mincondCpp for gs = 999.99    cpp:1321-1328
    double hf500p;   // |Hf|^0.2 of mincondCpp for rs = 500       cpp:1321-1328
    int shadowmask;  // 1: runmicro1Cpp, 0: runmicro2Cpp         cpp:2218 vs 2499
};

// ---- the output ring as its consumers see it -------------------------------------------------------------------------
// Doubles per tile-day block of the tiled ring = lanes of the solver's workgroup for `cpb` cells per tile.
#define RING_BLOCK(cpb) ((((cpb) * 24 + 63) / 64) * 64)
__host__ __device__ inline int ring_block_doubles(int cpb) { return RING_BLOCK(cpb); }
// Place of (cell of the tile, hour of the day) inside a block: the lane that computes it in k_solve.  21-cell tiles:
// wave w = hour / 3 holds [3 hours x cells 0..15 | 3 hours x cells 16..20 | one padding double]; others hour-major.
__host__ __device__ inline int ring_pos(int cpb, int cell, int hour) {
    if (cpb == 21) {
        const int w = hour / 3, hh = hour - 3 * w;
        return 64 * w + (cell < 16 ? 16 * hh + cell : 48 + 5 * hh + (cell - 16));
    }
    return hour * cpb + cell;
}
// ... and back: the (cell, hour) of place `pos` of a block; false for a padding place
__host__ __device__ inline bool ring_unpos(int cpb, int pos, int* cell, int* hour) {
    if (cpb == 21) {
        const int w = pos >> 6, l = pos & 63;
        if (l < 48) { *cell = l & 15; *hour = 3 * w + (l >> 4); return true; }
        const int j = l - 48;
        *cell = 16 + j % 5; *hour = 3 * w + j / 5;
        return j < 15;
    }
    *cell = pos % cpb; *hour = pos / cpb;
    return pos < 24 * cpb;
}
// One variable of one ring slot: value of cell c (0-based, column-major) at step k of the slot.
//   cpb == 0  linear [step][N] (reqhgt < 0, staging buffers)
//   cpb  > 0  tiled (k_solve's layout): tile = c / cpb, block of day k / 24 at tile * tile_stride + day * day_stride
struct RingView {
    const double* base;
    int64_t N;
    int64_t tile_stride, day_stride;
    int32_t cpb;
    __host__ __device__ inline int64_t index(int64_t c, int64_t k) const {
        if (cpb == 0) return c + N * k;
        const uint32_t cc = (uint32_t)c, t = cc / (uint32_t)cpb, cell = cc - t * (uint32_t)cpb;   // N < 2^31 (checked by the host)
        const uint32_t kk = (uint32_t)k, d = kk / 24u, h = kk - 24u * d;
        return (int64_t)t * tile_stride + (int64_t)d * day_stride + ring_pos(cpb, (int)cell, (int)h);
    }
    __device__ inline double at(int64_t c, int64_t k) const { return base[index(c, k)]; }
};

struct CellSetupArgs {
    int64_t N;
    // vegp
    const double *hgt, *pai, *x, *gsmax, *leafr, *leaft, *clump, *leafd, *paia, *leafden;
    const double* hgt0;  // layer-0 height: the NA test of the cell (cpp:2182 / 2759)
    // soilc
    const double *Smin, *Smax, *gref, *soilb, *Psie, *Vq, *Vm, *Mc, *rho, *slope, *aspect, *twi, *svfa;
    const double *hor, *wsa;    // [24][N], [8][N]: only their finiteness is looked at here (FL_REGULAR)
    const double *lats, *lons;  // array forcing, else null
    const double *crowpos, *ccolpos;  // coarse array forcing: [rows], [cols]; else null
    const double *elevd, *pkfac;      // coarse array forcing with altitude correction: [N]; else null
    int64_t rows;
    double lat, lon;
    double tfact, twi_mean;
    Globals g;
    // The per-cell constant table is TILE-MAJOR: one image per (layer, tile of cpb cells), [CF_COUNT + 32 rows][cpb] doubles
    // (the 24 horizon and 8 wind-shelter values follow the CF_ rows), images tile_image_doubles(cpb) apart — what the solver
    // copies into LDS is one contiguous, line-aligned run (one page) instead of 122 row segments 8 B x N apart.
    double* cellc;      // this layer's images
    int32_t cpb;
};
// doubles between consecutive tile images: (cell fields + 32 direction rows) x cpb, rounded up to whole 128-byte lines
int64_t tile_image_doubles(int cpb);

struct TimeSetupArgs {
    int nsteps;
    const int32_t *year, *month, *day;
    const double* hour;
    const double* raw[15];  // TF_TC .. TF_DTRP, each [tsteps]
    const double* winddir;
    double lat, lon;
    double* tt;  // [ndays][TF_COUNT][24]
};

struct DateSetupArgs {
    int nsteps;
    const int32_t *year, *month, *day;
    const double* hour;
    const double* winddir;
    double* dt;       // [tsteps][4]: sin(dec), cos(dec), eot, hour
    int32_t* windex;  // [tsteps]
};

struct SolveArgs {
    int64_t N;
    const double* cellc;  // [layers][ntiles][tile image], see CellSetupArgs
    int64_t ntiles_total; // tiles of the raster = images per layer
    const int32_t* daylayer;  // [ndays] vegetation layer of each day, -1: no layer covers it; null: layer 0
    const double* tt;     // vector forcing: [ndays][TF_COUNT][24]
    // array forcing: the slot of the TILED forcing ring — block of (tile, day d of the launch, series f) at
    //   af_base + tile * af_tile_stride + d * af_day_stride + f * ring_block_doubles(cells per tile), lane order (ring_pos),
    // series in TF_TC .. TF_DTRP order.  Coarse array forcing (crows > 0): af_base = [15][crows*ccols][tsteps], af_stride
    // elements between the series.
    const double* af_base;
    int64_t af_stride;
    int64_t af_tile_stride, af_day_stride;
    const double* dt;       // [tsteps][4]
    const int32_t* windex;  // [tsteps]
    const double* mxtc;     // [N]
    // coarse array forcing (af_base = [15][crows*ccols][tsteps], whole series resident): crows > 0
    int32_t crows, ccols;
    int32_t altcorrect;     // 0, 1 (fixed lapse rate), 2 (humidity-dependent)
    // outputs.  out_sel packs, 4 bits per variable, the slab index of variable v (15 = not requested).
    // reqhgt >= 0, the TILED ring (RingView below): the block of (tile, day d of the slot, slab s) starts at
    //   out_base + tile * out_tile_stride + d * out_day_stride + s * out_var_stride
    // and holds ring_block_doubles(cells per tile) values in the solver's lane order (ring_pos).
    // reqhgt < 0, the linear ring: slab s is [slot steps][N] at out_base + s * out_stride.
    double* out_base;
    int64_t out_stride;
    uint64_t out_sel;
    int64_t slot_step0;     // linear ring: first step of this launch inside the slot
    int64_t out_tile_stride, out_day_stride, out_var_stride;
    int32_t slot_day0;      // tiled ring: first day of this launch inside the slot
    // reqhgt < 0
    double* tgser;          // [N][tsteps]
    double* ddsum;          // [N]
    int32_t day0, ndays;
    int32_t total_days;  // days of the whole series (the time table's extent)
    int32_t need_pass2;  // 0: no requested output comes from pass 2 (Tz, tleaf, relhum, Rlwdown, Rlwup)
    int32_t need_tv;     // 0: no requested output comes from TVaboveground (cpp:2287-2303)
    // tiles of this launch: tile_list[ntiles_launch] (null: tiles 0 .. ntiles_launch-1; ntiles_launch <= 0: all)
    const int32_t* tile_list;
    int64_t ntiles_launch;
    // fast-clamp launches: tiles in which a canary tripped, redone by k_solve_fix for the launch's days
    int32_t* fix_count;
    int32_t* fix_list;      // [fix_cap]
    int32_t fix_cap;
    Globals g;
};

struct BelowArgs {
    int64_t N;
    int32_t tsteps, complete, hiy, per_cell_pointm;
    double reqhgt, mat;
    const double* cellflag_hgt;  // hgt [N] (NA test)
    const double* tg;            // [N][tsteps]
    const double* ddsum;         // [N]
    const double *Tgp, *Tbp;     // [tsteps] or [N][tsteps]; complete==0 only
    double* scratch;             // [N][2*ndays]
    double* tz;                  // [N][tsteps]
};

struct BioclimArgs {
    int64_t N;
    int32_t tsteps;
    RingView tz;           // Tz or tleaf, tsteps steps from step 0 of the slot
    RingView soilm;
    const int32_t *wetq, *dryq, *hotq, *colq;
    int32_t nwet, ndry, nhot, ncol;
    double* bio;           // [19][N]
};
void launch_bioclim(const BioclimArgs& a, hipStream_t s);
// The same nineteen values STREAMED (round 5): the solver runs in day chunks into a small ring and each chunk is folded into
// per-cell running state — sums, day extremes, quarter sums in the reference's index order (cpp:3245-3560) — so that nothing of
// size cells x steps is ever allocated.  State rows, [kBioStateRows][N]:
//   0 tz at step 0 (the NA test, cpp:3505)   1 bio1's running sum   2 bio2's sum of daily ranges   3..14 the twelve daily means
//   15 bio5's maximum   16 bio6's minimum   17..20 tz quarter sums   21 bio12's sum   22 / 23 soil moisture max / min
//   24 its sum over all steps   25..28 soil moisture quarter sums
constexpr int kBioStateRows = 29;
struct BioAccArgs {
    int64_t N;
    RingView tz, soilm;        // step 0 of the views = the chunk's first step
    int32_t day0, ndays;       // the chunk: absolute first day, whole days
    const int32_t* q[4];       // wettest / driest / hottest / coldest quarter's step indices, ascending (checked by the host)
    int32_t qlo[4], qhi[4];    // the entries that fall into this chunk
    double* state;
};
void launch_bioclim_acc(const BioAccArgs& a, hipStream_t s);
struct BioFinArgs {
    int64_t N;
    int32_t tsteps;
    const double* state;
    double* bio;               // [19][N]
    // the soil moisture series again for the variance's second pass (cpp:3391-3398): it is a function of the cell's constants and
    // the point model's soil moisture alone (soil_spread, cpp:1021-1032), made here with the very function the solver's lanes run
    const double* cellc;       // tile-major constant table
    int64_t ntiles_total;
    int32_t cpb;
    const int32_t* daylayer;   // or null
    const double* tt;          // time table [day][TF_COUNT][24]
};
void launch_bioclim_fin(const BioFinArgs& a, hipStream_t s);
void launch_fill(double* p, int64_t n, double v, hipStream_t s);
// dst[ci + ncells*k] = src(cells[ci], step0 + k), k < nsteps
void launch_gather_cells(const RingView& src, int64_t step0, int64_t nsteps, const int64_t* cells, int64_t ncells, double* dst,
                         hipStream_t s);
// dst[c + N*k] = src(c, step0 + k), k < nsteps: the reference's [rows, cols, steps] layout out of the tiled ring
void launch_untile(const RingView& src, int64_t step0, int64_t nsteps, double* dst, hipStream_t s);
// the other way (array forcing's upload path): dst(c, k) = src[c + N*k], k < nsteps, `dst` a tiled view (its base is written)
void launch_tile_series(const double* src, int64_t nsteps, const RingView& dst, hipStream_t s);
// per-cell maximum over time of the bilinearly interpolated coarse temperature [crows*ccols][tsteps]
// `force`: the 15 coarse slabs (stride elements apart); elevd / pkfac null without altitude correction
void launch_mxtc_coarse(const double* force, int64_t stride, int crows, int ccols, int tsteps, const double* rowpos,
                        const double* colpos, int64_t rows, int64_t N, int altcorrect, const double* elevd,
                        const double* pkfac, double* mx, hipStream_t s);
struct PackNcArgs {
    RingView src[10];        // each variable's slot view
    int64_t step0;           // first step (of the slot) of this launch
    double scale[10];
    int32_t fill_only[10];
    int32_t nv;
    int32_t missval;
    int64_t rows, cols;
    int64_t rec_words;       // record length in 4-byte words (2 + nv * rows * cols)
    int32_t* dst;            // first record
};
// at most 65535 / nv steps per launch
void launch_pack_nc(const PackNcArgs& a, int64_t nsteps, hipStream_t s);
void launch_pack_transpose(const RingView& src, int64_t step0, int64_t rows, int64_t cols, int64_t nsteps, double scale,
                           int32_t* dst, hipStream_t s);
// out2: twi_scratch_doubles() doubles; [0] = sum, [1] = count on completion
void launch_twi_partial(const double* twi, int64_t n, double tfact, double* out2, hipStream_t s);
int twi_scratch_doubles();
void launch_cell_setup(const CellSetupArgs& a, hipStream_t s);
void launch_time_setup(const TimeSetupArgs& a, hipStream_t s);
void launch_date_setup(const DateSetupArgs& a, hipStream_t s);
void launch_mxtc(const double* tc, int64_t N, int nsteps, double* mx, hipStream_t s);
// fast: vector forcing, reqhgt >= 0 only — the min / max clamp variant followed by the fix-up kernel (needs fix_count /
// fix_list; the caller zeroes *fix_count on the stream first); the tiles and days handed over must be REGULAR
// soil_daily: every day of the launch carries kSoilDaily (vector forcing): the per cell-day soil state is computed once per
// tile and day and shared through LDS
void launch_solve(const SolveArgs& a, int cells_per_block, bool af, bool bg, bool fast, bool soil_daily, hipStream_t s);
// out[t] = 1 if every valid cell of tile t (cpb consecutive cells) is FL_REGULAR in all layers
void launch_tile_regular(const double* cellc, int64_t N, int layers, int cpb, uint8_t* out, hipStream_t s);
// A launch over a SUBSET of the cells (mcf_plan_run_days_cells): the wanted cells gathered into dense tiles of their own.
//   cells_class  cls[c] = 0 (need[c] == 0) / 1 (wanted, its tile tile_regular — launch_tile_regular's flags; null: all) / 2 (wanted, not);
//                blockcnt[2 b + k - 1] = cells of class k among cells 256 b .. 256 b + 255
//   cells_place  list[blockoff[2 b + k - 1] + rank of the cell among its block's class-k cells] = c  (ascending in each class)
//   gather_image the sub-tiles' images [layers][ntiles_sub][tile_image_doubles] from the plan's; list entry < 0: an all-zero
//                column (flags 0: not valid)
//   scatter_cells the sub-ring's values of `ndays` days to the cells' own places in a ring slot (`ring` = the slot's first day
//                to write); both rings [tile][day][variable][block], `day_doubles` = variables x block
void launch_cells_class(const uint8_t* need, const uint8_t* tile_regular, int64_t N, int cpb, uint8_t* cls, int32_t* blockcnt,
                        hipStream_t s);
void launch_cells_place(const uint8_t* cls, int64_t N, const int32_t* blockoff, int32_t* list, hipStream_t s);
void launch_gather_image(const int32_t* list, int64_t ntiles_sub, const double* src, int64_t ntiles_src, int layers, int cpb,
                         double* dst, hipStream_t s);
void launch_scatter_cells(const int32_t* list, int64_t ntiles_sub, const double* sub, int64_t sub_tile_stride, double* ring,
                          int64_t tile_stride, int64_t day_doubles, int cpb, int ndays, hipStream_t s);
void launch_belowground(const BelowArgs& a, hipStream_t s);
void launch_selftest_math(int kind, const double* x, const double* y, double* out, int64_t n, hipStream_t s);

// a plan's ring slot as another translation unit's kernels address it (mcf_snow.hip writes the snow-day microclimate into it):
// views of the requested variables (has[v] = 0: not requested), the plan's stream, its cell count and its device
}  // namespace mcf
struct mcf_plan;
namespace mcf {
int plan_ring_views(mcf_plan* p, int slot, RingView views[10], int32_t has[10], hipStream_t* stream, int64_t* N, int* device,
                    int* slot_days);

int cell_field_count();
void print_variant_stats();  // hook for the timing variants under tools/variants/ (empty in the shipped library)
int soil_daily_bit();        // kSoilDaily, likewise
int step_irregular_bit();   // kStepIrregular of the packed TF_IDX value (last time field)
int time_field_count();
double hf_pow02(double rs);

}  // namespace mcf
