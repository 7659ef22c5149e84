/*
 * mcf.h — C ABI of libmcfhip: the MI355X (gfx950) grid microclimate solver.
 *
 * Drop-in boundary for the reference's `.Call` entry points (all paths below are
 * under the upstream repository root):
 *
 *   mcf_runmicro1()  replaces  _microclimf_runmicro1Cpp   src/RcppExports.cpp:250-272
 *                    (R stub   runmicro1Cpp               R/RcppExports.R:72-74,
 *                     body     runmicro1Cpp               src/microclimfCpp.cpp:2052-2337)
 *   mcf_runmicro2()  replaces  _microclimf_runmicro2Cpp   src/RcppExports.cpp:275-297
 *                    (R stub   runmicro2Cpp               R/RcppExports.R:76-78,
 *                     body     runmicro2Cpp               src/microclimfCpp.cpp:2340-2621)
 *   mcf_runmicro3/4() replace  _microclimf_runmicro3Cpp/4Cpp (time-varying vegetation,
 *                     bodies src/microclimfCpp.cpp:2624-3226), R stubs R/RcppExports.R:80-86
 *
 * Everything is plain pointers + sizes; no R, Rcpp or torch types.  All arrays
 * are IEEE fp64, column-major with the raster row as the fastest index, exactly
 * as R hands them to the reference:
 *
 *   matrix  [rows, cols]          element (i,j)     at  i + rows*j
 *   3-D     [rows, cols, n]       element (i,j,k)   at  i + rows*j + rows*cols*k
 *
 * "NA" follows Rcpp's NumericVector::is_na, i.e. any NaN.  Cells whose `hgt`
 * is NA are skipped and every requested output holds R's NA_real_ bit pattern
 * (0x7FF00000000007A2) there, as do time steps past the last whole day
 * (ndays = tsteps / 24 truncates, src/microclimfCpp.cpp:2116).
 *
 * Two levels:
 *   1. one-shot host calls (mcf_runmicro1 / mcf_runmicro2): host pointers in,
 *      host pointers out; the library stages everything through HBM in day
 *      chunks.  This is what the R glue binds.
 *   2. the plan API: inputs are uploaded once and stay resident in HBM, day
 *      chunks are solved into a device-resident output ring, and the caller
 *      decides what (if anything) is copied back.  bench.py and multi-GPU row
 *      tiling use this level.
 *
 * There is no CPU fallback: every solver / kernel entry point fails with
 * MCF_ERR_NO_DEVICE when no HIP device is usable.  Host-side by nature, as in the
 * reference, and therefore usable without a device: the one-point time-series model
 * (mcf_bigleaf, mcf_soilm, mcf_pointmprocess, mcf_weatherhgt) and the file side of
 * the writetonc sink (mcf_nc_create, mcf_nc_write_host, mcf_nc_close), and the
 * flow-accumulation sweep behind soilc$twi (mcf_flowacc, mcf_topidx).
 */
#ifndef MCF_H
#define MCF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCF_ABI_VERSION 6   /* 2: mcf_grid_inputs grew the coarse-forcing fields; 3: tiled output ring (mcf_plan_ring_layout);
                             * 4: mcf_nc_spec grew format / deflate_level (zero = the behaviour of version 3); 5: mcf_runmicrosnow1 / mcf_snowrun_*;
                             * 6: mcf_snowdriver_in grew af_wsa_s (the former `reserved`) / af_wind at its end — read only with array weather */

/* Output variables, in the order of the reference's returned list
 * (src/microclimfCpp.cpp:2326-2335) and of its `out` logical(10). */
enum {
    MCF_OUT_TZ = 0,       /* "Tz"        air T at reqhgt (reqhgt>0), ground T (==0), soil T (<0) */
    MCF_OUT_TLEAF = 1,    /* "tleaf"     leaf T (reqhgt>0 only)                              */
    MCF_OUT_RELHUM = 2,   /* "relhum"    relative humidity % (reqhgt>0 only)                 */
    MCF_OUT_SOILM = 3,    /* "soilm"     distributed volumetric soil moisture                */
    MCF_OUT_WINDSPEED = 4,/* "windspeed" wind speed at reqhgt                                */
    MCF_OUT_RDIRDOWN = 5, /* "Rdirdown"  downward direct SW at reqhgt                        */
    MCF_OUT_RDIFDOWN = 6, /* "Rdifdown"  downward diffuse SW                                 */
    MCF_OUT_RLWDOWN = 7,  /* "Rlwdown"   downward LW (reqhgt>=0)                             */
    MCF_OUT_RSWUP = 8,    /* "Rswup"     upward SW                                           */
    MCF_OUT_RLWUP = 9,    /* "Rlwup"     upward LW (reqhgt>=0)                               */
    MCF_NOUT = 10
};

/* Error codes (0 = success).  mcf_last_error() gives the message of the most
 * recent failure on the calling thread. */
enum {
    MCF_OK = 0,
    MCF_ERR_ARG = 1,        /* NULL/ill-sized argument                                  */
    MCF_ERR_NO_DEVICE = 2,  /* no usable HIP device / kernels not loadable              */
    MCF_ERR_HIP = 3,        /* a HIP runtime call failed (message has the HIP error)    */
    MCF_ERR_NOMEM = 4,      /* request does not fit device memory                       */
    MCF_ERR_STATE = 5       /* plan used out of order                                   */
};

/* obstime data.frame (src/microclimfCpp.cpp:2057-2060).  R passes year/month/day
 * as doubles that Rcpp coerces to int; the glue does that coercion. */
typedef struct mcf_obstime {
    const int32_t *year, *month, *day; /* [tsteps] */
    const double *hour;                /* [tsteps] decimal hour */
} mcf_obstime;

/* climdata (src/microclimfCpp.cpp:2062-2071 data.frame columns temp, es, ea,
 * tdew, pres, swdown, difrad, lwdown, windspeed, winddir; :2350-2359 list
 * entries tc, es, ea, tdew, pk, ...).  Vector forcing: each [tsteps].  Array
 * forcing: each [rows,cols,tsteps] except winddir, which stays [tsteps]. */
typedef struct mcf_climate {
    const double *tc, *es, *ea, *tdew, *pk, *swdown, *difrad, *lwdown, *windspeed, *winddir;
} mcf_climate;

/* pointm (src/microclimfCpp.cpp:2073-2082; :2361-2370, where G is named Gp).
 * T0p and DDp are accepted by the reference but never read; they are not part
 * of this ABI.  Tg/Tbp are only read when reqhgt < 0 and complete == 0 and may
 * be NULL otherwise.  Shapes as mcf_climate. */
typedef struct mcf_pointm {
    const double *soilm, *Tg, *Tbp, *G, *umu, *kp, *muGp, *dtrp;
} mcf_pointm;

/* vegp list (src/microclimfCpp.cpp:2084-2094), each [rows,cols]; with time-varying
 * vegetation (runmicro3Cpp / runmicro4Cpp, src/microclimfCpp.cpp:2667-2679) each is
 * [rows,cols,veg_layers]. */
typedef struct mcf_vegp {
    const double *hgt, *pai, *x, *gsmax, *leafr, *leaft, *clump, *leafd, *paia, *leafden;
} mcf_vegp;

/* soilc list (src/microclimfCpp.cpp:2096-2111): 13 matrices + wsa[rows,cols,8]
 * + hor[rows,cols,24]. */
typedef struct mcf_soilc {
    const double *Smin, *Smax, *gref, *soilb, *Psie, *Vq, *Vm, *Mc, *rho;
    const double *slope, *aspect, *twi, *svfa;
    const double *wsa; /* [rows,cols,8]  */
    const double *hor; /* [rows,cols,24] */
} mcf_soilc;

typedef struct mcf_grid_inputs {
    int64_t rows, cols, tsteps;
    int32_t array_forcing; /* 0: runmicro1Cpp/3Cpp geometry, 1: runmicro2Cpp/4Cpp geometry, 2: coarse arrays (below) */
    int32_t veg_layers;    /* 0 or 1: static vegetation; >1: runmicro3Cpp/4Cpp `dfsel` layers    */
    mcf_obstime obstime;
    mcf_climate clim;
    mcf_pointm pointm;
    mcf_vegp vegp;
    mcf_soilc soilc;
    double lat, lon;           /* vector forcing (runmicro1Cpp args lat, lon)            */
    const double *lats, *lons; /* array forcing  (runmicro2Cpp args lats, lons), [rows,cols] */
    /* dfsel of runmicro3Cpp/4Cpp (src/microclimfCpp.cpp:2629-2640): layer l drives the steps
     * lyr_st[l] .. lyr_ed[l] (0-based, whole days); NULL when veg_layers <= 1 */
    const int32_t *lyr_st, *lyr_ed;
    /* array_forcing == 2 — COARSE array forcing, the MI355X-native form of `.runmodel2Cpp` (R/internal.R:1175-1343):
     * the reference resamples every coarse climate / point-model variable to the fine raster (`.cca` ->
     * terra::resample, bilinear; R/internal.R:523-542, 1224-1277) and hands runmicro2Cpp full [rows,cols,T] arrays —
     * 120 B per cell-step to upload, hold and read back, the reason a tile is capped at 2e7 cell-steps
     * (R/Cppwrappers.R:469).  Here the coarse arrays stay coarse: clim.{tc,pk,swdown,difrad,lwdown,windspeed},
     * coarse_relhum, coarse_winddir and pointm.{soilm,G,umu,kp,muGp,dtrp} are [coarse_rows, coarse_cols, tsteps]; the
     * solver interpolates them bilinearly per cell-step and derives es, ea, tdew (.satvap, .dewpoint after
     * interpolating temp and relhum), wind speed (from interpolated u, v components) and the raster-mean wind
     * direction exactly where `.runmodel2Cpp` does.  clim.es, ea, tdew, winddir are ignored.
     * coarse_rowpos[i] / coarse_colpos[j]: position of raster row i / column j in units of coarse rows / columns,
     * 0 = centre of the first coarse row / column, clamped by the caller to [0, coarse_rows-1] / [0, coarse_cols-1]
     * (edge replication).  reqhgt < 0 needs complete = 1 in this mode. */
    int32_t coarse_rows, coarse_cols;
    const double *coarse_rowpos, *coarse_colpos;
    const double *coarse_relhum, *coarse_winddir;
    /* `.runmodel2Cpp`'s altcorrect (R/internal.R:1233-1251): 0 none; 1 fixed lapse rate 5 K/km; 2 humidity-dependent
     * lapse rate (`.lapserate`, R/internal.R:545-550).  With 1 or 2: coarse_dtm [coarse_rows, coarse_cols] (elevation
     * of the climate cells, NA read as 0) and fine_dtm [rows, cols]; pressure goes to sea level on the coarse grid and
     * back up on the fine one, temperature moves by lapse rate x (interpolated coarse elevation - fine elevation);
     * es, ea, tdew come from the UNcorrected temperature, as in the reference. */
    int32_t coarse_altcorrect;
    const double *coarse_dtm, *fine_dtm;
    /* 0 or rows: dense arrays.  > rows: every raster array above ([rows, cols(, layers | steps)]: vegp, soilc, wsa, hor, lats,
     * lons, fine_dtm, the array-forcing series) AND the output arrays of the one-shot entry points are row blocks of a
     * taller column-major raster with `row_pitch` rows — read and written in place (how mcf_runmicro1_multi hands a block
     * to a device without copying the host arrays). */
    int64_t row_pitch;
} mcf_grid_inputs;

typedef struct mcf_options {
    double reqhgt, zref;
    double Sminp, Smaxp; /* accepted for signature parity; unused by the reference (cpp:975) */
    double tfact;
    double mat;          /* mean annual temperature, reqhgt<0 && !complete */
    int32_t complete;
    int32_t out[MCF_NOUT];
    int32_t device;          /* HIP device ordinal                                      */
    int32_t days_per_chunk;  /* 0 = choose from free HBM                                */
    int32_t cells_per_block; /* 0 = default (21: two 8-wave workgroups per CU); 16, 21, 32, 42 (42: vector forcing only —
                              * array forcing is given its 32-cell tiles instead) */
} mcf_options;

/* Host output buffers, each [rows,cols,tsteps] or NULL when out[v]==0. */
typedef struct mcf_outputs {
    double *var[MCF_NOUT];
} mcf_outputs;

int mcf_abi_version(void);
const char *mcf_last_error(void);
/* Number of visible HIP devices (0 when none); never fails. */
int mcf_device_count(void);

/* One-shot host-to-host solves. */
int mcf_runmicro1(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out);
int mcf_runmicro2(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out);
/* The same solves on several devices of one node from ONE process — the R drop-in's way to a multi-GPU node (R has no
 * torch.distributed): the raster is cut into `n_blocks` contiguous row blocks of about equal valid-cell count (boundaries
 * on multiples of 10 rows; R/internal.R:909-925 is the reference's own tiling), block b goes to devices[b % n_devices],
 * one host thread per device.  The solver's one global reduction (mean of log(twi)/tfact, src/microclimfCpp.cpp:993-1004)
 * is taken over the whole raster first and installed in every block, so the result is bit for bit the single-device one.
 * n_devices = 0: every visible device; n_blocks = 0: one block per device (more blocks than devices are time-sliced).
 * mcf_runmicro3_multi / 4_multi: the same with time-varying vegetation (declared behind mcf_runmicro3 / 4 below). */
typedef struct mcf_multi {
    int32_t n_devices;
    const int32_t *devices;
    int32_t n_blocks;
} mcf_multi;
int mcf_runmicro1_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_multi *multi, mcf_outputs *out);
int mcf_runmicro2_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_multi *multi, mcf_outputs *out);

/* Time-varying vegetation: replace _microclimf_runmicro3Cpp / _microclimf_runmicro4Cpp
 * (src/RcppExports.cpp:300-322 / 326-348; bodies src/microclimfCpp.cpp:2624-2924 / 2926-3226).
 * Same physics with the vegetation layer chosen per day from `dfsel`; steps outside every
 * layer's range stay NA. */
int mcf_runmicro3(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out);
int mcf_runmicro4(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out);
int mcf_runmicro3_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_multi *multi, mcf_outputs *out);
int mcf_runmicro4_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_multi *multi, mcf_outputs *out);

/* ---- plan API: HBM-resident inputs, device output ring ---------------------- */
typedef struct mcf_plan mcf_plan;

/* Uploads all static inputs (and, for vector forcing, builds the per-timestep
 * table on the device).  `ring_days` is the capacity of each output ring slot in
 * days, `ring_slots` the number of slots (>=1).  For reqhgt<0 the ring must hold
 * the whole series (ring_days >= tsteps/24, enforced). */
int mcf_plan_create(const mcf_grid_inputs *in, const mcf_options *opt,
                    int32_t ring_days, int32_t ring_slots, mcf_plan **plan);
void mcf_plan_destroy(mcf_plan *plan);

/* Raster-wide mean of log(twi)/tfact over non-NA cells (src/microclimfCpp.cpp:
 * 993-1004) is the solver's only global reduction.  mcf_plan_create computes it
 * for the plan's own cells; row-tiled multi-GPU runs fetch the partial
 * (sum,count), all-reduce them (RCCL) and install the global mean. */
int mcf_plan_twi_partial(mcf_plan *plan, double *sum, int64_t *count);
int mcf_plan_set_twi_mean(mcf_plan *plan, double mean);

/* Array forcing only: upload the [rows,cols,24*ndays] slabs of every forcing
 * array for days [day0, day0+ndays) into the plan's forcing buffer (slot). */
int mcf_plan_upload_forcing_days(mcf_plan *plan, const mcf_grid_inputs *in,
                                 int32_t day0, int32_t ndays, int32_t slot);

/* Solve days [day0, day0+ndays) into ring slot `slot` (async on the plan's
 * stream).  The slot holds the reference's [rows,cols,24*ndays] arrays
 * (src/microclimfCpp.cpp:2292-2303) in a device-internal order — see
 * mcf_plan_ring_layout; mcf_plan_fetch* / mcf_nc_write_plan return them in the
 * reference's layout. */
int mcf_plan_run_days(mcf_plan *plan, int32_t day0, int32_t ndays, int32_t slot);
/* The same with the days written at day `slot_day0` of the slot instead of its start (vector forcing, reqhgt >= 0): several
 * runs of days of one chunk, each at its own place (the snow branch's no-snow days). */
int mcf_plan_run_days_at(mcf_plan *plan, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0);
/* ... and for a SUBSET of the tiles: skip_tile[t] != 0 (t < n_tiles = ceil(cells / mcf_ring_layout.cells_per_tile); tile t =
 * cells t * cells_per_tile ... of the column-major raster) leaves tile t's blocks of these days untouched in the slot.  The
 * snow run uses it for the days that are no-snow days AND snow days: `.runmicrosnow1` (R/internal.R:3632-3655) overwrites
 * every snow-covered cell-step of such a day with gridmicrosnow1's value, so the solver's values of a tile whose cells all lie
 * under snow for the whole run of days are dead — mcf_snowplan_covered_tiles finds those tiles.  skip_tile = NULL: all. */
int mcf_plan_run_days_masked(mcf_plan *plan, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0,
                             const uint8_t *skip_tile, int64_t n_tiles);
/* ... and for a SUBSET of the CELLS (vector forcing, reqhgt >= 0): need_cell — DEVICE memory on the plan's device, complete when
 * the call is made, one byte per cell of the column-major raster (n_cells = rows x cols) — marks the cells whose values of these
 * days are wanted; every other cell's values in the slot stay as they are.  The marked cells are gathered into dense tiles of
 * their own (their constants copied from the plan's tiles; the cells of the plan's regular tiles and those of its other tiles
 * apart, so that every cell meets the instantiation it meets in a launch of the plan's own tiles), solved into a ring of their
 * own and copied to their places in the slot: the same bits as mcf_plan_run_days_at's.  Where tiles are the unit (mcf_plan_run_days_masked) one snow-free cell keeps its
 * whole 21-cell tile in the launch; on a day that is both a snow day and a no-snow day the cells the merge of `.runmicrosnow1`
 * (R/internal.R:3632-3655) keeps from the no-snow model are a few per cent scattered over most tiles —
 * mcf_snowplan_free_cells finds them.  *n_gathered (may be NULL) = marked cells.  Memory: the gathered tiles' constants, and
 * their ring for as many of the days at a time as an eighth of the plan's ring holds (MCF_CELLS_RING_GB overrides). */
int mcf_plan_run_days_cells(mcf_plan *plan, int32_t day0, int32_t ndays, int32_t slot, int32_t slot_day0,
                            const uint8_t *need_cell, int64_t n_cells, int64_t *n_gathered);
/* Vector forcing: replace the series' maximum air temperature (src/microclimfCpp.cpp:2159-2168; it caps the
 * Penman-Monteith temperature excess, cpp:1236).  The snow branch solves a SUBSET of the days and the reference takes
 * the maximum over that subset. */
int mcf_plan_set_mxtc(mcf_plan *plan, double mxtc);
/* Array forcing (at the raster's resolution): the same per CELL (src/microclimfCpp.cpp:2467-2471) over the days with
 * dayflag[d] != 0 — the temperature series of `in` streamed once more, a run of flagged days at a time.  Also since this
 * version: mcf_plan_run_days_at accepts a run of days INSIDE the days a slot's forcing was uploaded for, with array forcing. */
int mcf_plan_set_mxtc_days(mcf_plan *plan, const mcf_grid_inputs *in, const int32_t *dayflag, int32_t ndays);
/* reqhgt<0: after every day has been solved into slot 0, smooth the stored
 * ground-temperature series into Tz (Tbelowgroundv, cpp:1474-1539). */
int mcf_plan_belowground(mcf_plan *plan);
int mcf_plan_sync(mcf_plan *plan);

/* Copy `nsteps` time steps of variable `var` from ring slot `slot` (starting at
 * step `step0` within the slot) to host memory. */
int mcf_plan_fetch(mcf_plan *plan, int32_t slot, int32_t var, int64_t step0,
                   int64_t nsteps, double *host_dst);
/* ... into a row block of a taller column-major array (row_pitch rows per column; 0 = dense). */
int mcf_plan_fetch_pitched(mcf_plan *plan, int32_t slot, int32_t var, int64_t step0,
                           int64_t nsteps, double *host_dst, int64_t row_pitch);
/* Sparse read-back for verification and point queries: `nsteps` steps of variable `var` for the `ncells`
 * cells listed in `cells` (0-based column-major cell indices i + rows*j, any order), gathered on the device
 * and returned as host_dst[ci + ncells*k].  Moves ncells*nsteps values instead of a whole [rows,cols,nsteps]
 * slab (bench.py checks a sample of the timed run's last ring slot against the oracle with it). */
int mcf_plan_fetch_cells(mcf_plan *plan, int32_t slot, int32_t var, int64_t step0, int64_t nsteps,
                         const int64_t *cells, int64_t ncells, double *host_dst);
/* The same, packed as `writetonc` stores it (R/dataprep.R:1064-1069 `atonc`, :1158-1167): int32
 * round-half-even(value * scale), transposed per step to [cols, rows] (east fastest), NA ->
 * NA_integer_ (INT32_MIN; ncvar_put writes the variable's missval -9999 for it).  writetonc's scales:
 * 100 for Tz, tleaf, soilm, windspeed; 1 for relhum and the radiation terms.  Halves the bytes that
 * cross PCIe and arrives in the file's layout.  `kernel_ms` (optional) receives the device time of the
 * pack kernel. */
int mcf_plan_fetch_packed(mcf_plan *plan, int32_t slot, int32_t var, int64_t step0, int64_t nsteps,
                          double scale, int32_t *host_dst, float *kernel_ms);
/* ---- writetonc sink ---------------------------------------------------------------------
 * Replaces `writetonc(mout, fileout, dtm, reqhgt, vars)` (R/dataprep.R:1063-1260; called per tile by
 * runmicro_big, R/Cppwrappers.R:531): the solver's outputs as int32 (x100 for Tz, tleaf, soilm,
 * windspeed; x1 for relhum and the radiation terms; round half even; NA -> missval -9999) on the
 * dimensions east, north, time, with writetonc's variable names, long names, units, `crs` variable and
 * time attributes.  Two containers (`format`):
 *   MCF_NC_CLASSIC  netCDF classic, 64-bit offsets, `time` as the record dimension, uncompressed — needs nothing on the
 *                   host and streams at disk speed (mcf_ncfile.hpp); every netCDF reader opens it like the reference's file;
 *   MCF_NC_NETCDF4  the reference's own container (ncvar_def(..., compression = 9), dataprep.R:1110-1111): HDF5 laid out by
 *                   the netCDF-4 conventions, chunked [1 step][row strip][cols], deflate `deflate_level`; written through
 *                   the host's HDF5 library, bound at run time (MCF_ERR_ARG with the reason if there is none), the chunks
 *                   deflated by a team of host threads (mcf_nc4file.hpp).
 * Both take the same records (the device packs them once, k_pack_nc).  The dataset keeps writetonc's orientation: north[i] belongs to raster row i,
 * with `north` the ASCENDING northings of dataprep.R:1073 (the reference does not flip the rows).
 *
 * `vars[v]` selects solver output v (MCF_OUT order); writetonc defines Tz, tleaf, relhum, soilm,
 * windspeed and the five radiation terms for reqhgt > 0, Tz, soilm and radiation for reqhgt == 0, Tz and
 * soilm below ground — anything else is MCF_ERR_ARG.  `reference_puts_only` = 1 reproduces the file the
 * reference really produces: its `ncvar_put` guards test names ("raddir", …) that never occur in `vars`
 * (dataprep.R:1163-1167) and the soilm put refers to an undefined object (:1161), so those variables
 * are defined but hold missval throughout; 0 writes what was evidently meant. */
enum { MCF_NC_CLASSIC = 0, MCF_NC_NETCDF4 = 1 };
typedef struct mcf_nc_spec {
    int32_t rows, cols;          /* raster rows (= length of north), columns (= length of east) */
    int64_t nsteps;
    const double *east;          /* [cols]  seq(xmin + res/2, xmax - res/2, res)                 */
    const double *north;         /* [rows]  seq(ymin + res/2, ymax - res/2, res)                 */
    const double *time_hours;    /* [nsteps] hours since 1970-01-01 00:00                        */
    const char *crs_wkt;         /* crs(dtm); may be NULL                                        */
    double reqhgt;
    int32_t vars[MCF_NOUT];
    int32_t reference_puts_only;
    int32_t format;              /* MCF_NC_CLASSIC (0) or MCF_NC_NETCDF4                                    */
    int32_t deflate_level;       /* MCF_NC_NETCDF4: 0 = writetonc's compression = 9, 1..9, -1 = none      */
} mcf_nc_spec;
typedef struct mcf_ncfile mcf_ncfile;
/* Host only (no device needed): create the file with header, coordinates and room for every record. */
int mcf_nc_create(const char *path, const mcf_nc_spec *spec, mcf_ncfile **nc);
/* Host only: records [step0, step0+nsteps) from host arrays; vars[v] = [rows, cols, nsteps] column-major
 * doubles as mcf_runmicro1..4 return them (NULL for variables the file does not hold). */
int mcf_nc_write_host(mcf_ncfile *nc, int64_t step0, int64_t nsteps, const double *const vars[MCF_NOUT]);
/* Records [file_step0, file_step0+nsteps) straight from a plan's ring slot: packed, transposed and
 * byte-ordered on the device (k_pack_nc, 4 B per value over PCIe instead of 8), written to the file
 * while the next piece is being packed and copied. */
int mcf_nc_write_plan(mcf_ncfile *nc, mcf_plan *plan, int32_t slot, int64_t slot_step0, int64_t file_step0,
                      int64_t nsteps, float *kernel_ms);
int mcf_nc_close(mcf_ncfile *nc);

/* Device address of a ring slot variable, for device-side consumers, and the layout it is to be read with.
 * reqhgt >= 0: the ring is TILED — the values k_solve's workgroup produces for one variable on one day (cells_per_tile
 * consecutive cells x 24 hours) are one block of block_doubles doubles in the workgroup's lane order, so that the solver
 * stores whole 128-byte lines of one contiguous run per tile and day instead of 240 row segments 8 B x cells apart.
 * Value (cell c, step k of the slot) of the variable is at dev_ptr[mcf_ring_index(layout, c, k)]:
 *   (c / cells_per_tile) * tile_stride + (k / 24) * day_stride + pos(c % cells_per_tile, k % 24),
 *   pos(cell, h) = h * cells_per_tile + cell, except for 21-cell tiles:
 *   64 * (h / 3) + (cell < 16 ? 16 * (h % 3) + cell : 48 + 5 * (h % 3) + cell - 16).
 * reqhgt < 0 (tiled == 0): linear, dev_ptr[c + cells * k], the reference's layout. */
int mcf_plan_slot_ptr(mcf_plan *plan, int32_t slot, int32_t var, void **dev_ptr);
typedef struct mcf_ring_layout {
    int32_t tiled;            /* 0: linear [step][cell]                                  */
    int32_t cells_per_tile;
    int32_t block_doubles;    /* doubles per (tile, day, variable) block                 */
    int32_t slot_days;
    int64_t cells;            /* rows * cols                                             */
    int64_t tile_stride;      /* doubles between consecutive tiles                       */
    int64_t day_stride;       /* doubles between consecutive days of a tile              */
} mcf_ring_layout;
int mcf_plan_ring_layout(mcf_plan *plan, mcf_ring_layout *layout);
/* Host-side index helper (no device needed); -1 when (cell, step) is outside the slot. */
int64_t mcf_ring_index(const mcf_ring_layout *layout, int64_t cell, int64_t step);

/* HIP-event timing on the plan's stream: start records an event, stop records
 * another, synchronises and returns the elapsed milliseconds between them. */
int mcf_plan_timer_start(mcf_plan *plan);
int mcf_plan_timer_stop(mcf_plan *plan, float *ms);
/* Accumulated device time (ms) and launch count of the solver kernel alone,
 * measured with per-launch HIP events when enabled. */
int mcf_plan_kernel_timing(mcf_plan *plan, int32_t enable);
int mcf_plan_kernel_stats(mcf_plan *plan, double *total_ms, int64_t *launches);

/* How the plan's launches were dispatched (vector forcing, reqhgt >= 0): the solver has two instantiations of its
 * clamps — one v_min_f64 / v_max_f64 each ("fast") for tiles whose cells' constants are all finite and in range, and the
 * reference's compare-and-select form for the others, for launches that contain a step with non-finite forcing, and for
 * tiles in which a fast wave met a NaN at a watched clamp (redone by a fix-up kernel).  Results are identical either way;
 * the counts are diagnostics (tests assert which path ran).  Synchronises the plan's stream. */
typedef struct mcf_dispatch_stats {
    int64_t fast_tiles, slow_tiles;   /* tiles per class (cells_per_block cells each)                    */
    int64_t irregular_days;           /* days with a non-finite / out-of-range forcing step             */
    int64_t fast_launches, slow_launches;
    int64_t canary_trips;             /* tiles redone by the fix-up kernel, summed over launches (one entry per
                                       * tile and launch, however many of its waves tripped)            */
} mcf_dispatch_stats;
int mcf_plan_dispatch_stats(mcf_plan *plan, mcf_dispatch_stats *stats);

/* ---- fused bioclim sink --------------------------------------------------------------
 * Replace _microclimf_runbioclim1Cpp / _microclimf_runbioclim2Cpp (bodies
 * src/microclimfCpp.cpp:3563-3588 / 3590-3616): the grid solver run with
 * out = {Tz | tleaf, soilm} followed by runbioclimCpp's per-cell reductions over time
 * (src/microclimfCpp.cpp:3245-3560).  Both stages stay on the device; only the requested
 * [rows,cols] matrices come back.  `opt->out` and `opt->complete` are ignored (the reference
 * passes its own mask and complete = true).  Time layout expected by the reference: steps
 * 0..287 twelve monthly days, 288..311 hottest day, 312..335 coldest day, then the quarter days
 * addressed by the 0-based index vectors. */
#define MCF_NBIO 19
typedef struct mcf_bioclim_sel {
    const int32_t *wetq, *dryq, *hotq, *colq; /* 0-based step indices                 */
    int32_t nwet, ndry, nhot, ncol;
    int32_t air;                              /* 1: Tz, 0: tleaf (runbioclim*Cpp `air`) */
    int32_t out[MCF_NBIO];                    /* bio1..bio19 requested                */
} mcf_bioclim_sel;
typedef struct mcf_bioclim_out {
    double *bio[MCF_NBIO]; /* each [rows,cols] or NULL; NA_real_ where the first step of Tz is NA */
} mcf_bioclim_out;
int mcf_runbioclim1(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel,
                    mcf_bioclim_out *out);
int mcf_runbioclim2(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel,
                    mcf_bioclim_out *out);
/* The time-varying-vegetation variants _microclimf_runbioclim3Cpp / 4Cpp (src/microclimfCpp.cpp:3620-3658 /
 * 3660-3700): vegetation arrays [rows, cols, 14] (deeper arrays: the first 14 layers) with the reference's fixed
 * dfsel of fourteen one-day layers; steps past the 336th belong to no layer and stay NA there too. */
int mcf_runbioclim3(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel,
                    mcf_bioclim_out *out);
int mcf_runbioclim4(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel,
                    mcf_bioclim_out *out);
/* The four over row blocks on several devices from one process (mcf_multi as for mcf_runmicro1_multi: the whole-raster twi
 * mean installed in every block, inputs read and the [rows, cols] results written in place through the row pitch): bit for
 * bit the single-device matrices. */
int mcf_runbioclim1_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel, const mcf_multi *multi,
                          mcf_bioclim_out *out);
int mcf_runbioclim2_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel, const mcf_multi *multi,
                          mcf_bioclim_out *out);
int mcf_runbioclim3_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel, const mcf_multi *multi,
                          mcf_bioclim_out *out);
int mcf_runbioclim4_multi(const mcf_grid_inputs *in, const mcf_options *opt, const mcf_bioclim_sel *sel, const mcf_multi *multi,
                          mcf_bioclim_out *out);

/* ---- terrain pre-compute (the solver's terrain inputs, built on the device) ------
 * Restates the R-side arithmetic of the reference's marshaller (R/internal.R):
 *   hor   = .horizon(dtm, 15*d), d = 0..23            R/internal.R:909-925, 1144
 *   svfa  = 0.5*cos(2*tan(mean(atan(hor))))+0.5       R/internal.R:1147-1148
 *   wsa   = .windsheltera(dtm, zref, s)               R/internal.R:949-991
 *   slope, aspect = terra::terrain (Horn), NA -> 0    R/internal.R:1124-1136
 * `dtm` is the caller's row block of the raster plus `halo_north` / `halo_south` extra rows
 * (column-major [(halo_north+rows+halo_south), cols]).  Outside the raster the reference's
 * zero padding applies; inside it a block needs 100 halo rows for hor/svfa and
 * 100 + 2.5*s for wsa (or every row up to the raster edge).  Outputs cover the own rows only
 * and may be NULL individually.  rows_total = 0 means "the block is the whole raster". */
typedef struct mcf_terrain_in {
    int64_t rows, cols;
    int32_t halo_north, halo_south;
    const double *dtm;
    double res;           /* cell size (m)                                   */
    double zref;          /* wind-shelter height, .windsheltera's whgt       */
    int32_t agg;          /* .windsheltera's s (0 -> 10)                     */
    int32_t reserved0;
    int64_t row0, rows_total;
} mcf_terrain_in;

typedef struct mcf_terrain_out {
    double *slope, *aspect; /* [rows,cols]    degrees                        */
    double *hor;            /* [rows,cols,24] tan(horizon angle)             */
    double *svfa;           /* [rows,cols]                                   */
    double *wsa;            /* [rows,cols,8]                                 */
} mcf_terrain_out;

int mcf_precompute_terrain(const mcf_terrain_in *in, const mcf_terrain_out *out, int32_t device);
/* The same for a WHOLE raster (no halos, no placement in `in`) over several devices from one process — the companion of
 * mcf_runmicro1_multi for BASELINE configs[3]'s geometry: contiguous row blocks, block b on devices[b % n_devices], each
 * with the halo rows its stencils need gathered out of the caller's array (the rows the one-process-per-GPU route exchanges
 * between ranks), results written into the caller's arrays in place.  n_devices = 0: every visible device; n_blocks = 0: one
 * block per device (more blocks than devices are time-sliced).  The values are those of the single-device call. */
int mcf_precompute_terrain_multi(const mcf_terrain_in *in, const mcf_terrain_out *out, const mcf_multi *multi);

/* ---- snow branch ----------------------------------------------------------------------
 *   mcf_gridmodelsnow1/2()  replace  _microclimf_gridmodelsnow1 / _microclimf_gridmodelsnow2
 *                           src/RcppExports.cpp:483-514  (R stubs R/RcppExports.R:108-114,
 *                           bodies src/microclimfCpp.cpp:4172-4423 / 4426-4673): the per-cell,
 *                           sequential-in-time snowpack energy and mass balance (snowoneB,
 *                           src/microclimfCpp.cpp:3835-3972) that `.snowmodel1/2` call once per
 *                           5-day chunk (R/internal.R:2587).
 *   mcf_gridmicrosnow1/2()  replace  _microclimf_gridmicrosnow1 / _microclimf_gridmicrosnow2
 *                           src/RcppExports.cpp:542-578  (R stubs R/RcppExports.R:124-130,
 *                           bodies src/microclimfCpp.cpp:4894-5056 / 5059-5214): the microclimate
 *                           of snow-covered cell-steps, written over the no-snow solver's output
 *                           (`.runmicrosnow1/2`, R/internal.R:3625).
 * Vector forcing (…1): climate and pointm entries are [tsteps]; array forcing (…2): they are
 * [rows,cols,tsteps] except winddir, and lats/lons replace lat/lon. */
enum { /* snowdenp's snow environments (src/microclimfCpp.cpp:3741-3749); any other name = Alpine */
    MCF_SNOWENV_ALPINE = 0, MCF_SNOWENV_MARITIME = 1, MCF_SNOWENV_PRAIRIE = 2, MCF_SNOWENV_TUNDRA = 3,
    MCF_SNOWENV_TAIGA = 4
};
/* Maps the reference's `snowenv` string to the enum (exact, case-sensitive match as in the reference). */
int32_t mcf_snowenv_from_name(const char *name);

/* climdata columns temp, relhum, pres, swdown, difrad, lwdown, windspeed, winddir, precip
 * (src/microclimfCpp.cpp:4206-4214; gridmicrosnow2 names the last one "prec", :5083) and, for
 * gridmicrosnow only, umu (:4911). */
typedef struct mcf_snow_climate {
    const double *temp, *relhum, *pres, *swdown, *difrad, *lwdown, *windspeed, *winddir, *precip, *umu;
} mcf_snow_climate;
/* pointm of gridmodelsnow (src/microclimfCpp.cpp:4181-4186): output of pointmodelsnow.  `tr` is
 * only used for its length (ndays = tr.size()/24, :4231) and is not part of this ABI. */
typedef struct mcf_snow_pointm {
    const double *Gp, *Tc, *RswabsG, *RlwabsG, *umu;
} mcf_snow_pointm;
/* vegp (src/microclimfCpp.cpp:4188-4191; gridmicrosnow adds paia, leafd, leafden :4913-4919). */
typedef struct mcf_snow_vegp {
    const double *pai, *hgt, *leaft, *clump; /* [rows,cols] */
    const double *paia, *leafd, *leafden;    /* gridmicrosnow only */
} mcf_snow_vegp;
/* other (src/microclimfCpp.cpp:4193-4204, :4921-4929). */
typedef struct mcf_snow_other {
    const double *slope, *aspect, *skyview; /* [rows,cols]                                   */
    const double *wsa;                      /* [rows,cols,8]                                 */
    const double *hor;                      /* [rows,cols,24]                                */
    double lat, lon;                        /* vector forcing                                */
    const double *lats, *lons;              /* array forcing, [rows,cols]                    */
    double zref;
    const double *isnowdc, *isnowdg;        /* gridmodelsnow: initial snow depth (m)         */
    const int32_t *isnowac, *isnowag;       /* gridmodelsnow: initial snow age (h), IntegerMatrix */
    const double *Smax;                     /* gridmicrosnow: written to soilm under snow    */
} mcf_snow_other;
typedef struct mcf_snow_inputs {
    int64_t rows, cols, tsteps;
    int32_t array_forcing;
    int32_t snowenv; /* MCF_SNOWENV_* (gridmodelsnow only) */
    mcf_obstime obstime;
    mcf_snow_climate clim;
    mcf_snow_pointm pointm; /* gridmodelsnow only */
    mcf_snow_vegp vegp;
    mcf_snow_other other;
} mcf_snow_inputs;
/* Returned list of gridmodelsnow (src/microclimfCpp.cpp:4412-4422): Tc, Tg, sdepc, sdepg, sden are
 * [rows,cols,tsteps]; agec, ageg, meltc, meltg are [rows,cols].  Any pointer may be NULL (not
 * wanted).  Cells with NA hgt hold NA_real_ everywhere. */
typedef struct mcf_snowmodel_out {
    double *Tc, *Tg, *sdepc, *sdepg, *sden;
    double *agec, *ageg, *meltc, *meltg;
} mcf_snowmodel_out;
int mcf_gridmodelsnow1(const mcf_snow_inputs *in, mcf_snowmodel_out *out, int32_t device);
int mcf_gridmodelsnow2(const mcf_snow_inputs *in, mcf_snowmodel_out *out, int32_t device);

/* snowm list of gridmicrosnow (src/microclimfCpp.cpp:4935-4939), each [rows,cols,tsteps]. */
typedef struct mcf_snowm {
    const double *Tc, *Tg, *totalSWE, *groundsnowdepth, *snowden;
} mcf_snowm;
/* `micro` holds the no-snow solver's output on entry and is updated IN PLACE for every cell-step
 * with totalSWE > 0 (src/microclimfCpp.cpp:4993-5038); only variables with out[v] != 0 are read
 * or written, their pointers must then be non-NULL. */
int mcf_gridmicrosnow1(const mcf_snow_inputs *in, const mcf_snowm *snowm, double reqhgt, double mat,
                       const int32_t out[MCF_NOUT], mcf_outputs *micro, int32_t device);
int mcf_gridmicrosnow2(const mcf_snow_inputs *in, const mcf_snowm *snowm, double reqhgt, double mat,
                       const int32_t out[MCF_NOUT], mcf_outputs *micro, int32_t device);

/* The chunk loop of `.snowmodel1` (R/internal.R:2553-2617) resident on the device: for every chunk of
 * `chunk_steps` hours (5 days) the terrain inputs are re-derived from dtm + ground snow depth
 * (slope/aspect with NA -> 0/180, hor x24, sky view, wsa with s = 10 if res <= 100 else 1:
 * R/internal.R:2566-2580 — the kernels of mcf_precompute_terrain), gridmodelsnow1 runs on the chunk
 * (:2587), snow-depth changes are redistributed by the topographic position index `.tpicalc`
 * (:2471-2485, 2589-2600) and depths / ages are fed back (:2607-2612).  Only the requested series cross
 * PCIe.  `base` carries obstime / clim / pointm for the whole series (vector forcing), vegp, and
 * other.{lat, lon, zref, isnowdc, isnowdg, isnowac, isnowag}; other.{slope, aspect, skyview, wsa, hor}
 * are ignored.  As in R, n5days = tsteps / chunk_steps chunks are run (`1:n5days` truncates; at least
 * one); later steps stay NA.  Kept on purpose: other$isnowdg is never updated inside the loop
 * (every chunk restarts the ground layer from the initial depth) and each chunk restarts the albedo
 * clock (gridmodelsnow1 calls snowalbCpp on its own slice).  terra's aggregate/resample inside
 * .tpicalc are restated as in mcf_precompute_terrain (block means from the top-left, bilinear between
 * block centres). */
typedef struct mcf_snowdriver_in {
    mcf_snow_inputs base;
    const double *dtm;      /* [rows,cols] elevations (m), NaN = NA                        */
    double res;             /* cell size (m)                                              */
    double tfact;           /* .tpicalc's tfact (runsnowmodel default 0.02)               */
    int32_t chunk_steps;    /* 0 -> 120                                                   */
    /* Array weather only (base.array_forcing != 0: `.snowmodel2`'s loop, R/internal.R:2950-3008; mcf_snowmodel2 and the
     * array-weather snow run).  af_wsa_s: the wind-shelter smoothing factor — `.snowmodel2` takes 10 only when res <= 100 AND
     * the coarse weather grid has at least ten cells a side (:2963-2964), which the fine arrays here no longer say; 0: as
     * `.snowmodel1` (10 if res <= 100 else 1).  af_wind: [tsteps] sqrt(wuv^2 + wvv^2) of the coarse wind components' spatial
     * means (:2907-2908), whose chunk means set the aggregation factor of the position index (:2981-2984, at least 2). */
    int32_t af_wsa_s;
    const double *af_wind;
} mcf_snowdriver_in;
/* Returned list of .snowmodel1 (R/internal.R:2619): each [rows,cols,tsteps] or NULL. */
typedef struct mcf_snowdriver_out {
    double *Tc, *Tg, *groundsnowdepth, *totalSWE, *snowden;
} mcf_snowdriver_out;
int mcf_snowmodel1(const mcf_snowdriver_in *in, mcf_snowdriver_out *out, int32_t device);
/* The chunk loop of `.snowmodel2` (R/internal.R:2950-3008: the part behind the resampling of the coarse arrays), device-resident:
 * `base` carries array weather at the raster's resolution — clim.{temp, relhum, pres, swdown, difrad, lwdown, windspeed, precip}
 * and pointm.{Gp, Tc, RswabsG, RlwabsG, umu} as [rows,cols,tsteps], clim.winddir [tsteps], other.{lats, lons} [rows,cols] —
 * what gridmodelsnow2 takes; a chunk's slices are uploaded as the loop reaches them (13 x 8 B per cell-step over PCIe: this is
 * the boundary the reference's own binding has).  Differences to the data.frame loop kept as the reference has them: the
 * aggregation factor from af_wind with a floor of 2, af_wsa_s.  One block (no `_multi` form yet). */
int mcf_snowmodel2(const mcf_snowdriver_in *in, mcf_snowdriver_out *out, int32_t device);
/* The same loop over several devices from ONE process (the companion of mcf_runmicro1_multi for BASELINE configs[4]): the
 * raster in `n_blocks` contiguous row blocks, block b a snow plan on devices[b % n_devices] driven by that device's host
 * thread; per chunk the blocks' snow surfaces meet in one host array (each block takes its halo rows from it) and the two
 * raster-wide means are formed from the blocks' (sum, count) in block order — the exchanges of the stepwise API below,
 * done in host memory.  Equal to mcf_snowmodel1 up to the summation order of those two means (1e-9 in the tests); a block's
 * series reach the caller's arrays chunk by chunk, downloaded through the row pitch straight into their rows.  n_devices = 0:
 * every visible device; n_blocks = 0: one per device.  Unlike the solver's _multi entries, EVERY block's snow plan stays
 * resident for the whole run (the chunk loop couples the blocks at every chunk): more blocks than devices do not lower a
 * device's memory footprint here, they only share its time. */
int mcf_snowmodel1_multi(const mcf_snowdriver_in *in, mcf_snowdriver_out *out, const mcf_multi *multi);

/* The same loop one step at a time, for a row block of a tiled raster (one plan per rank): between the
 * steps the caller exchanges what couples the blocks — halo rows of the snow surface for the terrain
 * stencil and .tpicalc's block means (point-to-point), and two raster-wide means as (sum, count)
 * all-reduces.  mcf_snowmodel1 is this sequence with one block.  `in->base` and `in->dtm` describe the own
 * rows; row0 / rows_total place them (rows_total = 0: the block is the raster).  Per chunk c:
 *   mcf_snowplan_surface()          own rows of dtm + ground snow depth          -> halo exchange
 *   mcf_snowplan_surface_partial()  (sum, count) of it over non-NA cells         -> all-reduce (only used when
 *                                   round(10*sqrt(mean wind)/res) >= min(dim)/2, .tpicalc's fallback)
 *   mcf_snowplan_prepare_chunk(c, ext, hn, hs, surface_mean, &s, &n)
 *                                   ext = [hn + rows + hs, cols] surface with halos (NULL, 0, 0: no
 *                                   neighbours); terrain refresh + tpic; returns the partial (sum, count)
 *                                   of tpic                                       -> all-reduce
 *   mcf_snowplan_run_chunk(c, tpic_mean, out)   gridmodelsnow1 on the chunk, redistribution, hand-over;
 *                                   copies the chunk into the block's [rows, cols, tsteps] host arrays
 * Halo needed: 100 + 2.5 s rows (s = 10 if res <= 100 else 1) and whole af x af blocks around the own
 * rows, or every row up to the raster edge; checked. */
typedef struct mcf_snowplan mcf_snowplan;
int mcf_snowplan_create(const mcf_snowdriver_in *in, int64_t row0, int64_t rows_total, int32_t device,
                        mcf_snowplan **plan);
void mcf_snowplan_destroy(mcf_snowplan *plan);
int32_t mcf_snowplan_chunks(const mcf_snowplan *plan);
int mcf_snowplan_surface(mcf_snowplan *plan, double *host_own);
/* The pack depth the loop hands from the chunk just run to the next one (`other$isnowdc <- (asc + cdsnow + dsnow2)[,,last]`,
 * R/internal.R:2607), own rows, [rows, cols].  Once a pack has melted this is a rounding residue (0 or +-1e-17 m) and
 * `sdepcp > 0` (src/microclimfCpp.cpp:4337) decides on it whether the next chunk runs the model: the reference's loop is
 * discontinuous in its own rounding there.  Exposed so that a checker can tell such ill-conditioned hand-overs from
 * wrong ones (tests/test_snow_gpu.py::test_full_year_chunk_loop_*). */
int mcf_snowplan_handover(mcf_snowplan *plan, double *host_isnowdc);
int mcf_snowplan_surface_partial(mcf_snowplan *plan, double *sum, double *count);
int mcf_snowplan_prepare_chunk(mcf_snowplan *plan, int32_t chunk, const double *ext, int32_t halo_north,
                               int32_t halo_south, double surface_mean, double *tpic_sum, double *tpic_count);
/* The same exchange without host staging (one process per GPU, RCCL send / recv on device buffers): pack_halo copies the own
 * block's first `rows_north` / last `rows_south` surface rows into the caller's DEVICE buffers as column-major
 * [rows_north, cols] / [rows_south, cols] pieces — what the neighbouring ranks receive as their halos — and returns when they
 * are complete; prepare_chunk_dev takes the received pieces as device pointers and puts [north; own; south] together on the
 * device.  Same results as the host-pointer pair above, bit for bit. */
int mcf_snowplan_pack_halo(mcf_snowplan *plan, int32_t rows_north, double *d_north, int32_t rows_south, double *d_south);
int mcf_snowplan_prepare_chunk_dev(mcf_snowplan *plan, int32_t chunk, const double *d_halo_north, int32_t halo_north,
                                   const double *d_halo_south, int32_t halo_south, double surface_mean, double *tpic_sum,
                                   double *tpic_count);
int mcf_snowplan_run_chunk(mcf_snowplan *plan, int32_t chunk, double tpic_mean, mcf_snowdriver_out *out);
/* ... with the chunk's series written into row blocks of taller column-major host arrays (`row_pitch` rows per column; 0 =
 * dense): one strided DMA per series straight into the caller's rows, no block-sized host buffer (mcf_snowmodel1_multi). */
int mcf_snowplan_run_chunk_pitched(mcf_snowplan *plan, int32_t chunk, double tpic_mean, const mcf_snowdriver_out *out, int64_t row_pitch);
/* `.tpicalc`'s aggregation factor of a chunk, round(10 * sqrt(mean wind speed of the chunk) / res) (R/internal.R:2589-2590):
 * what sizes the halo a row block needs around its rows (100 + 3 s + 2 af rows, or all rows up to the raster edge). */
int mcf_snowplan_chunk_af(const mcf_snowplan *plan, int32_t chunk, int32_t *af);
/* applycpp3 (src/microclimfCpp.cpp:5553-5588) of the totalSWE series of the chunk just run, straight from the device:
 * result / count [steps of that chunk] as mcf_applycpp3 gives them.  `.runmicrosnow1` decides snow / no-snow days on the
 * per-step minimum and maximum of totalSWE (R/internal.R:3592-3594); row-block ranks combine them with one all-reduce. */
int mcf_snowplan_apply3(mcf_snowplan *plan, int32_t chunk, int32_t fun, double *result, double *count);

/* ---- the snow-day microclimate inside the chunk loop: `.runmicrosnow1` (R/internal.R:3581-3659) device-resident ----------
 * The reference runs the no-snow solver on the days with a snow-free cell somewhere, gridmicrosnow1
 * (src/microclimfCpp.cpp:4894-5056) on the days with snow somewhere, and merges by day, with the whole year's snow series in
 * memory.  Here the series exist one 5-day chunk at a time, and gridmicrosnow1 needs two things of the WHOLE snow-day series
 * before its first step — the per-cell mean snow damping depth (cpp:4713-4737) and the day list itself (its albedo clock and
 * maximum temperature run over the subset) — so the year is walked twice:
 *   pass 1  chunk loop as before (prepare_chunk, run_chunk, apply3 -> snowdaysfun); per chunk
 *           mcf_snowplan_meand_accumulate(chunk, snowday[]) adds the chunk's snow days to the running sum
 *   between mcf_snowplan_micro_setup(subset inputs, day map, reqhgt, mat, out mask): the snow-day subset's weather (incl. umu),
 *           `.sortl2` vegetation, bare-ground terrain, Smax; builds its step table; finishes the mean damping depth;
 *           mcf_plan_set_mxtc(solver plan, max temperature of the NO-snow subset) (cpp:2159-2168 over the subset the
 *           reference hands to runmicro1Cpp); mcf_snowplan_reset() puts the hand-over state back to the series' start
 *   pass 2  chunk loop again; per chunk the solver on the chunk's no-snow days at their own place in the ring slot
 *           (mcf_plan_run_days_at), then mcf_snowplan_microsnow(solver plan, chunk, slot, nosnowday[]) writes the snow
 *           microclimate over it: snow-covered cell-steps of snow days get gridmicrosnow1's values, snow-free cell-steps keep
 *           the solver's value when the day is a no-snow day as well and are NA otherwise (the reference's blank template).
 * The slot then holds `.runmicrosnow1`'s merged output for the chunk's days. */
int mcf_snowplan_reset(mcf_snowplan *plan);
/* Pass 2 need not walk the whole series again: mcf_snowplan_checkpoint(chunk), called in pass 1 BEFORE the chunk's
 * prepare_chunk, keeps the state the chunk starts from (pack depth handed over, snow surface, the two ages: 24 bytes per cell)
 * on the device; mcf_snowplan_restore(chunk) puts it back.  A chunk without a snow day contributes only the no-snow solver's
 * days to the merged output, so pass 2 restores and re-runs the chunks that hold one and skips the others.  (On a row-tiled
 * raster every rank takes the same decision: the day classes come from all-reduced extremes.) */
/* Sparse read-back of the plan's device arrays for n cells (0-based, column-major index within the block): out is
 * [n, depth] column-major, depth = 1 (hand-over state, slope, aspect, sky view), 8 (wind shelter), 24 (horizons) or the
 * chunk's steps (the series of the chunk run last; MCF_SNOWPLAN_TOTALSWE after the redistribution).  For in-run checks of a
 * sample of cells against a CPU model (bench.py --config 4); no counterpart in the reference. */
enum {
    MCF_SNOWPLAN_ISNOWDC = 0, MCF_SNOWPLAN_ISNOWAC = 1, MCF_SNOWPLAN_ISNOWAG = 2, MCF_SNOWPLAN_SLOPE = 3, MCF_SNOWPLAN_ASPECT = 4,
    MCF_SNOWPLAN_SKYVIEW = 5, MCF_SNOWPLAN_WSA = 6, MCF_SNOWPLAN_HOR = 7, MCF_SNOWPLAN_TC = 8, MCF_SNOWPLAN_TG = 9,
    MCF_SNOWPLAN_SDEPG = 10, MCF_SNOWPLAN_SDEN = 11, MCF_SNOWPLAN_TOTALSWE = 12
};
int mcf_snowplan_fetch_cells(mcf_snowplan *plan, int32_t what, const int64_t *cells, int32_t n, double *out, int32_t *depth);
/* ... and need not re-run a chunk whose series still fit the device: mcf_snowplan_keep_chunk(chunk, reserve_bytes, &kept),
 * called in pass 1 after run_chunk and everything that reads the chunk's series (apply3, meand_accumulate), hands the chunk's
 * five series buffers over to a cache and gives the plan fresh ones — as long as `reserve_bytes` of device memory stay free
 * (kept = 0 otherwise; a year of one rank's block of configs[4] is 38 snow chunks x 10 GB against 288 GB of HBM).
 * mcf_snowplan_microsnow reads a kept chunk where it lies: pass 2 skips restore / prepare_chunk / run_chunk for it.
 * mcf_snowplan_release_kept (before the next year's pass 1) returns the sets to a pool the next year draws from: allocating
 * 10 GB takes about 0.25 s, more than re-running the chunk — the cache pays from a plan's second year on. */
int mcf_snowplan_keep_chunk(mcf_snowplan *plan, int32_t chunk, int64_t reserve_bytes, int32_t *kept);
/* Would mcf_snowplan_keep_chunk keep a chunk now (a pooled set, or room for a new one beside reserve_bytes)? */
int mcf_snowplan_can_keep(mcf_snowplan *plan, int64_t reserve_bytes, int32_t *yes);
/* A ceiling on what mcf_snowplan_keep_chunk may ALLOCATE over the plan's lifetime (pooled sets count once); < 0: none. */
int mcf_snowplan_set_keep_budget(mcf_snowplan *plan, int64_t bytes);
/* Which of the five device series the following mcf_snowplan_run_chunk calls write: bit 0 Tc, 1 Tg, 2 totalSWE, 3 ground snow
 * depth, 4 snow density (default 31).  Pass 1 of the two-pass run needs only totalSWE (the day classes) and the density (the
 * mean damping depth) of a chunk that will not stay in HBM — pass 2 re-runs it —, and the five stores per cell-step are what a
 * snow-free chunk costs.  A series that is off cannot be asked for in run_chunk's host outputs; a chunk run with series off
 * cannot be kept or handed to mcf_snowplan_microsnow (MCF_ERR_STATE). */
int mcf_snowplan_set_series(mcf_snowplan *plan, uint32_t mask);
int mcf_snowplan_release_kept(mcf_snowplan *plan);
int mcf_snowplan_checkpoint(mcf_snowplan *plan, int32_t chunk);
int mcf_snowplan_restore(mcf_snowplan *plan, int32_t chunk);
int mcf_snowplan_meand_accumulate(mcf_snowplan *plan, int32_t chunk, const int32_t *snowday /* [days of the chunk] */);
/* reuse_static != 0: the vegetation, terrain and Smax matrices of the previous set-up stay (only the series and the day map
 * are new). */
int mcf_snowplan_micro_setup(mcf_snowplan *plan, const mcf_snow_inputs *subset, const int32_t *subset_day_of_day,
                             int32_t ndays, double reqhgt, double mat, const int32_t out[MCF_NOUT], int32_t reuse_static);
/* Pass 2, before the solver's run of the chunk's days [day, day + ndays) (days of the chunk, 0-based): skip_tile[t] = 1
 * where every cell of the solver plan's tile t has a vegetation height and a snow water equivalent > 0 at every step of
 * those days — mcf_snowplan_microsnow will overwrite all of the tile's values of these days —, 0 elsewhere; all 0 when the
 * solver plan holds an output gridmicrosnow1's `out` mask leaves to the solver.  *n_covered = number of ones. */
int mcf_snowplan_covered_tiles(mcf_snowplan *plan, mcf_plan *solver, int32_t chunk, int32_t day, int32_t ndays,
                               uint8_t *skip_tile, int64_t n_tiles, int64_t *n_covered);
/* The same question per cell, answered on the device: (*need_cell)[c] = 1 where cell c is NOT under snow at every step of
 * those days, or has no vegetation height (gridmicrosnow1 skips it) — the cells whose solver values survive the merge —, 0
 * elsewhere; all 1 when the solver plan holds an output gridmicrosnow1's `out` mask leaves to the solver.  *need_cell is device
 * memory of the snow plan (valid until the next call, complete on return), for mcf_plan_run_days_cells; *n_need = number of
 * ones. */
int mcf_snowplan_free_cells(mcf_snowplan *plan, mcf_plan *solver, int32_t chunk, int32_t day, int32_t ndays,
                            const uint8_t **need_cell, int64_t *n_need);
int mcf_snowplan_microsnow(mcf_snowplan *plan, mcf_plan *solver, int32_t chunk, int32_t slot,
                           const int32_t *nosnowday /* [days of the chunk] */);

/* ---- `runmicro(..., snow = TRUE)` for data.frame weather as ONE call: `.snowmodel1` + `.runmicrosnow1` -----------------------
 * The reference's `.runmicrosnow1(micropoint, reqhgt, vegp, soilc, dtm, smod, ...)` (R/internal.R:3581-3659, called from
 * runmicro at R/Cppwrappers.R:382) takes `smod` — five [rows, cols, tsteps] arrays that `runsnowmodel` -> `.snowmodel1`
 * (R/internal.R:2498-2619) produced — from host memory, runs the grid solver on the days with a snow-free cell somewhere
 * (`.runmicronosnow` on subsetpointmodel(days = nosnowdays), :3603-3606), gridmicrosnow1 on the days with snow somewhere
 * (:3612-3625) and merges by day (:3633-3656).  This entry is that whole sequence with the snow series kept on the device:
 * the chunk loop of `.snowmodel1` is driven here (mcf_snowplan_*), the two models meet in the solver's output ring, and only
 * the merged [rows, cols, tsteps] outputs — and the snow series too, if `smod` asks for them — cross PCIe.  An R session
 * reaches it through r/mcfhip_overrides.R (runsnowmodel returns a light handle instead of the arrays, `.runmicrosnow1` passes
 * it on).
 *   grid    what `.runmicronosnow` -> runmicro1Cpp would get for EVERY day of the series (vector forcing); with time-varying
 *           vegetation (veg_layers > 1: runmicro3Cpp) the WHOLE-series layer table — a no-snow day runs with the layer it has
 *           in the whole series, which is what `.runmodel3Cpp` on the day subset comes to (R/internal.R:252-270, 1391-1399)
 *   snow    `.snowmodel1`'s inputs as for mcf_snowmodel1 (obstime / climate / pointmodelsnow output for the whole series, the
 *           `.sortl` vegetation, initial depths and ages, dtm, res, tfact)
 *   micro   gridmicrosnow1's inputs as `.prepsnowinputs1` (R/internal.R:3375-3443) makes them, but for the WHOLE series
 *           (tsteps = the series' length; the entry takes the snow-day subset itself): obstime, climate incl. umu =
 *           smod$umu, `.sortl2` vegetation, bare-ground terrain (slope, aspect, skyview, wsa, hor), lat, lon, zref, Smax.
 *           May be NULL only if the year has no snow day.
 *   mat     micropoint$matemp
 * `opt`: reqhgt >= 0 (below ground the reference smooths whole series: use mcf_runmicro1 + mcf_gridmicrosnow1 on host
 * arrays), out[] as for mcf_runmicro1; with reqhgt == 0 gridmicrosnow1 is given the reference's fixed mask (:3616-3619).
 * `out`: [rows, cols, tsteps] per requested variable; `smod` (optional, members may be NULL): `.snowmodel1`'s returned arrays.
 * Days that are in neither class (max totalSWE <= 0 and min != 0: a melted pack's negative rounding residue) are NA in
 * `out` — the reference's merge indexes past its arrays there (:3650-3655).  Steps past the last whole 5-day chunk are
 * no-snow days, as `.runmicrosnow1`'s NA -> 0 makes them (:3586).
 * The staged form, for callers whose gridmicrosnow1 inputs depend on the day classes (`.sortl2` weights time-varying
 * vegetation layers by the snow-covered steps): create -> pass1 (returns snowday[] / nosnowday[], one flag per day) -> pass2. */
typedef struct mcf_microsnow_in {
    const mcf_grid_inputs *grid;
    const mcf_snowdriver_in *snow;
    const mcf_snow_inputs *micro;
    double mat;
} mcf_microsnow_in;
int mcf_runmicrosnow1(const mcf_microsnow_in *in, const mcf_options *opt, mcf_outputs *out, const mcf_snowdriver_out *smod);
/* The same run with ARRAY weather: `.snowmodel2`'s loop (R/internal.R:2950-3008) + `.runmicrosnow2` (:3661-3745).  Everything is
 * at the raster's resolution, as the reference's bindings take it after `.cca` / resample:
 *   grid    runmicro2Cpp's arguments for every day (array_forcing = 1: climate and point-model arrays [rows,cols,tsteps], lats / lons)
 *   snow    mcf_snowmodel2's input (gridmodelsnow2's arrays, af_wind, af_wsa_s)
 *   micro   gridmicrosnow2's inputs for the WHOLE series (array_forcing = 1: weather arrays incl. umu, winddir [tsteps], lats / lons)
 *   mat     the point models' mean annual temperature (`matemp`, :3682-3687)
 * A chunk's slices of the snow model's arrays, its no-snow days of the solver's fifteen and its snow days of the microclimate's nine
 * are uploaded as the loops reach them (the boundary is PCIe-bound by construction: 8 B x 13 / 15 / 9 per cell-step); the solver's
 * per-cell temperature cap is taken over the no-snow days (mcf_plan_set_mxtc_days), the microclimate's over the snow days.  One
 * block on one device (no _multi form); the staged entries below take either geometry. */
int mcf_runmicrosnow2(const mcf_microsnow_in *in, const mcf_options *opt, mcf_outputs *out, const mcf_snowdriver_out *smod);
/* The same over row blocks on several devices from one process (mcf_multi as for mcf_runmicro1_multi; equal-row blocks as
 * mcf_snowmodel1_multi): per chunk the blocks' snow surfaces meet in one host array, the raster-wide means (snow surface, tpi,
 * the solver's twi mean) and the per-step extremes of totalSWE are combined in block order.  One block: bit for bit
 * mcf_runmicrosnow1; more blocks: equal up to the summation order of those means. */
int mcf_runmicrosnow1_multi(const mcf_microsnow_in *in, const mcf_options *opt, const mcf_multi *multi, mcf_outputs *out,
                            const mcf_snowdriver_out *smod);
typedef struct mcf_snowrun mcf_snowrun;
/* `in->micro` and `in->mat` are not read here; multi = NULL: one block on opt->device.  The caller's arrays must stay valid
 * until mcf_snowrun_destroy. */
int mcf_snowrun_create(const mcf_microsnow_in *in, const mcf_options *opt, const mcf_multi *multi, mcf_snowrun **run);
void mcf_snowrun_destroy(mcf_snowrun *run);
int32_t mcf_snowrun_days(const mcf_snowrun *run);     /* tsteps / 24 */
/* What pass 2 was spared (diagnostics; tests assert which path ran): stats[0] tile-days of the solver's days that are snow days as
 * well, [1] of them left out (tiles wholly under snow, mcf_plan_run_days_masked; or, where few cells are not under snow, all tiles
 * but the ones those cells fill, mcf_plan_run_days_cells), [2] snow chunks whose series had stayed in HBM,
 * [3] snow chunks re-run from their checkpoints. */
int mcf_snowrun_stats(const mcf_snowrun *run, int64_t stats[4]);
/* Keep pass 1's snow chunks in device memory for pass 2, up to `bytes` in all (0: off, the default — on a fresh handle the
 * allocation costs more than re-running the chunks).  The sets are pooled in the handle: a second period on the same handle
 * (mcf_snowrun_pass1 again) allocates nothing and pass 2 re-runs only what did not fit.  Outputs are bit for bit the unkept
 * run's.  Reference: none (`.runmicrosnow1`, R/internal.R:3581-3659, holds the whole series in host memory). */
int mcf_snowrun_keep(mcf_snowrun *run, int64_t bytes);
/* snowday / nosnowday: [mcf_snowrun_days] or NULL */
int mcf_snowrun_pass1(mcf_snowrun *run, const mcf_snowdriver_out *smod, int32_t *snowday, int32_t *nosnowday);
int mcf_snowrun_pass2(mcf_snowrun *run, const mcf_snow_inputs *micro, double mat, mcf_outputs *out);

/* applycpp3 (src/microclimfCpp.cpp:5553-5588; `.runmicrosnow1/2` use it on totalSWE, R/internal.R:3592-3593):
 * reduction of a [rows,cols,tsteps] array over space, per time step, skipping NA.  fun: 0 mean, 1 sum,
 * 2 max, 3 min (max / min of an all-NA step: -Inf / +Inf, mean: NaN).  `count` (optional, [tsteps])
 * receives the number of non-NA cells so that row blocks of a tiled raster can be combined: sum and
 * count add, max / min combine by max / min (microclimf_amd/distributed.py).  At most 65535 steps per call. */
enum { MCF_APPLY_MEAN = 0, MCF_APPLY_SUM = 1, MCF_APPLY_MAX = 2, MCF_APPLY_MIN = 3 };
int mcf_applycpp3(const double *a, int64_t rows, int64_t cols, int64_t tsteps, int32_t fun, double *result,
                  double *count, int32_t device);

/* ---- host-side point model ---------------------------------------------------------
 * The O(tsteps) serial series that produce the grid solver's `pointm` in the reference:
 *   mcf_soilm()          soilmCpp        src/microclimfCpp.cpp:931-972   (_microclimf_soilmCpp)
 *   mcf_bigleaf()        BigLeafCpp      src/microclimfCpp.cpp:710-881   (_microclimf_BigLeafCpp)
 *   mcf_pointmprocess()  pointmprocess   src/microclimfCpp.cpp:5265-5323 (_microclimf_pointmprocess)
 *   mcf_weatherhgt()     weatherhgtCpp   src/microclimfCpp.cpp:884-929   (_microclimf_weatherhgtCpp)
 * as `runpointmodel` chains them (R/Cppwrappers.R:119-138).  They run on the HOST, as in the reference: one
 * point, iterated over the whole series with running means — nothing to parallelise and not part of the GPU hot
 * path (which has no CPU implementation).  vegp / groundp are read positionally as the reference does:
 * vegp = (h, pai, x, clump, lref, ltra, leafd, em, gsmax, q50), groundp = (gref, slope, aspect, em, rho, Vm, Vq,
 * Mc, b, psi_e, Smax, Smin).  One guard the reference lacks: its circular means index outside their arrays when
 * the window is longer than the series; mcf_bigleaf rejects yearG for 2..89 days and series under 6 steps. */
typedef struct mcf_point_weather {
    const double *temp, *relhum, *pres, *swdown, *difrad, *lwdown, *windspeed, *precip; /* [n]; precip: soilm only */
} mcf_point_weather;
typedef struct mcf_bigleaf_out { /* caller-allocated [n] each (src/microclimfCpp.cpp:867-879) */
    double *Tc, *Tg, *H, *G, *psih, *psim, *phih, *OL, *uf, *RabsG, *albedo;
    double err;
    int32_t iters;
} mcf_bigleaf_out;
int mcf_bigleaf(int64_t n, const mcf_obstime *obstime, const mcf_point_weather *weather, const double *vegp,
                const double *groundp, const double *soilm, double lat, double lon, double dTmx, double zref,
                int32_t maxiter, double bwgt, double tol, int32_t yearG, mcf_bigleaf_out *out);
int mcf_soilm(int64_t n, const mcf_point_weather *weather, double rmu, double mult, double pwr, double Smax,
              double Smin, double Ksat, double a, double *soilm_days /* [n / 24] */, int64_t *ndays);
int mcf_pointmprocess(int64_t n, const double *windspeed, const double *tc, const double *rh, const double *pk,
                      const double *uf, const double *soilm, const double *RabsG, double zref, double h, double pai,
                      double rho, double Vm, double Vq, double Mc, double *umu, double *kp, double *muGp,
                      double *DDp, double *T0p, double *dtrp);
int mcf_weatherhgt(int64_t n, const mcf_obstime *obstime, const mcf_point_weather *weather, double zin, double uzin,
                   double zout, double lat, double lon, double *temp, double *relhum, double *windspeed);
/* _microclimf_pointmodelsnow (src/microclimfCpp.cpp:4000-4169; called by `.snowmodel1`, R/internal.R:2536): the snow
 * branch's point model.  vegp = (pai, hgt, ltra, clump), other = (slope, aspect, lat, lon, zref, initial depth,
 * initial age); snowenv: MCF_SNOWENV_*; the reference's defaults are tol = 0.5, maxiter = 100.  Outputs: [n] each,
 * sdepc / sdepg [n + 1] (the depth after the last step is kept, as in the reference).  Host code. */
typedef struct mcf_pointsnow_out {
    double *Tc, *Tg, *sdepc, *sdepg, *sdenc, *sdeng, *G, *RswabsG, *RlwabsG, *tr, *umu, *sublmelt, *tempmelt, *rainmelt,
        *sstemp;
    double mxdif;
    int32_t iters;
} mcf_pointsnow_out;
int mcf_pointmodelsnow(int64_t n, const mcf_obstime *obstime, const mcf_point_weather *weather, const double *vegp,
                       const double *other, int32_t snowenv, double tol, double maxiter, mcf_pointsnow_out *out);
/* ---- fast snow method for subset runs (R/internal.R:2627-2776 `.snowmodelq1`): its three kernels of arithmetic ----
 * mcf_canintfrac replaces _microclimf_canintfrac (src/microclimfCpp.cpp:5417-5450): frac = canopysnowintCpp(hgt, pai,
 *   uf, prec, tc, Li) / prec per cell, 0.5 everywhere when prec is not > 0, NaN where hgt is NA.  Host code.
 * mcf_meltmu replaces _microclimf_meltmu (:5454-5492): per cell sum over the n steps of max(0, (stemp - tc) * skyview +
 *   tc) over sum of max(0, stemp); 1 everywhere when the denominator is 0; NaN where skyview is NA.  Host code.
 * mcf_tpicalc replaces `.tpicalc` (R/internal.R:2483-2496) on the device: exp((aggregate(dtm, af, na.rm) resampled
 *   bilinearly - dtm) * tfact), or the raster mean in place of the aggregate when af >= min(dim) / 2, values below 0.05
 *   set to 0.1 and above 10 to 10 as in the reference, divided by the raster mean.  [rows, cols] column-major. */
int mcf_canintfrac(int64_t cells, const double *hgt, const double *pai, double uf, double prec, double tc, double Li,
                   double *frac);
int mcf_meltmu(int64_t cells, const double *skyview, int64_t n, const double *stemp, const double *tc, double *mu);
/* mcf_meltmu2 replaces _microclimf_meltmu2 (:5495-5527, `.snowmodelq2`): the same with stemp / tc per cell, [cells, n]
 * with the cell index fastest (an R array [rows, cols, n]); 0.5 where the denominator is 0.  Host code. */
int mcf_meltmu2(int64_t cells, int64_t n, const double *mu, const double *stemp, const double *tc, double *out);
int mcf_tpicalc(int64_t rows, int64_t cols, const double *dtm, int32_t af, double tfact, double *tpic, int32_t device);
/* manCpp (src/microclimfCpp.cpp:597-627): circular trailing mean, via daily means for windows beyond 48 steps. */
int mcf_man(int64_t n, const double *x, int32_t window, double *out);

/* ---- topographic wetness index (soilc$twi) ------------------------------------------------
 * mcf_flowacc replaces _microclimf_flowaccCpp (src/microclimfCpp.cpp:5368-5408, with flowdirCpp :5326-5366),
 * mcf_topidx the R function `.topidx` around it (R/internal.R:861-874).  Host code as in the reference: one
 * elevation-ordered sweep over the whole raster, run once per raster (see mcf_hydro.cpp for the reference
 * behaviours kept: self-pointing pits, the 9999.99 m ceiling, tie order, NA cells = -2147483648 in `fa`).
 * `dtm`, `fa`, `twi`: [rows, cols] column-major; NaN = NA. */
int mcf_flowacc(int64_t rows, int64_t cols, const double *dtm, double *fa);
int mcf_topidx(int64_t rows, int64_t cols, const double *dtm, double xres, double yres, double *twi);

/* Diagnostics: evaluate one of the solver's lean device elementary functions
 * elementwise on host arrays (kind 0 exp, 1 log, 2 x/y, 3 sqrt, 4 1/x, 5 satvap
 * (cpp:480-490), 6 x^y); used by tests to bound their error against libm. */
int mcf_selftest_math(int32_t kind, const double *x, const double *y, double *out, int64_t n,
                      int32_t device);

/* Information. */
int64_t mcf_plan_valid_cells(const mcf_plan *plan); /* cells with non-NA hgt */
int64_t mcf_plan_bytes(const mcf_plan *plan);       /* device bytes held      */

#ifdef __cplusplus
}
#endif
#endif /* MCF_H */
