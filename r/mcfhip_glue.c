/*
 * mcfhip_glue.c — R .Call() shim over libmcfhip's C ABI (include/mcf.h).
 *
 * Stands where the reference's generated Rcpp glue stands:
 *   _microclimf_runmicro1Cpp  src/RcppExports.cpp:250-272  ->  mcfhip_runmicro1
 *   _microclimf_runmicro2Cpp  src/RcppExports.cpp:275-297  ->  mcfhip_runmicro2
 *   _microclimf_runmicro3Cpp  src/RcppExports.cpp:300-322  ->  mcfhip_runmicro3  (dfsel + the same 15)
 *   _microclimf_runmicro4Cpp  src/RcppExports.cpp:326-348  ->  mcfhip_runmicro4
 *   _microclimf_gridmodelsnow1/2  src/RcppExports.cpp:483-514  ->  mcfhip_gridmodelsnow1/2
 *   _microclimf_gridmicrosnow1/2  src/RcppExports.cpp:542-578  ->  mcfhip_gridmicrosnow1/2
 * Same 15 arguments in the same order as R/RcppExports.R:72-78, same named-list
 * result (src/microclimfCpp.cpp:2326-2335).  Uses only R's C API (Rinternals.h);
 * no Rcpp.  NOT compiled in the build image (R is not installed there; tests/test_r_glue_syntax_cpu.py runs
 * `gcc -fsyntax-only` on it against declarations-only headers — a syntax check, nothing more): build with
 *   R CMD SHLIB mcfhip_glue.c -I../include -L../microclimf_amd/csrc -lmcfhip
 *
 * Error discipline: libmcfhip never longjmps; it returns a status and
 * has released every device resource by then, so Rf_error() is raised from here only
 * after the call has returned.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stddef.h>

#include "mcf.h"

/* The fill loops below walk these structs as arrays of `const double *` in the order of their name tables: a reorder or
 * an added field in include/mcf.h must fail the build, not shift the pointers silently. */
#define MCF_PTR_STRUCT(type, first, last, n)                                                           \
    _Static_assert(sizeof(type) == (n) * sizeof(const double *), #type " is not " #n " pointers");     \
    _Static_assert(offsetof(type, first) == 0, #type "." #first " is not the first field");            \
    _Static_assert(offsetof(type, last) == ((n) - 1) * sizeof(const double *), #type "." #last " is not the last field")
MCF_PTR_STRUCT(mcf_vegp, hgt, leafden, 10);
MCF_PTR_STRUCT(mcf_soilc, Smin, hor, 15);
MCF_PTR_STRUCT(mcf_snow_climate, temp, umu, 10);
MCF_PTR_STRUCT(mcf_snow_pointm, Gp, umu, 5);
MCF_PTR_STRUCT(mcf_snow_vegp, pai, leafden, 7);
_Static_assert(offsetof(mcf_vegp, paia) == 8 * sizeof(const double *), "mcf_vegp.paia");
_Static_assert(offsetof(mcf_soilc, slope) == 9 * sizeof(const double *), "mcf_soilc.slope");
_Static_assert(offsetof(mcf_soilc, wsa) == 13 * sizeof(const double *), "mcf_soilc.wsa");
_Static_assert(offsetof(mcf_snow_climate, precip) == 8 * sizeof(const double *), "mcf_snow_climate.precip");
_Static_assert(offsetof(mcf_snow_vegp, clump) == 3 * sizeof(const double *), "mcf_snow_vegp.clump");

static SEXP elt(SEXP list, const char *name, const char *alt) {
    SEXP names = getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < XLENGTH(list); ++i) {
        const char *n = CHAR(STRING_ELT(names, i));
        if (strcmp(n, name) == 0 || (alt && strcmp(n, alt) == 0)) return VECTOR_ELT(list, i);
    }
    Rf_error("mcfhip: list element '%s' not found", name);
    return R_NilValue;
}

/* Entry points whose C structs outlive the .Call (mcfhip_snowrun_create: the library reads the caller's arrays until the run
 * is destroyed) set g_keep to a list that then holds every vector a coercion below had to create.  EVERY entry that coerces
 * arguments clears it first: fill_inputs / fill_snowdriver can leave through Rf_error (a longjmp) while it is set, and the
 * list — reachable only from the unwound frame's external pointer — may be collected before the next .Call. */
static SEXP g_keep = NULL;
static int g_nkeep = 0;
static void keep(SEXP x) {
    if (!g_keep) return;
    if (g_nkeep >= LENGTH(g_keep)) Rf_error("mcfhip: too many coerced arguments to keep alive");
    SET_VECTOR_ELT(g_keep, g_nkeep++, x);
}
/* numeric column/array as double*; *np counts PROTECTs added by coercion */
static const double *dbl(SEXP x, int *np) {
    if (TYPEOF(x) != REALSXP) { x = PROTECT(coerceVector(x, REALSXP)); ++*np; keep(x); }
    return REAL(x);
}
static const int *intcol(SEXP x, int *np) {   /* obstime$year etc. arrive as doubles (int:1085) */
    if (TYPEOF(x) != INTSXP) { x = PROTECT(coerceVector(x, INTSXP)); ++*np; keep(x); }
    return INTEGER(x);
}

/* Fills in/opt from the R arguments shared by runmicro*Cpp and runbioclim*Cpp; *np counts PROTECTs. */
static void fill_inputs(mcf_grid_inputs *in, mcf_options *opt, int *np, int array_forcing, SEXP dfsel,
                        SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                        SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP complete, SEXP mat, SEXP out) {

    memset(in, 0, sizeof *in); memset(opt, 0, sizeof *opt);

    SEXP hgt = elt(vegp, "hgt", NULL);
    SEXP dim = getAttrib(hgt, R_DimSymbol);
    if (dfsel == R_NilValue) {
        if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 2) Rf_error("mcfhip: vegp$hgt must be a matrix");
    } else {
        /* runmicro3Cpp/4Cpp: vegetation arrays [rows, cols, layers] + dfsel (lyr, st, ed), cpp:2629-2640 */
        if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 3) Rf_error("mcfhip: vegp$hgt must be a 3-D array");
        in->veg_layers = INTEGER(dim)[2];
        if (XLENGTH(elt(dfsel, "st", NULL)) != in->veg_layers) Rf_error("mcfhip: dfsel rows != vegetation layers");
        in->lyr_st = intcol(elt(dfsel, "st", NULL), np);
        in->lyr_ed = intcol(elt(dfsel, "ed", NULL), np);
    }
    in->rows = INTEGER(dim)[0]; in->cols = INTEGER(dim)[1];
    in->tsteps = XLENGTH(elt(obstime, "year", NULL));
    in->array_forcing = array_forcing;
    in->obstime.year = intcol(elt(obstime, "year", NULL), np);
    in->obstime.month = intcol(elt(obstime, "month", NULL), np);
    in->obstime.day = intcol(elt(obstime, "day", NULL), np);
    in->obstime.hour = dbl(elt(obstime, "hour", NULL), np);
    /* climdata: data.frame columns (1Cpp, cpp:2062-2071) or list entries (2Cpp, cpp:2350-2359) */
    in->clim.tc = dbl(elt(climdata, "temp", "tc"), np);
    in->clim.es = dbl(elt(climdata, "es", NULL), np);
    in->clim.ea = dbl(elt(climdata, "ea", NULL), np);
    in->clim.tdew = dbl(elt(climdata, "tdew", NULL), np);
    in->clim.pk = dbl(elt(climdata, "pres", "pk"), np);
    in->clim.swdown = dbl(elt(climdata, "swdown", NULL), np);
    in->clim.difrad = dbl(elt(climdata, "difrad", NULL), np);
    in->clim.lwdown = dbl(elt(climdata, "lwdown", NULL), np);
    in->clim.windspeed = dbl(elt(climdata, "windspeed", NULL), np);
    in->clim.winddir = dbl(elt(climdata, "winddir", NULL), np);
    in->pointm.soilm = dbl(elt(pointm, "soilm", NULL), np);
    in->pointm.Tg = dbl(elt(pointm, "Tg", NULL), np);
    in->pointm.Tbp = dbl(elt(pointm, "Tbp", NULL), np);
    in->pointm.G = dbl(elt(pointm, "G", "Gp"), np);
    in->pointm.umu = dbl(elt(pointm, "umu", NULL), np);
    in->pointm.kp = dbl(elt(pointm, "kp", NULL), np);
    in->pointm.muGp = dbl(elt(pointm, "muGp", NULL), np);
    in->pointm.dtrp = dbl(elt(pointm, "dtrp", NULL), np);
    static const char *vn[10] = {"hgt", "pai", "x", "gsmax", "leafr", "leaft", "clump", "leafd", "paia", "leafden"};
    const double **vp = (const double **)&in->vegp;
    for (int i = 0; i < 10; ++i) vp[i] = dbl(elt(vegp, vn[i], NULL), np);
    static const char *sn[15] = {"Smin", "Smax", "gref", "soilb", "Psie", "Vq", "Vm", "Mc", "rho", "slope",
                                 "aspect", "twi", "svfa", "wsa", "hor"};
    const double **sp = (const double **)&in->soilc;
    for (int i = 0; i < 15; ++i) sp[i] = dbl(elt(soilc, sn[i], NULL), np);
    if (array_forcing) { in->lats = dbl(lat, np); in->lons = dbl(lon, np); }
    else { in->lat = asReal(lat); in->lon = asReal(lon); }

    opt->reqhgt = asReal(reqhgt); opt->zref = asReal(zref);
    opt->Sminp = asReal(Sminp); opt->Smaxp = asReal(Smaxp);
    opt->tfact = asReal(tfact); opt->mat = asReal(mat);
    opt->complete = asLogical(complete) == TRUE;
    /* `out` may be logical or numeric 0/1 after `out2*out` (int:1161) */
    SEXP outl = PROTECT(coerceVector(out, LGLSXP)); ++*np;
    if (LENGTH(outl) != MCF_NOUT) Rf_error("mcfhip: out must have 10 elements");
    for (int v = 0; v < MCF_NOUT; ++v) opt->out[v] = LOGICAL(outl)[v] == TRUE;
    /* Tg/Tbp are only read for reqhgt < 0 && !complete; the marshaller passes Tbp = 0 otherwise (int:1096) */
    if (!(opt->reqhgt < 0 && !opt->complete)) { in->pointm.Tg = NULL; in->pointm.Tbp = NULL; }

}

static SEXP run(int array_forcing, SEXP dfsel, SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc,
                SEXP reqhgt, SEXP zref, SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact,
                SEXP complete, SEXP mat, SEXP out) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_grid_inputs in;
    mcf_options opt;
    mcf_outputs res;
    memset(&res, 0, sizeof res);
    fill_inputs(&in, &opt, &np, array_forcing, dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                Sminp, Smaxp, tfact, complete, mat, out);
    static const char *on[MCF_NOUT] = {"Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown",
                                       "Rlwdown", "Rswup", "Rlwup"};
    int nreq = 0;
    for (int v = 0; v < MCF_NOUT; ++v) nreq += opt.out[v];
    SEXP ans = PROTECT(allocVector(VECSXP, nreq)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, nreq)); ++np;
    R_xlen_t n = (R_xlen_t)in.rows * in.cols * in.tsteps;   /* long vectors: no 2^31 overflow (cpp:2118) */
    for (int v = 0, k = 0; v < MCF_NOUT; ++v) {
        if (!opt.out[v]) continue;
        SEXP a = PROTECT(allocVector(REALSXP, n)); ++np;
        SEXP d = PROTECT(allocVector(INTSXP, 3)); ++np;
        INTEGER(d)[0] = (int)in.rows; INTEGER(d)[1] = (int)in.cols; INTEGER(d)[2] = (int)in.tsteps;
        setAttrib(a, R_DimSymbol, d);
        SET_VECTOR_ELT(ans, k, a);
        SET_STRING_ELT(nms, k, mkChar(on[v]));
        res.var[v] = REAL(a);
        ++k;
    }
    setAttrib(ans, R_NamesSymbol, nms);

    /* options(mcfhip.devices = c(0L, 1L, ...)) — set by mcfhip_enable(devices = ) — sends the solves through
     * the one-process multi-device entry points: row blocks dealt to the listed HIP devices, the same bits as one device
     * (include/mcf.h mcf_runmicro1_multi).  options(mcfhip.blocks = n): more row blocks than devices (time-sliced). */
    SEXP dv = GetOption1(install("mcfhip.devices"));
    int rc;
    if (dv != R_NilValue && LENGTH(dv) > 0) {
        SEXP dvi = PROTECT(coerceVector(dv, INTSXP)); ++np;
        SEXP nbo = GetOption1(install("mcfhip.blocks"));
        mcf_multi mu;
        mu.n_devices = LENGTH(dvi);
        mu.devices = INTEGER(dvi);
        mu.n_blocks = nbo == R_NilValue ? 0 : asInteger(nbo);
        rc = dfsel == R_NilValue
                 ? (array_forcing ? mcf_runmicro2_multi(&in, &opt, &mu, &res) : mcf_runmicro1_multi(&in, &opt, &mu, &res))
                 : (array_forcing ? mcf_runmicro4_multi(&in, &opt, &mu, &res) : mcf_runmicro3_multi(&in, &opt, &mu, &res));
    } else {
        rc = dfsel == R_NilValue
                 ? (array_forcing ? mcf_runmicro2(&in, &opt, &res) : mcf_runmicro1(&in, &opt, &res))
                 : (array_forcing ? mcf_runmicro4(&in, &opt, &res) : mcf_runmicro3(&in, &opt, &res));
    }
    if (rc != MCF_OK) {
        char msg[600];
        strncpy(msg, mcf_last_error(), sizeof msg - 1); msg[sizeof msg - 1] = 0;
        UNPROTECT(np);
        Rf_error("mcfhip (%d): %s", rc, msg);
    }
    UNPROTECT(np);
    return ans;
}

SEXP mcfhip_runmicro1(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                      SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP complete, SEXP mat, SEXP out) {
    return run(0, R_NilValue, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, complete, mat, out);
}
SEXP mcfhip_runmicro2(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                      SEXP lats, SEXP lons, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP complete, SEXP mat, SEXP out) {
    return run(1, R_NilValue, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, complete, mat, out);
}
/* _microclimf_runmicro3Cpp / 4Cpp (src/RcppExports.cpp:300-348): dfsel first, then the same 15 */
SEXP mcfhip_runmicro3(SEXP dfsel, SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt,
                      SEXP zref, SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP complete, SEXP mat,
                      SEXP out) {
    return run(0, dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, complete, mat, out);
}
SEXP mcfhip_runmicro4(SEXP dfsel, SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt,
                      SEXP zref, SEXP lats, SEXP lons, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP complete, SEXP mat,
                      SEXP out) {
    return run(1, dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, complete, mat, out);
}

/* _microclimf_runbioclim1Cpp / 2Cpp (src/microclimfCpp.cpp:3563-3616): 19 arguments; returns the list
 * bio1..bio19 (requested ones) of [rows, cols] matrices.  Solver and reductions stay on the device. */
static SEXP run_bioclim(int array_forcing, int layered, SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc,
                        SEXP reqhgt, SEXP zref, SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP mat,
                        SEXP out, SEXP wetq, SEXP dryq, SEXP hotq, SEXP colq, SEXP air) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_grid_inputs in;
    mcf_options opt;
    SEXP mask = PROTECT(allocVector(LGLSXP, MCF_NOUT)); ++np;
    for (int v = 0; v < MCF_NOUT; ++v) LOGICAL(mask)[v] = TRUE;
    SEXP tru = PROTECT(ScalarLogical(TRUE)); ++np;
    fill_inputs(&in, &opt, &np, array_forcing, R_NilValue, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat,
                lon, Sminp, Smaxp, tfact, tru, mat, mask);
    mcf_bioclim_sel sel;
    mcf_bioclim_out bo;
    memset(&sel, 0, sizeof sel); memset(&bo, 0, sizeof bo);
    sel.wetq = intcol(wetq, &np); sel.nwet = LENGTH(wetq);
    sel.dryq = intcol(dryq, &np); sel.ndry = LENGTH(dryq);
    sel.hotq = intcol(hotq, &np); sel.nhot = LENGTH(hotq);
    sel.colq = intcol(colq, &np); sel.ncol = LENGTH(colq);
    sel.air = asLogical(air) == TRUE;
    SEXP outl = PROTECT(coerceVector(out, LGLSXP)); ++np;
    if (LENGTH(outl) != MCF_NBIO) Rf_error("mcfhip: out must have 19 elements");
    int nreq = 0;
    for (int v = 0; v < MCF_NBIO; ++v) { sel.out[v] = LOGICAL(outl)[v] == TRUE; nreq += sel.out[v]; }
    SEXP ans = PROTECT(allocVector(VECSXP, nreq)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, nreq)); ++np;
    for (int v = 0, k = 0; v < MCF_NBIO; ++v) {
        if (!sel.out[v]) continue;
        SEXP a = PROTECT(allocMatrix(REALSXP, (int)in.rows, (int)in.cols)); ++np;
        char nm[8];
        snprintf(nm, sizeof nm, "bio%d", v + 1);
        SET_VECTOR_ELT(ans, k, a);
        SET_STRING_ELT(nms, k, mkChar(nm));
        bo.bio[v] = REAL(a);
        ++k;
    }
    setAttrib(ans, R_NamesSymbol, nms);
    /* layered (runbioclim3Cpp / 4Cpp): vegp holds [rows, cols, 14] arrays; the library installs the fixed dfsel */
    /* options(mcfhip.devices) — mcfhip_enable(devices = ) —: row blocks over the listed devices, same bits (mcf_runbioclim1_multi) */
    SEXP dv = GetOption1(install("mcfhip.devices"));
    int rc;
    if (dv != R_NilValue && LENGTH(dv) > 0) {
        SEXP dvi = PROTECT(coerceVector(dv, INTSXP)); ++np;
        SEXP nbo = GetOption1(install("mcfhip.blocks"));
        mcf_multi mu;
        mu.n_devices = LENGTH(dvi);
        mu.devices = INTEGER(dvi);
        mu.n_blocks = nbo == R_NilValue ? 0 : asInteger(nbo);
        rc = layered ? (array_forcing ? mcf_runbioclim4_multi(&in, &opt, &sel, &mu, &bo) : mcf_runbioclim3_multi(&in, &opt, &sel, &mu, &bo))
                     : (array_forcing ? mcf_runbioclim2_multi(&in, &opt, &sel, &mu, &bo) : mcf_runbioclim1_multi(&in, &opt, &sel, &mu, &bo));
    } else {
        rc = layered ? (array_forcing ? mcf_runbioclim4(&in, &opt, &sel, &bo) : mcf_runbioclim3(&in, &opt, &sel, &bo))
                     : (array_forcing ? mcf_runbioclim2(&in, &opt, &sel, &bo) : mcf_runbioclim1(&in, &opt, &sel, &bo));
    }
    if (rc != MCF_OK) {
        char msg[600];
        strncpy(msg, mcf_last_error(), sizeof msg - 1); msg[sizeof msg - 1] = 0;
        UNPROTECT(np);
        Rf_error("mcfhip (%d): %s", rc, msg);
    }
    UNPROTECT(np);
    return ans;
}
SEXP mcfhip_runbioclim1(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                        SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP mat, SEXP out, SEXP wetq,
                        SEXP dryq, SEXP hotq, SEXP colq, SEXP air) {
    return run_bioclim(0, 0, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, mat, out,
                       wetq, dryq, hotq, colq, air);
}
SEXP mcfhip_runbioclim2(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                        SEXP lats, SEXP lons, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP mat, SEXP out, SEXP wetq,
                        SEXP dryq, SEXP hotq, SEXP colq, SEXP air) {
    return run_bioclim(1, 0, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, mat,
                       out, wetq, dryq, hotq, colq, air);
}
/* _microclimf_runbioclim3Cpp / 4Cpp (src/microclimfCpp.cpp:3620-3700): the same with 14-layer vegetation */
SEXP mcfhip_runbioclim3(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                        SEXP lat, SEXP lon, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP mat, SEXP out, SEXP wetq,
                        SEXP dryq, SEXP hotq, SEXP colq, SEXP air) {
    return run_bioclim(0, 1, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, mat, out,
                       wetq, dryq, hotq, colq, air);
}
SEXP mcfhip_runbioclim4(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP soilc, SEXP reqhgt, SEXP zref,
                        SEXP lats, SEXP lons, SEXP Sminp, SEXP Smaxp, SEXP tfact, SEXP mat, SEXP out, SEXP wetq,
                        SEXP dryq, SEXP hotq, SEXP colq, SEXP air) {
    return run_bioclim(1, 1, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, mat,
                       out, wetq, dryq, hotq, colq, air);
}

/* ---- snow branch ------------------------------------------------------------------------------
 * _microclimf_gridmodelsnow1/2 (bodies src/microclimfCpp.cpp:4172-4673): (obstime, climdata, pointm,
 * vegp, other, snowenv) -> list(Tc, Tg, sdepc, sdepg, sden, agec, ageg, meltc, meltg). */
static SEXP opt_elt(SEXP list, const char *name, const char *alt) {   /* like elt(), NULL when absent */
    SEXP names = getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < XLENGTH(list); ++i) {
        const char *n = CHAR(STRING_ELT(names, i));
        if (strcmp(n, name) == 0 || (alt && strcmp(n, alt) == 0)) return VECTOR_ELT(list, i);
    }
    return R_NilValue;
}
static void fill_snow(mcf_snow_inputs *in, int *np, int array_forcing, int micro, SEXP obstime, SEXP climdata,
                      SEXP pointm, SEXP vegp, SEXP other) {
    memset(in, 0, sizeof *in);
    SEXP dim = getAttrib(elt(vegp, "pai", NULL), R_DimSymbol);
    if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 2) Rf_error("mcfhip: vegp$pai must be a matrix");
    in->rows = INTEGER(dim)[0]; in->cols = INTEGER(dim)[1];
    in->tsteps = XLENGTH(elt(obstime, "year", NULL));
    in->array_forcing = array_forcing;
    in->obstime.year = intcol(elt(obstime, "year", NULL), np);
    in->obstime.month = intcol(elt(obstime, "month", NULL), np);
    in->obstime.day = intcol(elt(obstime, "day", NULL), np);
    in->obstime.hour = dbl(elt(obstime, "hour", NULL), np);
    static const char *cn[9] = {"temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir",
                                "precip"};
    const double **cp = (const double **)&in->clim;
    for (int i = 0; i < 9; ++i) cp[i] = dbl(elt(climdata, cn[i], i == 8 ? "prec" : NULL), np);   /* cpp:5083 */
    if (micro) in->clim.umu = dbl(elt(climdata, "umu", NULL), np);
    if (pointm != R_NilValue) {
        static const char *pn[5] = {"Gp", "Tc", "RswabsG", "RlwabsG", "umu"};
        const double **pp = (const double **)&in->pointm;
        for (int i = 0; i < 5; ++i) pp[i] = dbl(elt(pointm, pn[i], NULL), np);
    }
    static const char *vn[7] = {"pai", "hgt", "leaft", "clump", "paia", "leafd", "leafden"};
    const double **vp = (const double **)&in->vegp;
    for (int i = 0; i < (micro ? 7 : 4); ++i) vp[i] = dbl(elt(vegp, vn[i], NULL), np);
    in->other.slope = dbl(elt(other, "slope", NULL), np);
    in->other.aspect = dbl(elt(other, "aspect", NULL), np);
    in->other.skyview = dbl(elt(other, "skyview", NULL), np);
    in->other.wsa = dbl(elt(other, "wsa", NULL), np);
    in->other.hor = dbl(elt(other, "hor", NULL), np);
    in->other.zref = asReal(elt(other, "zref", NULL));
    if (array_forcing) {   /* cpp:4457-4458 "lats"/"lons"; cpp:5091-5092 "lat"/"lon" */
        in->other.lats = dbl(elt(other, "lats", "lat"), np);
        in->other.lons = dbl(elt(other, "lons", "lon"), np);
    } else {
        in->other.lat = asReal(elt(other, "lat", NULL));
        in->other.lon = asReal(elt(other, "lon", NULL));
    }
    if (micro) {
        in->other.Smax = dbl(elt(other, "Smax", NULL), np);
    } else {
        in->other.isnowdc = dbl(elt(other, "isnowdc", NULL), np);
        in->other.isnowdg = dbl(elt(other, "isnowdg", NULL), np);
        in->other.isnowac = intcol(elt(other, "isnowac", NULL), np);   /* IntegerMatrix, cpp:4203-4204 */
        in->other.isnowag = intcol(elt(other, "isnowag", NULL), np);
    }
}
static void raise_last(int rc, int np) {
    char msg[600];
    strncpy(msg, mcf_last_error(), sizeof msg - 1); msg[sizeof msg - 1] = 0;
    UNPROTECT(np);
    Rf_error("mcfhip (%d): %s", rc, msg);
}
static SEXP run_snowmodel(int array_forcing, SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP other,
                          SEXP snowenv) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_snow_inputs in;
    fill_snow(&in, &np, array_forcing, 0, obstime, climdata, pointm, vegp, other);
    in.snowenv = mcf_snowenv_from_name(CHAR(asChar(snowenv)));
    static const char *on[9] = {"Tc", "Tg", "sdepc", "sdepg", "sden", "agec", "ageg", "meltc", "meltg"};
    mcf_snowmodel_out res;
    double **rp = (double **)&res;
    SEXP ans = PROTECT(allocVector(VECSXP, 9)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, 9)); ++np;
    for (int v = 0; v < 9; ++v) {
        SEXP a;
        if (v < 5) {
            a = PROTECT(allocVector(REALSXP, (R_xlen_t)in.rows * in.cols * in.tsteps)); ++np;
            SEXP d = PROTECT(allocVector(INTSXP, 3)); ++np;
            INTEGER(d)[0] = (int)in.rows; INTEGER(d)[1] = (int)in.cols; INTEGER(d)[2] = (int)in.tsteps;
            setAttrib(a, R_DimSymbol, d);
        } else {
            a = PROTECT(allocMatrix(REALSXP, (int)in.rows, (int)in.cols)); ++np;
        }
        SET_VECTOR_ELT(ans, v, a);
        SET_STRING_ELT(nms, v, mkChar(on[v]));
        rp[v] = REAL(a);
    }
    setAttrib(ans, R_NamesSymbol, nms);
    int rc = array_forcing ? mcf_gridmodelsnow2(&in, &res, 0) : mcf_gridmodelsnow1(&in, &res, 0);
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return ans;
}
SEXP mcfhip_gridmodelsnow1(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP other, SEXP snowenv) {
    return run_snowmodel(0, obstime, climdata, pointm, vegp, other, snowenv);
}
SEXP mcfhip_gridmodelsnow2(SEXP obstime, SEXP climdata, SEXP pointm, SEXP vegp, SEXP other, SEXP snowenv) {
    return run_snowmodel(1, obstime, climdata, pointm, vegp, other, snowenv);
}
/* _microclimf_gridmicrosnow1/2 (bodies src/microclimfCpp.cpp:4894-5214): (reqhgt, obstime, climdata, snowm,
 * micro, vegp, other, mat, out) -> the requested fields of `micro`, updated where snow lies.  The reference
 * writes into `micro`'s own storage; this shim updates duplicates and leaves the argument alone. */
static SEXP run_microsnow(int array_forcing, SEXP reqhgt, SEXP obstime, SEXP climdata, SEXP snowm, SEXP micro,
                          SEXP vegp, SEXP other, SEXP mat, SEXP out) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_snow_inputs in;
    fill_snow(&in, &np, array_forcing, 1, obstime, climdata, R_NilValue, vegp, other);
    mcf_snowm sm;
    sm.Tc = dbl(elt(snowm, "Tc", NULL), &np);
    sm.Tg = dbl(elt(snowm, "Tg", NULL), &np);
    sm.totalSWE = dbl(elt(snowm, "totalSWE", NULL), &np);
    sm.groundsnowdepth = dbl(elt(snowm, "groundsnowdepth", NULL), &np);
    sm.snowden = dbl(elt(snowm, "snowden", NULL), &np);
    static const char *on[MCF_NOUT] = {"Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown",
                                       "Rlwdown", "Rswup", "Rlwup"};
    SEXP outl = PROTECT(coerceVector(out, LGLSXP)); ++np;
    if (LENGTH(outl) != MCF_NOUT) Rf_error("mcfhip: out must have 10 elements");
    int32_t sel[MCF_NOUT];
    int nreq = 0;
    for (int v = 0; v < MCF_NOUT; ++v) { sel[v] = LOGICAL(outl)[v] == TRUE; nreq += sel[v]; }
    mcf_outputs res;
    memset(&res, 0, sizeof res);
    SEXP ans = PROTECT(allocVector(VECSXP, nreq)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, nreq)); ++np;
    for (int v = 0, k = 0; v < MCF_NOUT; ++v) {
        if (!sel[v]) continue;
        SEXP src = elt(micro, on[v], NULL);
        SEXP a = PROTECT(TYPEOF(src) == REALSXP ? duplicate(src) : coerceVector(src, REALSXP)); ++np;
        SET_VECTOR_ELT(ans, k, a);
        SET_STRING_ELT(nms, k, mkChar(on[v]));
        res.var[v] = REAL(a);
        ++k;
    }
    setAttrib(ans, R_NamesSymbol, nms);
    int rc = array_forcing ? mcf_gridmicrosnow2(&in, &sm, asReal(reqhgt), asReal(mat), sel, &res, 0)
                           : mcf_gridmicrosnow1(&in, &sm, asReal(reqhgt), asReal(mat), sel, &res, 0);
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return ans;
}
SEXP mcfhip_gridmicrosnow1(SEXP reqhgt, SEXP obstime, SEXP climdata, SEXP snowm, SEXP micro, SEXP vegp, SEXP other,
                           SEXP mat, SEXP out) {
    return run_microsnow(0, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out);
}
SEXP mcfhip_gridmicrosnow2(SEXP reqhgt, SEXP obstime, SEXP climdata, SEXP snowm, SEXP micro, SEXP vegp, SEXP other,
                           SEXP mat, SEXP out) {
    return run_microsnow(1, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out);
}

/* _microclimf_applycpp3 (src/microclimfCpp.cpp:5553-5588): (a [rows,cols,tsteps], fun_name) -> numeric(tsteps) */
SEXP mcfhip_applycpp3(SEXP a, SEXP fun_name) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    SEXP dim = getAttrib(a, R_DimSymbol);
    if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 3) Rf_error("mcfhip: applycpp3 needs a 3-D array");
    const char *fn = CHAR(asChar(fun_name));
    int fun = !strcmp(fn, "mean") ? MCF_APPLY_MEAN : !strcmp(fn, "sum") ? MCF_APPLY_SUM
            : !strcmp(fn, "max") ? MCF_APPLY_MAX : !strcmp(fn, "min") ? MCF_APPLY_MIN : -1;
    if (fun < 0) Rf_error("Unknown function name");
    const double *pa = dbl(a, &np);
    SEXP ans = PROTECT(allocVector(REALSXP, INTEGER(dim)[2])); ++np;
    int rc = mcf_applycpp3(pa, INTEGER(dim)[0], INTEGER(dim)[1], INTEGER(dim)[2], fun, REAL(ans), NULL, 0);
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return ans;
}
/* array_forcing: `.snowmodel2`'s loop — weather / pointm elements are [rows, cols, tsteps] arrays (winddir a vector), other has
 * lats / lons; the caller sets af_wind / af_wsa_s afterwards (mcfhip_snowrun_create) */
static void fill_snowdriver_g(mcf_snowdriver_in *din, int *np, int array_forcing, SEXP obstime, SEXP weather, SEXP pointm, SEXP vegp,
                              SEXP other, SEXP snowenv, SEXP dtm, SEXP res, SEXP tfact);
static void fill_snowdriver(mcf_snowdriver_in *din, int *np, SEXP obstime, SEXP weather, SEXP pointm, SEXP vegp, SEXP other,
                            SEXP snowenv, SEXP dtm, SEXP res, SEXP tfact) {
    fill_snowdriver_g(din, np, 0, obstime, weather, pointm, vegp, other, snowenv, dtm, res, tfact);
}
static void fill_snowdriver_g(mcf_snowdriver_in *din, int *np, int array_forcing, SEXP obstime, SEXP weather, SEXP pointm, SEXP vegp,
                              SEXP other, SEXP snowenv, SEXP dtm, SEXP res, SEXP tfact) {
    memset(din, 0, sizeof *din);
    mcf_snow_inputs *in = &din->base;
    in->array_forcing = array_forcing ? 1 : 0;
    /* fill_snow() wants the terrain members the loop recomputes: take what the driver needs by hand */
    SEXP dim = getAttrib(elt(vegp, "pai", NULL), R_DimSymbol);
    if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 2) Rf_error("mcfhip: vegp$pai must be a matrix");
    in->rows = INTEGER(dim)[0]; in->cols = INTEGER(dim)[1];
    in->tsteps = XLENGTH(elt(obstime, "year", NULL));
    in->obstime.year = intcol(elt(obstime, "year", NULL), np);
    in->obstime.month = intcol(elt(obstime, "month", NULL), np);
    in->obstime.day = intcol(elt(obstime, "day", NULL), np);
    in->obstime.hour = dbl(elt(obstime, "hour", NULL), np);
    static const char *cn[9] = {"temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir",
                                "precip"};
    const double **cp = (const double **)&in->clim;
    for (int i = 0; i < 9; ++i) cp[i] = dbl(elt(weather, cn[i], NULL), np);
    static const char *pn[5] = {"Gp", "Tc", "RswabsG", "RlwabsG", "umu"};
    const double **pp = (const double **)&in->pointm;
    for (int i = 0; i < 5; ++i) pp[i] = dbl(elt(pointm, pn[i], NULL), np);
    static const char *vn[4] = {"pai", "hgt", "leaft", "clump"};
    const double **vp = (const double **)&in->vegp;
    for (int i = 0; i < 4; ++i) vp[i] = dbl(elt(vegp, vn[i], NULL), np);
    if (in->array_forcing) {                 /* `.snowmodel2`: per-cell latitudes / longitudes (R/internal.R:2937-2941) */
        in->other.lats = dbl(elt(other, "lats", NULL), np);
        in->other.lons = dbl(elt(other, "lons", NULL), np);
    } else {
        in->other.lat = asReal(elt(other, "lat", NULL));
        in->other.lon = asReal(elt(other, "lon", NULL));
    }
    in->other.zref = asReal(elt(other, "zref", NULL));
    in->other.isnowdc = dbl(elt(other, "isnowdc", NULL), np);
    in->other.isnowdg = dbl(elt(other, "isnowdg", NULL), np);
    in->other.isnowac = intcol(elt(other, "isnowac", NULL), np);
    in->other.isnowag = intcol(elt(other, "isnowag", NULL), np);
    in->snowenv = mcf_snowenv_from_name(CHAR(asChar(snowenv)));
    din->dtm = dbl(dtm, np);
    din->res = asReal(res);
    din->tfact = asReal(tfact);
}
/* the `for (day in 1:n5days)` loop of .snowmodel1 (R/internal.R:2563-2617) in one call */
SEXP mcfhip_snowmodel1(SEXP obstime, SEXP weather, SEXP pointm, SEXP vegp, SEXP other, SEXP snowenv, SEXP dtm,
                       SEXP res, SEXP tfact) {
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_snowdriver_in din;
    fill_snowdriver(&din, &np, obstime, weather, pointm, vegp, other, snowenv, dtm, res, tfact);
    mcf_snow_inputs *in = &din.base;
    static const char *on[5] = {"Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden"};
    mcf_snowdriver_out res5;
    double **rp = (double **)&res5;
    SEXP ans = PROTECT(allocVector(VECSXP, 5)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, 5)); ++np;
    for (int v = 0; v < 5; ++v) {
        SEXP a = PROTECT(allocVector(REALSXP, (R_xlen_t)in->rows * in->cols * in->tsteps)); ++np;
        SEXP d = PROTECT(allocVector(INTSXP, 3)); ++np;
        INTEGER(d)[0] = (int)in->rows; INTEGER(d)[1] = (int)in->cols; INTEGER(d)[2] = (int)in->tsteps;
        setAttrib(a, R_DimSymbol, d);
        SET_VECTOR_ELT(ans, v, a);
        SET_STRING_ELT(nms, v, mkChar(on[v]));
        rp[v] = REAL(a);
    }
    setAttrib(ans, R_NamesSymbol, nms);
    /* options(mcfhip.devices) — mcfhip_enable(devices = ) — sends the chunk loop over the listed devices too (row blocks, one
     * snow plan per block: include/mcf.h mcf_snowmodel1_multi) */
    SEXP dv = GetOption1(install("mcfhip.devices"));
    int rc;
    if (dv != R_NilValue && LENGTH(dv) > 0) {
        SEXP dvi = PROTECT(coerceVector(dv, INTSXP)); ++np;
        SEXP nbo = GetOption1(install("mcfhip.blocks"));
        mcf_multi mu;
        mu.n_devices = LENGTH(dvi);
        mu.devices = INTEGER(dvi);
        mu.n_blocks = nbo == R_NilValue ? 0 : asInteger(nbo);
        rc = mcf_snowmodel1_multi(&din, &res5, &mu);
    } else {
        rc = mcf_snowmodel1(&din, &res5, 0);
    }
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return ans;
}

/* ---- runmicro(snow = TRUE) device-resident (include/mcf.h mcf_snowrun_*): `.snowmodel1`'s chunk loop + `.runmicrosnow1`
 * (R/internal.R:2563-2617, 3581-3659) without the year's snow arrays in R.  Three calls (r/mcfhip_overrides.R):
 *   h    <- .Call("mcfhip_snowrun_create", grid, snow)  grid: the fifteen arguments of runmicro1Cpp for the whole series, in order;
 *                                                     snow: list(obstime, weather, pointm, vegp, other, snowenv, dtm, res, tfact)
 *   days <- .Call("mcfhip_snowrun_pass1", h)            list(snowdays, nosnowdays): 1-based day numbers, as snowdaysfun's callers make
 *   mout <- .Call("mcfhip_snowrun_pass2", h, obstime, weather, vegp, other, mat)   gridmicrosnow1's inputs for the WHOLE series
 * The external pointer keeps every R vector the library reads alive and destroys the run when it is collected. */
typedef struct snowrun_box {
    mcf_snowrun *run;
    mcf_grid_inputs grid;
    mcf_options opt;
    mcf_snowdriver_in snow;
    mcf_microsnow_in in;
} snowrun_box;
static void snowrun_finalize(SEXP xp) {
    snowrun_box *b = (snowrun_box *)R_ExternalPtrAddr(xp);
    if (!b) return;
    if (b->run) mcf_snowrun_destroy(b->run);
    free(b);
    R_ClearExternalPtr(xp);
}
SEXP mcfhip_snowrun_create(SEXP grid, SEXP snow) {
    if (TYPEOF(grid) != VECSXP || LENGTH(grid) != 15) Rf_error("mcfhip: grid must be the list of runmicro1Cpp's / runmicro2Cpp's fifteen arguments");
    /* nine elements: data.frame weather (`.snowmodel1`); eleven: ARRAY weather at the raster's resolution (`.snowmodel2`'s loop +
     * `.runmicrosnow2`: grid = runmicro2Cpp's arguments) with af_wind = sqrt(wuv^2 + wvv^2) per step and af_wsa_s (include/mcf.h
     * mcf_snowdriver_in) appended.  NOT EXERCISED: no R here, and r/mcfhip_overrides.R has no stand-in for `.runmicrosnow2` yet. */
    if (TYPEOF(snow) != VECSXP || (LENGTH(snow) != 9 && LENGTH(snow) != 11))
        Rf_error("mcfhip: snow must be list(obstime, weather, pointm, vegp, other, snowenv, dtm, res, tfact[, af_wind, af_wsa_s])");
    const int arrayw = LENGTH(snow) == 11;
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    snowrun_box *b = (snowrun_box *)calloc(1, sizeof *b);
    if (!b) Rf_error("mcfhip: out of memory");
    /* prot = list(grid, snow, coerced vectors): alive as long as the external pointer */
    SEXP prot = PROTECT(allocVector(VECSXP, 3)); ++np;
    SEXP keepl = PROTECT(allocVector(VECSXP, 128)); ++np;
    SET_VECTOR_ELT(prot, 0, grid); SET_VECTOR_ELT(prot, 1, snow); SET_VECTOR_ELT(prot, 2, keepl);
    SEXP xp = PROTECT(R_MakeExternalPtr(b, R_NilValue, prot)); ++np;
    R_RegisterCFinalizerEx(xp, snowrun_finalize, TRUE);
    g_keep = keepl; g_nkeep = 0;
#define G(i) VECTOR_ELT(grid, i)
    fill_inputs(&b->grid, &b->opt, &np, arrayw, R_NilValue, G(0), G(1), G(2), G(3), G(4), G(5), G(6), G(7), G(8), G(9), G(10), G(11), G(12),
                G(13), G(14));
#undef G
#define S(i) VECTOR_ELT(snow, i)
    fill_snowdriver_g(&b->snow, &np, arrayw, S(0), S(1), S(2), S(3), S(4), S(5), S(6), S(7), S(8));
    if (arrayw) {
        b->snow.af_wind = dbl(S(9), &np);
        b->snow.af_wsa_s = asInteger(S(10));
    }
#undef S
    g_keep = NULL;
    b->in.grid = &b->grid; b->in.snow = &b->snow; b->in.micro = NULL; b->in.mat = 0.0;
    SEXP dv = GetOption1(install("mcfhip.devices"));
    int rc;
    if (dv != R_NilValue && LENGTH(dv) > 0) {
        SEXP dvi = PROTECT(coerceVector(dv, INTSXP)); ++np;
        SEXP nbo = GetOption1(install("mcfhip.blocks"));
        mcf_multi mu;
        mu.n_devices = LENGTH(dvi);
        mu.devices = INTEGER(dvi);
        mu.n_blocks = nbo == R_NilValue ? 0 : asInteger(nbo);
        rc = mcf_snowrun_create(&b->in, &b->opt, &mu, &b->run);
    } else {
        rc = mcf_snowrun_create(&b->in, &b->opt, NULL, &b->run);
    }
    if (rc != MCF_OK) raise_last(rc, np);      /* (the finalizer frees the box) */
    SEXP kg = GetOption1(install("mcfhip.keep_gb"));     /* mcfhip_enable(keep_gb = ...): pass 1's chunks stay in HBM for pass 2 */
    if (kg != R_NilValue && LENGTH(kg) > 0 && asReal(kg) > 0) {
        rc = mcf_snowrun_keep(b->run, (int64_t)(asReal(kg) * 1073741824.0));
        if (rc != MCF_OK) raise_last(rc, np);
    }
    UNPROTECT(np);
    return xp;
}
static snowrun_box *snowrun_of(SEXP h) {
    if (TYPEOF(h) != EXTPTRSXP || !R_ExternalPtrAddr(h)) Rf_error("mcfhip: not a live snow run");
    return (snowrun_box *)R_ExternalPtrAddr(h);
}
SEXP mcfhip_snowrun_pass1(SEXP h) {
    snowrun_box *b = snowrun_of(h);
    const int nd = mcf_snowrun_days(b->run);
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    SEXP sf = PROTECT(allocVector(INTSXP, nd)); ++np;
    SEXP nf = PROTECT(allocVector(INTSXP, nd)); ++np;
    const int rc = mcf_snowrun_pass1(b->run, NULL, INTEGER(sf), INTEGER(nf));
    if (rc != MCF_OK) raise_last(rc, np);
    int ns = 0, nn = 0;
    for (int d = 0; d < nd; ++d) { ns += INTEGER(sf)[d] != 0; nn += INTEGER(nf)[d] != 0; }
    SEXP sd = PROTECT(allocVector(INTSXP, ns)); ++np;
    SEXP nsd = PROTECT(allocVector(INTSXP, nn)); ++np;
    for (int d = 0, i = 0, j = 0; d < nd; ++d) {
        if (INTEGER(sf)[d]) INTEGER(sd)[i++] = d + 1;
        if (INTEGER(nf)[d]) INTEGER(nsd)[j++] = d + 1;
    }
    SEXP ans = PROTECT(allocVector(VECSXP, 2)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, 2)); ++np;
    SET_VECTOR_ELT(ans, 0, sd); SET_VECTOR_ELT(ans, 1, nsd);
    SET_STRING_ELT(nms, 0, mkChar("snowdays")); SET_STRING_ELT(nms, 1, mkChar("nosnowdays"));
    setAttrib(ans, R_NamesSymbol, nms);
    UNPROTECT(np);
    return ans;
}
SEXP mcfhip_snowrun_pass2(SEXP h, SEXP obstime, SEXP weather, SEXP vegp, SEXP other, SEXP mat) {
    snowrun_box *b = snowrun_of(h);
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_snow_inputs micro;
    const int have = obstime != R_NilValue;      /* NULL inputs: a year without a snow day */
    if (have) fill_snow(&micro, &np, b->grid.array_forcing ? 1 : 0, 1, obstime, weather, R_NilValue, vegp, other);
    static const char *on[MCF_NOUT] = {"Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown",
                                       "Rlwdown", "Rswup", "Rlwup"};
    mcf_outputs res;
    memset(&res, 0, sizeof res);
    int nreq = 0;
    for (int v = 0; v < MCF_NOUT; ++v) nreq += b->opt.out[v];
    SEXP ans = PROTECT(allocVector(VECSXP, nreq)); ++np;
    SEXP nms = PROTECT(allocVector(STRSXP, nreq)); ++np;
    const R_xlen_t n = (R_xlen_t)b->grid.rows * b->grid.cols * b->grid.tsteps;
    for (int v = 0, k = 0; v < MCF_NOUT; ++v) {
        if (!b->opt.out[v]) continue;
        SEXP a = PROTECT(allocVector(REALSXP, n)); ++np;
        SEXP d = PROTECT(allocVector(INTSXP, 3)); ++np;
        INTEGER(d)[0] = (int)b->grid.rows; INTEGER(d)[1] = (int)b->grid.cols; INTEGER(d)[2] = (int)b->grid.tsteps;
        setAttrib(a, R_DimSymbol, d);
        SET_VECTOR_ELT(ans, k, a);
        SET_STRING_ELT(nms, k, mkChar(on[v]));
        res.var[v] = REAL(a);
        ++k;
    }
    setAttrib(ans, R_NamesSymbol, nms);
    const int rc = mcf_snowrun_pass2(b->run, have ? &micro : NULL, asReal(mat), &res);
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return ans;
}

/* writetonc(mout, fileout, dtm, reqhgt, vars) (R/dataprep.R:1063-1260): the R side hands over what it takes terra for (the
 * cell-centre coordinates, the hours since 1970, the projection's text); the dataset — names, long names, units, packing,
 * missval, crs variable — is made by libmcfhip (mcf_nc_*).  format: "classic" or "netcdf4" (the reference's container). */
SEXP mcfhip_writetonc(SEXP mout, SEXP fileout, SEXP east, SEXP north, SEXP hours, SEXP crs_wkt, SEXP reqhgt, SEXP vars,
                      SEXP format) {
    static const char *on[MCF_NOUT] = {"Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown",
                                       "Rswup", "Rlwup"};
    int np = 0;
    g_keep = NULL; g_nkeep = 0;     /* an earlier entry may have left through Rf_error with the list still set: it is unprotected then */
    mcf_nc_spec sp;
    memset(&sp, 0, sizeof sp);
    const double *ptr[MCF_NOUT] = {0};
    if (TYPEOF(vars) != STRSXP || LENGTH(vars) < 1) Rf_error("mcfhip: vars must name at least one variable");
    SEXP dim = R_NilValue;
    for (int i = 0; i < LENGTH(vars); ++i) {
        const char *nm = CHAR(STRING_ELT(vars, i));
        int v = -1;
        for (int k = 0; k < MCF_NOUT; ++k) if (strcmp(nm, on[k]) == 0) v = k;
        if (v < 0) Rf_error("mcfhip: writetonc knows no variable '%s'", nm);
        SEXP a = elt(mout, nm, NULL);
        SEXP d = getAttrib(a, R_DimSymbol);
        if (TYPEOF(d) != INTSXP || LENGTH(d) != 3) Rf_error("mcfhip: mout$%s must be a [rows, cols, steps] array", nm);
        if (dim != R_NilValue && (INTEGER(d)[0] != INTEGER(dim)[0] || INTEGER(d)[1] != INTEGER(dim)[1] || INTEGER(d)[2] != INTEGER(dim)[2]))
            Rf_error("mcfhip: the arrays of mout differ in shape");
        dim = d;
        sp.vars[v] = 1;
        ptr[v] = dbl(a, &np);
    }
    sp.rows = INTEGER(dim)[0]; sp.cols = INTEGER(dim)[1]; sp.nsteps = INTEGER(dim)[2];
    if (XLENGTH(east) != sp.cols || XLENGTH(north) != sp.rows || XLENGTH(hours) != sp.nsteps)
        Rf_error("mcfhip: dtm / mout$tme do not match the arrays (east %ld, north %ld, time %ld)", (long)XLENGTH(east),
                 (long)XLENGTH(north), (long)XLENGTH(hours));
    sp.east = dbl(east, &np); sp.north = dbl(north, &np); sp.time_hours = dbl(hours, &np);
    sp.crs_wkt = CHAR(asChar(crs_wkt));
    sp.reqhgt = asReal(reqhgt);
    sp.reference_puts_only = 0;      /* what writetonc evidently means, not the file its never-true `%in%` guards produce */
    sp.format = strcmp(CHAR(asChar(format)), "netcdf4") == 0 ? MCF_NC_NETCDF4 : MCF_NC_CLASSIC;
    mcf_ncfile *nc = NULL;
    int rc = mcf_nc_create(CHAR(asChar(fileout)), &sp, &nc);
    if (rc == MCF_OK) {
        rc = mcf_nc_write_host(nc, 0, sp.nsteps, ptr);
        const int rc2 = mcf_nc_close(nc);
        if (rc == MCF_OK) rc = rc2;
    }
    if (rc != MCF_OK) raise_last(rc, np);
    UNPROTECT(np);
    return R_NilValue;
}

static const R_CallMethodDef CallEntries[] = {
    {"mcfhip_runmicro1", (DL_FUNC)&mcfhip_runmicro1, 15},
    {"mcfhip_runmicro2", (DL_FUNC)&mcfhip_runmicro2, 15},
    {"mcfhip_runmicro3", (DL_FUNC)&mcfhip_runmicro3, 16},
    {"mcfhip_runmicro4", (DL_FUNC)&mcfhip_runmicro4, 16},
    {"mcfhip_runbioclim1", (DL_FUNC)&mcfhip_runbioclim1, 19},
    {"mcfhip_runbioclim2", (DL_FUNC)&mcfhip_runbioclim2, 19},
    {"mcfhip_runbioclim3", (DL_FUNC)&mcfhip_runbioclim3, 19},
    {"mcfhip_runbioclim4", (DL_FUNC)&mcfhip_runbioclim4, 19},
    {"mcfhip_gridmodelsnow1", (DL_FUNC)&mcfhip_gridmodelsnow1, 6},
    {"mcfhip_gridmodelsnow2", (DL_FUNC)&mcfhip_gridmodelsnow2, 6},
    {"mcfhip_gridmicrosnow1", (DL_FUNC)&mcfhip_gridmicrosnow1, 9},
    {"mcfhip_gridmicrosnow2", (DL_FUNC)&mcfhip_gridmicrosnow2, 9},
    {"mcfhip_applycpp3", (DL_FUNC)&mcfhip_applycpp3, 2},
    {"mcfhip_snowmodel1", (DL_FUNC)&mcfhip_snowmodel1, 9},
    {"mcfhip_snowrun_create", (DL_FUNC)&mcfhip_snowrun_create, 2},
    {"mcfhip_snowrun_pass1", (DL_FUNC)&mcfhip_snowrun_pass1, 1},
    {"mcfhip_snowrun_pass2", (DL_FUNC)&mcfhip_snowrun_pass2, 6},
    {"mcfhip_writetonc", (DL_FUNC)&mcfhip_writetonc, 9},
    {NULL, NULL, 0}};

void R_init_mcfhip_glue(DllInfo *dll) {
    if (mcf_abi_version() != MCF_ABI_VERSION)      /* the glue was compiled against another include/mcf.h */
        Rf_error("mcfhip: libmcfhip ABI version %d, glue built for %d", mcf_abi_version(), MCF_ABI_VERSION);
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
