/*
 * mcf_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, un-hoisted restatement of the reference's grid
 * microclimate solver (ilyamaclean/microclimf v2.0.0, src/microclimfCpp.cpp).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg can check / time the HIP path against the reference's
 * algorithm on a box where neither R nor the reference source exist.  Nothing
 * in microclimf_amd/ may import, link or call it.
 *
 * Every function cites the reference lines it follows ("cpp:" =
 * src/microclimfCpp.cpp, "hdr:" = src/microclimfheaders.h).  The evaluation
 * order of the reference is kept (no hoisting, no algebraic rewrites) so that
 * the independently hoisted HIP kernels are cross-checked by it.
 *
 * PARITY PINNING: the reference cannot be compiled in the build container (it
 * needs Rcpp.h / R, which are absent, and building it against stand-in headers
 * is not allowed), and the reference ships no golden vectors for this path: its
 * only test that touches this arithmetic (tests/testthat/
 * test-microclimatemodel_wrapper.R) asserts interval bounds for one point.
 * That test (and test-BigLeafCpp.R) is replayed against this file by
 * oracle/replay_reference_tests.py (tests/test_oracle_reference_bounds.py): all
 * 33 of the reference's assertions hold.  Beyond those bounds the numeric
 * parity of this oracle with the reference is UNPINNED ("parity unpinned") —
 * see DESIGN.md §2.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mcf.h"
#include "mcf_oracle.h"

/* cpp:14-19 */
static const double PI_ = 3.14159265358979323846;
static const double TORAD = 3.14159265358979323846 / 180.0;
static const double SB = 5.67e-8;
static const double THETAM = 0.365;
static const double KA = 0.4;
static const double OMDY = (2.0 * 3.14159265358979323846) / (24.0 * 3600.0);

double orc_na_real(void) {
    union { uint64_t u; double d; } v;
    v.u = 0x7FF00000000007A2ULL; /* R's NA_real_ */
    return v.d;
}

/* cpp:24-26 radem */
static double radem(double tc) { return pow(tc + 273.15, 4.0); }

/* cpp:28-37 juldayCpp.  `yadj / 100` is an int division in the reference. */
int orc_julday(int year, int month, int day) {
    double dd = day + 0.5;
    int madj = month + (month < 3) * 12;
    int yadj = year + (month < 3) * -1;
    double j = trunc(365.25 * (yadj + 4716)) + trunc(30.6001 * (madj + 1)) + dd - 1524.5;
    int b = (int)(2 - trunc((double)(yadj / 100)) + trunc(trunc((double)(yadj / 100)) / 4));
    int jd = (int)(j + (j > 2299160) * b);
    return jd;
}

/* cpp:39-46 soltimeCpp */
static double soltime(int jd, double lt, double lon) {
    double m = 6.24004077 + 0.01720197 * (jd - 2451545.0);
    double eot = -7.659 * sin(m) + 9.863 * sin(2 * m + 3.5932);
    return lt + (4.0 * lon + eot) / 60.0;
}

/* cpp:48-83 solpositionCpp */
orc_solmodel orc_solposition(double lat, double lon, int year, int month, int day, double lt) {
    int jd = orc_julday(year, month, day);
    double st = soltime(jd, lt, lon);
    double latr = lat * PI_ / 180.0;
    double tt = 0.261799 * (st - 12);
    double dec = (PI_ * 23.5 / 180) * cos(2 * PI_ * ((jd - 159.5) / 365.25));
    double coh = sin(dec) * sin(latr) + cos(dec) * cos(latr) * cos(tt);
    double z = acos(coh) * (180 / PI_);
    double sh = sin(dec) * sin(latr) + cos(dec) * cos(latr) * cos(tt);
    double hh = atan(sh / sqrt(1 - sh * sh));
    double sazi = cos(dec) * sin(tt) / cos(hh);
    double cazi = (sin(latr) * cos(dec) * cos(tt) - cos(latr) * sin(dec)) /
                  sqrt(pow(cos(dec) * sin(tt), 2) +
                       pow(sin(latr) * cos(dec) * cos(tt) - cos(latr) * sin(dec), 2));
    double sqt = 1 - sazi * sazi;
    if (sqt < 0) sqt = 0;
    double azi = 180 + (180 * atan(sazi / sqrt(sqt))) / PI_;
    if (cazi < 0) {
        if (sazi < 0) azi = 180 - azi;
        else azi = 540 - azi;
    }
    orc_solmodel s;
    s.zend = z;
    s.zenr = z * TORAD;
    s.azid = azi;
    s.azir = azi * TORAD;
    return s;
}

/* cpp:85-102 solarindexCpp */
double orc_solarindex(double slope, double aspect, double zend, double azid, int shadowmask) {
    double si;
    if (zend > 90.0 && !shadowmask) {
        si = 0;
    } else {
        if (slope == 0.0) {
            si = cos(zend * TORAD);
        } else {
            si = cos(zend * TORAD) * cos(slope * TORAD) +
                 sin(zend * TORAD) * sin(slope * TORAD) * cos((azid - aspect) * TORAD);
        }
    }
    if (si < 0.0) si = 0.0;
    return si;
}

/* cpp:104-132 cankCpp */
orc_kstruct orc_cank(double zenr, double x, double si) {
    double k;
    if (zenr > (PI_ / 2.0)) zenr = PI_ / 2.0;
    if (si < 0.0) si = 0.0;
    if (x == 1.0) {
        k = 1.0 / (2.0 * cos(zenr));
    } else if (isinf(x)) {
        k = 1.0;
    } else if (x == 0.0) {
        k = tan(zenr);
    } else {
        k = sqrt(x * x + (tan(zenr) * tan(zenr))) / (x + 1.774 * pow((x + 1.182), -0.733));
    }
    if (k > 6000.0) k = 6000.0;
    double kd = k * cos(zenr) / si;
    if (si == 0) kd = 1.0;
    double Kc = 1.0 / si;
    if (si == 0.0) Kc = 600.0;
    orc_kstruct o;
    o.k = k;
    o.kd = kd;
    o.Kc = Kc;
    return o;
}

/* cpp:134-162 twostreamdifCpp */
typedef struct {
    double p1, p2, p3, p4, om, a, gma, J, del, h, u1, S1, D1, D2;
} tsdif_t;

static tsdif_t twostreamdif_params(double pait, double x, double lref, double ltra, double gref) {
    tsdif_t p;
    p.om = lref + ltra;
    p.a = 1.0 - p.om;
    p.del = lref - ltra;
    p.J = 1.0 / 3.0;
    if (x != 1.0) {
        double mla = 9.65 * pow((3.0 + x), -1.65);
        if (mla > PI_ / 2.0) mla = PI_ / 2.0;
        p.J = cos(mla) * cos(mla);
    }
    p.gma = 0.5 * (p.om + p.J * p.del);
    p.h = sqrt(p.a * p.a + 2.0 * p.a * p.gma);
    p.S1 = exp(-p.h * pait);
    p.u1 = p.a + p.gma * (1.0 - 1.0 / gref);
    double u2 = p.a + p.gma * (1.0 - gref);
    p.D1 = (p.a + p.gma + p.h) * (p.u1 - p.h) * 1.0 / p.S1 - (p.a + p.gma - p.h) * (p.u1 + p.h) * p.S1;
    p.D2 = (u2 + p.h) * 1.0 / p.S1 - (u2 - p.h) * p.S1;
    p.p1 = (p.gma / (p.D1 * p.S1)) * (p.u1 - p.h);
    p.p2 = (-p.gma * p.S1 / p.D1) * (p.u1 + p.h);
    p.p3 = (1.0 / (p.D2 * p.S1)) * (u2 + p.h);
    p.p4 = (-p.S1 / p.D2) * (u2 - p.h);
    return p;
}

/* cpp:164-185 twostreamdirCpp */
typedef struct {
    double sig, p5, p6, p7, p8, p9, p10;
} tsdir_t;

static tsdir_t twostreamdir_params(double pait, double om, double a, double gma, double J, double del,
                                   double h, double gref, double kd, double u1, double S1, double D1,
                                   double D2) {
    tsdir_t p;
    double sig = kd * kd + gma * gma - pow((a + gma), 2.0);
    double ss = 0.5 * (om + J * del / kd) * kd;
    double sstr = om * kd - ss;
    double S2 = exp(-kd * pait);
    double u2 = a + gma * (1.0 - gref);
    p.p5 = -ss * (a + gma - kd) - gma * sstr;
    double v1 = ss - (p.p5 * (a + gma + kd)) / sig;
    double v2 = ss - gma - (p.p5 / sig) * (u1 + kd);
    p.p6 = (1.0 / D1) * ((v1 / S1) * (u1 - h) - (a + gma - h) * S2 * v2);
    p.p7 = (-1.0 / D1) * ((v1 * S1) * (u1 + h) - (a + gma + h) * S2 * v2);
    p.sig = -sig;
    p.p8 = sstr * (a + gma + kd) - gma * ss;
    double v3 = (sstr + gma * gref - (p.p8 / p.sig) * (u2 - kd)) * S2;
    p.p9 = (-1 / D2) * ((p.p8 / (p.sig * S1)) * (u2 + h) + v3);
    p.p10 = (1 / D2) * (((p.p8 * S1) / p.sig) * (u2 - h) + v3);
    return p;
}

/* cpp:294-299 zeroplanedisCpp */
double orc_zeroplanedis(double h, double pai) {
    if (pai < 0.001) pai = 0.001;
    return (1.0 - (1.0 - exp(-sqrt(7.5 * pai))) / sqrt(7.5 * pai)) * h;
}

/* cpp:302-310 roughlengthCpp */
double orc_roughlength(double h, double pai, double d, double psi_h) {
    double Be = sqrt(0.003 + (0.2 * pai) / 2);
    double zm = (h - d) * exp(-KA / Be) * exp(KA * psi_h);
    if (zm > (0.9 * (h - d))) zm = 0.9 * (h - d);
    if (zm < 0.0005) zm = 0.0005;
    return zm;
}

/* cpp:373-380 gturbCpp */
static double gturb(double uf, double d, double zm, double zref, double ph, double psi_h, double gmin) {
    double z0 = 0.2 * zm + d;
    double ln = log((zref - d) / (z0 - d));
    double g = (KA * ph * uf) / (ln + psi_h);
    if (g < gmin) g = gmin;
    return g;
}

/* cpp:382-389 psiwfromthetaCpp */
static double psiwfromtheta(double theta, double Smax, double psi_e, double b) {
    psi_e = fabs(psi_e);
    double Se = theta / Smax;
    if (Se > 1.0) Se = 1.0;
    return -psi_e * pow(Se, -b) * 0.01;
}

/* cpp:391-440 stomparamsCpp; struct hdr:82-87 */
typedef struct {
    double Rsmx, psiw0, kk, rat;
} stomp_t;

static stomp_t stomparams(double hgt, double lat, double x) {
    stomp_t o;
    o.Rsmx = 420.0; o.psiw0 = -3.1; o.kk = 0.34; o.rat = 0.9;          /* C3 grass */
    if (hgt < 1.0 && fabs(lat) < 22.5) {                                /* C4 grass */
        o.Rsmx = 450.0; o.psiw0 = -2.7; o.kk = 0.39; o.rat = 0.9;
    }
    if (hgt >= 1.0 && hgt < 7.0) {                                      /* shrub */
        o.Rsmx = 430.0; o.psiw0 = -4.0; o.kk = 0.28; o.rat = 0.75;
    }
    if (hgt >= 7.0) {
        if (fabs(lat) < 22.5) {                                         /* tropical broadleaf */
            o.Rsmx = 500.0; o.psiw0 = -1.75; o.kk = 0.67; o.rat = 0.4;
        } else if (x < 0.8 || fabs(lat) > 58.0) {                       /* needleleaf */
            o.Rsmx = 420.0; o.psiw0 = -4.09; o.kk = 0.29; o.rat = 0.6;
        } else {                                                        /* deciduous */
            o.Rsmx = 500.0; o.psiw0 = -2.51; o.kk = 0.46; o.rat = 0.45;
        }
    }
    return o;
}

/* cpp:442-458 stomcondCpp */
static double stomcond(double Rswabs, double theta, double gsmax, double Smax, double psi_e, double b,
                       stomp_t st) {
    if (Rswabs <= 0.0) return 0.0;
    if (Rswabs > st.Rsmx) Rswabs = st.Rsmx;
    double gs = gsmax * pow(2.0, -(st.Rsmx - Rswabs) / (0.2 * st.Rsmx));
    double thetan = st.rat * theta + (1 - st.rat) * THETAM;
    double psiw = psiwfromtheta(thetan, Smax, psi_e, b);
    if (psiw < st.psiw0) psiw = st.psiw0;
    double mu = 1.0 - (exp(-st.kk * psiw) - 1.0) / (exp(-st.kk * st.psiw0) - 1.0);
    double gs2 = mu * gsmax;
    if (gs > gs2) gs = gs2;
    return gs;
}

/* cpp:460-477 canopycondCpp */
static double canopycond(double Rsw, double Rdif, double k, double om, double theta, double gsmax,
                         double PAI, double Smax, double psi_e, double b, stomp_t st) {
    double Gs = 9999.99;
    if (!isnan(om)) {
        double P_sun = (1.0 - exp(-k * PAI)) / k;
        double P_shade = PAI - P_sun;
        double Rshade_abs = Rdif * ((1.0 - exp(-PAI)) / PAI) * (1.0 - om);
        double Rsun_abs = (Rsw - Rdif) * k * (1 - om) + Rshade_abs;
        double gs_sun = stomcond(Rsun_abs, theta, gsmax, Smax, psi_e, b, st);
        double gs_shade = stomcond(Rshade_abs, theta, gsmax, Smax, psi_e, b, st);
        Gs = gs_sun * P_sun + gs_shade * P_shade;
    }
    return Gs;
}

/* cpp:480-490 satvapCpp (ice branch for tc <= 0) */
double orc_satvap(double tc) {
    if (tc > 0) return 0.61078 * exp(17.27 * tc / (tc + 237.3));
    return 0.61078 * exp(21.875 * tc / (tc + 265.5));
}

/* cpp:561-572 maCpp: circular trailing mean */
static void ma_circ(const double *x, int m, int n, double *y) {
    for (int i = 0; i < m; ++i) {
        double sum = 0.0;
        for (int j = 0; j < n; ++j) sum += x[(i - j + m) % m];
        y[i] = sum / n;
    }
}

/* cpp:597-627 manCpp */
void orc_man(const double *x, int m, int n, double *z) {
    if (n <= 48) {
        ma_circ(x, m, n, z);
        return;
    }
    int numDays = m / 24;
    double *d = (double *)calloc((size_t)(numDays > 0 ? numDays : 1), sizeof(double));
    double *y = (double *)calloc((size_t)(numDays > 0 ? numDays : 1), sizeof(double));
    double *zz = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (int i = 0; i < numDays; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 24; ++j) sum += x[i * 24 + j];
        d[i] = sum / 24.0;
    }
    int n2 = n / 24;
    ma_circ(d, numDays, n2, y);
    for (int i = 0; i < numDays; ++i)
        for (int j = 0; j < 24; ++j) zz[i * 24 + j] = y[i];
    ma_circ(zz, m, 24, z);
    free(d); free(y); free(zz);
}

/* cpp:517-559 hourtodayCpp(..., rephour = true) for stat in {max,min,mean} */
static void hourtoday(const double *h, int m, int stat, double *out) {
    int numDays = m / 24;
    for (int i = 0; i < numDays; ++i) {
        double s = h[i * 24];
        if (stat == 0) { for (int j = 1; j < 24; ++j) s = fmax(s, h[i * 24 + j]); }
        else if (stat == 1) { for (int j = 1; j < 24; ++j) s = fmin(s, h[i * 24 + j]); }
        else { s = 0.0; for (int j = 0; j < 24; ++j) s += h[i * 24 + j]; s /= 24; }
        for (int j = 0; j < 24; ++j) out[i * 24 + j] = s;
    }
}

/* cpp:628-636 soilpfun; hdr:110-114 */
typedef struct { double c1, c3, c4; } soilc_t;
static soilc_t soilpfun(double Vm, double Vq, double Mc, double rho) {
    (void)rho;
    soilc_t o;
    double frs = Vm + Vq;
    o.c1 = (0.57 + 1.73 * Vq + 0.93 * Vm) / (1.0 - 0.74 * Vq - 0.49 * Vm) - 2.8 * frs * (1.0 - frs);
    o.c3 = 1.0 + 2.6 * pow(Mc, -0.5);
    o.c4 = 0.03 + 0.7 * frs * frs;
    return o;
}

/* Row-block tests: a rank solving one block of a larger raster must use the mean of the
 * WHOLE raster (cpp:1004); the test installs it here (enable = 0 restores cpp:993-1004). */
static int g_mean_override_on = 0;
static double g_mean_override = 0.0;
void orc_set_twi_mean_override(double mean, int enable) {
    g_mean_override = mean;
    g_mean_override_on = enable;
}

/* cpp:975-1019 soildCppm: tadd = log(twi)/tfact - mean over non-NA cells */
void orc_soild_tadd(const double *twi, int64_t n_cells, int64_t rows, int64_t cols, double tfact,
                    double *tadd) {
    /* the reference sums i-outer / j-inner (row-major walk of a column-major matrix) */
    double sum = 0.0;
    int64_t count = 0;
    (void)n_cells;
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < cols; ++j) {
            double v = twi[i + rows * j];
            if (!isnan(v)) {
                sum += log(v) / tfact;
                count++;
            }
        }
    double me = sum / (double)count;
    if (g_mean_override_on) me = g_mean_override;
    for (int64_t c = 0; c < rows * cols; ++c) {
        double v = twi[c];
        tadd[c] = isnan(v) ? orc_na_real() : log(v) / tfact - me;
    }
}

/* cpp:1021-1032 soildCpp */
double orc_soild(double soilm, double Smin, double Smax, double tadd) {
    double rge = Smax - Smin;
    double theta = (soilm - Smin) / rge;
    if (theta > 0.9999) theta = 0.9999;
    if (theta < 0.0001) theta = 0.0001;
    double lt = log(theta / (1 - theta));
    double sm = lt + tadd;
    sm = 1 / (1 + exp(-sm));
    return sm * rge + Smin;
}

/* cpp:1034-1084 twostreamdif; hdr:46-68 tirstruct */
typedef struct {
    double albd, Rddn_g, Rdup_z, Rddn_z, gi, trdn, trdu, amx, pait, paiaa, om, omp, a, gma, J, del, h,
        u1, S1, D1, D2;
} tir_t;

static tir_t twostreamdif(double pai, double paia, double x, double lref, double ltra, double clump,
                          double gref) {
    tir_t o;
    o.pait = pai / (1.0 - clump);
    tsdif_t p = twostreamdif_params(o.pait, x, lref, ltra, gref);
    o.om = p.om; o.omp = 0.5 * o.om; o.a = p.a; o.gma = p.gma; o.J = p.J; o.del = p.del; o.h = p.h;
    o.u1 = p.u1; o.S1 = p.S1; o.D1 = p.D1; o.D2 = p.D2;
    o.gi = 0.0;
    if (clump > 0.0) o.gi = pow(clump, paia / pai);
    if (o.gi > 0.99) o.gi = 0.99;
    double giu = 0.0;
    if (clump > 0.0) giu = pow(clump, (pai - paia) / pai);
    if (giu > 0.99) giu = 0.99;
    double trd = o.gi * o.gi;
    o.trdn = pow(clump, 2.0);
    o.trdu = giu * giu;
    o.paiaa = paia / (1.0 - o.gi);
    o.amx = gref;
    if (o.amx < lref) o.amx = lref;
    o.albd = (1.0 - o.trdn * o.trdn) * (p.p1 + p.p2) + o.trdn * o.trdn * gref;
    if (o.albd > o.amx) o.albd = o.amx;
    if (o.albd < 0.01) o.albd = 0.01;
    o.Rddn_g = (1.0 - o.trdn) * (p.p3 * exp(-p.h * o.pait) + p.p4 * exp(p.h * o.pait)) + o.trdn;
    if (o.Rddn_g > 1.0) o.Rddn_g = 1.0;
    if (o.Rddn_g < 0.0) o.Rddn_g = 0.0;
    o.Rdup_z = (1.0 - o.trdu * o.trdn) * (p.p1 * exp(-p.h * o.paiaa) + p.p2 * exp(p.h * o.paiaa)) +
               o.trdu * o.trdn * gref;
    if (o.Rdup_z > 1.0) o.Rdup_z = 1.0;
    if (o.Rdup_z < 0.0) o.Rdup_z = 0.0;
    o.Rddn_z = (1.0 - trd) * (p.p3 * exp(-p.h * o.paiaa) + p.p4 * exp(p.h * o.paiaa)) + trd;
    if (o.Rddn_z > 1.0) o.Rddn_z = 1.0;
    if (o.Rddn_z < 0.0) o.Rddn_z = 0.0;
    return o;
}

/* cpp:1086-1178 twostreamCpp; hdr:69-81 radmodel2 */
typedef struct {
    double radGsw, radGlw, radCsw, radClw, Rbdown, Rddown, Rdup, radLsw, radLpar, lwout, zend;
} rad_t;

static rad_t twostream(double pai, double clump, double gref, double svfa, double si, double tc,
                       double Rsw, double Rdif, double Rlw, orc_solmodel solp, orc_kstruct kp,
                       tsdir_t d, tir_t tir) {
    rad_t o;
    if (Rsw > 0.0) {
        double cosz = cos(solp.zenr);
        if (pai > 0.0) {
            double trbn = pow(clump, kp.Kc);
            if (trbn > 0.999) trbn = 0.999;
            if (trbn < 0.0) trbn = 0.0;
            double trb = pow(tir.gi, kp.Kc);
            if (trb > 0.999) trb = 0.999;
            if (trb < 0.0) trb = 0.0;
            double albb = (1.0 - tir.trdn * trbn) * ((d.p5 / -d.sig) + d.p6 + d.p7) + tir.trdn * trbn * gref;
            if (albb > tir.amx) albb = tir.amx;
            if (albb < 0.01) albb = 0.01;
            double Rdbdn_g = (1.0 - trbn) * ((d.p8 / d.sig) * exp(-kp.kd * tir.pait) +
                                             d.p9 * exp(-tir.h * tir.pait) + d.p10 * exp(tir.h * tir.pait));
            if (Rdbdn_g > tir.amx) Rdbdn_g = tir.amx;
            if (Rdbdn_g < 0.0) Rdbdn_g = 0.0;
            double Rdbup_z = (1.0 - tir.trdu * trbn) * ((d.p5 / -d.sig) * exp(-kp.kd * tir.paiaa) +
                                                        d.p6 * exp(-tir.h * tir.paiaa) +
                                                        d.p7 * exp(tir.h * tir.paiaa)) +
                             tir.trdu * trbn * gref;
            if (Rdbup_z > tir.amx) Rdbup_z = tir.amx;
            if (Rdbup_z < 0.0) Rdbup_z = 0.0;
            double Rdbdn_z = (1.0 - trb) * ((d.p8 / d.sig) * exp(-kp.kd * tir.paiaa) +
                                            d.p9 * exp(-tir.h * tir.paiaa) + d.p10 * exp(tir.h * tir.paiaa));
            if (Rdbdn_z > tir.amx) Rdbdn_z = tir.amx;
            if (Rdbdn_z < 0.0) Rdbdn_z = 0.0;
            double Rbeam = (Rsw - Rdif) / cosz;
            if (Rbeam > 1352.0) Rbeam = 1352.0;
            double Rb = Rbeam * cosz;
            double trg = trb + (1 - trb) * exp(-kp.kd * tir.pait);
            double Rbc = (trg * si + (1 - trg) * cosz) * Rbeam;
            double Rbdn_g = trbn + (1.0 - trbn) * exp(-kp.kd * tir.pait);
            if (Rbdn_g > 1.0) Rbdn_g = 1.0;
            if (Rbdn_g < 0.0) Rbdn_g = 0.0;
            o.radGsw = (1.0 - gref) * (tir.Rddn_g * Rdif * svfa + Rdbdn_g * Rb + Rbdn_g * Rbeam * si);
            double maxg = (1.0 - gref) * (Rdif * svfa + Rbeam * si);
            if (o.radGsw > maxg) o.radGsw = maxg;
            o.radCsw = (1.0 - tir.albd) * Rdif * svfa + (1.0 - albb) * Rbc;
            o.Rbdown = (trb + (1.0 - trb) * exp(-kp.kd * tir.paiaa)) * Rbeam;
            o.Rddown = tir.Rddn_z * Rdif * svfa + Rdbdn_z * Rb;
            o.Rdup = tir.Rdup_z * Rdif * svfa + Rdbup_z * Rb;
            o.radLsw = 0.5 * (1.0 - tir.om) * (o.Rddown + o.Rdup + kp.k * cosz * o.Rbdown);
            o.radLpar = 0.5 * (1.0 - tir.omp) * (o.Rddown + o.Rdup + kp.k * cosz * o.Rbdown);
        } else {
            o.Rbdown = (Rsw - Rdif) / cosz;
            o.Rddown = Rdif * svfa;
            o.Rdup = gref * (Rdif * svfa + (Rsw - Rdif));
            o.radGsw = (1.0 - gref) * (svfa * Rdif + si * o.Rbdown);
            o.radCsw = o.radGsw;
            o.radLsw = 0.0;
            o.radLpar = 0.0;
        }
    } else {
        o.Rbdown = 0.0; o.Rddown = 0.0; o.Rdup = 0.0; o.radGsw = 0.0; o.radCsw = 0.0;
        o.radLsw = 0.0; o.radLpar = 0.0;
    }
    if (pai > 0.0) {
        double trdif = (1.0 - tir.trdn) * exp(-tir.pait) + tir.trdn;
        o.lwout = 0.97 * SB * radem(tc);
        o.radGlw = 0.97 * (trdif * svfa * Rlw + (1.0 - trdif) * o.lwout);
        o.radClw = 0.97 * svfa * Rlw;
    } else {
        o.lwout = 0.97 * SB * radem(tc);
        o.radGlw = 0.97 * svfa * Rlw;
        o.radClw = o.radGlw;
    }
    o.zend = solp.zend;
    return o;
}

/* cpp:1179-1187 windtiCpp; hdr:93-97 */
typedef struct { double d, zm, a; } tiw_t;
static tiw_t windti(double h, double pai) {
    tiw_t o;
    o.d = orc_zeroplanedis(h, pai);
    o.zm = orc_roughlength(h, pai, o.d, 0.0);
    if (o.zm < 1e-6) o.zm = 1e-6;
    o.a = pai / h;
    return o;
}

/* cpp:1189-1218 windCpp; hdr:98-102 */
typedef struct { double uf, uz, gHa; } wind_t;
static wind_t wind(double reqhgt, double zref, double h, double pai, double uref, double umu, double ws,
                   tiw_t tiw) {
    (void)pai;
    wind_t o;
    if (isnan(ws)) ws = 1.0;
    if (ws < 0.05) ws = 0.05;
    double ufs = (KA * uref) / log((zref - tiw.d) / tiw.zm);
    o.uf = ufs * umu * ws;
    if (o.uf < 0.001) o.uf = 0.001;
    o.uz = o.uf;
    if (reqhgt > 0) {
        if (reqhgt >= h) {
            o.uz = (o.uf / KA) * log((reqhgt - tiw.d) / tiw.zm);
        } else {
            double uh = (o.uf / KA) * log((h - tiw.d) / tiw.zm);
            if (uh < o.uf) uh = o.uf;
            double Be = o.uf / uh;
            if (Be < 0.001) Be = 0.001;
            double Lc = pow(0.25 * tiw.a, -1.0);
            double Lm = 2 * pow(Be, 3.0) * Lc;
            o.uz = uh * exp(Be * (reqhgt - h) / Lm);
        }
        if (o.uz > uref) o.uz = uref;
    }
    o.gHa = gturb(o.uf, tiw.d, tiw.zm, zref, 43, 0, 0.0001);
    return o;
}

/* cpp:1220-1247 PenmanMonteith2Cpp; hdr:103-109 */
typedef struct { double Ts, H, L, Rem, mu; } penmon_t;
static penmon_t penman2(double Rabs, double gHa, double gV, double tc, double mxtc, double pk, double ea,
                        double es, double G, double surfwet, double tdew) {
    double De = orc_satvap(tc + 0.5) - orc_satvap(tc - 0.5);
    double gHr = gHa + (4 * 0.97 * SB * pow(tc + 273.15, 3.0)) / 29.3;
    double Rem = 0.97 * SB * radem(tc);
    double la;
    if (tc >= 0) la = 45068.7 - 42.8428 * tc;
    else la = 51078.69 - 4.338 * tc - 0.06367 * tc * tc;
    double m = la * (gV / pk);
    double L = m * (es - ea) * surfwet;
    double dT = (Rabs - Rem - L - G) / (29.3 * gHr + m * De);
    double dTmx = -0.6273 * mxtc + 49.79;
    if (dT > dTmx) dT = dTmx;
    if (dT > 80.0) dT = 80.0;
    penmon_t o;
    o.Ts = dT + tc;
    if (o.Ts < tdew) o.Ts = tdew;
    o.H = 29.3 * gHa * (o.Ts - tc);
    o.L = m * (orc_satvap(o.Ts) - ea) * surfwet;
    o.Rem = 0.97 * SB * radem(o.Ts);
    o.mu = la * (43.0 / pk);
    return o;
}

/* hdr:119-128 soilpstruct */
typedef struct { double Smax, Smin, soilb, psi_e, Vq, Vm, Mc, rho; } soilp_t;

/* cpp:1249-1260 soilcondCpp */
typedef struct { double k, DD; } soilk_t;
static soilk_t soilcond(double rho, double soilm, soilc_t sp) {
    double cs = (2400 * rho / 2.64 + 4180.0 * soilm);
    double ph = (rho * (1.0 - soilm) + soilm) * 1000.0;
    double c2 = 1.06 * rho * soilm;
    soilk_t o;
    o.k = sp.c1 + c2 * soilm - (sp.c1 - sp.c4) * exp(-pow(sp.c3 * soilm, 4.0));
    double kap = o.k / (cs * ph);
    o.DD = pow(2.0 * kap / OMDY, 0.5);
    return o;
}

/* cpp:1262-1275 soiltempG0 */
typedef struct { double Tg, Rnet, surfwet, radabs; } soilG0_t;
static soilG0_t soiltempG0(double tc, double es, double ea, double pk, double radGsw, double radGlw,
                           double tdew, double gHa, double soilm, double mxtc, soilp_t sp) {
    soilG0_t o;
    o.radabs = radGsw + radGlw;
    double matric = -fabs(sp.psi_e) * pow(soilm / sp.Smax, -sp.soilb);
    o.surfwet = exp((0.018 * matric) / (8.31 * (tc + 273.15)));
    if (o.surfwet > 1.0) o.surfwet = 1.0;
    penmon_t pm = penman2(o.radabs, gHa, gHa, tc, mxtc, pk, ea, es, 0.0, o.surfwet, tdew);
    o.Tg = pm.Ts;
    o.Rnet = o.radabs - pm.Rem;
    return o;
}

/* cpp:1277-1296 soiltemp_hrCpp */
typedef struct { double Tg, G, DD; } soilhr_t;
static soilhr_t soiltemp_hr(double tc, double es, double ea, double pk, double radabs, double surfwet,
                            double tdew, double gHa, double soilm, double mxtc, double Gp, double dtr,
                            double dtrp, double muGp, double kp, double Rdmx, soilc_t sc, soilp_t sp) {
    double dtR = dtr / dtrp;
    soilk_t kd = soilcond(sp.rho, soilm, sc);
    double Gmu = dtR * (kd.k * muGp) / (kp * kd.DD);
    soilhr_t o;
    o.G = Gp * Gmu;
    if (o.G > 0.6 * Rdmx) o.G = 0.6 * Rdmx;
    if (o.G < -0.6 * Rdmx) o.G = -0.6 * Rdmx;
    penmon_t pm = penman2(radabs, gHa, gHa, tc, mxtc, pk, ea, es, o.G, surfwet, tdew);
    o.Tg = pm.Ts;
    o.DD = kd.DD;
    return o;
}

/* cpp:1298-1313 TVabove */
typedef struct { double Tz, ez; } abovecan_t;
static abovecan_t TVabove(double reqhgt, double zref, double h, double d, double zm, double T0, double tc,
                          double ea, double surfwet) {
    (void)h;
    double zh = 0.2 * zm;
    double estl = orc_satvap(T0);
    abovecan_t o;
    if (reqhgt > (d + zh)) {
        double lnr = log((reqhgt - d) / zh) / log((zref - d) / zh);
        o.Tz = tc + (T0 - tc) * (1 - lnr);
        o.ez = ea + (estl - ea) * surfwet * (1 - lnr);
    } else {
        o.Tz = T0;
        o.ez = ea + (estl - ea) * surfwet;
    }
    return o;
}

/* cpp:1316-1331 mincondCpp */
static double mincond(double leafabs, double gs, double tc, double leafd) {
    double Rnet = leafabs - 0.97 * SB * radem(tc);
    double rs = 500.0;
    if (gs > 0.0) rs = 1 / gs;
    if (rs > 500.0) rs = 500.0;
    double Hlf = 1.09767 * pow(rs, 0.2672778);
    double Hf = -1.0 / (1.0 + exp(2.0 - Hlf));
    double H = Hf * Rnet;
    double gmin = 0.0463 * pow(fabs(H) / leafd, 0.2);
    if (gmin < 0.05) gmin = 0.05;
    return gmin;
}

/* cpp:1333-1364 leaftemp */
typedef struct { double tleaf, H, L, lwdn, lwup; } leaft_t;
static leaft_t leaftemp(double Tcan, double Tg, double tc, double mxtc, double pk, double ea, double es,
                        double uz, double tdew, double surfwet, double radLsw, double Rddown,
                        double Rbdown, double Rlw, double pai, double paia, double leafd, double gsmax,
                        double PARabs, double theta, double Smax, double psi_e, double soilb, stomp_t st) {
    (void)Rddown; (void)Rbdown;
    leaft_t o;
    double lwcan = 0.97 * SB * radem(Tcan);
    double lwgro = 0.97 * SB * radem(Tg);
    double paig = pai - paia;
    o.lwup = exp(-paig) * lwgro + (1 - exp(-paig)) * lwcan;
    o.lwdn = exp(-paia) * Rlw + (1 - exp(-paia)) * lwcan;
    double lwabs = 0.97 * 0.5 * (o.lwup + o.lwdn);
    double leafabs = radLsw + lwabs;
    double gh = 0.135 * sqrt(uz / leafd) * 1.4;
    double gmin = mincond(leafabs, 999.99, Tcan, leafd);
    if (gh < gmin) gh = gmin;
    double gV = gh;
    if (gsmax < 999.99) {
        gV = 0.0;
        double gs = stomcond(PARabs, theta, gsmax, Smax, psi_e, soilb, st);
        gmin = mincond(leafabs, gs, Tcan, leafd);
        if (gh < gmin) gh = gmin;
        if (gs > 0.0) gV = 1 / (1 / gh + 1 / gs);
    }
    penmon_t pm = penman2(leafabs, gh, gV, tc, mxtc, pk, ea, es, 0.0, surfwet, tdew);
    o.tleaf = pm.Ts;
    o.H = pm.H;
    o.L = pm.L;
    return o;
}

/* cpp:1365-1380 rhcanopy */
static double rhcanopy(double uf, double h, double d, double z) {
    double a2 = 0.4 * (1.0 - (d / h)) / pow(1.25, 2);
    double inth = 4.293251 * h;
    if (z != h) {
        inth = (2.0 * h *
                ((48 * atan((sqrt(5.0) * sin((PI_ * z) / h)) / (cos((PI_ * z) / h) + 1))) / pow(5.0, 1.5) +
                 (32.0 * sin((PI_ * z) / h)) /
                     ((cos((PI_ * z) / h) + 1) *
                      ((25.0 * pow(sin((PI_ * z) / h), 2.0)) / pow((cos((PI_ * z) / h) + 1.0), 2.0) + 5.0)))) /
               PI_;
    }
    double mu = uf / (a2 * h) * 1.0 / (uf * uf);
    double rHa = inth * mu;
    if (rHa < 0.001) rHa = 0.001;
    return rHa;
}

/* cpp:1381-1409 TVbelow */
static double TVbelow(double zref, double z, double d, double h, double pai, double uf, double leafden,
                      double Flux, double Fluxz, double SH, double SG, double mxnear) {
    (void)zref;
    double Rc = rhcanopy(uf, h, d, h);
    double Kc = h / Rc;
    double Kg = 1.0 / rhcanopy(uf, h, d, z);
    double Kh = 1.0 / (Rc - rhcanopy(uf, h, d, z));
    Kg = Kg / z;
    Kh = Kh / (h - z);
    double SC = SH + Flux / Kc;
    double farg = (Kg * SG + Kh * SH + Kc * SC) / (Kg + Kh + Kc);
    double SN = Fluxz * leafden;
    double near = (3.047519 + 0.128642 * log(pai)) * SN;
    if (fabs(near) > mxnear) {
        if (near > 0.0) near = mxnear;
        else near = -mxnear;
    }
    if (isnan(near)) near = 0;
    return near + farg;
}

static double max4(double a, double b, double c, double d) {
    /* std::max({a,b,c,d}) keeps the first maximal element under operator< */
    double m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    if (m < d) m = d;
    return m;
}
static double min4(double a, double b, double c, double d) {
    double m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    if (d < m) m = d;
    return m;
}

/* cpp:1411-1472 TVaboveground; hdr:151-157 */
typedef struct { double Tz, tleaf, rh, lwdn, lwup; } above_t;
static above_t TVaboveground(double reqhgt, double zref, double tc, double pk, double ea, double es,
                             double tdew, double Rsw, double Rdif, double Rlw, double soilm, double hgt,
                             double pai, double paia, double vegx, double leafd, double leafden,
                             double Smin, double Smax, double psi_e, double soilb, double gsmax,
                             double mxtc, stomp_t st, tir_t tir, rad_t rv, tiw_t tiw, wind_t wv,
                             soilhr_t gv) {
    above_t o;
    double eT = orc_satvap(gv.Tg) - ea;
    if (eT < 0.001) eT = 0.001;
    double plf = 0.8753 - 1.7126 * log(eT);
    double gwet = 1.0 / (1.0 + exp(-plf));
    double surfwet = (soilm - Smin) / (Smax - Smin);
    if (surfwet > gwet) gwet = surfwet;
    /* NB the reference hands the zenith in DEGREES to cankCpp here (cpp:1425) */
    orc_kstruct kp = orc_cank(rv.zend, vegx, cos(rv.zend * TORAD));
    double gS = canopycond(Rsw, Rdif, kp.k, tir.omp, soilm, gsmax, pai, Smax, psi_e, soilb, st);
    double gV = 0.0;
    if (gS > 0.0) gV = 1.0 / (1.0 / wv.gHa + 1 / gS);
    double Rabs = rv.radCsw + rv.radClw;
    penmon_t pm = penman2(Rabs, wv.gHa, gV, tc, mxtc, pk, ea, es, gv.G, surfwet, tdew);
    double Tcan = pm.Ts;
    double ez = 0;
    if (reqhgt >= hgt) {
        abovecan_t tv = TVabove(reqhgt, zref, hgt, tiw.d, tiw.zm, Tcan, tc, ea, surfwet);
        o.Tz = tv.Tz;
        o.tleaf = Tcan;
        o.lwup = 0.97 * SB * radem(Tcan);
        o.lwdn = Rlw;
        ez = tv.ez;
    } else {
        leaft_t tvl = leaftemp(Tcan, gv.Tg, tc, mxtc, pk, ea, es, wv.uz, tdew, surfwet, rv.radLsw,
                               rv.Rddown, rv.Rbdown, Rlw, pai, paia, leafd, gsmax, rv.radLpar, soilm, Smax,
                               psi_e, soilb, st);
        o.tleaf = tvl.tleaf;
        double Flux = pm.H * (1.0 - exp(-pai));
        double Fluxz = tvl.H;
        abovecan_t tv = TVabove(hgt, zref, hgt, tiw.d, tiw.zm, Tcan, tc, ea, surfwet);
        double SH = tv.Tz * 29.3 * 43.0;
        double SG = gv.Tg * 29.3 * 43.0;
        double mxnear = fabs(o.tleaf - tv.Tz) * 29.3 * 43.0;
        o.Tz = TVbelow(zref, reqhgt, tiw.d, hgt, pai, wv.uf, leafden, Flux, Fluxz, SH, SG, mxnear) /
               (29.3 * 43.0);
        Flux = pm.L * (1.0 - exp(-pai));
        Fluxz = tvl.L;
        SH = tv.ez * pm.mu;
        SG = orc_satvap(gv.Tg) * gwet * pm.mu;
        mxnear = fabs(orc_satvap(o.tleaf) - tv.ez) * pm.mu;
        ez = TVbelow(zref, reqhgt, tiw.d, hgt, pai, wv.uf, leafden, Flux, Fluxz, SH, SG, mxnear) / pm.mu;
        o.lwdn = tvl.lwdn;
        o.lwup = tvl.lwup;
    }
    o.rh = (ez / orc_satvap(o.Tz)) * 100.0;
    if (o.rh > 100.0) o.rh = 100.0;
    double tmx = max4(o.tleaf, tc, gv.Tg, Tcan) + 2.0;
    double tmn = min4(o.tleaf, tc, gv.Tg, Tcan) - 2.0;
    if (o.Tz > tmx) o.Tz = tmx;
    if (o.Tz < tmn) o.Tz = tmn;
    return o;
}

/* cpp:1474-1539 Tbelowgroundv.  Tz must not alias Tg. */
void orc_tbelowground(double reqhgt, const double *Tg, const double *Tgp, const double *Tbp, int tsteps,
                      double meanD, double mat, int hiy, int complete, double *Tz) {
    for (int i = 0; i < tsteps; ++i) Tz[i] = Tg[i];
    if (!(reqhgt < 0)) return;
    double nb = -118.35 * reqhgt / meanD;
    int n = (int)round(nb);
    if (complete) {
        if (n < tsteps) {
            orc_man(Tg, tsteps, n, Tz);
        } else {
            double sumT = 0;
            for (int i = 0; i < tsteps; ++i) sumT = sumT + Tg[i];
            double meanT = sumT / tsteps;
            for (int i = 0; i < tsteps; ++i) Tz[i] = meanT;
        }
        return;
    }
    /* hourtodayCpp returns (tsteps/24)*24 elements (cpp:517-519) and cpp:1501-1515 index them up to tsteps: with a
     * ragged last day the reference reads past the end.  Zero-filled full-length arrays here: the tail's rat is 0/0. */
    size_t nb_ = (size_t)(tsteps > 0 ? tsteps : 1) * sizeof(double);
    double *Tzd = (double *)calloc(1, nb_), *Tbpd = (double *)calloc(1, nb_);
    double *gmx = (double *)calloc(1, nb_), *gmn = (double *)calloc(1, nb_), *gme = (double *)calloc(1, nb_);
    double *pmx = (double *)calloc(1, nb_), *pmn = (double *)calloc(1, nb_), *pme = (double *)calloc(1, nb_);
    hourtoday(Tbp, tsteps, 2, Tbpd);
    hourtoday(Tg, tsteps, 0, gmx); hourtoday(Tg, tsteps, 1, gmn); hourtoday(Tg, tsteps, 2, gme);
    hourtoday(Tgp, tsteps, 0, pmx); hourtoday(Tgp, tsteps, 1, pmn); hourtoday(Tgp, tsteps, 2, pme);
    for (int i = 0; i < tsteps; ++i) {
        double rat = (gmx[i] - gmn[i]) / (pmx[i] - pmn[i]);
        double dif = gme[i] - pme[i];
        Tzd[i] = rat * (Tbp[i] - Tbpd[i]) + Tbpd[i] + dif;
    }
    if (nb > 1.0 && nb <= 24.0) {
        double w1 = 1.0 / nb, w2 = nb / 24.0;
        double wgt = w1 / (w1 + w2);
        for (int i = 0; i < tsteps; ++i) Tz[i] = wgt * Tg[i] + (1 - wgt) * Tzd[i];
    }
    if (nb > 24.0) {
        if (nb < hiy) {
            double w1 = 24.0 / nb, w2 = nb / hiy;
            double wgt = w1 / (w1 + w2);
            for (int i = 0; i < tsteps; ++i) Tz[i] = wgt * Tzd[i] + (1 - wgt) * mat;
        } else {
            for (int i = 0; i < tsteps; ++i) Tz[i] = mat;
        }
    }
    free(Tzd); free(Tbpd); free(gmx); free(gmn); free(gme); free(pmx); free(pmn); free(pme);
}

/* Half-away-from-zero rounding then C remainder, as cpp:2166-2167.  A negative
 * direction would index out of bounds in the reference; it is wrapped here. */
static int dir_index(double v, double step, int n) {
    int r = ((int)round(v / step)) % n;
    if (r < 0) r += n;
    return r;
}

/*
 * Grid drivers: runmicro1Cpp cpp:2052-2337 (array_forcing == 0) and
 * runmicro2Cpp cpp:2340-2621 (array_forcing == 1).  Outputs must be
 * [rows,cols,tsteps] buffers (or NULL); they are filled with NA_real_ first.
 */
int orc_run_grid(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out) {
    const int64_t rows = in->rows, cols = in->cols, N = rows * cols;
    const int tsteps = (int)in->tsteps;
    const int ndays = tsteps / 24;
    const int af = in->array_forcing;
    const int layered = in->veg_layers > 1;
    const int nlyrs = layered ? in->veg_layers : 1;
    const double reqhgt = opt->reqhgt, zref = opt->zref;
    const double na = orc_na_real();
    double *O[MCF_NOUT];
    for (int v = 0; v < MCF_NOUT; ++v) {
        O[v] = (opt->out[v] && out->var[v]) ? out->var[v] : NULL;
        if (O[v]) for (int64_t q = 0; q < N * (int64_t)tsteps; ++q) O[v][q] = na;
    }
    /* per-timestep quantities (vector forcing) cpp:2153-2169 */
    int *sindex = (int *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(int));
    int *windex = (int *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(int));
    orc_solmodel *sol = (orc_solmodel *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(orc_solmodel));
    double mxtc_vec = -273.15;
    for (int k = 0; k < tsteps; ++k) {
        windex[k] = dir_index(in->clim.winddir[k], 45.0, 8);
        if (!af) {
            sol[k] = orc_solposition(in->lat, in->lon, in->obstime.year[k], in->obstime.month[k],
                                     in->obstime.day[k], in->obstime.hour[k]);
            sindex[k] = dir_index(sol[k].azid, 15.0, 24);
            if (in->clim.tc[k] > mxtc_vec) mxtc_vec = in->clim.tc[k];
        }
    }
    int hiy = 365 * 24;
    if (tsteps > 0 && in->obstime.year[0] % 4 == 0) hiy = 366 * 24;
    double *tadd = (double *)calloc((size_t)(N > 0 ? N : 1), sizeof(double));
    orc_soild_tadd(in->soilc.twi, N, rows, cols, opt->tfact, tadd);
    double *Tg = (double *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(double));
    double *DD = (double *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(double));
    double *Tzv = (double *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(double));
    double *Tgpv = (double *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(double));
    double *Tbpv = (double *)calloc((size_t)(tsteps > 0 ? tsteps : 1), sizeof(double));

    const mcf_vegp *V = &in->vegp;
    const mcf_soilc *S = &in->soilc;
    for (int64_t i = 0; i < rows; ++i) {
        for (int64_t j = 0; j < cols; ++j) {
            const int64_t c = i + rows * j;
            if (isnan(V->hgt[c])) continue;   /* layer 0 decides, cpp:2182 / 2759 */
            const double gref = S->gref[c];
            soilp_t spa;
            spa.Smax = S->Smax[c]; spa.Smin = S->Smin[c]; spa.soilb = S->soilb[c]; spa.psi_e = S->Psie[c];
            spa.Vq = S->Vq[c]; spa.Vm = S->Vm[c]; spa.Mc = S->Mc[c]; spa.rho = S->rho[c];
            soilc_t sc = soilpfun(S->Vm[c], S->Vq[c], S->Mc[c], S->rho[c]);
            memset(Tg, 0, sizeof(double) * (size_t)tsteps);
            memset(DD, 0, sizeof(double) * (size_t)tsteps);
            double mxtc = mxtc_vec;
            if (af) { /* cpp:2467-2471 */
                mxtc = -273.15;
                for (int k = 0; k < tsteps; ++k)
                    if (in->clim.tc[c + N * k] > mxtc) mxtc = in->clim.tc[c + N * k];
            }
            /* runmicro3Cpp/4Cpp (cpp:2760-2768, 3062-3070): one pass per vegetation layer over the
             * days dfsel assigns to it; with static vegetation a single layer covers every day */
            for (int lyr = 0; lyr < nlyrs; ++lyr) {
            const int64_t cl = c + N * lyr;
            const double hgt = V->hgt[cl], pai = V->pai[cl], x = V->x[cl];
            tir_t tir = twostreamdif(pai, V->paia[cl], x, V->leafr[cl], V->leaft[cl], V->clump[cl], gref);
            stomp_t st = stomparams(hgt, af ? in->lats[c] : in->lat, x);
            tiw_t tiw = windti(hgt, pai);
            const int lst = layered ? in->lyr_st[lyr] : 0;
            const int lnd = layered ? (in->lyr_ed[lyr] - in->lyr_st[lyr] + 1) / 24 : ndays;
            for (int dy = 0; dy < lnd; ++dy) {
                double Rmx = -999.9, tmx = -999.0, tmn = 999.0;
                double surfwet[24], radabs[24], soilmday[24], radCsw[24], radClw[24], Rddown[24], Rbdown[24],
                    radLsw[24], radLpar[24], uf[24], uzday[24], gHa[24], zendday[24];
                for (int hr = 0; hr < 24; ++hr) {
                    const int k = dy * 24 + hr + lst;
                    const int64_t idx = c + N * k;
                    const int64_t f = af ? idx : k; /* forcing index */
                    orc_solmodel solp;
                    double si;
                    int sidx;
                    if (af) { /* cpp:2497-2504 (shadowmask defaults to false) */
                        solp = orc_solposition(in->lats[c], in->lons[c], in->obstime.year[k],
                                               in->obstime.month[k], in->obstime.day[k], in->obstime.hour[k]);
                        si = orc_solarindex(S->slope[c], S->aspect[c], solp.zend, solp.azid, 0);
                        sidx = dir_index(solp.azid, 15.0, 24);
                    } else { /* cpp:2218-2223 (shadowmask = true) */
                        solp = sol[k];
                        si = orc_solarindex(S->slope[c], S->aspect[c], solp.zend, solp.azid, 1);
                        sidx = sindex[k];
                    }
                    zendday[hr] = solp.zend;
                    if (si < 0.0) si = 0.0;
                    double ws = S->wsa[windex[k] * N + c];
                    double ha = S->hor[sidx * N + c];
                    double sa = (PI_ / 2.0) - solp.zenr;
                    if (ha > tan(sa)) si = 0.0;
                    double soild = orc_soild(in->pointm.soilm[f], S->Smin[c], S->Smax[c], tadd[c]);
                    soilmday[hr] = soild;
                    if (O[MCF_OUT_SOILM]) O[MCF_OUT_SOILM][idx] = soild;
                    orc_kstruct kpp = orc_cank(solp.zenr, x, si);
                    tsdir_t tsd = twostreamdir_params(tir.pait, tir.om, tir.a, tir.gma, tir.J, tir.del, tir.h,
                                                      gref, kpp.kd, tir.u1, tir.S1, tir.D1, tir.D2);
                    rad_t rm = twostream(pai, V->clump[cl], gref, S->svfa[c], si, in->clim.tc[f],
                                         in->clim.swdown[f], in->clim.difrad[f], in->clim.lwdown[f], solp, kpp,
                                         tsd, tir);
                    radCsw[hr] = rm.radCsw; radClw[hr] = rm.radClw; Rddown[hr] = rm.Rddown;
                    Rbdown[hr] = rm.Rbdown; radLsw[hr] = rm.radLsw; radLpar[hr] = rm.radLpar;
                    if (O[MCF_OUT_RDIRDOWN]) O[MCF_OUT_RDIRDOWN][idx] = rm.Rbdown;
                    if (O[MCF_OUT_RDIFDOWN]) O[MCF_OUT_RDIFDOWN][idx] = rm.Rddown;
                    if (O[MCF_OUT_RSWUP]) O[MCF_OUT_RSWUP][idx] = rm.Rdup;
                    double reqhgt2 = reqhgt;
                    if (reqhgt2 < 0.00001) reqhgt2 = 0.00001;
                    wind_t wm = wind(reqhgt2, zref, hgt, pai, in->clim.windspeed[f], in->pointm.umu[f], ws, tiw);
                    uf[hr] = wm.uf; uzday[hr] = wm.uz; gHa[hr] = wm.gHa;
                    if (O[MCF_OUT_WINDSPEED]) O[MCF_OUT_WINDSPEED][idx] = wm.uz;
                    soilG0_t g0 = soiltempG0(in->clim.tc[f], in->clim.es[f], in->clim.ea[f], in->clim.pk[f],
                                             rm.radGsw, rm.radGlw, in->clim.tdew[f], wm.gHa, soild, mxtc, spa);
                    double Rval = fabs(g0.Rnet);
                    if (Rmx < Rval) Rmx = Rval;
                    if (tmx < g0.Tg) tmx = g0.Tg;
                    if (tmn > g0.Tg) tmn = g0.Tg;
                    surfwet[hr] = g0.surfwet;
                    radabs[hr] = g0.radabs;
                }
                double dtr = tmx - tmn;
                for (int hr = 0; hr < 24; ++hr) {
                    const int k = dy * 24 + hr + lst;
                    const int64_t idx = c + N * k;
                    const int64_t f = af ? idx : k;
                    soilhr_t gv = soiltemp_hr(in->clim.tc[f], in->clim.es[f], in->clim.ea[f], in->clim.pk[f],
                                              radabs[hr], surfwet[hr], in->clim.tdew[f], gHa[hr], soilmday[hr],
                                              mxtc, in->pointm.G[f], dtr, in->pointm.dtrp[f],
                                              in->pointm.muGp[f], in->pointm.kp[f], Rmx, sc, spa);
                    Tg[k] = gv.Tg;
                    DD[k] = gv.DD;
                    if (reqhgt >= 0.0) {
                        rad_t rv;
                        memset(&rv, 0, sizeof rv);
                        rv.zend = zendday[hr]; rv.radCsw = radCsw[hr]; rv.radClw = radClw[hr];
                        rv.Rddown = Rddown[hr]; rv.Rbdown = Rbdown[hr]; rv.radLsw = radLsw[hr];
                        rv.radLpar = radLpar[hr];
                        wind_t wv;
                        wv.uf = uf[hr]; wv.uz = uzday[hr]; wv.gHa = gHa[hr];
                        double reqhgt2 = reqhgt;
                        if (reqhgt2 < 0.00001) reqhgt2 = 0.00001;
                        above_t tv = TVaboveground(reqhgt2, zref, in->clim.tc[f], in->clim.pk[f], in->clim.ea[f],
                                                   in->clim.es[f], in->clim.tdew[f], in->clim.swdown[f],
                                                   in->clim.difrad[f], in->clim.lwdown[f], soilmday[hr], hgt, pai,
                                                   V->paia[cl], x, V->leafd[cl], V->leafden[cl], S->Smin[c],
                                                   S->Smax[c], S->Psie[c], S->soilb[c], V->gsmax[cl], mxtc, st,
                                                   tir, rv, tiw, wv, gv);
                        if (reqhgt > 0.0) {
                            if (O[MCF_OUT_TZ]) O[MCF_OUT_TZ][idx] = tv.Tz;
                        } else {
                            if (O[MCF_OUT_TZ]) O[MCF_OUT_TZ][idx] = Tg[k];
                        }
                        if (O[MCF_OUT_RLWDOWN]) O[MCF_OUT_RLWDOWN][idx] = tv.lwdn;
                        if (O[MCF_OUT_RLWUP]) O[MCF_OUT_RLWUP][idx] = tv.lwup;
                        if (reqhgt > 0.0) {
                            if (O[MCF_OUT_TLEAF]) O[MCF_OUT_TLEAF][idx] = tv.tleaf;
                            if (O[MCF_OUT_RELHUM]) O[MCF_OUT_RELHUM][idx] = tv.rh;
                        }
                    }
                }
            }
            } /* layers */
            if (reqhgt < 0.0 && O[MCF_OUT_TZ]) { /* cpp:2307-2320 / 2587-2604 */
                double sumD = 0.0;
                for (int k = 0; k < tsteps; ++k) {
                    sumD += DD[k];
                    if (!opt->complete) {
                        Tgpv[k] = in->pointm.Tg[af ? c + N * k : k];
                        Tbpv[k] = in->pointm.Tbp[af ? c + N * k : k];
                    }
                }
                double meanD = sumD / (double)tsteps;
                orc_tbelowground(reqhgt, Tg, Tgpv, Tbpv, tsteps, meanD, opt->mat, hiy, opt->complete, Tzv);
                for (int k = 0; k < tsteps; ++k) O[MCF_OUT_TZ][c + N * k] = Tzv[k];
            }
        }
    }
    free(sindex); free(windex); free(sol); free(tadd); free(Tg); free(DD); free(Tzv); free(Tgpv); free(Tbpv);
    return 0;
}
