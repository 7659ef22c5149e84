// mcf_pointmodel.cpp — the HOST-SIDE point model that produces the grid solver's `pointm` inputs
// (second half): soilmCpp, BigLeafCpp (+ RadswabsCpp, GFluxCpp), pointmprocess, weatherhgtCpp.
//
// These are O(tsteps) serial time series for ONE point (an iterated energy balance with running means over the
// series); the reference runs them on the CPU once per model run (R/Cppwrappers.R:119-138) and so does this file:
// they are host code by nature, not a fallback of anything — the grid solver (k_solve) never calls them and has no
// CPU path.  With them a caller can go from a weather table to `pointm` and into mcf_runmicro* without R.
//
// "cpp:" = the reference's src/microclimfCpp.cpp.  Evaluation order follows the reference so that the results can be
// compared with the test oracle to rounding (tests/test_pointmodel_cpu.py).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mcf.h"

namespace mcf {
int api_fail(int code, const std::string& msg);   // mcf_api.hip
}

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kToRad = kPi / 180.0;
constexpr double kSb = 5.67e-8;
constexpr double kThetam = 0.365;
constexpr double kKa = 0.4;
constexpr double kOmdy = (2.0 * kPi) / (24.0 * 3600.0);
using Vec = std::vector<double>;

inline double radem(double tc) { return pow(tc + 273.15, 4.0); }                      // cpp:24-26
inline double satvap(double tc) {                                                      // cpp:480-490
    return tc > 0 ? 0.61078 * exp(17.27 * tc / (tc + 237.3)) : 0.61078 * exp(21.875 * tc / (tc + 265.5));
}
inline double dewpoint(double ea) {                                                    // cpp:493-496
    const double l = log(ea / 0.6112);
    return 243.5 * l / (17.67 - l);
}
inline double phair(double tc, double pk) { return 44.6 * (pk / 101.3) * (273.15 / (tc + 273.15)); }   // cpp:280-285
inline double cpair(double tc) { return 2e-05 * pow(tc, 2.0) + 0.0002 * tc + 29.119; }                 // cpp:287-291

// ---- sun, cpp:28-102 -------------------------------------------------------------------------------
int julday(int year, int month, int day) {
    const double dd = day + 0.5;
    const int madj = month + (month < 3) * 12;
    const int yadj = year + (month < 3) * -1;
    const double j = trunc(365.25 * (yadj + 4716)) + trunc(30.6001 * (madj + 1)) + dd - 1524.5;
    const int b = (int)(2 - trunc((double)(yadj / 100)) + trunc(trunc((double)(yadj / 100)) / 4));
    return (int)(j + (j > 2299160) * b);
}
struct Sun { double zend, zenr, azid; };
Sun sun_position(double lat, double lon, int year, int month, int day, double lt) {
    const int jd = julday(year, month, day);
    const double m = 6.24004077 + 0.01720197 * (jd - 2451545.0);
    const double eot = -7.659 * sin(m) + 9.863 * sin(2 * m + 3.5932);
    const double st = lt + (4.0 * lon + eot) / 60.0;
    const double latr = lat * kPi / 180.0;
    const double tt = 0.261799 * (st - 12);
    const double dec = (kPi * 23.5 / 180) * cos(2 * kPi * ((jd - 159.5) / 365.25));
    const double coh = sin(dec) * sin(latr) + cos(dec) * cos(latr) * cos(tt);
    const double z = acos(coh) * (180 / kPi);
    const double sh = sin(dec) * sin(latr) + cos(dec) * cos(latr) * cos(tt);
    const double hh = atan(sh / sqrt(1 - sh * sh));
    const double sazi = cos(dec) * sin(tt) / cos(hh);
    const double cazi = (sin(latr) * cos(dec) * cos(tt) - cos(latr) * sin(dec)) /
                        sqrt(pow(cos(dec) * sin(tt), 2) + pow(sin(latr) * cos(dec) * cos(tt) - cos(latr) * sin(dec), 2));
    double sqt = 1 - sazi * sazi;
    if (sqt < 0) sqt = 0;
    double azi = 180 + (180 * atan(sazi / sqrt(sqt))) / kPi;
    if (cazi < 0) azi = sazi < 0 ? 180 - azi : 540 - azi;
    return {z, z * kToRad, azi};
}
double solar_index(double slope, double aspect, double zend, double azid) {   // shadowmask = false
    double si;
    if (zend > 90.0) si = 0;
    else if (slope == 0.0) si = cos(zend * kToRad);
    else si = cos(zend * kToRad) * cos(slope * kToRad) + sin(zend * kToRad) * sin(slope * kToRad) * cos((azid - aspect) * kToRad);
    return si < 0.0 ? 0.0 : si;
}
struct Ext { double k, kd, Kc; };
Ext canopy_k(double zenr, double x, double si) {                              // cankCpp, cpp:104-132
    if (zenr > kPi / 2.0) zenr = kPi / 2.0;
    if (si < 0.0) si = 0.0;
    double k;
    if (x == 1.0) k = 1.0 / (2.0 * cos(zenr));
    else if (isinf(x)) k = 1.0;
    else if (x == 0.0) k = tan(zenr);
    else k = sqrt(x * x + tan(zenr) * tan(zenr)) / (x + 1.774 * pow(x + 1.182, -0.733));
    if (k > 6000.0) k = 6000.0;
    Ext e{k, k * cos(zenr) / si, 1.0 / si};
    if (si == 0) { e.kd = 1.0; e.Kc = 600.0; }
    return e;
}

// ---- two-stream coefficients, cpp:134-185 ----------------------------------------------------------------
struct Dif { double p1, p2, p3, p4, om, a, gma, J, del, h, u1, S1, D1, D2; };
Dif two_stream_dif(double pait, double x, double lref, double ltra, double gref) {
    Dif p;
    p.om = lref + ltra; p.a = 1.0 - p.om; p.del = lref - ltra; p.J = 1.0 / 3.0;
    if (x != 1.0) {
        double mla = 9.65 * pow(3.0 + x, -1.65);
        if (mla > kPi / 2.0) mla = kPi / 2.0;
        p.J = cos(mla) * cos(mla);
    }
    p.gma = 0.5 * (p.om + p.J * p.del);
    p.h = sqrt(p.a * p.a + 2.0 * p.a * p.gma);
    p.S1 = exp(-p.h * pait);
    p.u1 = p.a + p.gma * (1.0 - 1.0 / gref);
    const double u2 = p.a + p.gma * (1.0 - gref);
    p.D1 = (p.a + p.gma + p.h) * (p.u1 - p.h) * 1.0 / p.S1 - (p.a + p.gma - p.h) * (p.u1 + p.h) * p.S1;
    p.D2 = (u2 + p.h) * 1.0 / p.S1 - (u2 - p.h) * p.S1;
    p.p1 = (p.gma / (p.D1 * p.S1)) * (p.u1 - p.h);
    p.p2 = (-p.gma * p.S1 / p.D1) * (p.u1 + p.h);
    p.p3 = (1.0 / (p.D2 * p.S1)) * (u2 + p.h);
    p.p4 = (-p.S1 / p.D2) * (u2 - p.h);
    return p;
}
struct Dir { double sig, p5, p6, p7, p8, p9, p10; };
Dir two_stream_dir(double pait, const Dif& f, double gref, double kd) {
    Dir p;
    const double sig = kd * kd + f.gma * f.gma - pow(f.a + f.gma, 2.0);
    const double ss = 0.5 * (f.om + f.J * f.del / kd) * kd;
    const double sstr = f.om * kd - ss;
    const double S2 = exp(-kd * pait);
    const double u2 = f.a + f.gma * (1.0 - gref);
    p.p5 = -ss * (f.a + f.gma - kd) - f.gma * sstr;
    const double v1 = ss - (p.p5 * (f.a + f.gma + kd)) / sig;
    const double v2 = ss - f.gma - (p.p5 / sig) * (f.u1 + kd);
    p.p6 = (1.0 / f.D1) * ((v1 / f.S1) * (f.u1 - f.h) - (f.a + f.gma - f.h) * S2 * v2);
    p.p7 = (-1.0 / f.D1) * ((v1 * f.S1) * (f.u1 + f.h) - (f.a + f.gma + f.h) * S2 * v2);
    p.sig = -sig;
    p.p8 = sstr * (f.a + f.gma + kd) - f.gma * ss;
    const double v3 = (sstr + f.gma * gref - (p.p8 / p.sig) * (u2 - kd)) * S2;
    p.p9 = (-1 / f.D2) * ((p.p8 / (p.sig * f.S1)) * (u2 + f.h) + v3);
    p.p10 = (1 / f.D2) * (((p.p8 * f.S1) / p.sig) * (u2 - f.h) + v3);
    return p;
}

// ---- aerodynamics, cpp:294-380 ---------------------------------------------------------------------------
double zeroplane(double h, double pai) {
    if (pai < 0.001) pai = 0.001;
    return (1.0 - (1.0 - exp(-sqrt(7.5 * pai))) / sqrt(7.5 * pai)) * h;
}
double roughlength(double h, double pai, double d, double psi_h) {
    const double Be = sqrt(0.003 + (0.2 * pai) / 2);
    double zm = (h - d) * exp(-kKa / Be) * exp(kKa * psi_h);
    if (zm > 0.9 * (h - d)) zm = 0.9 * (h - d);
    if (zm < 0.0005) zm = 0.0005;
    return zm;
}
double psi_m(double ze) {
    double v;
    if (ze < 0) {
        const double x = pow(1.0 - 15.0 * ze, 0.25);
        v = log(pow((1.0 + x) / 2.0, 2.0) * (1 + pow(x, 2.0)) / 2.0) - 2.0 * atan(x) + kPi / 2.0;
    } else v = -4.7 * ze;
    if (v < -4.0) v = -4.0;
    if (v > 3.0) v = 3.0;
    return v;
}
double psi_h(double ze) {
    double v;
    if (ze < 0) {
        const double y = sqrt(1.0 - 9.0 * ze);
        v = log(pow((1.0 + y) / 2.0, 2.0));
    } else v = -(4.7 * ze) / 0.74;
    if (v < -4.0) v = -4.0;
    if (v > 3.0) v = 3.0;
    return v;
}
double phi_h(double ze) {
    double v;
    if (ze < 0) {
        const double phim = 1 / pow(1.0 - 16.0 * ze, 0.25);
        v = pow(phim, 2.0);
    } else v = 1 + ((6.0 * ze) / (1.0 + ze));
    if (v > 1.5) v = 1.5;
    if (v < 0.5) v = 0.5;
    return v;
}
double g_free(double leafd, double H) {
    const double d = 0.71 * leafd;
    const double dT = 0.7045388 * pow(d * pow(H, 4.0), 0.2);
    double g = 0.0375 * pow(dT / d, 0.25);
    if (g < 0.1) g = 0.1;
    return g;
}
double g_turb(double uf, double d, double zm, double zref, double ph, double psih, double gmin) {
    const double z0 = 0.2 * zm + d;
    double g = (kKa * ph * uf) / (log((zref - d) / (z0 - d)) + psih);
    if (g < gmin) g = gmin;
    return g;
}

// ---- stomata, cpp:382-477 ----------------------------------------------------------------------------------
struct Stomp { double Rsmx, psiw0, kk, rat; };
Stomp stom_params(double hgt, double lat, double x) {
    Stomp o{420.0, -3.1, 0.34, 0.9};
    if (hgt < 1.0 && fabs(lat) < 22.5) o = {450.0, -2.7, 0.39, 0.9};
    if (hgt >= 1.0 && hgt < 7.0) o = {430.0, -4.0, 0.28, 0.75};
    if (hgt >= 7.0) {
        if (fabs(lat) < 22.5) o = {500.0, -1.75, 0.67, 0.4};
        else if (x < 0.8 || fabs(lat) > 58.0) o = {420.0, -4.09, 0.29, 0.6};
        else o = {500.0, -2.51, 0.46, 0.45};
    }
    return o;
}
double stom_cond(double Rswabs, double theta, double gsmax, double Smax, double psi_e, double b, const Stomp& st) {
    if (Rswabs <= 0.0) return 0.0;
    if (Rswabs > st.Rsmx) Rswabs = st.Rsmx;
    double gs = gsmax * pow(2.0, -(st.Rsmx - Rswabs) / (0.2 * st.Rsmx));
    const double thetan = st.rat * theta + (1 - st.rat) * kThetam;
    double Se = thetan / Smax;
    if (Se > 1.0) Se = 1.0;
    double psiw = -fabs(psi_e) * pow(Se, -b) * 0.01;
    if (psiw < st.psiw0) psiw = st.psiw0;
    const double mu = 1.0 - (exp(-st.kk * psiw) - 1.0) / (exp(-st.kk * st.psiw0) - 1.0);
    const double gs2 = mu * gsmax;
    if (gs > gs2) gs = gs2;
    return gs;
}
double canopy_cond(double Rsw, double Rdif, double k, double om, double theta, double gsmax, double PAI, double Smax,
                   double psi_e, double b, const Stomp& st) {
    if (isnan(om)) return 9999.99;
    const double P_sun = (1.0 - exp(-k * PAI)) / k;
    const double P_shade = PAI - P_sun;
    const double Rshade = Rdif * ((1.0 - exp(-PAI)) / PAI) * (1.0 - om);
    const double Rsun = (Rsw - Rdif) * k * (1 - om) + Rshade;
    return stom_cond(Rsun, theta, gsmax, Smax, psi_e, b, st) * P_sun + stom_cond(Rshade, theta, gsmax, Smax, psi_e, b, st) * P_shade;
}
// PenmanMonteithCpp, cpp:498-514
double penman(double Rabs, double gHa, double gV, double tc, double te, double pk, double ea, double em, double G, double erh) {
    const double Rema = em * kSb * radem(tc);
    const double la = te >= 0 ? 45068.7 - 42.8428 * te : 51078.69 - 4.338 * te - 0.06367 * te * te;
    const double cp = cpair(te);
    const double Da = satvap(tc) - ea;
    const double gR = (4.0 * em * kSb * pow(te + 273.15, 3.0)) / cp;
    const double De = satvap(te + 0.5) - satvap(te - 0.5);
    return tc + ((Rabs - Rema - la * (gV / pk) * Da * erh - G) / (cp * (gHa + gR) + la * (gV / pk) * De * erh));
}

// ---- series helpers, cpp:517-594 -----------------------------------------------------------------------------
enum Stat { MAX, MIN, MEAN };
void hour_to_day(const Vec& h, Stat stat, Vec& out) {   // rephour = true
    const size_t nd = h.size() / 24;
    for (size_t i = 0; i < nd; ++i) {
        double s = h[i * 24];
        if (stat == MAX) for (int j = 1; j < 24; ++j) s = fmax(s, h[i * 24 + j]);
        else if (stat == MIN) for (int j = 1; j < 24; ++j) s = fmin(s, h[i * 24 + j]);
        else { s = 0.0; for (int j = 0; j < 24; ++j) s += h[i * 24 + j]; s /= 24; }
        for (int j = 0; j < 24; ++j) out[i * 24 + j] = s;
    }
}
// maCpp: circular trailing mean.  The reference indexes x[(i - j + m) % m], which leaves the array for windows
// longer than the series (i - j + m < 0); callers here reject those cases up front (see mcf_bigleaf).
void moving_mean(const Vec& x, int n, Vec& y) {
    const int m = (int)x.size();
    for (int i = 0; i < m; ++i) {
        double sum = 0.0;
        for (int j = 0; j < n; ++j) sum += x[(size_t)((i - j + m) % m)];
        y[(size_t)i] = sum / n;
    }
}
void yearly_mean(const Vec& x, Vec& z) {                 // mayCpp: daily means, 91-day circular mean, back to hours
    const size_t nd = x.size() / 24;
    Vec d(nd), y(nd);
    for (size_t i = 0; i < nd; ++i) {
        double s = 0.0;
        for (int j = 0; j < 24; ++j) s += x[i * 24 + j];
        d[i] = s / 24.0;
    }
    moving_mean(d, 91, y);
    for (size_t i = 0; i < nd; ++i)
        for (int j = 0; j < 24; ++j) z[i * 24 + j] = y[i];
}

// ---- RadswabsCpp, cpp:187-278 ----------------------------------------------------------------------------------
struct PointIn {
    int64_t n;
    const int32_t *year, *month, *day;
    const double* hour;
};
void shortwave_absorbed(const PointIn& t, double pai, double x, double lref, double ltra, double clump, double gref,
                        double slope, double aspect, double lat, double lon, const double* Rsw, const double* Rdif,
                        Vec& radG, Vec& radC, double* albedo) {
    const size_t n = (size_t)t.n;
    if (pai > 0.0) {
        double pait = pai;
        if (clump > 0.0) pait = pai / (1 - clump);
        const Dif p = two_stream_dif(pait, x, lref, ltra, gref);
        const double trd = clump * clump;
        double amx = gref;
        if (amx < lref) amx = lref;
        double albd = gref * (trd * trd) + (1.0 - trd * trd) * (p.p1 + p.p2);
        if (albd > amx) albd = amx;
        if (albd < 0.01) albd = 0.01;
        const double groundRdd = trd + (1.0 - trd) * (p.p3 * exp(-p.h * pait) + p.p4 * exp(p.h * pait));
        for (size_t i = 0; i < n; ++i) {
            if (Rsw[i] > 0.0) {
                Sun sp = sun_position(lat, lon, t.year[i], t.month[i], t.day[i], t.hour[i]);
                const double si = solar_index(slope, aspect, sp.zend, sp.azid);
                if (sp.zenr > kPi / 2.0) sp.zenr = kPi / 2.0;
                const double cosz = cos(sp.zenr);
                const Ext kp = canopy_k(sp.zenr, x, si);
                const Dir d = two_stream_dir(pait, p, gref, kp.kd);
                double Rbeam = (Rsw[i] - Rdif[i]) / cosz;
                if (Rbeam > 1352.0) Rbeam = 1352.0;
                double trb = pow(clump, kp.Kc);
                if (trb > 0.999) trb = 0.999;
                if (trb < 0.0) trb = 0.0;
                const double Rb = Rbeam * cosz;
                const double trg = trb + (1 - trb) * exp(-kp.kd * pait);
                const double Rbc = (trg * si + (1 - trg) * cosz) * Rbeam;
                double albb = trd * trb * gref + (1.0 - trd * trb) * (d.p5 / -d.sig + d.p6 + d.p7);
                if (albb > amx) albb = amx;
                if (albb < 0.01) albb = 0.01;
                double groundRbdd = trb + (1.0 - trb) * ((d.p8 / d.sig) * exp(-kp.kd * pait) + d.p9 * exp(-p.h * pait) +
                                                         d.p10 * exp(p.h * pait));
                if (groundRbdd > amx) groundRbdd = amx;
                if (groundRbdd < 0.0) groundRbdd = 0.0;
                radC[i] = (1.0 - albd) * Rdif[i] + (1.0 - albb) * Rbc;
                const double Rgdif = groundRdd * Rdif[i] + groundRbdd * Rb;
                radG[i] = (1.0 - gref) * (Rgdif + exp(-kp.kd * pait) * Rbeam * si);
                albedo[i] = 1.0 - (radC[i] / (Rdif[i] + Rb));
                if (albedo[i] > amx) albedo[i] = amx;
                if (albedo[i] < 0.01) albedo[i] = 0.01;
            } else {
                radG[i] = 0; radC[i] = 0; albedo[i] = lref;
            }
        }
    } else {
        for (size_t i = 0; i < n; ++i) {
            albedo[i] = gref;
            if (Rsw[i] > 0) {
                Sun sp = sun_position(lat, lon, t.year[i], t.month[i], t.day[i], t.hour[i]);
                const double si = solar_index(slope, aspect, sp.zend, sp.azid);
                if (sp.zenr > kPi / 2.0) sp.zenr = kPi / 2.0;
                const double dirr = (Rsw[i] - Rdif[i]) / cos(sp.zenr);
                radG[i] = (1 - gref) * (Rdif[i] + si * dirr);
                radC[i] = radG[i];
            } else {
                radG[i] = 0; radC[i] = 0;
            }
        }
    }
}

// ---- GFluxCpp, cpp:641-707 ---------------------------------------------------------------------------------------
void ground_flux(const double* Tg, const double* soilm, size_t n, double rho, double Vm, double Vq, double Mc, Vec& Gmax,
                 Vec& Gmin, int iter, bool yearG, Vec& G) {
    const double frs = Vm + Vq;
    const double c1 = (0.57 + 1.73 * Vq + 0.93 * Vm) / (1.0 - 0.74 * Vq - 0.49 * Vm) - 2.8 * frs * (1.0 - frs);
    const double c3 = 1.0 + 2.6 * pow(Mc, -0.5);
    const double c4 = 0.03 + 0.7 * frs * frs;
    const double mu1 = 2400.0 * rho / 2.64, mu2 = 1.06 * rho;
    Vec Tgv(Tg, Tg + n), Td(n), Gmu(n), dT(n), k(n), kap(n), Gmud(n);
    hour_to_day(Tgv, MEAN, Td);
    for (size_t i = 0; i < n; ++i) {
        const double cs = mu1 + 4180 * soilm[i];
        const double ph = (rho * (1.0 - soilm[i]) + soilm[i]) * 1000;
        const double c2 = mu2 * soilm[i];
        k[i] = c1 + c2 * soilm[i] - (c1 - c4) * exp(-pow(c3 * soilm[i], 4.0));
        kap[i] = k[i] / (cs * ph);
        const double DD = sqrt(2 * kap[i] / kOmdy);
        Gmu[i] = sqrt(2) * (k[i] / DD) * 0.5;
        dT[i] = Tg[i] - Td[i];
    }
    moving_mean(Gmu, 6, Gmud);
    moving_mean(dT, 6, G);
    for (size_t i = 0; i < n; ++i) G[i] = G[i] * Gmud[i] * 1.1171;
    if (iter == 0) {
        hour_to_day(G, MIN, Gmin);
        hour_to_day(G, MAX, Gmax);
    }
    for (size_t i = 0; i < n; ++i) {
        if (G[i] < Gmin[i]) G[i] = Gmin[i];
        if (G[i] > Gmax[i]) G[i] = Gmax[i];
    }
    if (yearG) {
        Vec kma(n), kama(n), dTy(n), madTy(n);
        yearly_mean(k, kma);
        yearly_mean(kap, kama);
        double sumTd = 0.0;
        for (size_t i = 0; i < n; ++i) sumTd += Td[i];
        for (size_t i = 0; i < n; ++i) dTy[i] = Td[i] - sumTd / n;
        yearly_mean(dTy, madTy);
        for (size_t i = 0; i < n; ++i) {
            const double omyr = (2 * kPi) / (n * 3600.0);
            const double Gmuy = sqrt(2) * kma[i] / sqrt(2 * kama[i] / omyr);
            G[i] = G[i] + madTy[i] * Gmuy * 1.1171;
        }
    }
}

int check_series(const mcf_obstime* t, const mcf_point_weather* w, int64_t n, bool need_precip) {
    if (n <= 0 || n > (1 << 28)) return mcf::api_fail(MCF_ERR_ARG, "point model: bad series length");
    if (!t || !t->year || !t->month || !t->day || !t->hour) return mcf::api_fail(MCF_ERR_ARG, "point model: null obstime");
    if (!w || !w->temp || !w->relhum || !w->pres || !w->swdown || !w->difrad || !w->lwdown || !w->windspeed)
        return mcf::api_fail(MCF_ERR_ARG, "point model: null weather column");
    if (need_precip && !w->precip) return mcf::api_fail(MCF_ERR_ARG, "point model: null precip");
    return MCF_OK;
}

}  // namespace

// BigLeafCpp, cpp:710-881
extern "C" int mcf_bigleaf(int64_t n64, const mcf_obstime* t, const mcf_point_weather* w, const double* vegp,
                           const double* groundp, const double* soilm, double lat, double lon, double dTmx, double zref,
                           int32_t maxiter, double bwgt, double tol, int32_t yearG, mcf_bigleaf_out* o) {
    int rc = check_series(t, w, n64, false);
    if (rc) return rc;
    if (!vegp || !groundp || !soilm || !o || !o->Tc || !o->Tg || !o->H || !o->G || !o->psih || !o->psim || !o->phih ||
        !o->OL || !o->uf || !o->RabsG || !o->albedo)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_bigleaf: null argument");
    const size_t n = (size_t)n64;
    if (n < 6) return mcf::api_fail(MCF_ERR_ARG, "mcf_bigleaf: the 6-hour running mean of GFluxCpp needs at least 6 steps");
    if (yearG && n / 24 > 1 && n / 24 < 90)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_bigleaf: yearG needs one day or at least 90 (the reference's 91-day circular "
                                          "mean reads outside its array for series in between)");
    const double h = vegp[0], pai = vegp[1], vegx = vegp[2], clump = vegp[3], lref = vegp[4], ltra = vegp[5],
                 leafd = vegp[6], em = vegp[7], gsmax = vegp[8];
    const double gref = groundp[0], slope = groundp[1], aspect = groundp[2], groundem = groundp[3], rho = groundp[4],
                 Vm = groundp[5], Vq = groundp[6], Mc = groundp[7], soilb = groundp[8], psie = groundp[9],
                 Smax = groundp[10], Smin = groundp[11];
    const double *tc = w->temp, *rh = w->relhum, *pk = w->pres, *Rsw = w->swdown, *Rdif = w->difrad, *Rlw = w->lwdown,
                 *wspeed = w->windspeed;
    const PointIn ti{n64, t->year, t->month, t->day, t->hour};
    Vec swG(n), swC(n);
    shortwave_absorbed(ti, pai, vegx, lref, ltra, clump, gref, slope, aspect, lat, lon, Rsw, Rdif, swG, swC, o->albedo);
    const double pait = pai / (1 - clump);
    const double trd = (1 - clump * clump) * exp(-pait) + clump * clump;
    const double d = zeroplane(h, pai);
    const double Belim = 0.4 / sqrt(0.003 + (0.2 * pai) / 2);
    Vec tcc(tc, tc + n), tcg(tc, tc + n), Gmin(n, -999.0), Gmax(n, 999.0), Gnew(n);
    for (size_t i = 0; i < n; ++i) {
        o->Tg[i] = tc[i]; o->Tc[i] = tc[i];
        o->psim[i] = 0; o->psih[i] = 0; o->phih[i] = 0; o->OL[i] = 0; o->G[i] = 0;
        o->uf[i] = 999.0; o->RabsG[i] = 999.0;
        o->H[i] = 0.5 * Rsw[i] - em * kSb * radem(tc[i]);
    }
    const Stomp st = stom_params(h, lat, vegx);
    const double om = 0.5 * (lref + ltra);
    double tstf = tol * 2, tst = 0;
    int iter = 0;
    while (tstf > tol) {
        tst = 0;
        for (size_t i = 0; i < n; ++i) {
            const double RemC = em * kSb * radem(o->Tc[i]);
            const double radClw = em * Rlw[i];
            const double radGlw = groundem * (trd * radClw + (1 - trd) * RemC);
            o->RabsG[i] = swG[i] + radGlw;
            const double RabsC = swC[i] + radClw;
            const double zm = roughlength(h, pai, d, o->psih[i]);
            o->uf[i] = (kKa * wspeed[i]) / (log((zref - d) / zm) + o->psim[i]);
            if (o->uf[i] < 0.0002) o->uf[i] = 0.0002;
            const double gmin = g_free(leafd, fabs(o->H[i])) * 2 * pai;
            double ph = phair(tcc[i], pk[i]);
            const double gHa = g_turb(o->uf[i], d, zm, zref, ph, o->psih[i], gmin);
            const Sun sp = sun_position(lat, lon, t->year[i], t->month[i], t->day[i], t->hour[i]);
            const Ext kp = canopy_k(sp.zenr, vegx, cos(sp.zenr));
            const double gC = canopy_cond(Rsw[i], Rdif[i], kp.k, om, soilm[i], gsmax, pai, Smax, psie, soilb, st);
            double gV = 1 / (1 / gHa + 1 / gC);
            if (gC == 0) gV = 0;
            const double ea = satvap(tc[i]) * rh[i] / 100;
            double Tcn = penman(RabsC, gHa, gV, tc[i], tcc[i], pk[i], ea, em, o->G[i], 1);
            const double tdew = dewpoint(ea);
            if (Tcn < tdew) Tcn = tdew;
            const double srh = (soilm[i] - Smin) / (Smax - Smin);
            double Tgn = penman(o->RabsG[i], gHa, gHa, tcg[i], tcc[i], pk[i], ea, em, o->G[i], srh);
            if (Tgn < tdew) Tgn = tdew;
            double dTc = Tcn - tc[i], dTg = Tgn - tc[i];
            if (dTc > dTmx) dTc = dTmx;
            if (dTg > dTmx) dTg = dTmx;
            Tcn = tc[i] + dTc;
            Tgn = tc[i] + dTg;
            const double tst2 = fabs(Tcn - o->Tc[i]), tst3 = fabs(Tgn - o->Tg[i]);
            if (tst2 > tst) tst = tst2;
            if (tst3 > tst) tst = tst3;
            o->Tc[i] = bwgt * o->Tc[i] + (1 - bwgt) * Tcn;
            o->Tg[i] = bwgt * o->Tg[i] + (1 - bwgt) * Tgn;
            tcc[i] = (o->Tc[i] + tc[i]) / 2;
            tcg[i] = (o->Tg[i] + tc[i]) / 2;
            const double Tk = 273.15 + tcc[i];
            ph = phair(tcc[i], pk[i]);
            const double cp = cpair(tcc[i]);
            o->H[i] = bwgt * o->H[i] + (1 - bwgt) * (gHa * cp * (Tcn - tc[i]));
            const double Rnet = RabsC - kSb * em * radem(o->Tc[i]);
            if (Rnet > 0 && o->H[i] > Rnet) o->H[i] = Rnet;
            if (fabs(o->H[i]) < 0.1) o->H[i] = 0.1;
            o->OL[i] = (ph * cp * pow(o->uf[i], 3.0) * Tk) / (-0.4 * 9.81 * o->H[i]);
            o->psim[i] = psi_m(zm / o->OL[i]) - psi_m((zref - d) / o->OL[i]);
            o->psih[i] = psi_h((0.2 * zm) / o->OL[i]) - psi_h((zref - d) / o->OL[i]);
            o->phih[i] = phi_h((zref - d) / o->OL[i]);
            const double ln1 = log((zref - d) / zm), ln2 = log((zref - d) / (0.2 * zm));
            if (o->psim[i] < -0.9 * ln1) o->psim[i] = -0.9 * ln1;
            if (o->psih[i] < -0.9 * ln2) o->psih[i] = -0.9 * ln2;
            if (o->psim[i] > 0.9 * ln1) o->psim[i] = 0.9 * ln1;
            if (o->psih[i] > 0.9 * ln2) o->psih[i] = 0.9 * ln2;
            if (o->psih[i] > 0.9 * Belim) o->psih[i] = 0.9 * Belim;
        }
        ground_flux(o->Tg, soilm, n, rho, Vm, Vq, Mc, Gmax, Gmin, iter, yearG != 0, Gnew);
        memcpy(o->G, Gnew.data(), n * sizeof(double));
        tstf = tst;
        ++iter;
        if (iter >= maxiter) tstf = 0;
    }
    o->err = tst;
    o->iters = iter;
    return MCF_OK;
}

// soilmCpp, cpp:931-972: two-layer daily bucket model
extern "C" int mcf_soilm(int64_t n64, const mcf_point_weather* w, double rmu, double mult, double pwr, double Smax,
                         double Smin, double Ksat, double a, double* soilm_days, int64_t* ndays) {
    if (!w || !w->temp || !w->swdown || !w->lwdown || !w->precip || !soilm_days || !ndays || n64 <= 0)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_soilm: null argument");
    const int64_t nd = n64 / 24;
    Vec rnetd((size_t)nd), rain((size_t)nd);
    for (int64_t dday = 0; dday < nd; ++dday) {
        double sr = 0.0, sp = 0.0;
        for (int hh = 0; hh < 24; ++hh) {
            const int64_t i = dday * 24 + hh;
            const double swrad = (1 - 0.15) * w->swdown[i];
            const double lwout = kSb * 0.95 * radem(w->temp[i]);
            double rnet = swrad - (lwout - w->lwdown[i]);
            if (rnet < 0) rnet = 0;
            sr += rnet;
            sp += w->precip[i];
        }
        rnetd[(size_t)dday] = sr / 24;
        rain[(size_t)dday] = sp;
    }
    double s1 = Smax, s2 = Smax;
    if (nd > 0) soilm_days[0] = Smax;
    for (int64_t i = 1; i < nd; ++i) {
        const double sav = (s1 + s2) / 2;
        const double dif = s2 - s1;
        s1 = s1 + rmu * rain[(size_t)i] - mult * rnetd[(size_t)i];
        const double k = Ksat * pow(sav / Smax, pwr);
        s1 = s1 + a * k * dif;
        s2 = s2 - ((a * k * dif) / 10);
        if (s1 > Smax) s1 = Smax;
        if (s2 > Smax) s2 = Smax;
        if (s1 < Smin) s1 = Smin;
        if (s2 < Smin) s2 = Smin;
        soilm_days[i] = (s1 + s2) / 2;
    }
    *ndays = nd;
    return MCF_OK;
}

// pointmprocess, cpp:5265-5323
extern "C" int mcf_pointmprocess(int64_t n64, const double* u2, const double* tc, const double* rh, const double* pk,
                                 const double* uf, const double* soilm, const double* RabsG, double zref, double h,
                                 double pai, double rho, double Vm, double Vq, double Mc, double* umu, double* kp,
                                 double* muGp, double* DDp, double* T0p, double* dtrp) {
    if (n64 <= 0 || !u2 || !tc || !rh || !pk || !uf || !soilm || !RabsG || !umu || !kp || !muGp || !DDp || !T0p || !dtrp)
        return mcf::api_fail(MCF_ERR_ARG, "mcf_pointmprocess: null argument");
    const size_t n = (size_t)n64;
    const double dp = zeroplane(h, pai);
    const double zmp = roughlength(h, pai, dp, 0);
    const double frs = Vm + Vq;                                                          // soilpfun, cpp:628-636
    const double c1 = (0.57 + 1.73 * Vq + 0.93 * Vm) / (1.0 - 0.74 * Vq - 0.49 * Vm) - 2.8 * frs * (1.0 - frs);
    const double c3 = 1.0 + 2.6 * pow(Mc, -0.5);
    const double c4 = 0.03 + 0.7 * frs * frs;
    for (size_t i = 0; i < n; ++i) {
        const double ufps = (kKa * u2[i]) / log((zref - dp) / zmp);
        umu[i] = uf[i] / ufps;
        const double cs = (2400 * rho / 2.64 + 4180 * soilm[i]);
        const double ph = (rho * (1.0 - soilm[i]) + soilm[i]) * 1000;
        const double c2 = 1.06 * rho * soilm[i];
        kp[i] = c1 + c2 * soilm[i] - (c1 - c4) * exp(-pow(c3 * soilm[i], 4.0));
        const double kap = kp[i] / (cs * ph);
        muGp[i] = pow(2.0 * kap / kOmdy, 0.5);
        DDp[i] = pow(2.0 * kap / kOmdy, 0.5);
        const double gHa = (0.4 * 43.0 * ufps) / log((zref - dp) / zmp);
        const double es = satvap(tc[i]);
        const double ea = es * rh[i] / 100.0;
        T0p[i] = penman(RabsG[i], gHa, gHa, tc[i], tc[i], pk[i], ea, 0.97, 0.0, 1.0);
        dtrp[i] = 0.0;
    }
    for (size_t dday = 0; dday < n / 24; ++dday) {
        double mx = T0p[dday * 24], mn = T0p[dday * 24];
        for (int j = 1; j < 24; ++j) {
            mx = fmax(mx, T0p[dday * 24 + j]);
            mn = fmin(mn, T0p[dday * 24 + j]);
        }
        for (int j = 0; j < 24; ++j) dtrp[dday * 24 + j] = mx - mn;
    }
    return MCF_OK;
}

// weatherhgtCpp, cpp:884-929
extern "C" int mcf_weatherhgt(int64_t n64, const mcf_obstime* t, const mcf_point_weather* w, double zin, double uzin,
                              double zout, double lat, double lon, double* temp, double* relhum, double* windspeed) {
    int rc = check_series(t, w, n64, false);
    if (rc) return rc;
    if (!temp || !relhum || !windspeed) return mcf::api_fail(MCF_ERR_ARG, "mcf_weatherhgt: null output");
    const size_t n = (size_t)n64;
    const double vegp[10] = {0.12, 1, 1, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100.0};
    const double groundp[12] = {0.15, 0.0, 180.0, 0.97, 1.529643, 0.509, 0.06, 0.5422, 5.2, 2.6, 0.419, 0.074};
    Vec soilm(n, 0.2), buf(11 * n);
    mcf_bigleaf_out bo;
    double** slots[11] = {&bo.Tc, &bo.Tg, &bo.H, &bo.G, &bo.psih, &bo.psim, &bo.phih, &bo.OL, &bo.uf, &bo.RabsG, &bo.albedo};
    for (int q = 0; q < 11; ++q) *slots[q] = buf.data() + (size_t)q * n;
    // the reference passes yearG = true (cpp:895); for 2..89 days that reads outside its arrays, and the annual
    // term is switched off here instead (it is exactly zero for a single day)
    const int yearG = (n / 24 <= 1 || n / 24 >= 90) ? 1 : 0;
    if ((rc = mcf_bigleaf(n64, t, w, vegp, groundp, soilm.data(), lat, lon, 25, 2, 20, 0.5, 0.5, yearG, &bo))) return rc;
    const double d = zeroplane(0.12, 1);
    for (size_t i = 0; i < n; ++i) {
        const double zm = roughlength(0.12, 1, d, bo.psih[i]);
        const double zh = 0.2 * zm;
        const double lnr = log((zout - d) / zh) / log((zin - d) / zh);
        temp[i] = (bo.Tc[i] - w->temp[i]) * (1 - lnr) + w->temp[i];
        const double ea = satvap(w->temp[i]) * w->relhum[i] / 100;
        double es = satvap(bo.Tc[i]) * sqrt(w->relhum[i] / 100);
        const double ez = ea + (es - ea) * (1 - lnr);
        es = satvap(temp[i]);
        relhum[i] = (ez / es) * 100;
        if (relhum[i] < 0.25 * w->relhum[i]) relhum[i] = 0.25 * w->relhum[i];
        if (relhum[i] > 100.0) relhum[i] = 100.0;
        const double lnru = log((zout - d) / zm) / log((uzin - d) / zm);
        windspeed[i] = w->windspeed[i] * lnru;
    }
    return MCF_OK;
}

// manCpp, cpp:597-627: circular trailing mean over `window` steps; windows beyond 48 h go through daily means
// (window / 24 days) and a 24-h mean of the result.  Used by `.soilbelowT` (R/internal.R:169-185).
extern "C" int mcf_man(int64_t n64, const double* x, int32_t window, double* out) {
    if (n64 <= 0 || n64 > INT32_MAX || !x || !out || window < 1) return mcf::api_fail(MCF_ERR_ARG, "mcf_man: bad length, window or null argument");
    const int m = (int)n64;
    Vec xv(x, x + m), z((size_t)m);
    // the reference's circular index (i - j + m) % m leaves the array when the window is longer than the series
    // (cpp:561-572); such calls are refused here
    if (window <= 48) {
        if (window > m) return mcf::api_fail(MCF_ERR_ARG, "mcf_man: the window is longer than the series");
        moving_mean(xv, window, z);
    } else {
        const size_t nd = (size_t)m / 24;
        if (nd == 0 || (size_t)(window / 24) > nd || m < 24)
            return mcf::api_fail(MCF_ERR_ARG, "mcf_man: the window is longer than the series");
        Vec d(nd), y(nd), zz((size_t)m, 0.0);
        for (size_t i = 0; i < nd; ++i) {
            double sum = 0.0;
            for (int j = 0; j < 24; ++j) sum += xv[i * 24 + j];
            d[i] = sum / 24.0;
        }
        moving_mean(d, window / 24, y);
        for (size_t i = 0; i < nd; ++i)
            for (int j = 0; j < 24; ++j) zz[i * 24 + j] = y[i];
        moving_mean(zz, 24, z);
    }
    for (int i = 0; i < m; ++i) out[i] = z[(size_t)i];
    return MCF_OK;
}

// =====================================================================================================================
// pointmodelsnow, cpp:4000-4169 (with canopysnowintCpp 3713-3739, snowdenp 3741-3749, snowalbCpp 3752-3771, radoneB
// 3773-3833, snowoneB 3835-3972, GFluxCppsnow 3974-3997): the snow branch's point model — one snowpack under a canopy,
// stepped through the series and relaxed until canopy and ground snow temperatures settle.  It feeds `.snowmodel1`
// (R/internal.R:2536-2541) with Gp, Tc, RswabsG, RlwabsG, umu and the pack depth that `.sortl` averages vegetation by.
// Host code, as in the reference: a serial recurrence in time inside an outer relaxation.
// =====================================================================================================================
namespace {

struct SnowDen { double a, b, c, d; };
SnowDen snow_density_params(int env) {                       // snowdenp
    static const SnowDen tab[5] = {{0.5975, 0.2237, 0.0012, 0.0038}, {0.5979, 0.2578, 0.001, 0.0038},
                                   {0.594, 0.2332, 0.0016, 0.0031}, {0.363, 0.2425, 0.0029, 0.0049}, {0.217, 0.217, 0.0, 0.0}};
    return tab[(env < 0 || env > 4) ? 0 : env];
}
double snow_density(const SnowDen& p, double depth, double age_hours) {
    return ((p.a - p.b) * (1.0 - exp(-p.c * depth / 100.0 - p.d * age_hours / 24.0)) + p.b) * 1000.0;
}
// the age clock counts HOURS since snowfall; `hs / 24` is an integer division in the reference (whole days)
void snow_albedo(const double* prec, size_t n, Vec& alb) {
    int hs = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i > 0) hs = prec[i] > 0 ? 0 : hs + 1;
        double a = (-9.8740 * log((double)(hs / 24)) + 78.3434) / 100.0;
        alb[i] = a > 0.95 ? 0.95 : a < 0.1 ? 0.1 : a;
    }
}
double canopy_snow_interception(double hgt, double pai, double uf, double prec, double tc, double Li) {
    if (hgt < 0.001) hgt = 0.001;
    if (pai < 0.001) pai = 0.001;
    const double Be = sqrt(0.003 + (0.2 * pai) / 2.0), uh = uf / Be;
    const double Lc = pow(0.25 * (pai / hgt), -1.0), Lm = 2.0 * pow(Be, 3.0) * Lc, k1 = Be / Lm;
    double uzm = (uh / (hgt * k1)) * (1 - exp(-k1 * hgt));
    if (uzm < uf) uzm = uf;
    const double rhos = 67.92 + 51.25 * exp(tc / 2.59);
    const double Lstr = 6.2 * (0.26 + 46 / rhos) * pai;
    const double kc = 1.0 / (2.0 * cos(atan(uzm / 0.8)));
    const double Cp = 1.0 - exp(-kc * pai);
    const double cis = (Lstr - Li) * (1.0 - exp(-(Cp / Lstr) * prec)) * 0.678;
    return cis > prec ? prec : cis;
}

struct SnowStepIn {
    int year, month, day;
    double hour, tc, ea, pk, u2, Rsw, Rdif, Rlw, prec, Tci, te;
    double pai, hgt, ltra, clump;                              // vegetation of the site
    double alb, sdenc, sdeng, sdepc, sdepg, agec, ageg;        // pack
    double slope, aspect, lat, lon, zref, psim, psih, G;
};
struct SnowStepOut {
    double Tc, Tg, mSc, mMc, mRc, uf, gHa, RswabsG, RlwabsG, tr, Tcp, agec, ageg, sdenc, sdeng, sdepc, sdepg, pai, hgt;
};
struct SnowRad { double RabsC, RswabsG, RlwabsG, tr; };

SnowRad snow_radiation(const SnowStepIn& q, double pai, double hgt, double ltra) {      // radoneB
    SnowRad o{};
    const double RlwabsC = 0.97 * q.Rlw, cld = q.clump * q.clump, pait = pai / (1.0 - q.clump);
    o.RlwabsG = RlwabsC;
    o.tr = (1.0 - cld) * exp(-pait) + cld;
    if (hgt > 0.0) o.RlwabsG = 0.97 * (o.tr * q.Rlw + (1.0 - o.tr) * 0.97 * kSb * radem(q.Tci));
    o.RabsC = RlwabsC;
    if (q.Rsw > 0.0) {
        const Sun sp = sun_position(q.lat, q.lon, q.year, q.month, q.day, q.hour);
        double si = solar_index(q.slope, q.aspect, sp.zend, sp.azid);
        if (si < 0.0) si = 0.0;
        const double cosz = cos(sp.zenr);
        double Rbeam = (q.Rsw - q.Rdif) / cosz;
        if (Rbeam > 1352.2) Rbeam = 1352.2;
        const double RswabsC = (1.0 - q.alb) * (q.Rdif + Rbeam * cosz);
        o.RabsC = RswabsC + RlwabsC;
        o.RswabsG = RswabsC;
        if (hgt > 0.0) {
            if (q.alb + ltra > 0.999) ltra = 0.999 - q.alb;
            const Dif f = two_stream_dif(pait, 1.0, q.alb, ltra, q.alb);
            const Ext kp = canopy_k(sp.zenr, 1.0, si);
            const Dir r = two_stream_dir(pait, f, q.alb, kp.kd);
            const double clb = pow(q.clump, kp.Kc);
            double Rddm = (1.0 - cld) * (f.p3 * exp(-f.h * pait) + f.p4 * exp(f.h * pait)) + cld;
            Rddm = Rddm > 1.0 ? 1.0 : Rddm < 0.0 ? 0.0 : Rddm;
            double Rdbm = (1.0 - clb) * ((r.p8 / r.sig) * exp(-kp.kd * pait) + r.p9 * exp(-f.h * pait) + r.p10 * exp(f.h * pait));
            Rdbm = Rdbm > 1.0 ? 1.0 : Rdbm < 0.0 ? 0.0 : Rdbm;
            double Rbgm = (1.0 - clb) * exp(-kp.kd * pait) + clb;
            Rbgm = Rbgm > 1.0 ? 1.0 : Rbgm < 0.0 ? 0.0 : Rbgm;
            o.RswabsG = (1.0 - q.alb) * (Rdbm * Rbeam * cosz) + Rddm * q.Rdif + (1.0 - q.alb) * (Rbgm * Rbeam * 0.5);
        }
    }
    return o;
}

double latent_molar(double t) { return t < 0.0 ? 51078.69 - 4.338 * t - 0.06367 * t * t : 45068.7 - 42.8428 * t; }

SnowStepOut snow_step(const SnowStepIn& q, const SnowDen& sd) {                       // snowoneB with umu = 1
    SnowStepOut o{};
    const double pai = q.hgt > q.sdepg ? q.pai * (q.hgt - q.sdepg) / q.hgt : 0.0;
    double hgt = q.hgt - q.sdepg;
    if (hgt < 0.0) hgt = 0.0;
    const double zi = (q.sdepg > 0.0 && hgt > 0.0) ? ((q.sdepc - q.sdepg) * q.sdenc) / (hgt * 1000.0) : 0.0;
    const SnowRad rad = snow_radiation(q, pai, hgt, q.ltra * exp(-10.1 * zi));
    double d = 0.0, zm = 0.005;
    if (hgt > 0.0) { d = zeroplane(hgt, pai); zm = roughlength(hgt, pai, d, q.psih); }
    if (zm < 0.0009) zm = 0.0009;
    o.hgt = hgt; o.pai = pai;
    o.uf = (kKa * q.u2) / (log((q.zref - d) / zm) + q.psim);
    o.gHa = g_turb(o.uf, d, zm, q.zref, phair(q.tc, q.pk), q.psih, 0.03);
    o.Tc = penman(rad.RabsC, o.gHa, o.gHa, q.tc, q.te, q.pk, q.ea, 0.97, q.G, 1.0);
    o.Tg = penman(rad.RswabsG + rad.RlwabsG, o.gHa, o.gHa, q.tc, q.te, q.pk, q.ea, 0.97, q.G, 1.0);
    const double tdew = dewpoint(q.ea);
    if (o.Tc < tdew) o.Tc = tdew;
    if (o.Tg < tdew) o.Tg = tdew;
    // whole pack: sublimation, temperature melt, rain melt (m of water equivalent per hour)
    double la = latent_molar(o.Tc);
    o.mSc = ((la * (o.gHa / q.pk) * (satvap(o.Tc) - q.ea)) / (la / 0.018015)) * 3.6;
    o.Tcp = o.Tc;
    if (o.Tc > 0.0) {
        o.mMc = ((583.3 * o.Tc * (q.sdepc * (q.sdenc / 1000))) / 334000.0) * 3.6;
        if (q.sdepc > 0.0) o.Tc = 0.0;
    }
    if (q.tc > 0.0) o.mRc = 0.0125 * q.tc * q.prec / 1000;
    // ground pack
    la = latent_molar(o.Tg);
    double mu = exp(-pai);
    if (mu > 1.0) mu = 1.0;
    const double mSg = ((la * (o.gHa / q.pk) * (satvap(o.Tg) - q.ea) * mu) / (la / 0.018015)) * 3.6;
    double mMg = 0.0;
    if (o.Tg > 0.0) {
        mMg = ((583.3 * o.Tg * (q.sdepg * (q.sdeng / 1000.0))) / 334000.0) * 3.6;
        if (q.sdepg > 0.0) o.Tg = 0.0;
    }
    double Li = 0.0;
    if (q.sdepc > 0.0) {
        double wg = q.sdepg / q.sdepc;
        wg = wg < 0.0 ? 0.0 : wg > 1.0 ? 1.0 : wg;
        Li = (q.sdepc - q.sdepg) * (wg * q.sdeng + (1.0 - wg) * q.sdenc);
    }
    if (Li < 0.0) Li = 0.0;
    double cis = canopy_snow_interception(hgt, pai, o.uf, q.prec, q.tc, Li);
    if (cis > q.prec) cis = q.prec;
    const double mRg = q.tc > 0.0 ? 0.0125 * q.tc * (q.prec - cis) / 1000.0 : 0.0;
    const double snowc = q.tc > 2.0 ? 0.0 : q.prec, snowg = q.tc > 2.0 ? 0.0 : q.prec - cis;
    const double swec = snowc / 1000.0 - o.mSc - o.mMc - o.mRc, sweg = snowg / 1000.0 - mSg - mMg - mRg;
    o.agec = q.agec + 1.0; o.ageg = q.ageg + 1.0;
    o.sdenc = snow_density(sd, q.sdepc, o.agec);
    o.sdeng = snow_density(sd, q.sdepg, o.ageg);
    o.sdepc = q.sdepc + (swec * 1000.0) / o.sdenc;
    o.sdepg = q.sdepg + (sweg * 1000.0) / o.sdeng;
    if (o.sdepc < 0.0) { o.sdepc = 0.0; o.agec = 0.0; }
    if (o.sdepg < 0.0) { o.sdepg = 0.0; o.ageg = 0.0; }
    o.RswabsG = rad.RswabsG; o.RlwabsG = rad.RlwabsG; o.tr = rad.tr;
    return o;
}

void snow_ground_flux(const double* Ts, const double* den, size_t n, Vec& G) {          // GFluxCppsnow
    Vec Gmu(n), dT(n), Td(n, 0.0), Gmud(n), t(Ts, Ts + n);
    hour_to_day(t, MEAN, Td);
    for (size_t i = 0; i < n; ++i) {
        const double k = 0.0442 * exp(5.181 * den[i] / 1000), kap = k / (den[i] * 2090);
        Gmu[i] = sqrt(2.0) * (k / sqrt(2.0 * kap / kOmdy)) * 0.5;
        dT[i] = Ts[i] - Td[i];
    }
    moving_mean(Gmu, 6, Gmud);
    moving_mean(dT, 6, G);
    for (size_t i = 0; i < n; ++i) G[i] = G[i] * Gmud[i] * 1.1171;
}

}  // namespace

extern "C" int mcf_pointmodelsnow(int64_t n64, const mcf_obstime* t, const mcf_point_weather* w, const double* vegp,
                                  const double* other, int32_t snowenv, double tol, double maxiter, mcf_pointsnow_out* o) {
    if (!t || !w || !vegp || !other || !o) return mcf::api_fail(MCF_ERR_ARG, "mcf_pointmodelsnow: null argument");
    int rc = check_series(t, w, n64, true);
    if (rc) return rc;
    const size_t n = (size_t)n64;
    double* outs[] = {o->Tc, o->Tg, o->sdenc, o->sdeng, o->G, o->RswabsG, o->RlwabsG, o->tr, o->umu, o->sublmelt, o->tempmelt,
                      o->rainmelt, o->sstemp, o->sdepc, o->sdepg};
    for (double* q : outs)
        if (!q) return mcf::api_fail(MCF_ERR_ARG, "mcf_pointmodelsnow: null output vector");
    Vec ea(n), te(w->temp, w->temp + n), salb(n), psih(n, 0.0), psim(n, 0.0), G(n), Tco(n), Tgo(n);
    for (size_t i = 0; i < n; ++i) ea[i] = satvap(w->temp[i]) * w->relhum[i] / 100.0;
    const double slope = other[0], aspect = other[1], lat = other[2], lon = other[3], zref = other[4], isnowd = other[5],
                 isnowa = other[6];
    snow_albedo(w->precip, n, salb);
    const SnowDen sd = snow_density_params(snowenv);
    for (size_t i = 0; i < n; ++i) {
        o->sdenc[i] = o->sdeng[i] = snow_density(sd, isnowd, 0.0);
        o->Tc[i] = o->Tg[i] = w->temp[i];
    }
    snow_ground_flux(w->temp, o->sdenc, n, G);
    double tst = 100.0, mxdif = 0.0;
    int iter = 0;
    while (tst > tol) {
        double agec = (double)(int)isnowa, ageg = (double)(int)isnowa;       // `int snowagec = isnowa`
        o->sdepc[0] = isnowd;
        o->sdepg[0] = isnowd * 0.5;
        for (size_t i = 0; i < n; ++i) { Tco[i] = o->Tc[i]; Tgo[i] = o->Tg[i]; }
        mxdif = 0.0;
        for (size_t i = 0; i < n; ++i) {
            SnowStepIn q{};
            q.year = t->year[i]; q.month = t->month[i]; q.day = t->day[i]; q.hour = t->hour[i];
            q.tc = w->temp[i]; q.ea = ea[i]; q.pk = w->pres[i]; q.u2 = w->windspeed[i]; q.prec = w->precip[i];
            q.Rsw = w->swdown[i]; q.Rdif = w->difrad[i]; q.Rlw = w->lwdown[i]; q.Tci = o->Tc[i]; q.te = te[i];
            q.pai = vegp[0]; q.hgt = vegp[1]; q.ltra = vegp[2]; q.clump = vegp[3];
            q.alb = salb[i]; q.sdenc = o->sdenc[i]; q.sdeng = o->sdeng[i]; q.sdepc = o->sdepc[i]; q.sdepg = o->sdepg[i];
            q.agec = agec; q.ageg = ageg;
            q.slope = slope; q.aspect = aspect; q.lat = lat; q.lon = lon; q.zref = zref;
            q.psim = psim[i]; q.psih = psih[i]; q.G = G[i];
            const SnowStepOut s = snow_step(q, sd);
            agec = (double)(int)s.agec; ageg = (double)(int)s.ageg;
            o->sdepc[i + 1] = s.sdepc;
            o->sdepg[i + 1] = s.sdepg;
            o->Tc[i] = 0.5 * Tco[i] + 0.5 * s.Tc;
            o->Tg[i] = 0.5 * Tgo[i] + 0.5 * s.Tg;
            mxdif = fmax(mxdif, fmax(fabs(o->Tc[i] - Tco[i]), fabs(o->Tg[i] - Tgo[i])));
            // stability of the surface layer for the next pass
            const double cp = cpair(w->temp[i]), ph = phair(w->temp[i], w->pres[i]);
            double H = cp * s.gHa * (o->Tc[i] - w->temp[i]);
            const double d = zeroplane(s.hgt, s.pai);
            double zm = roughlength(s.hgt, s.pai, d, psih[i]);
            if (zm < 0.001) zm = 0.001;
            if (fabs(H) < 0.1) H = 0.1;
            const double LL = (ph * cp * pow(s.uf, 3.0) * (w->temp[i] + 273.15)) / (-kKa * 9.81 * H);
            psim[i] = psi_m(zm / LL) - psi_m((zref - d) / LL);
            psih[i] = psi_h((0.2 * zm) / LL) - psi_h((zref - d) / LL);
            const double Belim = 0.4 / sqrt(0.003 + (0.2 * s.pai) / 2.0);
            const double ln1 = log((zref - d) / zm), ln2 = log((zref - d) / (0.2 * zm));
            if (psim[i] < -0.9 * ln1) psim[i] = -0.9 * ln1;
            if (psih[i] < -0.9 * ln2) psih[i] = -0.9 * ln2;
            if (psim[i] > 0.9 * ln1) psim[i] = 0.9 * ln1;
            if (psih[i] > 0.9 * ln2) psih[i] = 0.9 * ln2;
            if (psih[i] > 0.9 * Belim) psih[i] = 0.9 * Belim;
            o->RswabsG[i] = s.RswabsG; o->RlwabsG[i] = s.RlwabsG; o->tr[i] = s.tr;
            o->umu[i] = s.uf / ((0.4 * w->windspeed[i]) / log((zref - d) / zm));
            te[i] = (o->Tc[i] + w->temp[i]) / 2.0;
            o->sublmelt[i] = s.mSc; o->tempmelt[i] = s.mMc; o->rainmelt[i] = s.mRc; o->sstemp[i] = s.Tcp;
        }
        snow_ground_flux(o->Tg, o->sdenc, n, G);
        tst = mxdif;
        if (++iter > maxiter) tst = 0;
    }
    for (size_t i = 0; i < n; ++i) o->G[i] = G[i];
    o->mxdif = mxdif;
    o->iters = iter;
    return MCF_OK;
}

// canintfrac (src/microclimfCpp.cpp:5417-5450): the share of a typical snowfall the canopy of each cell holds back
extern "C" int mcf_canintfrac(int64_t cells, const double* hgt, const double* pai, double uf, double prec, double tc, double Li,
                              double* frac) {
    if (cells <= 0 || !hgt || !pai || !frac) return mcf::api_fail(MCF_ERR_ARG, "mcf_canintfrac: null argument or no cells");
    for (int64_t i = 0; i < cells; ++i) {
        if (isnan(hgt[i])) frac[i] = NAN;
        else frac[i] = prec > 0.0 ? canopy_snow_interception(hgt[i], pai[i], uf, prec, tc, Li) / prec : 0.5;   // NaN prec: 0.5 too
    }
    return MCF_OK;
}
// meltmu (src/microclimfCpp.cpp:5454-5492): degree hours of the snow surface with each cell's sky view scaling its
// departure from air temperature, over the degree hours of the point model's surface
extern "C" int mcf_meltmu(int64_t cells, const double* skyview, int64_t n, const double* stemp, const double* tc, double* mu) {
    if (cells <= 0 || n < 0 || !skyview || !mu || (n > 0 && (!stemp || !tc)))
        return mcf::api_fail(MCF_ERR_ARG, "mcf_meltmu: null argument or no cells");
    double dhp = 0.0;
    for (int64_t k = 0; k < n; ++k)
        if (stemp[k] > 0.0) dhp += stemp[k];
    for (int64_t i = 0; i < cells; ++i) {
        if (!(dhp > 0.0)) { mu[i] = 1.0; continue; }          // NA cells included, as in the reference
        if (isnan(skyview[i])) { mu[i] = NAN; continue; }
        double dhm = 0.0;
        for (int64_t k = 0; k < n; ++k) {
            const double s2 = (stemp[k] - tc[k]) * skyview[i] + tc[k];
            if (s2 > 0.0) dhm += s2;
        }
        mu[i] = dhm / dhp;
    }
    return MCF_OK;
}
// meltmu2 (src/microclimfCpp.cpp:5495-5527): meltmu with per-cell series, stemp / tc [cells, n] with the cell index fastest
extern "C" int mcf_meltmu2(int64_t cells, int64_t n, const double* mu, const double* stemp, const double* tc, double* out) {
    if (cells <= 0 || n < 0 || !mu || !out || (n > 0 && (!stemp || !tc)))
        return mcf::api_fail(MCF_ERR_ARG, "mcf_meltmu2: null argument or no cells");
    for (int64_t i = 0; i < cells; ++i) {
        if (isnan(mu[i])) { out[i] = NAN; continue; }
        double dhp = 0.0, dhm = 0.0;
        for (int64_t k = 0; k < n; ++k) {
            const double st = stemp[i + cells * k], ta = tc[i + cells * k];
            if (st > 0.0) dhp += st;
            const double s2 = (st - ta) * mu[i] + ta;
            if (s2 > 0.0) dhm += s2;
        }
        out[i] = dhp > 0.0 ? dhm / dhp : 0.5;
    }
    return MCF_OK;
}
