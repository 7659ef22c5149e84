// mcf_snow_device.hpp — device physics of the snow branch.
//
// Restates, for one lane = one cell, the reference's snowpack energy / mass balance
//   snowoneB + radoneB + canopysnowintCpp          src/microclimfCpp.cpp:3713-3972  ("cpp:")
// and snow microclimate
//   snowabovepoint, belowpointsnow                  cpp:4739-4891
// Everything that depends only on the time step (vapour pressures, Penman-Monteith coefficients,
// sun position, albedo, daily radiation extremes) is gathered in StepT: with data.frame climate
// it is tabulated once per step on the device (k_snow_steps / k_snow_days), with array climate
// the same derive functions run per cell-step.
//
// Unlike the no-snow solver this path has zero and non-finite operands by construction (log(0) in the albedo,
// pow(0, Kc) for clump = 0, NaN propagation past the last whole day), and the reference's comparison directions are
// kept so that NaNs fall through the same branches.  The lean fp64 routines of mcf_device.hpp are therefore used under
// EXPLICIT OPERAND GUARDS (gexp / glog / gsqrt / gpow0 below): each returns what libm returns for the operands outside
// the lean routine's domain (NaN stays NaN, exp saturates to 0 / inf, log(0) = -inf, log(< 0) = NaN) and the lean value
// inside it.  Divisions by quantities that are finite and non-zero by construction go through fdiv (a * 1/b, 1.5 ulp);
// the few whose divisor can be 0 or infinite stay IEEE divisions.  Trigonometry stays device libm (per step or per cell,
// not in the recurrence), except cos(atan(x)) = 1/sqrt(1 + x^2) in the interception model.
#pragma once
#include "mcf_device.hpp"

namespace mcf {
namespace snow {

// The workgroup's copies of kExp2Tab / kLogTab: ONE static LDS array per kernel, reached from any depth of the call tree
// through this accessor.  EVERY kernel that can reach gexp / glog / gpow0 calls snow_tables_init() first (before any early
// return) — a kernel that did not would read uninitialised LDS; tests/test_snow_gpu.py runs each of them against the oracle.
__device__ __forceinline__ double* snow_tables() {
    __shared__ __attribute__((aligned(16))) double t[512 + 256];
    return t;
}
__device__ __forceinline__ void snow_tables_init() {
    double* t = snow_tables();
    const int n = (int)(blockDim.x * blockDim.y * blockDim.z), tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    for (int i = tid; i < 512; i += n) t[i] = kLogTab[i];
    for (int i = tid; i < 256; i += n) t[512 + i] = kExp2Tab[i];
    __syncthreads();
}
__device__ __forceinline__ void snow_mathk(MathK& K) {
    K.set();
    double* t = snow_tables();        // exp / log through the LDS tables of mcf_device.hpp (fexp_tab, flog_tab)
    K.ltab = t; K.logtab = true;
    K.tab = t + 512; K.table = true;
}
// exp(x) for NaN or |x| < 1e90: the table route saturates by itself (v_cvt_i32_f64 and v_ldexp_f64 saturate: exactly 0 for
// very negative, inf for very positive arguments; NaN stays NaN).  Only an INFINITE argument would come out as NaN — none is
// reachable: the one source of an infinity on this path, pai / (1 - clump) at clump = 1, is a NaN in the lean division
// already.  (Round 2 clamped the argument to [-750, 710] in front of every call: 8 VALU instructions x 16-20 calls a step.)
__device__ __forceinline__ double gexp(double x) {
    MathK K;
    snow_mathk(K);
    return fexp(x, K);
}
__device__ __forceinline__ double glog(double x) {      // log(x) for any operand but +inf: log(0) = -inf, log(< 0) = log(NaN) = NaN
    MathK K;
    snow_mathk(K);
    const double r = flog(x, K);                        // (garbage for x <= 0, replaced below)
    const double bad = (x == 0.0) ? -__longlong_as_double(0x7FF0000000000000LL) : __longlong_as_double(0x7FF8000000000000LL);
    return x > 0.0 ? r : bad;
}
__device__ __forceinline__ double gsqrt(double x) {     // sqrt(x) for ANY operand
    if (x > 1e-300 && x < 1e300) return fsqrt(x);
    return ::sqrt(x);
}
__device__ __forceinline__ double gdiv(double a, double b) { return fdiv(a, b); }   // b finite, non-zero by construction (or NaN)
// pow(b, e) for b >= 0 and e > 0 (clump^Kc): pow(0, e) = 0
__device__ __forceinline__ double gpow0(double b, double e) {
    // b = 0 is the common special case (clump = 0: every bare cell, and a wave that holds one ran the device's 220-instruction
    // pow for all its lanes): pow(0, e > 0) = 0.  Anything else outside b > 0 (NaNs, e <= 0) still goes to libm.
    if (b > 0.0) return gexp(e * glog(b));
    if (b == 0.0 && e > 0.0) return 0.0;
    // What is left is outside the model's domain (a negative or NaN clumping factor, a non-positive exponent).  Round 5: C's pow
    // semantics written out with the lean routines instead of the device libm's pow — 220 instructions and the registers of a
    // second kernel on a path no valid input takes; inlined into the snow-microclimate kernels it cost them 44-48 B of scratch
    // per lane, whose spill traffic was the "1.4 x write amplification" of profiles/r04_c4_aux_pmc_summary.json.
    if (b != b || e != e) return (e == 0.0 || b == 1.0) ? 1.0 : b + e;        // pow(x, 0) = pow(1, y) = 1 even for NaN
    if (e == 0.0) return 1.0;
    if (b == 0.0) return __longlong_as_double(0x7FF0000000000000LL);           // pow(+-0, e < 0): +inf up to the sign of an odd integer e
    // b < 0: defined for integer exponents only
    if (e != __builtin_rint(e)) return __longlong_as_double(0x7FF8000000000000LL);
    const double m = gexp(e * glog(-b));
    return (__builtin_fmod(e, 2.0) != 0.0) ? -m : m;
}

__device__ __forceinline__ double svp(double tc) {  // cpp:480-490 satvapCpp
    const bool w = tc > 0;                              // one division: the constant pair is selected, not the quotient
    const double a = w ? 17.27 : 21.875, b = w ? 237.3 : 265.5;
    return 0.61078 * gexp(gdiv(a * tc, tc + b));
}
__device__ __forceinline__ double rad4(double tc) {  // cpp:24-26 radem
    double t = tc + 273.15;
    t *= t;
    return t * t;
}
__device__ __forceinline__ double dewpoint(double ea) {  // cpp:493-496
    const double l = glog(ea / 0.6112);
    return 243.5 * l / (17.67 - l);
}
// latent heat of vaporisation / sublimation, the `T < 0` flavour of cpp:3879-3884, 4837-4842
__device__ __forceinline__ double latent_lt0(double t) {
    return t < 0.0 ? 51078.69 - 4.338 * t - 0.06367 * t * t : 45068.7 - 42.8428 * t;
}
__device__ __forceinline__ double zeroplane(double h, double pai) {  // cpp:294-299
    if (pai < 0.001) pai = 0.001;
    const double s = gsqrt(7.5 * pai);
    return (1.0 - gdiv(1.0 - gexp(-s), s)) * h;
}
__device__ __forceinline__ double roughlen0(double h, double pai, double d) {  // cpp:302-310, psi_h = 0
    const double Be = gsqrt(0.003 + (0.2 * pai) / 2);
    double zm = (h - d) * gexp(gdiv(-kKa, Be));
    if (zm > (0.9 * (h - d))) zm = 0.9 * (h - d);
    if (zm < 0.0005) zm = 0.0005;
    return zm;
}

// ---- two-stream coefficients for spherical leaves (x = 1 -> J = 1/3), cpp:134-185 -------------
struct TsDif { double om, a, gma, del, h, S1, iS1, iD1, iD2, u1, u2, D1, D2, p1, p2, p3, p4; };
// igref = 1 / gref (the snow albedo: a step value with data.frame climate).  Three reciprocals (1/S1, 1/D1, 1/D2) serve the
// four p's here and ts_dir's six quotients; round 2 took eleven.
__device__ __forceinline__ TsDif ts_dif(double pait, double lref, double ltra, double gref, double igref) {
    TsDif p;
    p.om = lref + ltra;
    p.a = 1.0 - p.om;
    p.del = lref - ltra;
    p.gma = 0.5 * (p.om + (1.0 / 3.0) * p.del);
    p.h = gsqrt(p.a * p.a + 2.0 * p.a * p.gma);
    p.S1 = gexp(-p.h * pait);
    p.u1 = p.a + p.gma * (1.0 - igref);
    p.u2 = p.a + p.gma * (1.0 - gref);
    const double iS1 = gdiv(1.0, p.S1);
    p.iS1 = iS1;                                        // = exp(+h pait) to 1.5 ulp: the callers' second exponential
    p.D1 = (p.a + p.gma + p.h) * (p.u1 - p.h) * iS1 - (p.a + p.gma - p.h) * (p.u1 + p.h) * p.S1;
    p.D2 = (p.u2 + p.h) * iS1 - (p.u2 - p.h) * p.S1;
    p.iD1 = gdiv(1.0, p.D1);
    p.iD2 = gdiv(1.0, p.D2);
    p.p1 = (p.gma * (p.iD1 * iS1)) * (p.u1 - p.h);
    p.p2 = (-p.gma * p.S1 * p.iD1) * (p.u1 + p.h);
    p.p3 = (p.iD2 * iS1) * (p.u2 + p.h);
    p.p4 = (-p.S1 * p.iD2) * (p.u2 - p.h);
    return p;
}
struct TsDir { double sig, isig, S2, p5, p6, p7, p8, p9, p10; };   // sig, isig: of the NEGATED determinant, as p8 .. p10 use it
__device__ __forceinline__ TsDir ts_dir(double pait, const TsDif& f, double gref, double kd) {
    TsDir p;
    const double ag = f.a + f.gma;
    const double sig = kd * kd + f.gma * f.gma - ag * ag;
    const double ss = 0.5 * (f.om + gdiv((1.0 / 3.0) * f.del, kd)) * kd;
    const double sstr = f.om * kd - ss;
    const double S2 = gexp(-kd * pait);
    p.S2 = S2;
    p.p5 = -ss * (ag - kd) - f.gma * sstr;
    const double isg = gdiv(1.0, sig);
    const double v1 = ss - (p.p5 * (ag + kd)) * isg;
    const double v2 = ss - f.gma - (p.p5 * isg) * (f.u1 + kd);
    p.p6 = f.iD1 * ((v1 * f.iS1) * (f.u1 - f.h) - (ag - f.h) * S2 * v2);
    p.p7 = -f.iD1 * ((v1 * f.S1) * (f.u1 + f.h) - (ag + f.h) * S2 * v2);
    p.sig = -sig;
    p.isig = -isg;
    p.p8 = sstr * (ag + kd) - f.gma * ss;
    const double p8s = p.p8 * p.isig;
    const double v3 = (sstr + f.gma * gref - p8s * (f.u2 - kd)) * S2;
    p.p9 = -f.iD2 * ((p8s * f.iS1) * (f.u2 + f.h) + v3);
    p.p10 = f.iD2 * ((p8s * f.S1) * (f.u2 - f.h) + v3);
    return p;
}
// cankCpp for x = 1 (cpp:104-132): k, kd = k cos(z)/si, Kc = 1/si
struct CanK { double k, kd, Kc; };
__device__ __forceinline__ CanK cank1(double kx, double kcos, double si) {
    if (si < 0.0) si = 0.0;
    CanK o;
    o.k = kx;
    o.Kc = gdiv(1.0, si);           // si == 0: overwritten below, whatever the quotient
    o.kd = kcos * o.Kc;
    if (si == 0) o.kd = 1.0;
    if (si == 0.0) o.Kc = 600.0;
    return o;
}
__device__ __forceinline__ double clamp01(double v) {  // `if (v > 1) v = 1; if (v < 0) v = 0;`
    if (v > 1.0) v = 1.0;
    if (v < 0.0) v = 0.0;
    return v;
}

// ---- per-time-step values ----------------------------------------------------------------------
struct SunT {          // sun position pieces of one step at one site
    double zend, zenr, azid;
    double cosz;       // cos(zenr)                                  cpp:3798
    double cz, sz;     // cos / sin (zend * torad)                   cpp:85-102
    double ca, sa;     // cos / sin (azid * torad)
    double kx, kcos;   // cankCpp(zenr, 1, .): k and k*cos(zenr')    cpp:104-132
    double tansa;      // horizon-test threshold                     cpp:4357-4359 / 4604-4606
};
// `degrees`: gridmodelsnow1 tests `ha > tan((90 - zend) * torad)`, every other caller
// `ha > tan(pi/2 - zenr)`.
__device__ __forceinline__ SunT sun_derive(const SolPos& sp, bool degrees) {
    SunT s;
    s.zend = sp.zend; s.zenr = sp.zenr; s.azid = sp.azid;
    s.cosz = cos(sp.zenr);
    s.cz = cos(sp.zend * kToRad);
    s.sz = sin(sp.zend * kToRad);
    s.ca = cos(sp.azid * kToRad);
    s.sa = sin(sp.azid * kToRad);
    double zr = sp.zenr;
    if (zr > (kPi / 2.0)) zr = kPi / 2.0;
    const double c = cos(zr);
    double k = 1.0 / (2.0 * c);
    if (k > 6000.0) k = 6000.0;
    s.kx = k;
    s.kcos = k * c;
    s.tansa = degrees ? tan((90 - sp.zend) * kToRad) : tan(kPi / 2.0 - sp.zenr);
    return s;
}
// Array climate: the sun at ONE CELL and step from the date part of the step (declination, the hour angle's time part A =
// 0.261799 (hour + eot / 60 - 12), tabulated per step as cos A / sin A) and the cell's constants (sin / cos of its latitude
// and of B = 0.261799 * 4 lon / 60, cpp:44, 54).  cpp:48-83 followed algebraically, as mcf_device.hpp derive_time_af does for
// the solver: with coh = cos(zenith)
//   cos(zenr) = cos(zend torad) = coh,  sin = sqrt(1 - coh^2),  tan(pi/2 - zenr) = coh / sin,  cos(hh) = sin(zenith),
//   sin(azimuth) = -sazi,  cos(azimuth) = -+sqrt(1 - sazi^2) by the sign of cazi (cpp:65-75),
// the 15-degree sector round(azid / 15) % 24 by comparing tangents — no inverse trigonometry, no libm call.  Round 5: until
// then k_snowmodel<true> / k_microsnow<true> called sol_site + sun_derive per cell-step — two inverse and twelve direct libm
// trigonometric functions, each with its large-argument reduction inlined: 168 VGPRs + 116 B of scratch in the snow model.
// Differences to the literal evaluation are rounding-level (1e-16).  `zend` is only ever compared with 90 (solar_index):
// 45 / 135 stand for "above / below the horizon"; zenr / azid are not read by any consumer and stay NaN.
struct SunCell { double sinlat, coslat, cosB, sinB; };
__device__ __forceinline__ SunCell sun_cell(double lat_deg, double lon_deg) {
    SunCell c;
    const double latr = lat_deg * kPi / 180.0, B = 0.261799 * (4.0 * lon_deg) / 60.0;
    c.sinlat = sin(latr); c.coslat = cos(latr); c.cosB = cos(B); c.sinB = sin(B);
    return c;
}
__device__ __forceinline__ SunT sun_at_cell(double sindec, double cosdec, double cosA, double sinA, const SunCell& c, int& sindex) {
    SunT s;
    const double ctt = cosA * c.cosB - sinA * c.sinB;
    const double stt = sinA * c.cosB + cosA * c.sinB;
    const double coh = sindec * c.sinlat + cosdec * c.coslat * ctt;            // cpp:56
    double s2 = 1.0 - coh * coh;
    if (s2 < 0.0) s2 = 0.0;
    const double sz = fsqrt(s2 > 1e-300 ? s2 : 1e-300);
    const bool up = coh >= 0.0;
    s.zend = up ? 45.0 : 135.0;
    s.zenr = s.azid = __longlong_as_double(0x7FF8000000000000LL);
    s.cosz = coh; s.cz = coh; s.sz = sz;
    s.tansa = fdiv(coh, sz);
    const double cc = up ? coh : 6.123233995736766e-17;                        // cos(pi/2) in fp64 (cpp:106)
    double k = 0.5 * frcp(cc);
    if (k > 6000.0) k = 6000.0;
    s.kx = k;
    s.kcos = k * cc;
    double sazi = fdiv(cosdec * stt, sz);                                       // cpp:59-61
    const double num = c.sinlat * cosdec * ctt - c.coslat * sindec;            // sign of cazi, cpp:62-64
    double sqt = 1.0 - sazi * sazi;
    if (sqt < 0.0) sqt = 0.0;
    if (sazi > 1.0) sazi = 1.0;
    if (sazi < -1.0) sazi = -1.0;
    const double rq = fsqrt(sqt > 1e-300 ? sqt : 1e-300);
    s.sa = -sazi;
    s.ca = num < 0.0 ? rq : -rq;
    // sindex = round(azid / 15) % 24: rotate by +7.5 degrees, quadrant, tangent tests
    const double xr = s.ca * 0.99144486137381038 - s.sa * 0.13052619222005157;
    const double yr = s.sa * 0.99144486137381038 + s.ca * 0.13052619222005157;
    int q;
    double u, v;
    if (yr >= 0.0) {
        if (xr > 0.0) { q = 0; u = xr; v = yr; } else { q = 1; u = yr; v = -xr; }
    } else {
        if (xr < 0.0) { q = 2; u = -xr; v = -yr; } else { q = 3; u = -yr; v = xr; }
    }
    const int n = (v >= u * 0.26794919243112270) + (v >= u * 0.57735026918962573) + (v >= u) + (v >= u * 1.7320508075688772) +
                  (v >= u * 3.7320508075688776);
    sindex = (6 * q + n) % 24;
    return s;
}
struct SiteK { double cS, sS, cA, sA; bool flat; };   // slope / aspect of the cell
__device__ __forceinline__ SiteK site_derive(double slope, double aspect) {
    SiteK k;
    k.cS = cos(slope * kToRad); k.sS = sin(slope * kToRad);
    k.cA = cos(aspect * kToRad); k.sA = sin(aspect * kToRad);
    k.flat = (slope == 0.0);
    return k;
}
// solarindexCpp (cpp:85-102) with cos((azid - aspect) torad) expanded
__device__ __forceinline__ double solar_index(const SunT& s, const SiteK& k, bool shadowmask) {
    double si;
    if (s.zend > 90.0 && !shadowmask) {
        si = 0;
    } else if (k.flat) {
        si = s.cz;
    } else {
        si = s.cz * k.cS + s.sz * k.sS * (s.ca * k.cA + s.sa * k.sA);
    }
    if (si < 0.0) si = 0.0;
    return si;
}

struct MetT {          // weather-only values of one step (snowpack model)
    double tc, prec, pk, ea, te, rcan, rem;
    double la, cp, Da, gR, De;   // PenmanMonteithCpp's step-only terms, cpp:498-514
    double tdew, ph, sint;       // dewpoint, molar density, max snow load per branch area (cpp:3728-3729)
    double rsw, rdif, rlw, umu, u2, gp, alb, ialb;   // ialb = 1 / alb
};
__device__ __forceinline__ void met_derive(MetT& m, double tc, double rh, double pk, double tci) {
    m.tc = tc; m.pk = pk;
    const double es = svp(tc);
    m.ea = es * rh / 100.0;                                    // cpp:4226
    m.te = (tci + tc) / 2.0;                                   // cpp:4227
    m.rem = 0.97 * kSb * rad4(tc);                             // cpp:4228, 501
    m.rcan = 0.97 * kSb * rad4(tci);                           // cpp:3785
    const double te = m.te;
    m.la = te >= 0 ? 45068.7 - 42.8428 * te : 51078.69 - 4.338 * te - 0.06367 * te * te;
    m.cp = 2e-05 * te * te + 0.0002 * te + 29.119;             // cpp:287-291
    m.Da = es - m.ea;
    const double tk = te + 273.15;
    m.gR = (4.0 * 0.97 * kSb * (tk * tk * tk)) / m.cp;
    m.De = svp(te + 0.5) - svp(te - 0.5);
    m.tdew = dewpoint(m.ea);
    m.ph = 44.6 * (pk / 101.3) * (273.15 / (tc + 273.15));     // cpp:280-285
    const double rhos = 67.92 + 51.25 * gexp(tc / 2.59);
    m.sint = 6.2 * (0.26 + 46 / rhos);
}
struct DayT { double rmx, rmn, rswmx, rlwmx, rswmn, rlwmn, gmx; };   // cpp:4231-4282
__device__ __forceinline__ void day_init(DayT& d) {
    d.rmx = -1352.0; d.rmn = 1352.0;
    d.rswmx = d.rlwmx = d.rswmn = d.rlwmn = 0.0;
    d.gmx = 0.0;
}
__device__ __forceinline__ void day_accum(DayT& d, double rnet, double rsw, double rlw) {
    if (d.rmx < rnet) { d.rmx = rnet; d.rswmx = rsw; d.rlwmx = rlw; }
    if (d.rmn > rnet) { d.rmn = rnet; d.rswmn = rsw; d.rlwmn = rlw; }
    if (fabs(rnet) > d.gmx) d.gmx = fabs(rnet);
}
// snowalbCpp (cpp:3752-3771): the logarithm's argument is the INTEGER quotient hs / 24
__device__ __forceinline__ double snow_albedo(int hs) {
    double alb = (-9.8740 * glog((double)(hs / 24)) + 78.3434) / 100.0;
    if (alb > 0.95) alb = 0.95;
    if (alb < 0.1) alb = 0.1;
    return alb;
}

// ---- snowpack state and one step of snowoneB ----------------------------------------------------
struct Pack { double sdenc, sdeng, sdepc, sdepg; int agec, ageg; };
struct PackOut { double Tc, Tg, melc, melg; };
// the cell's constants (rasters, slope / aspect sines and cosines), read through `p[f * cs]` AT THEIR USES: k_snowmodel keeps
// them in its workgroup's LDS table (cs = 256) — nine doubles fewer per lane across the whole step
enum CellVF : int { CV_PAI, CV_HGT, CV_CLUMP, CV_LTRA, CV_SKYVIEW, CV_CS, CV_SS, CV_CA, CV_SA, CV_SLOPE, CV_COUNT };
struct CellV {
    const double* p;
    int cs;
    __device__ __forceinline__ double C(int f) const { return p[f * cs]; }
    __device__ __forceinline__ SiteK site() const {
        SiteK k;
        k.cS = C(CV_CS); k.sS = C(CV_SS); k.cA = C(CV_CA); k.sA = C(CV_SA); k.flat = C(CV_SLOPE) == 0.0;
        return k;
    }
};

__device__ __forceinline__ double snow_density(const double* sdp, double depth, double age_h) {  // cpp:3952-3955
    return ((sdp[0] - sdp[1]) * (1.0 - gexp(-sdp[2] * depth * (1.0 / 100.0) - sdp[3] * age_h * (1.0 / 24.0))) + sdp[1]) * 1000.0;
}

// The body of the k loop of gridmodelsnow1/2 for a step that passed `snowtest` (cpp:4340-4396).
__device__ __forceinline__ void pack_step(const MetT& m, const DayT& dy, const SunT& sun, const CellV& c, double ha,
                                          double ws, const double* sdp, double zref, Pack& s, PackOut& o) {
    // ground heat flux of the cell from the point model's (cpp:4341-4354)
    double paip = c.C(CV_PAI);
    const double ihgt = gdiv(1.0, c.C(CV_HGT) > 0.0 ? c.C(CV_HGT) : 1.0);      // only used under hgt > sdepg >= 0
    const bool emerged = c.C(CV_HGT) > s.sdepg;
    if (emerged) paip = paip * (c.C(CV_HGT) - s.sdepg) * ihgt;
    const double dtR = dy.rmx - dy.rmn;
    const double epaip = gexp(-paip);                  // = exp(-pai) of the ground pack's sublimation below when `emerged`
    const double trS = c.C(CV_SKYVIEW) * epaip;
    const double dmxS = trS * dy.rswmx + trS * dy.rlwmx + (1 - trS) * m.rem - m.rem;
    const double dmnS = trS * dy.rswmn + trS * dy.rlwmn + (1 - trS) * m.rem - m.rem;
    double G = m.gp * ((dmxS - dmnS) / dtR);       // IEEE: dtR is 0 past the last whole day (0/0 = NaN there)
    if (G > dy.gmx) G = dy.gmx;
    if (G < -dy.gmx) G = -dy.gmx;
    // terrain-adjusted forcing (cpp:4355-4367)
    double smu = 1.0;
    if (ha > sun.tansa) smu = 0.0;
    const double u2p = m.umu * ws * m.u2;
    const double Rdif = m.rdif * c.C(CV_SKYVIEW);
    const double Rsw = (m.rsw - m.rdif) * smu + Rdif;
    const double Rlw = m.rlw * c.C(CV_SKYVIEW);
    // vegetation above the ground snow (cpp:3840-3852)
    double pai = 0.0;
    if (emerged) pai = paip;                           // the same expression, cpp:3841 / 4343
    double hgt = c.C(CV_HGT) - s.sdepg;
    if (hgt < 0.0) hgt = 0.0;
    double zi = 0.0;
    if (s.sdepg > 0.0 && hgt > 0.0) zi = gdiv((s.sdepc - s.sdepg) * s.sdenc, hgt * 1000.0);
    double ltra = c.C(CV_LTRA) * gexp(-10.1 * zi);
    // radoneB (cpp:3773-3833)
    const double RlwabsC = 0.97 * Rlw;
    double RlwabsG = RlwabsC;
    const double cld = c.C(CV_CLUMP) * c.C(CV_CLUMP);
    const double pait = gdiv(pai, 1.0 - c.C(CV_CLUMP));
    const double ept = gexp(-pait);
    const double tr = (1.0 - cld) * ept + cld;
    if (hgt > 0.0) RlwabsG = 0.97 * (tr * Rlw + (1.0 - tr) * m.rcan);
    double RabsC = RlwabsC, RswabsG = 0.0;
    if (Rsw > 0.0) {
        const double si = solar_index(sun, c.site(), false);
        double Rbeam = gdiv(Rsw - Rdif, sun.cosz);
        if (Rbeam > 1352.2) Rbeam = 1352.2;
        const double RswabsC = (1.0 - m.alb) * (Rdif + Rbeam * sun.cosz);
        RabsC = RswabsC + RlwabsC;
        RswabsG = RswabsC;
        if (hgt > 0.0) {
            if ((m.alb + ltra) > 0.999) ltra = 0.999 - m.alb;
            const TsDif f = ts_dif(pait, m.alb, ltra, m.alb, m.ialb);
            const CanK kp = cank1(sun.kx, sun.kcos, si);
            const TsDir d = ts_dir(pait, f, m.alb, kp.kd);
            const double clb = gpow0(c.C(CV_CLUMP), kp.Kc);
            const double ehp = f.iS1;                       // exp(h pait) = 1 / S1
            const double ekp = d.S2;                        // exp(-kd pait), ts_dir's own
            const double Rddm = clamp01((1.0 - cld) * (f.p3 * f.S1 + f.p4 * ehp) + cld);
            const double Rdbm = clamp01((1.0 - clb) * ((d.p8 * d.isig) * ekp + d.p9 * f.S1 + d.p10 * ehp));
            const double Rbgm = clamp01((1.0 - clb) * ekp + clb);
            const double RdifG = (1.0 - m.alb) * (Rdbm * Rbeam * sun.cosz) + Rddm * Rdif;
            const double RdirG = (1.0 - m.alb) * (Rbgm * Rbeam * 0.5);
            RswabsG = RdifG + RdirG;
        }
    }
    const double RabsG = RswabsG + RlwabsG;
    // turbulent exchange (cpp:3857-3869)
    double d0 = 0.0, zm = 0.005;
    if (hgt > 0.0) {
        d0 = zeroplane(hgt, pai);
        zm = roughlen0(hgt, pai, d0);
    }
    if (zm < 0.0009) zm = 0.0009;
    const double izm = gdiv(1.0, zm);
    const double uf = gdiv(kKa * u2p, glog((zref - d0) * izm));
    double gHa = gdiv(kKa * m.ph * uf, glog(gdiv(zref - d0, 0.2 * zm + d0 - d0)));   // gturbCpp, cpp:373-380
    if (gHa < 0.03) gHa = 0.03;
    // surface temperatures (cpp:3871-3875, PenmanMonteithCpp cpp:498-514 with gV = gHa, erh = 1)
    const double gpk = gdiv(gHa, m.pk);
    const double lg = m.la * gpk;
    const double den = m.cp * (gHa + m.gR) + lg * m.De;
    const double iden = gdiv(1.0, den);
    double Tc = m.tc + ((RabsC - m.rem - lg * m.Da - G) * iden);
    double Tg = m.tc + ((RabsG - m.rem - lg * m.Da - G) * iden);
    if (Tc < m.tdew) Tc = m.tdew;
    if (Tg < m.tdew) Tg = m.tdew;
    // canopy + ground pack: sublimation, melt, rain melt (cpp:3878-3901)
    double la = latent_lt0(Tc);
    double L = la * gpk * (svp(Tc) - m.ea);
    la = la * (1.0 / 0.018015);
    const double mSc = gdiv(L, la) * 3.6;
    double mMc = 0.0;
    if (Tc > 0.0) {
        const double S = s.sdepc * (s.sdenc * 0.001);
        mMc = ((583.3 * Tc * S) * (1.0 / 334000.0)) * 3.6;
        if (s.sdepc > 0.0) Tc = 0.0;
    }
    double mRc = 0.0;
    if (m.tc > 0.0) mRc = 0.0125 * m.tc * m.prec * 0.001;
    // ground pack (cpp:3904-3922)
    la = latent_lt0(Tg);
    double mu = emerged ? epaip : 1.0;                 // exp(-pai), pai = 0 under the pack
    if (mu > 1.0) mu = 1.0;
    L = la * gpk * (svp(Tg) - m.ea) * mu;
    la = la * (1.0 / 0.018015);
    const double mSg = gdiv(L, la) * 3.6;
    double mMg = 0.0;
    if (Tg > 0.0) {
        const double S = s.sdepg * (s.sdeng * 0.001);
        mMg = ((583.3 * Tg * S) * (1.0 / 334000.0)) * 3.6;
        if (s.sdepg > 0.0) Tg = 0.0;
    }
    // canopy interception (cpp:3924-3934, canopysnowintCpp cpp:3713-3739)
    double Li = 0.0;
    if (s.sdepc > 0.0) {
        double wgtg = gdiv(s.sdepg, s.sdepc);
        if (wgtg < 0.0) wgtg = 0.0;
        if (wgtg > 1.0) wgtg = 1.0;
        Li = (s.sdepc - s.sdepg) * (wgtg * s.sdeng + (1.0 - wgtg) * s.sdenc);
    }
    if (Li < 0.0) Li = 0.0;
    // With prec = 0 the intercepted amount is (Lstr - Li) (1 - exp(-0)) 0.678 = +-0 and only `prec - cis` reads it: the
    // whole model (three exponentials, two square roots, five divisions) is skipped on dry steps — unless a NaN is on its
    // way through Li or uf, which the reference would pass on to the ground pack.
    double cis = 0.0;
    if (!(m.prec == 0.0 && Li == Li && uf == uf)) {
        double h = hgt, p = pai;
        if (h < 0.001) h = 0.001;
        if (p < 0.001) p = 0.001;
        const double Be = gsqrt(0.003 + (0.2 * p) * 0.5);
        const double uh = gdiv(uf, Be);
        const double Lc = gdiv(h, 0.25 * p);                       // 1 / (0.25 * (p / h))
        const double Lm = 2.0 * (Be * Be * Be) * Lc;
        const double k1 = gdiv(Be, Lm);
        double uzm = gdiv(uh, h * k1) * (1 - gexp(-k1 * h));
        if (uzm < uf) uzm = uf;
        const double Lstr = m.sint * p;
        const double tz = uzm * (1.0 / 0.8);
        const double kc = 0.5 * gsqrt(1.0 + tz * tz);              // 1 / (2 cos(atan(t))) = sqrt(1 + t^2) / 2
        const double Cp = 1.0 - gexp(-kc * p);
        const double I1 = (Lstr - Li) * (1.0 - gexp(-gdiv(Cp, Lstr) * m.prec));
        cis = I1 * 0.678;
        if (cis > m.prec) cis = m.prec;
    }
    double mRg = 0.0;
    if (m.tc > 0.0) mRg = 0.0125 * m.tc * (m.prec - cis) * 0.001;
    // mass balance, density, age (cpp:3941-3965)
    double snowc = m.prec, snowg = m.prec - cis;
    if (m.tc > 2.0) { snowc = 0.0; snowg = 0.0; }
    const double swec = snowc * 0.001 - mSc - mMc - mRc;
    const double sweg = snowg * 0.001 - mSg - mMg - mRg;
    double agec = (double)s.agec + 1.0, ageg = (double)s.ageg + 1.0;
    const double sdenc = snow_density(sdp, s.sdepc, agec);
    const double sdeng = snow_density(sdp, s.sdepg, ageg);
    double sdepc = s.sdepc + gdiv(swec * 1000.0, sdenc);
    double sdepg = s.sdepg + gdiv(sweg * 1000.0, sdeng);
    if (sdepc < 0.0) { sdepc = 0.0; agec = 0.0; }
    if (sdepg < 0.0) { sdepg = 0.0; ageg = 0.0; }
    s.sdenc = sdenc; s.sdeng = sdeng; s.sdepc = sdepc; s.sdepg = sdepg;
    s.agec = (int)agec; s.ageg = (int)ageg;
    o.Tc = Tc; o.Tg = Tg;
    o.melc = mSc + mMc + mRc;
    o.melg = mSg + mMg + mRg;
}

// ---- snow microclimate: snowabovepoint (cpp:4739-4866) ------------------------------------------
// Round 3: the weather-only terms of a step (MicroMet: three saturation pressures, the dew point's logarithm, the
// Penman-Monteith constants) come from the step table with data.frame climate (k_micro_steps; the kernels' day is
// blockIdx.y, so a row is read through the scalar cache) and from the same function per cell-step with array climate; every
// division whose divisor is finite and non-zero by construction is the lean one; pow(5, 1.5) and sqrt(5) are constants (the
// device evaluated both at every in-canopy cell-step: ~230 VALU instructions); the five powers of the clumping factor
// share their logarithms; exp(+h pai) is the reciprocal of exp(-h pai); sin / cos of pi z / h, z < h, by a [0, pi] routine.
struct MicroMet {
    double es, ea, tdew;      // satvap(tc), vapour pressure, dew point                   cpp:4745-4747
    double De, gHrad, Rem;    // PenmanMonteith2Cpp's step-only operands                  cpp:1220-1247
    double la_pm, lat0;       // latent heat, the `>= 0` and the `< 0` flavours           cpp:1226-1229, 4837-4842
    double dTmx;              // leaf temperature cap from the series' maximum            cpp:1355
    double ipk;               // 1 / pressure
};
__device__ __forceinline__ MicroMet micro_met(double tc, double relhum, double pk, double mxtc) {
    MicroMet m;
    m.es = svp(tc);
    m.ea = m.es * relhum / 100.0;
    m.tdew = dewpoint(m.ea);
    m.De = svp(tc + 0.5) - svp(tc - 0.5);
    const double tk = tc + 273.15;
    m.gHrad = (4 * 0.97 * kSb * (tk * tk * tk)) / 29.3;
    m.Rem = 0.97 * kSb * rad4(tc);
    m.la_pm = tc >= 0 ? 45068.7 - 42.8428 * tc : 51078.69 - 4.338 * tc - 0.06367 * tc * tc;
    m.lat0 = latent_lt0(tc);
    m.dTmx = -0.6273 * mxtc + 49.79;
    m.ipk = gdiv(1.0, pk);
    return m;
}
// the cell's values: raw rasters, glog(clump) where clump > 0, three reciprocals.  Read through `cell[f * cs]` AT THEIR USES:
// k_microsnow_ring points this at its workgroup's LDS table (cs = 64), so that a value occupies registers only around its
// use; the lane-per-(cell, day) kernels point it at a local array (cs = 1: registers, as before).
enum MicroCellF : int { MQ_HGT, MQ_PAI, MQ_PAIA, MQ_LEAFD, MQ_CLUMP, MQ_LTRA, MQ_LEAFDEN, MQ_SVFA, MQ_LNCLUMP, MQ_IHGT, MQ_ILEAFD, MQ_IPAI,
                        MQ_COUNT };
struct MicroIn {
    double reqhgt, zref, tc, pk, u2, Rsw, Rdif, Rlw;               // step
    const double* cell;
    int cs;
    __device__ __forceinline__ double C(int f) const { return cell[f * cs]; }
    double si, ws, umu;
    int shadowmask;
    double Tg, Tc, sdepc, sdepg, sden, alb, ialb;                  // snowpoint2; ialb = 1 / alb
};
struct MicroOut { double Tz, tleaf, rh, uz, Rbdown, Rddown, Rlwdn, Rdup, Rlwup; };

// sin(x) and cos(x) for x in [0, pi] (quadrant reduction with a two-part pi/2, the classic degree-13 / degree-14 kernels on
// [-pi/4, pi/4]: < 1 ulp); anything else goes to the device libm
__device__ __forceinline__ void sincos_0pi(double x, double& sn, double& cs) {
    if (x >= 0.0 && x <= 3.1416) {
        const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);          // 2/pi: k = 0, 1, 2
        double r = fma(-k, 0x1.921fb54400000p+0, x);                        // exact: pi/2's leading 33 bits
        r = fma(-k, 0x1.0b4611a626331p-34, r);
        const double z = r * r;
        const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                          z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
        const double sr = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
        const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                          z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
        const double hz = 0.5 * z, w = 1.0 - hz;
        const double cr = w + (((1.0 - w) - hz) + z * pc);
        const bool odd = (k == 1.0);
        const double a = odd ? cr : sr, b = odd ? sr : cr;
        sn = (k == 2.0) ? -a : a;
        cs = (k == 0.0) ? b : -b;
    } else {
        // outside [0, pi]: not reachable from rh_canopy (0 < z < h) except with NaN operands, which propagate through the same
        // reduction (round 5: the device libm's sin / cos — Payne-Hanek reduction, the registers of a second kernel — are gone
        // from the snow-microclimate kernels); moderate arguments reduce by quadrants, anything beyond 1e9 is given up as NaN
        const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);
        double r = fma(-k, 0x1.921fb54400000p+0, x);
        r = fma(-k, 0x1.0b4611a626331p-34, r);
        if (!(fabs(x) < 1e9)) r = __longlong_as_double(0x7FF8000000000000LL);
        const double z = r * r;
        const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                          z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
        const double sr = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
        const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                          z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
        const double hz = 0.5 * z, w = 1.0 - hz;
        const double cr = w + (((1.0 - w) - hz) + z * pc);
        const int q = (int)__builtin_fmod(__builtin_fmod(k, 4.0) + 4.0, 4.0);
        sn = q == 0 ? sr : q == 1 ? cr : q == 2 ? -sr : -cr;
        cs = q == 0 ? cr : q == 1 ? -sr : q == 2 ? -cr : sr;
    }
}
__device__ __forceinline__ double rh_canopy(double uf, double h, double ih, double d, double z) {  // cpp:1365-1380; ih = 1/h
    const double a2 = 0.4 * (1.0 - (d * ih)) * (1.0 / (1.25 * 1.25));
    double inth = 4.293251 * h;
    if (z != h) {
        double sn, c1;
        sincos_0pi((kPi * z) * ih, sn, c1);
        c1 = c1 + 1;
        const double ic1 = gdiv(1.0, c1);                          // c1 = 0 only at z = h
        const double t = sn * ic1;                                 // tan(pi z / 2h)
        // sqrt(5) and pow(5, 1.5), correctly rounded
        inth = (2.0 * h * (48 * atan(0x1.1e3779b97f4a8p+1 * t) * (1.0 / 0x1.65c55827df1d2p+3) +
                           gdiv(32.0 * sn, c1 * (25.0 * (t * t) + 5.0)))) * (1.0 / kPi);
    }
    const double mu = gdiv(1.0, (a2 * h) * uf);                    // uf / (a2 h) / uf^2
    double r = inth * mu;
    if (r < 0.001) r = 0.001;
    return r;
}
struct BelowK { double Kg, Kh, Kc, iKc, iKs; };   // TVbelow's diffusivities (cpp:1385-1390): shared by T and e
__device__ __forceinline__ BelowK below_k(double z, double d, double h, double uf) {
    const double ih = gdiv(1.0, h);
    const double Rc = rh_canopy(uf, h, ih, d, h);
    const double rz = rh_canopy(uf, h, ih, d, z);
    BelowK k;
    k.iKc = Rc * ih;
    k.Kc = h * gdiv(1.0, Rc);
    k.Kg = gdiv(1.0, rz * z);
    k.Kh = gdiv(1.0, (Rc - rz) * (h - z));                         // Rc = rz: NaN here, inf / inf = NaN in the reference
    k.iKs = gdiv(1.0, k.Kg + k.Kh + k.Kc);
    return k;
}
__device__ __forceinline__ double tv_below(const BelowK& k, double lnpai, double leafden, double Flux, double Fluxz,
                                           double SH, double SG, double mxnear) {   // cpp:1391-1409
    const double SC = SH + Flux * k.iKc;
    const double farg = (k.Kg * SG + k.Kh * SH + k.Kc * SC) * k.iKs;
    double near = (3.047519 + 0.128642 * lnpai) * (Fluxz * leafden);
    if (fabs(near) > mxnear) near = near > 0.0 ? mxnear : -mxnear;
    if (isnan(near)) near = 0;
    return near + farg;
}
struct AboveTV { double Tz, ez; };
// TVabove (cpp:1298-1313, surfwet = 1) with the two logarithms handed in: Lz5 = log((z - d) / zh), Lref5 = log((zref - d) / zh), zh = 0.2 zm
__device__ __forceinline__ AboveTV tv_above_l(double z, double d, double zm, double Lz5, double Lref5, double T0, double tc,
                                              double ea) {
    const double estl = svp(T0);
    AboveTV o;
    if (z > (d + 0.2 * zm)) {
        const double lnr = gdiv(Lz5, Lref5);
        o.Tz = tc + (T0 - tc) * (1 - lnr);
        o.ez = ea + (estl - ea) * (1 - lnr);
    } else {
        o.Tz = T0;
        o.ez = ea + (estl - ea);
    }
    return o;
}
__device__ __forceinline__ double max4(double a, double b, double c, double d) {  // std::max({..})
    double m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    if (m < d) m = d;
    return m;
}
__device__ __forceinline__ double min4(double a, double b, double c, double d) {
    double m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    if (d < m) m = d;
    return m;
}

// `sink(i, v)` (i = the output's index in the reference's list: 4 wind speed, 5 Rdirdown, 6 Rdifdown, 7 Rlwdown, 8 Rswup,
// 9 Rlwup) is offered each of these six the moment it is final; a sink that returns true has taken it (k_microsnow_ring
// stores it into the ring there and then: the six values are not carried through the rest of the function — ten registers),
// the default leaves it in the returned struct.
struct MicroNoSink { __device__ __forceinline__ bool operator()(int, double) const { return false; } };
template <class Sink = MicroNoSink>
__device__ __forceinline__ MicroOut micro_above(const MicroIn& q, const MicroMet& mm, const SunT& sun, Sink sink = Sink()) {
    MicroOut out;
    double reqhgt = q.reqhgt;
    if (reqhgt == 0.0) reqhgt = 0.001;
    const double es = mm.es, ea = mm.ea, tdew = mm.tdew;
    double hgts = q.C(MQ_HGT) - q.sdepg;
    if (hgts < 0.0) hgts = 0.0;
    const double ipk = mm.ipk;
    double pais = 0.0, ihgts = 0.0, frac = 0.0;
    if (hgts > 0.0) {                                            // windtiCpp cpp:1179-1187
        frac = hgts * q.C(MQ_IHGT);                                    // hgt > sdepg >= 0 here
        ihgts = gdiv(1.0, hgts);
        pais = q.C(MQ_PAI) * frac;
    }
    // Roughness and wind (windtiCpp, windCpp cpp:1189-1218, gturbCpp) read nothing the radiation block makes and the radiation
    // block reads nothing of theirs: they are evaluated BEHIND it (same operations, same values) — five doubles fewer alive
    // through the two-stream algebra, the fullest stretch of the function —, once for both canopy classes (below).
    // One logarithm per height: Lref = log((zref - d) / zm) serves the friction velocity, and — as Lref + log 5 = log((zref - d) /
    // (0.2 zm)) — gturbCpp's conductance and the reference height of the temperature / vapour profile; Lz = log((z - d) / zm) of
    // the wind profile's own height z (reqhgt above the canopy, the canopy top inside it) is that profile's too.  (The reference
    // takes three more logarithms of the same quotients; its `0.2 zm + d - d` differs from 0.2 zm in the last bits only.)
    double d = 0.0, zm = 1e-5, uf, uz, gHa, Lref5, Lz5 = 0.0;
    constexpr double kLn5 = 0x1.9c041f7ed8d33p+0;
    auto wind = [&]() {
        double wa = 0.0;
        if (hgts > 0.0) {
            d = zeroplane(hgts, pais);
            zm = roughlen0(hgts, pais, d);
            if (zm < 1e-6) zm = 1e-6;
            wa = pais * ihgts;
        }
        double ws = q.ws;
        if (isnan(ws)) ws = 1.0;
        if (ws < 0.05) ws = 0.05;
        const double izm = gdiv(1.0, zm);
        const double Lref = glog((q.zref - d) * izm);
        Lref5 = Lref + kLn5;
        uf = gdiv(kKa * q.u2, Lref) * q.umu * ws;
        if (uf < 0.001) uf = 0.001;
        uz = uf;
        if (reqhgt > 0) {
            if (reqhgt >= hgts) {
                const double Lz = glog((reqhgt - d) * izm);
                Lz5 = Lz + kLn5;
                uz = (uf * (1.0 / kKa)) * Lz;
            } else {
                const double Lz = glog((hgts - d) * izm);
                Lz5 = Lz + kLn5;
                double uh = (uf * (1.0 / kKa)) * Lz;
                if (uh < uf) uh = uf;
                double Be = gdiv(uf, uh);
                if (Be < 0.001) Be = 0.001;
                const double Lm = 2 * (Be * Be * Be) * gdiv(1.0, 0.25 * wa);
                uz = uh * gexp(gdiv(Be * (reqhgt - hgts), Lm));
            }
            if (uz > q.u2) uz = q.u2;
        }
        gHa = gdiv(kKa * 43 * uf, Lref5);                       // gturbCpp(.., 43, 0, 0.0001)
        if (gHa < 0.0001) gHa = 0.0001;
        if (!sink(4, uz)) out.uz = uz;
    };
    // The radiation of both classes first, then roughness and wind ONCE for every lane, then the classes' temperatures: a wave
    // that holds cells of both classes (vegetation above the pack beside buried vegetation) runs the wind block — 250 of the
    // loop's 3 150 instructions — once, not once per class (round 5: -1 % on configs[4]'s snow-day stage, whose waves are
    // mostly of one class).
    double ez;
    const bool above = reqhgt >= hgts;
    double paias = 0.0, radLsw = 0.0;
    if (above) {                                                 // above the canopy, cpp:4768-4798
        {
            double Rbdown = 0.0, Rddown = 0.0, Rdup = 0.0;
            if (q.Rsw > 0.0) {
                Rddown = q.Rdif * q.C(MQ_SVFA);
                if (q.si > 0.0 && q.shadowmask > 0) {
                    Rbdown = gdiv(q.Rsw - q.Rdif, q.si);
                    if (Rbdown > 1352.0) Rbdown = 1352.0;
                    Rdup = q.alb * q.Rsw * q.C(MQ_SVFA);
                } else {
                    Rdup = q.alb * q.Rdif * q.C(MQ_SVFA);
                }
            }
            const double lwdn = q.C(MQ_SVFA) * q.Rlw, lwup = q.C(MQ_SVFA) * 0.97 * kSb * rad4(q.Tc);
            if (!sink(5, Rbdown)) out.Rbdown = Rbdown;
            if (!sink(6, Rddown)) out.Rddown = Rddown;
            if (!sink(7, lwdn)) out.Rlwdn = lwdn;
            if (!sink(8, Rdup)) out.Rdup = Rdup;
            if (!sink(9, lwup)) out.Rlwup = lwup;
        }
    } else {                                                     // inside the canopy, cpp:4799-4857
        paias = q.C(MQ_PAIA) * frac;                                   // (hgts > reqhgt > 0 here)
        double zi = 0.0;
        if (q.sdepg > 0.0) zi = ((q.sdepc - q.sdepg) * q.sden) * (ihgts * (1.0 / 1000.0));
        double ltras = q.C(MQ_LTRA) * gexp(-10.1 * zi);
        if ((ltras + q.alb) > 0.999) ltras = 0.999 - q.alb;
        // clumps = clump^(pais / pai) and its own powers through ONE logarithm: log(clumps) = (pais / pai) log(clump)
        const double ipais = gdiv(1.0, pais);
        double clumps = q.C(MQ_CLUMP), lncl = 0.0;
        if (q.C(MQ_CLUMP) > 0.0) {
            lncl = (pais * q.C(MQ_IPAI)) * q.C(MQ_LNCLUMP);
            clumps = gexp(lncl);
        }
        const double i1c = gdiv(1.0, 1.0 - clumps);
        double pait = pais;
        if (q.C(MQ_CLUMP) > 0.0) pait = pais * i1c;
        // twostreamdif (cpp:1034-1084) with lref = gref = snow albedo
        const double pait2 = pais * i1c;
        const TsDif f = ts_dif(pait2, q.alb, ltras, q.alb, q.ialb);
        double gi = 0.0, giu = 0.0, lngi = 0.0;
        if (clumps > 0.0) {
            lngi = (paias * ipais) * lncl;
            gi = gexp(lngi);
            giu = gexp(((pais - paias) * ipais) * lncl);
        }
        if (gi > 0.99) { gi = 0.99; lngi = -0x1.495453e6fd4bcp-7; }   // log(0.99)
        if (giu > 0.99) giu = 0.99;
        const double trd = gi * gi, trdn = clumps * clumps, trdu = giu * giu;
        const double paiaa = gdiv(paias, 1.0 - gi);
        const double amx = q.alb;                                // max(gref, lref), both the albedo
        const double eh_a = gexp(-f.h * paiaa), eH_a = gdiv(1.0, eh_a);
        double Rdup_z = (1.0 - trdu * trdn) * (f.p1 * eh_a + f.p2 * eH_a) + trdu * trdn * q.alb;
        Rdup_z = clamp01(Rdup_z);
        const double Rddn_z = clamp01((1.0 - trd) * (f.p3 * eh_a + f.p4 * eH_a) + trd);
        // the direct-beam coefficients use twostreamdifCpp(pait, ..) (cpp:4814) but tir's D1, D2 come from pait2;
        // the reference passes tspdif's own (cpp:4817), so they are recomputed when the two differ
        const TsDif fd = (pait == pait2) ? f : ts_dif(pait, q.alb, ltras, q.alb, q.ialb);
        const CanK kp = cank1(sun.kx, sun.kcos, q.si);
        const TsDir dr = ts_dir(pait, fd, q.alb, kp.kd);
        // twostreamCpp (cpp:1086-1178)
        double Rbdown = 0.0, Rddown = 0.0, Rdup = 0.0;
        if (q.Rsw > 0.0) {
            const double cosz = sun.cosz;
            const double icosz = gdiv(1.0, cosz);
            if (pais > 0.0) {
                double trbn = clumps > 0.0 ? gexp(kp.Kc * lncl) : gpow0(clumps, kp.Kc);
                if (trbn > 0.999) trbn = 0.999;
                if (trbn < 0.0) trbn = 0.0;
                double trb = gi > 0.0 ? gexp(kp.Kc * lngi) : gpow0(gi, kp.Kc);
                if (trb > 0.999) trb = 0.999;
                if (trb < 0.0) trb = 0.0;
                const double ek_a = gexp(-kp.kd * paiaa);
                const double isig = dr.isig;
                double Rdbup_z = (1.0 - trdu * trbn) * ((dr.p5 * -isig) * ek_a + dr.p6 * eh_a + dr.p7 * eH_a) +
                                 trdu * trbn * q.alb;
                if (Rdbup_z > amx) Rdbup_z = amx;
                if (Rdbup_z < 0.0) Rdbup_z = 0.0;
                double Rdbdn_z = (1.0 - trb) * ((dr.p8 * isig) * ek_a + dr.p9 * eh_a + dr.p10 * eH_a);
                if (Rdbdn_z > amx) Rdbdn_z = amx;
                if (Rdbdn_z < 0.0) Rdbdn_z = 0.0;
                double Rbeam = (q.Rsw - q.Rdif) * icosz;
                if (Rbeam > 1352.0) Rbeam = 1352.0;
                const double Rb = Rbeam * cosz;
                Rbdown = (trb + (1.0 - trb) * ek_a) * Rbeam;
                Rddown = Rddn_z * q.Rdif * q.C(MQ_SVFA) + Rdbdn_z * Rb;
                Rdup = Rdup_z * q.Rdif * q.C(MQ_SVFA) + Rdbup_z * Rb;
                radLsw = 0.5 * (1.0 - f.om) * (Rddown + Rdup + kp.k * cosz * Rbdown);
            } else {
                Rbdown = (q.Rsw - q.Rdif) * icosz;
                Rddown = q.Rdif * q.C(MQ_SVFA);
                Rdup = q.alb * (q.Rdif * q.C(MQ_SVFA) + (q.Rsw - q.Rdif));
            }
        }
        if (q.shadowmask == 0) Rbdown = 0.0;
        if (!sink(5, Rbdown)) out.Rbdown = Rbdown;
        if (!sink(6, Rddown)) out.Rddown = Rddown;
        if (!sink(8, Rdup)) out.Rdup = Rdup;
    }
    wind();
    if (above) {
        const AboveTV tv = tv_above_l(reqhgt, d, zm, Lz5, Lref5, q.Tc, q.tc, ea);
        out.Tz = tv.Tz;
        out.tleaf = q.Tc;
        ez = tv.ez;
    } else {
        // leaftemp (cpp:1333-1364) with gsmax = 999.999: gV = gh
        const double lwcan = 0.97 * kSb * rad4(q.Tc);
        const double lwgro = 0.97 * kSb * rad4(q.Tg);
        const double eg = gexp(-(pais - paias)), eaa = gexp(-paias);
        const double lwup = eg * lwgro + (1 - eg) * lwcan;
        const double lwdn = eaa * q.Rlw + (1 - eaa) * lwcan;
        const double leafabs = radLsw + 0.97 * 0.5 * (lwup + lwdn);
        if (!sink(7, lwdn)) out.Rlwdn = lwdn;
        if (!sink(9, lwup)) out.Rlwup = lwup;
        const double ileafd = q.C(MQ_ILEAFD);
        double gh = 0.135 * gsqrt(uz * ileafd) * 1.4;
        {   // mincondCpp(leafabs, 999.99, Tcan, leafd) cpp:1316-1331
            const double Rnet = leafabs - lwcan;
            // Hf = -1 / (1 + exp(2 - 1.09767 * rs^0.2672778)) with rs = 1 / 999.99: a constant of the model
            const double Hf = -0x1.1be70d7011323p-3;
            double gmin = 0.0463 * gpow0(fabs(Hf * Rnet) * ileafd, 0.2);
            if (gmin < 0.05) gmin = 0.05;
            if (gh < gmin) gh = gmin;
        }
        // PenmanMonteith2Cpp (cpp:1220-1247), G = 0, surfwet = 1
        const double gHr = gh + mm.gHrad;
        const double mpm = mm.la_pm * (gh * ipk);
        double dT = gdiv(leafabs - mm.Rem - mpm * (es - ea), 29.3 * gHr + mpm * mm.De);
        if (dT > mm.dTmx) dT = mm.dTmx;
        if (dT > 80.0) dT = 80.0;
        double tleaf = dT + q.tc;
        if (tleaf < tdew) tleaf = tdew;
        const double Hl = 29.3 * gh * (tleaf - q.tc);
        const double estl = svp(tleaf);
        const double Ll = mpm * (estl - ea);
        out.tleaf = tleaf;
        // Lagrangian below-canopy profile (cpp:4827-4851)
        const BelowK bk = below_k(reqhgt, d, hgts, uf);
        const double lnpai = glog(pais);
        const double H = 29.3 * gHa * (q.Tc - q.tc);
        const double fr = 1.0 - eg * eaa;                         // exp(-pais) = exp(-(pais - paias)) exp(-paias)
        const AboveTV tv = tv_above_l(hgts, d, zm, Lz5, Lref5, q.Tc, q.tc, ea);
        out.Tz = tv_below(bk, lnpai, q.C(MQ_LEAFDEN), H * fr, Hl, tv.Tz * 29.3 * 43.0, q.Tg * 29.3 * 43.0,
                          fabs(tleaf - tv.Tz) * 29.3 * 43.0) * (1.0 / (29.3 * 43));
        const double la = mm.lat0;
        const double mmg = la * (gHa * ipk);
        const double mu = la * (43 * ipk);
        ez = tv_below(bk, lnpai, q.C(MQ_LEAFDEN), (mmg * (es - ea)) * fr, Ll, tv.ez * mu, svp(q.Tg) * mu,
                      fabs(estl - tv.ez) * mu) * gdiv(1.0, mu);
    }
    out.rh = gdiv(ez, svp(out.Tz)) * 100.0;
    if (out.rh > 100.0) out.rh = 100.0;
    const double tmx = max4(out.tleaf, q.tc, q.Tg, q.Tc) + 2.0;
    const double tmn = min4(out.tleaf, q.tc, q.Tg, q.Tc) - 2.0;
    if (out.Tz > tmx) out.Tz = tmx;
    if (out.Tz < tmn) out.Tz = tmn;
    return out;
}

// belowpointsnow (cpp:4868-4891)
__device__ __forceinline__ double micro_below(double reqhgt, double meanD, double tg, double Tzd, double Tza,
                                              double hiy) {
    const double nb = -118.35 * reqhgt / meanD;
    double Tz = tg;
    if (nb > 1.0) {
        if (nb <= 24.0) {
            const double w1 = 1.0 / nb, w2 = nb / 24.0;
            const double wgt = w1 / (w1 + w2);
            Tz = wgt * tg + (1 - wgt) * Tzd;
        } else if (nb <= hiy) {
            const double w1 = 24.0 / nb, w2 = nb / hiy;
            const double wgt = w1 / (w1 + w2);
            Tz = wgt * Tzd + (1 - wgt) * Tza;
        } else {
            Tz = Tza;
        }
    }
    return Tz;
}

}  // namespace snow
}  // namespace mcf
