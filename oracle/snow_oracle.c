/*
 * snow_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mcf_oracle.c).
 *
 * Plain-C restatement of the reference's snow branch (src/microclimfCpp.cpp:3713-5214): the
 * snowpack energy / mass balance (snowoneB and its callers pointmodelsnow, gridmodelsnow1/2)
 * and the snow microclimate (snowabovepoint, belowpointsnow, gridmicrosnow1/2).  Included by
 * oracle_unit.c after mcf_oracle.c and pointmodel.c, whose static helpers it reuses.
 *
 * PARITY PINNING: the reference's only test of this arithmetic, tests/testthat/
 * test-pointmodelsnow.R, drives pointmodelsnow -> snowoneB -> radoneB / canopysnowintCpp /
 * snowalbCpp / GFluxCppsnow; it is replayed against orc_pointmodelsnow by
 * oracle/replay_reference_tests.py (all of its assertions hold).  The grid functions share
 * snowoneB with it.  snowabovepoint / belowpointsnow have no reference test: "parity unpinned"
 * beyond what they share with the pinned main path (windCpp, twostreamCpp, leaftemp, TVabove,
 * TVbelow).
 */

/* cpp:3713-3739 canopysnowintCpp (Sh = 6.2) */
static double canopysnowint(double hgt, double pai, double uf, double prec, double tc, double Li) {
    const double Sh = 6.2;
    if (hgt < 0.001) hgt = 0.001;
    if (pai < 0.001) pai = 0.001;
    double Be = sqrt(0.003 + (0.2 * pai) / 2.0);
    double uh = uf / Be;
    double a = pai / hgt;
    double Lc = pow(0.25 * a, -1.0);
    double Lm = 2.0 * pow(Be, 3.0) * Lc;
    double k1 = Be / Lm;
    double uzm = (uh / (hgt * k1)) * (1 - exp(-k1 * hgt));
    if (uzm < uf) uzm = uf;
    double rhos = 67.92 + 51.25 * exp(tc / 2.59);
    double S = Sh * (0.26 + 46 / rhos);
    double Lstr = S * pai;
    double Z = atan(uzm / 0.8);
    double kc = 1.0 / (2.0 * cos(Z));
    double Cp = 1.0 - exp(-kc * pai);
    double k2 = Cp / Lstr;
    double I1 = (Lstr - Li) * (1.0 - exp(-k2 * prec));
    double cis = I1 * 0.678;
    if (cis > prec) cis = prec;
    return cis;
}

/* cpp:3741-3749 snowdenp; the enum is include/mcf.h's MCF_SNOWENV_* */
static void snowdenp(int snowenv, double sdp[4]) {
    static const double tab[5][4] = {
        {0.5975, 0.2237, 0.0012, 0.0038}, /* default ("Alpine" and anything unknown) */
        {0.5979, 0.2578, 0.001, 0.0038},  /* Maritime */
        {0.594, 0.2332, 0.0016, 0.0031},  /* Prairie  */
        {0.363, 0.2425, 0.0029, 0.0049},  /* Tundra   */
        {0.217, 0.217, 0.0, 0.0}};        /* Taiga    */
    if (snowenv < 0 || snowenv > 4) snowenv = 0;
    for (int i = 0; i < 4; ++i) sdp[i] = tab[snowenv][i];
}

/* cpp:3752-3771 snowalbCpp.  `hs[i] / 24` is an INTEGER division in the reference, so the
 * argument of the logarithm is the whole number of days since snowfall (log(0) = -inf on the
 * first day, capped to 0.95). */
static void snowalb(const double *prec, int64_t stride, int tsteps, double *alb) {
    int hs = 0;
    for (int i = 0; i < tsteps; ++i) {
        if (i > 0) {
            if (prec[(int64_t)i * stride] > 0) hs = 0;
            else hs = hs + 1;
        }
        alb[i] = (-9.8740 * log((double)(hs / 24)) + 78.3434) / 100.0;
        if (alb[i] > 0.95) alb[i] = 0.95;
        if (alb[i] < 0.1) alb[i] = 0.1;
    }
}

void orc_snowalb(const double *prec, int tsteps, double *alb) { snowalb(prec, 1, tsteps, alb); }

/* hdr:159-245 structs of the snow model */
typedef struct { int year, month, day; double hour; } obspoint_t;
typedef struct { double tc, ea, pk, u2, Rsw, Rdif, Rlw, prec, Tci, te; } climpoint_t;
typedef struct { double pai, hgt, ltra, clump; } vegpoint_t;
typedef struct { double sdenc, sdeng, sdepc, sdepg, snowagec, snowageg, alb; } snowpoint_t;
typedef struct { double slope, aspect, lat, lon, zref, psim, psih, G; } otherpoint_t;
typedef struct { double RabsC, RswabsG, RlwabsG, tr; } snowrad_t;
typedef struct {
    double Tc, mSc, mMc, mRc, Tg, mSg, mMg, mRg, cis, uf, RswabsG, RlwabsG, tr, gHa, Tcp, snowagec, snowageg,
        sdenc, sdeng, sdepc, sdepg, pai, hgt;
} snowmod_t;

/* cpp:3773-3833 radoneB */
static snowrad_t radoneB(obspoint_t obstime, climpoint_t clim, vegpoint_t vegp, snowpoint_t snow,
                         otherpoint_t other) {
    snowrad_t out = {0.0, 0.0, 0.0, 0.0};
    double RlwabsC = 0.97 * clim.Rlw;
    out.RlwabsG = RlwabsC;
    double cld = vegp.clump * vegp.clump;
    double pait = vegp.pai / (1.0 - vegp.clump);
    out.tr = (1.0 - cld) * exp(-pait) + cld;
    if (vegp.hgt > 0.0) {
        double Rsky = out.tr * clim.Rlw;
        double Rcan = (1.0 - out.tr) * 0.97 * SB * radem(clim.Tci);
        out.RlwabsG = 0.97 * (Rsky + Rcan);
    }
    out.RabsC = RlwabsC;
    out.RswabsG = 0.0;
    if (clim.Rsw > 0.0) {
        orc_solmodel solp = orc_solposition(other.lat, other.lon, obstime.year, obstime.month, obstime.day,
                                            obstime.hour);
        double si = orc_solarindex(other.slope, other.aspect, solp.zend, solp.azid, 0);
        if (solp.zend > 90.0) solp.zend = 90.0;
        if (si < 0.0) si = 0.0;
        double cosz = cos(solp.zenr);
        double Rbeam = (clim.Rsw - clim.Rdif) / cosz;
        if (Rbeam > 1352.2) Rbeam = 1352.2;
        double RswabsC = (1.0 - snow.alb) * (clim.Rdif + Rbeam * cosz);
        out.RabsC = RswabsC + RlwabsC;
        out.RswabsG = RswabsC;
        if (vegp.hgt > 0.0) {
            if ((snow.alb + vegp.ltra) > 0.999) vegp.ltra = 0.999 - snow.alb;
            tsdif_t tspdif = twostreamdif_params(pait, 1.0, snow.alb, vegp.ltra, snow.alb);
            orc_kstruct kp = orc_cank(solp.zenr, 1.0, si);
            tsdir_t tspdir = twostreamdir_params(pait, tspdif.om, tspdif.a, tspdif.gma, tspdif.J, tspdif.del,
                                                 tspdif.h, snow.alb, kp.kd, tspdif.u1, tspdif.S1, tspdif.D1,
                                                 tspdif.D2);
            double clb = pow(vegp.clump, kp.Kc);
            double Rddm = (1.0 - cld) * (tspdif.p3 * exp(-tspdif.h * pait) + tspdif.p4 * exp(tspdif.h * pait)) +
                          cld;
            if (Rddm > 1.0) Rddm = 1.0;
            if (Rddm < 0.0) Rddm = 0.0;
            double Rdbm = (1.0 - clb) * ((tspdir.p8 / tspdir.sig) * exp(-kp.kd * pait) +
                                         tspdir.p9 * exp(-tspdif.h * pait) + tspdir.p10 * exp(tspdif.h * pait));
            if (Rdbm > 1.0) Rdbm = 1.0;
            if (Rdbm < 0.0) Rdbm = 0.0;
            double Rbgm = (1.0 - clb) * exp(-kp.kd * pait) + clb;
            if (Rbgm > 1.0) Rbgm = 1.0;
            if (Rbgm < 0.0) Rbgm = 0.0;
            double RdifG = (1.0 - snow.alb) * (Rdbm * Rbeam * cosz) + Rddm * clim.Rdif;
            double RdirG = (1.0 - snow.alb) * (Rbgm * Rbeam * 0.5);
            out.RswabsG = RdifG + RdirG;
        }
    }
    return out;
}

/* cpp:3835-3972 snowoneB */
static snowmod_t snowoneB(obspoint_t obstime, climpoint_t clim, vegpoint_t vegp, snowpoint_t snow,
                          otherpoint_t other, const double sdp[4], double umu) {
    snowmod_t out;
    memset(&out, 0, sizeof out);
    double pai = 0.0;
    if (vegp.hgt > snow.sdepg) pai = vegp.pai * (vegp.hgt - snow.sdepg) / vegp.hgt;
    double hgt = vegp.hgt - snow.sdepg;
    if (hgt < 0.0) hgt = 0.0;
    double zi = 0.0;
    if (snow.sdepg > 0.0 && hgt > 0.0) zi = ((snow.sdepc - snow.sdepg) * snow.sdenc) / (hgt * 1000.0);
    double ltra = vegp.ltra * exp(-10.1 * zi);
    vegp.hgt = hgt;
    vegp.ltra = ltra;
    vegp.pai = pai;
    snowrad_t rad = radoneB(obstime, clim, vegp, snow, other);
    double RabsG = rad.RswabsG + rad.RlwabsG;
    double d = 0.0;
    double zm = 0.005;
    if (vegp.hgt > 0.0) {
        d = orc_zeroplanedis(hgt, pai);
        zm = orc_roughlength(hgt, pai, d, other.psih);
    }
    if (zm < 0.0009) zm = 0.0009;
    out.hgt = hgt;
    out.pai = pai;
    out.uf = (KA * clim.u2) / (log((other.zref - d) / zm) + other.psim);
    out.uf = out.uf * umu;
    double ph = phair(clim.tc, clim.pk);
    out.gHa = gturb(out.uf, d, zm, other.zref, ph, other.psih, 0.03);
    out.Tc = penman(rad.RabsC, out.gHa, out.gHa, clim.tc, clim.te, clim.pk, clim.ea, 0.97, other.G, 1.0);
    out.Tg = penman(RabsG, out.gHa, out.gHa, clim.tc, clim.te, clim.pk, clim.ea, 0.97, other.G, 1.0);
    double tdew = dewpoint_cpp(clim.ea);
    if (out.Tc < tdew) out.Tc = tdew;
    if (out.Tg < tdew) out.Tg = tdew;
    /* canopy + ground pack */
    double la;
    if (out.Tc < 0.0) la = 51078.69 - 4.338 * out.Tc - 0.06367 * out.Tc * out.Tc;
    else la = 45068.7 - 42.8428 * out.Tc;
    double L = la * (out.gHa / clim.pk) * (orc_satvap(out.Tc) - clim.ea);
    la = la / 0.018015;
    out.mSc = (L / la) * 3.6;
    double Tcp = out.Tc;
    out.mMc = 0.0;
    if (out.Tc > 0.0) {
        double S = snow.sdepc * (snow.sdenc / 1000);
        double Fm = 583.3 * out.Tc * S;
        out.mMc = (Fm / 334000.0) * 3.6;
        if (snow.sdepc > 0.0) out.Tc = 0.0;
    }
    out.mRc = 0.0;
    if (clim.tc > 0.0) out.mRc = 0.0125 * clim.tc * clim.prec / 1000;
    /* ground pack */
    if (out.Tg < 0.0) la = 51078.69 - 4.338 * out.Tg - 0.06367 * out.Tg * out.Tg;
    else la = 45068.7 - 42.8428 * out.Tg;
    double mu = exp(-vegp.pai);
    if (mu > 1.0) mu = 1.0;
    L = la * (out.gHa / clim.pk) * (orc_satvap(out.Tg) - clim.ea) * mu;
    la = la / 0.018015;
    out.mSg = (L / la) * 3.6;
    out.mMg = 0.0;
    if (out.Tg > 0.0) {
        double S = snow.sdepg * (snow.sdeng / 1000.0);
        double Fm = 583.3 * out.Tg * S;
        out.mMg = (Fm / 334000.0) * 3.6;
        if (snow.sdepg > 0.0) out.Tg = 0.0;
    }
    double Li = 0.0;
    if (snow.sdepc > 0.0) {
        double wgtg = snow.sdepg / snow.sdepc;
        if (wgtg < 0.0) wgtg = 0.0;
        if (wgtg > 1.0) wgtg = 1.0;
        double sdencc = wgtg * snow.sdeng + (1.0 - wgtg) * snow.sdenc;
        Li = (snow.sdepc - snow.sdepg) * sdencc;
    }
    if (Li < 0.0) Li = 0.0;
    out.cis = canopysnowint(vegp.hgt, vegp.pai, out.uf, clim.prec, clim.tc, Li);
    if (out.cis > clim.prec) out.cis = clim.prec;
    out.mRg = 0.0;
    if (clim.tc > 0.0) out.mRg = 0.0125 * clim.tc * (clim.prec - out.cis) / 1000.0;
    double snowc = clim.prec;
    double snowg = clim.prec - out.cis;
    if (clim.tc > 2.0) {
        snowc = 0.0;
        snowg = 0.0;
    }
    double swec = snowc / 1000.0 - out.mSc - out.mMc - out.mRc;
    double sweg = snowg / 1000.0 - out.mSg - out.mMg - out.mRg;
    out.snowagec = snow.snowagec + 1.0;
    out.snowageg = snow.snowageg + 1.0;
    out.sdenc = ((sdp[0] - sdp[1]) * (1.0 - exp(-sdp[2] * snow.sdepc / 100.0 - sdp[3] * out.snowagec / 24.0)) +
                 sdp[1]) * 1000.0;
    out.sdeng = ((sdp[0] - sdp[1]) * (1.0 - exp(-sdp[2] * snow.sdepg / 100.0 - sdp[3] * out.snowageg / 24.0)) +
                 sdp[1]) * 1000.0;
    out.sdepc = snow.sdepc + (swec * 1000.0) / out.sdenc;
    out.sdepg = snow.sdepg + (sweg * 1000.0) / out.sdeng;
    if (out.sdepc < 0.0) {
        out.sdepc = 0.0;
        out.snowagec = 0.0;
    }
    if (out.sdepg < 0.0) {
        out.sdepg = 0.0;
        out.snowageg = 0.0;
    }
    out.RswabsG = rad.RswabsG;
    out.RlwabsG = rad.RlwabsG;
    out.tr = rad.tr;
    out.Tcp = Tcp;
    return out;
}

/* cpp:3974-3997 GFluxCppsnow */
static void gfluxsnow(const double *snowt, const double *snowden, int tsteps, double *G) {
    double *Gmu = (double *)calloc((size_t)tsteps + 1, sizeof(double));
    double *dT = (double *)calloc((size_t)tsteps + 1, sizeof(double));
    double *Td = (double *)calloc((size_t)tsteps + 1, sizeof(double));
    double *Gmud = (double *)calloc((size_t)tsteps + 1, sizeof(double));
    hourtoday(snowt, tsteps, 2, Td);
    for (int i = 0; i < tsteps; ++i) {
        double k = 0.0442 * exp(5.181 * snowden[i] / 1000);
        double kap = k / (snowden[i] * 2090);
        double DD = sqrt(2.0 * kap / OMDY);
        Gmu[i] = sqrt(2.0) * (k / DD) * 0.5;
        dT[i] = snowt[i] - Td[i];
    }
    ma_circ(Gmu, tsteps, 6, Gmud);
    ma_circ(dT, tsteps, 6, G);
    for (int i = 0; i < tsteps; ++i) G[i] = G[i] * Gmud[i] * 1.1171;
    free(Gmu); free(dT); free(Td); free(Gmud);
}

/* cpp:4000-4169 pointmodelsnow.  vegp = (pai, hgt, ltra, clump), other = (slope, aspect, lat, lon,
 * zref, isnowd, isnowa).  sdepc / sdepg have tsteps + 1 entries, everything else tsteps. */
int orc_pointmodelsnow(int tsteps, const int *year, const int *month, const int *day, const double *hour,
                       const double *tc, const double *rh, const double *pk, const double *Rsw,
                       const double *Rdif, const double *Rlw, const double *u2, const double *prec,
                       const double *vegp, const double *other, int snowenv, double tol, double maxiter,
                       orc_pointsnow_out *o) {
    size_t n = (size_t)tsteps + 1;
    double *ea = (double *)calloc(n, sizeof(double)), *te = (double *)calloc(n, sizeof(double));
    double *salb = (double *)calloc(n, sizeof(double)), *H = (double *)calloc(n, sizeof(double));
    double *psih = (double *)calloc(n, sizeof(double)), *psim = (double *)calloc(n, sizeof(double));
    double *phih = (double *)calloc(n, sizeof(double));
    double *Tco = (double *)calloc(n, sizeof(double)), *Tgo = (double *)calloc(n, sizeof(double));
    for (int i = 0; i < tsteps; ++i) {
        ea[i] = orc_satvap(tc[i]) * rh[i] / 100.0;
        te[i] = tc[i];
        phih[i] = 1.0;
    }
    double slope = other[0], aspect = other[1], lat = other[2], lon = other[3], zref = other[4];
    double isnowd = other[5], isnowa = other[6];
    snowalb(prec, 1, tsteps, salb);
    for (int i = 0; i < tsteps; ++i) {
        double Rabs = (1 - salb[i]) * Rsw[i] + 0.97 * Rlw[i];
        H[i] = 0.5 * Rabs;
    }
    double sdp[4];
    snowdenp(snowenv, sdp);
    for (int i = 0; i < tsteps; ++i) {
        o->sdenc[i] = ((sdp[0] - sdp[1]) * (1 - exp(-sdp[2] * isnowd / 100.0 - sdp[3] * 0)) + sdp[1]) * 1000.0;
        o->sdeng[i] = o->sdenc[i];
    }
    gfluxsnow(tc, o->sdenc, tsteps, o->G);
    for (int i = 0; i < tsteps; ++i) {
        o->Tc[i] = tc[i];
        o->Tg[i] = tc[i];
    }
    double tst = 100.0;
    int iter = 0;
    double mxdif = 0.0;
    obspoint_t obstimeo;
    climpoint_t climo;
    vegpoint_t vegpo;
    vegpo.pai = vegp[0]; vegpo.hgt = vegp[1]; vegpo.clump = vegp[3]; vegpo.ltra = vegp[2];
    otherpoint_t othero;
    memset(&othero, 0, sizeof othero);
    othero.slope = slope; othero.aspect = aspect; othero.lat = lat; othero.lon = lon; othero.zref = zref;
    snowpoint_t snowo;
    while (tst > tol) {
        int snowagec = (int)isnowa;
        int snowageg = (int)isnowa;
        o->sdepc[0] = isnowd;
        o->sdepg[0] = isnowd * 0.5;
        memcpy(Tco, o->Tc, (size_t)tsteps * sizeof(double));
        memcpy(Tgo, o->Tg, (size_t)tsteps * sizeof(double));
        mxdif = 0.0;
        for (int i = 0; i < tsteps; ++i) {
            obstimeo.year = year[i]; obstimeo.month = month[i]; obstimeo.day = day[i]; obstimeo.hour = hour[i];
            climo.tc = tc[i]; climo.ea = ea[i]; climo.pk = pk[i]; climo.u2 = u2[i]; climo.prec = prec[i];
            climo.Rsw = Rsw[i]; climo.Rdif = Rdif[i]; climo.Rlw = Rlw[i]; climo.Tci = o->Tc[i]; climo.te = te[i];
            othero.psim = psim[i]; othero.psih = psih[i]; othero.G = o->G[i];
            snowo.alb = salb[i]; snowo.sdenc = o->sdenc[i]; snowo.sdeng = o->sdeng[i];
            snowo.sdepc = o->sdepc[i]; snowo.sdepg = o->sdepg[i];
            snowo.snowagec = snowagec; snowo.snowageg = snowageg;
            snowmod_t smod = snowoneB(obstimeo, climo, vegpo, snowo, othero, sdp, 1.0);
            o->Tc[i] = smod.Tc;
            o->Tg[i] = smod.Tg;
            snowagec = (int)smod.snowagec;
            snowageg = (int)smod.snowageg;
            o->sdepc[i + 1] = smod.sdepc;
            o->sdepg[i + 1] = smod.sdepg;
            o->Tc[i] = 0.5 * Tco[i] + 0.5 * o->Tc[i];
            o->Tg[i] = 0.5 * Tgo[i] + 0.5 * o->Tg[i];
            double abs1 = fabs(o->Tc[i] - Tco[i]);
            double abs2 = fabs(o->Tg[i] - Tgo[i]);
            if (mxdif < abs1) mxdif = abs1;
            if (mxdif < abs2) mxdif = abs2;
            double cp = cpair(tc[i]);
            double ph = phair(tc[i], pk[i]);
            H[i] = cp * smod.gHa * (o->Tc[i] - tc[i]);
            double d = orc_zeroplanedis(smod.hgt, smod.pai);
            double zm = orc_roughlength(smod.hgt, smod.pai, d, psih[i]);
            if (zm < 0.001) zm = 0.001;
            double Tk = tc[i] + 273.15;
            if (fabs(H[i]) < 0.1) H[i] = 0.1;
            double LL = (ph * cp * pow(smod.uf, 3.0) * Tk) / (-KA * 9.81 * H[i]);
            psim[i] = dpsim(zm / LL) - dpsim((zref - d) / LL);
            psih[i] = dpsih((0.2 * zm) / LL) - dpsih((zref - d) / LL);
            phih[i] = dphih((zref - d) / LL);
            double Belim = 0.4 / sqrt(0.003 + (0.2 * smod.pai) / 2.0);
            double ln1 = log((zref - d) / zm);
            double ln2 = log((zref - d) / (0.2 * zm));
            if (psim[i] < -0.9 * ln1) psim[i] = -0.9 * ln1;
            if (psih[i] < -0.9 * ln2) psih[i] = -0.9 * ln2;
            if (psim[i] > 0.9 * ln1) psim[i] = 0.9 * ln1;
            if (psih[i] > 0.9 * ln2) psih[i] = 0.9 * ln2;
            if (psih[i] > 0.9 * Belim) psih[i] = 0.9 * Belim;
            o->RswabsG[i] = smod.RswabsG;
            o->RlwabsG[i] = smod.RlwabsG;
            o->tr[i] = smod.tr;
            double ufps = (0.4 * u2[i]) / log((zref - d) / zm);
            o->umu[i] = smod.uf / ufps;
            te[i] = (o->Tc[i] + tc[i]) / 2.0;
            o->sublmelt[i] = smod.mSc;
            o->tempmelt[i] = smod.mMc;
            o->rainmelt[i] = smod.mRc;
            o->sstemp[i] = smod.Tcp;
        }
        gfluxsnow(o->Tg, o->sdenc, tsteps, o->G);
        tst = mxdif;
        ++iter;
        if (iter > maxiter) tst = 0;
    }
    o->mxdif = mxdif;
    o->iters = iter;
    free(ea); free(te); free(salb); free(H); free(psih); free(psim); free(phih); free(Tco); free(Tgo);
    return 0;
}

static int is_na(double v) { return isnan(v); } /* Rcpp::NumericMatrix::is_na: any NaN */

/* cpp:4172-4423 gridmodelsnow1 (in->array_forcing == 0) and cpp:4426-4673 gridmodelsnow2 (== 1).
 * The two bodies differ in more than their indexing, and every difference is kept:
 *   1: per-timestep ea/te/Rnet, day statistics and albedo shared by all cells; horizon test
 *      `ha > tan((90 - zend) * torad)`; meltc starts from 0.0 (cpp:4333), meltg from NA.
 *   2: all of that per cell; horizon test `ha > tan(pi/2 - zenr)`; meltc AND meltg start from NA
 *      (cpp:4490-4491 are never zeroed), so they come back NA. */
int orc_gridmodelsnow(const mcf_snow_inputs *in, mcf_snowmodel_out *out) {
    const int64_t rows = in->rows, cols = in->cols;
    const int tsteps = (int)in->tsteps;
    const int64_t N = rows * cols;
    const int af = in->array_forcing;
    const int64_t fs = af ? N : 1; /* stride between time steps of a forcing array */
    const double NA = orc_na_real();
    const mcf_snow_climate *cl = &in->clim;
    const mcf_snow_pointm *pm = &in->pointm;
    double *arr3[5] = {out->Tc, out->Tg, out->sdepc, out->sdepg, out->sden};
    for (int v = 0; v < 5; ++v)
        if (arr3[v]) for (int64_t q = 0; q < N * tsteps; ++q) arr3[v][q] = NA;
    double *arr2[4] = {out->agec, out->ageg, out->meltc, out->meltg};
    for (int v = 0; v < 4; ++v)
        if (arr2[v]) for (int64_t q = 0; q < N; ++q) arr2[v][q] = NA;
    size_t n = (size_t)tsteps + 1;
    double *salb = (double *)calloc(n, sizeof(double)), *ea = (double *)calloc(n, sizeof(double));
    double *te = (double *)calloc(n, sizeof(double)), *Rnet = (double *)calloc(n, sizeof(double));
    double *Rmx = (double *)calloc(n, sizeof(double)), *Rmn = (double *)calloc(n, sizeof(double));
    double *Rswmn = (double *)calloc(n, sizeof(double)), *Rlwmn = (double *)calloc(n, sizeof(double));
    double *Rswmx = (double *)calloc(n, sizeof(double)), *Rlwmx = (double *)calloc(n, sizeof(double));
    double *Gmx = (double *)calloc(n, sizeof(double));
    double *zend = (double *)calloc(n, sizeof(double)), *azid = (double *)calloc(n, sizeof(double));
    int *sindex = (int *)calloc(n, sizeof(int)), *windex = (int *)calloc(n, sizeof(int));
    const int ndays = tsteps / 24;
    double sdp[4];
    snowdenp(in->snowenv, sdp);
    for (int i = 0; i < tsteps; ++i) windex[i] = (int)round(cl->winddir[i] / 45) % 8;
    if (!af) {
        snowalb(cl->precip, 1, tsteps, salb);
        for (int i = 0; i < tsteps; ++i) {
            orc_solmodel sp = orc_solposition(in->other.lat, in->other.lon, in->obstime.year[i],
                                              in->obstime.month[i], in->obstime.day[i], in->obstime.hour[i]);
            zend[i] = sp.zend;
            azid[i] = sp.azid;
            sindex[i] = (int)round(azid[i] / 15) % 24;
        }
    }
    for (int64_t j = 0; j < cols; ++j) {
        for (int64_t i = 0; i < rows; ++i) {
            const int64_t c = i + rows * j;
            const double hgt = in->vegp.hgt[c];
            if (is_na(hgt)) continue;
            /* time-class quantities: shared (1, recomputed here per cell for simplicity) or per cell (2) */
            const int64_t off = af ? c : 0;
            if (af) snowalb(cl->precip + c, N, tsteps, salb);
            for (int k = 0; k < tsteps; ++k) {
                int64_t idx = off + fs * k;
                if (!af) {
                    ea[k] = orc_satvap(cl->temp[idx]) * cl->relhum[idx] / 100.0;
                    te[k] = (pm->Tc[idx] + cl->temp[idx]) / 2.0;
                }
                double Rem = 0.97 * SB * radem(cl->temp[idx]);
                Rnet[k] = pm->RswabsG[idx] + pm->RlwabsG[idx] - Rem;
                Rmx[k] = Rmn[k] = Rswmn[k] = Rlwmn[k] = Rswmx[k] = Rlwmx[k] = Gmx[k] = 0.0;
            }
            for (int d = 0; d < ndays; ++d) {
                double Rmxd = -1352.0, Rmnd = 1352.0, Rswmnd = 0.0, Rlwmnd = 0.0, Rswmxd = 0.0, Rlwmxd = 0.0;
                double Gmxd = 0.0;
                for (int h = 0; h < 24; ++h) {
                    int k = d * 24 + h;
                    int64_t idx = off + fs * k;
                    if (Rmxd < Rnet[k]) { Rmxd = Rnet[k]; Rswmxd = cl->swdown[idx]; Rlwmxd = cl->lwdown[idx]; }
                    if (Rmnd > Rnet[k]) { Rmnd = Rnet[k]; Rswmnd = cl->swdown[idx]; Rlwmnd = cl->lwdown[idx]; }
                    if (fabs(Rnet[k]) > Gmxd) Gmxd = fabs(Rnet[k]);
                }
                for (int h = 0; h < 24; ++h) {
                    int k = d * 24 + h;
                    Rmx[k] = Rmxd; Rmn[k] = Rmnd; Rswmn[k] = Rswmnd; Rlwmn[k] = Rlwmnd;
                    Rswmx[k] = Rswmxd; Rlwmx[k] = Rlwmxd; Gmx[k] = Gmxd;
                }
            }
            vegpoint_t vegpo;
            vegpo.pai = in->vegp.pai[c]; vegpo.hgt = hgt; vegpo.clump = in->vegp.clump[c];
            vegpo.ltra = in->vegp.leaft[c];
            otherpoint_t othero;
            othero.zref = in->other.zref; othero.psim = 0.0; othero.psih = 0.0; othero.G = 0.0;
            othero.slope = in->other.slope[c]; othero.aspect = in->other.aspect[c];
            othero.lat = af ? in->other.lats[c] : in->other.lat;
            othero.lon = af ? in->other.lons[c] : in->other.lon;
            const double skyview = in->other.skyview[c];
            int snowagec = in->other.isnowac[c];
            int snowageg = in->other.isnowag[c];
            double sdencp = ((sdp[0] - sdp[1]) * (1 - exp(-sdp[2] * in->other.isnowdc[c] / 100.0 -
                                                          sdp[3] * snowagec / 24.0)) + sdp[1]) * 1000.0;
            double sdengp = ((sdp[0] - sdp[1]) * (1 - exp(-sdp[2] * in->other.isnowdg[c] * 0.5 / 100.0 -
                                                          sdp[3] * snowageg / 24.0)) + sdp[1]) * 1000.0;
            double sdepcp = in->other.isnowdc[c];
            double sdepgp = in->other.isnowdg[c];
            double meltc = af ? NA : 0.0; /* cpp:4333 only exists in gridmodelsnow1 */
            double meltg = NA;
            for (int k = 0; k < tsteps; ++k) {
                const int64_t idx = off + fs * k;   /* forcing index */
                const int64_t odx = c + N * k;      /* output index  */
                const double tc = cl->temp[idx], prec = cl->precip[idx];
                int snowtest = 0;
                if (sdepcp > 0.0) snowtest = 1;
                if (tc < 2.0 && prec > 0.0) snowtest = 1;
                if (snowtest > 0) {
                    double paip = in->vegp.pai[c];
                    if (hgt > sdepgp) paip = paip * (hgt - sdepgp) / hgt;
                    double dtR = Rmx[k] - Rmn[k];
                    double trS = skyview * exp(-paip);
                    double Rem = 0.97 * SB * radem(tc);
                    double dmxS = trS * Rswmx[k] + trS * Rlwmx[k] + (1 - trS) * Rem - Rem;
                    double dmnS = trS * Rswmn[k] + trS * Rlwmn[k] + (1 - trS) * Rem - Rem;
                    double Gmu = (dmxS - dmnS) / dtR;
                    double G = pm->Gp[idx] * Gmu;
                    if (G > Gmx[k]) G = Gmx[k];
                    if (G < -Gmx[k]) G = -Gmx[k];
                    double ha, smu = 1.0;
                    if (!af) {
                        ha = in->other.hor[(int64_t)sindex[k] * N + c];
                        double sa = 90 - zend[k];
                        if (ha > tan(sa * TORAD)) smu = 0.0;
                    } else {
                        orc_solmodel solp = orc_solposition(othero.lat, othero.lon, in->obstime.year[k],
                                                            in->obstime.month[k], in->obstime.day[k],
                                                            in->obstime.hour[k]);
                        int si_ = (int)round(solp.azid / 15.0) % 24;
                        ha = in->other.hor[(int64_t)si_ * N + c];
                        double sa = PI_ / 2.0 - solp.zenr;
                        if (ha > tan(sa)) smu = 0.0;
                    }
                    double ws = in->other.wsa[(int64_t)windex[k] * N + c];
                    double u2p = pm->umu[idx] * ws * cl->windspeed[idx];
                    double Rdifp = cl->difrad[idx] * skyview;
                    double Rdirp = (cl->swdown[idx] - cl->difrad[idx]) * smu;
                    double Rswp = Rdirp + Rdifp;
                    double Rlwp = cl->lwdown[idx] * skyview;
                    double eak, tek;
                    if (!af) { eak = ea[k]; tek = te[k]; }
                    else {
                        eak = orc_satvap(tc) * cl->relhum[idx] / 100.0;
                        tek = (pm->Tc[idx] + tc) / 2.0;
                    }
                    obspoint_t obstimeo;
                    obstimeo.year = in->obstime.year[k]; obstimeo.month = in->obstime.month[k];
                    obstimeo.day = in->obstime.day[k]; obstimeo.hour = in->obstime.hour[k];
                    climpoint_t climo;
                    climo.tc = tc; climo.ea = eak; climo.pk = cl->pres[idx]; climo.u2 = u2p; climo.prec = prec;
                    climo.Rsw = Rswp; climo.Rdif = Rdifp; climo.Rlw = Rlwp; climo.Tci = pm->Tc[idx]; climo.te = tek;
                    othero.G = G;
                    snowpoint_t snowo;
                    snowo.alb = salb[k]; snowo.sdenc = sdencp; snowo.sdeng = sdengp;
                    snowo.sdepc = sdepcp; snowo.sdepg = sdepgp;
                    snowo.snowagec = (double)snowagec; snowo.snowageg = (double)snowageg;
                    snowmod_t smod = snowoneB(obstimeo, climo, vegpo, snowo, othero, sdp, 1.0);
                    if (out->Tc) out->Tc[odx] = smod.Tc;
                    if (out->Tg) out->Tg[odx] = smod.Tg;
                    if (out->sdepc) out->sdepc[odx] = smod.sdepc;
                    if (out->sdepg) out->sdepg[odx] = smod.sdepg;
                    if (out->sden) out->sden[odx] = smod.sdenc;
                    sdencp = smod.sdenc;
                    sdengp = smod.sdeng;
                    sdepcp = smod.sdepc;
                    sdepgp = smod.sdepg;
                    snowagec = (int)smod.snowagec;
                    snowageg = (int)smod.snowageg;
                    double melc = smod.mSc + smod.mMc + smod.mRc;
                    double melg = smod.mSg + smod.mMg + smod.mRg;
                    meltc = meltc + melc;
                    meltc = meltc + (melc * 1000.0) / smod.sdenc;
                    meltg = meltg + (melg * 1000.0) / smod.sdeng;
                } else {
                    if (out->Tc) out->Tc[odx] = 0.0;
                    if (out->Tg) out->Tg[odx] = 0.0;
                    if (out->sdepc) out->sdepc[odx] = 0.0;
                    if (out->sdepg) out->sdepg[odx] = 0.0;
                    if (out->sden) out->sden[odx] = sdp[1] * 1000.0;
                }
            }
            if (out->agec) out->agec[c] = snowagec;
            if (out->ageg) out->ageg[c] = snowageg;
            if (out->meltc) out->meltc[c] = meltc;
            if (out->meltg) out->meltg[c] = meltg;
        }
    }
    free(salb); free(ea); free(te); free(Rnet); free(Rmx); free(Rmn); free(Rswmn); free(Rlwmn);
    free(Rswmx); free(Rlwmx); free(Gmx); free(zend); free(azid); free(sindex); free(windex);
    return 0;
}

/* hdr:192-200 snowpoint2, hdr:246-256 snowmicro */
typedef struct { double snowtempg, snowtempc, sdepc, sdepg, sdenc, albg, albc; } snowpoint2_t;
typedef struct { double Tz, tleaf, rh, uz, Rbdown, Rddown, Rlwdn, Rdup, Rlwup; } snowmicro_t;

/* cpp:4739-4866 snowabovepoint */
static snowmicro_t snowabovepoint(double reqhgt, double zref, double tc, double relhum, double pk, double u2,
                                  double Rsw, double Rdif, double Rlw, double hgt, double pai, double paia,
                                  double leafd, double clump, double ltra, double leafden, orc_solmodel solp,
                                  double si, double svfa, int shadowmask, double ws, double umu, double mxtc,
                                  snowpoint2_t snowp) {
    if (reqhgt == 0.0) reqhgt = 0.001;
    snowmicro_t out;
    memset(&out, 0, sizeof out);
    double es = orc_satvap(tc);
    double ea = es * relhum / 100.0;
    double tdew = dewpoint_cpp(ea);
    double hgts = hgt - snowp.sdepg;
    if (hgts < 0.0) hgts = 0.0;
    double pais = 0.0;
    tiw_t tiw;
    tiw.a = 0.0; /* left uninitialised by the reference when hgts <= 0; only read when reqhgt < hgts */
    if (hgts > 0.0) {
        pais = pai * hgts / hgt;
        tiw = windti(hgts, pais);
    } else {
        tiw.d = 0.0;
        tiw.zm = 1e-5;
    }
    wind_t wnd = wind(reqhgt, zref, hgts, pais, u2, umu, ws, tiw);
    out.uz = wnd.uz;
    double ez;
    if (reqhgt >= hgts) {
        if (Rsw > 0.0) {
            out.Rddown = Rdif * svfa;
            if (si > 0.0) {
                if (shadowmask > 0) {
                    out.Rbdown = (Rsw - Rdif) / si;
                    if (out.Rbdown > 1352.0) out.Rbdown = 1352.0;
                    out.Rdup = snowp.albc * Rsw * svfa;
                } else {
                    out.Rbdown = 0.0;
                    out.Rdup = snowp.albc * Rdif * svfa;
                }
            } else {
                out.Rbdown = 0.0;
                out.Rdup = snowp.albc * Rdif * svfa;
            }
        } else {
            out.Rbdown = 0.0;
            out.Rddown = 0.0;
            out.Rdup = 0.0;
        }
        out.Rlwdn = svfa * Rlw;
        out.Rlwup = svfa * 0.97 * SB * radem(snowp.snowtempc);
        abovecan_t tv = TVabove(reqhgt, zref, hgts, tiw.d, tiw.zm, snowp.snowtempc, tc, ea, 1.0);
        out.Tz = tv.Tz;
        out.tleaf = snowp.snowtempc;
        ez = tv.ez;
    } else {
        double paias = 0.0;
        if (hgts > 0.0) paias = paia * hgts / hgt;
        double zi = 0.0;
        if (snowp.sdepg > 0.0) zi = ((snowp.sdepc - snowp.sdepg) * snowp.sdenc) / (hgts * 1000.0);
        double ltras = ltra * exp(-10.1 * zi);
        double sm = ltras + snowp.albc;
        if (sm > 0.999) ltras = 0.999 - snowp.albc;
        double clumps = clump;
        if (clump > 0.0) clumps = pow(clump, pais / pai);
        double pait = pais;
        if (clump > 0.0) pait = pais / (1.0 - clumps);
        tsdif_t tspdif = twostreamdif_params(pait, 1.0, snowp.albc, ltras, snowp.albg);
        tir_t tir = twostreamdif(pais, paias, 1.0, snowp.albc, ltras, clumps, snowp.albg);
        orc_kstruct kp = orc_cank(solp.zenr, 1.0, si);
        tsdir_t tspdir = twostreamdir_params(pait, tspdif.om, tspdif.a, tspdif.gma, tspdif.J, tspdif.del, tspdif.h,
                                             snowp.albg, kp.kd, tspdif.u1, tspdif.S1, tspdif.D1, tspdif.D2);
        rad_t rad = twostream(pais, clumps, snowp.albg, svfa, si, tc, Rsw, Rdif, Rlw, solp, kp, tspdir, tir);
        if (shadowmask == 0) rad.Rbdown = 0.0;
        stomp_t stomp;
        memset(&stomp, 0, sizeof stomp); /* uninitialised in the reference; unused as gsmax = 999.999 */
        leaft_t tvl = leaftemp(snowp.snowtempc, snowp.snowtempg, tc, mxtc, pk, ea, es, wnd.uz, tdew, 1.0,
                               rad.radLsw, rad.Rddown, rad.Rbdown, Rlw, pais, paias, leafd, 999.999, rad.radLpar,
                               0.4, 0.4, 2.6, 5.2, stomp);
        out.tleaf = tvl.tleaf;
        double H = 29.3 * wnd.gHa * (snowp.snowtempc - tc);
        double Flux = H * (1.0 - exp(-pais));
        double Fluxz = tvl.H;
        abovecan_t tv = TVabove(hgts, zref, hgts, tiw.d, tiw.zm, snowp.snowtempc, tc, ea, 1.0);
        double SH = tv.Tz * 29.3 * 43.0;
        double SG = snowp.snowtempg * 29.3 * 43.0;
        double mxnear = fabs(out.tleaf - tv.Tz) * 29.3 * 43.0;
        out.Tz = TVbelow(zref, reqhgt, tiw.d, hgts, pais, wnd.uf, leafden, Flux, Fluxz, SH, SG, mxnear) /
                 (29.3 * 43);
        double la;
        if (tc < 0) la = 51078.69 - 4.338 * tc - 0.06367 * tc * tc;
        else la = 45068.7 - 42.8428 * tc;
        double m = la * (wnd.gHa / pk);
        double L = m * (es - ea);
        Flux = L * (1.0 - exp(-pais));
        Fluxz = tvl.L;
        double mu = la * (43 / pk);
        SH = tv.ez * mu;
        SG = orc_satvap(snowp.snowtempg) * mu;
        mxnear = fabs(orc_satvap(out.tleaf) - tv.ez) * mu;
        ez = TVbelow(zref, reqhgt, tiw.d, hgts, pais, wnd.uf, leafden, Flux, Fluxz, SH, SG, mxnear) / mu;
        out.Rbdown = rad.Rbdown;
        out.Rddown = rad.Rddown;
        out.Rdup = rad.Rdup;
        out.Rlwdn = tvl.lwdn;
        out.Rlwup = tvl.lwup;
    }
    out.rh = (ez / orc_satvap(out.Tz)) * 100.0;
    if (out.rh > 100.0) out.rh = 100.0;
    double tmx = max4(out.tleaf, tc, snowp.snowtempg, snowp.snowtempc) + 2.0;
    double tmn = min4(out.tleaf, tc, snowp.snowtempg, snowp.snowtempc) - 2.0;
    if (out.Tz > tmx) out.Tz = tmx;
    if (out.Tz < tmn) out.Tz = tmn;
    return out;
}

/* cpp:4868-4891 belowpointsnow */
static double belowpointsnow(double reqhgt, double meanD, double snowtempg, double Tzd, double Tza, double hiy) {
    double nb = -118.35 * reqhgt / meanD;
    double Tz = snowtempg;
    if (nb > 1.0) {
        if (nb <= 24.0) {
            double w1 = 1.0 / nb;
            double w2 = nb / 24.0;
            double wgt = w1 / (w1 + w2);
            Tz = wgt * snowtempg + (1 - wgt) * Tzd;
        } else {
            if (nb <= hiy) {
                double w1 = 24.0 / nb;
                double w2 = nb / hiy;
                double wgt = w1 / (w1 + w2);
                Tz = wgt * Tzd + (1 - wgt) * Tza;
            } else {
                Tz = Tza;
            }
        }
    }
    return Tz;
}

/* cpp:4894-5056 gridmicrosnow1 (array_forcing == 0) / cpp:5059-5214 gridmicrosnow2 (== 1), with
 * snowdayan (cpp:4679-4712) and meanDsnow (cpp:4713-4737) inlined per cell.  Differences kept:
 *   1: mxtc over the whole series, si NA-fallback cos(zend * torad)
 *   2: mxtc and albedo per cell, si NA-fallback cos(zenr). */
int orc_gridmicrosnow(const mcf_snow_inputs *in, const mcf_snowm *sm, double reqhgt, double mat,
                      const int32_t *outsel, mcf_outputs *micro) {
    const int64_t rows = in->rows, cols = in->cols;
    const int tsteps = (int)in->tsteps;
    const int64_t N = rows * cols;
    const int af = in->array_forcing;
    const int64_t fs = af ? N : 1;
    const mcf_snow_climate *cl = &in->clim;
    size_t n = (size_t)tsteps + 1;
    double *salb = (double *)calloc(n, sizeof(double)), *Tzd = (double *)calloc(n, sizeof(double));
    int *windex = (int *)calloc(n, sizeof(int));
    const int ndays = tsteps / 24;
    double mxtc = -273.15;
    for (int i = 0; i < tsteps; ++i) windex[i] = (int)round(cl->winddir[i] / 45) % 8;
    if (!af) {
        snowalb(cl->precip, 1, tsteps, salb);
        for (int i = 0; i < tsteps; ++i) if (cl->temp[i] > mxtc) mxtc = cl->temp[i];
    }
    const int y0 = in->obstime.year[0];
    const int hiy = (y0 % 4 == 0 && (y0 % 100 != 0 || y0 % 400 == 0)) ? 366 * 24 : 365 * 24;
    for (int64_t j = 0; j < cols; ++j) {
        for (int64_t i = 0; i < rows; ++i) {
            const int64_t c = i + rows * j;
            const double hgt = in->vegp.hgt[c];
            if (is_na(hgt)) continue;
            const int64_t off = af ? c : 0;
            /* snowdayan / meanDsnow: NA when the first step of their input is NA */
            for (int k = 0; k < tsteps; ++k) Tzd[k] = orc_na_real();
            if (!is_na(sm->Tg[c])) {
                for (int d = 0; d < ndays; ++d) {
                    double sumd = 0.0;
                    for (int h = 0; h < 24; ++h) sumd += sm->Tg[c + N * (d * 24 + h)];
                    double meand = sumd / 24.0;
                    for (int h = 0; h < 24; ++h) Tzd[d * 24 + h] = meand;
                }
            }
            double meanD = orc_na_real();
            if (!is_na(sm->snowden[c])) {
                double sumD = 0.0;
                for (int k = 0; k < tsteps; ++k) {
                    double den = sm->snowden[c + N * k];
                    double co = 0.0442 * exp(5.181 * den / 1000.0);
                    double kap = co / (den * 2090.0);
                    sumD += sqrt(2.0 * kap / OMDY);
                }
                meanD = sumD / (double)tsteps;
            }
            if (af) {
                mxtc = -273.15;
                for (int k = 0; k < tsteps; ++k) if (cl->temp[c + N * k] > mxtc) mxtc = cl->temp[c + N * k];
                snowalb(cl->precip + c, N, tsteps, salb);
            }
            const double lat = af ? in->other.lats[c] : in->other.lat;
            const double lon = af ? in->other.lons[c] : in->other.lon;
            for (int k = 0; k < tsteps; ++k) {
                const int64_t idx = off + fs * k;
                const int64_t odx = c + N * k;
                if (!(sm->totalSWE[odx] > 0.0)) continue;
                double reqhgts = reqhgt - sm->groundsnowdepth[odx];
                if (reqhgts >= 0.0) {
                    orc_solmodel solp = orc_solposition(lat, lon, in->obstime.year[k], in->obstime.month[k],
                                                        in->obstime.day[k], in->obstime.hour[k]);
                    int sindex = (int)round(solp.azid / 15) % 24;
                    int shadowmask = 1;
                    double ha = in->other.hor[(int64_t)sindex * N + c];
                    double sa = (PI_ / 2.0) - solp.zenr;
                    double si = orc_solarindex(in->other.slope[c], in->other.aspect[c], solp.zend, solp.azid, 1);
                    if (is_na(si)) si = af ? cos(solp.zenr) : cos(solp.zend * TORAD);
                    if (ha > tan(sa)) shadowmask = 0;
                    double ws = in->other.wsa[(int64_t)windex[k] * N + c];
                    snowpoint2_t snowp;
                    snowp.snowtempg = sm->Tg[odx]; snowp.snowtempc = sm->Tc[odx];
                    snowp.sdepc = sm->totalSWE[odx] / sm->snowden[odx];
                    snowp.sdepg = sm->groundsnowdepth[odx]; snowp.sdenc = sm->snowden[odx];
                    snowp.albc = salb[k]; snowp.albg = salb[k];
                    snowmicro_t apv = snowabovepoint(reqhgts, in->other.zref, cl->temp[idx], cl->relhum[idx],
                                                     cl->pres[idx], cl->windspeed[idx], cl->swdown[idx],
                                                     cl->difrad[idx], cl->lwdown[idx], hgt, in->vegp.pai[c],
                                                     in->vegp.paia[c], in->vegp.leafd[c], in->vegp.clump[c],
                                                     in->vegp.leaft[c], in->vegp.leafden[c], solp, si,
                                                     in->other.skyview[c], shadowmask, ws, cl->umu[idx], mxtc,
                                                     snowp);
                    if (outsel[0]) micro->var[0][odx] = apv.Tz;
                    if (outsel[1]) micro->var[1][odx] = apv.tleaf;
                    if (outsel[2]) micro->var[2][odx] = apv.rh;
                    if (outsel[4]) micro->var[4][odx] = apv.uz;
                    if (outsel[5]) micro->var[5][odx] = apv.Rbdown;
                    if (outsel[6]) micro->var[6][odx] = apv.Rddown;
                    if (outsel[7]) micro->var[7][odx] = apv.Rlwdn;
                    if (outsel[8]) micro->var[8][odx] = apv.Rdup;
                    if (outsel[9]) micro->var[9][odx] = apv.Rlwup;
                } else {
                    double bpv = belowpointsnow(reqhgts, meanD, sm->Tg[odx], Tzd[k], mat, hiy);
                    if (outsel[0]) micro->var[0][odx] = bpv;
                    if (outsel[1]) micro->var[1][odx] = bpv;
                    if (outsel[2]) micro->var[2][odx] = 100.0;
                    if (outsel[4]) micro->var[4][odx] = 0.0;
                    if (outsel[5]) micro->var[5][odx] = 0.0;
                    if (outsel[6]) micro->var[6][odx] = 0.0;
                    if (outsel[7]) micro->var[7][odx] = 0.0;
                    if (outsel[8]) micro->var[8][odx] = 0.0;
                    if (outsel[9]) micro->var[9][odx] = 0.0;
                }
                if (outsel[3]) micro->var[3][odx] = in->other.Smax[c];
            }
        }
    }
    free(salb); free(Tzd); free(windex);
    return 0;
}
