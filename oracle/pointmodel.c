/*
 * pointmodel.c — TEST INFRASTRUCTURE (part of the CPU oracle, see mcf_oracle.c).
 *
 * Restatement of the reference functions that its own tests drive around the grid
 * solver's arithmetic, so that those tests can be replayed against the oracle
 * (oracle/replay_reference_tests.py):
 *
 *   tests/testthat/test-microclimatemodel_wrapper.R  ->  clearskyradCpp cpp:5220, solpositionvCpp
 *       cpp:5244, BigLeafCpp cpp:710-881 (+ GFluxCpp cpp:641-707, RadswabsCpp cpp:187-278 and the
 *       stability functions cpp:312-371), microclimatemodel_wrapper cpp:5783-5952
 *   tests/testthat/test-BigLeafCpp.R                 ->  BigLeafCpp
 *
 * It is compiled into the same translation unit as mcf_oracle.c (oracle_unit.c) and
 * shares its static helpers.  Not on the product path.
 */
#include "pointmodel.h"

/* cpp:280-285 phairCpp, cpp:287-291 cpairCpp */
static double phair(double tc, double pk) { return 44.6 * (pk / 101.3) * (273.15 / (tc + 273.15)); }
static double cpair(double tc) { return 2e-05 * pow(tc, 2.0) + 0.0002 * tc + 29.119; }

/* cpp:312-328 dpsimCpp */
static double dpsim(double ze) {
    double psim;
    if (ze < 0) {
        double x = pow((1.0 - 15.0 * ze), 0.25);
        psim = log(pow((1.0 + x) / 2.0, 2.0) * (1 + pow(x, 2.0)) / 2.0) - 2.0 * atan(x) + PI_ / 2.0;
    } else {
        psim = -4.7 * ze;
    }
    if (psim < -4.0) psim = -4.0;
    if (psim > 3.0) psim = 3.0;
    return psim;
}
/* cpp:330-345 dpsihCpp */
static double dpsih(double ze) {
    double psih;
    if (ze < 0) {
        double y = sqrt(1.0 - 9.0 * ze);
        psih = log(pow((1.0 + y) / 2.0, 2.0));
    } else {
        psih = -(4.7 * ze) / 0.74;
    }
    if (psih < -4.0) psih = -4.0;
    if (psih > 3.0) psih = 3.0;
    return psih;
}
/* cpp:347-362 dphihCpp */
static double dphih(double ze) {
    double phih;
    if (ze < 0) {
        double phim = 1 / pow((1.0 - 16.0 * ze), 0.25);
        phih = pow(phim, 2.0);
    } else {
        phih = 1 + ((6.0 * ze) / (1.0 + ze));
    }
    if (phih > 1.5) phih = 1.5;
    if (phih < 0.5) phih = 0.5;
    return phih;
}
/* cpp:364-371 gfreeCpp */
static double gfree(double leafd, double H) {
    double d = 0.71 * leafd;
    double dT = 0.7045388 * pow((d * pow(H, 4.0)), 0.2);
    double gha = 0.0375 * pow(dT / d, 0.25);
    if (gha < 0.1) gha = 0.1;
    return gha;
}
/* cpp:493-496 dewpointCpp */
static double dewpoint_cpp(double ea) { return 243.5 * log(ea / 0.6112) / (17.67 - log(ea / 0.6112)); }

/* cpp:498-514 PenmanMonteithCpp */
static double penman(double Rabs, double gHa, double gV, double tc, double te, double pk, double ea, double em,
                     double G, double erh) {
    double Rema = em * SB * radem(tc);
    double la;
    if (te >= 0) la = 45068.7 - 42.8428 * te;
    else la = 51078.69 - 4.338 * te - 0.06367 * te * te;
    double cp = cpair(te);
    double Da = orc_satvap(tc) - ea;
    double gR = (4.0 * em * SB * pow(te + 273.15, 3.0)) / cp;
    double De = orc_satvap(te + 0.5) - orc_satvap(te - 0.5);
    return tc + ((Rabs - Rema - la * (gV / pk) * Da * erh - G) / (cp * (gHa + gR) + la * (gV / pk) * De * erh));
}

/* cpp:574-594 mayCpp: daily means, 91-day circular mean, expanded to hourly */
static void may_circ(const double *x, int m, double *z) {
    int nd = m / 24;
    double *d = (double *)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    double *y = (double *)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    for (int i = 0; i < nd; ++i) {
        double s = 0.0;
        for (int j = 0; j < 24; ++j) s += x[i * 24 + j];
        d[i] = s / 24.0;
    }
    ma_circ(d, nd, 91, y);
    for (int i = 0; i < nd; ++i)
        for (int j = 0; j < 24; ++j) z[i * 24 + j] = y[i];
    free(d); free(y);
}

/* cpp:187-278 RadswabsCpp */
static void radswabs(double pai, double x, double lref, double ltra, double clump, double gref, double slope,
                     double aspect, double lat, double lon, const int *year, const int *month, const int *day,
                     const double *lt, const double *Rsw, const double *Rdif, int n, double *radGsw,
                     double *radCsw, double *albedo) {
    if (pai > 0.0) {
        double pait = pai;
        if (clump > 0.0) pait = pai / (1 - clump);
        tsdif_t p = twostreamdif_params(pait, x, lref, ltra, gref);
        double trd = clump * clump;
        double amx = gref;
        if (amx < lref) amx = lref;
        double albd = gref * (trd * trd) + (1.0 - trd * trd) * (p.p1 + p.p2);
        if (albd > amx) albd = amx;
        if (albd < 0.01) albd = 0.01;
        double groundRdd = trd + (1.0 - trd) * (p.p3 * exp(-p.h * pait) + p.p4 * exp(p.h * pait));
        for (int i = 0; i < n; ++i) {
            if (Rsw[i] > 0.0) {
                orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], lt[i]);
                double si = orc_solarindex(slope, aspect, sp.zend, sp.azid, 0);
                if (sp.zenr > PI_ / 2.0) sp.zenr = PI_ / 2.0;
                double cosz = cos(sp.zenr);
                orc_kstruct kp = orc_cank(sp.zenr, x, si);
                tsdir_t d = twostreamdir_params(pait, p.om, p.a, p.gma, p.J, p.del, p.h, gref, kp.kd, p.u1,
                                                p.S1, p.D1, p.D2);
                double Rbeam = (Rsw[i] - Rdif[i]) / cosz;
                if (Rbeam > 1352.0) Rbeam = 1352.0;
                double trb = pow(clump, kp.Kc);
                if (trb > 0.999) trb = 0.999;
                if (trb < 0.0) trb = 0.0;
                double Rb = Rbeam * cosz;
                double trg = trb + (1 - trb) * exp(-kp.kd * pait);
                double Rbc = (trg * si + (1 - trg) * cosz) * Rbeam;
                double albb = trd * trb * gref + (1.0 - trd * trb) * (d.p5 / -d.sig + d.p6 + d.p7);
                if (albb > amx) albb = amx;
                if (albb < 0.01) albb = 0.01;
                double groundRbdd = trb + (1.0 - trb) * ((d.p8 / d.sig) * exp(-kp.kd * pait) +
                                                         d.p9 * exp(-p.h * pait) + d.p10 * exp(p.h * pait));
                if (groundRbdd > amx) groundRbdd = amx;
                if (groundRbdd < 0.0) groundRbdd = 0.0;
                radCsw[i] = (1.0 - albd) * Rdif[i] + (1.0 - albb) * Rbc;
                double Rgdif = groundRdd * Rdif[i] + groundRbdd * Rb;
                radGsw[i] = (1.0 - gref) * (Rgdif + exp(-kp.kd * pait) * Rbeam * si);
                albedo[i] = 1.0 - (radCsw[i] / (Rdif[i] + Rb));
                if (albedo[i] > amx) albedo[i] = amx;
                if (albedo[i] < 0.01) albedo[i] = 0.01;
            } else {
                radGsw[i] = 0;
                radCsw[i] = 0;
                albedo[i] = lref;
            }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            albedo[i] = gref;
            if (Rsw[i] > 0) {
                orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], lt[i]);
                double si = orc_solarindex(slope, aspect, sp.zend, sp.azid, 0);
                if (sp.zenr > PI_ / 2.0) sp.zenr = PI_ / 2.0;
                double dirr = (Rsw[i] - Rdif[i]) / cos(sp.zenr);
                radGsw[i] = (1 - gref) * (Rdif[i] + si * dirr);
                radCsw[i] = radGsw[i];
            } else {
                radGsw[i] = 0;
                radCsw[i] = 0;
            }
        }
    }
}

/* cpp:641-707 GFluxCpp.  Gmin/Gmax are in/out (set on iter == 0). */
static void gflux(const double *Tg, const double *soilm, int n, double rho, double Vm, double Vq, double Mc,
                  double *Gmax, double *Gmin, int iter, int yearG, double *G) {
    double frs = Vm + Vq;
    double c1 = (0.57 + 1.73 * Vq + 0.93 * Vm) / (1.0 - 0.74 * Vq - 0.49 * Vm) - 2.8 * frs * (1.0 - frs);
    double c3 = 1.0 + 2.6 * pow(Mc, -0.5);
    double c4 = 0.03 + 0.7 * frs * frs;
    double mu1 = 2400.0 * rho / 2.64;
    double mu2 = 1.06 * rho;
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *Td = (double *)calloc(1, nb), *Gmu = (double *)calloc(1, nb), *dT = (double *)calloc(1, nb);
    double *k = (double *)calloc(1, nb), *kap = (double *)calloc(1, nb), *Gmud = (double *)calloc(1, nb);
    hourtoday(Tg, n, 2, Td);
    for (int i = 0; i < n; ++i) {
        double cs = mu1 + 4180 * soilm[i];
        double ph = (rho * (1.0 - soilm[i]) + soilm[i]) * 1000;
        double c2 = mu2 * soilm[i];
        k[i] = c1 + c2 * soilm[i] - (c1 - c4) * exp(-pow(c3 * soilm[i], 4.0));
        kap[i] = k[i] / (cs * ph);
        double DD = sqrt(2 * kap[i] / OMDY);
        Gmu[i] = sqrt(2) * (k[i] / DD) * 0.5;
        dT[i] = Tg[i] - Td[i];
    }
    ma_circ(Gmu, n, 6, Gmud);
    ma_circ(dT, n, 6, G);
    for (int i = 0; i < n; ++i) G[i] = G[i] * Gmud[i] * 1.1171;
    if (iter == 0) {
        hourtoday(G, n, 1, Gmin);
        hourtoday(G, n, 0, Gmax);
    }
    for (int i = 0; i < n; ++i) {
        if (G[i] < Gmin[i]) G[i] = Gmin[i];
        if (G[i] > Gmax[i]) G[i] = Gmax[i];
    }
    if (yearG) {
        double *kma = (double *)calloc(1, nb), *kama = (double *)calloc(1, nb), *dTy = (double *)calloc(1, nb),
               *madTy = (double *)calloc(1, nb);
        may_circ(k, n, kma);
        may_circ(kap, n, kama);
        double sumTd = 0.0;
        for (int i = 0; i < n; ++i) sumTd += Td[i];
        for (int i = 0; i < n; ++i) dTy[i] = Td[i] - sumTd / n;
        may_circ(dTy, n, madTy);
        for (int i = 0; i < n; ++i) {
            double omyr = (2 * PI_) / (n * 3600.0);
            double Gmuy = sqrt(2) * kma[i] / sqrt(2 * kama[i] / omyr);
            G[i] = G[i] + madTy[i] * Gmuy * 1.1171;
        }
        free(kma); free(kama); free(dTy); free(madTy);
    }
    free(Td); free(Gmu); free(dT); free(k); free(kap); free(Gmud);
}

/* cpp:710-881 BigLeafCpp.  vegp/groundp are read POSITIONALLY as the reference does. */
int orc_bigleaf(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                const double *wspeed, const double *vegp, const double *groundp, const double *soilm, double lat,
                double lon, double dTmx, double zref, int maxiter, double bwgt, double tol, int yearG,
                orc_bigleaf_out *o) {
    double h = vegp[0], pai = vegp[1], vegx = vegp[2], clump = vegp[3], lref = vegp[4], ltra = vegp[5],
           leafd = vegp[6], em = vegp[7], gsmax = vegp[8];
    double gref = groundp[0], slope = groundp[1], aspect = groundp[2], groundem = groundp[3], rho = groundp[4],
           Vm = groundp[5], Vq = groundp[6], Mc = groundp[7], soilb = groundp[8], psie = groundp[9],
           Smax = groundp[10], Smin = groundp[11];
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *swG = (double *)calloc(1, nb), *swC = (double *)calloc(1, nb);
    radswabs(pai, vegx, lref, ltra, clump, gref, slope, aspect, lat, lon, year, month, day, hour, Rsw, Rdif, n, swG,
             swC, o->albedo);
    double pait = pai / (1 - clump);
    double trd = (1 - clump * clump) * exp(-pait) + clump * clump;
    double d = orc_zeroplanedis(h, pai);
    double Belim = 0.4 / sqrt(0.003 + (0.2 * pai) / 2);
    double *tcc = (double *)calloc(1, nb), *tcg = (double *)calloc(1, nb), *Gmin = (double *)calloc(1, nb),
           *Gmax = (double *)calloc(1, nb), *Gnew = (double *)calloc(1, nb);
    for (int i = 0; i < n; ++i) {
        o->Tg[i] = tc[i]; o->Tc[i] = tc[i]; tcc[i] = tc[i]; tcg[i] = tc[i];
        o->psim[i] = 0; o->psih[i] = 0; o->phih[i] = 0; o->OL[i] = 0; o->G[i] = 0;
        Gmin[i] = -999.0; Gmax[i] = 999.0; o->uf[i] = 999.0; o->RabsG[i] = 999.0;
        o->H[i] = 0.5 * Rsw[i] - em * SB * radem(tc[i]);
    }
    stomp_t st = stomparams(h, lat, vegx);
    double om = 0.5 * (lref + ltra);
    double tstf = tol * 2, tst = 0;
    int iter = 0;
    while (tstf > tol) {
        tst = 0;
        for (int i = 0; i < n; ++i) {
            double RemC = em * SB * radem(o->Tc[i]);
            double radClw = em * Rlw[i];
            double radGlw = groundem * (trd * radClw + (1 - trd) * RemC);
            o->RabsG[i] = swG[i] + radGlw;
            double RabsC = swC[i] + radClw;
            double zm = orc_roughlength(h, pai, d, o->psih[i]);
            o->uf[i] = (KA * wspeed[i]) / (log((zref - d) / zm) + o->psim[i]);
            if (o->uf[i] < 0.0002) o->uf[i] = 0.0002;
            double gmin = gfree(leafd, fabs(o->H[i])) * 2 * pai;
            double ph = phair(tcc[i], pk[i]);
            double gHa = gturb(o->uf[i], d, zm, zref, ph, o->psih[i], gmin);
            orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], hour[i]);
            orc_kstruct kp = orc_cank(sp.zenr, vegx, cos(sp.zenr));
            double gC = canopycond(Rsw[i], Rdif[i], kp.k, om, soilm[i], gsmax, pai, Smax, psie, soilb, st);
            double gV = 1 / (1 / gHa + 1 / gC);
            if (gC == 0) gV = 0;
            double ea = orc_satvap(tc[i]) * rh[i] / 100;
            double Tcn = penman(RabsC, gHa, gV, tc[i], tcc[i], pk[i], ea, em, o->G[i], 1);
            double tdew = dewpoint_cpp(ea);
            if (Tcn < tdew) Tcn = tdew;
            double srh = (soilm[i] - Smin) / (Smax - Smin);
            double Tgn = penman(o->RabsG[i], gHa, gHa, tcg[i], tcc[i], pk[i], ea, em, o->G[i], srh);
            if (Tgn < tdew) Tgn = tdew;
            double dTc = Tcn - tc[i], dTg = Tgn - tc[i];
            if (dTc > dTmx) dTc = dTmx;
            if (dTg > dTmx) dTg = dTmx;
            Tcn = tc[i] + dTc;
            Tgn = tc[i] + dTg;
            double tst2 = fabs(Tcn - o->Tc[i]), tst3 = fabs(Tgn - o->Tg[i]);
            if (tst2 > tst) tst = tst2;
            if (tst3 > tst) tst = tst3;
            o->Tc[i] = bwgt * o->Tc[i] + (1 - bwgt) * Tcn;
            o->Tg[i] = bwgt * o->Tg[i] + (1 - bwgt) * Tgn;
            tcc[i] = (o->Tc[i] + tc[i]) / 2;
            tcg[i] = (o->Tg[i] + tc[i]) / 2;
            double Tk = 273.15 + tcc[i];
            ph = phair(tcc[i], pk[i]);
            double cp = cpair(tcc[i]);
            o->H[i] = bwgt * o->H[i] + (1 - bwgt) * (gHa * cp * (Tcn - tc[i]));
            double Rnet = RabsC - SB * em * radem(o->Tc[i]);
            if (Rnet > 0 && o->H[i] > Rnet) o->H[i] = Rnet;
            if (fabs(o->H[i]) < 0.1) o->H[i] = 0.1;
            o->OL[i] = (ph * cp * pow(o->uf[i], 3.0) * Tk) / (-0.4 * 9.81 * o->H[i]);
            o->psim[i] = dpsim(zm / o->OL[i]) - dpsim((zref - d) / o->OL[i]);
            o->psih[i] = dpsih((0.2 * zm) / o->OL[i]) - dpsih((zref - d) / o->OL[i]);
            o->phih[i] = dphih((zref - d) / o->OL[i]);
            double ln1 = log((zref - d) / zm), ln2 = log((zref - d) / (0.2 * zm));
            if (o->psim[i] < -0.9 * ln1) o->psim[i] = -0.9 * ln1;
            if (o->psih[i] < -0.9 * ln2) o->psih[i] = -0.9 * ln2;
            if (o->psim[i] > 0.9 * ln1) o->psim[i] = 0.9 * ln1;
            if (o->psih[i] > 0.9 * ln2) o->psih[i] = 0.9 * ln2;
            if (o->psih[i] > 0.9 * Belim) o->psih[i] = 0.9 * Belim;
        }
        gflux(o->Tg, soilm, n, rho, Vm, Vq, Mc, Gmax, Gmin, iter, yearG, Gnew);
        memcpy(o->G, Gnew, nb);
        tstf = tst;
        ++iter;
        if (iter >= maxiter) tstf = 0;
    }
    o->err = tst;
    o->iters = iter;
    free(swG); free(swC); free(tcc); free(tcg); free(Gmin); free(Gmax); free(Gnew);
    return 0;
}

/* cpp:5220-5242 clearskyradCpp */
void orc_clearskyrad(int n, const int *year, const int *month, const int *day, const double *lt, double lat,
                     double lon, const double *tc, const double *rh, const double *pk, double *Ic) {
    for (int i = 0; i < n; ++i) {
        Ic[i] = 0.0;
        orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], lt[i]);
        if (sp.zend <= 90.0) {
            double m = 35 * cos(sp.zenr) * pow(1224.0 * cos(sp.zenr) * cos(sp.zenr) + 1.0, -0.5);
            double TrTpg = 1.021 - 0.084 * sqrt(m * 0.00949 * pk[i] + 0.051);
            double xx = log(rh[i] / 100.0) + ((17.27 * tc[i]) / (237.3 + tc[i]));
            double Td = (237.3 * xx) / (17.27 - xx);
            double u = exp(0.1133 - log(3.78) + 0.0393 * Td);
            double Tw = 1 - 0.077 * pow(u * m, 0.3);
            double Ta = 0.935 * m;
            Ic[i] = 1352.778 * cos(sp.zenr) * (TrTpg * Tw * Ta);
        }
    }
}

/* cpp:5244-5262 solpositionvCpp (si with shadowmask = true) */
void orc_solpositionv(int n, const int *year, const int *month, const int *day, const double *lt, double lat,
                      double lon, double slope, double aspect, double *zen, double *azi, double *si) {
    for (int i = 0; i < n; ++i) {
        orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], lt[i]);
        zen[i] = sp.zend;
        azi[i] = sp.azid;
        si[i] = orc_solarindex(slope, aspect, sp.zend, sp.azid, 1);
    }
}

/* cpp:5783-5952 microclimatemodel_wrapper.  Output arrays of length n; unused ones are left untouched. */
int orc_wrapper(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                const double *wspeed, const double *BL_Tg, const double *BL_G, const double *BL_uf,
                const double *vegp, const double *groundp, double reqhgt, double zref, double lat, double lon,
                orc_wrapper_out *o) {
    double hgt = vegp[0], pai = vegp[1], vegx = vegp[2], clump = vegp[3], lref = vegp[4], ltra = vegp[5],
           leafd = vegp[6], gsmax = vegp[8];
    double gref = groundp[0], slope = groundp[1], aspect = groundp[2], rho = groundp[4], Vm = groundp[5],
           Vq = groundp[6], Mc = groundp[7], soilb = groundp[8], psie = groundp[9], Smax = groundp[10],
           Smin = groundp[11];
    double mxtc = -999.99;
    for (int i = 0; i < n; ++i) {
        o->soilm[i] = 0.3;
        if (tc[i] > mxtc) mxtc = tc[i];
    }
    double dp = 0.0, zmp = 0.0;
    if (reqhgt >= 0) {
        double paia = 0;
        if (reqhgt < hgt) paia = (1.0 - reqhgt / hgt) * pai;
        double pait = pai;
        if (clump > 0.0) pait = pai / (1.0 - clump);
        tsdif_t p = twostreamdif_params(pait, vegx, lref, ltra, gref);
        tir_t tir = twostreamdif(pai, paia, vegx, lref, ltra, clump, gref);
        stomp_t st;
        tiw_t tiw;
        memset(&st, 0, sizeof st);
        memset(&tiw, 0, sizeof tiw);
        if (reqhgt > 0.0) {
            st = stomparams(hgt, lat, vegx);
            tiw = windti(hgt, pai);
            dp = orc_zeroplanedis(hgt, pai);
            zmp = orc_roughlength(hgt, pai, dp, 0);
        }
        for (int i = 0; i < n; ++i) {
            orc_solmodel sp = orc_solposition(lat, lon, year[i], month[i], day[i], hour[i]);
            double si = orc_solarindex(slope, aspect, sp.zend, sp.azid, 0);
            orc_kstruct kp = orc_cank(sp.zenr, vegx, si);
            tsdir_t d = twostreamdir_params(pait, p.om, p.a, p.gma, p.J, p.del, p.h, gref, kp.kd, p.u1, p.S1,
                                            p.D1, p.D2);
            rad_t rv = twostream(pai, clump, gref, 1.0, si, tc[i], Rsw[i], Rdif[i], Rlw[i], sp, kp, d, tir);
            o->Rdirdown[i] = rv.Rbdown;
            o->Rdifdown[i] = rv.Rddown;
            o->Rswup[i] = rv.Rdup;
            if (reqhgt == 0) {
                o->Rlwdown[i] = rv.radGlw;
                o->Rlwup[i] = 0.97 * SB * radem(BL_Tg[i]);
                o->Tz[i] = BL_Tg[i];
            } else {
                double ufps = (KA * wspeed[i]) / log((zref - dp) / zmp);
                double umu = BL_uf[i] / ufps;
                wind_t wv = wind(reqhgt, zref, hgt, pai, zref, umu, 1.0, tiw);
                o->uz[i] = wv.uz;
                double es = orc_satvap(tc[i]);
                double ea = es * (rh[i] / 100.0);
                double tdew = dewpoint_cpp(ea);
                double leafden = pai / hgt;
                soilhr_t gv;
                gv.Tg = BL_Tg[i]; gv.G = BL_G[i]; gv.DD = 0;
                above_t a = TVaboveground(reqhgt, zref, tc[i], pk[i], ea, es, tdew, Rsw[i], Rdif[i], Rlw[i],
                                          o->soilm[i], hgt, pai, paia, vegx, leafd, leafden, Smin, Smax, psie, soilb,
                                          gsmax, mxtc, st, tir, rv, tiw, wv, gv);
                o->Tz[i] = a.Tz; o->tleaf[i] = a.tleaf; o->rh[i] = a.rh;
                o->Rlwdown[i] = a.lwdn; o->Rlwup[i] = a.lwup;
            }
        }
    } else {
        soilc_t sc = soilpfun(Vm, Vq, Mc, rho);
        double sumD = 0.0;
        for (int i = 0; i < n; ++i) sumD += soilcond(rho, o->soilm[i], sc).DD;
        double meanD = sumD / (double)n;
        double nbv = -118.35 * reqhgt / meanD;
        int nn = (int)round(nbv);
        orc_man(BL_Tg, n, nn, o->Tz);
    }
    return 0;
}

/* ---- runbioclimCpp, cpp:3245-3560 (one cell's series; bio[] gets all 19 values) ---------- */
static double std_dev(const double *v, int n) { /* calc_std_dev cpp:3227-3244 */
    if (n <= 1) return orc_na_real();
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    double mean = s / n, ss = 0.0;
    for (int i = 0; i < n; ++i) ss += pow(v[i] - mean, 2.0);
    return sqrt(ss / (n - 1));
}
static double qmean(const double *x, const int *q, int nq) { /* bioclim8..11, 16..19 */
    double o = 0.0;
    for (int i = 0; i < nq; ++i) o = o + x[q[i]];
    return o / 72.0;
}
void orc_bioclim_cell(const double *Tz, const double *soilm, int tsteps, const int *wetq, int nwet,
                      const int *dryq, int ndry, const int *hotq, int nhot, const int *colq, int ncol,
                      double *bio) {
    double o = 0.0;
    for (int i = 0; i < 288; ++i) o = o + Tz[i];                                 /* bioclim1 */
    bio[0] = o / 288.0;
    double dtr[12], mon[12];
    int index = 0;
    for (int day = 0; day < 12; day++) {                                         /* bioclim2 */
        double tmx = -273.15, tmn = 273.15;
        for (int hr = 0; hr < 24; hr++) {
            if (Tz[index] > tmx) tmx = Tz[index];
            if (Tz[index] < tmn) tmn = Tz[index];
            index++;
        }
        dtr[day] = tmx - tmn;
    }
    o = 0.0;
    for (int day = 0; day < 12; day++) o = o + dtr[day];
    bio[1] = o / 12;
    index = 0;
    for (int mth = 0; mth < 12; mth++) {                                         /* bioclim4 */
        mon[mth] = 0.0;
        for (int hr = 0; hr < 24; hr++) { mon[mth] = mon[mth] + Tz[index]; index++; }
        mon[mth] = mon[mth] / 24;
    }
    bio[3] = std_dev(mon, 12) * 100.0;
    double tmx = -273.15, tmn = 273.15;
    for (int i = 288; i < 312; i++) if (Tz[i] > tmx) tmx = Tz[i];                /* bioclim5 */
    for (int i = 312; i < 336; i++) if (Tz[i] < tmn) tmn = Tz[i];                /* bioclim6 */
    bio[4] = tmx;
    bio[5] = tmn;
    bio[7] = qmean(Tz, wetq, nwet); bio[8] = qmean(Tz, dryq, ndry);
    bio[9] = qmean(Tz, hotq, nhot); bio[10] = qmean(Tz, colq, ncol);
    double me = 0.0;
    for (int i = 0; i < 288; i++) me = me + soilm[i];                            /* bioclim12 */
    me = me / 288.0;
    bio[11] = me;
    double mx = 0.0, mn = 1.0;
    for (int i = 0; i < tsteps; i++) {                                           /* bioclim13, 14 */
        if (soilm[i] > mx) mx = soilm[i];
        if (soilm[i] < mn) mn = soilm[i];
    }
    bio[12] = mx;
    bio[13] = mn;
    bio[14] = me / std_dev(soilm, tsteps);                                       /* bioclim15 (sic) */
    bio[15] = qmean(soilm, wetq, nwet); bio[16] = qmean(soilm, dryq, ndry);
    bio[17] = qmean(soilm, hotq, nhot); bio[18] = qmean(soilm, colq, ncol);
    bio[6] = bio[4] - bio[5];                                                    /* cpp:3533 */
    bio[2] = bio[1] / bio[6];                                                    /* cpp:3534 */
}

/* cpp:884-929 weatherhgtCpp: temperature, humidity and wind moved from zin / uzin to zout above a short
 * reference canopy, with the diabatic correction of a BigLeafCpp run (dTmx 25, zref 2, maxiter 20, yearG true) */
int orc_weatherhgt(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                   const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                   const double *ws, double zin, double uzin, double zout, double lat, double lon, double *Tz,
                   double *Rh, double *Uz) {
    const double vegp[10] = {0.12, 1, 1, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100.0};
    const double groundp[12] = {0.15, 0.0, 180.0, 0.97, 1.529643, 0.509, 0.06, 0.5422, 5.2, 2.6, 0.419, 0.074};
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *soilm = (double *)malloc(nb);
    double *buf = (double *)calloc(11, nb);
    for (int i = 0; i < n; ++i) soilm[i] = 0.2;
    orc_bigleaf_out bo;
    double **slots[11] = {&bo.Tc, &bo.Tg, &bo.H, &bo.G, &bo.psih, &bo.psim, &bo.phih, &bo.OL, &bo.uf, &bo.RabsG,
                          &bo.albedo};
    for (int q = 0; q < 11; ++q) *slots[q] = buf + (size_t)q * (n > 0 ? n : 1);
    int rc = orc_bigleaf(n, year, month, day, hour, tc, rh, pk, Rsw, Rdif, Rlw, ws, vegp, groundp, soilm, lat, lon, 25,
                         2, 20, 0.5, 0.5, 1, &bo);
    if (rc == 0) {
        double d = orc_zeroplanedis(0.12, 1);
        for (int i = 0; i < n; ++i) {
            double zm = orc_roughlength(0.12, 1, d, bo.psih[i]);
            double zh = 0.2 * zm;
            double lnr = log((zout - d) / zh) / log((zin - d) / zh);
            Tz[i] = (bo.Tc[i] - tc[i]) * (1 - lnr) + tc[i];
            double ea = orc_satvap(tc[i]) * rh[i] / 100;
            double es = orc_satvap(bo.Tc[i]) * sqrt(rh[i] / 100);
            double ez = ea + (es - ea) * (1 - lnr);
            es = orc_satvap(Tz[i]);
            Rh[i] = (ez / es) * 100;
            if (Rh[i] < 0.25 * rh[i]) Rh[i] = 0.25 * rh[i];
            if (Rh[i] > 100.0) Rh[i] = 100.0;
            double lnru = log((zout - d) / zm) / log((uzin - d) / zm);
            Uz[i] = ws[i] * lnru;
        }
    }
    free(soilm); free(buf);
    return rc;
}

/* cpp:931-972 soilmCpp: two-layer daily bucket model; writes n / 24 daily values, returns their number */
int orc_soilm(int n, const double *temp, const double *swdown, const double *lwdown, const double *rainh, double rmu,
              double mult, double pwr, double Smax, double Smin, double Ksat, double a, double *soilm) {
    int nd = n / 24;
    double *rnetd = (double *)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    double *rain = (double *)calloc((size_t)(nd > 0 ? nd : 1), sizeof(double));
    for (int d = 0; d < nd; ++d) {
        double sr = 0.0, sp = 0.0;
        for (int h = 0; h < 24; ++h) {
            int i = d * 24 + h;
            double swrad = (1 - 0.15) * swdown[i];
            double lwout = SB * 0.95 * radem(temp[i]);
            double lwnet = lwout - lwdown[i];
            double rnet = swrad - lwnet;
            if (rnet < 0) rnet = 0;
            sr += rnet;
            sp += rainh[i];
        }
        rnetd[d] = sr / 24;
        rain[d] = sp;
    }
    double s1 = Smax, s2 = Smax;
    if (nd > 0) soilm[0] = Smax;
    for (int i = 1; i < nd; ++i) {
        double sav = (s1 + s2) / 2;
        double dif = s2 - s1;
        s1 = s1 + rmu * rain[i] - mult * rnetd[i];
        double k = Ksat * pow(sav / Smax, pwr);
        s1 = s1 + a * k * dif;
        s2 = s2 - ((a * k * dif) / 10);
        if (s1 > Smax) s1 = Smax;
        if (s2 > Smax) s2 = Smax;
        if (s1 < Smin) s1 = Smin;
        if (s2 < Smin) s2 = Smin;
        soilm[i] = (s1 + s2) / 2;
    }
    free(rnetd); free(rain);
    return nd;
}

/* cpp:5265-5323 pointmprocess: the point-model quantities the grid solver scales from (umu, kp, muGp, dtrp; DDp
 * and T0p are computed but never read by runmicro*Cpp).  dtrp is filled for whole days, 0 beyond. */
void orc_pointmprocess(int n, const double *u2, const double *tc, const double *rh, const double *pk, const double *uf,
                       const double *soilm, const double *RabsG, double zref, double h, double pai, double rho,
                       double Vm, double Vq, double Mc, double *umu, double *kp, double *muGp, double *DDp,
                       double *T0p, double *dtrp) {
    double dp = orc_zeroplanedis(h, pai);
    double zmp = orc_roughlength(h, pai, dp, 0);
    soilc_t sp = soilpfun(Vm, Vq, Mc, rho);
    for (int i = 0; i < n; ++i) {
        double ufps = (KA * u2[i]) / log((zref - dp) / zmp);
        umu[i] = uf[i] / ufps;
        double cs = (2400 * rho / 2.64 + 4180 * soilm[i]);
        double ph = (rho * (1.0 - soilm[i]) + soilm[i]) * 1000;
        double c2 = 1.06 * rho * soilm[i];
        kp[i] = sp.c1 + c2 * soilm[i] - (sp.c1 - sp.c4) * exp(-pow(sp.c3 * soilm[i], 4.0));
        double kap = kp[i] / (cs * ph);
        muGp[i] = pow(2.0 * kap / OMDY, 0.5);
        DDp[i] = pow(2.0 * kap / OMDY, 0.5);
        double gHa = (0.4 * 43.0 * ufps) / log((zref - dp) / zmp);
        double es = orc_satvap(tc[i]);
        double ea = es * rh[i] / 100.0;
        T0p[i] = penman(RabsG[i], gHa, gHa, tc[i], tc[i], pk[i], ea, 0.97, 0.0, 1.0);
        dtrp[i] = 0.0;
    }
    int nd = n / 24;
    for (int d = 0; d < nd; ++d) {
        double mx = T0p[d * 24], mn = T0p[d * 24];
        for (int j = 1; j < 24; ++j) {
            mx = fmax(mx, T0p[d * 24 + j]);
            mn = fmin(mn, T0p[d * 24 + j]);
        }
        for (int j = 0; j < 24; ++j) dtrp[d * 24 + j] = mx - mn;
    }
}
