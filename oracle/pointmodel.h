/* pointmodel.h — TEST INFRASTRUCTURE (see pointmodel.c). */
#ifndef MCF_ORACLE_POINTMODEL_H
#define MCF_ORACLE_POINTMODEL_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_bigleaf_out {   /* caller-allocated arrays of length n; cpp:867-879 */
    double *Tc, *Tg, *H, *G, *psih, *psim, *phih, *OL, *uf, *RabsG, *albedo;
    double err;
    int iters;
} orc_bigleaf_out;

typedef struct orc_wrapper_out {   /* caller-allocated arrays of length n; cpp:5908-5950 */
    double *Tz, *tleaf, *rh, *uz, *Rdirdown, *Rdifdown, *Rswup, *Rlwdown, *Rlwup, *soilm;
} orc_wrapper_out;

int orc_bigleaf(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                const double *wspeed, const double *vegp, const double *groundp, const double *soilm, double lat,
                double lon, double dTmx, double zref, int maxiter, double bwgt, double tol, int yearG,
                orc_bigleaf_out *o);
void orc_clearskyrad(int n, const int *year, const int *month, const int *day, const double *lt, double lat,
                     double lon, const double *tc, const double *rh, const double *pk, double *Ic);
void orc_solpositionv(int n, const int *year, const int *month, const int *day, const double *lt, double lat,
                      double lon, double slope, double aspect, double *zen, double *azi, double *si);
int orc_wrapper(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                const double *wspeed, const double *BL_Tg, const double *BL_G, const double *BL_uf,
                const double *vegp, const double *groundp, double reqhgt, double zref, double lat, double lon,
                orc_wrapper_out *o);
void orc_bioclim_cell(const double *Tz, const double *soilm, int tsteps, const int *wetq, int nwet,
                      const int *dryq, int ndry, const int *hotq, int nhot, const int *colq, int ncol,
                      double *bio);
int orc_weatherhgt(int n, const int *year, const int *month, const int *day, const double *hour, const double *tc,
                   const double *rh, const double *pk, const double *Rsw, const double *Rdif, const double *Rlw,
                   const double *ws, double zin, double uzin, double zout, double lat, double lon, double *Tz,
                   double *Rh, double *Uz);
int orc_soilm(int n, const double *temp, const double *swdown, const double *lwdown, const double *rainh, double rmu,
              double mult, double pwr, double Smax, double Smin, double Ksat, double a, double *soilm);
void orc_pointmprocess(int n, const double *u2, const double *tc, const double *rh, const double *pk, const double *uf,
                       const double *soilm, const double *RabsG, double zref, double h, double pai, double rho,
                       double Vm, double Vq, double Mc, double *umu, double *kp, double *muGp, double *DDp,
                       double *T0p, double *dtrp);
#ifdef __cplusplus
}
#endif
#endif
