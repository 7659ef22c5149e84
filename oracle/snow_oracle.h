/* snow_oracle.h — TEST INFRASTRUCTURE (see snow_oracle.c). */
#ifndef MCF_ORACLE_SNOW_H
#define MCF_ORACLE_SNOW_H
#include <stdint.h>
#include "../include/mcf.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_pointsnow_out { /* caller-allocated; cpp:4151-4168 */
    double *Tc, *Tg, *sdepc, *sdepg, *sdenc, *sdeng, *G, *RswabsG, *RlwabsG, *tr, *umu, *sublmelt, *tempmelt,
        *rainmelt, *sstemp;  /* sdepc, sdepg: tsteps + 1; the rest: tsteps */
    double mxdif;
    int iters;
} orc_pointsnow_out;

int orc_pointmodelsnow(int tsteps, const int *year, const int *month, const int *day, const double *hour,
                       const double *tc, const double *rh, const double *pk, const double *Rsw,
                       const double *Rdif, const double *Rlw, const double *u2, const double *prec,
                       const double *vegp, const double *other, int snowenv, double tol, double maxiter,
                       orc_pointsnow_out *o);
void orc_snowalb(const double *prec, int tsteps, double *alb); /* snowalbCpp cpp:3752-3771 */
int orc_gridmodelsnow(const mcf_snow_inputs *in, mcf_snowmodel_out *out);
int orc_gridmicrosnow(const mcf_snow_inputs *in, const mcf_snowm *sm, double reqhgt, double mat,
                      const int32_t *outsel, mcf_outputs *micro);
#ifdef __cplusplus
}
#endif
#endif
