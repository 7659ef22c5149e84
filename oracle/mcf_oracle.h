/* mcf_oracle.h — TEST INFRASTRUCTURE (see mcf_oracle.c).  Public surface of the
 * CPU restatement used as the parity checker and as bench.py's cpu_baseline. */
#ifndef MCF_ORACLE_H
#define MCF_ORACLE_H
#include <stdint.h>
#include "../include/mcf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_solmodel { double zend, zenr, azid, azir; } orc_solmodel; /* hdr:5-10  */
typedef struct orc_kstruct { double k, kd, Kc; } orc_kstruct;                /* hdr:11-15 */

double orc_na_real(void);
int orc_julday(int year, int month, int day);
orc_solmodel orc_solposition(double lat, double lon, int year, int month, int day, double lt);
double orc_solarindex(double slope, double aspect, double zend, double azid, int shadowmask);
orc_kstruct orc_cank(double zenr, double x, double si);
double orc_zeroplanedis(double h, double pai);
double orc_roughlength(double h, double pai, double d, double psi_h);
double orc_satvap(double tc);
double orc_soild(double soilm, double Smin, double Smax, double tadd);
void orc_soild_tadd(const double *twi, int64_t n_cells, int64_t rows, int64_t cols, double tfact,
                    double *tadd);
void orc_set_twi_mean_override(double mean, int enable);
void orc_man(const double *x, int m, int n, double *z);
void orc_tbelowground(double reqhgt, const double *Tg, const double *Tgp, const double *Tbp, int tsteps,
                      double meanD, double mat, int hiy, int complete, double *Tz);
/* runmicro1Cpp / runmicro2Cpp restatement, same argument structs as the product ABI */
int orc_run_grid(const mcf_grid_inputs *in, const mcf_options *opt, mcf_outputs *out);

#ifdef __cplusplus
}
#endif
#endif
