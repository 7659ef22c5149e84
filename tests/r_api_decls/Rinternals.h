/* SYNTAX-CHECK STAND-IN (see README.md) — the subset of R's C API r/mcfhip_glue.c uses, declarations only */
#ifndef MCF_TEST_RINTERNALS_H
#define MCF_TEST_RINTERNALS_H
#include <stddef.h>
#include <R_ext/Boolean.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol;
extern double R_NaReal;
extern int R_NaInt;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
int TYPEOF(SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
double *REAL(SEXP);
int *INTEGER(SEXP);
int *LOGICAL(SEXP);
const char *R_CHAR(SEXP);
#define CHAR(x) R_CHAR(x)
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_duplicate(SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_mkChar(const char *);
SEXP Rf_install(const char *);
SEXP Rf_GetOption1(SEXP);
double Rf_asReal(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
Rboolean Rf_isNull(SEXP);
SEXP Rf_ScalarLogical(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_asChar(SEXP);
/* external pointers (Writing R Extensions 5.13) */
typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
#define allocVector Rf_allocVector
#define coerceVector Rf_coerceVector
#define duplicate Rf_duplicate
#define getAttrib Rf_getAttrib
#define setAttrib Rf_setAttrib
#define mkChar Rf_mkChar
#define install Rf_install
#define GetOption1 Rf_GetOption1
#define asReal Rf_asReal
#define asInteger Rf_asInteger
#define asLogical Rf_asLogical
#define isNull Rf_isNull
#define ScalarLogical Rf_ScalarLogical
#define ScalarReal Rf_ScalarReal
#define ScalarInteger Rf_ScalarInteger
#define allocMatrix Rf_allocMatrix
#define asChar Rf_asChar
#endif
