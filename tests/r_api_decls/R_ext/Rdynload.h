/* SYNTAX-CHECK STAND-IN (see README.md) */
#ifndef MCF_TEST_RDYNLOAD_H
#define MCF_TEST_RDYNLOAD_H
#include <R_ext/Boolean.h>
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
