/* SYNTAX-CHECK STAND-IN (see README.md) — declarations only */
#ifndef MCF_TEST_R_H
#define MCF_TEST_R_H
#include <stddef.h>
#include <R_ext/Boolean.h>
void Rf_error(const char *, ...) __attribute__((noreturn));
void Rf_warning(const char *, ...);
#endif
