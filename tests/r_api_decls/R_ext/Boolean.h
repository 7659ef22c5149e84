/* SYNTAX-CHECK STAND-IN (see README.md) */
#ifndef MCF_TEST_R_BOOLEAN_H
#define MCF_TEST_R_BOOLEAN_H
typedef enum { FALSE = 0, TRUE } Rboolean;
#endif
