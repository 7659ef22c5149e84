/* oracle_unit.c — TEST INFRASTRUCTURE: single translation unit of the CPU oracle
 * (the grid-solver restatement plus the point-model functions that the reference's
 * own tests drive; the latter reuse the former's static helpers). */
#include "mcf_oracle.c"
#include "pointmodel.c"
