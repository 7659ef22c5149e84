/* oracle_unit.c — TEST INFRASTRUCTURE: single translation unit of the CPU oracle
 * (the grid-solver restatement plus the point-model functions that the reference's
 * own tests drive, and the snow branch; the latter two reuse the former's static helpers). */
#include "mcf_oracle.c"
#include "pointmodel.c"
#include "snow_oracle.h"
#include "snow_oracle.c"

#ifdef ORC_COVERAGE
/* gcov build (tools/oracle_branch_coverage.py): libgcov's dump entry is hidden, re-export it */
void __gcov_dump(void);
void orc_cov_dump(void) { __gcov_dump(); }
#endif
