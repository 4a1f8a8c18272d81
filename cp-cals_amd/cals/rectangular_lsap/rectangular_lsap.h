/* extern/rectangular_lsap/rectangular_lsap.h of HPAC/CP-CALS (SciPy's assignment solver), same
 * contract: ROW-major nr x nc cost; (a[k], b[k]), k < min(nr, nc), = assigned (row, column) pairs
 * sorted by row.  Implemented in libcals.so (cals/utils.cpp) -- own code for the same published
 * algorithm (Crouse 2016) with the same tie rules; pinned against the reference's file in the tests. */
#ifndef CALS_AMD_RECTANGULAR_LSAP_H
#define CALS_AMD_RECTANGULAR_LSAP_H

#define RECTANGULAR_LSAP_INFEASIBLE -1
#define RECTANGULAR_LSAP_INVALID -2

#ifdef __cplusplus
extern "C" {
#endif

#include <stdbool.h>
#include <stdint.h>

int solve_rectangular_linear_sum_assignment(intptr_t nr, intptr_t nc, double *input_cost, bool maximize,
                                            int64_t *a, int64_t *b);

#ifdef __cplusplus
}
#endif
#endif
