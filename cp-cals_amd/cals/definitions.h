// include/definitions.h of HPAC/CP-CALS: the index type and the two statement macros the front-ends use.
#ifndef CALS_AMD_DEFINITIONS_H
#define CALS_AMD_DEFINITIONS_H

#include <cstddef>

#ifndef NDEBUG
#define DEBUG(exp) exp
#else
#define DEBUG(exp) \
  do {             \
  } while (0);
#endif

// The reference compiles its per-iteration timer matrices in only with -DWITH_TIME=1
// (CMakeLists.txt:195,201).  Here the CalsReport fields always exist; WITH_TIME only sets the default of
// CalsParams::with_time (whether a run fills them).
#if WITH_TIME
#define TIME(exp) exp
#define CALS_AMD_WITH_TIME_DEFAULT true
#else
#define TIME(exp) ;
#define CALS_AMD_WITH_TIME_DEFAULT false
#endif

typedef size_t dim_t;

#endif
