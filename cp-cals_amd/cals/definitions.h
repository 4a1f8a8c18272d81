// What include/definitions.h of HPAC/CP-CALS gives its front-ends: the index type dim_t and the two
// statement macros DEBUG(...) / TIME(...).
#ifndef CALS_AMD_DEFINITIONS_H
#define CALS_AMD_DEFINITIONS_H

#include <cstddef>

using dim_t = size_t;

// DEBUG(stmt): stmt in debug builds, an empty statement under NDEBUG
#ifdef NDEBUG
#define DEBUG(exp) \
  do {             \
  } while (0);
#else
#define DEBUG(exp) exp
#endif

// TIME(stmt): stmt when the caller is compiled with -DWITH_TIME=1.  The reference compiles its per-iteration
// timer matrices in only then (CMakeLists.txt:195,201); here the CalsReport / AlsReport fields always exist and
// WITH_TIME only sets the default of CalsParams::with_time / AlsParams::with_time (whether a run fills them).
#if defined(WITH_TIME) && WITH_TIME
#define TIME(exp) exp
#define CALS_AMD_WITH_TIME_DEFAULT true
#else
#define TIME(exp) ;
#define CALS_AMD_WITH_TIME_DEFAULT false
#endif

#endif
