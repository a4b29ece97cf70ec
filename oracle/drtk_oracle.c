/* TEST INFRASTRUCTURE ONLY -- see drtk_oracle.h.  Plain C restatement of the reference's CPU hot
 * path; the generic body lives in drtk_oracle_body.inc and is instantiated for float and double.
 * Build: oracle/build.py  (gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC). */
#include "drtk_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int drtk_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* nthreads: <=0 -> all available, otherwise clamp to what OpenMP offers. */
static int drtk_oracle_resolve_threads(int nthreads) {
  const int mx = drtk_oracle_max_threads();
  if (nthreads <= 0) return mx;
  return nthreads < mx ? nthreads : mx;
}

/* tests/f64_distance.py: accumulate |term| instead of term in the backward functions (see DRTK_MAG in the body). */
static int drtk_oracle_abs_accumulate = 0;
void drtk_oracle_set_abs_accumulate(int on) { drtk_oracle_abs_accumulate = on != 0; }

#define REAL float
#define SFX f32
#define REAL_EPS 1e-8f /* cuda_math_helper.h:62-64 */
#define REAL_SQRT sqrtf
#define REAL_FABS fabsf
#define REAL_MAX_VALUE FLT_MAX
#include "drtk_oracle_body.inc"
#undef REAL_MAX_VALUE
#define M_SQRT sqrtf
#define M_FLOOR floorf
#define M_CEIL ceilf
#define M_LOG2 log2f
#define M_EXP2 exp2f
#define M_FABS fabsf
#define M_FMOD fmodf
#include "drtk_oracle_mipmap.inc"
#undef M_SQRT
#undef M_FLOOR
#undef M_CEIL
#undef M_LOG2
#undef M_EXP2
#undef M_FABS
#undef M_FMOD
#undef REAL
#undef SFX
#undef REAL_EPS
#undef REAL_SQRT
#undef REAL_FABS

#define REAL double
#define SFX f64
#define REAL_EPS 1e-16 /* cuda_math_helper.h:67-69 */
#define REAL_SQRT sqrt
#define REAL_FABS fabs
#define REAL_MAX_VALUE DBL_MAX
#include "drtk_oracle_body.inc"
#undef REAL_MAX_VALUE
#define M_SQRT sqrt
#define M_FLOOR floor
#define M_CEIL ceil
#define M_LOG2 log2
#define M_EXP2 exp2
#define M_FABS fabs
#define M_FMOD fmod
#include "drtk_oracle_mipmap.inc"
