/*
 * pddp_oracle.c - instantiates the CPU oracle for double and float.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP kernels (see
 * pddp_oracle.h).  Built by oracle/Makefile into oracle/_build/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pddp_oracle.h"

#define REAL double
#define FN(x) x##_f64
#define SQRT_(x) sqrt(x)
#define SIN_(x) sin(x)
#define COS_(x) cos(x)
#define FABS_(x) fabs(x)
#include "pddp_oracle_impl.inc"
#undef REAL
#undef FN
#undef SQRT_
#undef SIN_
#undef COS_
#undef FABS_

#define REAL float
#define FN(x) x##_f32
#define SQRT_(x) sqrtf(x)
#define SIN_(x) sinf(x)
#define COS_(x) cosf(x)
#define FABS_(x) fabsf(x)
#include "pddp_oracle_impl.inc"
#undef REAL
#undef FN
#undef SQRT_
#undef SIN_
#undef COS_
#undef FABS_
