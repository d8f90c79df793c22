/* oracle/fsm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 * Instantiates the CPU restatement (fsm_oracle_impl.h) for float and double.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math: the
 * arithmetic must round exactly like the reference compiled with g++). */
#include "fsm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#define REAL float
#define SFX(name) name##_f32
#define FABS(x) fabsf(x)
#define REAL_MAX FLT_MAX
#define REAL_EPS FLT_EPSILON
#include "fsm_oracle_impl.h"
#undef REAL
#undef SFX
#undef FABS
#undef REAL_MAX
#undef REAL_EPS

#define REAL double
#define SFX(name) name##_f64
#define FABS(x) fabs(x)
#define REAL_MAX DBL_MAX
#define REAL_EPS DBL_EPSILON
#include "fsm_oracle_impl.h"
