/* oracle/pm_interpose.c -- TEST INFRASTRUCTURE. Linked ONLY into oracle/_ref/libtracer_ref_pm.so (oracle/Makefile),
 * with -Wl,-Bsymbolic-functions, so that the reference's own code -- glm::sin / cos / acos / pow, which end in
 * calls to sinf / cosf / acosf / powf (externals/glm/core/func_trigonometric.inl:70-123, func_exponential.inl:26) --
 * reaches the portable routines of oracle/pmath.h instead of glibc's. SURVEY.md 7.2-2 option (a): the reference
 * host build and the HIP kernels then share every arithmetic step, and their films can be compared bit for bit at
 * any size; the pure glibc effect is what remains between this build and oracle/_ref/libtracer_ref.so.
 * tanf (Renderer.cpp:168, host only, also host-side in the product) is left to glibc. */
#define _GNU_SOURCE
#include <math.h>
#include "pmath.h"

/* g++ merges sin(x) and cos(x) of one argument into one sincosf call: the reference build imports only that name */
void sincosf(float x, float* s, float* c) { pm_sincosf(x, s, c); }
float sinf(float x) { float s, c; pm_sincosf(x, &s, &c); return s; }
float cosf(float x) { float s, c; pm_sincosf(x, &s, &c); return c; }
float acosf(float x) { return pm_acosf(x); }
float powf(float x, float y) { return pm_powf(x, y); }
