/* oracle/pmath.h -- TEST INFRASTRUCTURE. "Portable math": the sin/cos/acos/gamma-pow
 * algorithms the HIP kernels implement (tracer_amd/csrc/pmath.h), restated for the CPU so the
 * oracle can be run in ORC_MATH_PORTABLE mode and compared BIT-EXACTLY with the GPU.
 *
 * Why: the reference calls glm::sin/cos/acos/pow -> host libm (func_trigonometric.inl:70-123,
 * func_exponential.inl:41). glibc's sinf/cosf/acosf/powf are not correctly rounded and no
 * GPU math library reproduces them bit for bit, so the libm-mode oracle can
 * only be matched to ~1 ulp per call. Every operation below is an IEEE-754 binary32/binary64
 * add, mul, div, sqrt, fma or round-to-nearest-integer in a fixed order, which x86-64 and
 * gfx950 evaluate identically.
 *
 * Accuracy (vs correctly-rounded, measured in tests/test_pmath.py): sin/cos <= 2 ulp on
 * [0, 2*pi], acos <= 3 ulp on [-1, 1], gamma pow <= 1 ulp.
 */
#ifndef TRC_ORACLE_PMATH_H
#define TRC_ORACLE_PMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* sin and cos of x (|x| up to ~1e4 keeps full accuracy; callers pass [0, 2*pi]). */
static inline void pm_sincosf(float x, float* sn, float* cs)
{
  /* Cody-Waite reduction by pi/2 in three parts */
  float q = rintf(x * 0.636619772367581343f);         /* 2/pi */
  float r = fmaf(q, -1.5703125f, x);                   /* pi/2 high  (exact product) */
  r = fmaf(q, -4.837512969970703125e-4f, r);           /* pi/2 mid   */
  r = fmaf(q, -7.54978995489188216e-8f, r);            /* pi/2 low   */
  int n = (int)q;
  float z = r * r;
  /* minimax polynomials on [-pi/4, pi/4] (Cephes single-precision coefficients) */
  float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  float s = fmaf(r * z, ps, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  float so = (n & 1) ? c : s;
  float co = (n & 1) ? s : c;
  if (n & 2) so = -so;
  if ((n + 1) & 2) co = -co;
  *sn = so; *cs = co;
}

static inline float pm_asin_poly(float x) /* |x| <= 0.5 */
{
  float z = x * x;
  float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = fmaf(p, z, 4.5470025998e-2f);
  p = fmaf(p, z, 7.4953002686e-2f);
  p = fmaf(p, z, 1.6666752422e-1f);
  return fmaf(x * z, p, x);
}

static inline float pm_acosf(float x)
{
  if (x > 0.5f) {
    float s = sqrtf(fmaf(-0.5f, x, 0.5f));
    return 2.0f * pm_asin_poly(s);
  } else if (x < -0.5f) {
    float s = sqrtf(fmaf(0.5f, x, 0.5f));
    return fmaf(-2.0f, pm_asin_poly(s), 3.14159274101257324f);
  } else {
    return 1.57079637050628662f - pm_asin_poly(x);
  }
}

/* x^y for the display gamma (raytrace.cu:173), evaluated in binary64; x is a float. */
static inline float pm_powf(float xf, float yf)
{
  if (!(xf > 0.0f)) return (xf == 0.0f) ? 0.0f : NAN;   /* y > 0 here: pow(+-0,y)=0, pow(<0,y)=NaN */
  if (xf > 3.0e38f) return xf;                          /* +inf */
  double x = (double)xf;
  uint64_t b; memcpy(&b, &x, 8);
  int e = (int)((b >> 52) & 0x7FF) - 1023;
  b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m; memcpy(&m, &b, 8);                           /* [1,2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }   /* [sqrt(.5), sqrt(2)) */
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  /* atanh series: log(m) = 2 (s + s^3/3 + ... + s^17/17) */
  double p = 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  p = p * s2 + 1.0;
  double l2 = (double)e + (2.0 * s * p) * 1.4426950408889634; /* log2(x) */
  double r = (double)yf * l2;
  if (r < -160.0) return 0.0f;
  if (r > 130.0) return INFINITY;
  double n = rint(r);
  double f = (r - n) * 0.6931471805599453;               /* |f| <= 0.3466 */
  double q = 1.0 / 479001600.0;
  q = q * f + 1.0 / 39916800.0;
  q = q * f + 1.0 / 3628800.0;
  q = q * f + 1.0 / 362880.0;
  q = q * f + 1.0 / 40320.0;
  q = q * f + 1.0 / 5040.0;
  q = q * f + 1.0 / 720.0;
  q = q * f + 1.0 / 120.0;
  q = q * f + 1.0 / 24.0;
  q = q * f + 1.0 / 6.0;
  q = q * f + 0.5;
  q = q * f + 1.0;
  q = q * f + 1.0;
  int64_t ni = (int64_t)n;
  uint64_t sb = (uint64_t)(ni + 1023) << 52;             /* 2^n, n in [-160,130] */
  double sc; memcpy(&sc, &sb, 8);
  return (float)(q * sc);
}

#endif
