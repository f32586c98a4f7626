// tracer_amd/csrc/pmath.h -- sin/cos/acos/gamma-pow for host and gfx950 device code.
//
// The reference reaches the host libm through glm::sin/cos/acos/pow
// (externals/glm/core/func_trigonometric.inl:70,87,134; func_exponential.inl:41). glibc's
// float functions are not correctly rounded and OCML's differ from them by 1-2 ulp, so no
// device routine can be bit-identical to the reference's host build here.
// These routines are therefore defined operation by operation -- IEEE-754 add, mul, div,
// sqrt, fma, round-to-nearest-integer only -- so that gfx950 and any IEEE host evaluate them
// to the same bits; they stay within 1-3 ulp of the libm results, which moves a sampled
// direction by ~1e-7 (measured effect on radiance: tests/test_parity_gpu.py).
#pragma once
#include "vecmath.h"
#include <cmath>
#include <cstring>

#if defined(__HIP_DEVICE_COMPILE__)
#include "crmath.h"
#endif

#pragma clang fp contract(off)

namespace trc {

// sqrtf: the short correctly-rounded sequence of crmath.h on the device, libm elsewhere (same values).
TRC_HD float pm_sqrtf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return sqrt_active(x);
#else
  return sqrtf(x);
#endif
}

// ---- division by a wave-uniform divisor --------------------------------------------------------
// a / d, correctly rounded, from rd = RN(1/d) (computed once on the host): the quotient estimate and
// the two fma residual corrections of the usual IEEE expansion (what hipcc emits between its
// v_div_scale and v_div_fixup steps), valid when nothing needs rescaling:
//   a == 0 or 2^-60 <= |a| <= 2^60,  2^-60 <= |d| <= 2^60.
// The camera code divides pixel coordinates (integers below 2^24) and 2r-1 with r a multiple of
// 2^-25 in (0, 1] by the image width / height, so the precondition holds by construction there.
// Checked against `/` for every x < w of many widths and 2^26 random numerators (tests/cpp/mathcheck).
TRC_HD float div_by_uniform(float a, float d, float rd) {
  float q = a * rd;
  q = fmaf(fmaf(-d, q, a), rd, q);
  return fmaf(fmaf(-d, q, a), rd, q);
}

// n / d and n % d for a wave-uniform d from magic = floor((2^32 - 1) / d): the high product
// undershoots the quotient by at most 2, fixed by two conditional subtractions.
struct UDiv { uint32_t d, magic; };
TRC_HD UDiv make_udiv(uint32_t d) { UDiv u; u.d = d; u.magic = d ? 0xFFFFFFFFu / d : 0u; return u; }
TRC_HD uint32_t umulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((unsigned long long)a * b) >> 32);
#endif
}
TRC_HD uint32_t udivmod(uint32_t n, UDiv u, uint32_t& rem) {
  uint32_t q = umulhi32(n, u.magic);
  uint32_t r = n - q * u.d;
  if (r >= u.d) { r -= u.d; ++q; }
  if (r >= u.d) { r -= u.d; ++q; }
  rem = r;
  return q;
}

// sin and cos of x (callers pass [0, 2*pi]; accuracy holds to |x| ~ 1e4)
TRC_HD void pm_sincosf(float x, float& sn, float& cs) {
  // Cody-Waite reduction by pi/2 in three parts
  float q = rintf(x * 0.636619772367581343f);
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188216e-8f, r);
  int n = (int)q;
  float z = r * r;
  // minimax polynomials on [-pi/4, pi/4]
  float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  float s = fmaf(r * z, ps, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  float so = (n & 1) ? c : s;
  float co = (n & 1) ? s : c;
  if (n & 2) so = -so;
  if ((n + 1) & 2) co = -co;
  sn = so; cs = co;
}

TRC_HD float pm_asin_poly(float x) { // |x| <= 0.5
  float z = x * x;
  float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = fmaf(p, z, 4.5470025998e-2f);
  p = fmaf(p, z, 7.4953002686e-2f);
  p = fmaf(p, z, 1.6666752422e-1f);
  return fmaf(x * z, p, x);
}

// acos(x) = 2 asin(sqrt((1 - x) / 2)) for x > 1/2, pi - 2 asin(sqrt((1 + x) / 2)) for x < -1/2, pi/2 - asin(x)
// between. Both tails take the root of fmaf(-0.5, |x|, 0.5) -- for x < -1/2 that is the fmaf(0.5, x, 0.5) of
// the defining formula, the same product and sum -- so one square root and one polynomial serve all three
// cases (lanes of a wavefront fall into all of them); pm_sqrtf is sqrtf, correctly rounded either way.
TRC_HD float pm_acosf(float x) {
  // |x| and the sign instead of three comparisons of x (x < 0 ? -x : x differs from |x| only for -0, whose fmaf below
  // is 0.5 either way; v_cmp is the expensive instruction of the VALU's simple classes)
  const float ax = __builtin_fabsf(x);
  const bool tail = ax > 0.5f, neg = x < 0.0f;
  const float s = pm_sqrtf(fmaf(-0.5f, ax, 0.5f));
  const float p = pm_asin_poly(tail ? s : x);
  if (tail) return neg ? fmaf(-2.0f, p, 3.14159274101257324f) : 2.0f * p;      // x < -1/2 : x > 1/2
  return 1.57079637050628662f - p;
}

TRC_HD double pm_bits2d(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)b);
#else
  double d; memcpy(&d, &b, 8); return d;
#endif
}
TRC_HD uint64_t pm_d2bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(d);
#else
  uint64_t b; memcpy(&b, &d, 8); return b;
#endif
}

// x^y for the display gamma (y = 1/2.2 > 0), evaluated in binary64 from a float argument.
TRC_HD float pm_powf(float xf, float yf) {
  if (!(xf > 0.0f)) return (xf == 0.0f) ? 0.0f : __builtin_nanf("");
  if (xf > 3.0e38f) return xf;
  double x = (double)xf;
  uint64_t b = pm_d2bits(x);
  int e = (int)((b >> 52) & 0x7FF) - 1023;
  b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m = pm_bits2d(b);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  double p = 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  p = p * s2 + 1.0;
  double l2 = (double)e + (2.0 * s * p) * 1.4426950408889634;
  double r = (double)yf * l2;
  if (r < -160.0) return 0.0f;
  if (r > 130.0) return __builtin_inff();
  double n = rint(r);
  double f = (r - n) * 0.6931471805599453;
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
  long long ni = (long long)n;
  double sc = pm_bits2d((uint64_t)(ni + 1023) << 52);
  return (float)(q * sc);
}

} // namespace trc
