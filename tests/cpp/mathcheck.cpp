// tests/cpp/mathcheck.cpp -- CPU check of the division shortcuts of tracer_amd/csrc/pmath.h against
// the plain C operators they replace (IEEE-754 `/`, `%`): built with g++, no GPU needed.
//   div_by_uniform(a, d, RN(1/d)) == a / d   for every pixel coordinate of many image sizes and for
//                                            random jitter numerators 2r-1, r = k * 2^-25
//   udivmod(n, make_udiv(d))      == n / d, n % d
#include "../../tracer_amd/csrc/pmath.h"
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }

int main(int argc, char** argv) {
  const long n_random = argc > 1 ? atol(argv[1]) : (1l << 24);
  long bad = 0, checked = 0;
  // every x < w for a spread of widths
  const uint32_t widths[] = {1, 2, 3, 5, 7, 31, 64, 100, 255, 256, 257, 333, 512, 640, 720, 1000, 1023, 1024, 1025,
                             1080, 1920, 2047, 2048, 3840, 4096, 7680, 8191, 16384, 65535, 65536};
  for (uint32_t w : widths) {
    const float fw = (float)w, rw = 1.0f / fw;
    for (uint32_t x = 0; x < w; ++x) {
      const float a = (float)x;
      if (trc::div_by_uniform(a, fw, rw) != a / fw) ++bad;
      ++checked;
    }
  }
  // random widths, random coordinates and jitter numerators
  for (long i = 0; i < n_random; ++i) {
    const uint32_t w = 1u + rnd() % ((i & 1) ? 65536u : 16777215u);
    const float fw = (float)w, rw = 1.0f / fw;
    const float a = (float)(rnd() % w);
    if (trc::div_by_uniform(a, fw, rw) != a / fw) ++bad;
    const float r = (float)(rnd() >> 7) * 0x1p-25f + 0x1p-25f;     // (0, 1]
    const float j = 2.0f * r - 1.0f;
    const float got = trc::div_by_uniform(j, fw, rw), want = j / fw;
    if (!(got == want) || (got == 0.0f && std::signbit(got) != std::signbit(want))) ++bad;
    checked += 2;
  }
  long bad_u = 0;
  for (long i = 0; i < n_random; ++i) {
    uint32_t d = rnd(); if ((i & 3) == 0) d = 1u + d % 4096u; else if ((i & 3) == 1) d = 1u + d % (1u << 22); if (d == 0) d = 1;
    uint32_t n = rnd(); if ((i & 7) == 0) n = 0xFFFFFFFFu - (n & 15u);
    uint32_t rem; const uint32_t q = trc::udivmod(n, trc::make_udiv(d), rem);
    if (q != n / d || rem != n % d) ++bad_u;
  }
  for (uint32_t d : {1u, 2u, 3u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu, 262144u})
    for (uint32_t n : {0u, 1u, d - 1u, d, d + 1u, 0xFFFFFFFEu, 0xFFFFFFFFu}) {
      uint32_t rem; const uint32_t q = trc::udivmod(n, trc::make_udiv(d), rem);
      if (q != n / d || rem != n % d) ++bad_u;
    }
  printf("div_by_uniform: %ld checked, %ld mismatches; udivmod: %ld random + edge cases, %ld mismatches\n",
         checked, bad, n_random, bad_u);
  return (bad || bad_u) ? 1 : 0;
}
