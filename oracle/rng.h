/* oracle/rng.h -- TEST INFRASTRUCTURE (part of the CPU oracle, never shipped).
 *
 * Counter-based RNG that is *injected* into the reference algorithm in place of
 * cuRAND (reference src/raytrace.cu:203-206 fills `rand_d` with 3 uniform floats
 * per pixel per iteration; cuRAND is not in /root/reference and not in this image,
 * so its XORWOW stream cannot be reproduced -> "matched RNG stream" per
 * BASELINE.json means: both sides draw from the function below).
 *
 * Generator: Philox-4x32-10 (Salmon et al., SC'11, "Parallel random numbers: as
 * easy as 1, 2, 3"), restated from the published algorithm; pinned by the
 * Random123 known-answer vectors in tests/test_rng.py.
 *
 *   counter = (index, iteration, dimension, 0)   key = (seed_lo, seed_hi)
 *   rand[index] at iteration `it`  :=  u01(out[0]), u01(out[1]), u01(out[2])
 *
 * u01 follows cuRAND's `_curand_uniform`: x * 2^-32 + 2^-33 evaluated in fp32,
 * giving the (0,1] range the reference's kernels were written against
 * (SURVEY.md Appendix A.10).
 */
#ifndef TRC_ORACLE_RNG_H
#define TRC_ORACLE_RNG_H

#include <stdint.h>

#define ORC_PHILOX_M0 0xD2511F53u
#define ORC_PHILOX_M1 0xCD9E8D57u
#define ORC_PHILOX_W0 0x9E3779B9u
#define ORC_PHILOX_W1 0xBB67AE85u

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)ORC_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)ORC_PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += ORC_PHILOX_W0; k1 += ORC_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* (0,1]: fp32 multiply then fp32 add, two roundings (no contraction). */
static inline float orc_u01(uint32_t x)
{
  volatile float m = (float)x * 2.3283064365386963e-10f; /* 2^-32 */
  return m + 1.1641532182693481e-10f;                    /* 2^-33 */
}

/* The injected stand-in for `rand_d[index]` of iteration `iter`. */
static inline void orc_rand3(uint64_t seed, uint32_t index, uint32_t iter, uint32_t dim, float r[3])
{
  uint32_t ctr[4] = { index, iter, dim, 0u };
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  uint32_t o[4];
  orc_philox4x32_10(ctr, key, o);
  r[0] = orc_u01(o[0]); r[1] = orc_u01(o[1]); r[2] = orc_u01(o[2]);
}

#endif
