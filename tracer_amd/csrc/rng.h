// tracer_amd/csrc/rng.h -- counter-based RNG replacing the reference's cuRAND fill.
//
// Reference: src/raytrace.cu:203-206 creates a cuRAND generator per call, seeds it with
// time*100 and writes 3 uniform floats per pixel into rand_d, which the kernels then read
// (raytrace.cu:81-82, :92, :157). Here `rand_d[index]` of iteration `iter` is a pure function
//   rand3(seed, index, iter, dim) = u01(Philox-4x32-10(ctr = {index, iter, dim, 0},
//                                                     key = {seed_lo, seed_hi}))[0..2]
// so no buffer is written or gathered, and every sample is reproducible on any GPU count.
// u01 is cuRAND's (0,1] mapping x*2^-32 + 2^-33 in fp32 (two roundings, no contraction).
#pragma once
#include "vecmath.h"

#pragma clang fp contract(off)

namespace trc {

TRC_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

TRC_HD float u01(uint32_t x) {
  float m = (float)x * 2.3283064365386963e-10f;
  return m + 1.1641532182693481e-10f;
}

TRC_HD f3 rand3(uint32_t seed_lo, uint32_t seed_hi, uint32_t index, uint32_t iter, uint32_t dim) {
  uint32_t o[4];
  philox4x32_10(index, iter, dim, 0u, seed_lo, seed_hi, o);
  return mk3(u01(o[0]), u01(o[1]), u01(o[2]));
}

} // namespace trc
