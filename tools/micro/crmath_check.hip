// Exhaustive check (all 2^32 float bit patterns) of short correctly-rounded sqrt / reciprocal
// sequences against the compiler's IEEE sqrtf and 1.0f/x on the actual gfx950 hardware.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#pragma clang fp contract(off)
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float sqrt_a(float x) {   // v_sqrt + one fma-Newton step with v_rsq
  float s = __builtin_amdgcn_sqrtf(x);
  float h = 0.5f * __builtin_amdgcn_rsqf(x);
  float r = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_b(float x) {   // v_sqrt + neighbour test (the compiler's core, no scaling)
  float s = __builtin_amdgcn_sqrtf(x);
  float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = (rm <= 0.0f) ? sm : s;
  s = (rp > 0.0f) ? sp : s;
  return s;
}
__device__ __forceinline__ float rcp_1(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_2(float x) {
  float r = rcp_1(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// v_rsq only (one transcendental instead of three): y ~ 1/sqrt(x); s = x*y ~ sqrt(x), corrected to RN(sqrt(x));
// then RN(1 / s) by Newton steps from y itself (y is within ~2 ulp of 1/s).
__device__ __forceinline__ float sqrt_c(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y;
  float h = 0.5f * y;
  float r = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float rsqrt_c(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y;
  float h = 0.5f * y;
  float q = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  float e = __builtin_fmaf(-q, y, 1.0f);
  return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float rsqrt_d(float x) {   // two Newton steps
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y;
  float h = 0.5f * y;
  float q = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  float r = __builtin_fmaf(__builtin_fmaf(-q, y, 1.0f), y, y);
  return __builtin_fmaf(__builtin_fmaf(-q, r, 1.0f), r, r);
}
// a / d from the correctly rounded reciprocal: the compiler's own expansion without its scaling and fix-up steps
__device__ __forceinline__ float div_1(float a, float d) {
  float r = rcp_1(d);
  float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
}
__device__ __forceinline__ float div_2(float a, float d) {
  float r = rcp_1(d);
  float q = a * r;
  q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
  return __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
}
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return (uint32_t)((z ^ (z >> 31)) >> 16);
}
// 2^34 pseudo-random (a, d) pairs: random signs and mantissas, exponents of a and d within +-40 of 1.0
// (quotient, residuals and reciprocal all stay normal), a quarter of the pairs with a few-bit mantissas.
__global__ void check_div(unsigned long long* stats) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t b = tid; b < (1ull << 34); b += stride) {
    uint32_t ra = mix32(2 * b + 1), rd = mix32(2 * b + 2), re = mix32(b ^ 0x5555555555ull);
    uint32_t ma = ra & 0x7fffffu, md = rd & 0x7fffffu;
    if ((re & 3u) == 0u) { ma &= 0x7f0000u; }
    if ((re & 12u) == 0u) { md &= 0x7e0000u; }
    const uint32_t ea = 87u + ((re >> 4) % 81u), ed = 87u + ((re >> 12) % 81u);
    const float a = __uint_as_float((ra & 0x80000000u) | (ea << 23) | ma);
    const float d = __uint_as_float((rd & 0x80000000u) | (ed << 23) | md);
    const float ref = a / d;
    if (__float_as_uint(div_1(a, d)) != __float_as_uint(ref)) atomicAdd(&stats[0], 1ull);
    if (__float_as_uint(div_2(a, d)) != __float_as_uint(ref)) atomicAdd(&stats[1], 1ull);
  }
}

// stats[test][0] = mismatches over all x, [1] = mismatches for normal positive x, [2],[3] = min/max exponent of a normal mismatch
__global__ void check(unsigned long long* stats) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t b = tid; b < (1ull << 32); b += stride) {
    const float x = __uint_as_float((uint32_t)b);
    const uint32_t ex = ((uint32_t)b >> 23) & 255u;
    const bool normal_pos = ((uint32_t)b >> 31) == 0 && ex >= 1 && ex <= 254;
    const bool normal_any = ex >= 1 && ex <= 254;
    const float rs = sqrtf(x), rr = 1.0f / x, rrs = 1.0f / sqrtf(x);
    const float t[8] = { sqrt_a(x), sqrt_b(x), rcp_1(x), rcp_2(x), rcp_2(sqrt_a(x)), sqrt_c(x), rsqrt_c(x), rsqrt_d(x) };
    const float ref[8] = { rs, rs, rr, rr, rrs, rs, rrs, rrs };
    const bool dom[8] = { normal_pos, normal_pos, normal_any, normal_any, normal_pos, normal_pos, normal_pos, normal_pos };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool same = __float_as_uint(t[k]) == __float_as_uint(ref[k]) || (t[k] != t[k] && ref[k] != ref[k]);
      if (!same) {
        atomicAdd(&stats[k * 4 + 0], 1ull);
        if (dom[k]) {
          if (k == 6) atomicAdd(&stats[64 + ex], 1ull);      // per-exponent histogram of the rsq-only 1/sqrt
          atomicAdd(&stats[k * 4 + 1], 1ull);
          atomicMin(&stats[k * 4 + 2], (unsigned long long)ex);
          atomicMax(&stats[k * 4 + 3], (unsigned long long)ex);
        }
      }
    }
  }
}

int main() {
  unsigned long long* d; static unsigned long long h[64 + 256];
  for (int k = 0; k < 8; ++k) { h[k*4] = 0; h[k*4+1] = 0; h[k*4+2] = 999; h[k*4+3] = 0; }
  CHECK(hipMalloc(&d, sizeof h)); CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  const char* names[8] = { "sqrt_a (sqrt,rsq,2 fma)", "sqrt_b (sqrt, +-1ulp test)", "rcp_1 (rcp + 1 NR)", "rcp_2 (rcp + 2 NR)", "rcp_2(sqrt_a(x)) vs 1/sqrt",
                           "sqrt_c (rsq only)", "rsqrt_c (rsq only, 1 NR)", "rsqrt_d (rsq only, 2 NR)" };
  for (int k = 0; k < 8; ++k)
    printf("%-30s mismatches: all inputs %llu, normal-range inputs %llu (biased exponents %llu..%llu)\n",
           names[k], h[k*4], h[k*4+1], h[k*4+2], h[k*4+3]);
  printf("rsqrt_c mismatches by biased exponent of x:");
  for (int e = 0; e < 256; ++e) if (h[64 + e]) printf(" %d:%llu", e, h[64 + e]);
  printf("\n");
  unsigned long long hd[2] = { 0, 0 };
  unsigned long long* dd; CHECK(hipMalloc(&dd, sizeof hd)); CHECK(hipMemcpy(dd, hd, sizeof hd, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check_div, dim3(8192), dim3(256), 0, 0, dd);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
  printf("div_1 (one residual correction)  mismatches in 2^34 pairs: %llu\ndiv_2 (two residual corrections) mismatches in 2^34 pairs: %llu\n", hd[0], hd[1]);
  return 0;
}
