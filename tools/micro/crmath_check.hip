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
    const float t[5] = { sqrt_a(x), sqrt_b(x), rcp_1(x), rcp_2(x), rcp_2(sqrt_a(x)) };
    const float ref[5] = { rs, rs, rr, rr, rrs };
    const bool dom[5] = { normal_pos, normal_pos, normal_any, normal_any, normal_pos };
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const bool same = __float_as_uint(t[k]) == __float_as_uint(ref[k]) || (t[k] != t[k] && ref[k] != ref[k]);
      if (!same) {
        atomicAdd(&stats[k * 4 + 0], 1ull);
        if (dom[k]) {
          atomicAdd(&stats[k * 4 + 1], 1ull);
          atomicMin(&stats[k * 4 + 2], (unsigned long long)ex);
          atomicMax(&stats[k * 4 + 3], (unsigned long long)ex);
        }
      }
    }
  }
}

int main() {
  unsigned long long* d; unsigned long long h[20];
  for (int k = 0; k < 5; ++k) { h[k*4] = 0; h[k*4+1] = 0; h[k*4+2] = 999; h[k*4+3] = 0; }
  CHECK(hipMalloc(&d, sizeof h)); CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  const char* names[5] = { "sqrt_a (sqrt,rsq,2 fma)", "sqrt_b (sqrt, +-1ulp test)", "rcp_1 (rcp + 1 NR)", "rcp_2 (rcp + 2 NR)", "rcp_2(sqrt_a(x)) vs 1/sqrt" };
  for (int k = 0; k < 5; ++k)
    printf("%-30s mismatches: all inputs %llu, normal-range inputs %llu (biased exponents %llu..%llu)\n",
           names[k], h[k*4], h[k*4+1], h[k*4+2], h[k*4+3]);
  return 0;
}
