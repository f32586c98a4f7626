// Microbenchmark: what VALU issue rate does an MI355X SIMD sustain for plain (unpacked, unfused)
// fp32 ops at a given occupancy? Sets the honest ceiling for code compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {        // 8 independent fma chains
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
      } else if (MODE == 1) { // 8 independent mul, add chains (unfused)
        x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
        x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
      } else {                // ONE dependent mul/add chain
        x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b;
        x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE> void run(const char* name, int blocks, int insts_per_iter) {
  float* out; CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double winst = (double)blocks * 4 * iters * 8 * insts_per_iter;         // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9 * 1024 / winst;              // cycles per wave-inst per SIMD at 2.4 GHz
    if (rep) printf("%-28s waves/SIMD %2d: %8.3f ms  %.2f cycles/wave-inst/SIMD (at 2.4 GHz)  %.1f Tlane-op/s\n",
                    name, blocks * 4 / 1024, ms, per_simd_cycles, winst * 64 / (ms * 1e-3) / 1e12);
  }
  CHECK(hipFree(out));
}

int main() {
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("fma x8 independent", blocks, 8);
    run<1>("mul,add x8 independent", blocks, 16);
    run<2>("mul,add 1 dependent chain", blocks, 16);
  }
  return 0;
}
