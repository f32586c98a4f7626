// Microbenchmark: the shader clock a VALU-bound kernel actually runs at on this GPU.
// s_memtime (clock64) counts shader-engine clocks, s_memrealtime (wall_clock64) a constant 100 MHz
// reference: their ratio over a long v_mul_f32 / v_add_f32 loop at full occupancy is the frequency
// that cycle-based ceilings (tools/micro/inst_cost.hip) have to be converted with.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/clock_probe tools/micro/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int blocks = 2048, iters = 20000;
  float* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2 * blocks];
    CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    double c = 0, w = 0;
    for (int b = 0; b < blocks; ++b) { c += (double)h[2 * b]; w += (double)h[2 * b + 1]; }
    const double mhz = c / w * 100.0;                       // wall clock is 100 MHz
    const double winst = (double)blocks * 4 * iters * 64;   // wave-instructions
    printf("rep %d: %.3f ms, shader clock %.0f MHz (clock64/wall_clock64), %.2f cycles per wave64 v_mul/v_add per SIMD at that clock, "
           "unfused fp32 ceiling %.1f Tflop/s\n", rep, ms, mhz, ms * 1e-3 * mhz * 1e6 * 1024 / winst, winst * 64 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
