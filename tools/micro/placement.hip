// tools/micro/placement.hip -- where do the waves of a grid of 256-thread blocks run, and when? Each wave records
// HW_ID (SIMD / CU / SE), XCC_ID and its start / end on the chip-wide 100 MHz counter around a fixed VALU loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/placement tools/micro/placement.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Rec { unsigned hw_id, xcc_id; unsigned long long r0, r1, c0, c1; };
__global__ __launch_bounds__(256) void k(Rec* rec, float* out, int iters, float a) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i + 1.5f;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u & 7]) : "v"(a));
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float r = 0; for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) {
    Rec& o = rec[blockIdx.x * 4 + (threadIdx.x >> 6)];
    o.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    o.xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    o.r0 = r0; o.r1 = r1; o.c0 = c0; o.c1 = c1;
  }
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  for (int per_cu : {1, 2, 4, 5, 8}) {
    const int blocks = p.multiProcessorCount * per_cu, iters = 6000;
    Rec* rec; float* out;
    CHECK(hipMalloc(&rec, sizeof(Rec) * blocks * 4)); CHECK(hipMalloc(&out, 4 * blocks * 256));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, rec, out, iters, 1.0f); CHECK(hipDeviceSynchronize()); }
    std::vector<Rec> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    std::map<unsigned long long, std::vector<int>> by_simd;
    for (size_t i = 0; i < h.size(); ++i) {
      t0 = std::min(t0, h[i].r0); t1 = std::max(t1, h[i].r1);
      const unsigned simd = (h[i].hw_id >> 4) & 3, cu = (h[i].hw_id >> 8) & 15, sh = (h[i].hw_id >> 12) & 1, se = (h[i].hw_id >> 13) & 7;
      by_simd[((unsigned long long)(h[i].xcc_id & 15) << 32) | (se << 16) | (sh << 12) | (cu << 4) | simd].push_back((int)i);
    }
    // per SIMD: waves hosted, and the largest number of them alive at one instant
    std::map<int, int> hist_hosted, hist_live;
    for (auto& kv : by_simd) {
      std::vector<std::pair<unsigned long long, int>> ev;
      for (int i : kv.second) { ev.push_back({h[i].r0, +1}); ev.push_back({h[i].r1, -1}); }
      std::sort(ev.begin(), ev.end());
      int live = 0, peak = 0; for (auto& e : ev) { live += e.second; peak = std::max(peak, live); }
      hist_hosted[(int)kv.second.size()]++; hist_live[peak]++;
    }
    double span = 0, cyc = 0; for (auto& r : h) { span += (double)(r.r1 - r.r0); cyc += (double)(r.c1 - r.c0); }
    printf("%d blocks/CU (%d blocks): launch span %.3f ms, mean wave span %.3f ms, clock %.0f MHz, distinct SIMDs seen %zu\n", per_cu, blocks,
           (t1 - t0) / 1e5, span / h.size() / 1e5, 100.0 * cyc / span, by_simd.size());
    printf("   waves hosted per SIMD over the launch:");
    for (auto& kv : hist_hosted) printf("  %d waves: %d SIMDs", kv.first, kv.second);
    printf("\n   peak waves alive at once per SIMD:    ");
    for (auto& kv : hist_live) printf("  %d: %d SIMDs", kv.first, kv.second);
    printf("\n   cycles per wave-instruction per SIMD, whole launch: %.2f\n", (t1 - t0) / 1e8 * (100.0 * cyc / span) * 1e6 / ((double)iters * 32 * per_cu));
    CHECK(hipFree(rec)); CHECK(hipFree(out));
  }
  return 0;
}
