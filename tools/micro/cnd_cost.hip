// Follow-up to inst_cost.hip: v_cndmask_b32 variants (mask in vcc / in an SGPR pair, fresh or stale
// mask, in-place or not), compare+select pairs, and min/max/med3 as select replacements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
enum { CND_VCC_STALE, CND_SGPR, CND_VCC_INIT, CND_NOTINPLACE, CMP_CND, CMP64_CND, MAXF, MED3, CMP_E64, CMP1_CND8, SAND_CND8, SAND_CND8_E64, P1, P2, P3, P5, NCLASS };
static const char* NAMES[] = {"cndmask vcc (never written)", "cndmask e64 sgpr-pair mask", "cndmask vcc written once by s_mov",
                              "cndmask vcc, dst != src", "v_cmp vcc + cndmask (pair)", "v_cmp_e64 sgpr + cndmask (pair)",
                              "v_max_f32", "v_med3_f32", "v_cmp_lt_f32_e64 -> sgpr pair", "1 v_cmp vcc + 8 cndmask vcc (per inst)", "1 s_and vcc + 8 cndmask vcc (per inst)", "1 s_and sgpr + 8 cndmask e64 (per inst)", "stale vcc: cndmask, v_mul alternating (per PAIR)", "v_cmp vcc + 2 cndmask vcc (per TRIPLE)", "v_cmp vcc, v_mul, cndmask vcc (per TRIPLE)", "stale vcc: cndmask, 3 v_mul (per QUAD)"};
template <int C>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, unsigned long long mask) {
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = i; }
  if (C == CND_VCC_INIT || C == CND_NOTINPLACE) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  unsigned long long m2 = mask;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (C == CND_VCC_STALE || C == CND_VCC_INIT) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == CND_SGPR) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mask));
        REP8(S)
#undef S
      } else if (C == CND_NOTINPLACE) {
#define S(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y[i]) : "v"(x[i]), "v"(a));
        REP8(S)
#undef S
      } else if (C == CMP_CND) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if (C == CMP64_CND) {
#define S(i) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(x[i]), "+s"(m2) : "v"(a));
        REP8(S)
#undef S
      } else if (C == MAXF) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == MED3) {
#define S(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == CMP1_CND8) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[0]), "v"(a) : "vcc");
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == SAND_CND8) {
        asm volatile("s_and_b64 vcc, %0, exec" : : "s"(mask) : "vcc", "scc");
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == SAND_CND8_E64) {
        asm volatile("s_and_b64 %0, %1, exec" : "=s"(m2) : "s"(mask) : "scc");
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(m2));
        REP8(S)
#undef S
      } else if (C == P1) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_mul_f32 %1, %1, %2" : "+v"(x[i]), "+v"(y[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == P2) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[i]), "+v"(y[i]) : "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if (C == P3) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_mul_f32 %1, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]), "+v"(y[i]) : "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if (C == P5) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_mul_f32 %1, %1, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %1, %1, %2" : "+v"(x[i]), "+v"(y[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == CMP_E64) {
#define S(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(x[i]), "v"(a));
        REP8(S)
#undef S
      }
    }
  }
  float r = (float)m2;
  for (int i = 0; i < 8; ++i) r += x[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int C> void run(int blocks, int per) {
  float* out; CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 8000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0x5555aaaa5555aaaaull);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  double units = (double)blocks * 4 * iters * 32;
  printf("%-36s waves/SIMD %2d: %8.3f ms  %6.2f cycles per %s per SIMD (2.4 GHz assumed)\n", NAMES[C], blocks * 4 / 1024, ms,
         ms * 1e-3 * 2.4e9 * 1024 / units, per == 2 ? "GROUP" : "inst"); fflush(stdout);
  CHECK(hipFree(out));
}
template <int C> void all(int per = 1) { for (int blocks : {256, 1024, 2048}) run<C>(blocks, per); }
int main() {
  all<CND_VCC_STALE>(); all<CND_VCC_INIT>(); all<CND_SGPR>(); all<CND_NOTINPLACE>(); all<CMP_CND>(2); all<CMP64_CND>(2);
  all<MAXF>(); all<MED3>(); all<CMP_E64>(); all<CMP1_CND8>(); all<SAND_CND8>(); all<SAND_CND8_E64>(); all<P1>(2); all<P2>(2); all<P3>(2); all<P5>(2);
  return 0;
}
