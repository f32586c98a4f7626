// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instruction
// classes the path tracer executes, measured with inline asm so the compiler cannot pack,
// fuse or drop anything. 8 independent accumulators per class -> throughput, not latency.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/inst_cost tools/micro/inst_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

enum { MUL, PKMUL, FMA, PKFMA, MOV, CNDMASK, CMP, RCP, SQRT, MADU64, MULLO, MULHI, MADU24, READLANE, DPPMOV, XOR, ADDU, PKADD, MULDEP, PKMULDEP, NCLASS };
static const char* NAMES[] = {"v_mul_f32", "v_pk_mul_f32", "v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32",
                              "v_rcp_f32", "v_sqrt_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
                              "v_readlane_b32", "v_mov_b32 dpp row_shr", "v_xor_b32", "v_add_u32", "v_pk_add_f32",
                              "v_mul_f32 dependent chain", "v_pk_mul_f32 dependent chain"};

template <int C>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a) {
  float x[8];
  double d[8];
  unsigned long long q[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; d[i] = threadIdx.x + i; q[i] = threadIdx.x + i; }
  int s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (C == MUL) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == PKMUL) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        REP8(S)
#undef S
      } else if (C == FMA) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == PKFMA) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        REP8(S)
#undef S
      } else if (C == MOV) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
        REP8(S)
#undef S
      } else if (C == CMP) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if (C == RCP) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == SQRT) {
#define S(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MADU64) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(x[i]), "v"(a) : "vcc");
        REP8(S)
#undef S
      } else if (C == MULLO) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == MULHI) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == MADU24) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == READLANE) {
#define S(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(x[i]));
        REP8(S)
#undef S
      } else if (C == DPPMOV) {
#define S(i) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == XOR) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == ADDU) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == PKADD) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        REP8(S)
#undef S
      } else if (C == MULDEP) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(a));
        REP8(S)
#undef S
      } else if (C == PKMULDEP) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1\n s_nop 0" : "+v"(d[0]) : "v"(d[1]));
        REP8(S)
#undef S
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += x[i] + (float)d[i] + (float)q[i];
  out[blockIdx.x * 256 + threadIdx.x] = r + s;
}

template <int C> void run(int blocks) {
  float* out; CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 8000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  double winst = (double)blocks * 4 * iters * 32;
  double cyc = ms * 1e-3 * 2.4e9 * 1024 / winst;
  printf("%-30s waves/SIMD %2d: %8.3f ms  %6.2f cycles/wave-inst/SIMD (2.4 GHz assumed)\n", NAMES[C], blocks * 4 / 1024, ms, cyc);
  CHECK(hipFree(out));
}

template <int C> void all() { for (int blocks : {256, 1024, 2048}) run<C>(blocks); }

int main() {
  all<MUL>(); all<PKMUL>(); all<FMA>(); all<PKFMA>(); all<PKADD>(); all<MOV>(); all<CNDMASK>(); all<CMP>(); all<RCP>(); all<SQRT>();
  all<MADU64>(); all<MULLO>(); all<MULHI>(); all<MADU24>(); all<READLANE>(); all<DPPMOV>(); all<XOR>(); all<ADDU>();
  all<MULDEP>(); all<PKMULDEP>();
  return 0;
}
