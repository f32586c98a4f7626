// tools/micro/issue_rate.hip -- VALU issue cost per instruction class on gfx950, measured in SHADER CLOCKS by the
// wavefronts themselves (s_memtime around the loop; s_memrealtime gives the clock the chip held), at 1, 2, 4 and
// 8 resident waves per SIMD, with 8 independent dependency chains per wave and three distinct source registers
// where the instruction takes three. cycles per wave-instruction per SIMD = median wave span / (instructions per
// wave x waves per SIMD): every SIMD hosts exactly W waves that start together and run the same loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_rate tools/micro/issue_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

enum { ADD, MUL, ADDMUL, FMA3, FMA_S, MUL_S, MIN, MAX3, MOV, CNDMASK, CMP_VCC, CMP_SGPR, CMP_CND, RCP, RSQ, SQRT, MULLO, MULHI, XOR, ADDU, PKMUL, PKFMA,
       FMA_ADD_MIX, TRANS_MIX, MUL_INLINE, MUL_LITERAL, ADD_E64_NEG, MUL_ABS, FMA_INLINE, CND_SGPRMASK, CND_VCC_FED, READLANE, READFIRST, AND, LSHL, CVT_F32_U32, RNDNE, MAD_U32_U24, BFE, MBCNT, MAX, MIN3, CMP_CLASS, CMP_U32, SUBREV, FMAAK, MIX_SGPR, MIX_CMP, MIX_MIN, MIX_MULLO, MIX_CMP_CND, MIX_SGPR2, MIX_PK1, MIX_PK3, PKADD, MIX_PKADD1, MIX_PK_2OF3, MAD_U64, MIX_MAD_U64, MIX_CND_SGPR, MIX_CND_SGPR1, NCLASS };
static const char* NAMES[] = {"v_add_f32", "v_mul_f32", "v_add_f32 / v_mul_f32 alternating", "v_fma_f32 (3 vgpr)", "v_fma_f32 (sgpr operand)",
  "v_mul_f32 (sgpr operand)", "v_min_f32", "v_max3_f32", "v_mov_b32", "v_cndmask_b32 (vcc)", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair",
  "v_cmp_lt_f32 vcc + v_cndmask (pair = 2 inst)", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_xor_b32", "v_add_u32",
  "v_pk_mul_f32", "v_pk_fma_f32", "3 v_add_f32 + 1 v_fma_f32 (4 inst)", "7 v_mul_f32 + 1 v_rcp_f32 (8 inst)",
  "v_mul_f32 (inline constant 2.0)", "v_mul_f32 (32-bit literal)", "v_add_f32 e64 with neg modifier", "v_mul_f32 e64 with |abs|", "v_fma_f32 (inline constant)",
  "v_cndmask_b32 e64 (sgpr-pair mask)", "v_cmp -> vcc ; v_cndmask vcc (other regs; 2 inst)", "v_readlane_b32", "v_readfirstlane_b32", "v_and_b32", "v_lshlrev_b32",
  "v_cvt_f32_u32", "v_rndne_f32", "v_mad_u32_u24", "v_bfe_u32", "v_mbcnt_lo_u32_b32", "v_max_f32", "v_min3_f32", "v_cmp_class_f32 -> sgpr", "v_cmp_lt_u32 -> sgpr",
  "v_subrev_f32", "v_fmaak_f32 (literal addend)",
  "3 v_mul_f32 + 1 v_mul_f32 sgpr (4 inst)", "3 v_mul_f32 + 1 v_cmp -> sgpr (4 inst)", "3 v_mul_f32 + 1 v_min_f32 (4 inst)", "3 v_mul_f32 + 1 v_mul_lo_u32 (4 inst)",
  "2 v_mul_f32 + v_cmp vcc + v_cndmask vcc (4 inst)", "1 v_mul_f32 + 1 v_mul_f32 sgpr (2 inst)",
  "1 v_mul_f32 + 1 v_pk_mul_f32 (2 inst)", "3 v_mul_f32 + 1 v_pk_mul_f32 (4 inst)", "v_pk_add_f32", "1 v_add_f32 + 1 v_pk_add_f32 (2 inst)",
  "1 v_mul_f32 + 2 v_pk_mul_f32 (3 inst)", "v_mad_u64_u32", "3 v_mul_f32 + 1 v_mad_u64_u32 (4 inst)",
  "3 v_mul_f32 + 1 v_cndmask_b32 e64 sgpr mask (4 inst)", "1 v_mul_f32 + 1 v_cndmask_b32 e64 sgpr mask (2 inst)"};

template <int C>
__global__ __launch_bounds__(256) void k(unsigned long long* span, unsigned long long* ref, float* out, int iters, float a, float b) {
  float x[8], y[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i + 1.5f; y[i] = a + i; d[i] = threadIdx.x + i; }
  float sa = a;   // wave-uniform -> SGPR
  unsigned long long s0 = 0x5555555555555555ull;
  int si = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (C == ADD) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MUL) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == ADDMUL) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i])); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[(i + 4) & 7]) : "v"(y[i]));
        S(0) S(1) S(2) S(3)
#undef S
      } else if (C == FMA3) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 3) & 7]));
        REP8(S)
#undef S
      } else if (C == FMA_S) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(sa), "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MUL_S) {
#define S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sa));
        REP8(S)
#undef S
      } else if (C == MIN) {
#define S(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MAX3) {
#define S(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 3) & 7]));
        REP8(S)
#undef S
      } else if (C == MOV) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == CMP_VCC) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(y[i]) : "vcc");
        REP8(S)
#undef S
      } else if (C == CMP_SGPR) {
#define S(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(s0) : "v"(x[i]), "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == CMP_CND) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
        S(0) S(1) S(2) S(3)
#undef S
      } else if (C == RCP) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == RSQ) {
#define S(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == SQRT) {
#define S(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MULLO) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MULHI) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == XOR) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == ADDU) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == PKMUL) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        REP8(S)
#undef S
      } else if (C == PKFMA) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
        REP8(S)
#undef S
      } else if (C == FMA_ADD_MIX) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[1]) : "v"(y[1]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2]) : "v"(y[2])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[3]) : "v"(y[3]), "v"(y[4]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[4]) : "v"(y[0])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[5]) : "v"(y[1]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[6]) : "v"(y[2])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[7]) : "v"(y[3]), "v"(y[4]));
      } else if (C == MUL_INLINE) {
#define S(i) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MUL_LITERAL) {
#define S(i) asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == ADD_E64_NEG) {
#define S(i) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MUL_ABS) {
#define S(i) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == FMA_INLINE) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == CND_SGPRMASK) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(s0));
        REP8(S)
#undef S
      } else if (C == CND_VCC_FED) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]) : "vcc");
        S(0) S(1) S(2) S(3)
#undef S
      } else if (C == READLANE) {
#define S(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(si) : "v"(x[i]));
        REP8(S)
#undef S
      } else if (C == READFIRST) {
#define S(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(si) : "v"(x[i]));
        REP8(S)
#undef S
      } else if (C == AND) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == LSHL) {
#define S(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == CVT_F32_U32) {
#define S(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == RNDNE) {
#define S(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MAD_U32_U24) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 3) & 7]));
        REP8(S)
#undef S
      } else if (C == BFE) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MBCNT) {
#define S(i) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x[i]));
        REP8(S)
#undef S
      } else if (C == MAX) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MIN3) {
#define S(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 3) & 7]));
        REP8(S)
#undef S
      } else if (C == CMP_CLASS) {
#define S(i) asm volatile("v_cmp_class_f32 %0, %1, %2" : "=s"(s0) : "v"(x[i]), "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == CMP_U32) {
#define S(i) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(s0) : "v"(x[i]), "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == SUBREV) {
#define S(i) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == FMAAK) {
#define S(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd" : "+v"(x[i]) : "v"(y[i]));
        REP8(S)
#undef S
      } else if (C == MIX_SGPR) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        M(0) M(1) M(2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[3]) : "s"(sa));
        M(4) M(5) M(6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[7]) : "s"(sa));
      } else if (C == MIX_CMP) {
        M(0) M(1) M(2) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(s0) : "v"(x[3]), "v"(y[3]));
        M(4) M(5) M(6) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(s0) : "v"(x[7]), "v"(y[7]));
      } else if (C == MIX_MIN) {
        M(0) M(1) M(2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[3]) : "v"(y[3]));
        M(4) M(5) M(6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[7]) : "v"(y[7]));
      } else if (C == MIX_MULLO) {
        M(0) M(1) M(2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[3]) : "v"(y[3]));
        M(4) M(5) M(6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[7]) : "v"(y[7]));
      } else if (C == MIX_CMP_CND) {
        M(0) M(1) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[2]) : "v"(y[2]), "v"(y[3]) : "vcc");
        M(4) M(5) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[6]) : "v"(y[6]), "v"(y[7]) : "vcc");
      } else if (C == MIX_SGPR2) {
        M(0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[1]) : "s"(sa)); M(2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[3]) : "s"(sa));
        M(4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[5]) : "s"(sa)); M(6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[7]) : "s"(sa));
#undef M
      } else if (C == MIX_PK1) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define P(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        M(0) P(0) M(1) P(2) M(2) P(4) M(3) P(6)
      } else if (C == MIX_PK3) {
        M(0) M(1) M(2) P(0) M(4) M(5) M(6) P(4)
      } else if (C == MIX_PK_2OF3) {
        M(0) P(0) P(2) M(1) P(4) P(6) M(2) P(1)
#undef P
      } else if (C == PKADD) {
#define P(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7)
      } else if (C == MIX_PKADD1) {
#define A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        A(0) P(0) A(1) P(2) A(2) P(4) A(3) P(6)
#undef A
#undef P
#undef M
      } else if (C == MAD_U64) {
#define Q(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d[i]) : "v"(x[i]), "v"(y[i]) : "vcc");
        Q(0) Q(1) Q(2) Q(3) Q(4) Q(5) Q(6) Q(7)
      } else if (C == MIX_MAD_U64) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        M(0) M(1) M(2) Q(3) M(4) M(5) M(6) Q(7)
#undef M
#undef Q
      } else if (C == MIX_CND_SGPR) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define K(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(s0));
        M(0) M(1) M(2) K(3) M(4) M(5) M(6) K(7)
      } else if (C == MIX_CND_SGPR1) {
        M(0) K(1) M(2) K(3) M(4) K(5) M(6) K(7)
#undef M
#undef K
      } else if (C == TRANS_MIX) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        S(0) S(1) S(2) S(3) S(4) S(5) S(6)
#undef S
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x[7]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += x[i] + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = r + (float)s0 + (float)si;
  if ((threadIdx.x & 63) == 0) { span[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; ref[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0; }
}

template <int C> void run(int waves_per_simd, int n_cu) {
  const int blocks = n_cu * waves_per_simd;        // a 256-thread block = one wave on each of a CU's 4 SIMDs
  float* out; unsigned long long *span, *ref;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&span, 8 * blocks * 4)); CHECK(hipMalloc(&ref, 8 * blocks * 4));
  const int iters = 6000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float wall_ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(256), 0, 0, span, ref, out, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&wall_ms, e0, e1));
  }
  std::vector<unsigned long long> hs(blocks * 4), hr(blocks * 4);
  CHECK(hipMemcpy(hs.data(), span, 8 * blocks * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hr.data(), ref, 8 * blocks * 4, hipMemcpyDeviceToHost));
  std::vector<double> cyc(blocks * 4), mhz(blocks * 4);
  for (int i = 0; i < blocks * 4; ++i) { cyc[i] = (double)hs[i]; mhz[i] = 100.0 * hs[i] / (double)hr[i]; }
  std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
  const double inst = (double)iters * 32;
  // wall: the whole launch by HIP events; "by wall" = cycles/wave-inst/SIMD if every SIMD issued its W x inst instructions over that time
  const double clk = mhz[mhz.size() / 2] * 1e6;
  printf("%-46s waves/SIMD %d: %6.2f cycles/wave-inst/SIMD  (clock %5.0f MHz; median wave span %.3f ms, launch %.3f ms -> %5.2f by wall)\n",
         NAMES[C], waves_per_simd, cyc[cyc.size() / 2] / (inst * waves_per_simd), mhz[mhz.size() / 2],
         cyc[cyc.size() / 2] / clk * 1e3, wall_ms, wall_ms * 1e-3 * clk / (inst * waves_per_simd));
  CHECK(hipFree(out)); CHECK(hipFree(span)); CHECK(hipFree(ref));
}

template <int C> void all(int n_cu) { for (int w : {1, 2, 4}) run<C>(w, n_cu); }
template <int C> void occupancy(int n_cu) { for (int w : {1, 2, 3, 4, 5, 6, 7, 8}) run<C>(w, n_cu); }

int main(int argc, char** argv) {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int n = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, n);
  if (argc > 1 && argv[1][0] == 'o') {      // "occ": the plain arithmetic at 1 .. 8 resident waves per SIMD
    occupancy<ADD>(n); occupancy<MUL>(n); occupancy<ADDMUL>(n); occupancy<FMA3>(n); occupancy<MOV>(n); occupancy<MIX_CMP>(n);
    return 0;
  }
  all<MAD_U64>(n); all<MIX_MAD_U64>(n); all<MIX_CND_SGPR>(n); all<MIX_CND_SGPR1>(n);
  if (argc > 2) return 0;                  // two arguments: only the 64-bit multiply-add
  all<MIX_PK1>(n); all<MIX_PK3>(n); all<MIX_PK_2OF3>(n); all<PKADD>(n); all<MIX_PKADD1>(n);
  if (argc > 1) return 0;                  // any argument: only the packed-arithmetic mixes
  all<ADD>(n); all<MUL>(n); all<ADDMUL>(n); all<FMA3>(n); all<FMA_S>(n); all<MUL_S>(n); all<MIN>(n); all<MAX3>(n); all<MOV>(n); all<CNDMASK>(n);
  all<CMP_VCC>(n); all<CMP_SGPR>(n); all<CMP_CND>(n); all<RCP>(n); all<RSQ>(n); all<SQRT>(n); all<MULLO>(n); all<MULHI>(n); all<XOR>(n); all<ADDU>(n);
  all<PKMUL>(n); all<PKFMA>(n); all<FMA_ADD_MIX>(n); all<TRANS_MIX>(n);
  all<MUL_INLINE>(n); all<MUL_LITERAL>(n); all<ADD_E64_NEG>(n); all<MUL_ABS>(n); all<FMA_INLINE>(n); all<CND_SGPRMASK>(n); all<CND_VCC_FED>(n);
  all<READLANE>(n); all<READFIRST>(n); all<AND>(n); all<LSHL>(n); all<CVT_F32_U32>(n); all<RNDNE>(n); all<MAD_U32_U24>(n); all<BFE>(n); all<MBCNT>(n);
  all<MAX>(n); all<MIN3>(n); all<CMP_CLASS>(n); all<CMP_U32>(n); all<SUBREV>(n); all<FMAAK>(n);
  all<MIX_SGPR>(n); all<MIX_CMP>(n); all<MIX_MIN>(n); all<MIX_MULLO>(n); all<MIX_CMP_CND>(n); all<MIX_SGPR2>(n);
  return 0;
}
