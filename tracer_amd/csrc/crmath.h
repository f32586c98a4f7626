// tracer_amd/csrc/crmath.h -- device-only helpers: lane predicates as wave masks, and short
// correctly-rounded 1/sqrt, sqrt, 1/x sequences with a per-wavefront IEEE fallback.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#pragma clang fp contract(off)

namespace trc {

// ---- lane predicates as wave masks ------------------------------------------------------------------
// A predicate that feeds a wave-level decision ("does any lane that matters ...") is kept as the 64-bit
// mask a v_cmp writes to an SGPR pair: masks combine with scalar-ALU instructions, `mask != 0` is one
// s_cmp, and lane_of() hands a mask back to v_cndmask / exec as a per-lane bool. __ballot() of a bool
// that is not itself a comparison costs two VALU instructions (v_cndmask 0/1 + v_cmp_ne) every time.
// A mask has zeros for the lanes that were inactive where it was computed.
using lmask = unsigned long long;
enum { F_OEQ = 1, F_OGT = 2, F_OGE = 3, F_OLT = 4, F_OLE = 5, F_ONE = 6, F_UNE = 14 };            // llvm::CmpInst float predicates
enum { I_EQ = 32, I_NE = 33, I_UGT = 34, I_UGE = 35, I_ULT = 36, I_ULE = 37 };      // ... integer predicates
#define TRC_FCMP(a, b, pred) __builtin_amdgcn_fcmpf((a), (b), (pred))
#define TRC_UCMP(a, b, pred) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), (pred))
__device__ __forceinline__ lmask exec_mask() { return __builtin_amdgcn_ballot_w64(true); }
__device__ __forceinline__ bool lane_of(lmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// ---- short correctly-rounded 1/sqrt and reciprocal -------------------------------------------------
// glm::normalize is v * (1 / sqrt(s)) with BOTH the square root and the division correctly rounded
// (func_geometric.inl:243, func_exponential.inl:151); hipcc's IEEE sqrtf + division cost 28 VALU
// instructions, most of them range scaling and special-case fix-ups. The sequences below were
// compared with those IEEE results for ALL 2^32 inputs on gfx950 (tools/micro/crmath_check.hip):
//   sqrt:  y = v_rsq(x), s = x * y, one fma residual, one fma correction by y/2 -- exact for every x >= 2^-102
//   1/x :  v_rcp + one Newton step                             -- exact for every normal |x| < 2^126
//   a/d :  r = that 1/d, q = a * r, two fma residual corrections: hipcc's own expansion of `/` without its
//          v_div_scale / v_div_fixup steps, i.e. the same arithmetic wherever those steps do nothing;
//          compared with `/` on 2^34 random pairs with 2^-40 <= |a|, |d| < 2^40 (no mismatch)
// Outside those ranges (zero vectors, denormals, inf, NaN) the wave takes the IEEE path; the
// choice is made per wavefront with a ballot, so the result is bit-identical either way.
// `need` = the lanes whose result is observable (dead / masked lanes must not force the slow path).
__device__ __forceinline__ float cr_rsqrt(float s, lmask need) {
  const lmask ok = TRC_UCMP(__float_as_uint(s) - 0x0D000000u, 0x7E800000u - 0x0D000000u, I_ULT);   // 2^-101 <= s < 2^126
  if (__builtin_expect((need & ~ok) == 0ull, 1)) {
    const float y = __builtin_amdgcn_rsqf(s);
    const float q0 = s * y;
    const float h = 0.5f * y;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, q0, s), h, q0);      // correctly rounded sqrt(s)
    const float r0 = __builtin_amdgcn_rcpf(q);
    return __builtin_fmaf(__builtin_fmaf(-q, r0, 1.0f), r0, r0);            // correctly rounded 1/q
  }
  return 1.0f / sqrtf(s);
}
__device__ __forceinline__ float cr_sqrt(float s, lmask need) {
  const lmask ok = TRC_UCMP(__float_as_uint(s) - 0x0D000000u, 0x7F800000u - 0x0D000000u, I_ULT);   // 2^-101 <= s < inf
  if (__builtin_expect((need & ~ok) == 0ull, 1)) {
    const float y = __builtin_amdgcn_rsqf(s);
    const float q0 = s * y;
    const float h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-q0, q0, s), h, q0);
  }
  return sqrtf(s);
}
__device__ __forceinline__ float cr_rcp(float x, lmask need) {
  const lmask ok = TRC_UCMP((__float_as_uint(x) & 0x7FFFFFFFu) - 0x00800000u, 0x7E800000u - 0x00800000u, I_ULT); // normal, |x| < 2^126
  if (__builtin_expect((need & ~ok) == 0ull, 1)) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
  }
  return 1.0f / x;
}
// sqrtf(x) for the lanes that are active here (shading code, which runs under divergent control flow)
__device__ __forceinline__ float sqrt_active(float x) { return cr_sqrt(x, exec_mask()); }

} // namespace trc
