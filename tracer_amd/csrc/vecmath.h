// tracer_amd/csrc/vecmath.h -- fp32 vector algebra for host and gfx950 device code.
//
// Primary-hit parity with the reference's host build is bit-level (BASELINE.json), so every
// helper spells out the operation ORDER of the vendored GLM 0.9.4.3 call it replaces
// (file:line under /root/reference/externals/glm) and the translation unit is compiled with
// -ffp-contract=off: a fused multiply-add would change the last bit of most expressions
// (SURVEY.md 7.2-1). IEEE `/` and sqrtf stay correctly rounded on device (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt; never -ffast-math).
#pragma once

#include <cstdint>
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
#define TRC_HD __host__ __device__ __forceinline__
#else
// host-only consumers of the scene-description API (tracer_amd/csrc/host/) built with a plain
// C++ compiler: pass -ffp-contract=off there too.
#define TRC_HD inline
#endif

namespace trc {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

TRC_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
TRC_HD f3 splat3(float s) { return mk3(s, s, s); }
TRC_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
TRC_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
TRC_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
TRC_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
TRC_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
TRC_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
TRC_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// glm::min / glm::max / glm::abs keep the reference's NaN behaviour (core/func_common.inl:296,
// :314, :44): plain compare-and-select, NOT fminf/fmaxf.
TRC_HD float gmin(float x, float y) { return x < y ? x : y; }
TRC_HD float gmax(float x, float y) { return x > y ? x : y; }
TRC_HD float gabs(float x) { return x >= 0.0f ? x : -x; }
TRC_HD float gclamp(float x, float lo, float hi) { return gmin(hi, gmax(lo, x)); } // :335
TRC_HD f3 gmin(f3 a, f3 b) { return mk3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
TRC_HD f3 gmax(f3 a, f3 b) { return mk3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }

// core/func_geometric.inl:163  x.x*y.x + x.y*y.y + x.z*y.z, left to right
TRC_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// core/func_geometric.inl:204-207
TRC_HD f3 cross(f3 x, f3 y) {
  return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// core/func_exponential.inl:151  genType(1) / sqrt(x)
TRC_HD float inversesqrt(float x) { return 1.0f / sqrtf(x); }
// core/func_geometric.inl:243-244  x * inversesqrt(dot-like sum)
TRC_HD f3 normalize(f3 v) {
  float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
  return v * inversesqrt(sqr);
}
// core/func_geometric.inl:279  I - N * dot(N, I) * genType(2)
TRC_HD f3 reflect(f3 I, f3 N) { return I - (N * dot(N, I)) * splat3(2.0f); }
// core/func_geometric.inl:311-317 (vec3, scalar eta)
TRC_HD f3 refract(f3 I, f3 N, float eta) {
  float dotValue = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
  if (k < 0.0f) return splat3(0.0f);
  return eta * I - (eta * dotValue + sqrtf(k)) * N;
}

// Affine 3x4 part of a glm::mat4, stored as the reference stores it (column-major:
// c[col][row]); the w row is never read by the path (Ray.inl:51-62 take vec3(m*v)).
struct Affine { float c[4][3]; };

// core/type_mat4x4.inl:695-699 with v.w == 1: ((m0*x + m1*y) + m2*z) + m3*1
TRC_HD f3 xform_point(const Affine& m, f3 p) {
  return mk3(m.c[0][0] * p.x + m.c[1][0] * p.y + m.c[2][0] * p.z + m.c[3][0] * 1.0f,
             m.c[0][1] * p.x + m.c[1][1] * p.y + m.c[2][1] * p.z + m.c[3][1] * 1.0f,
             m.c[0][2] * p.x + m.c[1][2] * p.y + m.c[2][2] * p.z + m.c[3][2] * 1.0f);
}
// same with v.w == 0 (the m3*0 term is kept: it decides the sign of a zero result)
TRC_HD f3 xform_dir(const Affine& m, f3 d) {
  return mk3(m.c[0][0] * d.x + m.c[1][0] * d.y + m.c[2][0] * d.z + m.c[3][0] * 0.0f,
             m.c[0][1] * d.x + m.c[1][1] * d.y + m.c[2][1] * d.z + m.c[3][1] * 0.0f,
             m.c[0][2] * d.x + m.c[1][2] * d.y + m.c[2][2] * d.z + m.c[3][2] * 0.0f);
}

} // namespace trc
