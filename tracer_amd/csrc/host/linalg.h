// tracer_amd/csrc/host/linalg.h -- the handful of 4x4 / quaternion operations the scene
// description needs on the host, GLM-free. Results must equal what the reference's host code
// computes with its vendored GLM 0.9.4.3, because object matrices and their inverses are inputs
// of the bit-exact hit test; each function names the GLM source whose operation order it keeps.
#pragma once
#include "../vecmath.h"
#include <cmath>

#pragma clang fp contract(off)

namespace trc {

struct f4 { float x, y, z, w; };
inline f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline f4 operator-(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline f4 operator*(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// column-major 4x4, memory-compatible with glm::mat4
struct mat4 {
  f4 col[4];
  float& at(int c, int r) { return (&col[c].x)[r]; }
  float at(int c, int r) const { return (&col[c].x)[r]; }
  const float* data() const { return &col[0].x; }
  float* data() { return &col[0].x; }
};
static_assert(sizeof(mat4) == 64, "mat4");

inline mat4 identity4() {
  mat4 m;
  m.col[0] = mk4(1, 0, 0, 0); m.col[1] = mk4(0, 1, 0, 0); m.col[2] = mk4(0, 0, 1, 0); m.col[3] = mk4(0, 0, 0, 1);
  return m;
}

// core/type_mat4x4.inl:689-700
inline f4 mul(const mat4& m, f4 v) {
  return mk4(m.at(0, 0) * v.x + m.at(1, 0) * v.y + m.at(2, 0) * v.z + m.at(3, 0) * v.w,
             m.at(0, 1) * v.x + m.at(1, 1) * v.y + m.at(2, 1) * v.z + m.at(3, 1) * v.w,
             m.at(0, 2) * v.x + m.at(1, 2) * v.y + m.at(2, 2) * v.z + m.at(3, 2) * v.w,
             m.at(0, 3) * v.x + m.at(1, 3) * v.y + m.at(2, 3) * v.z + m.at(3, 3) * v.w);
}

// core/type_mat4x4.inl:757-778
inline mat4 mul(const mat4& a, const mat4& b) {
  mat4 r;
  for (int j = 0; j < 4; ++j)
    r.col[j] = a.col[0] * b.at(j, 0) + a.col[1] * b.at(j, 1) + a.col[2] * b.at(j, 2) + a.col[3] * b.at(j, 3);
  return r;
}

// core/func_matrix.inl:521-581: cofactor expansion, then every element divided by the determinant
inline mat4 inverse(const mat4& m) {
  auto sub = [&](int a, int b, int c, int d, int e, int f, int g, int h) {
    return m.at(a, b) * m.at(c, d) - m.at(e, f) * m.at(g, h);
  };
  const float c00 = sub(2, 2, 3, 3, 3, 2, 2, 3), c02 = sub(1, 2, 3, 3, 3, 2, 1, 3), c03 = sub(1, 2, 2, 3, 2, 2, 1, 3);
  const float c04 = sub(2, 1, 3, 3, 3, 1, 2, 3), c06 = sub(1, 1, 3, 3, 3, 1, 1, 3), c07 = sub(1, 1, 2, 3, 2, 1, 1, 3);
  const float c08 = sub(2, 1, 3, 2, 3, 1, 2, 2), c10 = sub(1, 1, 3, 2, 3, 1, 1, 2), c11 = sub(1, 1, 2, 2, 2, 1, 1, 2);
  const float c12 = sub(2, 0, 3, 3, 3, 0, 2, 3), c14 = sub(1, 0, 3, 3, 3, 0, 1, 3), c15 = sub(1, 0, 2, 3, 2, 0, 1, 3);
  const float c16 = sub(2, 0, 3, 2, 3, 0, 2, 2), c18 = sub(1, 0, 3, 2, 3, 0, 1, 2), c19 = sub(1, 0, 2, 2, 2, 0, 1, 2);
  const float c20 = sub(2, 0, 3, 1, 3, 0, 2, 1), c22 = sub(1, 0, 3, 1, 3, 0, 1, 1), c23 = sub(1, 0, 2, 1, 2, 0, 1, 1);
  const f4 sa = mk4(+1, -1, +1, -1), sb = mk4(-1, +1, -1, +1);
  const f4 f0 = mk4(c00, c00, c02, c03), f1 = mk4(c04, c04, c06, c07), f2 = mk4(c08, c08, c10, c11);
  const f4 f3_ = mk4(c12, c12, c14, c15), f4_ = mk4(c16, c16, c18, c19), f5 = mk4(c20, c20, c22, c23);
  const f4 v0 = mk4(m.at(1, 0), m.at(0, 0), m.at(0, 0), m.at(0, 0));
  const f4 v1 = mk4(m.at(1, 1), m.at(0, 1), m.at(0, 1), m.at(0, 1));
  const f4 v2 = mk4(m.at(1, 2), m.at(0, 2), m.at(0, 2), m.at(0, 2));
  const f4 v3 = mk4(m.at(1, 3), m.at(0, 3), m.at(0, 3), m.at(0, 3));
  mat4 inv;
  inv.col[0] = sa * (v1 * f0 - v2 * f1 + v3 * f2);
  inv.col[1] = sb * (v0 * f0 - v2 * f3_ + v3 * f4_);
  inv.col[2] = sa * (v0 * f1 - v1 * f3_ + v3 * f5);
  inv.col[3] = sb * (v0 * f2 - v1 * f4_ + v2 * f5);
  const f4 row0 = mk4(inv.at(0, 0), inv.at(1, 0), inv.at(2, 0), inv.at(3, 0));
  const float det = dot(m.col[0], row0);
  for (int c = 0; c < 4; ++c) {
    inv.col[c].x /= det; inv.col[c].y /= det; inv.col[c].z /= det; inv.col[c].w /= det;
  }
  return inv;
}

// core/func_trigonometric.inl:41, :56
inline float radians(float deg) { const float pi = 3.1415926535897932384626433832795f; return deg * (pi / 180.0f); }
inline float degrees(float rad) { const float pi = 3.1415926535897932384626433832795f; return rad * (180.0f / pi); }

// gtc/matrix_transform.inl:44-82 (angle in degrees: the reference builds GLM without GLM_FORCE_RADIANS)
inline mat4 rotate(const mat4& m, float angle_deg, f3 v) {
  const float a = radians(angle_deg);
  const float c = cosf(a), s = sinf(a);
  const f3 axis = normalize(v);
  const f3 temp = (1.0f - c) * axis;
  float R[3][3];
  R[0][0] = c + temp.x * axis.x;
  R[0][1] = 0 + temp.x * axis.y + s * axis.z;
  R[0][2] = 0 + temp.x * axis.z - s * axis.y;
  R[1][0] = 0 + temp.y * axis.x - s * axis.z;
  R[1][1] = c + temp.y * axis.y;
  R[1][2] = 0 + temp.y * axis.z + s * axis.x;
  R[2][0] = 0 + temp.z * axis.x + s * axis.y;
  R[2][1] = 0 + temp.z * axis.y - s * axis.x;
  R[2][2] = c + temp.z * axis.z;
  mat4 r;
  for (int j = 0; j < 3; ++j) r.col[j] = m.col[0] * R[j][0] + m.col[1] * R[j][1] + m.col[2] * R[j][2];
  r.col[3] = m.col[3];
  return r;
}

// gtc/matrix_transform.inl:85-97
inline mat4 scale(const mat4& m, f3 v) {
  mat4 r;
  r.col[0] = m.col[0] * v.x; r.col[1] = m.col[1] * v.y; r.col[2] = m.col[2] * v.z; r.col[3] = m.col[3];
  return r;
}

struct quat { float x, y, z, w; };

// gtc/quaternion.inl:770-790 (degrees)
inline quat angleAxis(float angle_deg, f3 v) {
  const float a = radians(angle_deg);
  const float s = sinf(a * 0.5f);
  quat q; q.w = cosf(a * 0.5f); q.x = v.x * s; q.y = v.y * s; q.z = v.z * s;
  return q;
}

// gtc/quaternion.inl:626-654 (mat3_cast widened to 4x4), gtx/quaternion.hpp:155 toMat4
inline mat4 toMat4(quat q) {
  mat4 r = identity4();
  r.at(0, 0) = 1 - 2 * q.y * q.y - 2 * q.z * q.z;
  r.at(0, 1) = 2 * q.x * q.y + 2 * q.w * q.z;
  r.at(0, 2) = 2 * q.x * q.z - 2 * q.w * q.y;
  r.at(1, 0) = 2 * q.x * q.y - 2 * q.w * q.z;
  r.at(1, 1) = 1 - 2 * q.x * q.x - 2 * q.z * q.z;
  r.at(1, 2) = 2 * q.y * q.z + 2 * q.w * q.x;
  r.at(2, 0) = 2 * q.x * q.z + 2 * q.w * q.y;
  r.at(2, 1) = 2 * q.y * q.z - 2 * q.w * q.x;
  r.at(2, 2) = 1 - 2 * q.x * q.x - 2 * q.y * q.y;
  return r;
}

} // namespace trc
