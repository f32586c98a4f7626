// tracer_amd/csrc/sparse_xform.h -- matrix * vector products that leave out entries which are exactly zero.
//
// An object placed by scale + translate has six exact zeros in the 3x3 part of m_matrix and m_matrixi; one
// rotated about a coordinate axis has four (the reference's own scene: main.cpp:289-393 -- only the
// icosahedron's matrices are dense). The reference still multiplies and adds them (type_mat4x4.inl:695-699:
// ((m0*x + m1*y) + m2*z) + m3*w, left to right). A term m*v with m == +-0 and v finite is +-0, and adding
// +-0 changes a sum only when the sum is itself a zero: walking both evaluations term by term, the partial
// sums are equal, or both are zeros of possibly different sign, and they become equal for good at the
// first kept term that is not zero. So
//     the product without the zero entries' terms is bit-identical to the reference's
//     if every skipped coordinate is finite and each row has one kept term that is not zero.
// The host guarantees the matrix side (classify_xform: every row keeps an entry with 2^-60 <= |m| < 2^60,
// everything finite); the kernels guarantee the vector side per wavefront (all coordinates of the lanes
// that matter in [2^-60, 2^60), intersect.h: coords_ok) and take the full product otherwise -- e.g. for
// the primary rays of a pinhole camera standing at x = 0. Used for m_matrixi in Ray.inl:52-53 (two
// products per object and segment); the hit's way back through m_matrix stays the full product.
#pragma once
#include "vecmath.h"

namespace trc {

// Zero patterns of the 3x3 part: off-diagonal zeros (scale), or one axis decoupled from the other two
// (rotation about that axis, any scale). c[col][row] as in Affine.
enum XformPattern : uint32_t { XP_FULL = 0, XP_DIAG = 1, XP_BLOCK_X = 2, XP_BLOCK_Y = 3, XP_BLOCK_Z = 4 };

constexpr bool xp_zero(uint32_t P, int col, int row) {
  return P == XP_DIAG ? col != row
       : P == XP_BLOCK_X ? (col == 0) != (row == 0)
       : P == XP_BLOCK_Y ? (col == 1) != (row == 1)
       : P == XP_BLOCK_Z ? (col == 2) != (row == 2)
       : false;
}

// the kept terms of row ROW, summed left to right
template <uint32_t P, int ROW>
TRC_HD float xp_row(const Affine& m, f3 p) {
  constexpr bool k0 = !xp_zero(P, 0, ROW), k1 = !xp_zero(P, 1, ROW), k2 = !xp_zero(P, 2, ROW);
  static_assert(k0 || k1 || k2, "a row keeps at least one term");
  float s;
  if constexpr (k0) {
    s = m.c[0][ROW] * p.x;
    if constexpr (k1) s = s + m.c[1][ROW] * p.y;
    if constexpr (k2) s = s + m.c[2][ROW] * p.z;
  } else if constexpr (k1) {
    s = m.c[1][ROW] * p.y;
    if constexpr (k2) s = s + m.c[2][ROW] * p.z;
  } else {
    s = m.c[2][ROW] * p.z;
  }
  return s;
}
// xform_point (vecmath.h) for a matrix of pattern P
template <uint32_t P>
TRC_HD f3 xform_point_sparse(const Affine& m, f3 p) {
  if constexpr (P == XP_FULL) return xform_point(m, p);
  else return mk3(xp_row<P, 0>(m, p) + m.c[3][0], xp_row<P, 1>(m, p) + m.c[3][1], xp_row<P, 2>(m, p) + m.c[3][2]);
}
// Host: the pattern with the most zeros that matrix `a` satisfies under the conditions above.
inline uint32_t classify_xform(const Affine& a) {
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 3; ++row)
      if (!std::isfinite(a.c[col][row])) return XP_FULL;
  const uint32_t order[4] = { XP_DIAG, XP_BLOCK_X, XP_BLOCK_Y, XP_BLOCK_Z };
  for (uint32_t P : order) {
    bool ok = true;
    for (int row = 0; row < 3 && ok; ++row) {
      bool anchor = false;                       // a kept entry whose term cannot vanish
      for (int col = 0; col < 3; ++col) {
        const float v = std::fabs(a.c[col][row]);
        if (xp_zero(P, col, row)) { if (v != 0.0f) ok = false; }
        else if (v >= 0x1p-60f && v < 0x1p60f) anchor = true;
      }
      ok = ok && anchor;
    }
    if (ok) return P;
  }
  return XP_FULL;
}

} // namespace trc
