// tracer_amd/csrc/bound.h -- host side: a world-space bounding sphere per SPHERE / CUBE / MESH object for the
// conservative (ray, object) pre-test of intersect.h (may_hit). Double arithmetic, run once per upload.
//
// The exact tests of Ray.inl work in OBJECT space on fl(m_matrixi * ray): what a ray can hit is therefore the set
// m_matrixi^-1(object) -- whatever m_matrix holds -- and the sphere is built from M' = m_matrixi^-1:
//   centre c = M'(o_c)             o_c = 0 for the unit sphere / cube, the middle of [bmin, bmax] for a mesh
//   radius rho = sigma_max(M') * r  r = 0.5 (sphere), sqrt(3)/2 (cube), half the AABB's diagonal (mesh)
// The kernels call a pair a certain miss when the ray's LINE stays farther from c than
//   rho + s,  s = kappa * (|p|_1 + |c|_1 + 2 rho),  kappa = 2^-7,  p = the ray's origin,
// or when p lies outside that inflated sphere and c lies behind p. `bound_a` = rho + kappa (|c|_1 + 2 rho), rounded up.
//
// Why that never rejects a pair the float tests accept. Notation: u = 2^-24, A = the 3x3 part of m_matrixi with singular
// values 1 / sigma_i (sigma_min <= sigma_max the object's world-space scales, cond = sigma_max / sigma_min), c0 = M'(0),
// D = |p - c|, L = the true distance of the ray's line from c, o' = A (p - c0) and d' = A d the exact object-space ray.
//   (1) What the tests see is fl(mi * p), fl(mi * (p + d)) and the re-normalised difference (Ray.inl:52-55). Each point is
//       off by <= E := 4 u (|p| + |c0|) / sigma_min (object units: the products mi * p and the translation cancel, their
//       rounding does not); the direction d' is the difference of two such points ONE world unit apart, so its angle is off
//       by theta <= 2 E / |d'|. At the object the computed ray is displaced by <= E + theta |o'|, and with p - c0 = -tau d + w,
//       w perpendicular to d, |w| ~ L: |o'| / |d'| <= D + cond L. In world units the displacement is therefore
//           <= sigma_max E (1 + 2 D + 2 cond L) = 4 u cond (|p| + |c0|) (1 + 2 D + 2 cond L)  --
//       it grows with L itself, with the distance and with the square of the conditioning: a flat or needle-shaped object
//       seen from afar is a noisy target in the reference's own arithmetic (tests/test_gpu_pairs.py found an ellipsoid of
//       scale (6, 0.1, 6) that a ray 180 units away "hits" although its line passes several units beside it).
//       An accepted pair has L <= rho + that displacement (+ (2)); with x = 8 u cond^2 (|p| + |c0|) <= 1/2 this gives
//           L <= (1 + 2 x) (rho + 4 u cond (|p| + |c0|) (1 + 2 D)),
//       which stays below rho + s/2 when   cond^2 rho <= 2^11   (then 2 x rho <= kappa (|p|_1 + |c|_1) / 4)
//                                   and    cond (1 + 2 D) <= 2^12 (then the second term is <= kappa (|p|_1 + |c|_1) / 4).
//       The first is a property of the object (checked here; x <= 1/2 likewise, from cond^2 (far + 2 |c0|_1) <= 2^20);
//       the second limits the DISTANCE up to which the pre-test may be used: `bound_far2` = far^2, far = 2^11 / cond - 1
//       world units (the reference's "point one unit along the ray" ties the precision to absolute coordinates). A ray
//       whose origin is farther from c keeps its pair.
//   (2) The tests' own rounding: slab and Moeller-Trumbore compare quantities with relative errors of a few u, i.e. their
//       acceptance region grows by a few u |o'| <= a few u cond D in world units -- far inside s. The sphere test's
//       m2 = l2 - s*s CANCELS: its absolute error is <= 10 u l2, l2 = |o'|^2 <= (D / sigma_min)^2, and the accepted radius
//       grows by at most that much in object space (sqrt(1/4 + e) <= 1/2 + e), by <= 10 u D^2 cond / sigma_min in world
//       space. That stays below kappa D / 4 while D <= 2^16/10 sigma_min / cond: spheres use far = min(the above,
//       2^11 sigma_min / cond) (a unit sphere scaled uniformly to radius r: 4096 r).
//   (3) The pre-test's own arithmetic: it takes the squared distance of the line from c as oc2 - b^2, oc = c - p,
//       b = oc . d, which is off by <= (|eta| + 4 u) oc2 for a direction with |d|^2 = 1 + eta. Since s >= kappa |oc|,
//       (rho + s)^2 - (rho + s/2)^2 >= 0.75 s^2 >= 0.75 * 2^-14 oc2: with |eta| < 2^-16 (checked per lane; a normalised
//       direction has |eta| <= 3 u, and a lane that fails the check keeps all its pairs) a rejected line is farther
//       from c than rho + s/2, which is what (1) and (2) need.
// Objects whose matrix is singular or not finite, with cond > 64, cond^2 rho > 2^11 or a scale below 1e-18 get bound_a = +inf: every ray "may
// hit" them. The margins are fat on purpose (s is ~1 % of the scene's extent in the Cornell box: a few more pairs survive).
// tests/test_gpu_pairs.py compares the pre-test with the exhaustive loop (TRC_PAIRS_CHECK builds) on the reference's scenes,
// random scenes and scenes built to sit at these limits.
#pragma once
#include <cmath>
#include <limits>
#include "scene.h"

namespace trc {

constexpr float kBoundKappa = 0x1p-7f;
constexpr double kBoundMaxCond = 64.0;

// extreme eigenvalues of a symmetric 3x3 matrix (cyclic Jacobi)
inline void sym3_eig_minmax(double a[3][3], double& lmin, double& lmax) {
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (std::fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
      }
  }
  lmin = std::fmin(a[0][0], std::fmin(a[1][1], a[2][2]));
  lmax = std::fmax(a[0][0], std::fmax(a[1][1], a[2][2]));
}

struct ObjectBound { f3 centre; float a, far2; };

inline ObjectBound object_bound(const DevObject& o) {
  const float inf = std::numeric_limits<float>::infinity();
  ObjectBound out; out.centre = mk3(0.0f, 0.0f, 0.0f); out.a = inf; out.far2 = 0.0f;
  if (o.mesh_type == REF_PLANE) return out;
  // A = 3x3 part of mi (A[row][col]), t = its translation
  double A[3][3], t[3];
  for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) A[row][col] = o.mi.c[col][row];
  for (int row = 0; row < 3; ++row) t[row] = o.mi.c[3][row];
  for (int i = 0; i < 3; ++i) { if (!std::isfinite(t[i])) return out; for (int j = 0; j < 3; ++j) if (!std::isfinite(A[i][j])) return out; }
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!(std::fabs(det) > 1e-200)) return out;
  double B[3][3];                                     // B = A^-1 = the 3x3 part of M'
  B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
  B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
  B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
  double S[3][3];                                     // B^T B
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { S[i][j] = 0; for (int k = 0; k < 3; ++k) S[i][j] += B[k][i] * B[k][j]; }
  double lmin, lmax;
  sym3_eig_minmax(S, lmin, lmax);
  if (!(lmin > 0.0) || !std::isfinite(lmax)) return out;
  const double smax = std::sqrt(lmax) * (1.0 + 1e-9), smin = std::sqrt(lmin) * (1.0 - 1e-9);
  if (!(smax / smin <= kBoundMaxCond)) return out;
  // An object scaled below ~5e-20: |m_matrixi d|^2 overflows float for a unit d, Ray.inl:55 then yields the ZERO direction,
  // and for a MESH the slab test passes whenever the object-space origin lies below bmax on every axis -- the reference
  // reports a "hit" at distance 0 (Ray.inl:171-185 with t = FLT_MAX; intersect.h keeps that case). Geometry says nothing
  // about such pairs: no bound (1 / smin is the largest |m_matrixi d| of a unit d).
  if (!(1.0 / smin <= 1e18)) return out;
  double oc[3] = { 0, 0, 0 }, r = 0.5;
  if (o.mesh_type == REF_CUBE) r = std::sqrt(0.75);
  if (o.mesh_type == REF_MESH) {
    double d2 = 0;
    const float lo[3] = { o.bmin.x, o.bmin.y, o.bmin.z }, hi[3] = { o.bmax.x, o.bmax.y, o.bmax.z };
    for (int i = 0; i < 3; ++i) {
      if (!std::isfinite(lo[i]) || !std::isfinite(hi[i])) return out;
      oc[i] = 0.5 * ((double)lo[i] + hi[i]); const double h = 0.5 * ((double)hi[i] - lo[i]); d2 += h * h;
    }
    r = std::sqrt(d2);
  }
  // centre: mi * c = o_c  ->  c = B (o_c - t)
  double c[3];
  for (int i = 0; i < 3; ++i) { c[i] = 0; for (int j = 0; j < 3; ++j) c[i] += B[i][j] * (oc[j] - t[j]); }
  const double rho = smax * r, cond = smax / smin;
  if (!(cond * cond * rho <= 2048.0)) return out;
  double far = 2048.0 / cond - 1.0;
  if (o.mesh_type == REF_SPHERE) far = std::fmin(far, 2048.0 * smin / cond);
  // x <= 1/2 in (1): c0 = B (0 - t)
  double l1c0 = 0;
  for (int i = 0; i < 3; ++i) { double v = 0; for (int j = 0; j < 3; ++j) v += B[i][j] * -t[j]; l1c0 += std::fabs(v); }
  if (!(far > 0.0) || !(cond * cond * (far + 2.0 * l1c0) <= 1048576.0)) return out;
  const double l1c = std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]);
  const double a = rho + (double)kBoundKappa * (l1c + 2.0 * rho);
  if (!std::isfinite(a) || a > 1e18) return out;
  out.centre = mk3((float)c[0], (float)c[1], (float)c[2]);     // rounding: <= u |c|, inside kappa |c|_1
  out.a = std::nextafter((float)a, inf);
  out.far2 = (float)(far * far * (1.0 - 1e-6));
  return out;
}

// MESH: a sphere in OBJECT space that holds every vertex -- for a round mesh much tighter than the AABB (an icosahedron's
// AABB lets 4.3 units of projected area through, its circumsphere 3.1), so a third of the rays that pass the AABB
// (Ray.inl:161-167) can skip the triangle loop. Centre = the AABB's middle (as floats), r = the largest vertex distance.
// intersect.h calls a ray a certain miss when m2 = l2 - s^2 (l = c - o, s = l . d, the kernel's object-space ray) exceeds
//   r^2 (1 + 2^-8) + 2^-18 l2
// -- the first term is `hull_r2`. Margins: m2 itself is off by <= 4 u l2 = 2^-22 l2 (|d| = 1 within 2 u after the
// re-normalisation of Ray.inl:55); Moeller-Trumbore (Ray.inl:189-226) accepts barycentric coordinates that are off by a few
// u |o - v0| / |e|, i.e. points up to delta ~ 10 u sqrt(l2) outside the triangle: (r + delta)^2 - r^2 <= 2^-9 r^2 + 2^9 delta^2
// <= 2^-9 r^2 + 2^-33 l2. Checked like the pre-test (TRC_PAIRS_CHECK builds count culled rays that hit a triangle).
struct MeshHull { f3 centre; float r2; };
inline MeshHull mesh_hull(const DevObject& o, const DevTri* tris) {
  MeshHull h;
  h.centre = mk3(0.5f * (o.bmin.x + o.bmax.x), 0.5f * (o.bmin.y + o.bmax.y), 0.5f * (o.bmin.z + o.bmax.z));
  double r2 = 0.0;
  bool ok = std::isfinite(h.centre.x) && std::isfinite(h.centre.y) && std::isfinite(h.centre.z);
  for (uint32_t t = 0; t < o.tri_count && ok; ++t) {
    const DevTri& T = tris[o.tri_first + t];
    const double v[3][3] = { { T.v0.x, T.v0.y, T.v0.z },
                             { (double)T.v0.x + T.e1.x, (double)T.v0.y + T.e1.y, (double)T.v0.z + T.e1.z },      // e1 = v1 - v0 rounded: <= u |v|, inside the margin
                             { (double)T.v0.x + T.e2.x, (double)T.v0.y + T.e2.y, (double)T.v0.z + T.e2.z } };
    for (int k = 0; k < 3; ++k) {
      const double dx = v[k][0] - h.centre.x, dy = v[k][1] - h.centre.y, dz = v[k][2] - h.centre.z;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (!std::isfinite(d2)) ok = false;
      r2 = std::fmax(r2, d2);
    }
  }
  const double padded = r2 * (1.0 + 1.0 / 256.0);
  h.r2 = (ok && padded < 1e37) ? std::nextafter((float)padded, std::numeric_limits<float>::infinity()) : std::numeric_limits<float>::infinity();
  return h;
}

}  // namespace trc
