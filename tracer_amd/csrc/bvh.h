// tracer_amd/csrc/bvh.h -- optional acceleration structure for big MESH objects.
//
// The reference tests every face of a MESH for every ray that passes its AABB (Ray.inl:172-177);
// that loop stays the default (kernels.hip: k_mesh_brute). With TRC_OPT_MESH_ACCEL = 1 the same
// contract -- closest positive t over all faces by the reference's own Moeller-Trumbore arithmetic,
// lowest face index among equal t -- is met by visiting only the leaves of a bounding-volume
// hierarchy whose boxes the ray's LINE comes within `pad` of. The exact test still decides; the
// hierarchy only has to be conservative about the FLOATING-POINT test, which can accept a ray that
// geometrically misses a triangle by a rounding error. Bound on that error (object space, |d| = 1,
// eps = 2^-24, EPS = 0.001 the |det| threshold of Ray.inl:196, E = longest edge of the mesh):
//   u = dot(tvec, pvec), pvec = cross(d, e2): |du| <= 7 eps |tvec| E;  |ddet| <= 6 eps E^2
//   bary.x = u / det with |det| >= EPS: |d bary.x| <= (7 eps |tvec| E + 6 eps E^2) / EPS + 2 eps, same for bary.y
//   so an accepted (bary.x, bary.y) is within eta = 14 eps E (|tvec| + E) / EPS of the exact one, and the
//   exact point v0 + x e1 + y e2 on the ray's line lies within 2 eta E of the triangle.
// With |tvec| <= |o - c| + R (c, R: the mesh's bounding sphere) the traversal pads every box by
//   pad(o) = kPadScale * E^2 * (|o - c| + R + E),   kPadScale = 4 * 14 eps / EPS = 3.4e-3   (2x safety),
// plus 1e-5 * (|o - c| + R) for the rounding of the slab test itself (50x its 3 eps relative error).
// Nodes are pruned by distance only when they start beyond best_t + 2 pad, so every face that ties
// the best t is still tested and the (t, face) key picks the lowest index.
#pragma once
#include "scene.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace trc {

struct alignas(16) BvhNode {        // 32 bytes
  f3 bmin; uint32_t first;          // inner: index of the left child (right = first + 1); leaf: first entry in the face list
  f3 bmax; uint32_t count;          // leaf: number of faces (> 0); inner: 0
};
static_assert(sizeof(BvhNode) == 32, "BvhNode");
constexpr uint32_t kBvhLeaf = 4;    // faces per leaf
constexpr int kBvhMaxDepth = 40;    // traversal stack entries per lane (median split: depth <= log2(n / leaf) + 2)
constexpr float kPadScale = 3.4e-3f;

struct BvhParams { f3 centre; float radius; float max_edge; uint32_t n_nodes; };

// Median split on the longest axis of the centroid bounds; children of a node are adjacent.
inline void build_bvh(const DevTri* tris, uint32_t n, std::vector<BvhNode>& nodes, std::vector<uint32_t>& faces,
                      BvhParams& prm) {
  struct Box { float lo[3], hi[3]; };
  std::vector<Box> tb(n);
  std::vector<float> cen(3ull * n);
  double emax = 0.0;
  Box all = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
  for (uint32_t i = 0; i < n; ++i) {
    const DevTri& t = tris[i];
    const double v[3][3] = { { t.v0.x, t.v0.y, t.v0.z },
                             { (double)t.v0.x + t.e1.x, (double)t.v0.y + t.e1.y, (double)t.v0.z + t.e1.z },
                             { (double)t.v0.x + t.e2.x, (double)t.v0.y + t.e2.y, (double)t.v0.z + t.e2.z } };
    for (int a = 0; a < 3; ++a) {
      const double lo = std::min(v[0][a], std::min(v[1][a], v[2][a])), hi = std::max(v[0][a], std::max(v[1][a], v[2][a]));
      tb[i].lo[a] = (float)std::nextafter((float)lo, -INFINITY); tb[i].hi[a] = (float)std::nextafter((float)hi, INFINITY);
      cen[3ull * i + a] = (float)(0.5 * (lo + hi));
      all.lo[a] = std::min(all.lo[a], tb[i].lo[a]); all.hi[a] = std::max(all.hi[a], tb[i].hi[a]);
    }
    const double e1 = std::sqrt((double)t.e1.x * t.e1.x + (double)t.e1.y * t.e1.y + (double)t.e1.z * t.e1.z);
    const double e2 = std::sqrt((double)t.e2.x * t.e2.x + (double)t.e2.y * t.e2.y + (double)t.e2.z * t.e2.z);
    const double dx = (double)t.e2.x - t.e1.x, dy = (double)t.e2.y - t.e1.y, dz = (double)t.e2.z - t.e1.z;
    emax = std::max(emax, std::max(std::max(e1, e2), std::sqrt(dx * dx + dy * dy + dz * dz)));
  }
  prm.centre = mk3(0.5f * (all.lo[0] + all.hi[0]), 0.5f * (all.lo[1] + all.hi[1]), 0.5f * (all.lo[2] + all.hi[2]));
  const double hx = 0.5 * ((double)all.hi[0] - all.lo[0]), hy = 0.5 * ((double)all.hi[1] - all.lo[1]), hz = 0.5 * ((double)all.hi[2] - all.lo[2]);
  prm.radius = (float)(std::sqrt(hx * hx + hy * hy + hz * hz) * 1.000001);
  prm.max_edge = (float)(emax * 1.000001);
  faces.resize(n);
  for (uint32_t i = 0; i < n; ++i) faces[i] = i;
  nodes.clear();
  nodes.reserve(2ull * n / kBvhLeaf + 16);
  nodes.push_back(BvhNode{});
  struct Job { uint32_t node, lo, hi; };
  std::vector<Job> todo{ { 0u, 0u, n } };
  while (!todo.empty()) {
    const Job j = todo.back(); todo.pop_back();
    Box b = { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } }, cb = b;
    for (uint32_t k = j.lo; k < j.hi; ++k) {
      const uint32_t f = faces[k];
      for (int a = 0; a < 3; ++a) {
        b.lo[a] = std::min(b.lo[a], tb[f].lo[a]); b.hi[a] = std::max(b.hi[a], tb[f].hi[a]);
        cb.lo[a] = std::min(cb.lo[a], cen[3ull * f + a]); cb.hi[a] = std::max(cb.hi[a], cen[3ull * f + a]);
      }
    }
    BvhNode& nd = nodes[j.node];
    nd.bmin = mk3(b.lo[0], b.lo[1], b.lo[2]); nd.bmax = mk3(b.hi[0], b.hi[1], b.hi[2]);
    const uint32_t cnt = j.hi - j.lo;
    if (cnt <= kBvhLeaf) { nd.first = j.lo; nd.count = cnt; continue; }
    int axis = 0;
    if (cb.hi[1] - cb.lo[1] > cb.hi[axis] - cb.lo[axis]) axis = 1;
    if (cb.hi[2] - cb.lo[2] > cb.hi[axis] - cb.lo[axis]) axis = 2;
    const uint32_t mid = j.lo + cnt / 2;
    std::nth_element(faces.begin() + j.lo, faces.begin() + mid, faces.begin() + j.hi,
                     [&](uint32_t x, uint32_t y) { const float cx = cen[3ull * x + axis], cy = cen[3ull * y + axis]; return cx < cy || (cx == cy && x < y); });
    const uint32_t left = (uint32_t)nodes.size();
    nodes.push_back(BvhNode{}); nodes.push_back(BvhNode{});
    nodes[j.node].first = left; nodes[j.node].count = 0;
    todo.push_back({ left, j.lo, mid });
    todo.push_back({ left + 1, mid, j.hi });
  }
  prm.n_nodes = (uint32_t)nodes.size();
}

// Bounding spheres for the exhaustive pass (kernels.hip: k_mesh_brute<true>): per face a centre (the centroid,
// as rounded to float) and, for the whole mesh, r_max = the largest distance from a face's centre to one of its
// vertices -- so every point of a face lies within r_max of its centre. With the bound above, a (ray, face) pair
// that the floating-point test accepts has the ray's line within pad(o) / 2 of the face, hence within
// r_max + pad(o) of the centre; the pass skips the exact test for a pair only when the line's distance from the
// centre is beyond that, with margins for the rounding of the distance itself (k_mesh_brute).
// `centres` gets one float4 per face (w = 0); prm as in build_bvh; origin_dist = |centre of the mesh|.
inline void mesh_bound_spheres(const DevTri* tris, uint32_t n, std::vector<float>& centres, BvhParams& prm,
                               float& r_max, float& origin_dist) {
  double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
  double emax = 0.0, rmax = 0.0;
  centres.assign(4ull * n, 0.0f);
  for (uint32_t i = 0; i < n; ++i) {
    const DevTri& t = tris[i];
    const double v[3][3] = { { t.v0.x, t.v0.y, t.v0.z },
                             { (double)t.v0.x + t.e1.x, (double)t.v0.y + t.e1.y, (double)t.v0.z + t.e1.z },
                             { (double)t.v0.x + t.e2.x, (double)t.v0.y + t.e2.y, (double)t.v0.z + t.e2.z } };
    float c[3];
    for (int a = 0; a < 3; ++a) {
      c[a] = (float)((v[0][a] + v[1][a] + v[2][a]) / 3.0);
      centres[4ull * i + a] = c[a];
      for (int k = 0; k < 3; ++k) { lo[a] = std::min(lo[a], v[k][a]); hi[a] = std::max(hi[a], v[k][a]); }
    }
    for (int k = 0; k < 3; ++k) {
      const double dx = v[k][0] - c[0], dy = v[k][1] - c[1], dz = v[k][2] - c[2];
      rmax = std::max(rmax, std::sqrt(dx * dx + dy * dy + dz * dz));
      const int k2 = (k + 1) % 3;
      const double ex = v[k][0] - v[k2][0], ey = v[k][1] - v[k2][1], ez = v[k][2] - v[k2][2];
      emax = std::max(emax, std::sqrt(ex * ex + ey * ey + ez * ez));
    }
  }
  if (n == 0) { for (int a = 0; a < 3; ++a) { lo[a] = 0.0; hi[a] = 0.0; } }
  const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
  const double hx = 0.5 * (hi[0] - lo[0]), hy = 0.5 * (hi[1] - lo[1]), hz = 0.5 * (hi[2] - lo[2]);
  prm.centre = mk3((float)cx, (float)cy, (float)cz);
  prm.radius = (float)(std::sqrt(hx * hx + hy * hy + hz * hz) * 1.00001 + 1e-6 * (std::fabs(cx) + std::fabs(cy) + std::fabs(cz)));
  prm.max_edge = (float)(emax * 1.00001);
  prm.n_nodes = 0;
  r_max = (float)(rmax * 1.00001);
  origin_dist = (float)(std::sqrt(cx * cx + cy * cy + cz * cz) * 1.00001);
}

} // namespace trc
