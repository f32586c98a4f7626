// tests/cpp/bound_check.cpp -- host check of tracer_amd/csrc/bound.h (plain C++, no GPU): for random object transforms the
// world-space bounding sphere of the (ray, object) pre-test must hold the object as the kernels see it, i.e. the image of the
// object under the inverse of m_matrixi -- and the rules that switch the bound off (conditioning, far limit) must hold.
//   bound_check [n_transforms = 20000] [seed = 1]      prints "ok ..." and returns 0, or the first violation and 1
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../tracer_amd/csrc/bound.h"

using namespace trc;

static void inv3(const double A[3][3], double B[3][3], double& det) {
  det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
        A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
  B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
  B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(argc > 2 ? atoi(argv[2]) : 1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  int bounded = 0, unbounded = 0;
  for (int it = 0; it < n; ++it) {
    // object -> world: rotation * scale * rotation (* a little shear), translation; scales over six decades, ratios up to 200
    double M[3][3] = { { 0 } }, t[3];
    auto rot = [&](double R[3][3]) {
      double q[4]; double s = 0; for (double& v : q) { v = N(rng); s += v * v; } s = std::sqrt(s); for (double& v : q) v /= s;
      R[0][0] = 1 - 2 * (q[2] * q[2] + q[3] * q[3]); R[0][1] = 2 * (q[1] * q[2] - q[0] * q[3]); R[0][2] = 2 * (q[1] * q[3] + q[0] * q[2]);
      R[1][0] = 2 * (q[1] * q[2] + q[0] * q[3]); R[1][1] = 1 - 2 * (q[1] * q[1] + q[3] * q[3]); R[1][2] = 2 * (q[2] * q[3] - q[0] * q[1]);
      R[2][0] = 2 * (q[1] * q[3] - q[0] * q[2]); R[2][1] = 2 * (q[2] * q[3] + q[0] * q[1]); R[2][2] = 1 - 2 * (q[1] * q[1] + q[2] * q[2]);
    };
    double R1[3][3], R2[3][3], sc[3];
    rot(R1); rot(R2);
    const double base = std::pow(10.0, 3.0 * U(rng)), ratio = std::pow(200.0, std::fabs(U(rng)));
    sc[0] = base; sc[1] = base * std::pow(ratio, U(rng) * 0.5 + 0.5); sc[2] = base * ratio;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[i][j] = 0; for (int k = 0; k < 3; ++k) M[i][j] += R1[i][k] * sc[k] * R2[k][j]; }
    for (double& v : t) v = base * 50.0 * U(rng) * (it % 5 == 0 ? 100.0 : 1.0);
    double Mi[3][3], det;
    inv3(M, Mi, det);
    DevObject o{};
    for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) { o.mi.c[col][row] = (float)Mi[row][col]; o.m.c[col][row] = (float)M[row][col]; }
    for (int row = 0; row < 3; ++row) {
      double v = 0; for (int k = 0; k < 3; ++k) v -= Mi[row][k] * t[k];
      o.mi.c[3][row] = (float)v; o.m.c[3][row] = (float)t[row];
    }
    o.mesh_type = it % 3 == 0 ? REF_SPHERE : it % 3 == 1 ? REF_CUBE : REF_MESH;
    o.bmin = mk3((float)(-1.0 - U(rng) * 0.5), (float)(-0.5 + U(rng) * 0.2), (float)(-2.0 + U(rng)));
    o.bmax = mk3((float)(1.0 + U(rng) * 0.5), (float)(0.5 + U(rng) * 0.2), (float)(2.0 + U(rng)));
    const ObjectBound b = object_bound(o);
    // the object as the kernels see it: x with mi(x) in the object, mi as stored (floats)
    double A[3][3], B[3][3], ta[3], d2;
    for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) A[row][col] = o.mi.c[col][row];
    for (int row = 0; row < 3; ++row) ta[row] = o.mi.c[3][row];
    inv3(A, B, d2);
    const double smax = sc[2], smin = sc[0], cond = smax / smin;
    // the documented rules (bound.h), from the singular values this transform was built with; transforms within 2 % of
    // a threshold are not judged
    const double r_obj = o.mesh_type == REF_SPHERE ? 0.5 : o.mesh_type == REF_CUBE ? std::sqrt(0.75) :
        0.5 * std::sqrt(((double)o.bmax.x - o.bmin.x) * ((double)o.bmax.x - o.bmin.x) + ((double)o.bmax.y - o.bmin.y) * ((double)o.bmax.y - o.bmin.y) +
                        ((double)o.bmax.z - o.bmin.z) * ((double)o.bmax.z - o.bmin.z));
    double far_rule = 2048.0 / cond - 1.0;
    if (o.mesh_type == REF_SPHERE) far_rule = std::fmin(far_rule, 2048.0 * smin / cond);
    const double l1c0 = std::fabs(t[0]) + std::fabs(t[1]) + std::fabs(t[2]);
    const double margins[4] = { cond / 64.0, cond * cond * smax * r_obj / 2048.0, far_rule > 0 ? 0.0 : 2.0, cond * cond * (far_rule + 2.0 * l1c0) / 1048576.0 };
    bool expect = true, near_limit = false;
    for (double m : margins) { if (m > 1.0) expect = false; if (m > 0.98 && m < 1.02) near_limit = true; }
    if (!near_limit && expect != std::isfinite(b.a)) {
      printf("transform %d: bound %s, rules say %s (cond %.2f, sigma_max %.3g, |c0|_1 %.3g)\n", it, std::isfinite(b.a) ? "present" : "absent",
             expect ? "present" : "absent", cond, smax, l1c0);
      return 1;
    }
    if (!std::isfinite(b.a)) { ++unbounded; continue; }
    ++bounded;
    for (int s = 0; s < 64; ++s) {
      double q[3];
      if (o.mesh_type == REF_SPHERE) { double l = 0; for (double& v : q) { v = N(rng); l += v * v; } l = std::sqrt(l); for (double& v : q) v *= 0.5 / l; }
      else if (o.mesh_type == REF_CUBE) { for (int k = 0; k < 3; ++k) q[k] = (s >> k) & 1 ? 0.5 : -0.5; }
      else { const float lo[3] = { o.bmin.x, o.bmin.y, o.bmin.z }, hi[3] = { o.bmax.x, o.bmax.y, o.bmax.z }; for (int k = 0; k < 3; ++k) q[k] = (s >> k) & 1 ? hi[k] : lo[k]; }
      double x[3], dist2 = 0;
      for (int i = 0; i < 3; ++i) { x[i] = 0; for (int k = 0; k < 3; ++k) x[i] += B[i][k] * (q[k] - ta[k]); }
      const double c[3] = { b.centre.x, b.centre.y, b.centre.z };
      for (int i = 0; i < 3; ++i) dist2 += (x[i] - c[i]) * (x[i] - c[i]);
      if (std::sqrt(dist2) > (double)b.a) { printf("transform %d: point %.6g from the centre, bound_a %.6g\n", it, std::sqrt(dist2), (double)b.a); return 1; }
    }
    const double far = std::sqrt((double)b.far2);
    if (far > 2048.0 / cond * 1.01) { printf("transform %d: far limit %.4g with cond %.2f\n", it, far, cond); return 1; }
  }
  // mesh_hull: every vertex inside
  for (int it = 0; it < 2000; ++it) {
    DevTri tris[8]; DevObject o{};
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    const double off[3] = { 100.0 * U(rng), 100.0 * U(rng), 100.0 * U(rng) };
    float verts[8][3][3];
    for (int k = 0; k < 8; ++k) for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) {
      verts[k][v][a] = (float)(off[a] + U(rng) * (a == 1 ? 0.1 : 3.0));
      lo[a] = std::fmin(lo[a], verts[k][v][a]); hi[a] = std::fmax(hi[a], verts[k][v][a]);
    }
    for (int k = 0; k < 8; ++k) {
      tris[k].v0 = mk3(verts[k][0][0], verts[k][0][1], verts[k][0][2]);
      tris[k].e1 = mk3(verts[k][1][0], verts[k][1][1], verts[k][1][2]) - tris[k].v0;
      tris[k].e2 = mk3(verts[k][2][0], verts[k][2][1], verts[k][2][2]) - tris[k].v0;
    }
    o.bmin = mk3(lo[0], lo[1], lo[2]); o.bmax = mk3(hi[0], hi[1], hi[2]); o.tri_first = 0; o.tri_count = 8; o.mesh_type = REF_MESH;
    const MeshHull h = mesh_hull(o, tris);
    for (int k = 0; k < 8; ++k) for (int v = 0; v < 3; ++v) {
      double d2 = 0; const double c[3] = { h.centre.x, h.centre.y, h.centre.z };
      for (int a = 0; a < 3; ++a) d2 += (verts[k][v][a] - c[a]) * (verts[k][v][a] - c[a]);
      if (d2 > (double)h.r2) { printf("mesh %d: vertex at %.6g^2 outside hull r2 %.6g\n", it, d2, (double)h.r2); return 1; }
    }
  }
  printf("ok: %d transforms (%d with a bound, %d without), 2000 mesh hulls\n", n, bounded, unbounded);
  return 0;
}
