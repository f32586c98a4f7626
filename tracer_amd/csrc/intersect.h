// tracer_amd/csrc/intersect.h -- closest-hit for gfx950: the arithmetic of the reference's
// src/Ray.inl in its exact operation order, over the flat scene records of scene.h.
#pragma once
#include "scene.h"

#pragma clang fp contract(off)

namespace trc {

__device__ __forceinline__ Hit no_hit() {   // Ray.inl:30  Hit() : m_t(-1), m_id(-1), vec3s zero
  Hit h; h.t = -1.0f; h.id = -1; h.pos = splat3(0.0f); h.nor = splat3(0.0f); return h;
}

// Ray.inl:50-57: two point transforms, direction RE-normalised in object space
__device__ __forceinline__ Ray to_object_space(const Ray& ray, const Affine& mi) {
  f3 pos = xform_point(mi, ray.pos);
  f3 point = xform_point(mi, ray.pos + ray.dir);
  Ray r; r.pos = pos; r.dir = normalize(point - pos);
  return r;
}

// Ray.inl:189-226 Moeller-Trumbore with barycentric normal interpolation and back-face cull.
// Returns t (or -1 for "no hit", the m_t of a default Hit) and the un-normalised normal.
__device__ __forceinline__ float hit_triangle(const Ray& ray, const DevTri& tri, f3& nor) {
  f3 pvec = cross(ray.dir, tri.e2);
  float det = dot(tri.e1, pvec);
  if (gabs(det) < kEps) return -1.0f;
  det = 1.0f / det;
  f3 tvec = ray.pos - tri.v0;
  float bx = dot(tvec, pvec) * det;
  if (bx < 0.0f || bx > 1.0f) return -1.0f;
  f3 qvec = cross(tvec, tri.e1);
  float by = dot(ray.dir, qvec) * det;
  if (by < 0.0f || (bx + by) > 1.0f) return -1.0f;
  float t = dot(tri.e2, qvec) * det;
  float bz = 1.0f - bx - by;
  f3 n = tri.n0 * bz + tri.n1 * bx + tri.n2 * by;
  if (dot(ray.dir, n) > 0) return -1.0f;
  nor = n;
  return t;
}

// Slab test shared by CUBE (Ray.inl:121-127) and MESH (Ray.inl:161-167).
__device__ __forceinline__ bool slab(const Ray& ray, f3 bmin, f3 bmax, float& tNear, float& tFar) {
  f3 tMin = (bmin - ray.pos) / ray.dir;
  f3 tMax = (bmax - ray.pos) / ray.dir;
  f3 t1 = gmin(tMin, tMax);
  f3 t2 = gmax(tMin, tMax);
  tNear = gmax(gmax(t1.x, t1.y), t1.z);
  tFar = gmin(gmin(t2.x, t2.y), t2.z);
  return !(tNear > tFar || tFar < 0.0f);
}

// Object-space primitive test, Ray.inl:102-187. `t < 0` means no hit (Ray.inl:93).
// For MESH the triangle loop runs over the pre-gathered stream in index order with the
// reference's strict `<`, so the lowest face index wins ties (Ray.inl:174).
__device__ __forceinline__ float hit_primitive(const Ray& r, const DevObject& o,
                                               const DevTri* __restrict__ tris,
                                               f3& pos, f3& nor, uint32_t& tri_tests) {
  const int type = o.mesh_type;
  if (type == REF_SPHERE) {                                   // Ray.inl:103-118
    f3 l = -r.pos;
    float s = dot(l, r.dir);
    float l2 = dot(l, l);
    if (s < 0.0f && l2 > 0.25f) return -1.0f;
    float m2 = l2 - s * s;
    if (m2 > 0.25f) return -1.0f;
    float q = sqrtf(0.25f - m2);
    float t = (l2 > 0.25f) ? s - q : s + q;
    pos = r.pos + r.dir * t;
    nor = normalize(pos);
    return t;
  } else if (type == REF_CUBE) {                              // Ray.inl:120-140
    float tNear, tFar;
    if (!slab(r, splat3(-0.5f), splat3(0.5f), tNear, tFar)) return -1.0f;
    float t = tNear > 0.0f ? tNear : tFar;
    pos = r.pos + r.dir * t;
    if      (pos.x < -0.5f + kEps) nor = mk3(-1.0f, 0.0f, 0.0f);
    else if (pos.x >  0.5f - kEps) nor = mk3( 1.0f, 0.0f, 0.0f);
    else if (pos.y < -0.5f + kEps) nor = mk3( 0.0f, -1.0f, 0.0f);
    else if (pos.y >  0.5f - kEps) nor = mk3( 0.0f, 1.0f, 0.0f);
    else if (pos.z < -0.5f + kEps) nor = mk3( 0.0f, 0.0f, -1.0f);
    else                           nor = mk3( 0.0f, 0.0f, 1.0f);
    return t;
  } else if (type == REF_PLANE) {                             // Ray.inl:142-158
    float t = -r.pos.y / r.dir.y;
    f3 p = r.pos + t * r.dir;
    if (p.x < 0.5f && p.z < 0.5f && p.x > -0.5f && p.z > -0.5f) {
      pos = p; nor = mk3(0.0f, 1.0f, 0.0f);
      return t;                                               // sign tested by the caller (:93)
    }
    return -1.0f;
  } else {                                                    // MESH, Ray.inl:160-186
    float tNear, tFar;
    if (!slab(r, o.bmin, o.bmax, tNear, tFar)) return -1.0f;
    float best = FLT_MAX;
    f3 bn = splat3(0.0f);
    const uint32_t first = o.tri_first, count = o.tri_count;
    for (uint32_t i = 0; i < count; ++i) {
      f3 n;
      float t = hit_triangle(r, tris[first + i], n);
      if (t > 0.0f && t < best) { best = t; bn = n; }
    }
    tri_tests += count;
    // Ray.inl:171-185: with no triangle hit the reference carries t = FLT_MAX and a zero
    // normal forward; the world-space distance test then rejects it because
    // |o + d*FLT_MAX|^2 overflows to inf for any normalised d.
    if (best == FLT_MAX) return -1.0f;
    pos = r.pos + r.dir * best;
    nor = bn;
    return best;
  }
}

// Ray::intersectScene, Ray.inl:66-84: closest by WORLD-space squared distance (not by t),
// strict `<` so the lowest object index wins ties; the surviving t is the object-space one.
__device__ __forceinline__ Hit closest_hit(const Ray& ray,
                                           const DevObject* __restrict__ objs, uint32_t nobj,
                                           const DevTri* __restrict__ tris, uint32_t& tri_tests) {
  float mindist = FLT_MAX;
  Hit best = no_hit();
  for (uint32_t i = 0; i < nobj; ++i) {
    const DevObject& o = objs[i];
    Ray r = to_object_space(ray, o.mi);                       // Ray.inl:88
    f3 opos, onor;
    float t = hit_primitive(r, o, tris, opos, onor, tri_tests);
    if (t < 0.0f) continue;                                   // Ray.inl:93
    // Ray.inl:59-64: position by the model matrix, normal by the model matrix too
    // (not its inverse transpose), then normalised.
    f3 wpos = xform_point(o.m, opos);
    f3 wnor = normalize(xform_dir(o.m, onor));
    if (t > 0.0f) {                                           // Ray.inl:72
      f3 d = ray.pos - wpos;
      float dist = dot(d, d);
      if (dist < mindist) { mindist = dist; best.t = t; best.id = (int)i; best.pos = wpos; best.nor = wnor; }
    }
  }
  return best;
}

} // namespace trc
