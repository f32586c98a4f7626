// tracer_amd/csrc/intersect.h -- closest-hit for gfx950: the arithmetic of the reference's
// src/Ray.inl in its exact operation order, over the flat scene records of scene.h.
//
// closest_hit() is WAVE-CONVERGENT: all 64 lanes of a wavefront must call it together (lanes
// without a live path pass live = false). That lets the MESH test turn the few lanes whose ray
// passed the AABB into wave-wide work: the triangles of a 64-triangle chunk are spread over the
// lanes and each passing ray is broadcast to them in turn (lane <-> triangle instead of
// lane <-> ray), with the reference's "first index wins" tie rule kept by a wave-wide
// min + lowest-lane select.
#pragma once
#include "scene.h"
#include "crmath.h"

#pragma clang fp contract(off)

#ifdef TRC_MARKERS
#define TRC_MARK(name) asm volatile("; MARK " name)
#else
#define TRC_MARK(name)
#endif

namespace trc {

// ---- wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ float read_lane(float v, int lane) {   // lane must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned umin_dpp(unsigned v) {
  unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xf, false);
  return o < v ? o : v;
}
// minimum over the 64 lanes, returned as a wave-uniform value (row_shr 1,2,4,8; row_bcast 15, 31)
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
  v = umin_dpp<0x111, 0xf>(v);
  v = umin_dpp<0x112, 0xf>(v);
  v = umin_dpp<0x114, 0xf>(v);
  v = umin_dpp<0x118, 0xf>(v);
  v = umin_dpp<0x142, 0xa>(v);
  v = umin_dpp<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}


// glm::normalize (func_geometric.inl:243-244) on the device
__device__ __forceinline__ f3 normalize_dev(f3 v, lmask need) {
  const float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
  return v * cr_rsqrt(sqr, need);
}

// Wave-uniform record -> SGPRs. The scene is read-only for the whole launch, but the compiler cannot know
// that a kernel's stores never alias it and falls back to per-lane vector loads of the same address
// (global_load_dwordx4 + s_waitcnt vmcnt(0) in front of every object). Reading through the constant
// address space states the invariance: the copy becomes s_load_dwordx8/x16 and the matrix entries are used
// as scalar operands of the VALU instructions. `p` must be wave-uniform.
template <typename T>
__device__ __forceinline__ T load_uniform(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  typedef const __attribute__((address_space(4))) unsigned int* const_dwords;
  const const_dwords q = (const_dwords)(unsigned long long)p;
  T out;
  unsigned int* o = reinterpret_cast<unsigned int*>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = q[i];
  return out;
}

// Where a loop body takes an object's wave-uniform matrix from.
//   XF_SGPR: scalar loads (s_load_dwordx4/x8 through the constant address space) -> the entries are SGPR operands of the
//            VALU instructions. No VGPRs, but a v_mul_f32 / v_add_f32 / v_fma_f32 with an SGPR source issues in 4.25 cycles
//            per wavefront instead of 2.5 (tools/micro/issue_rate.hip) -- and each entry feeds two of them per trip;
//   XF_LDS : broadcast ds_read_b128 from the kernel's LDS copy of the DevXform table -> the entries are VGPRs (12 per
//            object while it is being tested), the arithmetic runs at the full rate and no scalar-cache wait precedes it.
enum XfSource { XF_SGPR = 0, XF_LDS = 1 };
template <int SRC>
__device__ __forceinline__ Affine object_mi(const DevXform* __restrict__ xf, const DevXform* __restrict__ xl, uint32_t k) {
  if (SRC == XF_LDS) return xl[k].mi;
  return load_uniform(&xf[k].mi);
}

__device__ __forceinline__ Hit no_hit() {   // Ray.inl:30  Hit() : m_t(-1), m_id(-1), vec3s zero
  Hit h; h.t = -1.0f; h.id = -1; h.pos = splat3(0.0f); h.nor = splat3(0.0f); return h;
}

// The vector-side condition of sparse_xform.h: the lanes whose six coordinates are all finite and in
// [2^-60, 2^60) (the sum catches NaN, inf and big values; v_min skips NaNs, which the sum has already caught).
__device__ __forceinline__ lmask coords_ok(f3 a, f3 b) {
  const float sum = __builtin_fabsf(a.x) + __builtin_fabsf(a.y) + __builtin_fabsf(a.z) +
                    __builtin_fabsf(b.x) + __builtin_fabsf(b.y) + __builtin_fabsf(b.z);
  const float la = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
  const float lb = __builtin_fminf(__builtin_fminf(__builtin_fabsf(b.x), __builtin_fabsf(b.y)), __builtin_fabsf(b.z));
  return TRC_FCMP(sum, 0x1p60f, F_OLT) & TRC_FCMP(__builtin_fminf(la, lb), 0x1p-60f, F_OGE);
}

// Ray.inl:50-57: two point transforms, direction RE-normalised in object space.
// P: a zero pattern of `mi` (sparse_xform.h). Anything but XP_FULL requires coords_ok(ray.pos, ray.pos + ray.dir)
// for every lane of `live`.
template <uint32_t P = XP_FULL>
__device__ __forceinline__ Ray to_object_space(const Ray& ray, const Affine& mi, lmask live) {
  f3 pos = xform_point_sparse<P>(mi, ray.pos);
  f3 point = xform_point_sparse<P>(mi, ray.pos + ray.dir);
  Ray r; r.pos = pos; r.dir = normalize_dev(point - pos, live);
  return r;
}

// Ray.inl:189-226 Moeller-Trumbore with barycentric normal interpolation and back-face cull.
// Returns t (or -1 for "no hit", the m_t of a default Hit) and the un-normalised normal.
__device__ __forceinline__ float hit_triangle(const Ray& ray, const DevTri& tri, f3& nor, lmask need = exec_mask()) {
  f3 pvec = cross(ray.dir, tri.e2);
  float det = dot(tri.e1, pvec);
  if (gabs(det) < kEps) return -1.0f;
  det = cr_rcp(det, need & exec_mask());
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

// Slab test shared by CUBE (Ray.inl:121-127) and MESH (Ray.inl:161-167). The six IEEE divisions share three
// reciprocals: q = a / d as a * RN(1/d) plus two residual corrections (see above), taken by the whole
// wavefront when every lane that matters has 2^-40 <= |a|, |d| < 2^40 for all nine operands (|d| <= 1 for
// a normalised direction, |a| <= |bound| + |origin|); otherwise the wavefront divides with `/`.
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }   // v_min3_f32
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32
__device__ __forceinline__ float div_core(float a, float d, float r) {
  float q = a * r;
  q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
  return __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
}
// Returns the lanes (of `need`) for which Ray.inl:127 / :167 does NOT return "no hit".
__device__ __forceinline__ lmask slab(const Ray& ray, f3 bmin, f3 bmax, float& tNear, float& tFar, lmask need) {
  const f3 a0 = bmin - ray.pos, a1 = bmax - ray.pos;
  const f3 d = ray.dir;
  f3 tMin, tMax;
  const float lo = __builtin_fminf(__builtin_fminf(min3f(__builtin_fabsf(a0.x), __builtin_fabsf(a0.y), __builtin_fabsf(a0.z)),
                                                   min3f(__builtin_fabsf(a1.x), __builtin_fabsf(a1.y), __builtin_fabsf(a1.z))),
                                   min3f(__builtin_fabsf(d.x), __builtin_fabsf(d.y), __builtin_fabsf(d.z)));
  const float hi = __builtin_fmaxf(max3f(__builtin_fabsf(a0.x), __builtin_fabsf(a0.y), __builtin_fabsf(a0.z)),
                                   max3f(__builtin_fabsf(a1.x), __builtin_fabsf(a1.y), __builtin_fabsf(a1.z)));
  const lmask ok = TRC_FCMP(lo, 0x1p-40f, F_OGE) & TRC_FCMP(hi, 0x1p40f, F_OLT);
  if (__builtin_expect((need & ~ok) == 0ull, 1)) {
    const float rx0 = __builtin_amdgcn_rcpf(d.x), ry0 = __builtin_amdgcn_rcpf(d.y), rz0 = __builtin_amdgcn_rcpf(d.z);
    const float rx = __builtin_fmaf(__builtin_fmaf(-d.x, rx0, 1.0f), rx0, rx0);
    const float ry = __builtin_fmaf(__builtin_fmaf(-d.y, ry0, 1.0f), ry0, ry0);
    const float rz = __builtin_fmaf(__builtin_fmaf(-d.z, rz0, 1.0f), rz0, rz0);
    tMin = mk3(div_core(a0.x, d.x, rx), div_core(a0.y, d.y, ry), div_core(a0.z, d.z, rz));
    tMax = mk3(div_core(a1.x, d.x, rx), div_core(a1.y, d.y, ry), div_core(a1.z, d.z, rz));
    // glm::min / glm::max are compare-and-select (func_common.inl:296, :314); here they may be v_min / v_max:
    // every quotient of a needed lane is finite and nonzero (|a|, |d| in range), so no signed-zero or NaN case
    // can tell the two apart. (A ray with a NaN coordinate has an all-NaN direction in every object space --
    // the normalisation spreads it -- so all six of its quotients are NaN and so are both results, either way.)
    tNear = max3f(__builtin_fminf(tMin.x, tMax.x), __builtin_fminf(tMin.y, tMax.y), __builtin_fminf(tMin.z, tMax.z));
    tFar = min3f(__builtin_fmaxf(tMin.x, tMax.x), __builtin_fmaxf(tMin.y, tMax.y), __builtin_fmaxf(tMin.z, tMax.z));
  } else {
    tMin = a0 / d;
    tMax = a1 / d;
    f3 t1 = gmin(tMin, tMax);
    f3 t2 = gmax(tMin, tMax);
    tNear = gmax(gmax(t1.x, t1.y), t1.z);
    tFar = gmin(gmin(t2.x, t2.y), t2.z);
  }
  return need & ~(TRC_FCMP(tNear, tFar, F_OGT) | TRC_FCMP(tFar, 0.0f, F_OLT));     // !(tNear > tFar || tFar < 0)
}

// Branch-free Moeller-Trumbore for the cooperative test: same arithmetic as hit_triangle(), every
// rejection folded into one predicate (the rejected lanes' values are never observed).
__device__ __forceinline__ lmask hit_triangle_all(const Ray& ray, const DevTri& tri, lmask valid, float& t, f3& nor) {
  const f3 pvec = cross(ray.dir, tri.e2);
  const float det = dot(tri.e1, pvec);
  lmask ok = valid & ~TRC_FCMP(__builtin_fabsf(det), kEps, F_OLT);
  const float inv = cr_rcp(det, ok);
  const f3 tvec = ray.pos - tri.v0;
  const float bx = dot(tvec, pvec) * inv;
  ok &= ~(TRC_FCMP(bx, 0.0f, F_OLT) | TRC_FCMP(bx, 1.0f, F_OGT));
  const f3 qvec = cross(tvec, tri.e1);
  const float by = dot(ray.dir, qvec) * inv;
  ok &= ~(TRC_FCMP(by, 0.0f, F_OLT) | TRC_FCMP(bx + by, 1.0f, F_OGT));
  t = dot(tri.e2, qvec) * inv;
  const float bz = 1.0f - bx - by;
  nor = tri.n0 * bz + tri.n1 * bx + tri.n2 * by;
  ok &= ~TRC_FCMP(dot(ray.dir, nor), 0.0f, F_OGT);
  return ok;
}

// Per-wave LDS mailboxes of the cooperative mesh test: one slot per passing ray.
struct alignas(16) MeshSlot {
  float px, py, pz, dx;            // ray (object space)
  float dy, dz;
  unsigned long long key;          // (bits of t) << 32 | face index : u64 min = closest t, lowest face on ties
  float nx, ny, nz; uint32_t pad;  // the winner's un-normalised normal
};
static_assert(sizeof(MeshSlot) == 48, "MeshSlot");
constexpr int kMaxWavesPerBlock = 4;
#ifndef TRC_MESH_SLOTS
#define TRC_MESH_SLOTS 64          // mailboxes per wavefront; with fewer than 64, mesh_closest() takes the passing rays in rounds
#endif
static_assert(TRC_MESH_SLOTS >= 1 && TRC_MESH_SLOTS <= 64, "TRC_MESH_SLOTS");

// MESH, Ray.inl:160-186: brute force over the object's triangles in face order, strict `<`
// (lowest face index wins ties, Ray.inl:174). Wave-convergent; `pass` marks the lanes whose
// ray passed the AABB. Returns the closest positive t (FLT_MAX if none) and its normal.
//
// The few passing rays of a wavefront become wave-wide work: they are compacted into LDS slots,
// and the 64 lanes enumerate (ray, triangle) PAIRS -- a mesh of G < 64 triangles takes 64/G rays per
// pass (the 20-triangle icosahedron: 3 rays on 60 lanes), larger meshes one ray per 64-triangle
// chunk. Every hit does one ds_min_u64 on its ray's key; the winner then leaves its normal in the
// slot and the ray's own lane collects (t, normal) at the end.
__device__ __forceinline__ float mesh_closest(const Ray& r, lmask mask, const DevTri* __restrict__ tris,
                                              const DevMeshExtra& ex, f3& bn, uint32_t& tri_tests) {
  __shared__ MeshSlot s_slots[kMaxWavesPerBlock * TRC_MESH_SLOTS];
  const uint32_t first = ex.tri_first, count = ex.tri_count;
  float best = FLT_MAX;
  bn = splat3(0.0f);
  if (__builtin_expect(mask == 0ull || count == 0u, 0)) return best;
  if (lane_of(mask)) tri_tests += count;
  const unsigned lane = lane_id();
  const unsigned G = count < 64u ? count : 64u;                  // triangles side by side in one pass
  const unsigned R = ex.rays_per_pass;                           // 64 / G rays side by side
  const unsigned n_chunks = (count + 63u) >> 6;
  lmask todo = mask;
  do {
  // this round: the first TRC_MESH_SLOTS passing rays that are still to do (all of them when there are 64 mailboxes)
  lmask round = todo;
  if (TRC_MESH_SLOTS < 64) round = todo & TRC_UCMP((unsigned)__popcll(todo & ((1ull << lane) - 1ull)), (unsigned)TRC_MESH_SLOTS, I_ULT);
  todo &= ~round;
  const bool pass = lane_of(round);
  const unsigned n_pass = __popcll(round);
  // cost model (VALU instructions per wave): pair enumeration ~ ceil(n_pass / R) * n_chunks * 110,
  // ray-per-lane ~ count * 95
  if (__builtin_expect(((n_pass + R - 1u) / R) * n_chunks * 5u <= count * 4u, 1)) {
    MeshSlot* __restrict__ slots = s_slots + (threadIdx.x >> 6) * TRC_MESH_SLOTS;
    const unsigned rank = __popcll(round & ((1ull << lane) - 1ull));      // < n_pass <= TRC_MESH_SLOTS
    const unsigned long long KEY_NONE = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;   // t = FLT_MAX
    if (pass) {
      MeshSlot& m = slots[rank];
      m.px = r.pos.x; m.py = r.pos.y; m.pz = r.pos.z; m.dx = r.dir.x; m.dy = r.dir.y; m.dz = r.dir.z;
      m.key = KEY_NONE;
    }
    const unsigned sub = (lane * ex.group_magic) >> 16;         // lane / G
    const unsigned tl = lane - sub * G;                         // lane % G
    const lmask lane_used = TRC_UCMP(sub, R, I_ULT);
    for (uint32_t base = 0; base < count; base += 64u) {
      const uint32_t ti = base + tl;
      const lmask tvalid = lane_used & TRC_UCMP(ti, count, I_ULT);
      const DevTri tri = tris[first + (lane_of(tvalid) ? ti : 0u)];
      for (unsigned s0 = 0; s0 < n_pass; s0 += R) {
        TRC_MARK("mesh_pass");
        const unsigned slot = s0 + sub;
        const lmask valid = tvalid & TRC_UCMP(slot, n_pass, I_ULT);
        MeshSlot& m = slots[lane_of(valid) ? slot : 0u];
        Ray rr;
        rr.pos = mk3(m.px, m.py, m.pz); rr.dir = mk3(m.dx, m.dy, m.dz);
        float t; f3 n;
        const lmask hit = hit_triangle_all(rr, tri, valid, t, n);
        // Ray.inl:174  thit.m_t > 0 && thit.m_t < hit.m_t (initially FLT_MAX); positive floats order as their bits
        const bool cand = lane_of(hit & TRC_FCMP(t, 0.0f, F_OGT) & TRC_FCMP(t, FLT_MAX, F_OLT));
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | ti;
        if (cand) atomicMin(&m.key, key);
        if (cand && m.key == key) { m.nx = n.x; m.ny = n.y; m.nz = n.z; }
        TRC_MARK("mesh_pass_end");
      }
    }
    if (pass) {
      const MeshSlot& m = slots[rank];
      best = __uint_as_float((unsigned)(m.key >> 32));
      if (best != FLT_MAX) bn = mk3(m.nx, m.ny, m.nz);
    }
  } else if (pass) {
    for (uint32_t i = 0; i < count; ++i) {
      f3 n;
      float t = hit_triangle(r, tris[first + i], n);
      if (t > 0.0f && t < best) { best = t; bn = n; }
    }
  }
  } while (todo != 0ull);
  return best;
}

// ---- closest hit ---------------------------------------------------------------------------------
// Ray::intersectScene, Ray.inl:66-84: closest by WORLD-space squared distance (not by t); the
// surviving t is the object-space one. The reference walks objects in index order with a strict
// `<`, i.e. among equal distances the LOWEST index wins; objects are visited here grouped by
// type, so the same winner is selected by comparing (distance, original index).
struct Closest {
  float mindist; float t; int id; f3 pos, nor;
};

// Ray.inl:93 (t >= 0 gets transformed), :72 (t > 0 competes), :59-64 (hit to world space)
__device__ __forceinline__ void consider(Closest& c, const Ray& ray, float t, uint32_t id,
                                         const Affine& m, f3 opos, f3 wnor) {
  const f3 wpos = xform_point(m, opos);                       // Ray.inl:61
  const f3 d = ray.pos - wpos;
  const float dist = dot(d, d);                               // Ray.inl:73-74
  const bool better = (t > 0.0f) & ((dist < c.mindist) | ((dist == c.mindist) & ((int)id < c.id)));
  if (better) { c.mindist = dist; c.t = t; c.id = (int)id; c.pos = wpos; c.nor = wnor; }
}

// ---- planes and spheres: hold the candidate, transform it once -----------------------------------
// A ray usually has a positive hit on ONE object of a group (one wall of a box, one of a few
// spheres), yet Ray.inl:59-64 + :72-74 (world position, world normal, distance, the running
// comparison) cost as much as the intersection itself. So each lane HOLDS its candidate (t, object
// space point, object number) while the group is walked, and the world-space half runs once per
// group with the held object's matrix gathered per lane. If any lane meets a second candidate
// while holding one, the wave flushes first -- every candidate still goes through consider() in
// object order, so the result is the one the plain loop produces.
// Meshes join the spheres' group: a mesh hit is held like a sphere's (with its un-normalised object-space normal
// `n` in place of normalize(p)), and the group is flushed after the last mesh -- the world-space half of the
// spheres and of the meshes then runs once per trip instead of once per sphere group and once per mesh.
struct Held { lmask any; float t; f3 p; uint32_t k; f3 n; };   // any: the lanes holding a candidate (wave mask)

// `mesh_first` (SPHERE only): held object numbers from here on are meshes (they follow the spheres and the cubes).
template <bool SPHERE>
__device__ __forceinline__ void flush_held(Closest& c, const Ray& ray, const DevXform* __restrict__ xg, Held& h,
                                           uint32_t mesh_first = 0xFFFFFFFFu) {
  if (h.any == 0ull) return;
  if (lane_of(h.any)) {
    const DevXform& x = xg[h.k];                                // per-lane gather (a handful of distinct records)
    const Affine m = x.m;
    f3 wn = x.nor_y;                                            // planes: object +y (Ray.inl:156, :62)
    if (SPHERE) {
      const lmask is_mesh = TRC_UCMP(h.k, mesh_first, I_UGE);
      const f3 sn = normalize_dev(h.p, h.any & ~is_mesh);       // Ray.inl:116 (a sphere's normal)
      const f3 on = lane_of(is_mesh) ? h.n : sn;
      wn = normalize_dev(xform_dir(m, on), h.any);              // Ray.inl:62
    }
    consider(c, ray, h.t, x.id, m, h.p, wn);
  }
  h.any = 0ull;
}
template <bool SPHERE>
__device__ __forceinline__ void hold(Closest& c, const Ray& ray, const DevXform* __restrict__ xg, Held& h,
                                     lmask cand, float t, f3 p, uint32_t k, uint32_t mesh_first = 0xFFFFFFFFu) {
  if (__builtin_expect((cand & h.any) != 0ull, 0)) flush_held<SPHERE>(c, ray, xg, h, mesh_first);
  if (lane_of(cand)) { h.t = t; h.p = p; h.k = k; }
  h.any |= cand;
}

// K planes at once. Ray.inl:142-158: t = -o.y/d.y, hit iff |x|,|z| < 0.5 (strict); the normal is
// always object +y -> DevXform::nor_y. Only t > 0 competes (Ray.inl:72).
template <int K, uint32_t P, int SRC>
__device__ __forceinline__ void planes(Closest& c, const Ray& ray, lmask live, const DevXform* __restrict__ xf,
                                       const DevXform* __restrict__ xg, uint32_t first, Held& h) {
  Ray r[K];
#pragma unroll
  for (int k = 0; k < K; ++k) r[k] = to_object_space<P>(ray, object_mi<SRC>(xf, xg, first + k), live);    // Ray.inl:88
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float tp = -r[k].pos.y / r[k].dir.y;
    const f3 p = r[k].pos + tp * r[k].dir;
    // Ray.inl:148-149  x < 0.5 && x > -0.5 is |x| < 0.5 for every float (NaN fails both forms); |.| is a source modifier
    const lmask in = TRC_FCMP(__builtin_fabsf(p.x), 0.5f, F_OLT) & TRC_FCMP(__builtin_fabsf(p.z), 0.5f, F_OLT);
    hold<false>(c, ray, xg, h, live & in & TRC_FCMP(tp, 0.0f, F_OGT), tp, p, first + k);
  }
}

// K spheres at once. Ray.inl:103-118: radius 0.5, geometric method, normal = normalize(pos),
// then Ray.inl:62 normalize(M * (n, 0)) -- both only for a held candidate (flush_held).
template <int K, uint32_t P, int SRC>
__device__ __forceinline__ void spheres(Closest& c, const Ray& ray, lmask live, const DevXform* __restrict__ xf,
                                        const DevXform* __restrict__ xg, uint32_t first, Held& h, uint32_t mesh_first) {
  Ray r[K];
#pragma unroll
  for (int k = 0; k < K; ++k) r[k] = to_object_space<P>(ray, object_mi<SRC>(xf, xg, first + k), live);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const f3 l = -r[k].pos;
    const float s = dot(l, r[k].dir);
    const float l2 = dot(l, l);
    const float m2 = l2 - s * s;
    const lmask outside = TRC_FCMP(l2, 0.25f, F_OGT);
    const lmask hit = live & ~(TRC_FCMP(s, 0.0f, F_OLT) & outside) & ~TRC_FCMP(m2, 0.25f, F_OGT);
    const float q = cr_sqrt(0.25f - m2, hit);
    const float th = lane_of(outside) ? s - q : s + q;
    const f3 p = r[k].pos + r[k].dir * th;
    hold<true>(c, ray, xg, h, hit & TRC_FCMP(th, 0.0f, F_OGT), th, p, first + k, mesh_first);
  }
}

// objects [lo, hi) of the group, two per trip
template <uint32_t P, int SRC>
__device__ __forceinline__ void planes_range(Closest& c, const Ray& ray, lmask live, const DevXform* __restrict__ xf,
                                             const DevXform* __restrict__ xg, uint32_t lo, uint32_t hi, Held& h) {
  uint32_t i = lo;
  for (; i + 2 <= hi; i += 2) planes<2, P, SRC>(c, ray, live, xf, xg, i, h);
  if (i < hi) planes<1, P, SRC>(c, ray, live, xf, xg, i, h);
}
template <uint32_t P, int SRC>
__device__ __forceinline__ void spheres_range(Closest& c, const Ray& ray, lmask live, const DevXform* __restrict__ xf,
                                              const DevXform* __restrict__ xg, uint32_t lo, uint32_t hi, Held& h, uint32_t mesh_first) {
  uint32_t i = lo;
  for (; i + 2 <= hi; i += 2) spheres<2, P, SRC>(c, ray, live, xf, xg, i, h, mesh_first);
  if (i < hi) spheres<1, P, SRC>(c, ray, live, xf, xg, i, h, mesh_first);
}

// Ray.inl:120-140: unit cube slab test; normal = first face within EPS in the order
// -x,+x,-y,+y,-z,else +z, taken from the per-object table of world normals.
template <uint32_t P, int SRC>
__device__ __forceinline__ void cube(Closest& c, const Ray& ray, lmask live, const DevXform& xf_g, const DevXform& xf_l, const DevCubeExtra& ex_g) {
  const DevXform xf = SRC == XF_LDS ? xf_l : load_uniform(&xf_g);
  const DevCubeExtra ex = load_uniform(&ex_g);
  const Ray r = to_object_space<P>(ray, xf.mi, live);
  float tNear, tFar;
  const bool hit = lane_of(slab(r, splat3(-0.5f), splat3(0.5f), tNear, tFar, live));
  const float th = tNear > 0.0f ? tNear : tFar;
  const f3 p = r.pos + r.dir * th;
  // six wave-uniform candidates: select from the last test to the first
  f3 wn = ex.axis_nor[5];
  if (p.z < -0.5f + kEps) wn = ex.axis_nor[4];
  if (p.y >  0.5f - kEps) wn = ex.axis_nor[3];
  if (p.y < -0.5f + kEps) wn = ex.axis_nor[2];
  if (p.x >  0.5f - kEps) wn = ex.axis_nor[1];
  if (p.x < -0.5f + kEps) wn = ex.axis_nor[0];
  consider(c, ray, hit ? th : -1.0f, xf.id, xf.m, p, wn);
}

// Ray.inl:160-186 (MESH): AABB slab test, then the closest triangle -- from mesh_closest(), or, for a
// big mesh whose brute-force pass ran before this kernel (kernels.hip: k_mesh_brute), from that pass's
// (t, face) key of this ray slot: the winning face is then evaluated once more for its normal
// (same arithmetic, same t).
// `xg`, `h`, `k`, `mesh_first`: the spheres' group of held candidates (the mesh is object number k of that table).
template <int SRC>
__device__ __forceinline__ void mesh(Closest& c, const Ray& ray, lmask live, const DevXform& xf_g,
                                     const DevMeshExtra& ex_g, const DevScene& sc, uint32_t slot, uint32_t& tri_tests,
                                     const DevXform* __restrict__ xg, Held& h, uint32_t k, uint32_t mesh_first) {
  const Affine mi = SRC == XF_LDS ? xg[k].mi : load_uniform(&xf_g.mi);
  const DevMeshExtra ex = load_uniform(&ex_g);
  const Ray r = to_object_space(ray, mi, live);
  float tNear, tFar;
  const lmask pass_m = slab(r, ex.bmin, ex.bmax, tNear, tFar, live);
  const bool pass = lane_of(pass_m);
  f3 bn;
  float tb;
  if (sc.mesh_keys != nullptr && ex.big_slot != kNotBig) {
    tb = FLT_MAX; bn = splat3(0.0f);
    if (pass) {
      tri_tests += ex.tri_count;
      const unsigned long long key = sc.mesh_keys[(size_t)ex.big_slot * sc.key_stride + slot];
      if ((uint32_t)(key >> 32) != 0x7f7fffffu) {
        f3 n;
        const float t = hit_triangle(r, sc.tris[ex.tri_first + (uint32_t)key], n);
        if (t > 0.0f) { tb = t; bn = n; }
      }
    }
  } else {
    tb = mesh_closest(r, pass_m, sc.tris, ex, bn, tri_tests);
  }
  // Ray.inl:171-185: with no triangle hit the reference carries t = FLT_MAX and a zero normal
  // forward; the world-space distance test then rejects it because |o + d*FLT_MAX|^2 overflows
  // to inf for any normalised d -> explicit miss here.
  const lmask hit_m = pass_m & TRC_FCMP(tb, FLT_MAX, F_UNE);
  const f3 p = r.pos + r.dir * tb;
  // tb > 0 for every hit (mesh_closest / the key pass keep positive t only): the candidate competes (Ray.inl:72);
  // its world-space half -- normalize(M * (bn, 0)), M * p, distance -- runs in flush_held()
  hold<true>(c, ray, xg, h, hit_m, tb, p, k, mesh_first);
  if (lane_of(hit_m)) h.n = bn;
}

// Wave-convergent (see the header comment).
// `slot`: this ray's row in the key table of the big-mesh passes (unused while sc.mesh_keys is null).
// `gather`: where the per-lane matrix gathers of the held candidates read the DevXform table from -- a copy
// of sc.xf in LDS when the kernel made one (k_queue), else sc.xf itself. Never null.
// SRC: XF_LDS when `gather` is such a copy in LDS and the object matrices shall be read from it as well (XfSource).
template <int SRC = XF_SGPR>
__device__ __forceinline__ Hit closest_hit(const Ray& ray, bool live_lane, const DevScene& sc, uint32_t& tri_tests, uint32_t slot,
                                           const DevXform* __restrict__ gather) {
  const lmask live = __ballot(live_lane);
  Closest c; c.mindist = FLT_MAX; c.t = -1.0f; c.id = -1; c.pos = splat3(0.0f); c.nor = splat3(0.0f);
  const DevXform* __restrict__ xf = sc.xf;
  const DevXform* __restrict__ xg = gather;
  uint32_t i = 0;
  Held h; h.any = 0ull; h.t = -1.0f; h.p = splat3(0.0f); h.k = 0u; h.n = splat3(0.0f);
  // Within a type the objects are sorted by the zero pattern of their m_matrixi (scene.h): one straight-line
  // loop body per pattern. A wavefront with a ray that fails the vector-side condition of sparse_xform.h takes
  // every object through the full product.
  const bool sparse = (live & ~coords_ok(ray.pos, ray.pos + ray.dir)) == 0ull;
  TRC_MARK("planes");
  planes_range<XP_FULL, SRC>(c, ray, live, xf, xg, 0u, sparse ? sc.plane_end[XP_FULL] : sc.n_plane, h);
  if (sparse) {
    planes_range<XP_DIAG, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_FULL], sc.plane_end[XP_DIAG], h);
    planes_range<XP_BLOCK_X, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_DIAG], sc.plane_end[XP_BLOCK_X], h);
    planes_range<XP_BLOCK_Y, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_BLOCK_X], sc.plane_end[XP_BLOCK_Y], h);
    planes_range<XP_BLOCK_Z, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_BLOCK_Y], sc.n_plane, h);
  }
  flush_held<false>(c, ray, xg, h);
  xf += sc.n_plane; xg += sc.n_plane;
  TRC_MARK("spheres");
  const uint32_t mesh_first = sc.n_sphere + sc.n_cube;          // numbering of the spheres' group: [spheres | cubes | meshes]
  spheres_range<XP_FULL, SRC>(c, ray, live, xf, xg, 0u, sparse ? sc.sphere_full : sc.n_sphere, h, mesh_first);
  if (sparse) spheres_range<XP_DIAG, SRC>(c, ray, live, xf, xg, sc.sphere_full, sc.n_sphere, h, mesh_first);
  xf += sc.n_sphere;
  TRC_MARK("cubes");
  const uint32_t cube_full = sparse ? sc.cube_full : sc.n_cube;
  for (i = 0; i < cube_full; ++i) cube<XP_FULL, SRC>(c, ray, live, xf[i], xg[sc.n_sphere + i], sc.cubes[i]);
  for (; i < sc.n_cube; ++i) cube<XP_DIAG, SRC>(c, ray, live, xf[i], xg[sc.n_sphere + i], sc.cubes[i]);
  xf += sc.n_cube;
  TRC_MARK("meshes");
  for (i = 0; i < sc.n_mesh; ++i) mesh<SRC>(c, ray, live, xf[i], sc.meshes[i], sc, slot, tri_tests, xg, h, mesh_first + i, mesh_first);
  flush_held<true>(c, ray, xg, h, mesh_first);                   // spheres and meshes
  TRC_MARK("closest_done");
  Hit out; out.t = c.t; out.id = c.id; out.pos = c.pos; out.nor = c.nor;
  return out;
}

} // namespace trc
