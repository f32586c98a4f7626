// tracer_amd/csrc/intersect.h -- closest-hit for gfx950: the arithmetic of the reference's
// src/Ray.inl in its exact operation order, over the flat scene records of scene.h.
//
// closest_hit() is WAVE-CONVERGENT: all 64 lanes of a wavefront must call it together (lanes
// without a live path pass live = false). That lets sparse work become wave-wide work: the few (ray, object)
// pairs that can hit a sphere, cube or mesh are compacted one per lane (lane <-> pair instead of lane <-> ray),
// and the few rays that pass a mesh's AABB are spread over the lanes as (ray, triangle) pairs; the reference's
// "closest, lowest index among equals" is kept by a u64 minimum over (distance or t, index) in LDS.
#pragma once
#include "scene.h"
#include "crmath.h"

#pragma clang fp contract(off)

#ifdef TRC_MARKERS
#define TRC_MARK(name) asm volatile("; MARK " name)
#else
#define TRC_MARK(name)
#endif

// Diagnostic builds (-DTRC_SECTION=n, tools/section_clocks.sh): the tri_tests counter sums the shader cycles that the
// wavefronts spend between TRC_SECT_BEGIN(n, ..) and TRC_SECT_END(n, ..) instead of counting triangle tests.
#ifndef TRC_SECTION
#define TRC_SECTION 0
#endif
#define TRC_SECT_BEGIN(n, acc) do { if (TRC_SECTION == (n)) (acc) -= (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#define TRC_SECT_END(n, acc)   do { if (TRC_SECTION == (n)) (acc) += (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
// Checking builds of the (ray, object) pre-test further down (tests/test_gpu_pairs.py): EVERY pair is tested exactly and carries the pre-test's verdict.
//   TRC_PAIRS_CHECK == 1: the tri_tests counter counts the pairs that the pre-test called a certain miss and that hit
//                         their object at t > 0 or passed a mesh's AABB -- must be zero;
//   TRC_PAIRS_CHECK == 2: it counts the pairs that the pre-test called a certain miss (what the product build skips).
#ifndef TRC_PAIRS_CHECK
#define TRC_PAIRS_CHECK 0
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

// A pointer that was read back from the kernel-argument segment or from a record (load_uniform) has lost its address
// space, and the compiler falls back to flat_load / flat_store / flat_atomic: instructions that count on BOTH memory
// counters (vmcnt and lgkmcnt), so that every later wait for an LDS read also waits for them -- and the kernels wait for
// LDS reads all the time. Every such pointer here points to global memory: say so.
template <typename T> using gptr = T __attribute__((address_space(1)))*;
template <typename T> __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)p; }
// a record of whole dwords from global memory (per-lane address)
template <typename T>
__device__ __forceinline__ T load_global(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  const gptr<const unsigned int> q = (gptr<const unsigned int>)p;
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
// this wavefront's mailboxes (one array per workgroup, shared by mesh_closest() and the (ray, object) pairs below)
__device__ __forceinline__ MeshSlot* wave_slots() {
  __shared__ MeshSlot s_slots[kMaxWavesPerBlock * TRC_MESH_SLOTS];
  return s_slots + (threadIdx.x >> 6) * TRC_MESH_SLOTS;
}

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
  const uint32_t first = ex.tri_first, count = ex.tri_count;
  float best = FLT_MAX;
  bn = splat3(0.0f);
  if (__builtin_expect(mask == 0ull || count == 0u, 0)) return best;
  const unsigned lane = lane_id();
  const unsigned G = count < 64u ? count : 64u;                  // triangles side by side in one pass
  const unsigned R = ex.rays_per_pass & 0xffu;                   // 64 / G rays side by side
  const unsigned R_magic = ex.rays_per_pass >> 8;                // n / R == (n * R_magic) >> 16 for n < 128
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
  if (__builtin_expect((((n_pass + R - 1u) * R_magic) >> 16) * n_chunks * 5u <= count * 4u, 1)) {
    MeshSlot* __restrict__ slots = wave_slots();
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
      const DevTri tri = load_global(&tris[first + (lane_of(tvalid) ? ti : 0u)]);
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
      float t = hit_triangle(r, load_global(&tris[first + i]), n);
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

// ---- planes: hold the candidate, transform it once -------------------------------------------------
// A ray usually has a positive hit on ONE plane of the scene (one wall of a box), yet Ray.inl:59-64 + :72-74 (world
// position, distance, the running comparison) cost as much as the intersection itself. So each lane HOLDS its
// candidate (t, object-space point, object number) while the planes are walked, and the world-space half runs once
// with the held object's matrix gathered per lane. If any lane meets a second candidate while holding one, the
// wave flushes first -- every candidate still goes through consider() in object order, so the result is the one
// the plain loop produces.
struct Held { lmask any; float t; f3 p; uint32_t k; };   // any: the lanes holding a candidate (wave mask)

__device__ __forceinline__ void flush_held(Closest& c, const Ray& ray, const DevXform* __restrict__ xg, Held& h) {
  if (h.any == 0ull) return;
  if (lane_of(h.any)) {
    const DevXform& x = xg[h.k];                                // per-lane gather (a handful of distinct records)
    consider(c, ray, h.t, x.id, x.m, h.p, x.nor_y);             // object +y (Ray.inl:156, :62)
  }
  h.any = 0ull;
}
__device__ __forceinline__ void hold(Closest& c, const Ray& ray, const DevXform* __restrict__ xg, Held& h,
                                     lmask cand, float t, f3 p, uint32_t k) {
  if (__builtin_expect((cand & h.any) != 0ull, 0)) flush_held(c, ray, xg, h);
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
    // (one comparison for both. v_max drops a NaN operand; one of p.x, p.z is a NaN without the other only for an infinite tp
    // times a zero direction component -- the other component is then not zero (dir.y is: tp = -pos.y / dir.y), its
    // coordinate infinite, and the test fails either way)
    const lmask in = TRC_FCMP(__builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.z)), 0.5f, F_OLT);
    hold(c, ray, xg, h, live & in & TRC_FCMP(tp, 0.0f, F_OGT), tp, p, first + k);
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

// ---- spheres, cubes, meshes: (ray, object) PAIRS ---------------------------------------------------------------------
// A ray of the Cornell box hits one of these objects on one segment in four, yet walking the object list
// (Ray.inl:70) makes all 64 lanes run every object's test: the two spheres, the light and the icosahedron's AABB cost
// ~900 of k_queue's ~1550 VALU instructions per trip. Instead:
//   1. every ray is tried against every object's world-space BOUNDING SPHERE (bound.h: ~20 instructions; conservative --
//      it only ever says "certain miss" for a pair the exact test rejects, margins derived there);
//   2. the surviving (ray, object) pairs are compacted, one per lane (at most 64 per pass, usually one pass);
//   3. each lane runs the exact test of ITS pair: it fetches the ray from the owning lane (ds_bpermute) and gathers the
//      object's matrices per lane -- the arithmetic of Ray.inl:50-57 and :103-186 operation for operation, with the
//      full matrix product (which the sparse products of sparse_xform.h equal bit for bit);
//   4. the candidates (Ray.inl:59-64 world position / normal, :73 distance) meet in the ray's mailbox: one ds_min_u64 on
//      (distance bits << 32 | object index) selects what the reference's strict `<` in index order selects
//      (Ray.inl:72-75: smallest distance, lowest index among equals); the owning lane merges the winner into its Closest.
// Distances are sums of squares (never negative), so their bit patterns order like their values.
static_assert(TRC_MESH_SLOTS == 64, "the pair list and the merge use one mailbox per lane");

__device__ __forceinline__ unsigned rank_in(lmask m) {          // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ float pull(int addr, float v) {      // v of the lane addr / 4
  return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

// Object k's bounding record (scene.h: DevXformTail), wave-uniform.
template <int SRC>
__device__ __forceinline__ DevXformTail bound_of(const DevXform* __restrict__ xf, const DevXform* __restrict__ xl, uint32_t k) {
#ifndef TRC_BOUND_LDS
#define TRC_BOUND_LDS 1
#endif
  return (TRC_BOUND_LDS && SRC == XF_LDS) ? *reinterpret_cast<const DevXformTail*>(&xl[k].nor_y)
                                          : load_uniform(reinterpret_cast<const DevXformTail*>(&xf[k].nor_y));
}
// The lanes of `live` whose ray MAY hit the object with bounding record b (bound.h).
// l1k = kappa |p|_1 of the ray's origin; unit = the lanes whose direction has |d.d - 1| < 2^-16. Every comparison is false
// for a NaN, so a ray with a NaN coordinate keeps all its pairs.
__device__ __forceinline__ lmask may_hit(const Ray& ray, float l1k, lmask live, lmask unit, const DevXformTail& b) {
  // (fused multiply-adds here: this arithmetic is the pre-test's own, not the reference's -- it only has to be accurate)
  const f3 oc = b.nor_y - ray.pos;
  const float bb = __builtin_fmaf(oc.z, ray.dir.z, __builtin_fmaf(oc.y, ray.dir.y, oc.x * ray.dir.x));
  const float oc2 = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
  // squared distance from c of the nearest point of the HALF-line: that of the line when c lies ahead (bb > 0), |oc|^2
  // when it lies behind. v_max drops a NaN operand, but a NaN direction never gets here (`unit`), and a NaN origin makes
  // oc2 a NaN: both comparisons below are then false.
  const float bp = __builtin_fmaxf(bb, 0.0f);
  const float d2 = __builtin_fmaf(-bp, bp, oc2);
  const float s = l1k + b.bound_a;                                  // l1k = kappa |p|_1
  const lmask miss = TRC_FCMP(d2, s * s, F_OGT) & TRC_FCMP(oc2, b.bound_far2, F_OLT);
  return live & ~(miss & unit);
}

// Pre-test + compaction: appends the surviving pairs of objects k, k + 1, ... to the list in the mailboxes' `pad` words
// (owning lane | object << 8) while a whole object's survivors still fit (64 entries); advances k and n_pairs.
// xf / xl: the DevXform table (global / the gather table) at its first non-plane record.
template <int SRC>
__device__ __forceinline__ void pairs_collect(const Ray& ray, lmask live, uint32_t n_obj,
                                              const DevXform* __restrict__ xf, const DevXform* __restrict__ xl,
                                              uint32_t& k, uint32_t& n_pairs) {
  if (k >= n_obj) return;
  MeshSlot* __restrict__ slots = wave_slots();
  const unsigned lane = lane_id();
  const float l1 = kBoundKappa * (__builtin_fabsf(ray.pos.x) + __builtin_fabsf(ray.pos.y) + __builtin_fabsf(ray.pos.z));
  const float dd = __builtin_fmaf(ray.dir.z, ray.dir.z, __builtin_fmaf(ray.dir.y, ray.dir.y, ray.dir.x * ray.dir.x));
  const lmask unit = TRC_FCMP(__builtin_fabsf(dd - 1.0f), 0x1p-16f, F_OLT);
  // two objects per trip: both records are read before the first is tested
  while (k < n_obj) {
    const bool two = k + 1u < n_obj;
    const DevXformTail b0 = bound_of<SRC>(xf, xl, k);
    const DevXformTail b1 = bound_of<SRC>(xf, xl, two ? k + 1u : k);
    lmask m0 = may_hit(ray, l1, live, unit, b0);
    lmask m1 = may_hit(ray, l1, live, unit, b1);
    uint32_t tag0 = lane, tag1 = lane;                              // entry: owning lane | (checking builds) 64 = "certain miss" | object << 8
    if (TRC_PAIRS_CHECK) { if (!lane_of(m0)) tag0 |= 64u; if (!lane_of(m1)) tag1 |= 64u; m0 = live; m1 = live; }
    const uint32_t c0 = __popcll(m0), c1 = __popcll(m1);
    if (n_pairs + c0 > 64u) break;                                  // the list is full: test what it holds first
    if (lane_of(m0)) slots[n_pairs + rank_in(m0)].pad = tag0 | (k << 8);
    n_pairs += c0;
    ++k;
    if (!two) break;
    if (n_pairs + c1 > 64u) break;
    if (lane_of(m1)) slots[n_pairs + rank_in(m1)].pad = tag1 | (k << 8);
    n_pairs += c1;
    ++k;
  }
}

// What lane j < n_pairs needs to start on pair j: the object and the ray, fetched from the owning lane.
struct PairFetch { lmask valid; uint32_t k; int src4; Ray w; bool flagged; };
__device__ __forceinline__ PairFetch pairs_fetch(const Ray& ray, uint32_t n_pairs) {
  MeshSlot* __restrict__ slots = wave_slots();
  PairFetch f;
  f.valid = TRC_UCMP(lane_id(), n_pairs, I_ULT);
  __builtin_amdgcn_wave_barrier();
  const uint32_t e = lane_of(f.valid) ? slots[lane_id()].pad : 0u;
  f.k = e >> 8;
  f.src4 = (int)((e & 63u) << 2);
  f.flagged = (e & 64u) != 0u;
  f.w.pos = mk3(pull(f.src4, ray.pos.x), pull(f.src4, ray.pos.y), pull(f.src4, ray.pos.z));
  f.w.dir = mk3(pull(f.src4, ray.dir.x), pull(f.src4, ray.dir.y), pull(f.src4, ray.dir.z));
  return f;
}

// One pass: every lane tests the pair it fetched. xg: the gather table at its first non-plane record;
// `slot`: the ray slot of the big-mesh passes (closest_hit()).
__device__ __forceinline__ void pairs_pass(Closest& c, const PairFetch& f, const DevScene& sc,
                                           const DevXform* __restrict__ xg, uint32_t slot, uint32_t& tri_tests) {
  MeshSlot* __restrict__ slots = wave_slots();
  const unsigned lane = lane_id();
  const lmask valid = f.valid;
  const uint32_t k = f.k;
  const int src4 = f.src4;
  const Ray w = f.w;                                                // the pair's ray (world space)
  const DevXform& X = xg[k];                                        // per-lane gather
  const Ray r = to_object_space<XP_FULL>(w, X.mi, valid);           // Ray.inl:88
  const uint32_t mesh_first = sc.n_sphere + sc.n_cube;
  const lmask is_s = valid & TRC_UCMP(k, sc.n_sphere, I_ULT);
  const lmask is_m = valid & TRC_UCMP(k, mesh_first, I_UGE);
  const lmask is_c = valid & ~(is_s | is_m);
  float th = -1.0f;                                                 // the pair's object-space t
  lmask cand = 0ull;                                                // pairs with a hit at t > 0 (Ray.inl:72)
  lmask hull_miss = 0ull;                                           // mesh pairs whose ray passes outside the sphere that holds the vertices
  if ((is_s | is_m) != 0ull) {
    // Ray.inl:103-118: radius 0.5 about the origin, geometric method -- and, in the same instructions, a mesh pair's
    // ray against the sphere that holds the mesh's vertices (centre hull_c: bound.h, mesh_hull)
    f3 l = -r.pos;
    if (lane_of(is_m)) l = mk3(X.hull_cx, X.hull_cy, X.hull_cz) - r.pos;
    const float s = dot(l, r.dir);
    const float l2 = dot(l, l);
    const float m2 = l2 - s * s;
    hull_miss = is_m & TRC_FCMP(m2, __builtin_fmaf(0x1p-18f, l2, X.hull_r2), F_OGT);
    const lmask outside = TRC_FCMP(l2, 0.25f, F_OGT);
    const lmask hit = is_s & ~(TRC_FCMP(s, 0.0f, F_OLT) & outside) & ~TRC_FCMP(m2, 0.25f, F_OGT);
    const float q = cr_sqrt(0.25f - m2, hit);
    th = lane_of(outside) ? s - q : s + q;
    cand = hit;                                                     // (and th > 0: one comparison for all types, below)
  }
  f3 bn = splat3(0.0f);                                             // MESH: the winning triangle's un-normalised normal
  lmask box_pass = 0ull, tri_hit = 0ull;                            // (tri_hit: mesh pairs whose ray hit a triangle)
  if ((is_c | is_m) != 0ull) {
    // Ray.inl:121-127 (CUBE, bounds +-0.5) and :161-167 (MESH, the mesh's AABB): one slab test
    f3 bmin = splat3(-0.5f), bmax = splat3(0.5f);
    if (lane_of(is_m)) { bmin = X.bmin; bmax = X.bmax; }
    float tNear, tFar;
    const lmask pass = slab(r, bmin, bmax, tNear, tFar, is_c | is_m);
    box_pass = pass;
    const float thc = tNear > 0.0f ? tNear : tFar;                  // Ray.inl:129
    if (lane_of(is_c)) th = thc;
    cand |= pass & is_c;
    const lmask pass_m = pass & is_m;
    if (pass_m != 0ull) {
      TRC_SECT_BEGIN(4, tri_tests);
      float tb = FLT_MAX;
      const uint32_t ray_slot = sc.mesh_keys != nullptr ? (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)slot) : 0u;
      for (uint32_t j = 0; j < sc.n_mesh; ++j) {
        const lmask aj = pass_m & TRC_UCMP(k, mesh_first + j, I_EQ);
        if (aj == 0ull) continue;
        const DevMeshExtra ex = load_uniform(&sc.meshes[j]);
        // a ray that passes the AABB counts the mesh's triangle tests (Ray.inl:168-170 runs them all); the tests themselves
        // are only for the rays that can reach a vertex (checking builds test every ray)
        if (TRC_SECTION == 0 && TRC_PAIRS_CHECK == 0 && lane_of(aj)) tri_tests += ex.tri_count;
        const lmask mj = TRC_PAIRS_CHECK ? aj : aj & ~hull_miss;
        if (mj == 0ull) continue;
        if (sc.mesh_keys != nullptr && ex.big_slot != kNotBig) {
          // a big mesh whose brute-force pass ran before this kernel (kernels.hip: k_mesh_brute): the winning face
          // is evaluated once more for its normal (same arithmetic, same t)
          if (lane_of(mj)) {
            const unsigned long long key = as_global(sc.mesh_keys)[(size_t)ex.big_slot * sc.key_stride + ray_slot];
            if ((uint32_t)(key >> 32) != 0x7f7fffffu) {
              f3 n;
              const float t = hit_triangle(r, load_global(&sc.tris[ex.tri_first + (uint32_t)key]), n);
              if (t > 0.0f) { tb = t; bn = n; }
            }
          }
        } else {
          f3 n;
          const float t = mesh_closest(r, mj, sc.tris, ex, n, tri_tests);
          if (lane_of(mj)) { tb = t; bn = n; }
        }
      }
      // Ray.inl:171-185: with no triangle hit the reference carries t = FLT_MAX and a zero normal forward, i.e. the point
      // o + d * FLT_MAX; the world-space distance test rejects it because that point's squared distance overflows to inf
      // (or is a NaN) -- for every d of unit length. The one exception is kept: a direction that overflowed in the
      // re-normalisation of Ray.inl:55 is the ZERO vector (|M d|^2 = inf, 1 / sqrt(inf) = 0: a caller's ray with
      // |d| > 1e19), the point is then the origin itself, its distance 0, and the reference reports the mesh "hit" at
      // t = FLT_MAX with a NaN normal (normalize of the zero vector, below). tb > 0 for every triangle hit.
      const lmask zero_dir = TRC_FCMP(__builtin_fabsf(r.dir.x) + __builtin_fabsf(r.dir.y) + __builtin_fabsf(r.dir.z), 0.0f, F_OEQ);
      if (lane_of(is_m)) th = tb;
      tri_hit = pass_m & TRC_FCMP(tb, FLT_MAX, F_UNE);
      cand |= tri_hit | (pass_m & zero_dir);
      TRC_SECT_END(4, tri_tests);
    }
  }
  cand &= TRC_FCMP(th, 0.0f, F_OGT);                                // Ray.inl:72 (a mesh pair's t is positive or FLT_MAX already)
  if (TRC_PAIRS_CHECK == 1 && lane_of(hull_miss & tri_hit)) ++tri_tests;           // a culled mesh pair that hit a triangle after all
  if (TRC_PAIRS_CHECK == 1 && f.flagged && lane_of(cand | (valid & is_m & box_pass))) {
    ++tri_tests;
#ifdef TRC_PAIRS_CHECK_PRINT
    printf("violation: object %u (id %u) ray p=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) th=%.9g cand=%d boxpass=%d centre=(%.9g %.9g %.9g) a=%.9g far2=%.9g\n", k, X.id,
           w.pos.x, w.pos.y, w.pos.z, w.dir.x, w.dir.y, w.dir.z, th, (int)lane_of(cand), (int)lane_of(box_pass), X.nor_y.x, X.nor_y.y, X.nor_y.z, X.bound_a, X.bound_far2);
#endif
  }
  if (TRC_PAIRS_CHECK == 2 && f.flagged && lane_of(valid)) ++tri_tests;
  __builtin_amdgcn_wave_barrier();
  TRC_SECT_BEGIN(5, tri_tests);
  const unsigned long long KEY_NONE = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;
  slots[lane].key = KEY_NONE;
  __builtin_amdgcn_wave_barrier();
  if (cand != 0ull) {
    // the hit in world space: Ray.inl:59-64, :73-74
    const f3 p = r.pos + r.dir * th;
    const Affine m = X.m;
    f3 on = bn;                                                     // object-space normal
    if ((cand & is_s) != 0ull) { const f3 sn = normalize_dev(p, cand & is_s); if (lane_of(is_s)) on = sn; }   // Ray.inl:116
    if ((cand & is_c) != 0ull) {
      // Ray.inl:132-137: the first face within EPS in the order -x, +x, -y, +y, -z, else +z -- as lane masks, two selects
      // per component
      const lmask x0 = TRC_FCMP(p.x, -0.5f + kEps, F_OLT), x1 = TRC_FCMP(p.x, 0.5f - kEps, F_OGT);
      const lmask y0 = TRC_FCMP(p.y, -0.5f + kEps, F_OLT), y1 = TRC_FCMP(p.y, 0.5f - kEps, F_OGT);
      const lmask z0 = TRC_FCMP(p.z, -0.5f + kEps, F_OLT);
      const lmask onx = x0 | x1, ony = ~onx & (y0 | y1);
      const f3 cn = mk3(lane_of(x0) ? -1.0f : (lane_of(onx) ? 1.0f : 0.0f),
                        lane_of(ony & y0) ? -1.0f : (lane_of(ony) ? 1.0f : 0.0f),
                        lane_of(onx | ony) ? 0.0f : (lane_of(z0) ? -1.0f : 1.0f));
      if (lane_of(is_c)) on = cn;
    }
    const f3 wn = normalize_dev(xform_dir(m, on), cand);            // Ray.inl:62
    const f3 wpos = xform_point(m, p);                              // Ray.inl:61
    const f3 d = w.pos - wpos;
    const float dist = dot(d, d);                                   // Ray.inl:73-74
    // a candidate must beat the initial (FLT_MAX, -1): dist < FLT_MAX (false for a NaN)
    const bool ok = lane_of(cand & TRC_FCMP(dist, FLT_MAX, F_OLT));
    const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | X.id;
    MeshSlot& o = slots[src4 >> 2];
    if (ok) atomicMin(&o.key, key);
    __builtin_amdgcn_wave_barrier();
    if (ok && o.key == key) { o.px = th; o.py = wpos.x; o.pz = wpos.y; o.dx = wpos.z; o.dy = wn.x; o.dz = wn.y; o.nx = wn.z; }
    __builtin_amdgcn_wave_barrier();
    const MeshSlot& mine = slots[lane];
    const unsigned long long kk = mine.key;
    if (kk != KEY_NONE) {
      const float dist_w = __uint_as_float((uint32_t)(kk >> 32));
      const int id_w = (int)(uint32_t)kk;
      if ((dist_w < c.mindist) | ((dist_w == c.mindist) & (id_w < c.id))) {      // Ray.inl:75 over (distance, index)
        c.mindist = dist_w; c.id = id_w; c.t = mine.px;
        c.pos = mk3(mine.py, mine.pz, mine.dx); c.nor = mk3(mine.dy, mine.dz, mine.nx);
      }
    }
  }
  TRC_SECT_END(5, tri_tests);
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
  Held h; h.any = 0ull; h.t = -1.0f; h.p = splat3(0.0f); h.k = 0u;
  // Within a type the objects are sorted by the zero pattern of their m_matrixi (scene.h): one straight-line
  // loop body per pattern. A wavefront with a ray that fails the vector-side condition of sparse_xform.h takes
  // every object through the full product.
  const bool sparse = (live & ~coords_ok(ray.pos, ray.pos + ray.dir)) == 0ull;
  // spheres, cubes, meshes: the pre-test and the fetches of the first pass go ahead of the planes, whose arithmetic
  // then runs while those LDS round trips are under way
  const uint32_t n_obj = sc.n_sphere + sc.n_cube + sc.n_mesh;
  uint32_t pk = 0, n_pairs = 0;
  TRC_SECT_BEGIN(6, tri_tests); TRC_SECT_BEGIN(2, tri_tests);
  pairs_collect<SRC>(ray, live, n_obj, xf + sc.n_plane, xg + sc.n_plane, pk, n_pairs);
  PairFetch pf = pairs_fetch(ray, n_pairs);
  TRC_SECT_END(2, tri_tests);
  TRC_MARK("planes");
  TRC_SECT_BEGIN(1, tri_tests);
  planes_range<XP_FULL, SRC>(c, ray, live, xf, xg, 0u, sparse ? sc.plane_end[XP_FULL] : sc.n_plane, h);
  if (sparse) {
    planes_range<XP_DIAG, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_FULL], sc.plane_end[XP_DIAG], h);
    planes_range<XP_BLOCK_X, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_DIAG], sc.plane_end[XP_BLOCK_X], h);
    planes_range<XP_BLOCK_Y, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_BLOCK_X], sc.plane_end[XP_BLOCK_Y], h);
    planes_range<XP_BLOCK_Z, SRC>(c, ray, live, xf, xg, sc.plane_end[XP_BLOCK_Y], sc.n_plane, h);
  }
  flush_held(c, ray, xg, h);
  TRC_SECT_END(1, tri_tests);
  xf += sc.n_plane; xg += sc.n_plane;
  TRC_MARK("spheres");
  while (n_pairs != 0u) {
    if (TRC_SECTION == 8 && lane_id() == 0) tri_tests += 64u * n_pairs;      // (diagnostic builds: pairs, passes; x64 like the clock sums)
    if (TRC_SECTION == 9 && lane_id() == 0) tri_tests += 64u;
    TRC_SECT_BEGIN(3, tri_tests);
    pairs_pass(c, pf, sc, xg, slot, tri_tests);
    TRC_SECT_END(3, tri_tests);
    n_pairs = 0;
    if (pk == n_obj) break;                                         // (usually: one pass)
    pairs_collect<SRC>(ray, live, n_obj, xf, xg, pk, n_pairs);
    pf = pairs_fetch(ray, n_pairs);
  }
  TRC_MARK("closest_done");
  TRC_SECT_END(6, tri_tests);
  Hit out; out.t = c.t; out.id = c.id; out.pos = c.pos; out.nor = c.nor;
  return out;
}

} // namespace trc
