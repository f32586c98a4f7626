// tracer_amd/csrc/kernels.hip -- hand-written gfx950 kernels for the per-bounce path-tracing
// loop (replaces reference src/raytrace.cu:29-177). Compiled with -ffp-contract=off.
//
// Three pipelines produce bit-identical films (tracer_abi.h: trc_pipeline):
//   wavefront : per-pixel ray state in HBM, one launch per bounce (reference structure)
//   compacted : same, survivors compacted by wavefront ballot into a dense queue each bounce
//   fused     : one persistent launch; state lives in VGPRs across bounces and iterations
// The film sum of a pixel is always iteration 1, 2, 3, ... in that order (raytrace.cu:170),
// and adding the black (0,0,0) of a missed / depth-capped path is skipped because x + 0 == x.
#include "kernels.h"
#include <cstddef>
#include "intersect.h"
#include "shade.h"
#include "rng.h"
#include "bvh.h"

#pragma clang fp contract(off)

namespace trc {

constexpr int kBlock = 256;
constexpr unsigned kDigestParts = 32;     // workgroups per object of k_scene_digest
#ifndef TRC_GATHER_MAX
#define TRC_GATHER_MAX 16
#endif
constexpr uint32_t kGatherMax = TRC_GATHER_MAX;   // DevXform records k_queue / k_bounce keep in LDS (2.5 KB)
// the records that held candidates gather per lane (intersect.h: flush_held): planes, spheres and -- through the
// spheres' group -- meshes; the table is the whole DevXform array
__host__ __device__ inline uint32_t gather_records(const DevScene& sc) { return sc.n_plane + sc.n_sphere + sc.n_cube + sc.n_mesh; }

static inline unsigned grid_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

// 16- and 8-byte elements of the per-path state (kernels.h: PathState) through global-address-space pointers
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f ldg4(const float4* p, size_t j) { return ((gptr<const v4f>)p)[j]; }
__device__ __forceinline__ v2f ldg2(const float2* p, size_t j) { return ((gptr<const v2f>)p)[j]; }
__device__ __forceinline__ void stg4(float4* p, size_t j, float x, float y, float z, float w) { v4f v = { x, y, z, w }; ((gptr<v4f>)p)[j] = v; }
__device__ __forceinline__ void stg2(float2* p, size_t j, float x, float y) { v2f v = { x, y }; ((gptr<v2f>)p)[j] = v; }

// Wavefront pipelines: the path slots of a launch sequence are split into kRegions contiguous regions (kernels.h:
// region_fill / region_cap). A workgroup belongs to one region; survivors of a bounce are compacted WITHIN their region,
// with one live count per region and bounce, each in a 256-byte block of its own -- so the queue reservations of a
// launch go to 256 addresses instead of one (same measurement as for the counters, kernels.h).
static_assert(kRegionBlock == (uint32_t)kBlock, "region blocks are workgroups");
// Which paths a region starts with: the tasks (path numbers: iteration_in_chunk * ntile + pixel_in_tile) are dealt to the
// regions 64 at a time, in rounds of kRegions tiles, and every round starts at another, scrambled region -- so that every
// region holds the same mix of pixels. (Dealt one contiguous range each, the regions of the headline workload differed by
// 13 % in the time they took: one covers the centre of the image twice, another the ceiling; dealt round robin, or with a
// fixed turn per round, a region of a 512-wide image kept meeting the same columns and rows: 25 %.) Slot `off` of region p
// starts as task region_task(p, off).
static_assert((kRegions & (kRegions - 1u)) == 0u, "kRegions is a power of two");
__device__ __forceinline__ uint32_t region_turn(uint32_t k) { return (k * 0x9E3779B1u) >> 20; }    // (scrambled, any width)
__device__ __forceinline__ uint32_t region_task(uint32_t p, uint32_t off) {
  const uint32_t k = off >> 6;
  return (k * kRegions + ((p + region_turn(k)) & (kRegions - 1u))) * 64u + (off & 63u);
}
__device__ __forceinline__ uint32_t region_initial(uint32_t n_slots, uint32_t p) {
  const uint32_t T = (n_slots + 63u) / 64u;                            // 64-task tiles in all
  const uint32_t K = T / kRegions, R = T % kRegions;                    // full rounds, tiles of the last round
  const uint32_t in_last = (p + region_turn(K)) & (kRegions - 1u);    // this region's place in round K
  const uint32_t mine = K + ((in_last < R) ? 1u : 0u);
  if (mine == 0u) return 0u;
  const bool has_last = R ? (in_last == R - 1u) : (((p + region_turn(K - 1u)) & (kRegions - 1u)) == kRegions - 1u);
  return (has_last && (n_slots & 63u)) ? (mine - 1u) * 64u + (n_slots & 63u) : mine * 64u;
}
// this workgroup's region, its first slot within the region, the region's live count
struct RegionView { uint32_t p, first, n_in; };
__device__ __forceinline__ RegionView region_view(uint32_t n_slots, uint32_t cap, const uint32_t* __restrict__ count_in) {
  const uint32_t bpr = cap / kBlock;
  RegionView v;
  v.p = blockIdx.x / bpr;
  v.first = (blockIdx.x - v.p * bpr) * kBlock;
  v.n_in = count_in ? count_in[v.p * kCountStride] : region_initial(n_slots, v.p);
  return v;
}

// ---- wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void count_add(unsigned long long* dst, unsigned long long lane_value) {
  unsigned long long s = wave_sum_u64(lane_value);
  if (lane_id() == 0 && s) atomicAdd(dst, s);
}

// ---- camera rays ------------------------------------------------------------------------------
// raytrace.cu:40-43 (mode 1: pixel corner, no jitter, pinhole)
__device__ __forceinline__ Ray primary_ray(const FrameParams& fp, uint32_t idx) {
  uint32_t x;
  const uint32_t y = udivmod(idx, fp.div_w, x);
  float u = (float)x / fp.w, v = (float)y / fp.h;
  Ray ray;
  ray.pos = fp.campos + fp.C + (2.0f * u - 1.0f) * fp.A + (2.0f * v - 1.0f) * fp.B;
  ray.dir = normalize(ray.pos - fp.campos);
  return ray;
}

// initBuffersKernel, raytrace.cu:72-96: +-1 pixel jitter and thin-lens depth of field, all
// from the one triple rand[idx] (x, y reused by jitter and lens).
__device__ __forceinline__ Ray camera_sample(const FrameParams& fp, uint32_t idx, uint32_t iter, lmask need = exec_mask()) {
  uint32_t x;
  const uint32_t y = udivmod(idx, fp.div_w, x);
  f3 r = rand3(fp.seed_lo, fp.seed_hi, idx, iter, 0u);
  // raytrace.cu:80-81; x, y < 2^24 and |2r - 1| in {0} u [2^-24, 1]: div_by_uniform's precondition
  float u = div_by_uniform((float)x, fp.fw, fp.rcp_w) + div_by_uniform(2.0f * r.x - 1.0f, fp.fw, fp.rcp_w);
  float v = div_by_uniform((float)y, fp.fh, fp.rcp_h) + div_by_uniform(2.0f * r.y - 1.0f, fp.fh, fp.rcp_h);
  Ray ray;
  ray.pos = fp.campos + fp.C + (2.0f * u - 1.0f) * fp.A + (2.0f * v - 1.0f) * fp.B;
  ray.dir = normalize_dev(ray.pos - fp.campos, need);
  f3 fpt = fp.focalDist * ray.dir + ray.pos;
  f2 disk = rand_point_disk(r.x, r.y, r.z);
  float dkx = fp.lensRadius * disk.x, dky = fp.lensRadius * disk.y;
  ray.pos = ray.pos + (dky * fp.nA + dkx * fp.nB);          // note the x/y swap, raytrace.cu:94
  ray.dir = normalize_dev(fpt - ray.pos, need);
  return ray;
}

// SGPR pressure. A persistent kernel that keeps all of FrameParams (68 dwords of kernel arguments) in SGPRs
// for its whole loop leaves too few for the loop body, and the register allocator parks the overflow in the
// lanes of a VGPR: every later use is a v_readlane -- a VALU instruction in a VALU-bound kernel (k_queue:
// ~100 of them per trip). Fields that a part of the loop needs are instead read again from the
// kernel-argument segment right there: s_load, scalar cache, no VALU. The empty asm makes the address opaque
// so that the optimiser does not hoist the loads back out of the loop.
// Only for kernels whose argument 0 is the FrameParams.
__device__ __forceinline__ const FrameParams* kernarg_params() {
  unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return reinterpret_cast<const FrameParams*>(ka);
}

// ---- one bounce (calcColorKernel body, raytrace.cu:126-159) -------------------------------------
// Wave-convergent: every lane of the wavefront calls it; lanes with live == false run along
// (their result is STEP_IDLE and their ray/col are left untouched).
enum StepResult { STEP_CONTINUE = 0, STEP_BLACK = 1, STEP_LIGHT = 2, STEP_IDLE = 3 };

// `fp_is_arg0`: the calling kernel's argument 0 is this FrameParams (see kernarg_params()).
struct NoHook { __device__ __forceinline__ void operator()(bool) const {} };
// `decided(cont)`: called by every lane (wave-convergent) as soon as the outcome of the trip is known -- the closest hit is
// found and its material read, `cont` says whether this lane's path goes on -- and before the shading arithmetic: the point
// of a trip where the fewest registers are live. k_bounce reserves the survivors' run in the next queue and issues the loads
// of its NEXT trip's ray state there, so that both are under way during the ~300 instructions of the bounce instead of in
// front of an idle wavefront.
// `lds_mats`: the kernel's LDS copy of the material table (indexed like sc.mats) where `mats_in_lds`.
template <class Hook = NoHook>
__device__ __forceinline__ int bounce_step(const FrameParams& fp_in, bool live, uint32_t pixel, uint32_t iter, int depth,
                                           Ray& ray, f3& col, uint32_t& tri_tests, uint32_t slot = 0u,
                                           const DevXform* __restrict__ gather = nullptr, bool gather_given = false,
                                           bool fp_is_arg0 = false, bool xf_in_lds = false, Hook decided = Hook(),
                                           const DevMaterial* lds_mats = nullptr, bool mats_in_lds = false) {
  DevScene sc_again;
  if (fp_is_arg0) sc_again = load_uniform(&kernarg_params()->sc);
  const DevScene& sc = fp_is_arg0 ? sc_again : fp_in.sc;
#ifndef TRC_XF_LDS
#define TRC_XF_LDS 1      // object matrices from the LDS copy of the DevXform table where the kernel made one (intersect.h: XfSource)
#endif
  Hit hit = (TRC_XF_LDS && xf_in_lds) ? closest_hit<XF_LDS>(ray, live, sc, tri_tests, slot, gather)
                                      : closest_hit<XF_SGPR>(ray, live, sc, tri_tests, slot, gather_given ? gather : sc.xf);
  const bool has_hit = live && hit.id >= 0;
  DevMaterial o;
  o.color = splat3(0.0f); o.emit = 0.0f; o.brdf = 0.0f; o.ior = 0.0f; o.mat_type = 0; o.inv_ior = 0.0f;
  if (mats_in_lds) o = lds_mats[has_hit ? hit.id : 0];
  else if (has_hit) o = load_global(&sc.mats[hit.id]);
  decided(has_hit && !(o.emit > 0.0f) && depth != kPathDepth - 1);
  if (!live) return STEP_IDLE;
  if (hit.id < 0) { col = splat3(0.0f); return STEP_BLACK; }               // :130-134
  FrameParams again;
  if (fp_is_arg0) again = load_uniform(kernarg_params());
  const FrameParams& fp = fp_is_arg0 ? again : fp_in;
  const f3 mcolor = o.color;
  const float emit = o.emit;
  if (emit > 0.0f) { col = col * (mcolor * emit); return STEP_LIGHT; }      // :137-141
  if (depth == kPathDepth - 1) { col = splat3(0.0f); return STEP_BLACK; }   // :144-147
  col = col * mcolor;                                                      // :151
  f3 rnd;
  if (fp.rng_mode == 0) {
    // :155 -- unsigned wrap-around of idx + (int)dot, then % size; ray is the PRE-bounce ray
    uint32_t randidx;
    udivmod(pixel + (uint32_t)(int)dot(ray.dir, ray.pos), fp.div_npix, randidx);
    rnd = rand3(fp.seed_lo, fp.seed_hi, randidx, iter, 0u);
#ifdef TRC_DUP_RNG          // cost-measuring build (tools/pmc_dup.sh)
    { uint32_t ri = randidx; asm volatile("" : "+v"(ri)); f3 r2 = rand3(fp.seed_lo, fp.seed_hi, ri, iter, 0u); if (r2.x != rnd.x) rnd = r2; }
#endif
  } else {
    rnd = rand3(fp.seed_lo, fp.seed_hi, pixel, iter, 1u + (uint32_t)depth);
  }
  ray.dir = bounce(o.mat_type, o.ior, o.inv_ior, ray.dir, hit.nor, rnd);              // :156-157
  ray.pos = hit.pos + kEps * ray.dir;                                      // :159
  return STEP_CONTINUE;
}

// ---- mode 1: raytraceKernel, raytrace.cu:29-62 ---------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_raytrace1(FrameParams fp, uint32_t* __restrict__ ldr,
                                                     int32_t* __restrict__ hit_id, float* __restrict__ hit_t,
                                                     float* __restrict__ hit_pos, float* __restrict__ hit_nor,
                                                     DeviceCounters* counters) {
  const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t idx = fp.pix0 + slot;
  const bool valid = slot < fp.ntile;        // no early return: closest_hit is wave-convergent
  Ray ray = primary_ray(fp, idx);
  uint32_t tt = 0;
  Hit hit = closest_hit(ray, valid, fp.sc, tt, slot, fp.sc.xf);
  if (counters) count_add(&counters->bank[blockIdx.x % kCounterBanks].tri_tests, (unsigned long long)tt);
  if (!valid) return;
  const f3 lightDir = mk3((float)0.267261, (float)0.801784, (float)0.534522);
  f3 col;
  if (hit.id < 0) col = ray.dir;
  else {
    const DevMaterial& o = fp.sc.mats[hit.id];
    if (o.emit > 0.0f) col = o.color;
    else {
      col = o.color * o.brdf;
      col = col * gmax(dot(lightDir, hit.nor), 0.0f);
    }
  }
  ldr[idx] = rgb_to_int(col);
  if (hit_id) hit_id[idx] = hit.id;
  if (hit_t) hit_t[idx] = hit.t;
  if (hit_pos) { hit_pos[3 * idx] = hit.pos.x; hit_pos[3 * idx + 1] = hit.pos.y; hit_pos[3 * idx + 2] = hit.pos.z; }
  if (hit_nor) { hit_nor[3 * idx] = hit.nor.x; hit_nor[3 * idx + 1] = hit.nor.y; hit_nor[3 * idx + 2] = hit.nor.z; }
}

// ---- ray batches: Ray::intersectScene (Ray.inl:66-84) for rays the caller supplies -------------------
// What main.cpp:202 does on the CPU for one picking ray, for n rays at once: rays6[i] = {pos.xyz, dir.xyz}
// (the direction as given: intersectScene does not normalise it either). Any output may be null.
__global__ __launch_bounds__(kBlock) void k_intersect_rays(FrameParams fp, const float* __restrict__ rays6, uint32_t n,
                                                          int32_t* __restrict__ hit_id, float* __restrict__ hit_t,
                                                          float* __restrict__ hit_pos, float* __restrict__ hit_nor,
                                                          DeviceCounters* counters) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;                    // no early return: closest_hit is wave-convergent
  Ray ray; ray.pos = splat3(0.0f); ray.dir = splat3(0.0f);
  if (valid) { const float* r = rays6 + 6ull * i; ray.pos = mk3(r[0], r[1], r[2]); ray.dir = mk3(r[3], r[4], r[5]); }
  uint32_t tt = 0;
  const Hit hit = closest_hit(ray, valid, fp.sc, tt, i, fp.sc.xf);
  // a query ray is not a path segment: live_segments and paths stay what was rendered (bench.py and
  // trc_multi_last_stats read them as such); its triangle tests were executed and are counted
  if (counters) count_add(&counters->bank[blockIdx.x % kCounterBanks].tri_tests, (unsigned long long)tt);
  if (!valid) return;
  if (hit_id) hit_id[i] = hit.id;
  if (hit_t) hit_t[i] = hit.t;
  if (hit_pos) { hit_pos[3ull * i] = hit.pos.x; hit_pos[3ull * i + 1] = hit.pos.y; hit_pos[3ull * i + 2] = hit.pos.z; }
  if (hit_nor) { hit_nor[3ull * i] = hit.nor.x; hit_nor[3ull * i + 1] = hit.nor.y; hit_nor[3ull * i + 2] = hit.nor.z; }
}

// ---- big meshes: brute-force closest triangle as a pass of its own ---------------------------------
// Ray.inl:172-177 tests EVERY face of a MESH for every ray that passed its AABB. For a mesh of a
// million triangles that loop is the whole cost of a frame, so the launch-per-bounce pipelines
// (mode 1, wavefront, compacted) take it out of closest_hit():
//   k_mesh_rays   every live ray is brought to the mesh's object space and slab-tested exactly as
//                 mesh() does; the passing rays are appended to a dense queue by wavefront ballot;
//   k_mesh_brute  a workgroup owns 512 queued rays (two per lane, in registers) and a contiguous
//                 segment of the triangle list, which it stages through LDS in tiles of 256 HotTri
//                 records (36 useful bytes each; coalesced copy, then every lane reads the SAME
//                 record: a broadcast, no bank conflicts). (ray block, segment) work items are
//                 handed to a fixed grid round-robin, segment-major, so the workgroups running
//                 at any moment stream the same part of the list out of L2.
//                 The closest hit of a ray is the minimum of (bits of t) << 32 | face over all
//                 segments, joined with one atomicMin per ray and segment: positive floats order
//                 as their bits, and among equal t the lowest face wins, which is the reference's
//                 strict `<` in face order (Ray.inl:174);
//   closest_hit() then looks the key up and evaluates the winning face once more for its normal.
// Per (ray, triangle) the exact Moeller-Trumbore arithmetic runs up to det and u = dot(tvec, pvec);
// 1/det and bary.x = u * (1/det) are NOT formed when the outcome of Ray.inl:196 / :203 is already
// certain from u and det themselves. With c = 0.500005:
//   |det| < EPS                    -> rejected at :196
//   |u - c*det| > |c*det|          -> u/det > 2c = 1.00001 or u/det < -1.5e-8, so fl(u * fl(1/det)) is
//                                     > 1 or < 0 at :203 (bary.x in [0, 1] <=> |u/det - 1/2| <= 1/2)
// c*det and the difference are each rounded once (2^-24 relative) and rounding is monotonic, so
// the strict comparison can only hold with the true values at least that far out -- margins of
// 1e-5 and 1.5e-8 against 1.2e-7 of error in u * fl(1/det) and no underflow to -0. A NaN makes the
// comparisons false, i.e. "not certain". Only when some lane of the wavefront is not certain --
// about one (ray, triangle) pair in 10^5 -- does the wavefront run the full test.
constexpr unsigned long long kKeyNone = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;   // t = FLT_MAX

// SOURCE 0: primary rays of mode 1 (slot = pixel). SOURCE 1: the live paths of a bounce pass
// (slot = position in the path-state arrays; n_in from the queue counter when compacting).
// SOURCE 2: a batch of caller rays, rays6[slot] = {pos, dir} (trc_intersect_rays).
template <int SOURCE>
__global__ __launch_bounds__(kBlock) void k_mesh_rays(FrameParams fp, uint32_t mesh_index, uint32_t n_slots, PathState cur,
                                                     const uint32_t* __restrict__ count_in, MeshRayQueue q,
                                                     unsigned long long* __restrict__ keys, const float* __restrict__ rays6) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool live;
  if (SOURCE == 1) {                 // the slots of a wavefront launch sequence, region by region
    const uint32_t cap = region_cap(n_slots);
    const RegionView rv = region_view(n_slots, cap, count_in);
    if (rv.first >= rv.n_in) return;
    live = rv.first + threadIdx.x < rv.n_in;
    i = rv.p * cap + rv.first + threadIdx.x;
  } else {
    if (blockIdx.x * kBlock >= n_slots) return;
    live = i < n_slots;
  }
  Ray ray; ray.pos = splat3(0.0f); ray.dir = splat3(0.0f);
  if (SOURCE == 0) {
    ray = primary_ray(fp, fp.pix0 + i);
  } else if (SOURCE == 2) {
    if (live) { const float* r = rays6 + 6ull * i; ray.pos = mk3(r[0], r[1], r[2]); ray.dir = mk3(r[3], r[4], r[5]); }
  } else {
    if (live) {
      const float4 A = cur.a[i], B = cur.b[i];
      live = __float_as_int(cur.c[i].y) >= 0;
      if (live) { ray.pos = mk3(A.x, A.y, A.z); ray.dir = mk3(A.w, B.x, B.y); }
    }
  }
  const DevXform& xf = fp.sc.xf[fp.sc.n_plane + fp.sc.n_sphere + fp.sc.n_cube + mesh_index];
  const DevMeshExtra& ex = fp.sc.meshes[mesh_index];
  const lmask live_m = __ballot(live);
  const Ray r = to_object_space(ray, xf.mi, live_m);                     // exactly as mesh() in intersect.h
  float tNear, tFar;
  const bool pass = lane_of(slab(r, ex.bmin, ex.bmax, tNear, tFar, live_m));
  // a ray with a NaN coordinate passes the slab test (Ray.inl:167 is written with `>` and `<`) and then
  // fails every triangle, because each t it produces is NaN; it gets the "no hit" key without being queued.
  const bool finite = (r.pos.x == r.pos.x) & (r.pos.y == r.pos.y) & (r.pos.z == r.pos.z) &
                      (r.dir.x == r.dir.x) & (r.dir.y == r.dir.y) & (r.dir.z == r.dir.z);
  if (pass) keys[(size_t)ex.big_slot * fp.sc.key_stride + i] = kKeyNone;
  const bool enq = pass & finite;
  const unsigned long long mask = __ballot(enq);
  const unsigned n = __popcll(mask);
  if (n == 0) return;
  const unsigned lane = lane_id();
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(q.count, n);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (enq) {
    const uint32_t j = base + __popcll(mask & ((1ull << lane) - 1ull));
    q.slot[j] = i;
    q.ox[j] = r.pos.x; q.oy[j] = r.pos.y; q.oz[j] = r.pos.z;
    q.dx[j] = r.dir.x; q.dy[j] = r.dir.y; q.dz[j] = r.dir.z;
  }
}

constexpr uint32_t kMeshItemWord = 16;   // MeshRayQueue::count[kMeshItemWord]: next work item of k_mesh_brute (64 spare words follow the count)
#ifndef TRC_RAYS_PER_LANE
#define TRC_RAYS_PER_LANE 4          // 2 / 3 / 4 / 6 / 8 rays per lane: 68.6 / 65.4 / 64.0-65.4 / 65.4 / 71.0 ms on configs[4] (107 VGPRs at 4: four waves per SIMD, more independent work in each)
#endif
constexpr int kRaysPerLane = TRC_RAYS_PER_LANE;    // rays a lane of k_mesh_brute carries: every LDS broadcast of a triangle record
                                   // (48 B x 64 lanes through the LDS return path) then feeds that many tests
constexpr uint32_t kBruteRays = kBlock * kRaysPerLane;   // rays per work item

// BOUND: a bounding-sphere pre-test in front of all that (meshes whose faces are small against the mesh, BigMesh::
// use_bound). The LDS tile then holds one float4 per face -- the centre of its bounding sphere -- and a pair goes
// on to the arithmetic above (its HotTri record read from memory) only if the ray's line passes within
//   thr = (r_max + pad(o)) * 1.00001 + 2e-6 * (reach + |mesh centre|)
// of the centre: r_max + pad(o) is as far as the line of a pair that the floating-point test accepts can be
// (bvh.h: mesh_bound_spheres), and the second term covers the rounding of the distance itself. The distance vector is
//   w = (c - o) x d = c x d - m,   m = o x d  (the ray's moment about the object-space origin, formed once per ray),
// evaluated as two fused multiply-adds per component (this arithmetic is the pre-test's own, not the reference's: it
// only has to be accurate): m carries <= 2^-23 |o| of rounding per component, the two fma of a component one rounding
// each on magnitudes <= |c| + |o|, so |w - w_true| <= sqrt(3) * 2^-22 (|c| + |o|) <= 8.3e-7 (reach + |mesh centre|)
// (|o| <= reach + |mesh centre|, |c| <= |mesh centre| + radius <= |mesh centre| + reach); the factor 1.00001 takes the
// three roundings of w.w and |d|^2 = 1 +- 2^-22.
// 10 VALU instructions per pair (round 2: 18, the cross product of c - o unfused; 26 for the exact arithmetic) and one
// LDS read per face instead of three.
template <bool BOUND>
__global__ __launch_bounds__(kBlock) void k_mesh_brute(MeshRayQueue q, const HotTri* __restrict__ hot,
                                                      const DevTri* __restrict__ cold, uint32_t tri_count, uint32_t n_tiles,
                                                      uint32_t tiles_per_seg, uint32_t n_segs,
                                                      unsigned long long* __restrict__ keys,
                                                      const float4* __restrict__ bnd, BvhParams prm, float r_max, float origin_dist,
                                                      DeviceCounters* counters) {
  __shared__ float4 s_tile[kMeshTile * (BOUND ? 1 : 3)];
  const unsigned long long clk0 = clock64(), ref0 = wall_clock64();
  const uint32_t nq = *q.count;
  const uint32_t n_rb = (nq + kBruteRays - 1) / kBruteRays;
  const uint32_t n_items = n_rb * n_segs;
  // Work items are taken from a counter (the word 16 behind the queue's count, zeroed with it), not dealt out by block
  // number: a SIMD issues its oldest wavefront first, so of the 8 workgroups a CU holds the first placed run fastest, and
  // with equal shares the launch waited for the last placed.
  __shared__ uint32_t s_item;
  while (true) {
    __syncthreads();                                       // (everyone has read the previous item number)
    if (threadIdx.x == 0) s_item = atomicAdd(q.count + kMeshItemWord, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= n_items) break;
    const uint32_t seg = item / n_rb;
    const uint32_t rb = item - seg * n_rb;
    // Lanes past the end of the queue carry a zero ray: det == 0, "certain" for every triangle.
    f3 o[kRaysPerLane], d[kRaysPerLane];
    uint32_t slot[kRaysPerLane];
    bool valid[kRaysPerLane];
    unsigned long long best[kRaysPerLane];
#pragma unroll
    for (int r = 0; r < kRaysPerLane; ++r) {
      const uint32_t j = rb * kBruteRays + r * kBlock + threadIdx.x;
      valid[r] = j < nq;
      o[r] = splat3(0.0f); d[r] = splat3(0.0f); slot[r] = 0; best[r] = kKeyNone;
      if (valid[r]) { slot[r] = q.slot[j]; o[r] = mk3(q.ox[j], q.oy[j], q.oz[j]); d[r] = mk3(q.dx[j], q.dy[j], q.dz[j]); }
    }
    float thr2[kRaysPerLane];        // BOUND: squared pre-test distance of this lane's rays; -1 = no ray, every pair certain
    f3 mom[kRaysPerLane];            // BOUND: o x d
#pragma unroll
    for (int r = 0; r < kRaysPerLane; ++r) {
      thr2[r] = -1.0f;
      mom[r] = splat3(0.0f);
      if (BOUND && valid[r]) {
        const f3 oc = o[r] - prm.centre;
        const float reach = sqrtf(dot(oc, oc)) + prm.radius;
        const float pad = kPadScale * prm.max_edge * prm.max_edge * (reach + prm.max_edge);
        const float thr = (r_max + pad) * 1.00001f + 2e-6f * (reach + origin_dist);
        thr2[r] = thr * thr;
        mom[r] = mk3(__builtin_fmaf(o[r].y, d[r].z, -(o[r].z * d[r].y)), __builtin_fmaf(o[r].z, d[r].x, -(o[r].x * d[r].z)),
                     __builtin_fmaf(o[r].x, d[r].y, -(o[r].y * d[r].x)));
      }
    }
    const uint32_t tile0 = seg * tiles_per_seg;
    const uint32_t tile1 = (tile0 + tiles_per_seg < n_tiles) ? tile0 + tiles_per_seg : n_tiles;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(hot);
    // this lane's record of the first tile (records are padded to whole tiles at upload)
    float4 r0, r1, r2;
    if (BOUND) {
      r0 = bnd[(size_t)tile0 * kMeshTile + threadIdx.x]; r1 = r0; r2 = r0;
    } else {
      r0 = src[((size_t)tile0 * kMeshTile + threadIdx.x) * 3 + 0];
      r1 = src[((size_t)tile0 * kMeshTile + threadIdx.x) * 3 + 1];
      r2 = src[((size_t)tile0 * kMeshTile + threadIdx.x) * 3 + 2];
    }
    for (uint32_t tile = tile0; tile < tile1; ++tile) {
      __syncthreads();                                     // everyone is done with the previous tile
      if (BOUND) s_tile[threadIdx.x] = r0;
      else { s_tile[threadIdx.x * 3 + 0] = r0; s_tile[threadIdx.x * 3 + 1] = r1; s_tile[threadIdx.x * 3 + 2] = r2; }
      __syncthreads();
      if (tile + 1 < tile1) {                              // next tile's record is in flight during the tests
        if (BOUND) r0 = bnd[(size_t)(tile + 1) * kMeshTile + threadIdx.x];
        else {
          r0 = src[((size_t)(tile + 1) * kMeshTile + threadIdx.x) * 3 + 0];
          r1 = src[((size_t)(tile + 1) * kMeshTile + threadIdx.x) * 3 + 1];
          r2 = src[((size_t)(tile + 1) * kMeshTile + threadIdx.x) * 3 + 2];
        }
      }
      if (__ballot(valid[0]) == 0ull) continue;            // a wavefront past the end of the queue only helps copying
      // one triangle record against this lane's rays
      auto test = [&](const float4& a, const float4& b, const float4& c, uint32_t k) {
        const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
        unsigned long long open_mask[kRaysPerLane], any_open = 0ull;
#pragma unroll
        for (int r = 0; r < kRaysPerLane; ++r) {
          const f3 pvec = cross(d[r], e2);                 // Ray.inl:193-194
          const float det = dot(e1, pvec);
          const f3 tvec = o[r] - v0;                       // Ray.inl:199
          const float u = dot(tvec, pvec);                 // bary.x = u * (1 / det), Ray.inl:201
          const float cd = det * 0.500005f;
          const float t1 = u - cd;
          // lanes that are NOT certain: !(|det| < EPS) & !(|u - c*det| > |c*det|), as one wave mask
          asm("v_cmp_nlt_f32 vcc, |%1|, %3\n\t"
              "v_cmp_ngt_f32 %0, |%2|, |%4|\n\t"
              "s_and_b64 %0, %0, vcc"
              : "=&s"(open_mask[r]) : "v"(det), "v"(t1), "s"(kEps), "v"(cd) : "vcc");
          any_open |= open_mask[r];
        }
        if (any_open == 0ull) return;
        // Second barycentric coordinate, same idea: v = dot(dir, qvec) is bary.y * det (Ray.inl:206-207) and
        // Ray.inl:208 rejects when bary.y < 0 or bary.x + bary.y > 1, which is certain from
        // |v - c*det| > |c*det| (bary.y outside [0, 1.00001]) or |(u + v) - c*det| > |c*det| (the sum is).
#pragma unroll
        for (int r = 0; r < kRaysPerLane; ++r) {
          if (open_mask[r] == 0ull) continue;
          const f3 pvec = cross(d[r], e2);
          const float det = dot(e1, pvec);
          const f3 tvec = o[r] - v0;
          const float u = dot(tvec, pvec);
          const float v = dot(d[r], cross(tvec, e1));
          const float cd = det * 0.500005f;
          const bool certain = (__builtin_fabsf(v - cd) > __builtin_fabsf(cd)) | (__builtin_fabsf((u + v) - cd) > __builtin_fabsf(cd));
          open_mask[r] &= __builtin_amdgcn_ballot_w64(!certain);
        }
        any_open = 0ull;
#pragma unroll
        for (int r = 0; r < kRaysPerLane; ++r) any_open |= open_mask[r];
        const uint32_t ti = tile * kMeshTile + k;          // face index within the mesh
        if (any_open == 0ull || ti >= tri_count) return;
        // the full test of Ray.inl:189-226 for the lanes that are still not certain
        const DevTri tri = cold[ti];
#pragma unroll
        for (int r = 0; r < kRaysPerLane; ++r) {
          if (open_mask[r] == 0ull) continue;
          float t; f3 n;
          const lmask hit = hit_triangle_all(Ray{o[r], d[r]}, tri, open_mask[r] & __ballot(valid[r]), t, n);
          if (lane_of(hit & TRC_FCMP(t, 0.0f, F_OGT) & TRC_FCMP(t, FLT_MAX, F_OLT))) {          // Ray.inl:174
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | ti;
            best[r] = key < best[r] ? key : best[r];
          }
        }
      };
      if (BOUND) {
        // the pre-test; !(n > thr2) as "unordered or <=", so that a NaN keeps the pair. The lanes (of either ray) that
        // cannot rule face centre c out:
        auto open_of = [&](const float4& c) -> lmask {
          lmask open = 0ull;
#pragma unroll
          for (int r = 0; r < kRaysPerLane; ++r) {
            // w = c x d - m: two fma per component, then w.w with two more
            const float wx = __builtin_fmaf(c.y, d[r].z, -__builtin_fmaf(c.z, d[r].y, mom[r].x));
            const float wy = __builtin_fmaf(c.z, d[r].x, -__builtin_fmaf(c.x, d[r].z, mom[r].y));
            const float wz = __builtin_fmaf(c.x, d[r].y, -__builtin_fmaf(c.y, d[r].x, mom[r].z));
            const float n = __builtin_fmaf(wz, wz, __builtin_fmaf(wy, wy, wx * wx));
            open |= TRC_FCMP(n, thr2[r], 13 /* FCMP_ULE */);
          }
          return open;           // (one comparison for both rays through v_min was measured: no faster, 78.1 against 77.9 ms)
        };
        auto exact = [&](uint32_t k) {
          const size_t rec = ((size_t)tile * kMeshTile + k) * 3;           // wave-uniform address
          test(src[rec], src[rec + 1], src[rec + 2], k);
        };
        // four faces per trip: one decision (and one branch) for all of them -- nearly always "none is open";
        // the next four centres are on their way from LDS meanwhile
        static_assert(kMeshTile % 4 == 0, "four faces per trip");
        float4 c0 = s_tile[0], c1 = s_tile[1], c2 = s_tile[2], c3 = s_tile[3];
        for (uint32_t k = 0; k < kMeshTile; k += 4) {
          const float4 a0 = c0, a1 = c1, a2 = c2, a3 = c3;
          if (k + 4 < kMeshTile) { c0 = s_tile[k + 4]; c1 = s_tile[k + 5]; c2 = s_tile[k + 6]; c3 = s_tile[k + 7]; }
          const lmask o0 = open_of(a0), o1 = open_of(a1), o2 = open_of(a2), o3 = open_of(a3);
          if (__builtin_expect((o0 | o1 | o2 | o3) == 0ull, 1)) continue;
          if (o0) exact(k);
          if (o1) exact(k + 1);
          if (o2) exact(k + 2);
          if (o3) exact(k + 3);
        }
        continue;
      }
      // two register sets take turns, so the next record's LDS read is in flight during a test
      float4 a0 = s_tile[0], b0 = s_tile[1], c0 = s_tile[2], a1, b1, c1;
      for (uint32_t k = 0; k < kMeshTile; k += 2) {
        a1 = s_tile[k * 3 + 3]; b1 = s_tile[k * 3 + 4]; c1 = s_tile[k * 3 + 5];
        test(a0, b0, c0, k);
        if (k + 2 < kMeshTile) { a0 = s_tile[k * 3 + 6]; b0 = s_tile[k * 3 + 7]; c0 = s_tile[k * 3 + 8]; }
        test(a1, b1, c1, k + 1);
      }
    }
#pragma unroll
    for (int r = 0; r < kRaysPerLane; ++r)
      if (best[r] != kKeyNone) atomicMin(&keys[slot[r]], best[r]);
  }
  if (counters && lane_id() == 0) {                  // the shader clock this pass ran at (bench.py: issue utilisation)
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_cycles, clock64() - clk0);
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_ref_ticks, wall_clock64() - ref0);
  }
}

// The same queue and key table, served by the bounding-volume hierarchy of bvh.h instead of the exhaustive
// loop (TRC_OPT_MESH_ACCEL = 1): one ray per lane, a per-lane stack in LDS, the exact hit_triangle() for
// every face of every leaf whose padded box the ray's line meets. See bvh.h for why that visits every face
// the exhaustive loop could accept.
__global__ __launch_bounds__(kBlock) void k_mesh_bvh(MeshRayQueue q, const BvhNode* __restrict__ nodes,
                                                    const uint32_t* __restrict__ faces, const DevTri* __restrict__ tris,
                                                    BvhParams prm, unsigned long long* __restrict__ keys) {
  __shared__ uint32_t s_stack[kBvhMaxDepth][kBlock];
  const uint32_t nq = *q.count;
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nq; j += gridDim.x * kBlock) {
    Ray r;
    r.pos = mk3(q.ox[j], q.oy[j], q.oz[j]); r.dir = mk3(q.dx[j], q.dy[j], q.dz[j]);
    const uint32_t slot = q.slot[j];
    const f3 oc = r.pos - prm.centre;
    const float reach = sqrtf(dot(oc, oc)) + prm.radius;
    const float pad = kPadScale * prm.max_edge * prm.max_edge * (reach + prm.max_edge);
    const float grow = pad + 1e-5f * reach;
    const f3 inv = mk3(1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z);
    unsigned long long best = kKeyNone;
    float best_t = FLT_MAX;
    int sp = 0;
    s_stack[sp++][threadIdx.x] = 0u;
    while (sp > 0) {
      const BvhNode nd = nodes[s_stack[--sp][threadIdx.x]];
      // the ray's LINE against the grown box; every comparison is written so that a NaN keeps the node
      const f3 t0 = (nd.bmin - splat3(grow) - r.pos) * inv, t1 = (nd.bmax + splat3(grow) - r.pos) * inv;
      const float tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
      const float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
      if (tn > tf || tf < -2.0f * pad || tn > best_t + 2.0f * pad) continue;
      if (nd.count == 0u) {
        if (sp + 2 <= kBvhMaxDepth) { s_stack[sp++][threadIdx.x] = nd.first; s_stack[sp++][threadIdx.x] = nd.first + 1u; }
        continue;
      }
      for (uint32_t k = 0; k < nd.count; ++k) {
        const uint32_t face = faces[nd.first + k];
        f3 n;
        const float t = hit_triangle(r, tris[face], n);                // Ray.inl:189-226, exactly
        if (t > 0.0f && t < FLT_MAX) {                                  // Ray.inl:174
          const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | face;
          if (key < best) { best = key; best_t = t; }
        }
      }
    }
    if (best != kKeyNone) atomicMin(&keys[slot], best);
  }
}

// ---- wavefront pipelines -------------------------------------------------------------------------
// The reference's structure (raytrace.cu:211-225): ray state in HBM, one launch per bounce. One launch
// sequence carries a CHUNK of iterations at once -- slot = iteration_in_chunk * ntile + pixel_in_tile, so
// a 512x512 image still gives every launch millions of paths -- and, as in the work-queue pipeline, a
// path stores its sample at stage[slot] when it ends (its radiance at the light, zero otherwise); k_accumulate
// then adds a pixel's samples in iteration order (raytrace.cu:170), which keeps the film bit-exact although the
// iterations of a chunk run side by side. The `pixel` plane of the state holds the path's slot (-1 once it ended).
__global__ __launch_bounds__(kBlock) void k_raygen(FrameParams fp, uint32_t iter_first, uint32_t iter_stride,
                                                  uint32_t n_slots, PathState st,
                                                  uint32_t* __restrict__ queue_count, DeviceCounters* counters) {
  const uint32_t cap = region_cap(n_slots);
  const RegionView rv = region_view(n_slots, cap, nullptr);
  const uint32_t idx = rv.p * cap + rv.first + threadIdx.x;    // slot
  const uint32_t task = region_task(rv.p, rv.first + threadIdx.x);
  if (rv.first == 0 && threadIdx.x == 0) {
    queue_count[rv.p * kCountStride] = rv.n_in;
#pragma unroll
    for (int d = 1; d <= kPathDepth; ++d) queue_count[(d * kRegions + rv.p) * kCountStride] = 0;
#pragma unroll
    for (int d = 0; d < kPathDepth; ++d) queue_count[kLiveCountWords + (d * kRegions + rv.p) * kCountStride] = 0;   // k_bounce's tile counters
  }
  const bool valid = rv.first + threadIdx.x < rv.n_in;
  if (valid) {
    uint32_t s;
    const uint32_t itk = udivmod(task, fp.div_tile, s);
    Ray ray = camera_sample(fp, fp.pix0 + s, iter_first + itk * iter_stride);
    st.a[idx] = make_float4(ray.pos.x, ray.pos.y, ray.pos.z, ray.dir.x);
    st.b[idx] = make_float4(ray.dir.y, ray.dir.z, 1.0f, 1.0f);          // throughput (1, 1, 1): raytrace.cu:99
    st.c[idx] = make_float2(1.0f, __int_as_float((int32_t)task));       // :101
  }
  count_add(&counters->bank[blockIdx.x % kCounterBanks].paths, valid ? 1ull : 0ull);
}

// FIRST: the bounce at depth 0 generates its camera rays itself (k_raygen's body) instead of reading a state that a
// launch of its own would have had to write first -- 40 bytes per path less written and read again, one launch less
// per chunk. The launcher zeroes the region counts with a memset node then, and keeps k_raygen only where a big-mesh
// pass needs the depth-0 rays in memory before this kernel runs.
#ifndef TRC_BOUNCE_WAVES
#define TRC_BOUNCE_WAVES 5         // resident waves per SIMD k_bounce is compiled for (6 = 80 VGPRs spills since the (ray, object) pairs)
#endif
// `bpr` persistent workgroups per region (as many as stay resident: 5 x 256 CUs / 256 regions); their wavefronts take
// the region's queue 64 slots at a time from the region's grab counter, so that a launch ends within one trip of its last
// wavefront (statically divided, the 16384 workgroups of a launch drained for 1/13 of its duration: mean occupancy 4.3 of 5
// wavefronts per SIMD) and the set-up -- the LDS tables, the kernel arguments, the counters -- is paid 1280 times per launch.
// The tile after the next one is reserved while the current one is traced (`decided`), so no trip waits for its reservation.
// All of k_bounce's arguments as one struct (FrameParams first: kernarg_params()), for the same reason as k_queue's: a loop
// that keeps 68 dwords of FrameParams, twenty state pointers and the rest in SGPRs leaves too few for its body, and the
// overflow lives in VGPR lanes -- round 2's k_bounce had 75-90 spilled SGPRs (every reload a v_readlane) and ran the same
// arithmetic a quarter slower than k_queue even where it loads no state at all. Each part of a trip now reads the fields
// it needs from the kernel-argument segment right there (s_load, scalar cache).
struct BounceArgs {
  FrameParams fp;
  uint32_t iter_first, iter_stride, n_slots, bpr;
  int depth;
  PathState cur, next;
  const uint32_t* count_in; uint32_t* count_out;
  uint32_t* grab;                     // this bounce's tile counters, one per region (kCountStride words apart, zero at launch)
  float* stage; DeviceCounters* counters;
};
static_assert(offsetof(BounceArgs, fp) == 0, "kernarg_params() reads the FrameParams at offset 0 of the kernel-argument segment");
__device__ __forceinline__ const BounceArgs* kernarg_bounce() { return reinterpret_cast<const BounceArgs*>(kernarg_params()); }

#ifdef TRC_BOUNCE_TRACE      // development build (tools/bounce_trace.py): when did every workgroup of every bounce start and end?
__device__ unsigned long long g_bounce_trace[kPathDepth * 32768 * 2];
} // namespace trc
extern "C" int trc_debug_bounce_trace(unsigned long long* dst, size_t n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(trc::g_bounce_trace), n * sizeof(unsigned long long));
}
namespace trc {
#endif
template <bool COMPACT, bool LDS_GATHER, bool FIRST>
__global__ __launch_bounds__(kBlock, TRC_BOUNCE_WAVES) void k_bounce(BounceArgs a) {
#ifdef TRC_BOUNCE_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  const FrameParams& fp = a.fp;
  const uint32_t cap = region_cap(a.n_slots);
  // this workgroup's region: workgroups p, p + 256, ... A SIMD issues its oldest wavefront first, so the workgroups placed
  // first (the lowest numbers: one per CU, then a second one per CU ...) run fastest -- 1.24 against 2.2 ms until the end
  // of one bounce with five resident generations -- and a region must own some of every generation to keep pace with the rest
  const uint32_t p = (blockIdx.x + blockIdx.x / kRegions) % kRegions;   // (+ generation: and some on every XCD)
  const uint32_t n_in = (COMPACT && !FIRST) ? a.count_in[p * kCountStride] : region_initial(a.n_slots, p);
  if (n_in == 0) return;                                              // (whole block: its region's queue is empty)
  __shared__ DevXform s_xf[kGatherMax ? kGatherMax : 1];              // as in k_queue: table of the held candidates' gathers
  __shared__ DevMaterial s_mat[kGatherMax ? kGatherMax : 1];          // and the materials (indexed by original object index)
  if (LDS_GATHER) {
    const uint32_t n_dw = gather_records(fp.sc) * (uint32_t)(sizeof(DevXform) / 4);
    for (uint32_t k = threadIdx.x; k < n_dw; k += kBlock) ((uint32_t*)s_xf)[k] = ((const uint32_t*)fp.sc.xf)[k];
    const uint32_t m_dw = gather_records(fp.sc) * (uint32_t)(sizeof(DevMaterial) / 4);
    for (uint32_t k = threadIdx.x; k < m_dw; k += kBlock) ((uint32_t*)s_mat)[k] = ((const uint32_t*)fp.sc.mats)[k];
    __syncthreads();
  }
  const uint32_t region_base = p * cap;
  const uint32_t n_tiles = (n_in + 63u) / 64u;
  const int depth = a.depth;
  const unsigned lane = lane_id();
  uint32_t tt = 0, n_live = 0, n_paths = 0;
  // A trip's ray state as it sits in memory. The loads of trip n + 1 are issued in the middle of trip n (bounce_step's
  // `decided` hook: ~300 instructions ahead of their use) instead of in front of an idle wavefront.
  struct State { int32_t task; float px, py, pz, dx, dy, dz, cr, cg, cb; };
  auto load_state = [&](uint32_t tile_) -> State {
    State L; L.task = -1; L.px = L.py = L.pz = L.dx = L.dy = L.dz = L.cr = L.cg = L.cb = 0.0f;
    if (FIRST || tile_ >= n_tiles) return L;                          // (wave-uniform)
    const PathState cur = load_uniform(&kernarg_bounce()->cur);
    if (tile_ * 64u + lane < n_in) {
      const uint32_t j = region_base + tile_ * 64u + lane;
      const v4f A = ldg4(cur.a, j), B = ldg4(cur.b, j);
      const v2f Cc = ldg2(cur.c, j);
      L.task = __float_as_int(Cc.y);                                  // -1: the path ended (raytrace.cu:119-124) or a dead slot
      L.px = A.x; L.py = A.y; L.pz = A.z; L.dx = A.w; L.dy = B.x; L.dz = B.y; L.cr = B.z; L.cg = B.w; L.cb = Cc.x;
    }
    return L;
  };
  // Reservations -- of the tiles a wavefront takes from its region's queue, and of its place in the next queue, kBounceBatch
  // slots at a time -- are atomics at the memory side, microseconds away. Each is made a whole trip before its answer is
  // needed: a wavefront holds the tile it traces, the next one (whose state it prefetches) and a pending third; the batch
  // it is filling and a pending second one. Their answers sit in lane 0 of `tile_after_v` / `out_next_v`.
  uint32_t tile, tile_next, out_off = 0, out_rem = 0;
  unsigned tile_after_v, out_next_v = 0;
  {
    unsigned v = 0, w = 0;
    if (lane == 0) {
      v = __hip_atomic_fetch_add(as_global(a.grab) + p * kCountStride, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (COMPACT) w = __hip_atomic_fetch_add(as_global(a.count_out) + p * kCountStride, 2u * kBounceBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tile = __builtin_amdgcn_readfirstlane(v);
    tile_next = tile + 1u;
    tile_after_v = v + 2u;
    if (COMPACT) { out_off = __builtin_amdgcn_readfirstlane(w); out_rem = kBounceBatch; out_next_v = w + kBounceBatch; }
  }
  State st = load_state(tile);
  while (tile < n_tiles) {
    const uint32_t first = tile * 64u;
    const uint32_t i = region_base + first + lane;                    // slot
    const bool in_queue = first + lane < n_in;
    int32_t task = -1;
    Ray ray; ray.pos = splat3(0.0f); ray.dir = splat3(0.0f);
    f3 col = splat3(0.0f);
    uint32_t pixel = 0, iter = 0;
    {
      const BounceArgs* ka = kernarg_bounce();
      if (FIRST) {
        task = in_queue ? (int32_t)region_task(p, first + lane) : -1;  // raytrace.cu:101
      } else {
        task = st.task;
        ray.pos = mk3(st.px, st.py, st.pz);
        ray.dir = mk3(st.dx, st.dy, st.dz);
        col = mk3(st.cr, st.cg, st.cb);
      }
      const bool active0 = task >= 0;
      uint32_t s = 0;
      const uint32_t itk = udivmod(active0 ? (uint32_t)task : 0u, load_uniform(&ka->fp.div_tile), s);
      pixel = load_uniform(&ka->fp.pix0) + s;
      iter = load_uniform(&ka->iter_first) + itk * load_uniform(&ka->iter_stride);
      if (FIRST) {
        const FrameParams cf = load_uniform(&ka->fp);                 // camera and image fields: only here
        ray = camera_sample(cf, pixel, iter);                         // initBuffersKernel, raytrace.cu:72-96
        col = splat3(1.0f);                                           // raytrace.cu:99
        n_paths += active0 ? 1u : 0u;
      }
    }
    // Once the outcome of the trip is known (bounce_step: `decided`): who survives (compaction by wavefront ballot: rank
    // within the wave from the ballot mask), and the loads of the next trip's state (past the queue's end: an empty state,
    // no loads). The vector-memory counter is in order, so the ONE wait of a trip -- after the shading, for these loads and
    // for last trip's reservations -- comes before every store of the trip: nothing waits for a store.
    unsigned long long mask = 0ull;
    auto decided = [&](bool cont_) {
      if (COMPACT) mask = __ballot(cont_);
      if (!FIRST) st = load_state(tile_next);
    };
    const bool active = task >= 0;                                    // raytrace.cu:119-124
    const int res = LDS_GATHER ? bounce_step(fp, active, pixel, iter, depth, ray, col, tt, i, s_xf, true, true, true, decided, s_mat, true)
                               : bounce_step(fp, active, pixel, iter, depth, ray, col, tt, i, fp.sc.xf, true, true, false, decided);
    const bool cont = (res == STEP_CONTINUE);
    const uint32_t tile_after = __builtin_amdgcn_readfirstlane(tile_after_v);          // (the wait)
    const uint32_t out_next = COMPACT ? (uint32_t)__builtin_amdgcn_readfirstlane(out_next_v) : 0u;
    if (!FIRST) asm volatile("" :: "v"(st.pz), "v"(st.cg), "v"(st.task));
    const BounceArgs* ka = kernarg_bounce();
    if (tile_after < n_tiles) {                                       // (wave-uniform; past the end nothing is left to reserve)
      if (lane == 0) tile_after_v = __hip_atomic_fetch_add(as_global(load_uniform(&ka->grab)) + p * kCountStride, 1u,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (res == STEP_LIGHT || res == STEP_BLACK) {
      // Every path writes its sample when it ends -- the radiance it carried to the light, or the zero that bounce_step()
      // left in `col` for a miss or the depth cap -- exactly once per slot of the chunk, as k_queue does: the staging
      // buffer needs no clearing (round 2 zero-filled 0.9 GB per 300 iterations here and wrote only at the light).
      gptr<float> dst = as_global(load_uniform(&ka->stage)) + 3ull * (uint32_t)task;
      dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
    }
    if (COMPACT) {
      const unsigned total = __popcll(mask);
      if (cont) {
        const unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t j = region_base + (rank < out_rem ? out_off + rank : out_next + (rank - out_rem));
        const PathState next = load_uniform(&ka->next);
        stg4(next.a, j, ray.pos.x, ray.pos.y, ray.pos.z, ray.dir.x);
        stg4(next.b, j, ray.dir.y, ray.dir.z, col.x, col.y);
        stg2(next.c, j, col.z, __int_as_float(task));
      }
      if (total >= out_rem) {                                         // batch full: go on in the next one, reserve another
        out_off = out_next + (total - out_rem);
        out_rem = kBounceBatch - (total - out_rem);                   // (total <= 64 = kBounceBatch: at least one slot is left)
        if (lane == 0) out_next_v = __hip_atomic_fetch_add(as_global(load_uniform(&ka->count_out)) + p * kCountStride, kBounceBatch,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        out_off += total; out_rem -= total;
      }
    } else if (active) {
      const PathState cur = load_uniform(&ka->cur);
      if (cont) {
        stg4(cur.a, i, ray.pos.x, ray.pos.y, ray.pos.z, ray.dir.x);
        stg4(cur.b, i, ray.dir.y, ray.dir.z, col.x, col.y);
        stg2(cur.c, i, col.z, __int_as_float(task));
      } else {
        stg2(cur.c, i, 0.0f, __int_as_float(-1));                     // raytrace.cu:132,139
      }
    }
    n_live += active ? 1u : 0u;
    tile = tile_next; tile_next = tile_after;
  }
  if (COMPACT) {
    // what this wavefront reserved in the next queue and did not fill: dead slots (no path: task -1)
    const uint32_t out_next = (uint32_t)__builtin_amdgcn_readfirstlane(out_next_v);
    const PathState next = load_uniform(&kernarg_bounce()->next);
    if (lane < out_rem) stg2(next.c, region_base + out_off + lane, 0.0f, __int_as_float(-1));
    static_assert(kBounceBatch == 64u, "one lane per slot of a batch");
    stg2(next.c, region_base + out_next + lane, 0.0f, __int_as_float(-1));
  }
  DeviceCounters* counters = load_uniform(&kernarg_bounce()->counters);
  if (FIRST) count_add(&counters->bank[blockIdx.x % kCounterBanks].paths, (unsigned long long)n_paths);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].live_segments, (unsigned long long)n_live);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].tri_tests, (unsigned long long)tt);
#ifdef TRC_BOUNCE_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 32768u) {
    g_bounce_trace[((size_t)depth * 32768u + blockIdx.x) * 2u] = trace_t0;
    g_bounce_trace[((size_t)depth * 32768u + blockIdx.x) * 2u + 1u] = wall_clock64();
  }
#endif
}

// ---- fused persistent pipeline ------------------------------------------------------------------
// One lane owns one pixel for the whole call. Its ray, throughput and film sum stay in VGPRs;
// a lane whose path ended starts that pixel's next iteration at once (path regeneration), so
// every trip of the loop is one closest-hit evaluation for (nearly) all 64 lanes.
__global__ __launch_bounds__(kBlock, 4) void k_fused(FrameParams fp, uint32_t iter_first, uint32_t iter_stride,
                                                 uint32_t n_iters, float* __restrict__ film,
                                                 DeviceCounters* counters) {
  const uint32_t pixel = fp.pix0 + blockIdx.x * kBlock + threadIdx.x;
  const bool valid = blockIdx.x * kBlock + threadIdx.x < fp.ntile;
  f3 acc = splat3(0.0f);
  if (valid) acc = mk3(film[3 * pixel], film[3 * pixel + 1], film[3 * pixel + 2]);
  uint32_t k = 0;                 // iterations of this call already accumulated
  uint32_t iter = iter_first;
  int depth = 0;
  bool need_new = true;
  bool done = !valid || n_iters == 0;
  Ray ray; ray.pos = splat3(0.0f); ray.dir = splat3(0.0f);
  f3 col = splat3(0.0f);
  uint32_t live = 0, tt = 0;
  const unsigned long long clk0 = clock64(), ref0 = wall_clock64();

  while (true) {
    if (!done && need_new) {
      iter = iter_first + k * iter_stride;
      if (iter == 1) acc = splat3(0.0f);                              // raytrace.cu:103-104
      const FrameParams cf = load_uniform(kernarg_params());          // camera fields: see kernarg_params()
      ray = camera_sample(cf, pixel, iter);
      col = splat3(1.0f);
      depth = 0;
      need_new = false;
    }
    if (__ballot(!done) == 0ull) break;
    const int res = bounce_step(fp, !done, pixel, iter, depth, ray, col, tt, 0u, nullptr, false, true);
    if (res == STEP_CONTINUE) {
      ++live; ++depth;
    } else if (res != STEP_IDLE) {
      ++live;
      if (res == STEP_LIGHT) acc = acc + col;                         // film += col, raytrace.cu:170
      ++k;
      need_new = true;
      done = (k == n_iters);
    }
  }
  if (valid) { film[3 * pixel] = acc.x; film[3 * pixel + 1] = acc.y; film[3 * pixel + 2] = acc.z; }
  count_add(&counters->bank[blockIdx.x % kCounterBanks].live_segments, (unsigned long long)live);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].tri_tests, (unsigned long long)tt);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].paths, valid ? (unsigned long long)n_iters : 0ull);
  if (lane_id() == 0) {
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_cycles, clock64() - clk0);
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_ref_ticks, wall_clock64() - ref0);
  }
}


// ---- persistent work-queue pipeline ------------------------------------------------------------
// The unit of scheduling is ONE PATH = (pixel, iteration). A fixed grid of persistent wavefronts
// (as many as stay resident) pulls task ids from a global counter, kGrab at a time per wave, and
// a lane whose path ended takes the next task at once. Every trip of the loop is therefore a
// closest-hit for all 64 lanes of every resident wave until the queue runs dry -- no idle
// lanes inside a wave (paths of different length) and no tail across waves (pixels of
// different cost), whatever the image size.
// The reference sums a pixel's samples in iteration order (raytrace.cu:170). To keep that sum
// bit-exact although samples now finish in any order on any lane, a path that reaches the
// light stores its radiance -- every other path a zero -- into a staging buffer stage[iteration][pixel]
// (12 B per path; 0.94 GB for 512x512x300, nothing against 288 GB of HBM) and
// k_accumulate adds the staged samples per pixel in iteration order afterwards.
#ifndef TRC_GRAB
#define TRC_GRAB 256          // measured on the headline workload: 128 -> 8.10 ms, 256 -> 8.08, 512 -> 8.16, 1024 -> 8.35 (tail of the last grabs)
#endif
constexpr uint32_t kGrab = TRC_GRAB;   // task ids a wavefront reserves at a time (fewer when the whole call is small: launch_queue)
// One camera ray parked in a wavefront's LDS ring (k_queue).
struct alignas(16) RaySlot { float px, py, pz, dx; float dy, dz; uint32_t pixel, itk; };
static_assert(sizeof(RaySlot) == 32, "RaySlot");
constexpr uint32_t kRing = 128;

#ifndef TRC_QUEUE_WAVES
#define TRC_QUEUE_WAVES 5          // resident waves per SIMD the kernel is compiled for
#endif

// All of k_queue's arguments as one struct (FrameParams first: kernarg_params()), so that the loop can read any
// of them again from the kernel-argument segment where it needs it.
struct QueueArgs {
  FrameParams fp;
  uint32_t iter_first, iter_stride, total, grab;   // total = tile pixels x iterations = number of path tasks
  float* stage; uint32_t* next_task; DeviceCounters* counters;
};
static_assert(offsetof(QueueArgs, fp) == 0, "kernarg_params() reads the FrameParams at offset 0 of the kernel-argument segment");
__device__ __forceinline__ const QueueArgs* kernarg_queue() { return reinterpret_cast<const QueueArgs*>(kernarg_params()); }

template <bool LDS_GATHER>   // the scene's plane + sphere records fit s_xf (the launcher picks the instantiation)
__global__ __launch_bounds__(kBlock, TRC_QUEUE_WAVES) void k_queue(QueueArgs qa) {
  const FrameParams& fp = qa.fp;
  __shared__ RaySlot s_ring[kMaxWavesPerBlock * kRing];
  RaySlot* __restrict__ ring = s_ring + (threadIdx.x >> 6) * kRing;
  // The plane / sphere records that held candidates gather per lane (intersect.h: flush_held), copied to LDS once.
  __shared__ DevXform s_xf[kGatherMax ? kGatherMax : 1];
  __shared__ DevMaterial s_mat[kGatherMax ? kGatherMax : 1];          // and the materials (indexed by original object index)
  const uint32_t n_gather = gather_records(fp.sc);
  if (LDS_GATHER) {
    const uint32_t n_dw = n_gather * (uint32_t)(sizeof(DevXform) / 4);
    for (uint32_t i = threadIdx.x; i < n_dw; i += kBlock) ((uint32_t*)s_xf)[i] = ((const uint32_t*)fp.sc.xf)[i];
    const uint32_t m_dw = n_gather * (uint32_t)(sizeof(DevMaterial) / 4);
    for (uint32_t i = threadIdx.x; i < m_dw; i += kBlock) ((uint32_t*)s_mat)[i] = ((const uint32_t*)fp.sc.mats)[i];
    __syncthreads();
  }
#ifdef TRC_LDS_PAD          // occupancy experiments: extra LDS so that fewer workgroups fit a CU
  __shared__ uint32_t s_pad[TRC_LDS_PAD];
  if (qa.total == 0xFFFFFFFFu) s_pad[threadIdx.x] = 1u;
#endif
  const unsigned lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t pool_next = 0, pool_end = 0;      // wave-uniform: task ids this wave has reserved
  uint32_t ring_head = 0, ring_count = 0;    // wave-uniform: camera rays waiting in this wave's LDS ring
  bool exhausted = false;                    // wave-uniform: the global queue is empty
  bool have = false;                         // this lane carries a live path
  uint32_t pixel = 0, itk = 0, iter = 0;
  int depth = 0;
  Ray ray; ray.pos = splat3(0.0f); ray.dir = splat3(0.0f);
  f3 col = splat3(1.0f);
  uint32_t live = 0, tt = 0, paths = 0;
  const unsigned long long clk0 = clock64(), ref0 = wall_clock64();

  while (true) {
    const unsigned long long need = __ballot(!have);
    if (need != 0ull) {
      const uint32_t n_need = __popcll(need);
      // Camera rays are generated 64 at a time by the WHOLE wavefront into the ring (ring_count < 64
      // here, so at most 127 of its kRing = 128 slots are ever in use); finished lanes then just pick
      // one up. Generating only for the lanes that happen to be idle ran this code at ~27 % lane use.
      while (ring_count < n_need && !exhausted) {
        const QueueArgs* ka = kernarg_queue();
        if (__builtin_expect(pool_next == pool_end, 0)) {
          const uint32_t grab = load_uniform(&ka->grab), total = load_uniform(&ka->total);
          uint32_t base = 0;
          if (lane == 0) base = __hip_atomic_fetch_add(as_global(load_uniform(&ka->next_task)), grab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (base >= total) { exhausted = true; break; }
          pool_next = base;
          pool_end = (total - base < grab) ? total : base + grab;
        }
        const uint32_t avail = pool_end - pool_next;
        const uint32_t take = avail < 64u ? avail : 64u;
        const bool gen = lane < take;
        TRC_MARK("regen");
        // the camera and image fields are needed only here, once per 64 new paths (see kernarg_params())
        const FrameParams cf = load_uniform(&ka->fp);
        uint32_t g_pixel;
        const uint32_t g_itk = udivmod(gen ? pool_next + lane : 0u, cf.div_tile, g_pixel);
        g_pixel += cf.pix0;
        const Ray g = camera_sample(cf, g_pixel, load_uniform(&ka->iter_first) + g_itk * load_uniform(&ka->iter_stride),
                                    TRC_UCMP(lane, take, I_ULT));
        if (gen) {
          RaySlot& o = ring[(ring_head + ring_count + lane) & (kRing - 1u)];
          o.px = g.pos.x; o.py = g.pos.y; o.pz = g.pos.z; o.dx = g.dir.x; o.dy = g.dir.y; o.dz = g.dir.z;
          o.pixel = g_pixel; o.itk = g_itk;
        }
        TRC_MARK("regen_end");
        pool_next += take;
        ring_count += take;
      }
      const uint32_t rank = __popcll(need & lt_mask);
      if (!have && rank < ring_count) {
        const RaySlot& o = ring[(ring_head + rank) & (kRing - 1u)];
        ray.pos = mk3(o.px, o.py, o.pz); ray.dir = mk3(o.dx, o.dy, o.dz);
        pixel = o.pixel; itk = o.itk;
        const QueueArgs* ka = kernarg_queue();
        iter = load_uniform(&ka->iter_first) + itk * load_uniform(&ka->iter_stride);
        col = splat3(1.0f);
        depth = 0;
        have = true;
        ++paths;
      }
      const uint32_t taken = n_need < ring_count ? n_need : ring_count;
      ring_head = (ring_head + taken) & (kRing - 1u);
      ring_count -= taken;
    }
    if (__ballot(have) == 0ull) break;
    const int res = LDS_GATHER ? bounce_step(fp, have, pixel, iter, depth, ray, col, tt, 0u, s_xf, true, true, true, NoHook(), s_mat, true)
                               : bounce_step(fp, have, pixel, iter, depth, ray, col, tt, 0u, fp.sc.xf, true, true);
    TRC_MARK("bounce_done");
    if (res == STEP_CONTINUE) {
      ++live; ++depth;
    } else if (res != STEP_IDLE) {
      ++live;
      // Every path writes its sample when it ends -- the radiance it carried to the light, or the zero that
      // bounce_step() left in `col` for a miss or the depth cap -- so the staging buffer needs no clearing
      // (some lane of a wavefront reaches the light on nine trips out of ten: the stores were issued anyway).
      const QueueArgs* ka = kernarg_queue();
      gptr<float> dst = as_global(load_uniform(&ka->stage)) + 3ull * ((unsigned long long)itk * load_uniform(&ka->fp.ntile) + (pixel - load_uniform(&ka->fp.pix0)));
      dst[0] = col.x; dst[1] = col.y; dst[2] = col.z;
      have = false;
    }
  }
  DeviceCounters* counters = load_uniform(&kernarg_queue()->counters);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].live_segments, (unsigned long long)live);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].tri_tests, (unsigned long long)tt);
  count_add(&counters->bank[blockIdx.x % kCounterBanks].paths, (unsigned long long)paths);
  if (lane == 0) {
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_cycles, clock64() - clk0);
    atomicAdd(&counters->bank[blockIdx.x % kCounterBanks].wave_ref_ticks, wall_clock64() - ref0);
  }
}

// film[pixel] += stage[0][pixel] + stage[1][pixel] + ... strictly in that order (raytrace.cu:170);
// the film restarts from zero when the first staged iteration is number 1 (raytrace.cu:103-104).
__global__ __launch_bounds__(kBlock) void k_accumulate(float* __restrict__ film, const float* __restrict__ stage,
                                                      uint32_t npix, uint32_t n_iters, int restart) {
  const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;   // one lane per film FLOAT: fully coalesced
  if (idx >= 3u * npix) return;
  float f = restart ? 0.0f : film[idx];
  const float* s = stage + idx;
  const size_t pitch = 3ull * npix;
  // the additions are a serial chain; the loads are not: 16 in flight per lane
  uint32_t k = 0;
  for (; k + 16 <= n_iters; k += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(&s[(k + j) * pitch]);
#pragma unroll
    for (int j = 0; j < 16; ++j) f = f + v[j];
  }
  for (; k < n_iters; ++k) f = f + s[k * pitch];
  film[idx] = f;
}

// ---- display transform, raytrace.cu:173 -----------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tonemap(const float* __restrict__ film, uint32_t npix, float iters,
                                                   uint32_t* __restrict__ ldr) {
  uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= npix) return;
  ldr[idx] = tonemap(mk3(film[3 * idx], film[3 * idx + 1], film[3 * idx + 2]), iters);
}

// ---- content digest of a caller-owned device scene (legacy launchers, tracer_abi.hip) ----------------
// The reference's pathtrace / raytrace1 receive `scene_d`, the device copy of Object[] that Renderer::initScene
// built (Renderer.cpp:112-151), every frame, and Renderer::updateScene edits it in place (:221-225). The shim
// behind those signatures keeps a flattened copy of the scene; this kernel tells it whether that copy is still
// good without reading anything back: a 64-bit sum over EVERY dword the scene consists of -- the Object records,
// the Mesh headers they point to and, for a MESH, its vertex / normal / face arrays -- each mixed with its
// position, so a moved matrix, a swapped material, a re-pointed or re-filled mesh array all change it.
// grid = (objects, kDigestParts); a MESH's arrays are spread over the kDigestParts workgroups of its object.
__device__ __forceinline__ unsigned long long digest_mix(uint32_t word, uint32_t index, uint32_t tag, uint32_t obj) {
  unsigned long long z = ((unsigned long long)word | ((unsigned long long)index << 32)) ^
                         (0x9E3779B97F4A7C15ull * (tag + 1u) + 0xD1B54A32D192ED03ull * (obj + 1u));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_scene_digest(const RefObject* __restrict__ objs, unsigned long long* __restrict__ out) {
  const uint32_t o = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  const uint32_t t = part * kBlock + threadIdx.x, stride = nparts * kBlock;
  unsigned long long h = 0ull;
  const uint32_t* ow = reinterpret_cast<const uint32_t*>(objs + o);
  if (part == 0 && threadIdx.x < sizeof(RefObject) / 4) h += digest_mix(ow[threadIdx.x], threadIdx.x, 1u, o);
  const RefMesh* m = objs[o].mesh;
  if (m != nullptr) {
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(m);
    if (part == 0 && threadIdx.x < sizeof(RefMesh) / 4) h += digest_mix(mw[threadIdx.x], threadIdx.x, 2u, o);
    if (m->type == REF_MESH) {
      const uint32_t nv = 3u * m->numVerts, nn = 3u * m->numNorms, nf = 6u * m->numFaces;
      const uint32_t* v = reinterpret_cast<const uint32_t*>(m->verts);
      const uint32_t* n = reinterpret_cast<const uint32_t*>(m->norms);
      const uint32_t* f = reinterpret_cast<const uint32_t*>(m->faces);
      if (v) for (uint32_t i = t; i < nv; i += stride) h += digest_mix(v[i], i, 3u, o);
      if (n) for (uint32_t i = t; i < nn; i += stride) h += digest_mix(n[i], i, 4u, o);
      if (f) for (uint32_t i = t; i < nf; i += stride) h += digest_mix(f[i], i, 5u, o);
    }
  }
  h = wave_sum_u64(h);
  if (lane_id() == 0 && h) atomicAdd(out, h);
}

// ---- launchers ------------------------------------------------------------------------------------

// One brute-force pass per big mesh over the rays of the kernel that follows (SOURCE as k_mesh_rays).
template <int SOURCE>
static void launch_big_mesh_passes(const FrameParams& fp, uint32_t n_slots, PathState cur, const uint32_t* count_in,
                                   const BigMeshPass& bmp, hipStream_t stream, const float* rays6 = nullptr) {
  for (uint32_t b = 0; b < bmp.n_big; ++b) {
    const BigMesh& m = bmp.big[b];
    const uint32_t n_tiles = (m.tri_count + kMeshTile - 1) / kMeshTile;
    // segments of >= 16 tiles (the 256 rays of a work item are loaded once per segment), at most 64 of them
#ifndef TRC_MESH_SEGS
#define TRC_MESH_SEGS 256        // (64 -> 128 -> 256 segments: 70.4 -> 70.1 -> 69.7 ms on configs[4]: the launch ends within one item of its last workgroup)
#endif
    uint32_t tiles_per_seg = (n_tiles + (TRC_MESH_SEGS - 1u)) / (uint32_t)TRC_MESH_SEGS;
    if (tiles_per_seg < 16u) tiles_per_seg = 16u;
    const uint32_t n_segs = (n_tiles + tiles_per_seg - 1) / tiles_per_seg;
    (void)hipMemsetAsync(bmp.q.count, 0, sizeof(uint32_t) * (kMeshItemWord + 1u), stream);   // count and k_mesh_brute's item counter
    hipLaunchKernelGGL(k_mesh_rays<SOURCE>, dim3(SOURCE == 1 ? kRegions * (region_cap(n_slots) / kBlock) : grid_for(n_slots)), dim3(kBlock), 0, stream,
                       fp, m.mesh_index, n_slots, cur, count_in, bmp.q, bmp.keys, rays6);
    if (bmp.bvh && bmp.bvh[b].nodes) {
      hipLaunchKernelGGL(k_mesh_bvh, dim3(bmp.grid_blocks), dim3(kBlock), 0, stream,
                         bmp.q, bmp.bvh[b].nodes, bmp.bvh[b].faces, fp.sc.tris + m.tri_first, bmp.bvh[b].prm,
                         bmp.keys + (size_t)b * fp.sc.key_stride);
      continue;
    }
    const bool bound = m.use_bound && !bmp.plain;
    hipLaunchKernelGGL((bound ? k_mesh_brute<true> : k_mesh_brute<false>), dim3(bmp.grid_blocks), dim3(kBlock), 0, stream,
                       bmp.q, fp.sc.hot + m.hot_first, fp.sc.tris + m.tri_first, m.tri_count, n_tiles,
                       tiles_per_seg, n_segs, bmp.keys + (size_t)b * fp.sc.key_stride,
                       reinterpret_cast<const float4*>(fp.sc.tri_bound) + m.hot_first, m.prm, m.r_max, m.origin_dist,
                       bmp.counters);
  }
}

void launch_raytrace1(const FrameParams& fp, uint32_t* ldr, int32_t* hit_id, float* hit_t,
                      float* hit_pos, float* hit_nor, DeviceCounters* counters, const BigMeshPass* bmp,
                      hipStream_t stream) {
  if (bmp) launch_big_mesh_passes<0>(fp, fp.ntile, PathState{}, nullptr, *bmp, stream);
  hipLaunchKernelGGL(k_raytrace1, dim3(grid_for(fp.ntile)), dim3(kBlock), 0, stream,
                     fp, ldr, hit_id, hit_t, hit_pos, hit_nor, counters);
}

void launch_intersect_rays(const FrameParams& fp, const float* rays6, uint32_t n, int32_t* hit_id, float* hit_t,
                           float* hit_pos, float* hit_nor, DeviceCounters* counters, const BigMeshPass* bmp,
                           hipStream_t stream) {
  if (n == 0) return;
  if (bmp) launch_big_mesh_passes<2>(fp, n, PathState{}, nullptr, *bmp, stream, rays6);
  hipLaunchKernelGGL(k_intersect_rays, dim3(grid_for(n)), dim3(kBlock), 0, stream,
                     fp, rays6, n, hit_id, hit_t, hit_pos, hit_nor, counters);
}

void launch_wavefront_chunk(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                            bool compact, PathState cur, PathState next, uint32_t* queue_count,
                            float* film, float* stage, DeviceCounters* counters, const BigMeshPass* bmp,
                            hipStream_t stream) {
  const uint32_t n_slots = fp.ntile * n_iters;
  const dim3 grid(kRegions * (region_cap(n_slots) / kBlock)), block(kBlock);
  // (no clearing of `stage`: k_bounce writes the sample of every path of the chunk when the path ends)
  // the depth-0 bounce generates the camera rays itself unless a big-mesh pass has to see them in memory first
  const bool fold_raygen = (bmp == nullptr);
  if (fold_raygen) (void)hipMemsetAsync(queue_count, 0, sizeof(uint32_t) * kQueueCountWords, stream);
  else hipLaunchKernelGGL(k_raygen, grid, block, 0, stream, fp, iter_first, iter_stride, n_slots, cur, queue_count, counters);
  const bool lds = gather_records(fp.sc) <= kGatherMax;
  // k_bounce: at most TRC_BOUNCE_BPR workgroups per region, their wavefronts taking the region's queue tile by tile
  const uint32_t bpr = region_bpr(n_slots);
  const dim3 bgrid(kRegions * bpr);
  for (int d = 0; d < kPathDepth; ++d) {
    if (bmp) launch_big_mesh_passes<1>(fp, n_slots, cur, compact ? queue_count + d * kRegions * kCountStride : nullptr, *bmp, stream);
    const bool first = fold_raygen && d == 0;
    const uint32_t* cin = compact ? queue_count + d * kRegions * kCountStride : nullptr;
    uint32_t* cout = compact ? queue_count + (d + 1) * kRegions * kCountStride : nullptr;
    const PathState out = compact ? next : cur;
    BounceArgs ba;
    ba.fp = fp; ba.iter_first = iter_first; ba.iter_stride = iter_stride; ba.n_slots = n_slots; ba.bpr = bpr; ba.depth = d;
    ba.cur = cur; ba.next = out; ba.count_in = cin; ba.count_out = cout; ba.grab = queue_count + kLiveCountWords + d * kRegions * kCountStride; ba.stage = stage; ba.counters = counters;
#define TRC_LAUNCH_BOUNCE(C, L, F) hipLaunchKernelGGL((k_bounce<C, L, F>), bgrid, block, 0, stream, ba)
    if (compact) { if (lds) { if (first) TRC_LAUNCH_BOUNCE(true, true, true); else TRC_LAUNCH_BOUNCE(true, true, false); }
                   else     { if (first) TRC_LAUNCH_BOUNCE(true, false, true); else TRC_LAUNCH_BOUNCE(true, false, false); } }
    else         { if (lds) { if (first) TRC_LAUNCH_BOUNCE(false, true, true); else TRC_LAUNCH_BOUNCE(false, true, false); }
                   else     { if (first) TRC_LAUNCH_BOUNCE(false, false, true); else TRC_LAUNCH_BOUNCE(false, false, false); } }
#undef TRC_LAUNCH_BOUNCE
    if (compact) { PathState t = cur; cur = next; next = t; }
  }
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(3u * fp.ntile)), dim3(kBlock), 0, stream,
                     film + 3ull * fp.pix0, (const float*)stage, fp.ntile, n_iters, iter_first == 1u ? 1 : 0);
}

void launch_fused(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                  float* film, DeviceCounters* counters, hipStream_t stream) {
  hipLaunchKernelGGL(k_fused, dim3(grid_for(fp.ntile)), dim3(kBlock), 0, stream,
                     fp, iter_first, iter_stride, n_iters, film, counters);
}

void launch_queue(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                  float* film, float* stage, uint32_t* next_task, unsigned resident_blocks,
                  DeviceCounters* counters, hipStream_t stream) {
  const size_t paths = (size_t)fp.ntile * n_iters;
  (void)hipMemsetAsync(next_task, 0, sizeof(uint32_t), stream);       // (k_queue writes every sample of `stage`)
  unsigned blocks = (unsigned)((paths + kBlock - 1) / kBlock);
#ifndef TRC_QUEUE_BLOCKS_PER_CU
#define TRC_QUEUE_BLOCKS_PER_CU TRC_QUEUE_WAVES
#endif
  resident_blocks = resident_blocks / 4u * TRC_QUEUE_BLOCKS_PER_CU;     // CUs x waves per SIMD (a block = 1 wave per SIMD)
  if (blocks > resident_blocks) blocks = resident_blocks;
  const bool lds_gather = gather_records(fp.sc) <= kGatherMax;
  // a small call (an interactive frame: one iteration) is spread over all resident wavefronts: about four
  // reservations per wavefront, 64 .. kGrab task ids each
  uint32_t grab = (uint32_t)(paths / ((size_t)blocks * 4u * 4u)) / 64u * 64u;
  grab = grab < 64u ? 64u : (grab > kGrab ? kGrab : grab);
  QueueArgs qa;
  qa.fp = fp; qa.iter_first = iter_first; qa.iter_stride = iter_stride; qa.total = (uint32_t)paths; qa.grab = grab;
  qa.stage = stage; qa.next_task = next_task; qa.counters = counters;
  hipLaunchKernelGGL((lds_gather ? k_queue<true> : k_queue<false>), dim3(blocks), dim3(kBlock), 0, stream, qa);
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(3u * fp.ntile)), dim3(kBlock), 0, stream,
                     film + 3ull * fp.pix0, (const float*)stage, fp.ntile, n_iters, iter_first == 1u ? 1 : 0);
}

void launch_scene_digest(const void* scene_d, uint32_t n_objects, unsigned long long* out, hipStream_t stream) {
  (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
  hipLaunchKernelGGL(k_scene_digest, dim3(n_objects, kDigestParts), dim3(kBlock), 0, stream,
                     reinterpret_cast<const RefObject*>(scene_d), out);
}

void launch_tonemap(const float* film, uint32_t npix, float iters, uint32_t* ldr, hipStream_t stream) {
  hipLaunchKernelGGL(k_tonemap, dim3(grid_for(npix)), dim3(kBlock), 0, stream, film, npix, iters, ldr);
}

} // namespace trc
