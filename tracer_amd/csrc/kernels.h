// tracer_amd/csrc/kernels.h -- launch interface between the C ABI (tracer_abi.cpp) and the
// gfx950 kernels (kernels.hip).
#pragma once
#include "scene.h"
#include "pmath.h"
#include "bvh.h"
#include "bound.h"
#include <hip/hip_runtime.h>

namespace trc {

// Everything a kernel needs that is uniform over the launch. Passed by value (kernarg
// segment -> SGPRs).
struct FrameParams {
  uint32_t w, h, npix;
  // The pixels this launch renders: [pix0, pix0 + ntile), whole rows (trc_set_tile; the whole image by
  // default). Kernels number their work 0 .. ntile-1 ("slot") and address the film, the RNG and the
  // camera with pixel = pix0 + slot, so a tile holds exactly the values the full frame has there.
  uint32_t pix0, ntile;
  UDiv div_w, div_npix, div_tile;   // pixel -> (x, y); % npix of raytrace.cu:155; task -> (iteration, slot)
  float fw, fh, rcp_w, rcp_h;   // (float)w, (float)h and their correctly rounded reciprocals (div_by_uniform)
  f3 campos, A, B, C;       // raytrace.cu:192
  f3 nA, nB;                // normalize(A), normalize(B) of raytrace.cu:94, hoisted to the host
  float lensRadius, focalDist;
  DevScene sc;              // packed scene (scene.h)
  uint32_t seed_lo, seed_hi;
  int rng_mode;             // trc_rng_mode
};

inline void set_dims(FrameParams& fp, uint32_t w, uint32_t h) {
  fp.w = w; fp.h = h; fp.npix = w * h;
  fp.div_w = make_udiv(w); fp.div_npix = make_udiv(fp.npix);
  fp.pix0 = 0; fp.ntile = fp.npix; fp.div_tile = fp.div_npix;
  fp.fw = (float)w; fp.fh = (float)h; fp.rcp_w = 1.0f / fp.fw; fp.rcp_h = 1.0f / fp.fh;
}

// Per-path state of the wavefront pipelines: 40 bytes per path in three planar arrays of 16-, 16- and 8-byte elements,
// so that a wavefront moves a path's state with three load and three store instructions, each covering 1 KB / 512 B of
// contiguous memory (round 2 kept ten planar dword arrays: ten instructions each way, and after compaction a wavefront's
// run of survivors starts anywhere, so each of them touched two or three partly written cache lines -- the stores alone
// cost a fifth of a bounce launch). Two of these are ping-ponged by the compacting pipeline.
//   a[i] = (pos.x, pos.y, pos.z, dir.x)    b[i] = (dir.y, dir.z, col.r, col.g)    c[i] = (col.b, bits of the task number)
// task = the path's slot in the sample staging (iteration-in-chunk * tile pixels + pixel); -1 once the path has ended
// (raytrace.cu:119-124: idx_d).
struct PathState {
  float4* a;
  float4* b;
  float2* c;
};

// Event counters, banked: a workgroup adds to bank blockIdx.x % kCounterBanks, each bank in a 256-byte
// block of its own, so the one-atomic-per-wavefront updates of a launch with hundreds of thousands of
// wavefronts go to 256 different addresses instead of queueing on one (measured: ~12 ns per same-address
// atomic, i.e. 3 ms per launch of 262 144 wavefronts). The host sums the banks.
constexpr unsigned kCounterBanks = 256;
struct alignas(256) CounterBank {
  unsigned long long live_segments;
  unsigned long long paths;
  unsigned long long tri_tests;
  unsigned long long wave_cycles;      // persistent kernels: sum of clock64() spans of their wavefronts
  unsigned long long wave_ref_ticks;   // ... and of wall_clock64() (100 MHz) spans
};
struct DeviceCounters { CounterBank bank[kCounterBanks]; };

// Wavefront pipelines (kernels.hip: region_cap): path slots are split into kRegions regions; the live count of
// region p entering bounce d is queue_count[(d * kRegions + p) * kCountStride] (256 bytes apart).
#ifndef TRC_REGIONS
#define TRC_REGIONS 256
#endif
constexpr uint32_t kRegions = TRC_REGIONS;
constexpr uint32_t kCountStride = 64;
// ... followed by the tile counters of k_bounce's wavefronts, grab[(d * kRegions + p) * kCountStride] for bounce d
constexpr uint32_t kLiveCountWords = (kPathDepth + 1) * kRegions * kCountStride;
constexpr uint32_t kQueueCountWords = 2 * kLiveCountWords;
// Layout of the path slots of a launch sequence: region p owns the slots [p * cap, (p + 1) * cap); it starts with `fill`
// paths (tasks p * fill ...), and the rest of `cap` is head room for the compacting bounce, whose wavefronts reserve their
// place in the next queue kBounceBatch slots at a time and leave what they did not use as dead slots (at most 3 batches per
// wavefront). `cap` is an ODD number of 256-slot blocks: the regions advance side by side, and with a power of two in the
// region stride the 256 wavefronts that work at the same offset of their regions met in the same memory channels.
#ifndef TRC_BOUNCE_BPR
#define TRC_BOUNCE_BPR 5           // k_bounce: workgroups per region (kernels.hip): 5 x 256 regions = one per wavefront slot of the chip
#endif
constexpr uint32_t kRegionBlock = 256;      // = the workgroup size of the kernels
constexpr uint32_t kBounceBatch = 64;
#if defined(__HIPCC__) || defined(__CUDACC__)
#define TRC_HOSTDEV __host__ __device__
#else
#define TRC_HOSTDEV
#endif
TRC_HOSTDEV inline uint32_t region_fill(uint32_t n_slots) {
  const uint32_t per = (n_slots + kRegions - 1) / kRegions;
  const uint32_t fill = (per + kRegionBlock - 1) / kRegionBlock * kRegionBlock;
  return fill ? fill : kRegionBlock;
}
TRC_HOSTDEV inline uint32_t region_bpr(uint32_t n_slots) {
  const uint32_t blocks = region_fill(n_slots) / kRegionBlock;
  return blocks < (uint32_t)TRC_BOUNCE_BPR ? blocks : (uint32_t)TRC_BOUNCE_BPR;
}
TRC_HOSTDEV inline uint32_t region_cap(uint32_t n_slots) {
  return ((region_fill(n_slots) / kRegionBlock + 3u * region_bpr(n_slots)) | 1u) * kRegionBlock;
}
// slots the state arrays (and the big-mesh key tables) of a launch sequence over n_slots paths hold
inline size_t wavefront_slots(uint32_t n_slots) { return (size_t)kRegions * region_cap(n_slots); }

// ---- big meshes (scene.h: kBigMeshTris): brute-force pass in front of a closest-hit kernel ------
struct BigMesh {                    // host-side list, in big_slot order
  uint32_t mesh_index, hot_first, tri_first, tri_count;
  // bounding spheres of the faces (bvh.h: mesh_bound_spheres) for the pre-test of k_mesh_brute<true>;
  // use_bound: the faces are small against the mesh (r_max <= radius / 4), so the pre-test rejects nearly every pair
  BvhParams prm; float r_max, origin_dist; uint32_t use_bound;
};
// Dense queue of the rays that passed one big mesh's AABB, in that mesh's object space.
struct MeshRayQueue {
  uint32_t* count;                 // rays queued (device counter)
  uint32_t* slot;                  // ray slot of queue entry j (row of the key table)
  float* ox; float* oy; float* oz; // object-space ray of entry j, planar
  float* dx; float* dy; float* dz;
};
// What the launchers need to run those passes: the scene's big meshes, the queue (reused mesh
// after mesh, bounce after bounce: everything is stream-ordered), the key table
// keys[big_slot * fp.sc.key_stride + slot] (== fp.sc.mesh_keys) and the size of the persistent grid.
struct DevBvh { const BvhNode* nodes; const uint32_t* faces; BvhParams prm; };   // device arrays of one big mesh's hierarchy
struct BigMeshPass {
  const BigMesh* big; uint32_t n_big;
  const DevBvh* bvh;               // per big mesh, or null: exhaustive loop (the reference's behaviour)
  bool plain;                      // exhaustive loop without the bounding-sphere pre-test (TRC_OPT_MESH_ACCEL = 2)
  MeshRayQueue q;
  unsigned long long* keys;
  unsigned grid_blocks;
  DeviceCounters* counters;        // or null: k_mesh_brute adds its wavefronts' clock spans (shader clock of the pass)
};

inline void set_tile(FrameParams& fp, uint32_t row0, uint32_t rows) {
  fp.pix0 = row0 * fp.w; fp.ntile = rows * fp.w; fp.div_tile = make_udiv(fp.ntile);
}

// mode 1 (raytrace.cu:29-62). hit_* may be null; bmp is null when the scene has no big mesh.
void launch_raytrace1(const FrameParams& fp, uint32_t* ldr, int32_t* hit_id, float* hit_t,
                      float* hit_pos, float* hit_nor, DeviceCounters* counters, const BigMeshPass* bmp,
                      hipStream_t stream);

// Ray::intersectScene for a batch of caller rays (device pointers; rays6[i] = {pos, dir}); outputs may be null;
// bmp as for launch_raytrace1 (slot = ray index).
void launch_intersect_rays(const FrameParams& fp, const float* rays6, uint32_t n, int32_t* hit_id, float* hit_t,
                           float* hit_pos, float* hit_nor, DeviceCounters* counters, const BigMeshPass* bmp,
                           hipStream_t stream);

// mode 2, wavefront pipelines: a chunk of `n_iters` iterations (iter_first, +stride, ...) in one launch
// sequence = raygen + kPathDepth bounce launches over ntile * n_iters path slots + k_accumulate.
// compact = false: state stays at its slot, dead paths flagged (pixel plane = -1).
// compact = true : survivors are written densely into `next` behind a wave-ballot prefix sum;
//                  per region (kRegions of them), one live count per bounce.
// cur / next hold wavefront_slots(ntile * n_iters) entries each, stage 3 * ntile * n_iters floats,
// queue_count kQueueCountWords words.
void launch_wavefront_chunk(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                            bool compact, PathState cur, PathState next, uint32_t* queue_count,
                            float* film, float* stage, DeviceCounters* counters, const BigMeshPass* bmp,
                            hipStream_t stream);

// mode 2, fused persistent pipeline: all iterations of the call in one launch.
void launch_fused(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                  float* film, DeviceCounters* counters, hipStream_t stream);

// mode 2, persistent work-queue pipeline: `n_iters` iterations (iter_first, +stride, ...) as
// npix*n_iters independent path tasks pulled from a global counter by `resident_blocks`
// persistent workgroups; per-path radiance goes through `stage` (3*npix*n_iters floats) and is
// added to the film in iteration order by a second kernel.
void launch_queue(const FrameParams& fp, uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters,
                  float* film, float* stage, uint32_t* next_task, unsigned resident_blocks,
                  DeviceCounters* counters, hipStream_t stream);

// 64-bit content digest of a device-resident Object[] in the reference's layout (objects, mesh headers, mesh
// arrays) into *out (device memory); see k_scene_digest.
void launch_scene_digest(const void* scene_d, uint32_t n_objects, unsigned long long* out, hipStream_t stream);

// display transform (raytrace.cu:173) over the whole film.
void launch_tonemap(const float* film, uint32_t npix, float iters, uint32_t* ldr, hipStream_t stream);

} // namespace trc
