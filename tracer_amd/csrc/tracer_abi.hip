// tracer_amd/csrc/tracer_abi.hip -- implementation of include/tracer_abi.h: context, scene
// flattening, launches, headless read-back. Host code only; kernels are in kernels.hip.
//
// Replaces reference src/Renderer.cpp:94-151 (buffer ownership, scene deep copy), :173-206
// (PBO map/unmap) and src/raytrace.cu:189-238 (launchers).
#include "../../include/tracer_abi.h"
#include "kernels.h"

#include <link.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#pragma clang fp contract(off)

using namespace trc;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

// One HIP runtime per process. PyTorch wheels bundle a libamdhip64.so of their own with the system
// library's SONAME; a process that has mapped BOTH (this library loaded first, torch imported afterwards) crashes inside
// libhsa-runtime64 as soon as both copies touch the GPU, and a torch stream or tensor pointer means nothing to the
// other copy. Where a context is made or a caller's device memory is bound, the mapped objects are therefore listed
// and two runtimes are an error, not a crash later (TRC_ALLOW_TWO_HIP_RUNTIMES=1 turns the check off).
int two_hip_runtimes(std::string& which) {
  const char* allow = getenv("TRC_ALLOW_TWO_HIP_RUNTIMES");
  if (allow && allow[0] == '1') return 0;
  std::vector<std::string> found;
  dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) -> int {
    const char* name = info->dlpi_name ? info->dlpi_name : "";
    const char* base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "libamdhip64.so", 14) == 0) {
      auto* v = static_cast<std::vector<std::string>*>(data);
      bool seen = false;
      for (const std::string& s : *v) seen = seen || s == name;
      if (!seen) v->push_back(name);
    }
    return 0;
  }, &found);
  if (found.size() < 2) return 0;
  which = found[0];
  for (size_t i = 1; i < found.size(); ++i) which += " and " + found[i];
  return (int)found.size();
}
#define ONE_HIP_RUNTIME(where)                                                                                              \
  do {                                                                                                                      \
    std::string which_;                                                                                                     \
    if (two_hip_runtimes(which_))                                                                                           \
      return fail(TRC_ERR_INVALID, std::string(where) + ": two HIP runtimes are mapped into this process (" + which_ +     \
                  "); load libtracer_amd.so after importing torch (tracer_amd.api.load_library() does) -- see INTEGRATION.md, " \
                  "One HIP runtime per process");                                                                           \
  } while (0)

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      return fail(TRC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    }                                                                                      \
  } while (0)

template <typename T> void free_dev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

} // namespace

// ---- static scratch of trc_pathtrace_static (include/tracer_abi.h) ---------------------------------------
// The work-queue pipeline's scratch -- the task counter and the per-path sample staging that trc_pathtrace
// keeps in hipMalloc'ed buffers -- as a file-scope device array, for callers that cannot allocate. Every GPU
// the library runs on holds its own copy (at its own address) until the process ends. In host code the
// array's NAME is only a handle for hipGetSymbolAddress: the address a kernel may use is the one that call
// returns on the device that is current, resolved once per device in trc_create and kept in g_scratch_ptr.
constexpr size_t kScratchHeader = 256;                         // [0, 4): the work queue's next-task counter
constexpr size_t kScratchStageBytes = size_t(64) << 20;        // sample staging: 12 bytes per path of a chunk
static __device__ __align__(256) unsigned char scratch[kScratchHeader + kScratchStageBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};                         // per device: where `scratch` lives there
// Per device: recorded behind the last launch of every trc_pathtrace_static call, waited for in front of the next one's
// first launch. All contexts of a GPU share that GPU's array; with this event two calls on different streams cannot
// overlap on the GPU even when the host never synchronises between them (a GPU-side dependency: no host wait).
hipEvent_t g_scratch_event[kMaxDevices] = {};

// `device` must be current. Resolves the device's copy of `scratch` on first use.
int resolve_scratch(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return fail(TRC_ERR_UNSUPPORTED, "device index beyond the static-scratch table");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(TRC_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    HIP_TRY(hipEventCreateWithFlags(&g_scratch_event[device], hipEventDisableTiming));
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return TRC_OK;
}
} // namespace

struct trc_ctx {
  int device = 0;
  uint32_t w = 0, h = 0, npix = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;
  bool timed = false;

  int rng_mode = TRC_RNG_REFERENCE;
  int pipeline = TRC_PIPE_AUTO;
  int stage_mb = 2048;
  int record_hits = 1;
  int mesh_accel = 0;                      // TRC_OPT_MESH_ACCEL
  int skip_tonemap = 0;                    // TRC_OPT_SKIP_TONEMAP
  uint32_t tile_row0 = 0, tile_rows = 0;   // trc_set_tile; rows == 0: the whole image

  // scene: one device allocation [xforms | cube extras | mesh extras | materials | triangles]
  void* scene_d = nullptr;
  size_t scene_cap = 0;
  DevScene sc = {};
  uint32_t nobj = 0;
  std::vector<DevObject> objs_h;     // flattened objects, caller order (for trc_update_object)
  std::vector<DevTri> tris_h;

  // outputs
  float* film_own = nullptr;   // 3*npix
  float* film = nullptr;       // film_own or a bound external buffer
  uint32_t* ldr = nullptr;     // npix
  int32_t* hit_id = nullptr; float* hit_t = nullptr; float* hit_pos = nullptr; float* hit_nor = nullptr;

  // wavefront state (allocated on first use)
  float* state_mem[2] = { nullptr, nullptr };
  size_t state_slots = 0;              // path slots the state arrays hold (a chunk of iterations x tile pixels)
  PathState state[2];
  uint32_t* queue_count = nullptr;

  // work-queue pipeline
  float* stage = nullptr; size_t stage_floats = 0;
  uint32_t* next_task = nullptr;
  unsigned resident_blocks = 0;

  // big meshes: brute-force pass of the launch-per-bounce pipelines (kernels.hip: k_mesh_brute)
  std::vector<BigMesh> big;
  unsigned long long* mesh_keys = nullptr; size_t mesh_keys_cap = 0;
  size_t mesh_slots = 0;               // ray slots per big mesh the key table and the queue hold
  float* mesh_queue_mem = nullptr;
  MeshRayQueue mesh_queue = {};
  unsigned brute_blocks = 0;
  std::vector<DevBvh> bvh;             // per big mesh, built on first use with TRC_OPT_MESH_ACCEL = 1

  // trc_intersect_rays: device staging of a host ray batch [rays 6n | pos 3n | nor 3n | t n | id n]
  float* batch_mem = nullptr; size_t batch_cap = 0;

  unsigned char* scratch_d = nullptr;  // this device's copy of the static `scratch` array (trc_pathtrace_static)

  DeviceCounters* counters = nullptr;
};

namespace {

int ensure_wavefront_state(trc_ctx* c, bool need_second, size_t slots) {
  if (slots > c->state_slots) { free_dev(c->state_mem[0]); free_dev(c->state_mem[1]); c->state_slots = slots; }
  const size_t n = c->state_slots;
  for (int b = 0; b < (need_second ? 2 : 1); ++b) {
    if (c->state_mem[b]) continue;
    HIP_TRY(hipMalloc((void**)&c->state_mem[b], sizeof(float) * 10ull * n));
    float* p = c->state_mem[b];
    PathState& s = c->state[b];
    s.a = (float4*)p; s.b = (float4*)(p + 4ull * n); s.c = (float2*)(p + 8ull * n);      // 16 + 16 + 8 bytes per slot
  }
  if (!c->queue_count) HIP_TRY(hipMalloc((void**)&c->queue_count, sizeof(uint32_t) * kQueueCountWords));
  return TRC_OK;
}

int ensure_hit_buffers(trc_ctx* c) {
  if (c->hit_id) return TRC_OK;
  HIP_TRY(hipMalloc((void**)&c->hit_id, sizeof(int32_t) * c->npix));
  HIP_TRY(hipMalloc((void**)&c->hit_t, sizeof(float) * c->npix));
  HIP_TRY(hipMalloc((void**)&c->hit_pos, sizeof(float) * 3ull * c->npix));
  HIP_TRY(hipMalloc((void**)&c->hit_nor, sizeof(float) * 3ull * c->npix));
  return TRC_OK;
}

void free_bvh(trc_ctx* c) {
  for (DevBvh& b : c->bvh) { (void)hipFree((void*)b.nodes); (void)hipFree((void*)b.faces); }
  c->bvh.clear();
}

// Build (host, median split) and upload the hierarchy of every big mesh; meshes live in object space, so
// trc_update_object leaves them valid and only trc_upload_scene discards them.
int ensure_bvh(trc_ctx* c) {
  if (c->mesh_accel != 1 || c->bvh.size() == c->big.size()) return TRC_OK;
  free_bvh(c);
  for (const BigMesh& m : c->big) {
    std::vector<BvhNode> nodes; std::vector<uint32_t> faces; BvhParams prm;
    build_bvh(c->tris_h.data() + m.tri_first, m.tri_count, nodes, faces, prm);
    DevBvh d; d.prm = prm; d.nodes = nullptr; d.faces = nullptr;
    HIP_TRY(hipMalloc((void**)&d.nodes, nodes.size() * sizeof(BvhNode)));
    c->bvh.push_back(d);
    HIP_TRY(hipMalloc((void**)&c->bvh.back().faces, faces.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy((void*)c->bvh.back().nodes, nodes.data(), nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy((void*)c->bvh.back().faces, faces.data(), faces.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  return TRC_OK;
}

// Buffers of the big-mesh brute-force passes: key table (one row of npix keys per big mesh) and
// the ray queue (npix entries: slot + 6 floats, and the counter).
int ensure_big_mesh_buffers(trc_ctx* c, size_t slots) {
  if (c->big.empty()) return TRC_OK;
  if (slots > c->mesh_slots) { free_dev(c->mesh_queue_mem); c->mesh_slots = slots; }
  const size_t n = c->mesh_slots;
  const size_t need = c->big.size() * n;
  if (need > c->mesh_keys_cap) {
    free_dev(c->mesh_keys); c->mesh_keys_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->mesh_keys, sizeof(unsigned long long) * need));
    c->mesh_keys_cap = need;
  }
  if (!c->mesh_queue_mem) {
    HIP_TRY(hipMalloc((void**)&c->mesh_queue_mem, sizeof(float) * (7ull * n + 64)));
    float* p = c->mesh_queue_mem;
    MeshRayQueue& q = c->mesh_queue;
    q.slot = (uint32_t*)p;
    q.ox = p + n; q.oy = p + 2ull * n; q.oz = p + 3ull * n;
    q.dx = p + 4ull * n; q.dy = p + 5ull * n; q.dz = p + 6ull * n;
    q.count = (uint32_t*)(p + 7ull * n);
  }
  { int rc = ensure_bvh(c); if (rc) return rc; }
  if (!c->brute_blocks) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->brute_blocks = (unsigned)prop.multiProcessorCount * 8u;      // 8 workgroups of 4 wavefronts per CU
  }
  return TRC_OK;
}

BigMeshPass big_mesh_pass(const trc_ctx* c) {
  BigMeshPass b;
  b.big = c->big.data(); b.n_big = (uint32_t)c->big.size();
  b.bvh = (c->mesh_accel == 1 && c->bvh.size() == c->big.size() && !c->bvh.empty()) ? c->bvh.data() : nullptr;
  b.plain = c->mesh_accel == 2;
  b.q = c->mesh_queue; b.keys = c->mesh_keys; b.grid_blocks = c->brute_blocks;
  b.counters = c->counters;
  return b;
}

f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

void affine_from_mat4(Affine& a, const float m[16]) {
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 3; ++row) a.c[col][row] = m[4 * col + row];
}

FrameParams make_params(const trc_ctx* c, const trc_camera* cam, uint64_t seed) {
  FrameParams fp;
  set_dims(fp, c->w, c->h);
  if (c->tile_rows) set_tile(fp, c->tile_row0, c->tile_rows);
  fp.campos = ld3(cam->campos); fp.A = ld3(cam->A); fp.B = ld3(cam->B); fp.C = ld3(cam->C);
  fp.nA = normalize(fp.A);       // raytrace.cu:94 evaluates these per thread; same IEEE ops here
  fp.nB = normalize(fp.B);
  fp.lensRadius = cam->lensRadius; fp.focalDist = cam->focalDist;
  fp.sc = c->sc;
  fp.seed_lo = (uint32_t)seed; fp.seed_hi = (uint32_t)(seed >> 32);
  fp.rng_mode = c->rng_mode;
  return fp;
}

// Ray.inl:62 for the six axis-aligned object-space normals (see DevObject::axis_nor).
void fill_axis_normals(DevObject& d) {
  const f3 axes[6] = { mk3(-1.0f, 0.0f, 0.0f), mk3(1.0f, 0.0f, 0.0f), mk3(0.0f, -1.0f, 0.0f),
                       mk3(0.0f, 1.0f, 0.0f), mk3(0.0f, 0.0f, -1.0f), mk3(0.0f, 0.0f, 1.0f) };
  for (int k = 0; k < 6; ++k) d.axis_nor[k] = normalize(xform_dir(d.m, axes[k]));
}

// Flatten the caller's Object[]/Mesh AoS into DevObject records + one triangle stream.
int flatten_scene(const RefObject* objs, uint32_t count, std::vector<DevObject>& out_objs,
                  std::vector<DevTri>& out_tris) {
  out_objs.resize(count);
  out_tris.clear();
  for (uint32_t i = 0; i < count; ++i) {
    const RefObject& o = objs[i];
    if (!o.mesh) return fail(TRC_ERR_INVALID, "object " + std::to_string(i) + " has a null m_mesh");
    const RefMesh& m = *o.mesh;
    DevObject d;
    memset(&d, 0, sizeof d);
    affine_from_mat4(d.mi, o.matrixi);
    affine_from_mat4(d.m, o.matrix);
    d.bmin = m.bmin; d.bmax = m.bmax;
    d.color = o.material.color; d.emit = o.material.emit; d.brdf = o.material.brdf; d.ior = o.material.n;
    d.mat_type = o.material.type;
    d.mesh_type = m.type;
    if (d.mesh_type < REF_MESH || d.mesh_type > REF_PLANE)
      return fail(TRC_ERR_INVALID, "object " + std::to_string(i) + ": unknown MeshType");
    if (d.mat_type < REF_DIFF || d.mat_type > REF_MIRR)
      return fail(TRC_ERR_INVALID, "object " + std::to_string(i) + ": unknown Material::Type");
    fill_axis_normals(d);
    d.tri_first = (uint32_t)out_tris.size();
    d.tri_count = 0;
    if (d.mesh_type == REF_MESH) {
      if (m.numFaces && (!m.verts || !m.norms || !m.faces))
        return fail(TRC_ERR_INVALID, "object " + std::to_string(i) + ": MESH with null arrays");
      d.tri_count = m.numFaces;
      out_tris.reserve(out_tris.size() + m.numFaces);
      for (uint32_t f = 0; f < m.numFaces; ++f) {
        const RefFace& fc = m.faces[f];
        for (int k = 0; k < 3; ++k)
          if (fc.v[k] >= m.numVerts || fc.n[k] >= m.numNorms)
            return fail(TRC_ERR_INVALID, "object " + std::to_string(i) + ": face index out of range");
        DevTri t;
        memset(&t, 0, sizeof t);
        // Mesh::getTriangle (Mesh.inl:56-62) + the edge vectors of Ray.inl:190-191
        t.v0 = m.verts[fc.v[0]];
        t.e1 = m.verts[fc.v[1]] - m.verts[fc.v[0]];
        t.e2 = m.verts[fc.v[2]] - m.verts[fc.v[0]];
        t.n0 = m.norms[fc.n[0]]; t.n1 = m.norms[fc.n[1]]; t.n2 = m.norms[fc.n[2]];
        out_tris.push_back(t);
      }
      const MeshHull hull = mesh_hull(d, out_tris.data());
      d.hull_c = hull.centre; d.hull_r2 = hull.r2;
    }
    out_objs[i] = d;
  }
  return TRC_OK;
}

// The zero pattern of m_matrixi an object is filed under (scene.h: DevScene::plane_end): planes use all five, the
// other types are not sorted (caller order within the type).
uint32_t filed_pattern(const DevObject& o) {
  return o.mesh_type == REF_PLANE ? classify_xform(o.mi) : (uint32_t)XP_FULL;
}

// Group the flattened objects by primitive type and lay everything out in one buffer
// (scene.h: DevScene). `base` is the device address the buffer will live at.
size_t pack_scene(const std::vector<DevObject>& objs, const std::vector<DevTri>& tris,
                  std::vector<unsigned char>& blob, std::vector<BigMesh>& big) {
  std::vector<DevXform> xf; std::vector<DevMeshExtra> meshes;
  std::vector<HotTri> hot;
  std::vector<float> bound;          // 4 floats per `hot` record
  big.clear();
  std::vector<DevMaterial> mats(objs.size());
  const int order[4] = { REF_PLANE, REF_SPHERE, REF_CUBE, REF_MESH };
  for (int g = 0; g < 4; ++g)
   for (uint32_t pat = XP_FULL; pat <= XP_BLOCK_Z; ++pat)            // within a type: sorted by filed_pattern()
    for (size_t i = 0; i < objs.size(); ++i) {
      const DevObject& o = objs[i];
      if (o.mesh_type != order[g] || filed_pattern(o) != pat) continue;
      DevXform x; memset(&x, 0, sizeof x);
      x.mi = o.mi; x.m = o.m; x.nor_y = o.axis_nor[3]; x.id = (uint32_t)i;
      if (o.mesh_type != REF_PLANE) {                 // the (ray, object) pre-test of intersect.h
        const ObjectBound b = object_bound(o);
        x.nor_y = b.centre; x.bound_a = b.a; x.bound_far2 = b.far2;
        x.bmin = o.bmin; x.bmax = o.bmax;
        if (o.mesh_type == REF_MESH) { x.hull_r2 = o.hull_r2; x.hull_cx = o.hull_c.x; x.hull_cy = o.hull_c.y; x.hull_cz = o.hull_c.z; }
      }
      xf.push_back(x);
      if (o.mesh_type == REF_MESH) {
        DevMeshExtra e; e.bmin = o.bmin; e.tri_first = o.tri_first; e.bmax = o.bmax; e.tri_count = o.tri_count;
        const uint32_t G = o.tri_count < 64u ? (o.tri_count ? o.tri_count : 1u) : 64u;
        e.group_magic = 65535u / G + 1u; e.rays_per_pass = (64u / G) | ((65535u / (64u / G) + 1u) << 8);
        e.big_slot = kNotBig; e.hot_first = 0;
        if (o.tri_count >= kBigMeshTris) {
          // the 36 bytes per triangle that every test of the brute-force pass reads, padded to whole
          // tiles with zero triangles (det == 0 -> rejected at Ray.inl:196 by every ray)
          e.big_slot = (uint32_t)big.size(); e.hot_first = (uint32_t)hot.size();
          BigMesh b; b.mesh_index = (uint32_t)meshes.size(); b.hot_first = e.hot_first;
          b.tri_first = o.tri_first; b.tri_count = o.tri_count;
          std::vector<float> centres;
          mesh_bound_spheres(tris.data() + o.tri_first, o.tri_count, centres, b.prm, b.r_max, b.origin_dist);
          b.use_bound = b.r_max <= 0.25f * b.prm.radius;
          bound.insert(bound.end(), centres.begin(), centres.end());
          big.push_back(b);
          for (uint32_t t = 0; t < o.tri_count; ++t) {
            const DevTri& src = tris[o.tri_first + t];
            HotTri h; memset(&h, 0, sizeof h);
            h.v0 = src.v0; h.e1 = src.e1; h.e2 = src.e2;
            hot.push_back(h);
          }
          HotTri zero; memset(&zero, 0, sizeof zero);
          while (hot.size() % kMeshTile) hot.push_back(zero);
          bound.resize(4 * hot.size(), 0.0f);
        }
        meshes.push_back(e);
      }
    }
  for (size_t i = 0; i < objs.size(); ++i) {
    DevMaterial& m = mats[i]; memset(&m, 0, sizeof m);
    m.color = objs[i].color; m.emit = objs[i].emit; m.brdf = objs[i].brdf; m.ior = objs[i].ior; m.mat_type = objs[i].mat_type;
    m.inv_ior = 1.0f / objs[i].ior;
  }
  auto append = [&](const void* p, size_t n) {
    size_t off = blob.size();
    blob.resize(off + ((n + 255) & ~size_t(255)));             // keep every section 256-byte aligned
    if (n) memcpy(blob.data() + off, p, n);
    return off;
  };
  blob.clear();
  append(xf.data(), xf.size() * sizeof(DevXform));
  append(meshes.data(), meshes.size() * sizeof(DevMeshExtra));
  append(mats.data(), mats.size() * sizeof(DevMaterial));
  append(tris.data(), tris.size() * sizeof(DevTri));
  append(hot.data(), hot.size() * sizeof(HotTri));
  append(bound.data(), bound.size() * sizeof(float));
  blob.resize(blob.size() + 256);                              // slack: empty sections still get valid addresses
  return blob.size();
}

DevScene scene_pointers(const std::vector<DevObject>& objs, const std::vector<DevTri>& tris, void* base) {
  DevScene sc; memset(&sc, 0, sizeof sc);
  uint32_t plane_pat[5] = { 0, 0, 0, 0, 0 };
  for (const DevObject& o : objs) {
    const uint32_t pat = filed_pattern(o);
    if (o.mesh_type == REF_PLANE) { ++sc.n_plane; ++plane_pat[pat]; }
    else if (o.mesh_type == REF_SPHERE) ++sc.n_sphere;
    else if (o.mesh_type == REF_CUBE) ++sc.n_cube;
    else ++sc.n_mesh;
  }
  for (uint32_t pat = XP_FULL, end = 0; pat < XP_BLOCK_Z; ++pat) { end += plane_pat[pat]; sc.plane_end[pat] = end; }
  auto up = [](size_t n) { return (n + 255) & ~size_t(255); };
  unsigned char* p = (unsigned char*)base;
  sc.xf = (const DevXform*)p;         p += up(objs.size() * sizeof(DevXform));
  sc.meshes = (const DevMeshExtra*)p; p += up(sc.n_mesh * sizeof(DevMeshExtra));
  sc.mats = (const DevMaterial*)p;    p += up(objs.size() * sizeof(DevMaterial));
  sc.tris = (const DevTri*)p;         p += up(tris.size() * sizeof(DevTri));
  sc.hot = (const HotTri*)p;
  size_t n_hot = 0;                   // as pack_scene lays them out: every big mesh padded to whole tiles
  for (const DevObject& o : objs)
    if (o.mesh_type == REF_MESH && o.tri_count >= kBigMeshTris) n_hot += (o.tri_count + kMeshTile - 1) / kMeshTile * kMeshTile;
  p += up(n_hot * sizeof(HotTri));
  sc.tri_bound = (const float*)p;
  return sc;
}

// (Re)upload the packed scene; grows the device allocation when needed.
int upload_packed(void*& dev, size_t& cap, DevScene& sc, const std::vector<DevObject>& objs,
                  const std::vector<DevTri>& tris, bool objects_only, std::vector<BigMesh>* big_out = nullptr) {
  std::vector<unsigned char> blob;
  std::vector<BigMesh> big;
  const size_t bytes = pack_scene(objs, tris, blob, big);
  if (big_out) big_out->swap(big);
  if (bytes > cap) {
    if (dev) { (void)hipFree(dev); dev = nullptr; cap = 0; }
    HIP_TRY(hipMalloc(&dev, bytes));
    cap = bytes;
    objects_only = false;
  }
  sc = scene_pointers(objs, tris, dev);
  const size_t n = objects_only ? (size_t)((const unsigned char*)sc.tris - (const unsigned char*)dev) : bytes;
  HIP_TRY(hipMemcpy(dev, blob.data(), n, hipMemcpyHostToDevice));
  return TRC_OK;
}

int begin_timing(trc_ctx* c) { HIP_TRY(hipEventRecord(c->ev_start, c->stream)); return TRC_OK; }
int end_timing(trc_ctx* c) { HIP_TRY(hipEventRecord(c->ev_stop, c->stream)); c->timed = true; return TRC_OK; }

} // namespace

extern "C" {

const char* trc_last_error_string(void) { return g_last_error.c_str(); }
const char* trc_version(void) { return "tracer_amd 0.1 (gfx950)"; }

int trc_create(int device, uint32_t width, uint32_t height, void* stream, trc_ctx** out) {
  if (!out) return fail(TRC_ERR_INVALID, "trc_create: out is null");
  *out = nullptr;
  if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
    return fail(TRC_ERR_INVALID, "trc_create: bad image size");
  ONE_HIP_RUNTIME("trc_create");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(TRC_ERR_INVALID, "trc_create: no such device");
  HIP_TRY(hipSetDevice(device));
  trc_ctx* c = new trc_ctx;
  c->device = device; c->w = width; c->h = height; c->npix = width * height;
  int rc = [&]() -> int {
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIP_TRY(hipEventCreate(&c->ev_start));
    HIP_TRY(hipEventCreate(&c->ev_mid));
    HIP_TRY(hipEventCreate(&c->ev_stop));
    HIP_TRY(hipMalloc((void**)&c->film_own, sizeof(float) * 3ull * c->npix));
    HIP_TRY(hipMalloc((void**)&c->ldr, sizeof(uint32_t) * c->npix));
    HIP_TRY(hipMalloc((void**)&c->counters, sizeof(DeviceCounters)));
    c->film = c->film_own;
    HIP_TRY(hipMemsetAsync(c->film_own, 0, sizeof(float) * 3ull * c->npix, c->stream));
    HIP_TRY(hipMemsetAsync(c->ldr, 0, sizeof(uint32_t) * c->npix, c->stream));
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(DeviceCounters), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->resident_blocks = (unsigned)prop.multiProcessorCount * 4u;   // work queue: 4 x 256 threads per CU = 4 waves per SIMD
    void* sp = nullptr;
    { int rs = resolve_scratch(c->device, &sp); if (rs != TRC_OK) return rs; }
    c->scratch_d = (unsigned char*)sp;
    return TRC_OK;
  }();
  if (rc != TRC_OK) { trc_destroy(c); return rc; }
  *out = c;
  return TRC_OK;
}

int trc_destroy(trc_ctx* c) {
  if (!c) return TRC_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  free_dev(c->scene_d);
  free_dev(c->film_own); free_dev(c->ldr);
  free_dev(c->hit_id); free_dev(c->hit_t); free_dev(c->hit_pos); free_dev(c->hit_nor);
  free_dev(c->state_mem[0]); free_dev(c->state_mem[1]); free_dev(c->queue_count);
  free_dev(c->counters); free_dev(c->stage); free_dev(c->next_task);
  free_dev(c->mesh_keys); free_dev(c->mesh_queue_mem); free_dev(c->batch_mem);
  free_bvh(c);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TRC_OK;
}

int trc_set_option(trc_ctx* c, int option, int value) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  switch (option) {
    case TRC_OPT_RNG_MODE:
      if (value != TRC_RNG_REFERENCE && value != TRC_RNG_DECORRELATED) return fail(TRC_ERR_INVALID, "bad rng mode");
      c->rng_mode = value; return TRC_OK;
    case TRC_OPT_PIPELINE:
      if (value < TRC_PIPE_WAVEFRONT || value > TRC_PIPE_AUTO) return fail(TRC_ERR_INVALID, "bad pipeline");
      c->pipeline = value; return TRC_OK;
    case TRC_OPT_RECORD_HITS:
      c->record_hits = value ? 1 : 0; return TRC_OK;
    case TRC_OPT_MESH_ACCEL:
      if (value < 0 || value > 2) return fail(TRC_ERR_INVALID, "bad mesh acceleration mode");
      c->mesh_accel = value; return TRC_OK;
    case TRC_OPT_STAGE_MB:
      if (value < 1) return fail(TRC_ERR_INVALID, "bad staging budget");
      c->stage_mb = value; return TRC_OK;
    case TRC_OPT_SKIP_TONEMAP:
      c->skip_tonemap = value ? 1 : 0; return TRC_OK;
    default: return fail(TRC_ERR_INVALID, "unknown option");
  }
}

int trc_set_tile(trc_ctx* c, uint32_t row0, uint32_t rows) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (rows == 0) { c->tile_row0 = 0; c->tile_rows = 0; return TRC_OK; }
  if (row0 >= c->h || rows > c->h - row0) return fail(TRC_ERR_INVALID, "trc_set_tile: rows outside the image");
  c->tile_row0 = row0; c->tile_rows = rows;
  return TRC_OK;
}

int trc_upload_scene(trc_ctx* c, const void* objects, uint32_t count) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (!objects || count == 0) return fail(TRC_ERR_INVALID, "trc_upload_scene: empty scene");
  std::vector<DevObject> objs; std::vector<DevTri> tris;
  int rc = flatten_scene((const RefObject*)objects, count, objs, tris);
  if (rc != TRC_OK) return rc;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->nobj = 0;
  free_bvh(c);
  rc = upload_packed(c->scene_d, c->scene_cap, c->sc, objs, tris, false, &c->big);
  if (rc != TRC_OK) return rc;
  c->objs_h.swap(objs);
  c->tris_h.swap(tris);
  c->nobj = count;
  return TRC_OK;
}

int trc_update_object(trc_ctx* c, uint32_t index, const float matrix[16], const float matrixi[16]) {
  if (!c || !matrix || !matrixi) return fail(TRC_ERR_INVALID, "null argument");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  if (index >= c->nobj) return fail(TRC_ERR_INVALID, "trc_update_object: index out of range");
  HIP_TRY(hipSetDevice(c->device));
  DevObject& d = c->objs_h[index];
  affine_from_mat4(d.m, matrix);
  affine_from_mat4(d.mi, matrixi);
  fill_axis_normals(d);
  // stream-ordered like the reference's cudaMemcpy on the default stream (Renderer.cpp:224)
  HIP_TRY(hipStreamSynchronize(c->stream));
  return upload_packed(c->scene_d, c->scene_cap, c->sc, c->objs_h, c->tris_h, true, &c->big);
}

int trc_raytrace1(trc_ctx* c, const trc_camera* cam) {
  if (!c || !cam) return fail(TRC_ERR_INVALID, "null argument");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  HIP_TRY(hipSetDevice(c->device));
  if (c->record_hits) { int rc = ensure_hit_buffers(c); if (rc) return rc; }
  FrameParams fp = make_params(c, cam, 0);
  int rc = ensure_big_mesh_buffers(c, c->npix); if (rc) return rc;
  BigMeshPass bmp = big_mesh_pass(c);
  const BigMeshPass* pass = c->big.empty() ? nullptr : &bmp;
  if (pass) { fp.sc.mesh_keys = c->mesh_keys; fp.sc.key_stride = (uint32_t)c->mesh_slots; }
  rc = begin_timing(c); if (rc) return rc;
  if (c->record_hits) launch_raytrace1(fp, c->ldr, c->hit_id, c->hit_t, c->hit_pos, c->hit_nor, c->counters, pass, c->stream);
  else launch_raytrace1(fp, c->ldr, nullptr, nullptr, nullptr, nullptr, c->counters, pass, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
  return end_timing(c);
}

// Ray::intersectScene for caller rays (main.cpp:202 calls it on the CPU for the picking ray).
int trc_intersect_rays_device(trc_ctx* c, const void* rays6_d, uint32_t n, void* ids_d, void* ts_d,
                              void* pos3_d, void* nor3_d) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  if (n == 0) return TRC_OK;
  if (!rays6_d) return fail(TRC_ERR_INVALID, "trc_intersect_rays: null ray array");
  if (n > 0x7FFFFFFFu) return fail(TRC_ERR_INVALID, "trc_intersect_rays: more than 2^31 - 1 rays");
  HIP_TRY(hipSetDevice(c->device));
  trc_camera none; memset(&none, 0, sizeof none);
  FrameParams fp = make_params(c, &none, 0);
  int rc = ensure_big_mesh_buffers(c, n); if (rc) return rc;
  BigMeshPass bmp = big_mesh_pass(c);
  const BigMeshPass* pass = c->big.empty() ? nullptr : &bmp;
  if (pass) { fp.sc.mesh_keys = c->mesh_keys; fp.sc.key_stride = (uint32_t)c->mesh_slots; }
  // (query rays are not path segments: k_intersect_rays adds its triangle tests to the counters, nothing else)
  launch_intersect_rays(fp, (const float*)rays6_d, n, (int32_t*)ids_d, (float*)ts_d, (float*)pos3_d, (float*)nor3_d,
                        c->counters, pass, c->stream);
  HIP_TRY(hipGetLastError());
  return TRC_OK;
}

int trc_intersect_rays(trc_ctx* c, const float* rays6, uint32_t n, int32_t* ids, float* ts, float* pos3, float* nor3) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  if (n == 0) return TRC_OK;
  if (!rays6) return fail(TRC_ERR_INVALID, "trc_intersect_rays: null ray array");
  if (n > 0x7FFFFFFFu) return fail(TRC_ERR_INVALID, "trc_intersect_rays: more than 2^31 - 1 rays");
  HIP_TRY(hipSetDevice(c->device));
  const size_t need = 14ull * n;               // floats
  if (need > c->batch_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_dev(c->batch_mem); c->batch_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->batch_mem, sizeof(float) * need));
    c->batch_cap = need;
  }
  float* d_rays = c->batch_mem; float* d_pos = d_rays + 6ull * n; float* d_nor = d_pos + 3ull * n;
  float* d_t = d_nor + 3ull * n; int32_t* d_id = (int32_t*)(d_t + n);
  HIP_TRY(hipMemcpyAsync(d_rays, rays6, sizeof(float) * 6ull * n, hipMemcpyHostToDevice, c->stream));
  int rc = trc_intersect_rays_device(c, d_rays, n, d_id, d_t, d_pos, d_nor); if (rc) return rc;
  if (ids) HIP_TRY(hipMemcpyAsync(ids, d_id, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (ts) HIP_TRY(hipMemcpyAsync(ts, d_t, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (pos3) HIP_TRY(hipMemcpyAsync(pos3, d_pos, sizeof(float) * 3ull * n, hipMemcpyDeviceToHost, c->stream));
  if (nor3) HIP_TRY(hipMemcpyAsync(nor3, d_nor, sizeof(float) * 3ull * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

// main.cpp:180-208: the pixel-corner ray through window position (x, y), y counted from the top as GLUT reports it.
int trc_pick(trc_ctx* c, const trc_camera* cam, uint32_t window_w, uint32_t window_h, int32_t x, int32_t y,
             int32_t* id, float* t, float* pos3, float* nor3) {
  if (!c || !cam) return fail(TRC_ERR_INVALID, "null argument");
  if (window_w == 0 || window_h == 0 || window_w > 0x7FFFFFFFu || window_h > 0x7FFFFFFFu)
    return fail(TRC_ERR_INVALID, "trc_pick: bad window size");
  const int W = (int)window_w, H = (int)window_h;
  const float u = (float)x / W, v = (float)(H - y) / H;                                  // main.cpp:196
  const f3 campos = ld3(cam->campos), A = ld3(cam->A), B = ld3(cam->B), Cc = ld3(cam->C);
  const f3 pos = campos + Cc + (2.0f * u - 1.0f) * A + (2.0f * v - 1.0f) * B;           // :198
  const f3 dir = normalize(pos - campos);                                                // :199
  const float ray6[6] = { pos.x, pos.y, pos.z, dir.x, dir.y, dir.z };
  return trc_intersect_rays(c, ray6, 1, id, t, pos3, nor3);                              // :202
}

int trc_pathtrace(trc_ctx* c, const trc_camera* cam, uint64_t seed,
                  uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters) {
  if (!c || !cam) return fail(TRC_ERR_INVALID, "null argument");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  if (iter_first == 0) return fail(TRC_ERR_INVALID, "trc_pathtrace: iterations are 1-based");
  if (n_iters == 0) return TRC_OK;
  if (n_iters > 1 && iter_stride == 0) return fail(TRC_ERR_INVALID, "trc_pathtrace: zero stride");
  if ((uint64_t)iter_first + (uint64_t)(n_iters - 1) * iter_stride > 0xFFFFFFFFull)
    return fail(TRC_ERR_INVALID, "trc_pathtrace: iteration number overflows 32 bits");
  HIP_TRY(hipSetDevice(c->device));
  FrameParams fp = make_params(c, cam, seed);
  const uint32_t last_iter = iter_first + (n_iters - 1) * iter_stride;
  // TRC_PIPE_AUTO: the work queue; the compacted pipeline when a big mesh makes the per-bounce brute-force pass
  // worth a launch; the fused kernel for a small call (an interactive frame: one iteration of 512x512 takes
  // 0.07 ms fused, 0.16 ms through the queue; the queue wins from about 2.4 M paths on:
  // 8 iterations of 512x512 take 0.48 ms fused and 0.51 ms through the queue, 12 take 0.72 and 0.52)
  const int pipeline = c->pipeline != TRC_PIPE_AUTO ? c->pipeline
                     : !c->big.empty() ? TRC_PIPE_COMPACTED
                     : (uint64_t)fp.ntile * n_iters <= 2400000ull ? TRC_PIPE_FUSED : TRC_PIPE_QUEUE;
  BigMeshPass bmp = {};
  const BigMeshPass* pass = nullptr;
  const bool wave = pipeline == TRC_PIPE_WAVEFRONT || pipeline == TRC_PIPE_COMPACTED;
  // Iterations per launch sequence (work queue and wavefront pipelines): what fits the staging budget, at
  // least one; paths per chunk < 2^31; the wavefront pipelines also keep 40-80 B of ray state per path, so
  // they stop at 2^27 paths per chunk (10.7 GB of ray state).
  uint32_t chunk_iters = 0;
  if (pipeline == TRC_PIPE_QUEUE || wave) {
    const size_t per_iter = 3ull * fp.ntile;
    size_t fit = ((size_t)c->stage_mb << 20) / (sizeof(float) * per_iter);
    const size_t cap31 = 0x7FFFFFFFull / fp.ntile;
    if (fit > cap31) fit = cap31;
    // (round 3, headline workload through the compacted pipeline: chunks of 2^23 / 24 / 25 / 26 / 27 path slots take
    //  13.4 / 11.7 / 11.0 / 10.7 / 10.3 ms -- every launch sequence ends in a tail and 288 GB hold the 80 bytes per slot easily)
    if (wave && fit > (size_t(1) << 27) / fp.ntile) fit = (size_t(1) << 27) / fp.ntile;
    if (fit < 1) fit = 1;
    chunk_iters = (uint32_t)(fit < n_iters ? fit : n_iters);
    const size_t need = per_iter * chunk_iters;
    if (need > c->stage_floats) {
      free_dev(c->stage); c->stage_floats = 0;
      HIP_TRY(hipMalloc((void**)&c->stage, sizeof(float) * need));
      c->stage_floats = need;
    }
  }
  if (wave) {
    const size_t slots = wavefront_slots(fp.ntile * chunk_iters);
    int rc = ensure_wavefront_state(c, pipeline == TRC_PIPE_COMPACTED, slots); if (rc) return rc;
    rc = ensure_big_mesh_buffers(c, slots); if (rc) return rc;
    if (!c->big.empty()) {
      bmp = big_mesh_pass(c); pass = &bmp;
      fp.sc.mesh_keys = c->mesh_keys; fp.sc.key_stride = (uint32_t)c->mesh_slots;
    }
  }
  if (pipeline == TRC_PIPE_QUEUE && !c->next_task) HIP_TRY(hipMalloc((void**)&c->next_task, sizeof(uint32_t)));
  int rc = begin_timing(c); if (rc) return rc;
  if (pipeline == TRC_PIPE_QUEUE) {
    for (uint32_t done = 0; done < n_iters; done += chunk_iters) {
      const uint32_t n = (n_iters - done < chunk_iters) ? n_iters - done : chunk_iters;
      launch_queue(fp, iter_first + done * iter_stride, iter_stride, n, c->film, c->stage, c->next_task,
                   c->resident_blocks, c->counters, c->stream);
    }
  } else if (pipeline == TRC_PIPE_FUSED) {
    launch_fused(fp, iter_first, iter_stride, n_iters, c->film, c->counters, c->stream);
  } else {
    const bool compact = pipeline == TRC_PIPE_COMPACTED;
    for (uint32_t done = 0; done < n_iters; done += chunk_iters) {
      const uint32_t n = (n_iters - done < chunk_iters) ? n_iters - done : chunk_iters;
      launch_wavefront_chunk(fp, iter_first + done * iter_stride, iter_stride, n, compact, c->state[0],
                             compact ? c->state[1] : c->state[0], c->queue_count,
                             c->film, c->stage, c->counters, pass, c->stream);
    }
  }
  HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
  if (!c->skip_tonemap) launch_tonemap(c->film, c->npix, (float)last_iter, c->ldr, c->stream);
  HIP_TRY(hipGetLastError());
  return end_timing(c);
}

// The work-queue pipeline of trc_pathtrace with its scratch in the static device array: no hipMalloc, no hipFree,
// no allocating library call on this path (the context, its film and the scene were allocated by trc_create /
// trc_upload_scene). Same kernels, same chunking rule, same film bit for bit.
int trc_pathtrace_static(trc_ctx* c, const trc_camera* cam, uint64_t seed,
                         uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters) {
  if (!c || !cam) return fail(TRC_ERR_INVALID, "null argument");
  if (c->nobj == 0) return fail(TRC_ERR_NO_SCENE, "no scene uploaded");
  if (iter_first == 0) return fail(TRC_ERR_INVALID, "trc_pathtrace_static: iterations are 1-based");
  if (n_iters == 0) return TRC_OK;
  if (n_iters > 1 && iter_stride == 0) return fail(TRC_ERR_INVALID, "trc_pathtrace_static: zero stride");
  if ((uint64_t)iter_first + (uint64_t)(n_iters - 1) * iter_stride > 0xFFFFFFFFull)
    return fail(TRC_ERR_INVALID, "trc_pathtrace_static: iteration number overflows 32 bits");
  if (!c->scratch_d || !c->resident_blocks) return fail(TRC_ERR_UNSUPPORTED, "trc_pathtrace_static: no static scratch on this device");
  HIP_TRY(hipSetDevice(c->device));
  FrameParams fp = make_params(c, cam, seed);
  const size_t per_iter = sizeof(float) * 3ull * fp.ntile;       // staging of one iteration of the tile
  if (per_iter > kScratchStageBytes)
    return fail(TRC_ERR_UNSUPPORTED, "trc_pathtrace_static: one iteration of " + std::to_string(fp.ntile) +
                " pixels needs " + std::to_string(per_iter) + " bytes of sample staging, the static scratch holds " +
                std::to_string(kScratchStageBytes) + " (5592405 pixels): render in bands (trc_set_tile) or call trc_pathtrace");
  const uint32_t last_iter = iter_first + (n_iters - 1) * iter_stride;
  size_t fit = kScratchStageBytes / per_iter;
  const size_t cap31 = 0x7FFFFFFFull / fp.ntile;
  if (fit > cap31) fit = cap31;
  const uint32_t chunk_iters = (uint32_t)(fit < n_iters ? fit : n_iters);
  // the addresses the kernels use come from hipGetSymbolAddress (c->scratch_d), never from the array's name
  uint32_t* next_task = reinterpret_cast<uint32_t*>(c->scratch_d);
  float* stage = reinterpret_cast<float*>(c->scratch_d + kScratchHeader);
  // Under stream capture (hipStreamBeginCapture on this context's stream) the call records kernel and memset nodes
  // only: the timing events and the cross-context guard event are left out -- an event recorded inside a capture
  // cannot be queried or waited for outside it -- so trc_last_*_ms report "nothing rendered" afterwards, and replays
  // of the captured graph are the caller's to order against other users of this GPU's scratch.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(c->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  // One call at a time enqueues on a device's scratch, and its first launch waits (on the GPU) for the previous
  // call's last one, whichever context and stream that was on.
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  hipEvent_t guard = g_scratch_event[c->device];
  if (!capturing) {
    HIP_TRY(hipStreamWaitEvent(c->stream, guard, 0));
    int rc = begin_timing(c); if (rc) return rc;
  }
  for (uint32_t done = 0; done < n_iters; done += chunk_iters) {
    const uint32_t n = (n_iters - done < chunk_iters) ? n_iters - done : chunk_iters;
    // launch_queue zeroes the task counter (stream-ordered) before every launch; k_queue writes every sample of the
    // chunk before k_accumulate reads it, so whatever an earlier call left in the staging area is never read
    launch_queue(fp, iter_first + done * iter_stride, iter_stride, n, c->film, stage, next_task,
                 c->resident_blocks, c->counters, c->stream);
  }
  if (!capturing) {
    HIP_TRY(hipEventRecord(guard, c->stream));                   // the scratch is free again from here on
    HIP_TRY(hipEventRecord(c->ev_mid, c->stream));
  }
  if (!c->skip_tonemap) launch_tonemap(c->film, c->npix, (float)last_iter, c->ldr, c->stream);
  HIP_TRY(hipGetLastError());
  if (capturing) { c->timed = false; return TRC_OK; }
  return end_timing(c);
}

int trc_reset_film(trc_ctx* c) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->film, 0, sizeof(float) * 3ull * c->npix, c->stream));
  return TRC_OK;
}

int trc_tonemap(trc_ctx* c, float iters) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  launch_tonemap(c->film, c->npix, iters, c->ldr, c->stream);
  HIP_TRY(hipGetLastError());
  return TRC_OK;
}

int trc_read_film(trc_ctx* c, float* rgb) {
  if (!c || !rgb) return fail(TRC_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(rgb, c->film, sizeof(float) * 3ull * c->npix, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

int trc_write_film(trc_ctx* c, const float* rgb) {
  if (!c || !rgb) return fail(TRC_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->film, rgb, sizeof(float) * 3ull * c->npix, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

int trc_read_ldr(trc_ctx* c, uint32_t* bgra) {
  if (!c || !bgra) return fail(TRC_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(bgra, c->ldr, sizeof(uint32_t) * c->npix, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

int trc_read_hits(trc_ctx* c, int32_t* ids, float* ts, float* pos3, float* nor3) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (!c->hit_id) return fail(TRC_ERR_INVALID, "trc_read_hits: no trc_raytrace1 with TRC_OPT_RECORD_HITS yet");
  HIP_TRY(hipSetDevice(c->device));
  if (ids) HIP_TRY(hipMemcpyAsync(ids, c->hit_id, sizeof(int32_t) * c->npix, hipMemcpyDeviceToHost, c->stream));
  if (ts) HIP_TRY(hipMemcpyAsync(ts, c->hit_t, sizeof(float) * c->npix, hipMemcpyDeviceToHost, c->stream));
  if (pos3) HIP_TRY(hipMemcpyAsync(pos3, c->hit_pos, sizeof(float) * 3ull * c->npix, hipMemcpyDeviceToHost, c->stream));
  if (nor3) HIP_TRY(hipMemcpyAsync(nor3, c->hit_nor, sizeof(float) * 3ull * c->npix, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

int trc_synchronize(trc_ctx* c) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TRC_OK;
}

int trc_film_device_ptr(trc_ctx* c, void** dev_ptr) {
  if (!c || !dev_ptr) return fail(TRC_ERR_INVALID, "null argument");
  *dev_ptr = c->film;
  return TRC_OK;
}

int trc_bind_film(trc_ctx* c, void* dev_ptr) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  if (dev_ptr && ((uintptr_t)dev_ptr & 15)) return fail(TRC_ERR_INVALID, "trc_bind_film: pointer not 16-byte aligned");
  if (dev_ptr) ONE_HIP_RUNTIME("trc_bind_film");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->film = dev_ptr ? (float*)dev_ptr : c->film_own;
  return TRC_OK;
}

int trc_get_counters(trc_ctx* c, trc_counters* out) {
  if (!c || !out) return fail(TRC_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  std::vector<DeviceCounters> hv(1);
  DeviceCounters& h = hv[0];
  HIP_TRY(hipMemcpyAsync(&h, c->counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memset(out, 0, sizeof *out);
  for (unsigned b = 0; b < kCounterBanks; ++b) {
    out->live_segments += h.bank[b].live_segments; out->paths += h.bank[b].paths; out->tri_tests += h.bank[b].tri_tests;
    out->wave_cycles += h.bank[b].wave_cycles; out->wave_ref_ticks += h.bank[b].wave_ref_ticks;
  }
  return TRC_OK;
}

int trc_reset_counters(trc_ctx* c) {
  if (!c) return fail(TRC_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(DeviceCounters), c->stream));
  return TRC_OK;
}

int trc_last_render_ms(trc_ctx* c, float* ms) {
  if (!c || !ms) return fail(TRC_ERR_INVALID, "null argument");
  if (!c->timed) return fail(TRC_ERR_INVALID, "trc_last_render_ms: nothing rendered yet");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
  return TRC_OK;
}

int trc_last_kernel_ms(trc_ctx* c, float* ms) {
  if (!c || !ms) return fail(TRC_ERR_INVALID, "null argument");
  if (!c->timed) return fail(TRC_ERR_INVALID, "trc_last_kernel_ms: nothing rendered yet");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev_mid));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_mid));
  return TRC_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// The reference's own launcher signatures (raytrace.cu:189-238), for a Renderer.cpp that keeps
// its device-side bookkeeping unchanged. `scene_d` is the device copy of Object[] that
// Renderer::initScene built (Renderer.cpp:112-151): Object records whose m_mesh points at a
// device Mesh whose m_verts/m_norms/m_faces point at device arrays. Each call reads the small
// Object array back (168 B per object) so that Renderer::updateScene's in-place matrix edits
// (Renderer.cpp:221-225) are honoured; meshes are immutable in the reference and are read back
// and flattened once per device Mesh pointer.
// ---------------------------------------------------------------------------------------------
namespace {

struct LegacyMesh {
  RefMesh host;                 // host copy with host pointers
  std::vector<f3> verts, norms;
  std::vector<RefFace> faces;
};

// What the shim keeps per caller scene (key: scene_d). The packed copy is reused as long as the content digest of
// the caller's device scene (kernels.hip: k_scene_digest -- objects, mesh headers AND mesh arrays) has not changed;
// any change -- Renderer::updateScene's in-place matrix edit, a mesh freed and another one landing at the same
// address, vertices rewritten in place -- makes the next call read the scene back and flatten it again.
struct LegacyScene {
  uint32_t n_objects = 0;
  bool valid = false;
  unsigned long long digest = 0;
  void* scene_d = nullptr; size_t scene_cap = 0;   // packed copy (device)
  DevScene sc = {};
  DeviceCounters* counters = nullptr;
  unsigned long long* digest_d = nullptr;          // device accumulator of k_scene_digest
  unsigned long long* digest_h = nullptr;          // pinned host landing place
};

std::mutex g_legacy_mutex;
std::unordered_map<const void*, LegacyScene*> g_legacy_scenes;  // key: scene_d

void legacy_free(LegacyScene* ls) {
  if (!ls) return;
  free_dev(ls->scene_d); free_dev(ls->counters); free_dev(ls->digest_d);
  if (ls->digest_h) (void)hipHostFree(ls->digest_h);
  delete ls;
}

int legacy_digest(LegacyScene* ls, const void* scene_d, uint32_t n, unsigned long long* out) {
  if (!ls->digest_d) HIP_TRY(hipMalloc((void**)&ls->digest_d, sizeof(unsigned long long)));
  if (!ls->digest_h) HIP_TRY(hipHostMalloc((void**)&ls->digest_h, sizeof(unsigned long long), hipHostMallocDefault));
  launch_scene_digest(scene_d, n, ls->digest_d, nullptr);          // default stream, like the launchers themselves
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ls->digest_h, ls->digest_d, sizeof(unsigned long long), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  *out = *ls->digest_h;
  return TRC_OK;
}

int legacy_prepare(const void* scene_d, uint32_t sceneSize, LegacyScene** out) {
  if (!scene_d || sceneSize == 0) return fail(TRC_ERR_INVALID, "legacy launcher: empty scene");
  LegacyScene*& ls = g_legacy_scenes[scene_d];
  if (!ls) ls = new LegacyScene;
  if (ls->valid && ls->n_objects == sceneSize) {
    // the frame-to-frame case: one small kernel and an 8-byte read-back, no flattening, no upload
    unsigned long long d = 0;
    int rc = legacy_digest(ls, scene_d, sceneSize, &d); if (rc) return rc;
    if (d == ls->digest) { *out = ls; return TRC_OK; }
  }
  ls->valid = false;
  // first sight of this scene, or it changed: read it back through checked copies (a bad pointer is a HIP error here)
  std::vector<RefObject> objs(sceneSize);
  HIP_TRY(hipMemcpy(objs.data(), scene_d, sizeof(RefObject) * sceneSize, hipMemcpyDeviceToHost));
  std::unordered_map<const void*, std::unique_ptr<LegacyMesh>> meshes;   // key: device Mesh*, for this read-back only
  for (uint32_t i = 0; i < sceneSize; ++i) {
    const void* key = objs[i].mesh;
    if (!key) return fail(TRC_ERR_INVALID, "legacy launcher: null m_mesh");
    auto it = meshes.find(key);
    if (it == meshes.end()) {
      std::unique_ptr<LegacyMesh> lm(new LegacyMesh);
      HIP_TRY(hipMemcpy(&lm->host, key, sizeof(RefMesh), hipMemcpyDeviceToHost));
      RefMesh& m = lm->host;
      if (m.type == REF_MESH) {
        lm->verts.resize(m.numVerts); lm->norms.resize(m.numNorms); lm->faces.resize(m.numFaces);
        if (m.numVerts) HIP_TRY(hipMemcpy(lm->verts.data(), m.verts, sizeof(f3) * m.numVerts, hipMemcpyDeviceToHost));
        if (m.numNorms) HIP_TRY(hipMemcpy(lm->norms.data(), m.norms, sizeof(f3) * m.numNorms, hipMemcpyDeviceToHost));
        if (m.numFaces) HIP_TRY(hipMemcpy(lm->faces.data(), m.faces, sizeof(RefFace) * m.numFaces, hipMemcpyDeviceToHost));
      }
      m.verts = lm->verts.data(); m.norms = lm->norms.data(); m.faces = lm->faces.data();
      it = meshes.emplace(key, std::move(lm)).first;
    }
    objs[i].mesh = &it->second->host;
  }
  std::vector<DevObject> dobjs; std::vector<DevTri> dtris;
  int rc = flatten_scene(objs.data(), sceneSize, dobjs, dtris);
  if (rc) return rc;
  rc = upload_packed(ls->scene_d, ls->scene_cap, ls->sc, dobjs, dtris, false);
  if (rc) return rc;
  if (!ls->counters) {
    HIP_TRY(hipMalloc((void**)&ls->counters, sizeof(DeviceCounters)));
    HIP_TRY(hipMemset(ls->counters, 0, sizeof(DeviceCounters)));
  }
  // the digest of what was just read (the caller's scene is stream-ordered with these default-stream calls)
  rc = legacy_digest(ls, scene_d, sceneSize, &ls->digest); if (rc) return rc;
  ls->n_objects = sceneSize;
  ls->valid = true;
  *out = ls;
  return TRC_OK;
}

FrameParams legacy_params(uint32_t w, uint32_t h, const float* campos, const float* A, const float* B,
                          const float* Cc, float lensRadius, float focalDist, const LegacyScene* ls,
                          uint32_t nobj, uint64_t seed) {
  FrameParams fp;
  set_dims(fp, w, h);
  fp.campos = ld3(campos); fp.A = ld3(A); fp.B = ld3(B); fp.C = ld3(Cc);
  fp.nA = normalize(fp.A); fp.nB = normalize(fp.B);
  fp.lensRadius = lensRadius; fp.focalDist = focalDist;
  fp.sc = ls->sc; (void)nobj;
  fp.seed_lo = (uint32_t)seed; fp.seed_hi = (uint32_t)(seed >> 32);
  fp.rng_mode = TRC_RNG_REFERENCE;
  return fp;
}

} // namespace

extern "C" {

int trc_legacy_invalidate(const void* scene_d) {
  std::lock_guard<std::mutex> lock(g_legacy_mutex);
  if (!scene_d) { for (auto& kv : g_legacy_scenes) kv.second->valid = false; return TRC_OK; }
  auto it = g_legacy_scenes.find(scene_d);
  if (it != g_legacy_scenes.end()) it->second->valid = false;
  return TRC_OK;
}

int trc_legacy_release(void) {
  std::lock_guard<std::mutex> lock(g_legacy_mutex);
  for (auto& kv : g_legacy_scenes) legacy_free(kv.second);
  g_legacy_scenes.clear();
  return TRC_OK;
}

void raytrace1(uint32_t* pbo_out, uint32_t w, uint32_t h, float time,
               const float* campos, const float* A, const float* B, const float* C,
               const void* scene_d, uint32_t sceneSize) {
  (void)time;
  std::lock_guard<std::mutex> lock(g_legacy_mutex);
  LegacyScene* ls = nullptr;
  if (!pbo_out || !campos || !A || !B || !C || w == 0 || h == 0 ||
      legacy_prepare(scene_d, sceneSize, &ls) != TRC_OK) {
    fprintf(stderr, "tracer_amd: raytrace1: %s\n", g_last_error.empty() ? "bad argument" : g_last_error.c_str());
    return;
  }
  FrameParams fp = legacy_params(w, h, campos, A, B, C, 0.0f, 0.0f, ls, sceneSize, 0);
  launch_raytrace1(fp, pbo_out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);   // default stream, as the reference
}

void pathtrace(uint32_t* pbo_out, uint32_t w, uint32_t h, float time,
               const float* campos, const float* A, const float* B, const float* C,
               float lensRadius, float focalDist,
               const void* scene_d, uint32_t sceneSize,
               void* rand_d, void* rays_d, void* col_d, int* idx_d,
               void* film_d, uint32_t filmIters) {
  (void)rand_d; (void)rays_d; (void)col_d; (void)idx_d;   // ray state lives in registers
  std::lock_guard<std::mutex> lock(g_legacy_mutex);
  LegacyScene* ls = nullptr;
  if (!pbo_out || !film_d || !campos || !A || !B || !C || w == 0 || h == 0 || filmIters == 0 ||
      legacy_prepare(scene_d, sceneSize, &ls) != TRC_OK) {
    fprintf(stderr, "tracer_amd: pathtrace: %s\n", g_last_error.empty() ? "bad argument" : g_last_error.c_str());
    return;
  }
  const uint64_t seed = (uint64_t)(time * 100.0f);        // raytrace.cu:205
  FrameParams fp = legacy_params(w, h, campos, A, B, C, lensRadius, focalDist, ls, sceneSize, seed);
  launch_fused(fp, filmIters, 1, 1, (float*)film_d, ls->counters, nullptr);
  launch_tonemap((const float*)film_d, fp.npix, (float)filmIters, pbo_out, nullptr);
}

} // extern "C"
