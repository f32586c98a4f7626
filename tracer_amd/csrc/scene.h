// tracer_amd/csrc/scene.h -- scene description as the caller hands it over (the reference's
// pointer-based AoS) and as the kernels read it (flat, pointer-free, wave-uniform records).
#pragma once
#include <cstddef>
#include "vecmath.h"
#include "sparse_xform.h"

namespace trc {

// ---- constants that are part of the contract: reference src/common.h:31,37,40,41 ----------
constexpr float kEps = 0.001f;        // EPS
constexpr int   kPathDepth = 5;       // PATH_DEPTH
constexpr float kPi = 3.14159265358979323846264338327950288f;  // common.h:17 (float M_PI)

// ---- caller-side layout (host memory), mirrors of the reference's structs -----------------
// Object.inl:24-29, Mesh.inl:22-41, Material.inl:19-34 on LP64. Only read, never written.
enum RefMeshType { REF_MESH = 0, REF_SPHERE = 1, REF_CUBE = 2, REF_PLANE = 3 };   // Mesh.inl:32
enum RefMatType  { REF_DIFF = 0, REF_TRANS = 1, REF_MIRR = 2 };                  // Material.inl:19

struct RefFace { uint32_t v[3]; uint32_t n[3]; };
struct RefMesh {
  uint32_t numVerts, numNorms, numFaces;
  f3* verts; f3* norms; RefFace* faces;
  f3 bmin, bmax;
  int32_t type;
};
struct RefMaterial { f3 color; float emit, brdf, n; int32_t type; };
struct RefObject {
  float matrix[16];   // glm::mat4, column-major
  float matrixi[16];
  RefMesh* mesh;
  RefMaterial material;
};
static_assert(sizeof(RefFace) == 24, "Mesh::Face layout");
static_assert(sizeof(RefMesh) == 72, "Mesh::Mesh layout");
static_assert(sizeof(RefMaterial) == 28, "Material::Material layout");
static_assert(sizeof(RefObject) == 168, "Object::Object layout");

// ---- flattened object (host-side intermediate, one per caller object, original order) ------
struct alignas(16) DevObject {
  Affine   mi;           // world -> object, 12 floats   (Object::m_matrixi)
  Affine   m;            // object -> world, 12 floats   (Object::m_matrix)
  f3       bmin;         // MESH only (Mesh::m_bmin/m_bmax)
  uint32_t tri_first;    // first record in the triangle stream
  f3       bmax;
  uint32_t tri_count;    // Mesh::m_numFaces
  f3       color;        // Material::m_color
  float    emit;         // Material::m_emit
  float    brdf;         // Material::m_brdf
  float    ior;          // Material::m_n
  int32_t  mat_type;     // RefMatType
  int32_t  mesh_type;    // RefMeshType
  // World-space normals of the axis-aligned object-space normals, i.e. what Ray.inl:62
  // normalize(vec3(m_matrix * vec4(n, 0))) yields for n = -x,+x,-y,+y,-z,+z. PLANE always
  // reports +y (Ray.inl:151) and CUBE one of the six (Ray.inl:132-137), so the per-ray matrix
  // product + normalisation is replaced by a table filled on the host with the same IEEE ops.
  f3       axis_nor[6];
  float    pad_[2];
  // MESH: the sphere that holds every vertex (bound.h: mesh_hull) -- a property of the mesh, computed when it is flattened,
  // not on every matrix update
  f3       hull_c;
  float    hull_r2;
};
static_assert(sizeof(DevObject) == 256, "DevObject");

// ---- device-side layout -------------------------------------------------------------------
// Objects are GROUPED BY PRIMITIVE TYPE on the device: [planes | spheres | cubes | meshes].
// Every lane of a wavefront walks the same plane list (Ray.inl:70), so a plane's record is read wave-uniformly (scalar
// loads, or broadcast reads of the kernel's LDS copy); the planes are tested two per trip (two independent dependency
// chains for the VALU to interleave), and sorting them by the zero pattern of the matrices lets the loop body leave out
// the multiplications by exact zeros (sparse_xform.h). The other types follow as [spheres | cubes | meshes]: the table
// that the (ray, object) pairs of intersect.h index per lane. The reference's tie rule "lowest object index wins"
// (strict `<` in index order, Ray.inl:75) is kept by comparing (distance, original index).
struct alignas(16) DevXform {      // 160 bytes
  Affine   mi;                     // world -> object (Object::m_matrixi)
  Affine   m;                      // object -> world (Object::m_matrix)
  f3       nor_y;                  // PLANE: world normal of object-space +y, what a hit reports; SPHERE / CUBE / MESH: centre of the
                                   // object's world-space bounding sphere (bound.h)
  uint32_t id;                     // index in the caller's Object[] (Hit::m_id)
  float    hull_r2;                // MESH: squared radius (with its margin, bound.h: mesh_hull) of the sphere about hull_c that holds
                                   // every vertex of the mesh, in OBJECT space -- tighter than the AABB for round meshes
  float    bound_a, bound_far2;    // SPHERE / CUBE / MESH: bound.h (radius + the constant part of the slack; far limit)
  float    hull_cx;
  f3       bmin; float hull_cy;    // MESH: Mesh::m_bmin / m_bmax once more (DevMeshExtra), where a per-lane gather of the record finds them
  f3       bmax; float hull_cz;
};
static_assert(sizeof(DevXform) == 160, "DevXform");
// the 32 bytes behind the matrices, as one record (what the (ray, object) pre-test reads)
struct alignas(16) DevXformTail { f3 nor_y; uint32_t id; float hull_r2; float bound_a, bound_far2; float hull_cx; };
static_assert(sizeof(DevXformTail) == 32 && offsetof(DevXform, nor_y) == 96 && offsetof(DevXform, bound_far2) == 96 + 24, "DevXformTail");
struct alignas(16) DevMeshExtra {
  f3 bmin; uint32_t tri_first; f3 bmax; uint32_t tri_count;
  // cooperative test (intersect.h): G = min(tri_count, 64) triangles side by side, 64 / G rays per pass;
  // lane / G == (lane * group_magic) >> 16 for lane < 64; rays_per_pass = R | (65535 / R + 1) << 8, R = 64 / G:
  // n / R == (n * (rays_per_pass >> 8)) >> 16 for n < 128
  uint32_t group_magic, rays_per_pass;
  // big meshes (tri_count >= kBigMeshTris): number among the scene's big meshes (row of the per-ray
  // key table of the brute-force pass, kernels.hip: k_mesh_brute) and first record in DevScene::hot;
  // kNotBig for a mesh that is tested inside closest_hit().
  uint32_t big_slot, hot_first;
};
constexpr uint32_t kNotBig = 0xFFFFFFFFu;
constexpr uint32_t kBigMeshTris = 1024;     // from here on a mesh gets its own brute-force pass
constexpr uint32_t kMeshTile = 256;         // triangles per LDS tile of that pass
struct alignas(16) DevMaterial  { f3 color; float emit; float brdf; float ior; int32_t mat_type; float inv_ior; };   // inv_ior = 1.0f / ior
static_assert(sizeof(DevMeshExtra) == 48 && sizeof(DevMaterial) == 32, "extras");

struct DevTri;
struct HotTri;
// Pointers into ONE device allocation holding the packed scene.
struct DevScene {
  const DevXform*     xf;          // n_plane + n_sphere + n_cube + n_mesh records, in that order
  const DevMeshExtra* meshes;      // n_mesh
  const DevMaterial*  mats;        // nobj, indexed by ORIGINAL object index
  const DevTri*       tris;        // triangle stream
  const HotTri*       hot;         // big meshes only: the 36 bytes per triangle every test reads
  const float*        tri_bound;   // big meshes only, indexed like `hot`: centre of the face's bounding sphere (x, y, z, 0)
  uint32_t n_plane, n_sphere, n_cube, n_mesh;
  // Set per launch by the host: closest (t, face) per ray slot per big mesh as found by the
  // brute-force pass that ran before this kernel, keys[big_slot * key_stride + slot]; null in the
  // persistent pipelines, which test big meshes inside closest_hit() like small ones.
  const unsigned long long* mesh_keys;
  uint32_t key_stride, n_big;
  // The planes are sorted by the zero pattern of mi (XformPattern order: FULL, DIAG, BLOCK_X, BLOCK_Y, BLOCK_Z), so that
  // the kernels run one specialised loop per pattern. plane_end[P] = end of pattern P's run (BLOCK_Z's ends at n_plane).
  // Spheres, cubes and meshes are tested as (ray, object) pairs with a per-lane matrix: always the full product.
  uint32_t plane_end[4];
};

// Triangle stream: faces pre-gathered at upload (replaces the Face -> verts/norms double
// indirection of Mesh::getTriangle, Mesh.inl:50-65) into 80-byte records that are contiguous
// in HBM, so a tile of T triangles is one coalesced T*80-byte copy into LDS and one record is
// five 16-byte broadcast reads. e1 = v1 - v0 and e2 = v2 - v0 (Ray.inl:190-191) are the same
// IEEE subtractions the reference performs per ray, done once here.
struct alignas(16) DevTri {
  f3 v0; f3 e1; f3 e2;
  f3 n0; f3 n1; f3 n2;
  float pad[2];
};
static_assert(sizeof(DevTri) == 80, "DevTri");

// What the brute-force pass streams: v0, e1, e2 of DevTri and nothing else (the normals are read
// from DevTri only for the handful of triangles that pass both barycentric tests). 48-byte records =
// three 16-byte LDS reads; a mesh's records are padded to a whole tile with all-zero triangles,
// which every ray rejects at |det| < EPS (Ray.inl:196).
struct alignas(16) HotTri { f3 v0; f3 e1; f3 e2; float pad[3]; };
static_assert(sizeof(HotTri) == 48, "HotTri");

struct Ray { f3 pos, dir; };
struct Hit { float t; int id; f3 pos, nor; };

} // namespace trc
