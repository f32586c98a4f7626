// tracer_amd/csrc/host/SceneApi.h -- host-side scene description with the reference's names:
// namespaces Material / Mesh / Object (reference src/Material.inl:19-34, src/Mesh.inl:22-41,
// src/Mesh.h:6-126, src/Object.inl:24-89). GLM-free; the structs are byte-compatible with the
// reference's (scene.h), so an Object::Object[] built here and one built by the reference's own
// Object.inl are interchangeable at trc_upload_scene / Renderer::initScene.
#pragma once
#include "../scene.h"
#include "linalg.h"
#include <string>
#include <vector>

namespace Material {
  enum Type { DIFF = trc::REF_DIFF, TRANS = trc::REF_TRANS, MIRR = trc::REF_MIRR };   // Material.inl:19
  using Material = trc::RefMaterial;
  // Material.inl:22-27: grey 0.5, no emission, brdf 1, n 1, diffuse
  inline Material defaults() { Material m; m.color = trc::splat3(0.5f); m.emit = 0.0f; m.brdf = 1.0f; m.n = 1.0f; m.type = DIFF; return m; }
}

namespace Mesh {
  enum MeshType { MESH = trc::REF_MESH, SPHERE = trc::REF_SPHERE, CUBE = trc::REF_CUBE, PLANE = trc::REF_PLANE }; // Mesh.inl:32
  using Face = trc::RefFace;
  using Mesh = trc::RefMesh;

  // Mesh.h:6-18. Analytic primitive with the unit bounding box. Caller owns the result (free with freeMesh).
  Mesh* newGeometry(MeshType type);
  // Mesh.h:20-126 dialect ("v", "vn", "f a//b c//d e//f") plus "f a/b/c" and "f a" forms; a mesh
  // without normals gets per-face geometric normals. Returns nullptr (and sets *err) on failure --
  // the reference calls exit(). The AABB follows Mesh.h:103-113 including its FLT_MIN start for
  // m_bmax, so meshes straddling the origin (all the reference's data files) get identical boxes.
  Mesh* loadObj(const std::string& filename, std::string* err = nullptr);
  // Build a MESH from arrays (verts/norms as xyz triples, faces as {v0,v1,v2,n0,n1,n2}).
  Mesh* fromArrays(const float* verts, uint32_t nverts, const float* norms, uint32_t nnorms,
                   const uint32_t* faces6, uint32_t nfaces);
  // Icosphere: `levels` 4-to-1 subdivisions of a base mesh's faces, vertices pushed to the unit
  // sphere, normals = positions (BASELINE.json configs[4]; no counterpart in the reference).
  Mesh* icosphere(const Mesh& base, int levels);
  float icosphereRadius(int levels);   // object-space radius of icosphere(base, levels): 2^max(0, levels-3)
  void freeMesh(Mesh* m);
}

namespace Object {
  using Object = trc::RefObject;
  // Object.inl:44-89. Every edit recomputes m_matrixi = inverse(m_matrix) (UPDATE_MAT_I, :54).
  // newObject returns malloc'd storage like the reference's (Object.inl:45); release with free().
  Object* newObject(Mesh::Mesh* mesh);
  void translate(Object& obj, trc::f3 amount);
  void rotate(Object& obj, trc::quat q);
  void rotateIsolate(Object& obj, trc::quat q);
  void scale(Object& obj, float amount);
  void scale(Object& obj, trc::f3 amount);
}
