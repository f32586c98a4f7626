// tracer_amd/csrc/host/SceneApi.cpp -- see SceneApi.h.
#include "SceneApi.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

#pragma clang fp contract(off)

using trc::f3;

namespace {

void set_err(std::string* err, const std::string& msg) { if (err) *err = msg; }

// Mesh.h:103-113: running min/max from FLT_MAX / FLT_MIN (sic)
void compute_bounds(Mesh::Mesh* m) {
  m->bmin = trc::splat3(FLT_MAX);
  m->bmax = trc::splat3(FLT_MIN);
  for (uint32_t i = 0; i < m->numVerts; ++i) {
    const f3 v = m->verts[i];
    m->bmin = trc::gmin(m->bmin, v);
    m->bmax = trc::gmax(m->bmax, v);
  }
}

template <typename T> T* dup_array(const std::vector<T>& v) {
  T* p = (T*)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

Mesh::Mesh* make_mesh(const std::vector<f3>& verts, const std::vector<f3>& norms, const std::vector<Mesh::Face>& faces) {
  Mesh::Mesh* m = (Mesh::Mesh*)calloc(1, sizeof(Mesh::Mesh));
  m->type = Mesh::MESH;
  m->numVerts = (uint32_t)verts.size(); m->numNorms = (uint32_t)norms.size(); m->numFaces = (uint32_t)faces.size();
  m->verts = dup_array(verts); m->norms = dup_array(norms); m->faces = dup_array(faces);
  compute_bounds(m);
  return m;
}

// one "a", "a/b", "a//c" or "a/b/c" token of an `f` line; indices are 1-based, 0 = absent
bool parse_corner(const std::string& tok, uint32_t& v, uint32_t& n) {
  v = 0; n = 0;
  unsigned a = 0, b = 0, c = 0;
  if (sscanf(tok.c_str(), "%u//%u", &a, &c) == 2) { v = a; n = c; return true; }
  if (sscanf(tok.c_str(), "%u/%u/%u", &a, &b, &c) == 3) { v = a; n = c; return true; }
  if (sscanf(tok.c_str(), "%u/%u", &a, &b) == 2) { v = a; return true; }
  if (sscanf(tok.c_str(), "%u", &a) == 1) { v = a; return true; }
  return false;
}

} // namespace

namespace Mesh {

Mesh* newGeometry(MeshType type) {
  Mesh* m = (Mesh*)calloc(1, sizeof(Mesh));
  m->type = type;
  m->bmin = trc::splat3(-0.5f);
  m->bmax = trc::splat3(0.5f);
  return m;
}

Mesh* fromArrays(const float* verts, uint32_t nverts, const float* norms, uint32_t nnorms,
                 const uint32_t* faces6, uint32_t nfaces) {
  std::vector<f3> v(nverts), n(nnorms);
  std::vector<Face> f(nfaces);
  if (nverts) memcpy(v.data(), verts, sizeof(f3) * nverts);
  if (nnorms) memcpy(n.data(), norms, sizeof(f3) * nnorms);
  if (nfaces) memcpy(f.data(), faces6, sizeof(Face) * nfaces);
  return make_mesh(v, n, f);
}

Mesh* loadObj(const std::string& filename, std::string* err) {
  std::ifstream in(filename.c_str());
  if (!in.good()) { set_err(err, "loading obj: cannot open " + filename); return nullptr; }
  std::vector<f3> verts, norms;
  std::vector<Face> faces;
  std::vector<bool> face_has_normals;
  std::string line;
  size_t lineno = 0;
  while (std::getline(in, line)) {
    ++lineno;
    float a, b, c;
    if (line.size() >= 2 && line[0] == 'v' && line[1] == ' ') {
      if (sscanf(line.c_str(), "v %f %f %f", &a, &b, &c) != 3) {
        set_err(err, filename + ":" + std::to_string(lineno) + ": vertex format"); return nullptr;
      }
      verts.push_back(trc::mk3(a, b, c));
    } else if (line.size() >= 3 && line[0] == 'v' && line[1] == 'n' && line[2] == ' ') {
      if (sscanf(line.c_str(), "vn %f %f %f", &a, &b, &c) != 3) {
        set_err(err, filename + ":" + std::to_string(lineno) + ": normal format"); return nullptr;
      }
      norms.push_back(trc::mk3(a, b, c));
    } else if (line.size() >= 2 && line[0] == 'f' && line[1] == ' ') {
      std::istringstream ss(line.substr(2));
      std::string tok;
      std::vector<uint32_t> cv, cn;
      while (ss >> tok) {
        uint32_t v, n;
        if (!parse_corner(tok, v, n) || v == 0) {
          set_err(err, filename + ":" + std::to_string(lineno) + ": face format"); return nullptr;
        }
        cv.push_back(v - 1); cn.push_back(n);
      }
      if (cv.size() < 3) { set_err(err, filename + ":" + std::to_string(lineno) + ": face with < 3 corners"); return nullptr; }
      // triangles as in the reference; larger polygons are fanned around corner 0
      for (size_t k = 1; k + 1 < cv.size(); ++k) {
        Face f;
        f.v[0] = cv[0]; f.v[1] = cv[k]; f.v[2] = cv[k + 1];
        const bool hn = cn[0] && cn[k] && cn[k + 1];
        f.n[0] = hn ? cn[0] - 1 : 0; f.n[1] = hn ? cn[k] - 1 : 0; f.n[2] = hn ? cn[k + 1] - 1 : 0;
        faces.push_back(f);
        face_has_normals.push_back(hn);
      }
    }
  }
  for (size_t i = 0; i < faces.size(); ++i) {
    Face& f = faces[i];
    for (int k = 0; k < 3; ++k)
      if (f.v[k] >= verts.size()) { set_err(err, filename + ": face " + std::to_string(i) + " vertex index out of range"); return nullptr; }
    if (face_has_normals[i]) {
      for (int k = 0; k < 3; ++k)
        if (f.n[k] >= norms.size()) { set_err(err, filename + ": face " + std::to_string(i) + " normal index out of range"); return nullptr; }
    } else {
      // no normals in the file: one geometric normal per face
      const f3 e1 = verts[f.v[1]] - verts[f.v[0]], e2 = verts[f.v[2]] - verts[f.v[0]];
      f3 g = trc::cross(e1, e2);
      const float l2 = trc::dot(g, g);
      g = l2 > 0.0f ? trc::normalize(g) : trc::mk3(0.0f, 1.0f, 0.0f);
      f.n[0] = f.n[1] = f.n[2] = (uint32_t)norms.size();
      norms.push_back(g);
    }
  }
  return make_mesh(verts, norms, faces);
}

Mesh* icosphere(const Mesh& base, int levels) {
  auto unit = [](f3 v) {
    float q = v.x * v.x + v.y * v.y;
    q = q + v.z * v.z;
    const float inv = 1.0f / sqrtf(q);
    return trc::mk3(v.x * inv, v.y * inv, v.z * inv);
  };
  std::vector<f3> verts(base.verts, base.verts + base.numVerts);
  for (f3& v : verts) v = unit(v);
  std::vector<uint32_t> tri(3ull * base.numFaces);
  for (uint32_t i = 0; i < base.numFaces; ++i) for (int k = 0; k < 3; ++k) tri[3ull * i + k] = base.faces[i].v[k];
  for (int l = 0; l < levels; ++l) {
    std::unordered_map<uint64_t, uint32_t> mid;
    mid.reserve(tri.size());
    auto midpoint = [&](uint32_t a, uint32_t b) {
      const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
      const uint64_t key = ((uint64_t)lo << 32) | hi;
      auto it = mid.find(key);
      if (it != mid.end()) return it->second;
      const f3 s = verts[lo] + verts[hi];
      const uint32_t id = (uint32_t)verts.size();
      verts.push_back(unit(trc::mk3(s.x * 0.5f, s.y * 0.5f, s.z * 0.5f)));
      mid.emplace(key, id);
      return id;
    };
    std::vector<uint32_t> next;
    next.reserve(tri.size() * 4);
    for (size_t f = 0; f < tri.size(); f += 3) {
      const uint32_t a = tri[f], b = tri[f + 1], c = tri[f + 2];
      const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      const uint32_t quad[12] = { a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca };
      next.insert(next.end(), quad, quad + 12);
    }
    tri.swap(next);
  }
  std::vector<Face> faces(tri.size() / 3);
  for (size_t i = 0; i < faces.size(); ++i)
    for (int k = 0; k < 3; ++k) { faces[i].v[k] = tri[3 * i + k]; faces[i].n[k] = tri[3 * i + k]; }
  // Ray.inl:196 rejects a triangle when |det| < EPS = 0.001 (absolute), and |det| <= |e1 x e2|: on the
  // unit sphere every face of level 5 and above is smaller than that and the mesh would be invisible.
  // From level 4 on the vertices are therefore scaled by 2^(level-3) (exact in binary) and the caller
  // scales the object down by the same factor; the normals stay the unit vectors.
  const float radius = icosphereRadius(levels);
  if (radius == 1.0f) return make_mesh(verts, verts, faces);
  std::vector<f3> scaled(verts);
  for (f3& v : scaled) v = trc::mk3(v.x * radius, v.y * radius, v.z * radius);
  return make_mesh(scaled, verts, faces);
}

float icosphereRadius(int levels) { return levels > 3 ? (float)(1u << (levels - 3)) : 1.0f; }

void freeMesh(Mesh* m) {
  if (!m) return;
  free(m->verts); free(m->norms); free(m->faces); free(m);
}

} // namespace Mesh

namespace Object {

static void update_inverse(Object& obj) {
  trc::mat4 m; memcpy(&m, obj.matrix, 64);
  const trc::mat4 mi = trc::inverse(m);
  memcpy(obj.matrixi, &mi, 64);
}

Object* newObject(Mesh::Mesh* mesh) {
  Object* o = (Object*)calloc(1, sizeof(Object));
  const trc::mat4 id = trc::identity4();
  memcpy(o->matrix, &id, 64); memcpy(o->matrixi, &id, 64);
  o->mesh = mesh;
  o->material = Material::defaults();
  return o;
}

void translate(Object& obj, f3 amount) {                       // Object.inl:56-59
  obj.matrix[12] += amount.x; obj.matrix[13] += amount.y; obj.matrix[14] += amount.z; obj.matrix[15] += 0.0f;
  update_inverse(obj);
}

void rotate(Object& obj, trc::quat q) {                        // Object.inl:61-65
  trc::mat4 m; memcpy(&m, obj.matrix, 64);
  const trc::mat4 r = trc::mul(trc::toMat4(q), m);
  memcpy(obj.matrix, &r, 64);
  update_inverse(obj);
}

void rotateIsolate(Object& obj, trc::quat q) {                 // Object.inl:68-79
  trc::mat4 m; memcpy(&m, obj.matrix, 64);
  const trc::f4 trans = m.col[3];
  m.col[3] = trc::mk4(0.0f, 0.0f, 0.0f, 0.0f);
  trc::mat4 r = trc::mul(trc::toMat4(q), m);
  r.col[3] = trans;
  memcpy(obj.matrix, &r, 64);
  update_inverse(obj);
}

void scale(Object& obj, f3 amount) {                           // Object.inl:86-89
  trc::mat4 m; memcpy(&m, obj.matrix, 64);
  const trc::mat4 r = trc::scale(m, amount);
  memcpy(obj.matrix, &r, 64);
  update_inverse(obj);
}

void scale(Object& obj, float amount) { scale(obj, trc::splat3(amount)); }   // Object.inl:81-84

} // namespace Object
