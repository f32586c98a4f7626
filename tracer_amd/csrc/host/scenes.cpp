// tracer_amd/csrc/host/scenes.cpp -- include/tracer_host.h: the reference's hard-coded scene
// (src/main.cpp:274-396) expressed through the Object/Mesh API of SceneApi.h, the default camera,
// and headless image writers.
#include "../../../include/tracer_host.h"
#include "Renderer.h"
#include "SceneApi.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#pragma clang fp contract(off)

using trc::f3; using trc::mk3; using trc::splat3; using trc::angleAxis;

struct trc_scene {
  std::vector<Object::Object> objs;
  std::vector<Mesh::Mesh*> meshes;   // owned
  ~trc_scene() { for (Mesh::Mesh* m : meshes) Mesh::freeMesh(m); }
  Mesh::Mesh* own(Mesh::Mesh* m) { meshes.push_back(m); return m; }
  Object::Object& add(Mesh::Mesh* m) {
    Object::Object* o = Object::newObject(m);
    objs.push_back(*o);
    free(o);
    return objs.back();
  }
};

namespace {

void add_walls_and_light(trc_scene& s) {
  const float H = 5.0f;                                   // BOX_HDIM, main.cpp:281
  const f3 X = mk3(1.0f, 0.0f, 0.0f), Z = mk3(0.0f, 0.0f, 1.0f);
  Mesh::Mesh* plane = s.own(Mesh::newGeometry(Mesh::PLANE));
  struct Wall { bool rot; float angle; f3 axis; f3 at; f3 color; };
  const Wall walls[5] = {
    { false, 0.0f, X, mk3(0.0f, -H, 0.0f), splat3(1.0f) },            // bottom -y, main.cpp:288-292
    { true, 180.0f, X, mk3(0.0f, H, 0.0f), splat3(1.0f) },            // top +y,    :294-300
    { true, 90.0f, X, mk3(0.0f, 0.0f, -H), splat3(1.0f) },            // back -z,   :302-307
    { true, 90.0f, Z, mk3(H, 0.0f, 0.0f), mk3(0.5f, 0.5f, 1.0f) },    // right +x,  :316-322
    { true, -90.0f, Z, mk3(-H, 0.0f, 0.0f), mk3(1.0f, 0.5f, 0.5f) },  // left -x,   :324-329
  };
  for (const Wall& w : walls) {
    Object::Object& o = s.add(plane);
    if (w.rot) Object::rotate(o, angleAxis(w.angle, w.axis));
    Object::scale(o, splat3(H) * 2.0f);
    Object::translate(o, w.at);
    o.material.color = w.color;
  }
  Object::Object& light = s.add(s.own(Mesh::newGeometry(Mesh::CUBE)));   // ceiling light, :332-337
  Object::scale(light, mk3(H, 0.5f, H));
  Object::translate(light, mk3(0.0f, H, 0.0f));
  light.material.color = splat3(1.0f);
  light.material.emit = 2.0f;
}

} // namespace

extern "C" {

int trc_scene_cornell(int variant, int param, const char* data_dir, trc_scene** out) {
  if (!out) return TRC_ERR_INVALID;
  *out = nullptr;
  if (variant < 0 || variant > 2 || param < 0 || param > 10) return TRC_ERR_INVALID;
  const std::string obj = std::string(data_dir ? data_dir : ".") + "/icosahedron.obj";
  std::string err;
  trc_scene* s = new trc_scene;
  s->objs.reserve(16);
  add_walls_and_light(*s);
  Mesh::Mesh* ico = Mesh::loadObj(obj, &err);
  if (!ico) { fprintf(stderr, "tracer_amd: %s\n", err.c_str()); delete s; return TRC_ERR_INVALID; }
  s->own(ico);
  if (variant == 0 || variant == 1) {
    Object::Object& mirror = s->add(s->own(Mesh::newGeometry(Mesh::SPHERE)));   // main.cpp:370-376
    Object::scale(mirror, 4.0f);
    Object::translate(mirror, mk3(-2.5f, -3.0f, -3.0f));
    mirror.material.color = splat3(1.0f);
    mirror.material.type = Material::MIRR;
    Object::Object& glass = s->add(s->own(Mesh::newGeometry(Mesh::SPHERE)));    // :378-385
    Object::scale(glass, 4.0f);
    Object::translate(glass, mk3(2.5f, -3.0f, 0.0f));
    glass.material.color = splat3(1.0f);
    glass.material.type = Material::TRANS;
    glass.material.n = 1.6f;
    Object::Object& solid = s->add(ico);                                         // :387-395
    Object::scale(solid, 1.5f);
    Object::rotate(solid, angleAxis(-10.0f, mk3(1.0f, 0.0f, 0.0f)));
    Object::rotate(solid, angleAxis(25.0f, mk3(0.0f, 1.0f, 0.0f)));
    Object::translate(solid, mk3(-1.0f, -3.7f, 3.0f));
    solid.material.color = mk3(1.0f, 1.0f, 0.0f);
    if (variant == 1) { solid.material.type = Material::TRANS; solid.material.n = 1.6f; }
  } else {
    Object::Object& ball = s->add(s->own(Mesh::icosphere(*ico, param)));
    Object::scale(ball, 3.0f / Mesh::icosphereRadius(param));      // world radius 3 at every level
    Object::translate(ball, mk3(0.0f, -2.0f, 0.0f));
    ball.material.color = mk3(1.0f, 1.0f, 0.0f);
  }
  *out = s;
  return TRC_OK;
}

// ---- scene description files (the reference's scene is compiled in, main.cpp:274-396) ----
// One statement per line, `#` starts a comment. Transforms are applied in the order written, through
// the same Object::scale / rotate / translate calls main.cpp makes, so a file that lists main.cpp's
// calls reproduces its Object[] bit for bit (tracer_amd/data/cornell.scene, tests/test_host_mirror.py).
//   camera [fov F] [zoom Z] [focal D] [lens R] [rotate RX RY]     ThirdPersonCamera set-up of main.cpp:275-279
//   mesh NAME plane|cube|sphere                                   Mesh::newGeometry
//   mesh NAME obj PATH                                            Mesh::loadObj, PATH relative to the file
//   mesh NAME icosphere BASE LEVELS                               Mesh::icosphere of mesh BASE
//   object MESH                                                   Object::newObject; the statements below edit it
//   scale S | scale X Y Z      rotate DEGREES X Y Z      translate X Y Z
//   color R G B    emit E    brdf B    ior N    material diff|trans|mirr
int trc_scene_load(const char* path, uint32_t aspect_w, uint32_t aspect_h, trc_scene** out, trc_camera* cam_out,
                   char* err, uint32_t err_len) {
  auto fail = [&](const std::string& msg) {
    if (err && err_len) { snprintf(err, err_len, "%s", msg.c_str()); }
    return (int)TRC_ERR_INVALID;
  };
  if (!out || !path) return fail("null argument");
  *out = nullptr;
  std::ifstream in(path);
  if (!in) return fail(std::string("cannot open ") + path);
  std::string dir(path);
  const size_t slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
  trc_scene* s = new trc_scene;
  s->objs.reserve(64);
  std::map<std::string, Mesh::Mesh*> meshes;
  ThirdPersonCamera cam;
  cam.setFOV(30.0f);                                   // FOV, common.h:40
  if (aspect_w && aspect_h) cam.setAspect(aspect_w, aspect_h);
  float rot_x = 0.0f, rot_y = 0.0f;
  Object::Object* cur = nullptr;
  std::string line;
  int lineno = 0;
  auto bad = [&](const std::string& what) {
    delete s;
    return fail(std::string(path) + ":" + std::to_string(lineno) + ": " + what);
  };
  while (std::getline(in, line)) {
    ++lineno;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    std::istringstream ls(line);
    std::string kw;
    if (!(ls >> kw)) continue;
    std::vector<std::string> a;
    for (std::string t; ls >> t;) a.push_back(t);
    auto num = [&](size_t i, float& v) {
      if (i >= a.size()) return false;
      char* end = nullptr;
      v = strtof(a[i].c_str(), &end);
      return end && *end == 0 && end != a[i].c_str();
    };
    if (kw == "camera") {
      for (size_t i = 0; i < a.size();) {
        float v, w;
        if (a[i] == "rotate") { if (!num(i + 1, v) || !num(i + 2, w)) return bad("camera rotate RX RY"); rot_x = v; rot_y = w; i += 3; continue; }
        if (!num(i + 1, v)) return bad("camera " + a[i] + " needs a number");
        if (a[i] == "fov") cam.setFOV(v);
        else if (a[i] == "zoom") cam.zoom(v);
        else if (a[i] == "focal") cam.m_focalDist = v;
        else if (a[i] == "lens") cam.m_lensRadius = v;
        else return bad("unknown camera parameter " + a[i]);
        i += 2;
      }
    } else if (kw == "mesh") {
      if (a.size() < 2) return bad("mesh NAME KIND ...");
      if (meshes.count(a[0])) return bad("mesh " + a[0] + " defined twice");
      Mesh::Mesh* m = nullptr;
      if (a[1] == "plane") m = Mesh::newGeometry(Mesh::PLANE);
      else if (a[1] == "cube") m = Mesh::newGeometry(Mesh::CUBE);
      else if (a[1] == "sphere") m = Mesh::newGeometry(Mesh::SPHERE);
      else if (a[1] == "obj") {
        if (a.size() != 3) return bad("mesh NAME obj PATH");
        std::string e;
        m = Mesh::loadObj(a[2][0] == '/' ? a[2] : dir + "/" + a[2], &e);
        if (!m) return bad(e);
      } else if (a[1] == "icosphere") {
        float lv;
        if (a.size() != 4 || !meshes.count(a[2]) || !num(3, lv) || lv < 0.0f || lv > 10.0f || lv != (float)(int)lv)
          return bad("mesh NAME icosphere BASE LEVELS (0..10)");
        if (meshes[a[2]]->type != Mesh::MESH) return bad("icosphere base must be a triangle mesh");
        m = Mesh::icosphere(*meshes[a[2]], (int)lv);
      } else return bad("unknown mesh kind " + a[1]);
      if (!m) return bad("could not create mesh " + a[0]);
      meshes[a[0]] = s->own(m);
    } else if (kw == "object") {
      if (a.size() != 1 || !meshes.count(a[0])) return bad("object MESH (a mesh defined above)");
      cur = &s->add(meshes[a[0]]);
    } else {
      if (!cur) return bad(kw + " before the first object");
      float x, y, z, w;
      if (kw == "scale") {
        if (a.size() == 1 && num(0, x)) Object::scale(*cur, x);                             // Object.inl:81
        else if (a.size() == 3 && num(0, x) && num(1, y) && num(2, z)) Object::scale(*cur, mk3(x, y, z));   // :86
        else return bad("scale S | scale X Y Z");
      } else if (kw == "rotate") {
        if (a.size() != 4 || !num(0, w) || !num(1, x) || !num(2, y) || !num(3, z)) return bad("rotate DEGREES X Y Z");
        Object::rotate(*cur, angleAxis(w, mk3(x, y, z)));
      } else if (kw == "translate") {
        if (a.size() != 3 || !num(0, x) || !num(1, y) || !num(2, z)) return bad("translate X Y Z");
        Object::translate(*cur, mk3(x, y, z));
      } else if (kw == "color") {
        if (a.size() != 3 || !num(0, x) || !num(1, y) || !num(2, z)) return bad("color R G B");
        cur->material.color = mk3(x, y, z);
      } else if (kw == "emit" || kw == "brdf" || kw == "ior") {
        if (a.size() != 1 || !num(0, x)) return bad(kw + " VALUE");
        (kw == "emit" ? cur->material.emit : kw == "brdf" ? cur->material.brdf : cur->material.n) = x;
      } else if (kw == "material") {
        if (a.size() != 1) return bad("material diff|trans|mirr");
        if (a[0] == "diff") cur->material.type = Material::DIFF;
        else if (a[0] == "trans") cur->material.type = Material::TRANS;
        else if (a[0] == "mirr") cur->material.type = Material::MIRR;
        else return bad("material diff|trans|mirr");
      } else return bad("unknown statement " + kw);
    }
  }
  if (s->objs.empty()) { lineno = 0; return bad("no objects"); }
  if (cam_out) {
    cam.rotate(rot_x, rot_y);
    cam.update();
    cameraFrame(cam, cam_out);
  }
  *out = s;
  return TRC_OK;
}

void trc_scene_free(trc_scene* s) { delete s; }
uint32_t trc_scene_size(const trc_scene* s) { return s ? (uint32_t)s->objs.size() : 0; }
const void* trc_scene_objects(const trc_scene* s) { return s ? s->objs.data() : nullptr; }

int trc_scene_translate(trc_scene* s, uint32_t i, const float a[3]) {
  if (!s || !a || i >= s->objs.size()) return TRC_ERR_INVALID;
  Object::translate(s->objs[i], mk3(a[0], a[1], a[2])); return TRC_OK;
}
int trc_scene_rotate(trc_scene* s, uint32_t i, float angle_deg, const float ax[3]) {
  if (!s || !ax || i >= s->objs.size()) return TRC_ERR_INVALID;
  Object::rotate(s->objs[i], angleAxis(angle_deg, mk3(ax[0], ax[1], ax[2]))); return TRC_OK;
}
int trc_scene_scale(trc_scene* s, uint32_t i, const float a[3]) {
  if (!s || !a || i >= s->objs.size()) return TRC_ERR_INVALID;
  Object::scale(s->objs[i], mk3(a[0], a[1], a[2])); return TRC_OK;
}
int trc_scene_add_obj(trc_scene* s, const char* path, uint32_t* index) {
  if (!s || !path) return TRC_ERR_INVALID;
  std::string err;
  Mesh::Mesh* m = Mesh::loadObj(path, &err);
  if (!m) { fprintf(stderr, "tracer_amd: %s\n", err.c_str()); return TRC_ERR_INVALID; }
  s->add(s->own(m));
  if (index) *index = (uint32_t)s->objs.size() - 1;
  return TRC_OK;
}
int trc_scene_set_material(trc_scene* s, uint32_t i, const float color[3], float emit, float brdf, float n, int type) {
  if (!s || !color || i >= s->objs.size() || type < Material::DIFF || type > Material::MIRR) return TRC_ERR_INVALID;
  Material::Material& m = s->objs[i].material;
  m.color = mk3(color[0], color[1], color[2]); m.emit = emit; m.brdf = brdf; m.n = n; m.type = type;
  return TRC_OK;
}

int trc_camera_default(uint32_t aspect_w, uint32_t aspect_h, float rot_x, float rot_y, trc_camera* out) {
  if (!out || aspect_w == 0 || aspect_h == 0) return TRC_ERR_INVALID;
  ThirdPersonCamera cam;
  cam.setFOV(30.0f);                 // FOV, common.h:40 / main.cpp:275
  cam.setAspect(aspect_w, aspect_h); // main.cpp:276
  cam.zoom(-13.0f);                  // main.cpp:277
  cam.m_focalDist = 13.0f;           // main.cpp:278
  cam.rotate(rot_x, rot_y);
  cam.update();                      // main.cpp:279
  cameraFrame(cam, out);
  return TRC_OK;
}

int trc_write_ppm(const char* path, const uint32_t* ldr, uint32_t w, uint32_t h) {
  if (!path || !ldr || !w || !h) return TRC_ERR_INVALID;
  FILE* f = fopen(path, "wb");
  if (!f) return TRC_ERR_INVALID;
  fprintf(f, "P6\n%u %u\n255\n", w, h);
  std::vector<unsigned char> row(3ull * w);
  for (uint32_t y = 0; y < h; ++y) {            // buffer row 0 is the bottom of the image
    const uint32_t* src = ldr + (size_t)(h - 1 - y) * w;
    for (uint32_t x = 0; x < w; ++x) {
      row[3 * x] = (src[x] >> 16) & 255; row[3 * x + 1] = (src[x] >> 8) & 255; row[3 * x + 2] = src[x] & 255;
    }
    fwrite(row.data(), 1, row.size(), f);
  }
  fclose(f);
  return TRC_OK;
}

int trc_write_pfm(const char* path, const float* film, uint32_t w, uint32_t h, float iters) {
  if (!path || !film || !w || !h || !(iters > 0.0f)) return TRC_ERR_INVALID;
  FILE* f = fopen(path, "wb");
  if (!f) return TRC_ERR_INVALID;
  fprintf(f, "PF\n%u %u\n-1.0\n", w, h);        // little-endian, rows bottom-to-top like the film
  std::vector<float> row(3ull * w);
  for (uint32_t y = 0; y < h; ++y) {
    for (size_t i = 0; i < row.size(); ++i) row[i] = film[(size_t)y * 3 * w + i] / iters;
    fwrite(row.data(), sizeof(float), row.size(), f);
  }
  fclose(f);
  return TRC_OK;
}

} // extern "C"
