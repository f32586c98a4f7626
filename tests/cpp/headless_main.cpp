// tests/cpp/headless_main.cpp -- what the reference's main.cpp does, minus the window: build the
// Cornell scene through the Object/Mesh API, hand it to a Renderer, call render(camera, timer)
// once per "frame" with timer += DELTA_T, and dump the film + LDR image. Exercises the C++
// host-side mirror (tracer_amd/csrc/host/) exactly as a user of the reference would.
//
//   headless_main <data_dir> <w> <h> <frames> <out_prefix> [mode: 1|2] [batch]
//
// Writes <out_prefix>.film (raw fp32 RGB, w*h*3), <out_prefix>.ldr (raw u32, w*h),
// <out_prefix>.ppm and prints the iteration count.
#include "../../tracer_amd/csrc/host/Renderer.h"
#include "../../include/tracer_host.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using trc::mk3; using trc::splat3; using trc::angleAxis;

static std::vector<Object::Object> scene;
static ThirdPersonCamera camera;

static bool initScene(const std::string& dataDir, unsigned w, unsigned h) {   // cf. main.cpp:274-396
  camera.setFOV(30.0f);
  camera.setAspect(w, h);
  camera.zoom(-13.0f);
  camera.m_focalDist = 13.0f;
  camera.update();

  const trc::f3 BOX_HDIM = splat3(5.0f);
  Object::Object* obj;
  Mesh::Mesh* planemesh = Mesh::newGeometry(Mesh::PLANE);

  obj = Object::newObject(planemesh);                                   // bottom
  Object::scale(*obj, BOX_HDIM * 2.0f);
  Object::translate(*obj, mk3(0.0f, -BOX_HDIM.y, 0.0f));
  obj->material.color = splat3(1.0f);
  scene.push_back(*obj);
  obj = Object::newObject(planemesh);                                   // top
  Object::rotate(*obj, angleAxis(180.0f, mk3(1.0f, 0.0f, 0.0f)));
  Object::scale(*obj, BOX_HDIM * 2.0f);
  Object::translate(*obj, mk3(0.0f, BOX_HDIM.y, 0.0f));
  obj->material.color = splat3(1.0f);
  scene.push_back(*obj);
  obj = Object::newObject(planemesh);                                   // back
  Object::rotate(*obj, angleAxis(90.0f, mk3(1.0f, 0.0f, 0.0f)));
  Object::scale(*obj, BOX_HDIM * 2.0f);
  Object::translate(*obj, mk3(0.0f, 0.0f, -BOX_HDIM.z));
  obj->material.color = splat3(1.0f);
  scene.push_back(*obj);
  obj = Object::newObject(planemesh);                                   // right
  Object::rotate(*obj, angleAxis(90.0f, mk3(0.0f, 0.0f, 1.0f)));
  Object::scale(*obj, BOX_HDIM * 2.0f);
  Object::translate(*obj, mk3(BOX_HDIM.x, 0.0f, 0.0f));
  obj->material.color = mk3(0.5f, 0.5f, 1.0f);
  scene.push_back(*obj);
  obj = Object::newObject(planemesh);                                   // left
  Object::rotate(*obj, angleAxis(-90.0f, mk3(0.0f, 0.0f, 1.0f)));
  Object::scale(*obj, BOX_HDIM * 2.0f);
  Object::translate(*obj, mk3(-BOX_HDIM.x, 0.0f, 0.0f));
  obj->material.color = mk3(1.0f, 0.5f, 0.5f);
  scene.push_back(*obj);

  obj = Object::newObject(Mesh::newGeometry(Mesh::CUBE));               // ceiling light
  Object::scale(*obj, mk3(BOX_HDIM.x, 0.5f, BOX_HDIM.z));
  Object::translate(*obj, mk3(0.0f, BOX_HDIM.y, 0.0f));
  obj->material.color = splat3(1.0f);
  obj->material.emit = 2.0f;
  scene.push_back(*obj);

  obj = Object::newObject(Mesh::newGeometry(Mesh::SPHERE));             // mirror sphere
  Object::scale(*obj, 4.0f);
  Object::translate(*obj, mk3(-2.5f, -3.0f, -3.0f));
  obj->material.color = splat3(1.0f);
  obj->material.type = Material::MIRR;
  scene.push_back(*obj);
  obj = Object::newObject(Mesh::newGeometry(Mesh::SPHERE));             // glass sphere
  Object::scale(*obj, 4.0f);
  Object::translate(*obj, mk3(2.5f, -3.0f, 0.0f));
  obj->material.color = splat3(1.0f);
  obj->material.type = Material::TRANS;
  obj->material.n = 1.6f;
  scene.push_back(*obj);

  std::string err;
  Mesh::Mesh* ico = Mesh::loadObj(dataDir + "/icosahedron.obj", &err);
  if (!ico) { fprintf(stderr, "%s\n", err.c_str()); return false; }
  obj = Object::newObject(ico);
  Object::scale(*obj, 1.5f);
  Object::rotate(*obj, angleAxis(-10.0f, mk3(1.0f, 0.0f, 0.0f)));
  Object::rotate(*obj, angleAxis(25.0f, mk3(0.0f, 1.0f, 0.0f)));
  Object::translate(*obj, mk3(-1.0f, -3.7f, 3.0f));
  obj->material.color = mk3(1.0f, 1.0f, 0.0f);
  scene.push_back(*obj);
  return true;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s data_dir w h frames out_prefix [mode] [batch]\n", argv[0]); return 2; }
  const std::string dataDir = argv[1];
  const unsigned w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]);
  const std::string out = argv[5];
  const int mode = argc > 6 ? atoi(argv[6]) : 2;
  const unsigned batch = argc > 7 ? atoi(argv[7]) : 0;

  if (!initScene(dataDir, w, h)) return 1;
  Renderer renderer(w, h);
  renderer.init();
  renderer.initScene(scene.data(), (unsigned)scene.size());
  renderer.setMode(mode == 1 ? Renderer::RAYTRACE : Renderer::PATHTRACE);

  float timer = 0.0f;
  if (batch) {                       // one persistent launch for all iterations
    timer += 0.01f;
    renderer.renderIterations(camera, timer, frames);
  } else {
    for (unsigned f = 0; f < frames; ++f) {   // main.cpp:125-131 draw(): timer += DELTA_T; render
      timer += 0.01f;
      renderer.render(camera, timer);
    }
  }
  if (!renderer.ok()) { fprintf(stderr, "render failed: %s\n", renderer.error().c_str()); return 1; }

  const float* film = renderer.film();
  const uint32_t* ldr = renderer.ldr();
  FILE* f = fopen((out + ".film").c_str(), "wb"); fwrite(film, sizeof(float), 3ull * w * h, f); fclose(f);
  f = fopen((out + ".ldr").c_str(), "wb"); fwrite(ldr, sizeof(uint32_t), (size_t)w * h, f); fclose(f);
  trc_write_ppm((out + ".ppm").c_str(), ldr, w, h);
  printf("iterations %u\n", renderer.getIterations());
  // main.cpp:180-208: what is under the middle of the window, and under its top-left corner region
  printf("pick centre %d, pick (w/4, h/8) %d\n", renderer.pick(camera, (int)w / 2, (int)h / 2, w, h),
         renderer.pick(camera, (int)w / 4, (int)h / 8, w, h));
  return renderer.ok() ? 0 : 1;
}
