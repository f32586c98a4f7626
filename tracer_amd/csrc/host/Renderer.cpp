// tracer_amd/csrc/host/Renderer.cpp -- see Renderer.h.
#include "Renderer.h"
#include <cstdio>

#pragma clang fp contract(off)

using namespace trc;

// Renderer.cpp:156-171. `fov` is used as a half-angle; B is scaled by 1/aspect.
void cameraFrame(const Camera& camera, trc_camera* out) {
  const f3 C = normalize(camera.getLookAt() - camera.getPosition());             // :159
  f3 A = normalize(cross(C, camera.getUp()));                                    // :164
  f3 B = 1.0f / camera.getAspect() * normalize(cross(A, C));                     // :165
  const float tanFOV = tanf(radians(camera.getFOV()));                           // :168
  A = A * tanFOV;                                                                // :169
  B = B * tanFOV;                                                                // :170
  const f3 p = camera.getPosition();
  out->campos[0] = p.x; out->campos[1] = p.y; out->campos[2] = p.z;
  out->A[0] = A.x; out->A[1] = A.y; out->A[2] = A.z;
  out->B[0] = B.x; out->B[1] = B.y; out->B[2] = B.z;
  out->C[0] = C.x; out->C[1] = C.y; out->C[2] = C.z;
  out->lensRadius = camera.m_lensRadius;
  out->focalDist = camera.m_focalDist;
}

Renderer::Renderer(unsigned w, unsigned h)
  : image_width(w), image_height(h), m_ctx(nullptr), sceneSize(0), filmIters(0), mode(RAYTRACE) {}

Renderer::~Renderer() { if (m_ctx) trc_destroy(m_ctx); }

void Renderer::check(int rc, const char* what) {
  if (rc != TRC_OK) {
    m_error = std::string(what) + ": " + trc_last_error_string();
    fprintf(stderr, "Renderer: %s\n", m_error.c_str());   // reported, never exit() (cf. checkCudaErrors)
  }
}

void Renderer::init() { init(0); }
void Renderer::init(int device) {
  if (m_ctx) return;
  check(trc_create(device, image_width, image_height, nullptr, &m_ctx), "trc_create");
}

void Renderer::initScene(Object::Object* scene, unsigned size) {
  if (!m_ctx) init();
  if (!m_ctx) return;
  sceneSize = size;
  check(trc_upload_scene(m_ctx, scene, size), "trc_upload_scene");
}

void Renderer::render(const Camera& camera, float time) {
  if (mode == PATHTRACE) { renderIterations(camera, time, 1); return; }
  if (!m_ctx) return;
  trc_camera cam; cameraFrame(camera, &cam);
  check(trc_raytrace1(m_ctx, &cam), "trc_raytrace1");                            // Renderer.cpp:177-181
}

void Renderer::renderIterations(const Camera& camera, float time, unsigned count) {
  if (!m_ctx || count == 0) return;
  trc_camera cam; cameraFrame(camera, &cam);
  const uint64_t seed = (uint64_t)(time * 100.0f);                               // raytrace.cu:205
  check(trc_pathtrace(m_ctx, &cam, seed, filmIters + 1, 1, count), "trc_pathtrace"); // ++filmIters, :183
  filmIters += count;
}

void Renderer::resetFilm() { filmIters = 0; }      // the next iteration is number 1 and zeroes the film
unsigned Renderer::getIterations() { return filmIters; }
void Renderer::setMode(Mode m) { mode = m; }

void Renderer::updateScene(int index, const Object::Object& obj) {
  if (!m_ctx) return;
  check(trc_update_object(m_ctx, (uint32_t)index, obj.matrix, obj.matrixi), "trc_update_object");
}

int Renderer::pick(const Camera& camera, int x, int y, unsigned windowW, unsigned windowH) {
  if (!m_ctx) return -1;
  trc_camera cam; cameraFrame(camera, &cam);                                    // main.cpp:187-194
  int32_t id = -1;
  check(trc_pick(m_ctx, &cam, windowW, windowH, x, y, &id, nullptr, nullptr, nullptr), "trc_pick");
  return id;
}

const uint32_t* Renderer::ldr() {
  m_ldr.resize((size_t)image_width * image_height);
  if (m_ctx) check(trc_read_ldr(m_ctx, m_ldr.data()), "trc_read_ldr");
  return m_ldr.data();
}

const float* Renderer::film() {
  m_film.resize(3 * (size_t)image_width * image_height);
  if (m_ctx) check(trc_read_film(m_ctx, m_film.data()), "trc_read_film");
  return m_film.data();
}
