// tracer_amd/csrc/host/Renderer.h -- the reference's Renderer interface (src/Renderer.h:10-55)
// on top of the C ABI (include/tracer_abi.h), headless: the CUDA-GL PBO, result texture and
// full-screen quad (Renderer.cpp:48-92, :193-206) are replaced by host-readable buffers.
#pragma once
#include "../../../include/tracer_abi.h"
#include "Camera.h"
#include "SceneApi.h"
#include <cstdint>
#include <string>
#include <vector>

// Camera frame exactly as Renderer::render derives it (Renderer.cpp:156-171).
void cameraFrame(const Camera& camera, trc_camera* out);

class Renderer {
public:
  enum Mode { RAYTRACE, PATHTRACE };                          // Renderer.h:13

  Renderer(unsigned w, unsigned h);                          // Renderer.h:15
  ~Renderer();

  void init();                                               // allocates the GPU context (device 0)
  void init(int device);
  void initScene(Object::Object* scene, unsigned size);      // Renderer.cpp:112-151
  void render(const Camera& camera, float time);             // Renderer.cpp:153-207

  void updateScene(int index, const Object::Object& obj);    // Renderer.cpp:221-225
  void setMode(Mode m);                                      // Renderer.cpp:217-219
  void resetFilm();                                          // Renderer.cpp:209-211
  unsigned getIterations();                                  // Renderer.cpp:213-215

  // ---- additions for headless use --------------------------------------------------------
  // `count` path-tracing iterations in one call (one persistent launch instead of `count`
  // render() calls); equivalent to calling render() `count` times with the same `time`.
  void renderIterations(const Camera& camera, float time, unsigned count);
  // Picking, main.cpp:180-208 (there on the CPU copy of the scene; here against the uploaded one): the object
  // under window position (x, y) -- y from the top, as GLUT reports it -- of a windowW x windowH window, or -1.
  int pick(const Camera& camera, int x, int y, unsigned windowW, unsigned windowH);
  const uint32_t* ldr();       // w*h pixels, 0x00RRGGBB, row 0 = bottom of the image
  const float* film();         // 3*w*h floats, un-normalised radiance sums (divide by getIterations())
  bool ok() const { return m_error.empty(); }
  const std::string& error() const { return m_error; }
  trc_ctx* context() { return m_ctx; }
  unsigned width() const { return image_width; }
  unsigned height() const { return image_height; }

private:
  void check(int rc, const char* what);

  unsigned image_width, image_height;
  trc_ctx* m_ctx;
  unsigned sceneSize;
  unsigned filmIters;
  Mode mode;
  std::vector<uint32_t> m_ldr;
  std::vector<float> m_film;
  std::string m_error;
};
