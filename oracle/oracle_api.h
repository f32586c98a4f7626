/* oracle/oracle_api.h -- TEST INFRASTRUCTURE. C API shared by the two CPU checkers:
 *
 *   oracle/_ref/libtracer_ref.so   the reference's own Ray/Object/Material/Mesh/Utils .inl +
 *                                  Camera.cpp + vendored GLM compiled in place with g++
 *                                  (oracle/ref_driver.cpp, built only where /root/reference exists)
 *   oracle/libtracer_oracle.so     plain-C restatement (oracle/tracer_oracle.c), GLM-free
 *
 * Both export exactly these symbols so that tests can run the same call against either.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these.
 *
 * Conventions: vec3 = 3 packed floats; mat4 = 16 floats column-major (GLM order, m[col][row]);
 * `iter` is the reference's 1-based filmIters (Renderer.cpp:183); film is the un-normalised
 * running sum of per-iteration radiance (raytrace.cu:170).
 */
#ifndef TRC_ORACLE_API_H
#define TRC_ORACLE_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_camera {
  float campos[3];
  float A[3];       /* screen-x half extent  (Renderer.cpp:164,169) */
  float B[3];       /* screen-y half extent  (Renderer.cpp:165,170) */
  float C[3];       /* view direction        (Renderer.cpp:159)     */
  float lensRadius; /* Camera.h:32 */
  float focalDist;  /* Camera.h:33 */
} orc_camera;

typedef struct orc_stats {
  uint64_t live_segments; /* threads of calcColorKernel passing raytrace.cu:119-124 */
  uint64_t miss;          /* raytrace.cu:130 */
  uint64_t light;         /* raytrace.cu:137 */
  uint64_t depthcap;      /* raytrace.cu:144 */
  uint64_t diff, trans, mirr; /* bounced, by Material::Type */
  uint64_t paths;         /* pixels x iterations */
} orc_stats;

/* RNG modes for the per-bounce random triple:
 *  0 = reference-faithful: rand[randidx] of the same iteration, randidx per raytrace.cu:155
 *  1 = decorrelated: rand3(seed, pixel, iter, dim = 1 + depth)                            */
enum { ORC_RNG_REF = 0, ORC_RNG_DECORRELATED = 1 };

/* math modes for sin/cos/acos/pow:
 *  0 = host libm (what the reference's host build calls through glm::sin etc.)
 *  1 = portable polynomials (oracle/pmath.h), the algorithm the HIP kernels implement.
 * The reference-linked build supports mode 0 only.                                      */
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };
int  orc_set_math_mode(int mode); /* returns 0 if supported */
const char* orc_name(void);       /* "reference-host-build" or "c-restatement" */

/* ---- scenes ---------------------------------------------------------------------- */
/* variant 0: main.cpp:274-396 as shipped.  variant 1: same, object 8 (icosahedron) made
 * TRANS with m_n = 1.6 (BASELINE.json configs[2]).
 * variant 2: walls + light + one MESH icosphere of `param` subdivision levels (configs[4]). */
void* orc_scene_create(int variant, int param, const char* data_dir);
/* Wrap an Object[] (reference host layout) built by someone else; records copied, meshes shared. */
void* orc_scene_wrap(const void* objects, unsigned n);
void  orc_scene_free(void* scene);
unsigned orc_scene_size(const void* scene);
/* pointer to the Object[] array in the reference's host memory layout (Object.inl:24-29,
 * 168 bytes per object on LP64, m_mesh -> Mesh.inl:34-41); valid until orc_scene_free. */
const void* orc_scene_objects(const void* scene);
/* flat export of one object (for fixtures). faces6 = numFaces x {v0,v1,v2,n0,n1,n2}. */
void orc_scene_object_info(const void* scene, unsigned i, float matrix[16], float matrixi[16],
                           int* mesh_type, unsigned counts[3] /*verts,norms,faces*/,
                           float bmin[3], float bmax[3],
                           float color[3], float emit_brdf_n[3], int* mat_type);
void orc_scene_object_mesh(const void* scene, unsigned i, float* verts, float* norms, unsigned* faces6);
/* Object::translate / rotate / scale on object i (Object.inl:56-89), recomputing the inverse. */
void orc_scene_translate(void* scene, unsigned i, const float amount[3]);
void orc_scene_rotate(void* scene, unsigned i, float angle_deg, const float axis[3]);
void orc_scene_scale(void* scene, unsigned i, const float amount[3]);

/* ---- camera ---------------------------------------------------------------------- */
/* ThirdPersonCamera as main.cpp:275-279 sets it up (FOV 30, zoom(-13), focalDist 13,
 * lensRadius 1 from Camera.cpp:17) with aspect = aspect_w/aspect_h, after `rot_x`,`rot_y`
 * radians of Camera::rotate; then the A/B/C basis of Renderer.cpp:156-171.              */
void orc_camera_default(unsigned aspect_w, unsigned aspect_h, float rot_x, float rot_y, orc_camera* out);

/* ---- kernels (restated launch bodies around the intersect/shade library) ---------- */
/* raytraceKernel, raytrace.cu:29-62. Any output pointer may be NULL. */
void orc_raytrace1(const void* scene, unsigned w, unsigned h, const orc_camera* cam,
                   uint32_t* pbo, int* ids, float* ts, float* pos3, float* nor3, int nthreads);

/* pathtrace(), raytrace.cu:190-226, for iterations iter_first, iter_first+stride, ...
 * (n_iters of them). film (3*w*h) is read-modify-written; it is zeroed first when the
 * iteration number is 1 (raytrace.cu:103-104). pbo (may be NULL) receives the LDR image of
 * the last iteration rendered, normalised by that iteration number (raytrace.cu:173).    */
void orc_pathtrace(const void* scene, unsigned w, unsigned h, const orc_camera* cam,
                   uint64_t seed, unsigned iter_first, unsigned iter_stride, unsigned n_iters,
                   int rng_mode, float* film, uint32_t* pbo, orc_stats* stats, int nthreads);

/* accumColorKernel's display transform alone: pbo = rgbToInt(pow(film/iters, 1/2.2)). */
void orc_tonemap(const float* film, unsigned n, float iters, uint32_t* pbo);

/* ---- unit-level entry points ---------------------------- */
void  orc_intersect_scene(const void* scene, const float ray6[6], float* t, int* id, float pos[3], float nor[3]);
void  orc_intersect_object(const void* scene, unsigned obj, const float ray6[6], float* t, float pos[3], float nor[3]);
/* Picking, main.cpp:180-208: the ray through window position (x, y) as GLUT reports it (y from the top:
 * uv = (x / window_w, (window_h - y) / window_h), pixel corner, no jitter), then Ray::intersectScene.
 * `cam` carries the A/B/C basis that main.cpp:187-194 recomputes with the formulas of Renderer.cpp:159-170. */
void  orc_pick(const void* scene, const orc_camera* cam, int window_w, int window_h, int x, int y,
               float ray6[6], float* t, int* id, float pos[3], float nor[3]);
void  orc_bounce(int mat_type, float mat_n, const float rd[3], const float nor[3], const float rnd[3], float out[3]);
float orc_reflectance(const float inc[3], const float nor[3], float n1, float n2);
void  orc_rand_vector(float r1, float r2, float out[3]);
void  orc_rand_vector_hem(float r1, float r2, const float nor[3], float out[3]);
void  orc_rand_point_disk(float r1, float r2, float r3, float out[2]);
void  orc_rand3_api(uint64_t seed, uint32_t index, uint32_t iter, uint32_t dim, float out[3]);
void  orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t orc_rgb_to_int(float r, float g, float b);

/* C restatement only: exit stages of Ray.inl:189-226 over the (ray, face) pairs of a mode-1 frame (every `step`-th pixel in
 * x and y) against MESH object `obj`: counts[0..3] = pairs leaving at :196, :203, :208, and running the whole test;
 * counts[4] = rays past the AABB, counts[5] = rays tried (tools/mesh_exit_rates.py -> the flop weight of configs[4]). */
void  orc_mesh_exit_histogram(const void* scene, unsigned w, unsigned h, const orc_camera* cam, unsigned obj, unsigned step,
                              uint64_t counts[6], int nthreads);

/* portable-math unit entry points (oracle/pmath.h); C restatement only */
void  orc_pm_sincos(float x, float* s, float* c);
float orc_pm_acos(float x);
float orc_pm_pow(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
