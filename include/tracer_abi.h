/* include/tracer_abi.h -- C ABI of libtracer_amd.so, the MI355X (gfx950) replacement for the
 * per-bounce path-tracing loop of nopjia/tracer.
 *
 * What it replaces (paths relative to the reference checkout):
 *   src/raytrace.cu:189-238   extern "C" pathtrace() / raytrace1() kernel launchers
 *   src/raytrace.cu:29-187    the kernels behind them
 *   src/Renderer.cpp:94-151   device buffer ownership + deep-copied scene upload
 *   src/Renderer.cpp:173-206  CUDA-GL PBO mapping (replaced by a headless buffer)
 * The reference binds this path from C++ (Renderer.cpp:7-23 declares the two extern "C"
 * launchers and Renderer::render calls them, Renderer.cpp:177-191). INTEGRATION.md shows the
 * Renderer.cpp a maintainer would write on top of this header.
 *
 * Conventions
 *   - every call returns 0 on success or a non-zero trc_status; trc_last_error_string() gives
 *     the text. Nothing here calls exit() (the reference's checkCudaErrors does).
 *   - plain pointers and sizes only. `objects` is the caller's HOST array of Object::Object in
 *     the reference's own memory layout (Object.inl:24-29; LP64: 168 bytes, m_mesh -> Mesh::Mesh
 *     of Mesh.inl:34-41, 72 bytes). It is read during the call and not retained.
 *   - a context is bound to one GPU and one HIP stream; calls on one context are not
 *     re-entrant. Rendering calls are asynchronous on that stream; the trc_read_* calls and
 *     trc_synchronize() wait for it.
 *   - `iter` is the reference's 1-based filmIters (Renderer.cpp:183). The film is the
 *     un-normalised running sum of per-iteration radiance (raytrace.cu:170) and is zeroed
 *     when iteration 1 is rendered (raytrace.cu:103-104).
 */
#ifndef TRACER_ABI_H
#define TRACER_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trc_ctx trc_ctx;

typedef enum trc_status {
  TRC_OK = 0,
  TRC_ERR_INVALID = 1,   /* bad argument (null pointer, zero size, index out of range, ...) */
  TRC_ERR_HIP = 2,       /* a HIP runtime call failed; see trc_last_error_string() */
  TRC_ERR_NO_SCENE = 3,  /* render call before trc_upload_scene */
  TRC_ERR_UNSUPPORTED = 4
} trc_status;

/* Camera frame as Renderer::render computes it (Renderer.cpp:156-171) and passes it to the
 * launchers (raytrace.cu:192-193): position, screen-x / screen-y half extents A and B
 * (already scaled by tan(fov) and 1/aspect), view direction C, thin-lens parameters
 * (Camera.h:32-33). */
typedef struct trc_camera {
  float campos[3];
  float A[3];
  float B[3];
  float C[3];
  float lensRadius;
  float focalDist;
} trc_camera;

/* Per-bounce random triple (raytrace.cu:155-157):
 *  TRC_RNG_REFERENCE     rand[randidx] of the same iteration, randidx = (idx + (int)dot(dir,pos)) % N,
 *                        i.e. the reference's reuse of one triple per pixel per iteration;
 *  TRC_RNG_DECORRELATED  an independent triple per (pixel, iteration, depth).            */
typedef enum trc_rng_mode { TRC_RNG_REFERENCE = 0, TRC_RNG_DECORRELATED = 1 } trc_rng_mode;

/* How one pathtrace call is executed on the GPU (results are bit-identical across all three):
 *  TRC_PIPE_WAVEFRONT   one raygen launch, PATH_DEPTH bounce launches over per-pixel ray state
 *                       in HBM, one accumulate launch per iteration (the reference's structure,
 *                       raytrace.cu:211-225), dead paths skipped by flag;
 *  TRC_PIPE_COMPACTED   same, but live paths are compacted into a queue after every bounce by
 *                       wavefront ballot (README.md:56 "stream compaction");
 *  TRC_PIPE_FUSED       one launch per call: a lane owns a pixel, ray state stays in registers
 *                       across bounces and iterations, a finished lane regenerates that
 *                       pixel's next sample;
 *  TRC_PIPE_QUEUE       persistent wavefronts pull (pixel, iteration) path tasks from a global
 *                       work queue, so lanes never idle on short paths or cheap pixels; the
 *                       samples pass through a staging buffer in HBM and are summed per pixel
 *                       in iteration order by a second kernel;
 *  TRC_PIPE_AUTO       TRC_PIPE_QUEUE (TRC_PIPE_FUSED for calls of fewer than 2.4 M paths, e.g. one
 *                       interactive frame), except for a scene with a MESH of 1024 triangles or more:
 *                       then TRC_PIPE_COMPACTED, where the brute-force loop over that mesh's faces
 *                       (Ray.inl:172-177) runs as a pass of its own before every bounce, over a
 *                       dense queue of the rays that passed its AABB, with the triangle list
 *                       staged through LDS in tiles (default).                              */
typedef enum trc_pipeline { TRC_PIPE_WAVEFRONT = 0, TRC_PIPE_COMPACTED = 1, TRC_PIPE_FUSED = 2,
                            TRC_PIPE_QUEUE = 3, TRC_PIPE_AUTO = 4 } trc_pipeline;

typedef enum trc_option {
  TRC_OPT_RNG_MODE = 0,     /* trc_rng_mode, default TRC_RNG_REFERENCE */
  TRC_OPT_PIPELINE = 1,     /* trc_pipeline, default TRC_PIPE_AUTO */
  TRC_OPT_RECORD_HITS = 2,  /* 0/1: raytrace1 also stores id/t/pos/nor per pixel (default 1) */
  TRC_OPT_STAGE_MB = 3,     /* TRC_PIPE_QUEUE: staging buffer budget in MiB (default 2048); a call is
                               split into chunks of iterations that fit */
  TRC_OPT_MESH_ACCEL = 4,   /* big meshes in the launch-per-bounce pipelines and mode 1: 0 = every face against every
                               ray that passed the AABB, as Ray.inl:172-177 does (default; for meshes of small faces a
                               conservative bounding-sphere test per pair skips the arithmetic of pairs that cannot
                               hit), 1 = bounding-volume hierarchy in front of the same exact triangle test
                               (same hits, same tie rule), 2 = as 0 without the bounding-sphere
                               test (validation) */
  TRC_OPT_SKIP_TONEMAP = 5  /* 0/1: trc_pathtrace / trc_pathtrace_static leave the LDR image alone (default 0). For the
                               ranks of a multi-GPU render: only the rank that holds the reduced film needs the display
                               transform, and it runs it itself with the total iteration count (trc_tonemap) */
} trc_option;

/* Device-side event counts since the last trc_reset_counters (units of SURVEY.md 8d). */
typedef struct trc_counters {
  uint64_t live_segments;   /* closest-hit evaluations for live paths (raytrace.cu:119-127) */
  uint64_t paths;           /* camera rays generated (pixels x iterations) */
  uint64_t tri_tests;       /* ray-triangle tests executed (Ray.inl:173) */
  /* persistent path kernels (TRC_PIPE_QUEUE, TRC_PIPE_FUSED): sum over their wavefronts of the
   * shader-clock cycles and of the constant 100 MHz reference ticks between a wavefront's first and
   * last instruction; 100 * wave_cycles / wave_ref_ticks = the shader clock in MHz the kernel ran at */
  uint64_t wave_cycles;
  uint64_t wave_ref_ticks;
} trc_counters;

/* ---- life cycle (replaces Renderer::initCUDAMemory/freeCUDAMemory, Renderer.cpp:94-110) -- */
/* `stream`: a hipStream_t to launch on, or NULL to create a private one. */
int trc_create(int device, uint32_t width, uint32_t height, void* stream, trc_ctx** out);
int trc_destroy(trc_ctx* ctx);
int trc_set_option(trc_ctx* ctx, int option, int value);
/* Image-tile partition (no counterpart in the reference, which renders every pixel on one GPU): from now on
 * trc_raytrace1 / trc_pathtrace render only image rows [row0, row0 + rows) and leave the rest of the film,
 * LDR and hit buffers untouched; rows == 0 returns to the whole image. The RNG is keyed by the absolute
 * pixel, so a tile holds exactly the values the full frame has there and disjoint tiles rendered on
 * different GPUs over zeroed films combine with the same single sum-reduce as the sample partition. */
int trc_set_tile(trc_ctx* ctx, uint32_t row0, uint32_t rows);

/* ---- scene (replaces Renderer::initScene / updateScene, Renderer.cpp:112-151, :221-225) -- */
int trc_upload_scene(trc_ctx* ctx, const void* objects, uint32_t count);
/* matrix / inverse as 16 floats each, column-major (glm::mat4 memory order). */
int trc_update_object(trc_ctx* ctx, uint32_t index, const float matrix[16], const float matrixi[16]);

/* ---- rendering (replaces raytrace1 / pathtrace, raytrace.cu:189-238) -------------------- */
/* mode 1: one closest-hit + flat shade per pixel into the LDR buffer (raytrace.cu:29-62). */
int trc_raytrace1(trc_ctx* ctx, const trc_camera* cam);
/* mode 2: iterations iter_first, iter_first + iter_stride, ... (n_iters of them) accumulated
 * into the film; the LDR buffer then holds rgbToInt(pow(film / last_iter, 1/2.2)) as after the
 * reference's last accumColorKernel (raytrace.cu:173). seed keys the RNG (rng.h).          */
int trc_pathtrace(trc_ctx* ctx, const trc_camera* cam, uint64_t seed,
                  uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters);
/* trc_pathtrace for callers that cannot allocate. NO COUNTERPART IN THE REFERENCE (whose launcher owns no memory
 * at all, raytrace.cu:189-226): it is for code under HIP graph capture, tight serving loops and harnesses that
 * forbid hipMalloc. Runs the work-queue pipeline (TRC_PIPE_QUEUE) with its scratch -- the task counter and the
 * 12-bytes-per-path sample staging -- in a static device array of 64 MiB + 256 B that every GPU the library runs on
 * holds until the process ends; the call itself performs no hipMalloc / hipFree / allocating library call. The film
 * and LDR image are those of the context (or trc_bind_film), bit-identical to trc_pathtrace's.
 * Limits: one iteration of the rendered tile must fit the staging area, i.e. at most 5 592 405 pixels per
 * trc_set_tile band (TRC_ERR_UNSUPPORTED beyond; a 3840x2160 frame takes two bands); more iterations than fit
 * are rendered in chunks.
 * Sharing: all contexts on one GPU share that GPU's array. Calls are ordered against each other ON THE GPU by a
 * per-device event -- each call's first launch waits for the previous call's last one, whatever context and stream
 * that ran on -- so contexts with private streams may call back to back without a host synchronisation in between
 * (tests/test_zz_no_alloc.py). Calls from several host threads are serialised while they enqueue.
 * Graph capture: a call made while the context's stream is capturing (hipStreamBeginCapture) records memset and
 * kernel nodes only -- no timing events (trc_last_*_ms then report "nothing rendered") and no guard event: replays of
 * such a graph must be ordered by the caller against any other use of this entry point on the same GPU. */
int trc_pathtrace_static(trc_ctx* ctx, const trc_camera* cam, uint64_t seed,
                         uint32_t iter_first, uint32_t iter_stride, uint32_t n_iters);

/* ---- ray queries (the interactive path's picking, main.cpp:180-208) ------------------------ */
/* Ray::intersectScene (Ray.inl:66-84) for n caller rays against the uploaded scene -- what main.cpp:202 does on
 * the CPU for the picking ray. rays6[i] = {pos.xyz, dir.xyz}, the direction used as given (intersectScene does
 * not normalise it either). Per ray: Hit::m_id (-1 = no hit), m_t (object-space parameter, -1 on a miss), m_pos,
 * m_nor (zero on a miss). HOST arrays; any output may be NULL; returns after the results have arrived. */
int trc_intersect_rays(trc_ctx* ctx, const float* rays6, uint32_t n,
                       int32_t* ids, float* ts, float* pos3, float* nor3);
/* Same with DEVICE arrays, asynchronous on the context's stream: no copies and no waiting -- except that a scene with
 * a MESH of 1024 faces or more keeps a key table and a ray queue of n entries per such mesh, (re)allocated by the first
 * call and by any call with a larger n than every call before it (hipFree / hipMalloc: a device-wide wait). Query rays
 * add their triangle tests to trc_counters.tri_tests and nothing to live_segments / paths, which stay what was rendered. */
int trc_intersect_rays_device(trc_ctx* ctx, const void* rays6_d, uint32_t n,
                              void* ids_d, void* ts_d, void* pos3_d, void* nor3_d);
/* main.cpp:180-208 in one call: the pixel-corner ray through window position (x, y) of a window_w x window_h
 * window -- y counted from the TOP as GLUT reports it, uv = (x / window_w, (window_h - y) / window_h), no jitter --
 * then intersectScene. *id = the clicked object (main.cpp:203-206: valid when >= 0). Outputs may be NULL. */
int trc_pick(trc_ctx* ctx, const trc_camera* cam, uint32_t window_w, uint32_t window_h, int32_t x, int32_t y,
             int32_t* id, float* t, float* pos3, float* nor3);

/* Renderer::resetFilm (Renderer.cpp:209-211); also zeroes the film buffer itself so that a
 * rank that never renders iteration 1 starts from zero. */
int trc_reset_film(trc_ctx* ctx);
/* LDR buffer = rgbToInt(pow(film / iters, 1/2.2)) (raytrace.cu:173), e.g. after a multi-GPU sum. */
int trc_tonemap(trc_ctx* ctx, float iters);

/* ---- headless output (replaces the PBO -> texture path, Renderer.cpp:193-206) ----------- */
int trc_read_film(trc_ctx* ctx, float* rgb /* 3*w*h */);
int trc_write_film(trc_ctx* ctx, const float* rgb /* 3*w*h */);   /* resume from a checkpoint */
int trc_read_ldr(trc_ctx* ctx, uint32_t* bgra /* w*h, 0x00RRGGBB as rgbToInt packs it */);
/* validation outputs of the last trc_raytrace1; any pointer may be NULL. */
int trc_read_hits(trc_ctx* ctx, int32_t* ids, float* ts, float* pos3, float* nor3);
int trc_synchronize(trc_ctx* ctx);

/* The film as a device pointer (3*w*h floats), for a sum-reduce over RCCL by the caller. */
int trc_film_device_ptr(trc_ctx* ctx, void** dev_ptr);
/* Use caller-owned device memory (3*w*h floats, 16-byte aligned) as the film from now on;
 * NULL returns to the context's own buffer. */
int trc_bind_film(trc_ctx* ctx, void* dev_ptr);

/* ---- measurement ------------------------------------------------------------------------ */
int trc_get_counters(trc_ctx* ctx, trc_counters* out);   /* waits for the stream */
int trc_reset_counters(trc_ctx* ctx);
/* GPU time of the last trc_raytrace1 / trc_pathtrace call between HIP events recorded on the
 * context's stream around its launches (waits for it). */
int trc_last_render_ms(trc_ctx* ctx, float* ms);
/* Same, up to the end of the path kernels only (excludes the display transform): for
 * TRC_PIPE_FUSED this is the duration of the single persistent launch. */
int trc_last_kernel_ms(trc_ctx* ctx, float* ms);

const char* trc_last_error_string(void);
const char* trc_version(void);

/* ---- the reference's own launcher signatures (raytrace.cu:189-199, :228-232) ------------- */
/* Same argument lists as the reference; glm::vec3 references are `const float*` at the ABI.
 * All buffers are DEVICE pointers owned by the caller exactly as Renderer.cpp allocates them
 * (Renderer.cpp:94-100), in the reference's layouts; `scene_d` is the device copy of Object[]
 * produced by Renderer::initScene (Renderer.cpp:112-151). `time` seeds the RNG as
 * (uint64)(time*100.0f) like curandSetPseudoRandomGeneratorSeed (raytrace.cu:205);
 * `rand_d` is accepted and left untouched (the RNG is counter-based).
 * The library keeps a flattened copy of each `scene_d` it has seen and re-validates it on every call with a
 * content digest computed on the GPU over the objects, their Mesh headers and the mesh arrays (one small kernel,
 * 8 bytes read back): Renderer::updateScene's in-place edits (Renderer.cpp:221-225), a mesh freed and replaced at
 * the same address, vertices rewritten in place are all picked up by the next frame.                        */
void pathtrace(uint32_t* pbo_out, uint32_t w, uint32_t h, float time,
               const float* campos, const float* A, const float* B, const float* C,
               float lensRadius, float focalDist,
               const void* scene_d, uint32_t sceneSize,
               void* rand_d, void* rays_d, void* col_d, int* idx_d,
               void* film_d, uint32_t filmIters);
void raytrace1(uint32_t* pbo_out, uint32_t w, uint32_t h, float time,
               const float* campos, const float* A, const float* B, const float* C,
               const void* scene_d, uint32_t sceneSize);

/* Housekeeping for the two launchers above (no counterpart in the reference, which never frees its scene,
 * Renderer.cpp:109): forget the copy kept for `scene_d` (NULL: for every scene) so that the next call reads it
 * back unconditionally; free everything the launchers have cached (call before unloading the library -- nothing
 * is freed behind the caller's back at exit). */
int trc_legacy_invalidate(const void* scene_d);
int trc_legacy_release(void);

#ifdef __cplusplus
}
#endif
#endif /* TRACER_ABI_H */
