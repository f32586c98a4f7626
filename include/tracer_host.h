/* include/tracer_host.h -- C entry points to the host-side scene description of
 * libtracer_amd.so, for callers that are not C++ (tests, bench.py). C++ callers use the
 * classes in tracer_amd/csrc/host/ (Camera, Renderer, namespaces Object/Mesh/Material), which
 * mirror reference src/Camera.h, src/Renderer.h, src/Object.inl, src/Mesh.h.
 *
 * Nothing here touches the GPU.
 */
#ifndef TRACER_HOST_H
#define TRACER_HOST_H

#include <stdint.h>
#include "tracer_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trc_scene trc_scene;

/* The reference's hard-coded scene (src/main.cpp:274-396) built through the Object/Mesh API:
 *   variant 0  as shipped: 5 walls, ceiling light, mirror sphere, glass sphere, icosahedron
 *   variant 1  same with the icosahedron made glass (TRANS, n = 1.6)   [BASELINE.json configs[2]]
 *   variant 2  walls + light + one MESH icosphere of `param` subdivision levels, scale 3 at
 *              (0,-2,0)                                                  [BASELINE.json configs[4]]
 * data_dir holds icosahedron.obj (the reference's data/icosahedron.obj). */
int trc_scene_cornell(int variant, int param, const char* data_dir, trc_scene** out);
/* Scene description file (the reference compiles its scene in, main.cpp:274-396). One
 * statement per line, `#` comments; transforms are applied in the order written through the same
 * Object::scale / rotate / translate calls main.cpp makes (tracer_amd/data/cornell.scene lists exactly
 * those calls and yields the same Object[] bit for bit):
 *   camera [fov F] [zoom Z] [focal D] [lens R] [rotate RX RY]
 *   mesh NAME plane|cube|sphere | mesh NAME obj PATH | mesh NAME icosphere BASE LEVELS
 *   object MESH   followed by   scale S | scale X Y Z | rotate DEGREES X Y Z | translate X Y Z |
 *                               color R G B | emit E | brdf B | ior N | material diff|trans|mirr
 * cam_out (may be NULL) receives the camera frame for an aspect_w x aspect_h image. On failure returns
 * TRC_ERR_INVALID with "file:line: reason" in err (may be NULL). */
int trc_scene_load(const char* path, uint32_t aspect_w, uint32_t aspect_h, trc_scene** out, trc_camera* cam_out,
                   char* err, uint32_t err_len);
void trc_scene_free(trc_scene* scene);
uint32_t trc_scene_size(const trc_scene* scene);
/* Object::Object[] in the reference's host layout; pass to trc_upload_scene. */
const void* trc_scene_objects(const trc_scene* scene);
/* Object::translate / rotate (angleAxis, degrees) / scale on object i (Object.inl:56-89). */
int trc_scene_translate(trc_scene* scene, uint32_t i, const float amount[3]);
int trc_scene_rotate(trc_scene* scene, uint32_t i, float angle_deg, const float axis[3]);
int trc_scene_scale(trc_scene* scene, uint32_t i, const float amount[3]);
/* Load an .obj (Mesh.h:20-126 dialect and common variants) and append it as a new object with
 * default material; returns its index in *index. */
int trc_scene_add_obj(trc_scene* scene, const char* obj_path, uint32_t* index);
int trc_scene_set_material(trc_scene* scene, uint32_t i, const float color[3], float emit, float brdf,
                           float n, int type);

/* ThirdPersonCamera as main.cpp:275-279 sets it up (FOV 30, zoom(-13), focal distance 13, lens
 * radius 1), aspect = aspect_w/aspect_h, orbited by (rot_x, rot_y) radians, turned into the
 * launcher's camera frame as Renderer::render does (Renderer.cpp:156-171). */
int trc_camera_default(uint32_t aspect_w, uint32_t aspect_h, float rot_x, float rot_y, trc_camera* out);

/* Headless image output (replaces the GL texture + quad, Renderer.cpp:195-206): binary PPM of
 * an LDR buffer (0x00RRGGBB, row 0 = bottom) and PFM of a film divided by `iters`. */
int trc_write_ppm(const char* path, const uint32_t* ldr, uint32_t w, uint32_t h);
int trc_write_pfm(const char* path, const float* film, uint32_t w, uint32_t h, float iters);

#ifdef __cplusplus
}
#endif
#endif
