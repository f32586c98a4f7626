/* include/tracer_multi.h -- C ABI of libtracer_amd_multi.so: one render spread over the GPUs of a node.
 *
 * NO COUNTERPART IN THE REFERENCE, which is single-GPU (no NCCL/MPI call sites). This is the
 * multi-GPU layer BASELINE.json's north_star defines for a C or C++ caller -- the maintainer of the
 * reference's Renderer::render (Renderer.cpp:153-207) -- who has no Python: one trc_ctx per GPU (include/tracer_abi.h),
 * the iterations split by sample index (rank r renders iterations r+1, r+1+G, ...) or the image split into bands of
 * rows, and ONE sum-reduce of the films to GPU 0 over RCCL/xGMI (ncclReduce, ncclFloat, ncclSum, inside
 * ncclGroupStart/End, each on its context's stream), then the display transform on GPU 0. Per-sample radiance does
 * not depend on G (the RNG is keyed by the absolute pixel and iteration); the reduced film differs from the one-GPU
 * film by float summation order only (<= 1e-5 * sqrt(iterations) relative).
 *
 * libtracer_amd.so itself does not link RCCL; this library does (and libtracer_amd.so). A Python / PyTorch process
 * should use tracer_amd.multi.render_partitioned instead (torch.distributed's RCCL), not load both.
 *
 * Every call returns 0 or a trc_status / the failing context's status; trc_multi_last_error() gives the text.
 * Calls on one trc_multi are not re-entrant.
 */
#ifndef TRACER_MULTI_H
#define TRACER_MULTI_H

#include <stdint.h>
#include "tracer_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trc_multi trc_multi;

/* One context + private stream per entry of `devices` (G = n_devices >= 1), and an RCCL communicator over them.
 * Device numbers must be distinct for RCCL; as a testing aid they may ALL be the same device (several contexts on
 * one GPU): the film sum is then a same-device kernel (film[0] += film[r], r = 1..G-1 in order) instead of RCCL. */
int trc_multi_create(const int* devices, int n_devices, uint32_t width, uint32_t height, trc_multi** out);
int trc_multi_destroy(trc_multi* m);
int trc_multi_size(const trc_multi* m);
trc_ctx* trc_multi_context(trc_multi* m, int rank);                      /* for per-GPU counters, timings, options */

/* trc_upload_scene / trc_update_object / trc_set_option on every context. */
int trc_multi_upload_scene(trc_multi* m, const void* objects, uint32_t count);
int trc_multi_update_object(trc_multi* m, uint32_t index, const float matrix[16], const float matrixi[16]);
int trc_multi_set_option(trc_multi* m, int option, int value);

/* Iterations 1 .. total_iters of mode 2 over all GPUs: films zeroed; by_tile == 0: GPU r renders iterations
 * r+1, r+1+G, ... of the whole frame; by_tile != 0: GPU r renders every iteration of its band of rows; one
 * sum-reduce onto GPU 0; GPU 0's LDR image = rgbToInt(pow(film / total_iters, 1/2.2)). Returns when GPU 0 holds the
 * result (waits for every stream). */
int trc_multi_pathtrace(trc_multi* m, const trc_camera* cam, uint64_t seed, uint32_t total_iters, int by_tile);

/* The combined result, from GPU 0 (trc_read_film / trc_read_ldr of context 0). */
int trc_multi_read_film(trc_multi* m, float* rgb);
int trc_multi_read_ldr(trc_multi* m, uint32_t* bgra);
/* Wall-clock milliseconds of the last trc_multi_pathtrace (first launch to reduced film), and the live path segments
 * all GPUs traced in it. */
int trc_multi_last_stats(trc_multi* m, double* ms, uint64_t* live_segments);
/* Where that time went, so that a scaling curve can explain itself:
 *   render_ms[r]   (G entries, may be NULL) GPU time of rank r's path kernels, between HIP events on its own stream;
 *   all_rendered_ms  host clock, from the first launch until every rank's render had finished;
 *   combine_ms       host clock, from there until GPU 0 held the reduced, tonemapped film: the sum-reduce over RCCL/xGMI
 *                    (or the same-device sum) plus the display transform. all_rendered_ms + combine_ms = ms of
 *                    trc_multi_last_stats. During a render the ranks' own display transforms are switched off
 *                    (TRC_OPT_SKIP_TONEMAP): only GPU 0 runs one, after the reduce. */
int trc_multi_last_breakdown(trc_multi* m, double* render_ms, double* all_rendered_ms, double* combine_ms);

const char* trc_multi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TRACER_MULTI_H */
