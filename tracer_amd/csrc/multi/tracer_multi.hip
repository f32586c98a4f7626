// tracer_amd/csrc/multi/tracer_multi.hip -- implementation of include/tracer_multi.h on top of the context API
// (include/tracer_abi.h) and RCCL. Host code plus one trivial kernel (the same-device film sum of the testing aid).
#include "../../../include/tracer_multi.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(TRC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(TRC_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); } while (0)
#define TRC_TRY(expr) do { int rc_ = (expr); if (rc_ != TRC_OK) return fail(rc_, std::string(#expr) + ": " + trc_last_error_string()); } while (0)

__global__ void k_film_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = dst[i] + src[i];
}

} // namespace

struct trc_multi {
  std::vector<int> devices;
  std::vector<trc_ctx*> ctx;
  std::vector<hipStream_t> streams;
  std::vector<ncclComm_t> comms;      // empty: same-device mode (no RCCL)
  uint32_t w = 0, h = 0;
  std::vector<hipEvent_t> rendered;   // per rank: recorded on its stream behind its render
  double last_ms = 0.0, last_all_rendered_ms = 0.0, last_combine_ms = 0.0;
  std::vector<double> last_render_ms;
  uint64_t last_segments = 0;
};

extern "C" {

const char* trc_multi_last_error(void) { return g_err.c_str(); }

int trc_multi_create(const int* devices, int n_devices, uint32_t width, uint32_t height, trc_multi** out) {
  if (!out) return fail(TRC_ERR_INVALID, "trc_multi_create: out is null");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(TRC_ERR_INVALID, "trc_multi_create: bad device list");
  bool all_same = true, distinct = true;
  for (int i = 0; i < n_devices; ++i) {
    all_same = all_same && devices[i] == devices[0];
    for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  }
  if (n_devices > 1 && !distinct && !all_same)
    return fail(TRC_ERR_INVALID, "trc_multi_create: device numbers must be distinct (or all equal: same-device testing mode)");
  trc_multi* m = new trc_multi;
  m->w = width; m->h = height;
  m->devices.assign(devices, devices + n_devices);
  int rc = [&]() -> int {
    for (int r = 0; r < n_devices; ++r) {
      HIP_TRY(hipSetDevice(devices[r]));
      hipStream_t s = nullptr;
      HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      m->streams.push_back(s);
      trc_ctx* c = nullptr;
      TRC_TRY(trc_create(devices[r], width, height, s, &c));
      m->ctx.push_back(c);
      hipEvent_t e = nullptr;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      m->rendered.push_back(e);
    }
    // (TRC_MULTI_RCCL_SINGLE=1: a communicator and the ncclReduce call for ONE GPU too -- a one-rank reduce is a no-op
    //  copy -- so that a one-GPU box can exercise the RCCL path end to end)
    const char* single = getenv("TRC_MULTI_RCCL_SINGLE");
    if ((n_devices > 1 && distinct) || (n_devices == 1 && single && single[0] == '1')) {
      m->comms.resize(n_devices);
      NCCL_TRY(ncclCommInitAll(m->comms.data(), n_devices, devices));
    }
    return TRC_OK;
  }();
  if (rc != TRC_OK) { const std::string keep = g_err; trc_multi_destroy(m); g_err = keep; return rc; }
  *out = m;
  return TRC_OK;
}

int trc_multi_destroy(trc_multi* m) {
  if (!m) return TRC_OK;
  for (ncclComm_t c : m->comms) if (c) (void)ncclCommDestroy(c);
  for (trc_ctx* c : m->ctx) (void)trc_destroy(c);
  for (size_t r = 0; r < m->rendered.size(); ++r) { (void)hipSetDevice(m->devices[r]); (void)hipEventDestroy(m->rendered[r]); }
  for (size_t r = 0; r < m->streams.size(); ++r) { (void)hipSetDevice(m->devices[r]); (void)hipStreamDestroy(m->streams[r]); }
  delete m;
  return TRC_OK;
}

int trc_multi_size(const trc_multi* m) { return m ? (int)m->ctx.size() : 0; }
trc_ctx* trc_multi_context(trc_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[rank] : nullptr; }

int trc_multi_upload_scene(trc_multi* m, const void* objects, uint32_t count) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  for (trc_ctx* c : m->ctx) TRC_TRY(trc_upload_scene(c, objects, count));
  return TRC_OK;
}

int trc_multi_update_object(trc_multi* m, uint32_t index, const float matrix[16], const float matrixi[16]) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  for (trc_ctx* c : m->ctx) TRC_TRY(trc_update_object(c, index, matrix, matrixi));
  return TRC_OK;
}

int trc_multi_set_option(trc_multi* m, int option, int value) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  for (trc_ctx* c : m->ctx) TRC_TRY(trc_set_option(c, option, value));
  return TRC_OK;
}

int trc_multi_pathtrace(trc_multi* m, const trc_camera* cam, uint64_t seed, uint32_t total_iters, int by_tile) {
  if (!m || !cam) return fail(TRC_ERR_INVALID, "null argument");
  if (total_iters == 0) return fail(TRC_ERR_INVALID, "trc_multi_pathtrace: no iterations");
  const uint32_t G = (uint32_t)m->ctx.size();
  const size_t n_floats = 3ull * m->w * m->h;
  const bool combine = G > 1 || !m->comms.empty();
  // a rank's own display transform is wasted work when a combine follows: GPU 0 runs the one that counts, after the reduce
  for (trc_ctx* c : m->ctx) { TRC_TRY(trc_reset_counters(c)); TRC_TRY(trc_set_option(c, TRC_OPT_SKIP_TONEMAP, combine ? 1 : 0)); }
  auto restore = [&]() { for (trc_ctx* c : m->ctx) (void)trc_set_option(c, TRC_OPT_SKIP_TONEMAP, 0); };
  std::vector<char> rendered_something(G, 0);
  const auto t0 = std::chrono::steady_clock::now();
  // every GPU: zero its film, render its share; all launches are asynchronous on the context's stream
  for (uint32_t r = 0; r < G; ++r) {
    trc_ctx* c = m->ctx[r];
    int rc = trc_reset_film(c);
    if (rc == TRC_OK && by_tile) {
      const uint32_t y0 = (uint32_t)((uint64_t)m->h * r / G), y1 = (uint32_t)((uint64_t)m->h * (r + 1) / G);
      if (y1 > y0) {
        rc = trc_set_tile(c, y0, y1 - y0);
        if (rc == TRC_OK) { rc = trc_pathtrace(c, cam, seed, 1, 1, total_iters); rendered_something[r] = 1; }
        (void)trc_set_tile(c, 0, 0);
      }
    } else if (rc == TRC_OK) {
      const uint32_t count = total_iters > r ? (total_iters - r + G - 1) / G : 0;      // iterations r+1, r+1+G, ...
      if (count) { rc = trc_pathtrace(c, cam, seed, r + 1, G, count); rendered_something[r] = 1; }
    }
    if (rc != TRC_OK) { restore(); return fail(rc, std::string("rank ") + std::to_string(r) + ": " + trc_last_error_string()); }
    (void)hipSetDevice(m->devices[r]);
    HIP_TRY(hipEventRecord(m->rendered[r], m->streams[r]));
  }
  restore();
  // the path's one exchange: sum of the films onto GPU 0
  if (combine) {
    std::vector<void*> film(G);
    for (uint32_t r = 0; r < G; ++r) TRC_TRY(trc_film_device_ptr(m->ctx[r], &film[r]));
    if (!m->comms.empty()) {
      NCCL_TRY(ncclGroupStart());
      for (uint32_t r = 0; r < G; ++r) {
        const ncclResult_t nr = ncclReduce(film[r], film[r], n_floats, ncclFloat, ncclSum, 0, m->comms[r], m->streams[r]);
        if (nr != ncclSuccess) { (void)ncclGroupEnd(); return fail(TRC_ERR_HIP, std::string("ncclReduce: ") + ncclGetErrorString(nr)); }
      }
      NCCL_TRY(ncclGroupEnd());
    } else {
      // same-device testing mode: the contexts share one GPU; stream 0 waits (on the GPU) for the other renders and adds
      // their films in rank order
      HIP_TRY(hipSetDevice(m->devices[0]));
      for (uint32_t r = 1; r < G; ++r) HIP_TRY(hipStreamWaitEvent(m->streams[0], m->rendered[r], 0));
      for (uint32_t r = 1; r < G; ++r)
        hipLaunchKernelGGL(k_film_add, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, m->streams[0],
                           (float*)film[0], (const float*)film[r], n_floats);
      HIP_TRY(hipGetLastError());
    }
    TRC_TRY(trc_tonemap(m->ctx[0], (float)total_iters));
  }
  // host clock: when every rank's render had finished, then when GPU 0 held the result
  for (uint32_t r = 0; r < G; ++r) { (void)hipSetDevice(m->devices[r]); HIP_TRY(hipEventSynchronize(m->rendered[r])); }
  const auto t1 = std::chrono::steady_clock::now();
  for (trc_ctx* c : m->ctx) TRC_TRY(trc_synchronize(c));
  const auto t2 = std::chrono::steady_clock::now();
  m->last_all_rendered_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  m->last_combine_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  m->last_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
  m->last_segments = 0;
  m->last_render_ms.assign(G, 0.0);
  for (uint32_t r = 0; r < G; ++r) {
    trc_counters k; TRC_TRY(trc_get_counters(m->ctx[r], &k)); m->last_segments += k.live_segments;
    float ms = 0.0f;
    if (rendered_something[r]) { TRC_TRY(trc_last_kernel_ms(m->ctx[r], &ms)); m->last_render_ms[r] = ms; }
  }
  return TRC_OK;
}

int trc_multi_read_film(trc_multi* m, float* rgb) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  TRC_TRY(trc_read_film(m->ctx[0], rgb));
  return TRC_OK;
}

int trc_multi_read_ldr(trc_multi* m, uint32_t* bgra) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  TRC_TRY(trc_read_ldr(m->ctx[0], bgra));
  return TRC_OK;
}

int trc_multi_last_stats(trc_multi* m, double* ms, uint64_t* live_segments) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  if (ms) *ms = m->last_ms;
  if (live_segments) *live_segments = m->last_segments;
  return TRC_OK;
}

int trc_multi_last_breakdown(trc_multi* m, double* render_ms, double* all_rendered_ms, double* combine_ms) {
  if (!m) return fail(TRC_ERR_INVALID, "null trc_multi");
  if (render_ms) for (size_t r = 0; r < m->ctx.size(); ++r) render_ms[r] = r < m->last_render_ms.size() ? m->last_render_ms[r] : 0.0;
  if (all_rendered_ms) *all_rendered_ms = m->last_all_rendered_ms;
  if (combine_ms) *combine_ms = m->last_combine_ms;
  return TRC_OK;
}

} // extern "C"
