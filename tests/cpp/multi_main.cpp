// tests/cpp/multi_main.cpp -- a C++ caller of the multi-GPU layer (include/tracer_multi.h), no Python in the process:
// what a maintainer of the reference's Renderer::render would write to spread one render over the GPUs of a node.
//
//   multi_main <data_dir> <w> <h> <total_iters> <by_tile 0|1> <out_prefix> <device> [<device> ...]
//
// Builds the Cornell scene through the host API, renders iterations 1..total_iters over the listed devices (the same
// device listed several times = the library's same-device testing mode), writes <out_prefix>.film / .ldr from GPU 0
// and prints one line of statistics.
#include "../../include/tracer_multi.h"
#include "../../include/tracer_host.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s data_dir w h total_iters by_tile out_prefix device...\n", argv[0]); return 2; }
  const std::string data = argv[1], out = argv[6];
  const uint32_t w = atoi(argv[2]), h = atoi(argv[3]), iters = atoi(argv[4]);
  const int by_tile = atoi(argv[5]);
  std::vector<int> devices;
  for (int i = 7; i < argc; ++i) devices.push_back(atoi(argv[i]));

  trc_scene* scene = nullptr;
  if (trc_scene_cornell(0, 0, data.c_str(), &scene) != 0) { fprintf(stderr, "scene failed\n"); return 1; }
  trc_camera cam;
  if (trc_camera_default(w, h, 0.0f, 0.0f, &cam) != 0) return 1;

  trc_multi* m = nullptr;
  if (trc_multi_create(devices.data(), (int)devices.size(), w, h, &m) != 0) { fprintf(stderr, "create: %s\n", trc_multi_last_error()); return 1; }
  if (trc_multi_upload_scene(m, trc_scene_objects(scene), trc_scene_size(scene)) != 0) { fprintf(stderr, "upload: %s\n", trc_multi_last_error()); return 1; }
  for (int rep = 0; rep < 2; ++rep)      // twice: the second call must start from zeroed films again
    if (trc_multi_pathtrace(m, &cam, 7, iters, by_tile) != 0) { fprintf(stderr, "pathtrace: %s\n", trc_multi_last_error()); return 1; }

  std::vector<float> film(3ull * w * h);
  std::vector<uint32_t> ldr((size_t)w * h);
  if (trc_multi_read_film(m, film.data()) != 0 || trc_multi_read_ldr(m, ldr.data()) != 0) { fprintf(stderr, "read: %s\n", trc_multi_last_error()); return 1; }
  FILE* f = fopen((out + ".film").c_str(), "wb"); fwrite(film.data(), sizeof(float), film.size(), f); fclose(f);
  f = fopen((out + ".ldr").c_str(), "wb"); fwrite(ldr.data(), sizeof(uint32_t), ldr.size(), f); fclose(f);
  double ms = 0; uint64_t segs = 0;
  trc_multi_last_stats(m, &ms, &segs);
  printf("gpus %d iterations %u by_tile %d ms %.3f live_segments %llu Mseg/s %.1f\n", trc_multi_size(m), iters, by_tile, ms,
         (unsigned long long)segs, segs / ms / 1e3);
  std::vector<double> render(devices.size());
  double all_rendered = 0, combine = 0;
  if (trc_multi_last_breakdown(m, render.data(), &all_rendered, &combine) != 0) { fprintf(stderr, "breakdown: %s\n", trc_multi_last_error()); return 1; }
  printf("breakdown all_rendered_ms %.3f combine_ms %.3f render_ms", all_rendered, combine);
  for (double r : render) printf(" %.3f", r);
  printf("\n");
  trc_multi_destroy(m);
  trc_scene_free(scene);
  return 0;
}
