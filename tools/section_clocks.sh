#!/bin/bash
# Where k_queue's wavefronts spend their cycles: one diagnostic build per section (-DTRC_SECTION=n: the tri_tests counter
# sums s_memtime differences over all wavefronts), each run once on the headline workload.
#   build here:  tools/section_clocks.sh build      run on the GPU box:  tools/section_clocks.sh run
SECTS="1 2 3 4 5 6 8 9"
if [ "$1" = build ]; then
  for n in $SECTS; do tools/build_variant.sh sect$n -DTRC_SECTION=$n > /dev/null 2>&1 & done; wait; ls tracer_amd/variants/ | grep sect
else
  for n in $SECTS; do
    TRC_LIB=$PWD/tracer_amd/variants/libtracer_sect$n.so timeout -k 10 60 python3 - $n <<'PY' 2>/dev/null || exit 1
import sys, os
sys.path.insert(0, os.getcwd())
import tracer_amd as ta
names = {1: "planes", 2: "pairs: pre-test + compaction", 3: "pairs: pass (all)", 4: "pairs: pass, mesh stage", 5: "pairs: pass, world half + merge", 6: "closest_hit (all)", 8: "(count) pairs", 9: "(count) passes"}
sc = ta.Scene(0); cam = ta.camera_default(512, 512)
tr = ta.Tracer(512, 512); tr.set_pipeline(3); tr.upload_scene(sc.objects_ptr, sc.size)
tr.pathtrace(cam, 1, 1, 300)
tr.reset_counters(); tr.pathtrace(cam, 1, 1, 300); c = tr.counters()
n = int(sys.argv[1])
cyc = c['tri_tests'] / 64      # all 64 lanes of a wavefront accumulate the same stamps
print(f"section {n} {names[n]:34s}: {cyc / c['wave_cycles'] * 100:6.2f} % of the wave cycles  ({cyc / (c['live_segments'] / 64):8.1f} wave cycles per 64 segments; kernel {tr.last_kernel_ms():.3f} ms)")
PY
  done
fi
