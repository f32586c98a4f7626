"""Minimal driver for rocprofv3 passes: the headline workload (Cornell 512x512, mode 2,
300 iterations) as `reps` launches of the selected pipeline, nothing else on the GPU.
    python3 tools/prof_run.py [pipeline=2] [reps=2] [iters=300] [w=512] [h=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracer_amd as ta

pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
w = int(sys.argv[4]) if len(sys.argv) > 4 else 512
h = int(sys.argv[5]) if len(sys.argv) > 5 else 512
sc = ta.Scene(int(os.environ.get('TRC_VARIANT', '0')))
cam = ta.camera_default(w, h)
tr = ta.Tracer(w, h)
tr.set_pipeline(pipe)
nobj = int(os.environ.get('TRC_NOBJ', sc.size))
tr.upload_scene(sc.objects_ptr, nobj)
for r in range(reps):
    tr.reset_counters()
    tr.pathtrace(cam, 1, 1, iters)
    ms = tr.last_kernel_ms()
    c = tr.counters()
    print(f"pipe {pipe} rep {r}: {ms:.3f} ms, {c['live_segments']} segs, {c['live_segments']/ms/1e3:.1f} Mseg/s, tri_tests {c['tri_tests']}, "
          f"paths {c['paths']}, shader_mhz {c['shader_mhz'] or 0:.1f}", flush=True)
tr.close()
