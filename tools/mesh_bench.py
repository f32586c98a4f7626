"""configs[4]: Cornell walls + one tessellated icosphere MESH, brute-force closest hit.
    python3 tools/mesh_bench.py [level=8] [w=1024] [h=1024] [mode2_iters=1] [pipeline=-1 (library default)] [accel=0|1|2]
Prints mode-1 and mode-2 kernel time and ray-triangle tests/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracer_amd as ta

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pipe = int(sys.argv[5]) if len(sys.argv) > 5 else -1
accel = int(sys.argv[6]) if len(sys.argv) > 6 else 0
t0 = time.time()
sc = ta.Scene(2, level)
print(f"scene built in {time.time()-t0:.2f} s (level {level}: {20 * 4**level} triangles)", flush=True)
cam = ta.camera_default(w, h)
tr = ta.Tracer(w, h)
if pipe >= 0:
    tr.set_pipeline(pipe)
t0 = time.time()
tr.upload_scene(sc.objects_ptr, sc.size)
print(f"scene uploaded in {time.time()-t0:.2f} s", flush=True)
tr.set_record_hits(True)
tr.set_mesh_accel(accel)
print("big meshes:", ["every face (the reference's loop; bounding-sphere pre-test)", "BVH (TRC_OPT_MESH_ACCEL = 1)",
                      "every face, no pre-test (TRC_OPT_MESH_ACCEL = 2)"][accel], flush=True)
for rep in range(2):
    tr.reset_counters()
    tr.raytrace1(cam)
    ms = tr.last_kernel_ms(); c = tr.counters()
    print(f"mode 1 rep {rep}: {ms:.2f} ms, tri_tests {c['tri_tests']}, {c['tri_tests']/ms/1e6:.2f} Gtests/s, shader_mhz {c['shader_mhz'] or 0:.1f}", flush=True)
if iters > 0:
    for rep in range(2):
        tr.reset_counters()
        tr.pathtrace(cam, 1, 1, iters)
        ms = tr.last_kernel_ms(); c = tr.counters()
        print(f"mode 2 rep {rep} ({iters} it): {ms:.2f} ms, segs {c['live_segments']}, tri_tests {c['tri_tests']:.4g}, "
              f"{c['tri_tests']/ms/1e6:.2f} Gtests/s, {c['live_segments']/ms/1e3:.1f} Mseg/s", flush=True)
tr.close()
