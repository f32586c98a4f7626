"""tools/crossover.py -- host time per trc_pathtrace call of 1 .. 32 iterations (512x512 Cornell box) through the fused and the
work-queue pipeline: where TRC_PIPE_AUTO's switch (2.4 M paths per call) comes from. Needs a GPU."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import tracer_amd as ta
sc = ta.Scene(0); cam = ta.camera_default(512, 512)
tr = ta.Tracer(512, 512); tr.upload_scene(sc.objects_ptr, sc.size)
for iters in (1, 2, 4, 8, 12, 16, 32):
    row = []
    for pipe in (2, 3):
        tr.set_pipeline(pipe)
        for _ in range(5): tr.pathtrace(cam, 1, 1, iters)
        tr.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n): tr.pathtrace(cam, 1, 1, iters)
        tr.synchronize(); row.append((time.perf_counter() - t0) / n * 1e3)
    print(f"{iters:3d} iterations: fused {row[0]:.4f} ms  queue {row[1]:.4f} ms")
