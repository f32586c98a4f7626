"""tools/bounce_trace.py -- when did every workgroup of every k_bounce launch start and end? (development aid; needs a build
with -DTRC_BOUNCE_TRACE: tools/build_variant.sh trace -DTRC_BOUNCE_TRACE [...], TRC_LIB=tracer_amd/variants/libtracer_trace.so)
    python3 tools/bounce_trace.py [pipe=1]
Prints per launch: workgroups that ran, span of the launch, and the distribution of workgroup start / end times within it."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracer_amd as ta

pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = h = 512
sc = ta.Scene(0); cam = ta.camera_default(w, h)
tr = ta.Tracer(w, h); tr.set_pipeline(pipe); tr.upload_scene(sc.objects_ptr, sc.size)
for _ in range(3):
    tr.pathtrace(cam, 1, 1, 300)
print("call: %.3f ms" % tr.last_kernel_ms())
L = tr.L
n = 5 * 32768 * 2
buf = np.zeros(n, dtype=np.uint64)
L.trc_debug_bounce_trace.argtypes = [C.c_void_p, C.c_size_t]
rc = L.trc_debug_bounce_trace(buf.ctypes.data, n)
assert rc == 0, rc
t = buf.reshape(5, 32768, 2).astype(np.int64)
for d in range(5):
    ran = t[d, :, 1] > 0
    if not ran.any():
        continue
    s, e = t[d, ran, 0], t[d, ran, 1]
    t0, t1 = s.min(), e.max()
    span = (t1 - t0) / 100.0          # us (100 MHz)
    q = lambda x: " ".join("%7.1f" % ((v - t0) / 100.0) for v in np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"depth {d}: {ran.sum():6d} workgroups, span {span:8.1f} us | start p0/10/50/90/100: {q(s)} | end: {q(e)} | mean life {np.mean(e - s) / 100.0:7.1f} us")
    ids = np.nonzero(ran)[0]
    # end time by region (blockIdx // bpr) for persistent grids
    if ran.sum() <= 4096 and ran.sum() % 256 == 0:
        g = ran.sum() // 256
        ee = ((t[d, :g * 256, 1] - t0) / 100.0).reshape(g, 256)     # [generation][block % 256]
        ee = np.stack([np.roll(ee[k], k) for k in range(g)])       # region = (block + generation) % 256
        rmax, rmin = ee.max(0), ee.min(0)
        print("   per region: end of its last workgroup p0/50/100: %.1f %.1f %.1f us; spread inside a region mean %.1f max %.1f us" %
              (rmax.min(), np.median(rmax), rmax.max(), (rmax - rmin).mean(), (rmax - rmin).max()))
        print("   regions ending first:", np.argsort(rmax)[:10].tolist(), " last:", np.argsort(rmax)[-10:].tolist())
        print("   mean end by generation:", " ".join("%.1f" % x for x in ee.mean(1)))
    if ran.sum() <= 4096:
        order = np.argsort(e)
        print("   earliest-ending blocks:", ids[order[:8]].tolist(), " latest:", ids[order[-8:]].tolist())
