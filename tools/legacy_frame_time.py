#!/usr/bin/env python3
"""tools/legacy_frame_time.py -- one interactive 512x512 path-tracing frame (the reference's use: one iteration per
displayed frame, Renderer.cpp:183-190) through the reference's own `pathtrace` signature (legacy shim: GPU content
digest of scene_d + fused kernel + display transform) against the same frame through trc_pathtrace. Needs a GPU."""
import ctypes as C
import json
import os
import struct
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "tests"))


def main():
    import numpy as np
    import torch
    import tracer_amd as ta
    from test_gpu_boundary import _device_scene
    L = ta.load_library()
    w = h = 512
    n = w * h
    sc = ta.Scene(0)
    scene_d, keep = _device_scene(torch, sc.raw_objects(), sc.size)
    cam = ta.camera_default(w, h)
    f3 = C.c_float * 3
    args_cam = [f3(*cam.campos), f3(*cam.A), f3(*cam.B), f3(*cam.C)]
    vp = C.c_void_p
    L.pathtrace.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, f3, f3, f3, f3, C.c_float, C.c_float,
                            vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32]
    L.pathtrace.restype = None
    pbo = torch.zeros(n, dtype=torch.int32, device="cuda")
    film = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
    dummy = torch.zeros(n * 6, dtype=torch.float32, device="cuda")
    idx = torch.zeros(n, dtype=torch.int32, device="cuda")

    def legacy(frames, first):
        for it in range(first, first + frames):
            L.pathtrace(pbo.data_ptr(), w, h, 0.01 * it, *args_cam, cam.lensRadius, cam.focalDist, scene_d.data_ptr(),
                        sc.size, dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), idx.data_ptr(), film.data_ptr(), it)
        torch.cuda.synchronize()

    legacy(20, 1)
    t0 = time.perf_counter(); legacy(200, 21); legacy_ms = (time.perf_counter() - t0) / 200 * 1e3
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    for it in range(1, 21):
        tr.pathtrace(cam, it, it, 1)
    tr.synchronize()
    t0 = time.perf_counter()
    for it in range(21, 221):
        tr.pathtrace(cam, it, it, 1)
    tr.synchronize()
    ctx_ms = (time.perf_counter() - t0) / 200 * 1e3
    tr.pathtrace(cam, 1, 1, 1); tr.synchronize()
    print(json.dumps({"frame": "cornell 512x512, one iteration per call, 200 calls back to back (host wall clock)",
                      "legacy_pathtrace_ms_per_frame": round(legacy_ms, 4), "trc_pathtrace_ms_per_frame": round(ctx_ms, 4),
                      "trc_pathtrace_gpu_ms": round(tr.last_render_ms(), 4)}))


if __name__ == "__main__":
    main()
