#!/usr/bin/env python3
"""tools/iters_scan.py -- k_queue launch time against the number of iterations per launch (512x512 Cornell box): the
intercept of the fit is what a launch costs besides its segments (start-up, the drain at the end, the accumulate pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tracer_amd as ta
sc = ta.Scene(0); cam = ta.camera_default(512, 512)
tr = ta.Tracer(512, 512); tr.set_pipeline(3); tr.upload_scene(sc.objects_ptr, sc.size)
xs, ys = [], []
for iters in (25, 50, 100, 150, 200, 300):
    ms = []
    for r in range(8):
        tr.pathtrace(cam, 1, 1, iters)
        if r >= 2: ms.append(tr.last_kernel_ms())
    xs.append(iters); ys.append(min(ms))
    print(f"{iters:4d} iterations: min {min(ms):.3f} ms  median {sorted(ms)[len(ms)//2]:.3f} ms", flush=True)
a, b = np.polyfit(xs, ys, 1)
print(f"fit: {b:.3f} ms + {a*1e3:.3f} us / iteration")
