"""Robustness check (development aid, run on the GPU box): a scene of 192 objects and scenes with empty object
groups through every pipeline against the oracle: python3 tools/many_objects.py"""
import os, sys, tempfile, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import tracer_amd as ta
from oracle.pyoracle import Oracle, Camera as OCam
from scene_gen import clutter_scene
o = Oracle("c"); o.set_math_mode(Oracle.MATH_PORTABLE)
with tempfile.TemporaryDirectory() as tmp:
    for (ns, npn, nc, nm) in [(120, 40, 20, 6), (0, 0, 0, 0), (1, 0, 0, 5), (0, 30, 0, 0), (0, 0, 9, 0)]:
        path = clutter_scene(os.path.join(tmp, "s.scene"), 5, n_spheres=ns, n_panels=npn, n_cubes=nc, n_meshes=nm)
        w, h = 96, 64
        sc = ta.Scene(path=path, aspect=(w, h)); s = o.scene_wrap(sc.objects_ptr, sc.size); cam = OCam.from_array(sc.camera.as_array())
        tr = ta.Tracer(w, h); tr.upload_scene(sc.objects_ptr, sc.size)
        ref = o.pathtrace(s, w, h, cam, 3, 1, 3, nthreads=16)
        for pipe in (0, 1, 2, 3):
            tr.set_pipeline(pipe); tr.reset_counters(); tr.pathtrace(sc.camera, 3, 1, 3)
            ok = tr.read_film().tobytes() == ref["film"].tobytes()
            print(sc.size, "objects pipe", pipe, "ok" if ok else "MISMATCH", tr.last_kernel_ms(), flush=True)
        tr.close()
