"""Parity soak (development aid, run on the GPU box): many seeded random scenes -- cluttered ones and
axis-aligned ones -- path-traced by every pipeline and compared bit for bit with the CPU oracle.
    python3 tools/soak.py [first_seed=100] [n_seeds=20]
With TRC_LIB=tracer_amd/variants/libtracer_pairs_check1.so (the checking build of the (ray, object) pre-test, intersect.h) the
tri_tests counter counts pairs that the pre-test called a certain miss and that hit after all: it must stay zero."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import tracer_amd as ta
from oracle.pyoracle import Oracle, Camera as OCam
from scene_gen import clutter_scene, aligned_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
o = Oracle("c")
o.set_math_mode(Oracle.MATH_PORTABLE)
bad = 0
checking = "pairs_check1" in os.environ.get("TRC_LIB", "")
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + n):
        r = np.random.default_rng(seed)
        w, h = int(r.integers(5, 20)) * 8, int(r.integers(5, 14)) * 8
        if seed % 3 == 0:
            path = aligned_scene(os.path.join(tmp, "s.scene"), lens=float(r.choice([0.0, 0.4, 1.0])))
        else:
            path = clutter_scene(os.path.join(tmp, "s.scene"), seed, n_spheres=int(r.integers(1, 14)), n_panels=int(r.integers(0, 7)),
                                 n_cubes=int(r.integers(0, 4)), n_meshes=int(r.integers(0, 4)))
        sc = ta.Scene(path=path, aspect=(w, h))
        s = o.scene_wrap(sc.objects_ptr, sc.size)
        cam = OCam.from_array(sc.camera.as_array())
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        iters = int(r.integers(2, 7))
        rng = int(seed % 2)
        ref = o.pathtrace(s, w, h, cam, seed, 1, iters, rng_mode=rng, nthreads=16)
        for pipe in (0, 1, 2, 3):
            tr.set_pipeline(pipe); tr.set_rng_mode(rng); tr.reset_counters()
            tr.pathtrace(sc.camera, seed, 1, iters)
            same = tr.read_film().tobytes() == ref["film"].tobytes() and tr.counters()["live_segments"] == ref["stats"]["live_segments"]
            if checking and tr.counters()["tri_tests"] != 0:
                same = False
                print(f"PRE-TEST VIOLATIONS: {tr.counters()['tri_tests']}", flush=True)
            if not same:
                bad += 1
                print(f"MISMATCH seed {seed} pipe {pipe} {w}x{h} iters {iters} rng {rng} objects {sc.size}", flush=True)
        tr.close()
        print(f"seed {seed}: {w}x{h}, {sc.size} objects, {iters} iterations, {ref['stats']['live_segments']} segments ok", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
