"""Parity soak (development aid, run on the GPU box): many seeded random scenes -- cluttered ones and
axis-aligned ones -- path-traced by every pipeline and compared bit for bit with the CPU oracle.
    python3 tools/soak.py [first_seed=100] [n_seeds=20] [--json record.json]
--json writes a record of the run (seeds, scenes, segments compared, mismatches, pre-test violations): profiles/r03_soak*.json
are such records, and tests/test_gpu_soak.py runs a fixed slice through the product and the checking build.
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

import json
import time
json_path = None
if "--json" in sys.argv:
    k = sys.argv.index("--json"); json_path = sys.argv[k + 1]; del sys.argv[k:k + 2]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t_start = time.time()
rec = {"tool": "tools/soak.py", "library": os.environ.get("TRC_LIB") or "tracer_amd/libtracer_amd.so", "first_seed": first, "n_seeds": n,
       "scenes": 0, "films_compared": 0, "segments_compared": 0, "objects": 0, "mismatches": 0, "pre_test_violations": 0,
       "what": "seeded random scenes (clutter_scene: 1-14 spheres, 0-6 panels, 0-3 cubes, 0-3 meshes, arbitrary transforms; every third: "
               "aligned_scene), 2-6 iterations, RNG mode = seed % 2, all four pipelines: film and live-segment count against the C oracle "
               "(portable math), bit for bit"}
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
            rec["films_compared"] += 1; rec["segments_compared"] += ref["stats"]["live_segments"]
            if checking and tr.counters()["tri_tests"] != 0:
                same = False
                rec["pre_test_violations"] += tr.counters()["tri_tests"]
                print(f"PRE-TEST VIOLATIONS: {tr.counters()['tri_tests']}", flush=True)
            if not same:
                bad += 1
                print(f"MISMATCH seed {seed} pipe {pipe} {w}x{h} iters {iters} rng {rng} objects {sc.size}", flush=True)
        tr.close()
        rec["scenes"] += 1; rec["objects"] += sc.size
        print(f"seed {seed}: {w}x{h}, {sc.size} objects, {iters} iterations, {ref['stats']['live_segments']} segments ok", flush=True)
print("mismatches:", bad)
rec["mismatches"] = bad
rec["checking_build"] = checking
rec["seconds"] = round(time.time() - t_start, 1)
if json_path:
    with open(json_path, "w") as f:
        json.dump(rec, f, indent=1)
sys.exit(1 if bad else 0)
