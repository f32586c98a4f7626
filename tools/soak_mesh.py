"""Soak of the bounding-sphere pre-test of the big-mesh pass (development aid, run on the GPU box): random big
meshes (subdivided icosahedra, stretched, placed far from their object-space origin), random cameras; mode 1
through the default loop (pre-test) and through the loop without it must agree in every bit of every pixel.
    python3 tools/soak_mesh.py [first_seed=0] [n_seeds=100] [--json record.json]
With TRC_LIB=tracer_amd/variants/libtracer_pairs_check1.so (checking build of intersect.h's (ray, object) pre-test and of the
mesh hull cull: every pair tested, rejections that hit after all counted in tri_tests) that count must stay zero too."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import tracer_amd as ta
from scene_gen import icosphere_obj, offset_mesh_scene

import json
import time
json_path = None
if "--json" in sys.argv:
    k = sys.argv.index("--json"); json_path = sys.argv[k + 1]; del sys.argv[k:k + 2]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_start = time.time()
violations = 0
faces_total = 0
bad = 0
seen = 0
checking = "pairs_check1" in os.environ.get("TRC_LIB", "")
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + n):
        r = np.random.default_rng(seed)
        offset = tuple(r.uniform(-1, 1, 3) * 10.0 ** r.uniform(0, 2.7))
        stretch = tuple(r.uniform(0.3, 2.5, 3)) if seed % 2 else (1.0, 1.0, 1.0)
        level = 3 + int(seed % 3 == 0)
        obj, nf = icosphere_obj(os.path.join(tmp, "m.obj"), level, offset, stretch, float(r.uniform(0.5, 3.0)))
        w, h = 128, 96
        sc = ta.Scene(path=offset_mesh_scene(os.path.join(tmp, "m.scene"), obj, offset, scale=float(r.uniform(0.7, 2.0)), lens=0.0), aspect=(w, h))
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        cam = ta.camera_default(w, h, float(r.uniform(-0.5, 0.5)), float(r.uniform(-0.4, 0.4)))
        out = {}
        tr.reset_counters()
        for accel in (0, 2):
            tr.set_mesh_accel(accel)
            tr.raytrace1(cam)
            out[accel] = tr.read_hits()
        faces_total += nf
        if checking and tr.counters()["tri_tests"] != 0:
            bad += 1
            violations += tr.counters()["tri_tests"]
            print(f"PRE-TEST / HULL VIOLATIONS seed {seed}: {tr.counters()['tri_tests']}", flush=True)
        same = all(out[0][k].tobytes() == out[2][k].tobytes() for k in ("ids", "ts", "pos", "nor"))
        npx = int((out[2]["ids"] == 6).sum())
        seen += npx
        if not same:
            bad += 1
            print(f"MISMATCH seed {seed}: {nf} faces offset {offset} stretch {stretch}", flush=True)
        tr.close()
        if seed % 10 == 0:
            print(f"seed {seed}: {nf} faces, |offset| {np.linalg.norm(offset):.1f}, {npx} mesh pixels ok", flush=True)
print("mismatches:", bad, "mesh pixels compared:", seen)
if json_path:
    with open(json_path, "w") as f:
        json.dump({"tool": "tools/soak_mesh.py", "library": os.environ.get("TRC_LIB") or "tracer_amd/libtracer_amd.so", "checking_build": checking,
                   "first_seed": first, "n_seeds": n, "meshes": n, "faces": faces_total, "mesh_pixels_compared": seen, "mismatches": bad,
                   "pre_test_violations": violations, "seconds": round(time.time() - t_start, 1),
                   "what": "random big meshes (icosphere level 3-4, stretched, up to 500 units from their object-space origin, random cameras), "
                           "mode 1 at 128x96 through the big-mesh pass with (TRC_OPT_MESH_ACCEL 0) and without (2) its bounding-sphere "
                           "pre-test: ids, t, position, normal of every pixel, bit for bit"}, f, indent=1)
sys.exit(1 if bad else 0)
