"""Worker of tests/test_gpu_pairs.py: renders a list of scenarios with the library that TRC_LIB names (the product build
or a checking build of the (ray, object) pre-test, intersect.h: TRC_PAIRS_CHECK) and prints one JSON line:
per scenario the tri_tests counter and a digest of every output.   usage: python pairs_check_worker.py TMPDIR"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import tracer_amd as ta  # noqa: E402
from scene_gen import aligned_scene, clutter_scene, stress_scene  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def ray_batches(tr, rng, n=16384):
    """Caller rays through trc_intersect_rays: unit and non-unit directions, near and far origins, axis-aligned
    directions, and rays that start on the surfaces the first batch hit."""
    out = []
    o = rng.uniform(-20, 20, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = np.concatenate([o, d], 1).astype(np.float32)
    a[:, 3:] /= np.linalg.norm(a[:, 3:].astype(np.float64), axis=1, keepdims=True).astype(np.float32)   # float32 unit vectors
    first = tr.intersect_rays(a); out.append(first)
    b = a.copy(); b[:, 3:] *= rng.uniform(0.1, 10.0, (n, 1)).astype(np.float32)                        # not unit length
    out.append(tr.intersect_rays(b))
    far = a.copy(); far[:, :3] = (rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.uniform(2, 6, (n, 1))).astype(np.float32)
    far[:, 3:] = -far[:, :3] / np.linalg.norm(far[:, :3], axis=1, keepdims=True)                        # aimed at the origin
    far[:, 3:] += rng.normal(size=(n, 3)).astype(np.float32) * 1e-3
    far[:, 3:] /= np.linalg.norm(far[:, 3:], axis=1, keepdims=True)
    out.append(tr.intersect_rays(far))
    ax = a.copy(); ax[:, 3:] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], (n, 1)).astype(np.float32)
    out.append(tr.intersect_rays(ax))
    # degenerate rays: NaN / inf coordinates, zero and denormal directions (the pre-test's comparisons are all false for a NaN,
    # and a direction that is not of unit length keeps every pair)
    bad = a[:1024].copy()
    sel = rng.integers(0, 6, len(bad))
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, 1e-42, 3e38], np.float32)[rng.integers(0, 6, len(bad))]
    bad[np.arange(len(bad)), sel] = vals
    bad[::7, 3:] = 0.0
    out.append(tr.intersect_rays(bad))
    hit = first["ids"] >= 0
    s = a[hit].copy()
    nd = rng.normal(size=(len(s), 3)); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    s[:, 3:] = nd.astype(np.float32)
    s[:, :3] = first["pos"][hit] + np.float32(0.001) * s[:, 3:]                                        # raytrace.cu:159
    if len(s):
        out.append(tr.intersect_rays(s))
        # grazing rays: aimed at points on or just beside the surfaces found above, from 0.01 to 10^5 units away
        tgt = first["pos"][hit].astype(np.float64) + rng.normal(size=(len(s), 3)) * 10.0 ** rng.uniform(-4, 0, (len(s), 1))
        away = rng.normal(size=(len(s), 3)); away /= np.linalg.norm(away, axis=1, keepdims=True)
        dist = 10.0 ** rng.uniform(-2, 5, (len(s), 1))
        g = np.concatenate([tgt + away * dist, -away], 1).astype(np.float32)
        g[:, 3:] /= np.linalg.norm(g[:, 3:].astype(np.float64), axis=1, keepdims=True).astype(np.float32)
        out.append(tr.intersect_rays(g))
    return out


def main(tmp):
    res = {}

    def run(name, sc, w, h, iters=3, pipes=(3, 1), batches=True):
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        cam = sc.camera if hasattr(sc, "camera") and sc.camera is not None else ta.camera_default(w, h)
        tr.reset_counters()
        dig = []
        tr.raytrace1(cam); hits = tr.read_hits()
        dig.append(digest(hits["ids"], hits["ts"], hits["pos"], hits["nor"]))
        for pipe in pipes:
            for rng_mode in (0, 1):
                tr.set_pipeline(pipe); tr.set_rng_mode(rng_mode)
                tr.pathtrace(cam, 7, 1, iters)
                dig.append(digest(tr.read_film()))
        if batches:
            for b in ray_batches(tr, np.random.default_rng(5)):
                dig.append(digest(b["ids"], b["ts"], b["pos"], b["nor"]))
        c = tr.counters()
        res[name] = dict(tri_tests=c["tri_tests"], digest=dig)
        tr.close()

    for v in (0, 1, 2):
        sc = ta.Scene(v)
        sc.camera = None
        run("cornell_v%d" % v, sc, 96, 64, pipes=(3, 2, 1, 0))
    for seed in (1, 2, 3, 4, 5, 6):
        run("clutter_%d" % seed, ta.Scene(path=clutter_scene(os.path.join(tmp, "c%d.scene" % seed), seed), aspect=(112, 80)), 112, 80)
    run("aligned", ta.Scene(path=aligned_scene(os.path.join(tmp, "a.scene"), 0.7), aspect=(96, 96)), 96, 96)
    for seed in (1, 2, 3, 4, 5, 6):
        run("stress_%d" % seed, ta.Scene(path=stress_scene(os.path.join(tmp, "s%d.scene" % seed), seed), aspect=(128, 96)), 128, 96)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
