#!/usr/bin/env python3
"""tools/make_golden.py -- regenerate tests/golden/*.npz from the REFERENCE-LINKED oracle
(oracle/_ref/libtracer_ref.so: the reference's own Ray/Object/Material/Mesh/Utils .inl +
Camera.cpp + vendored GLM compiled with g++, see oracle/Makefile).

Runs only where /root/reference exists (this build container). The fixtures are data --
inputs and expected outputs -- and are what pins the plain-C oracle and the HIP path on
machines that do not have the reference (the GPU box).

    make -C oracle && python tools/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.pyoracle import Oracle  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def scene_arrays(o, s, prefix):
    d = {}
    n = o.scene_size(s)
    d[prefix + "n"] = np.array([n], np.int32)
    for i in range(n):
        ob = o.scene_object(s, i)
        for k, v in ob.items():
            d[f"{prefix}{i}_{k}"] = np.asarray(v)
    return d


def main():
    R = Oracle("ref")
    assert R.name() == "reference-host-build"
    os.makedirs(OUT, exist_ok=True)
    meta = load_meta()          # keys written by the other sections (make_film300) survive
    meta.update({"generator": "tools/make_golden.py", "oracle": R.name(),
                 "flags": "g++ -std=c++14 -O2 -ffp-contract=off -DNDEBUG"})
    meta.setdefault("hashes", {})

    # ---- scenes and cameras -------------------------------------------------------------
    scenes = {}
    d = {}
    for variant, param in ((0, 0), (1, 0), (2, 1)):
        s = R.scene_create(variant, param)
        scenes[(variant, param)] = s
        d.update(scene_arrays(R, s, f"v{variant}p{param}_"))
    cams = [(512, 512, 0.0, 0.0), (256, 256, 0.0, 0.0), (3840, 2160, 0.0, 0.0), (512, 512, 0.3, -0.2),
            (640, 480, -1.1, 0.4), (128, 96, 2.5, 1.0)]
    d["camera_args"] = np.array(cams, np.float64)
    d["cameras"] = np.stack([R.camera_default(int(a), int(b), c, e).as_array() for a, b, c, e in cams])
    np.savez_compressed(os.path.join(OUT, "scenes.npz"), **d)

    # ---- mode 1 (BASELINE.json configs[0]: 256x256 default scene) ---------------------------
    s0 = scenes[(0, 0)]
    cam = R.camera_default(256, 256)
    r = R.raytrace1(s0, 256, 256, cam)
    ids, cnt = np.unique(r["ids"], return_counts=True)
    meta["mode1_256_id_histogram"] = {int(i): int(c) for i, c in zip(ids, cnt)}
    meta["hashes"]["mode1_256_pos"] = sha(r["pos"]); meta["hashes"]["mode1_256_nor"] = sha(r["nor"])
    np.savez_compressed(os.path.join(OUT, "mode1_256.npz"), cam=cam.as_array(), ids=r["ids"].astype(np.int8),
                        ts=r["ts"], pbo=r["pbo"])
    # other shapes / scenes, smaller
    d = {}
    cases = [("v0_96x64_rot", (0, 0), 96, 64, (96, 64, 0.3, -0.2)),
             ("v1_64x64", (1, 0), 64, 64, (64, 64, 0.0, 0.0)),
             ("v2_64x64", (2, 1), 64, 64, (64, 64, 0.0, 0.0)),
             ("v0_33x17", (0, 0), 33, 17, (33, 17, -0.7, 0.1))]
    for name, key, w, h, cargs in cases:
        c = R.camera_default(*cargs)
        rr = R.raytrace1(scenes[key], w, h, c)
        d[name + "_cam"] = c.as_array(); d[name + "_ids"] = rr["ids"].astype(np.int8); d[name + "_ts"] = rr["ts"]
        d[name + "_pbo"] = rr["pbo"]; d[name + "_pos"] = rr["pos"]; d[name + "_nor"] = rr["nor"]
        d[name + "_shape"] = np.array([w, h, key[0], key[1]], np.int32)
    np.savez_compressed(os.path.join(OUT, "mode1_small.npz"), **d)

    # ---- mode 2: per-sample radiance and short accumulations ---------------------------------
    d = {}
    cases = [("v0_64x64", (0, 0), 64, 64, (64, 64, 0.0, 0.0), 11),
             ("v1_64x48_rot", (1, 0), 64, 48, (64, 48, 0.3, -0.2), 12),
             ("v0_33x17", (0, 0), 33, 17, (33, 17, -0.7, 0.1), 13)]
    for name, key, w, h, cargs, seed in cases:
        c = R.camera_default(*cargs)
        d[name + "_cam"] = c.as_array()
        d[name + "_shape"] = np.array([w, h, key[0], key[1], seed], np.int32)
        for rng in (0, 1):
            # single samples: iteration 1 and iteration 7 on a zero film
            for it in (1, 7):
                rr = R.pathtrace(scenes[key], w, h, c, seed, it, 1, rng_mode=rng)
                d[f"{name}_rng{rng}_sample{it}"] = rr["film"]
            rr = R.pathtrace(scenes[key], w, h, c, seed, 1, 6, rng_mode=rng)
            d[f"{name}_rng{rng}_film6"] = rr["film"]; d[f"{name}_rng{rng}_pbo6"] = rr["pbo"]
            d[f"{name}_rng{rng}_stats6"] = np.array([rr["stats"][k] for k in
                                                     ("live_segments", "miss", "light", "depthcap", "diff", "trans", "mirr", "paths")], np.int64)
            # strided subset (rank 1 of 2): iterations 2, 4, 6
            rr = R.pathtrace(scenes[key], w, h, c, seed, 2, 3, iter_stride=2, rng_mode=rng)
            d[f"{name}_rng{rng}_film_2_4_6"] = rr["film"]
    np.savez_compressed(os.path.join(OUT, "mode2_small.npz"), **d)
    # 512x512 headline config: iteration-1 sample hash + statistics of 4 iterations
    c512 = R.camera_default(512, 512)
    rr = R.pathtrace(s0, 512, 512, c512, 1, 1, 4)
    meta["mode2_512_seed1_4it_stats"] = rr["stats"]
    meta["hashes"]["mode2_512_seed1_film4"] = sha(rr["film"])
    meta["mode2_512_seed1_film4_mean"] = [float(x) for x in (rr["film"].mean(0) / 4)]

    # ---- unit-level known answers ---------------------------------
    rs = np.random.RandomState(1234)
    d = {}
    nray = 400
    org = rs.uniform(-4.5, 4.5, (nray, 3)).astype(np.float32)
    org[: nray // 2] = np.array([0, 0, 13], np.float32)          # half from the camera
    tgt = rs.uniform(-5, 5, (nray, 3)).astype(np.float32)
    dirs = tgt - org
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([org, dirs], 1).astype(np.float32)
    d["rays"] = rays
    for key, s in (("v0", s0), ("v1", scenes[(1, 0)])):
        t_all = np.zeros((nray, 9), np.float32); pos_all = np.zeros((nray, 9, 3), np.float32); nor_all = np.zeros((nray, 9, 3), np.float32)
        st = np.zeros(nray, np.float32); sid = np.zeros(nray, np.int32); sp = np.zeros((nray, 3), np.float32); sn = np.zeros((nray, 3), np.float32)
        for i in range(nray):
            for ob in range(9):
                t_all[i, ob], pos_all[i, ob], nor_all[i, ob] = R.intersect_object(s, ob, rays[i])
            st[i], sid[i], sp[i], sn[i] = R.intersect_scene(s, rays[i])
        d[key + "_obj_t"] = t_all; d[key + "_obj_pos"] = pos_all; d[key + "_obj_nor"] = nor_all
        d[key + "_scene_t"] = st; d[key + "_scene_id"] = sid; d[key + "_scene_pos"] = sp; d[key + "_scene_nor"] = sn
    nb = 300
    rd = rs.normal(size=(nb, 3)); rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    nor = rs.normal(size=(nb, 3)); nor = (nor / np.linalg.norm(nor, axis=1, keepdims=True)).astype(np.float32)
    rnd = rs.uniform(0, 1, (nb, 3)).astype(np.float32)
    d["b_rd"] = rd; d["b_nor"] = nor; d["b_rnd"] = rnd
    for mt, ior in ((0, 1.0), (1, 1.6), (1, 1.2), (2, 1.0)):
        d[f"bounce_{mt}_{ior}"] = np.stack([R.bounce(mt, ior, rd[i], nor[i], rnd[i]) for i in range(nb)])
    d["reflectance_1_1.6"] = np.array([R.reflectance(rd[i], nor[i], 1.0, 1.6) for i in range(nb)], np.float32)
    d["reflectance_1.6_1"] = np.array([R.reflectance(rd[i], nor[i], 1.6, 1.0) for i in range(nb)], np.float32)
    d["rand_vector"] = np.stack([R.rand_vector(float(rnd[i, 0]), float(rnd[i, 1])) for i in range(nb)])
    d["rand_point_disk"] = np.stack([R.rand_point_disk(*[float(x) for x in rnd[i]]) for i in range(nb)])
    d["rand_vector_hem"] = np.stack([R.rand_vector_hem(float(rnd[i, 0]), float(rnd[i, 1]), nor[i]) for i in range(nb)])
    cols = np.concatenate([rs.uniform(-0.2, 1.2, (200, 3)), [[0, 0, 0], [1, 1, 1], [0.5, 0.25, 0.999999], [1 - 2 ** -24, 2 ** -24, 0.1]]]).astype(np.float32)
    d["rgb_in"] = cols
    d["rgb_out"] = np.array([R.rgb_to_int(*[float(x) for x in c]) for c in cols], np.uint32)
    films = rs.uniform(0, 20, (200, 3)).astype(np.float32)
    d["tonemap_film"] = films; d["tonemap_iters"] = np.array([7.0], np.float32)
    d["tonemap_pbo"] = R.tonemap(films, 7.0)
    d["rand3_seed5"] = np.stack([R.rand3(5, i * 7919, 3, i % 4) for i in range(64)])
    np.savez_compressed(os.path.join(OUT, "units.npz"), **d)

    save_meta(meta)


def load_meta():
    p = os.path.join(OUT, "meta.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def save_meta(meta):
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def make_film300(R, P):
    """The film the benchmark renders -- BASELINE.json configs[1]: default scene, 512x512, seed 1, iterations 1..300,
    reference RNG reuse (raytrace.cu:162-177: the running sum over all iterations) -- from both reference builds
    (~2.5 minutes each, single-threaded). Stored: sha1 of the film and of the LDR image of the portable-math build
    (what the HIP path must equal byte for byte; bench.py checks it after its timed loop), the film's mean, the
    segment statistics, and -- as data -- the pixels in which the glibc build differs, with its values there, so that a
    film equal to the portable-math build's can be patched into the glibc build's and checked against ITS sha1."""
    w = h = 512
    n = 300
    out = {}
    for key, o in (("pm", P), ("glibc", R)):
        s = o.scene_create(0, 0)
        out[key] = o.pathtrace(s, w, h, o.camera_default(w, h), 1, 1, n)
    a, b = out["pm"], out["glibc"]
    diff = np.nonzero((a["film"] != b["film"]).any(1))[0].astype(np.int32)
    ldiff = np.nonzero(a["pbo"] != b["pbo"])[0].astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "mode2_film300.npz"), shape=np.array([w, h, 0, 0, 1, n], np.int32),
                        glibc_diff_idx=diff, glibc_diff_film=b["film"][diff], pm_diff_film=a["film"][diff],
                        glibc_ldr_diff_idx=ldiff, glibc_ldr_diff_val=b["pbo"][ldiff])
    meta = load_meta()
    meta.setdefault("hashes", {})
    meta["hashes"]["mode2_512_seed1_film300_pm"] = sha(a["film"])
    meta["hashes"]["mode2_512_seed1_ldr300_pm"] = sha(a["pbo"])
    meta["hashes"]["mode2_512_seed1_film300_glibc"] = sha(b["film"])
    meta["hashes"]["mode2_512_seed1_ldr300_glibc"] = sha(b["pbo"])
    meta["mode2_512_seed1_film300_pm_mean"] = [float(x) for x in (a["film"].astype(np.float64).mean(0) / n)]
    meta["mode2_512_seed1_300it_stats_pm"] = a["stats"]
    meta["mode2_512_seed1_300it_stats_glibc"] = b["stats"]
    meta["mode2_512_seed1_film300_glibc_vs_pm_pixels"] = int(len(diff))
    save_meta(meta)
    print("film300:", len(diff), "of", w * h, "pixels differ between the glibc and the portable-math reference builds;",
          "live segments", a["stats"]["live_segments"], "/", b["stats"]["live_segments"])


def make_full(R, P):
    """Per-sample radiance of the reference host build at the BASELINE.json sizes: configs[1] 512x512 (iterations 1
    and 7, both RNG modes; a 4-iteration film), configs[2] 1024x1024 glass icosahedron (iteration 1), configs[3]
    3840x2160 (iteration 1, the whole frame). One sample's radiance is a product of material colours (or zero): the
    4K frame is stored as a palette of its distinct RGB triples + one index per pixel (exact, ~1.2 MB compressed).

    R = the reference build as it is (glibc sinf/cosf/acosf/powf); P = the same sources with those four bound to
    oracle/pmath.h (oracle/pm_interpose.c), i.e. the reference's code on the arithmetic the HIP kernels use. Where P's
    output equals R's byte for byte only a flag is stored (`*_pm_equal`); else the differing pixels (`*_pm_diff_*`)."""
    d = {}
    sR = {0: R.scene_create(0, 0), 1: R.scene_create(1, 0)}
    sP = {0: P.scene_create(0, 0), 1: P.scene_create(1, 0)}

    def sample(key, variant, w, h, cam, it, n, rng):
        a = R.pathtrace(sR[variant], w, h, cam, 1, it, n, rng_mode=rng)
        b = P.pathtrace(sP[variant], w, h, cam, 1, it, n, rng_mode=rng)
        diff = np.nonzero(np.abs(a["film"] - b["film"]).max(1) > 0)[0].astype(np.int64)
        d[key + "_pm_diff_idx"] = diff
        d[key + "_pm_diff_val"] = b["film"][diff]
        return a, b

    c = R.camera_default(512, 512)
    d["c1_cam"] = c.as_array(); d["c1_shape"] = np.array([512, 512, 0, 0, 1], np.int32)
    for rng in (0, 1):
        for it in (1, 7):
            d[f"c1_rng{rng}_sample{it}"] = sample(f"c1_rng{rng}_sample{it}", 0, 512, 512, c, it, 1, rng)[0]["film"]
    a, b = sample("c1_rng0_film4", 0, 512, 512, c, 1, 4, 0)
    d["c1_rng0_film4"] = a["film"]; d["c1_rng0_pbo4"] = a["pbo"]; d["c1_rng0_pbo4_pm"] = b["pbo"]
    c = R.camera_default(1024, 1024)
    d["c2_cam"] = c.as_array(); d["c2_shape"] = np.array([1024, 1024, 1, 0, 1], np.int32)
    d["c2_rng0_sample1"] = sample("c2_rng0_sample1", 1, 1024, 1024, c, 1, 1, 0)[0]["film"]
    c = R.camera_default(3840, 2160)
    w, h = 3840, 2160
    d["c3_cam"] = c.as_array(); d["c3_shape"] = np.array([w, h, 0, 0, 1], np.int32)
    film = sample("c3_rng0_sample1", 0, w, h, c, 1, 1, 0)[0]["film"]
    pal, idx = np.unique(film.view(np.dtype((np.void, 12))).ravel(), return_inverse=True)
    assert len(pal) < 256
    d["c3_rng0_sample1_palette"] = pal.view(np.float32).reshape(-1, 3)
    d["c3_rng0_sample1_index"] = idx.astype(np.uint8)
    assert d["c3_rng0_sample1_palette"][d["c3_rng0_sample1_index"]].tobytes() == film.tobytes()
    for k in sorted(d):
        if k.endswith("_pm_diff_idx"):
            print(k, len(d[k]), "pixels differ between the glibc and the portable-math reference builds")
    np.savez_compressed(os.path.join(OUT, "mode2_full.npz"), **d)


def make_picks(R):
    """main.cpp:180-208 through the reference build: picking rays of a 512x512 window (WINDOW_W/H, common.h:33-34) on
    a 23x23 grid of window positions plus the corners, default and rotated camera, default and glass scene."""
    d = {}
    xs = list(range(0, 512, 23)) + [511, 512]
    xy = np.array([(x, y) for y in xs for x in xs], np.int32)
    d["xy"] = xy; d["window"] = np.array([512, 512], np.int32)
    for name, variant, cargs in (("v0", 0, (512, 512, 0.0, 0.0)), ("v1_rot", 1, (512, 512, 0.3, -0.2))):
        s = R.scene_create(variant, 0)
        c = R.camera_default(*cargs)
        out = [R.pick(s, c, 512, 512, int(x), int(y)) for x, y in xy]
        d[name + "_cam"] = c.as_array(); d[name + "_variant"] = np.array([variant], np.int32)
        d[name + "_ray"] = np.stack([o[0] for o in out]); d[name + "_t"] = np.array([o[1] for o in out], np.float32)
        d[name + "_id"] = np.array([o[2] for o in out], np.int32)
        d[name + "_pos"] = np.stack([o[3] for o in out]); d[name + "_nor"] = np.stack([o[4] for o in out])
    np.savez_compressed(os.path.join(OUT, "picks.npz"), **d)


if __name__ == "__main__":
    which = sys.argv[1:]
    if not which:
        main()
    if not which or "full" in which or "picks" in which or "film300" in which:
        Rr = Oracle("ref")
        assert Rr.name() == "reference-host-build"
        if not which or "full" in which or "film300" in which:
            Pp = Oracle("ref_pm")
            assert Pp.name() == "reference-host-build+portable-math"
            if not which or "full" in which:
                make_full(Rr, Pp)
            if not which or "film300" in which:
                make_film300(Rr, Pp)
        if not which or "picks" in which:
            make_picks(Rr)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))
