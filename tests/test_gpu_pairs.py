"""The (ray, object) pre-test of intersect.h (bounding spheres from bound.h) must be CONSERVATIVE: it may only call a pair
a certain miss if the exact test of Ray.inl rejects it (and, for a mesh, if the ray fails the AABB, so that the count of
triangle tests stays the reference's). Checked with the library's checking builds (tracer_amd/variants, built by
__graft_entry__.build()): every pair is tested exactly and carries the pre-test's verdict.
  libtracer_pairs_check1.so  counts the pairs called a certain miss that hit after all        -> 0 in every scenario
  libtracer_pairs_check2.so  counts the pairs called a certain miss                           -> most pairs of most scenes
and every output (mode-1 hits, films of several pipelines, caller ray batches) of both equals the product build's bit for
bit. Scenarios: the Cornell variants, random cluttered scenes, the axis-aligned scene, and scenes with tiny, huge, far-away,
badly conditioned and skewed objects (tests/scene_gen.py: stress_scene); ray batches with non-unit directions, origins up
to 10^6 away, axis-aligned directions and origins on surfaces."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
VARIANTS = os.path.join(REPO, "tracer_amd", "variants")


def run_worker(lib, tmp):
    env = dict(os.environ)
    if lib:
        env["TRC_LIB"] = os.path.join(VARIANTS, lib)
        assert os.path.exists(env["TRC_LIB"]), "checking build missing: run __graft_entry__.build()"
    else:
        env.pop("TRC_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "pairs_check_worker.py"), str(tmp)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_pre_test_is_conservative(tmp_path):
    product = run_worker(None, tmp_path)
    viol = run_worker("libtracer_pairs_check1.so", tmp_path)
    skipped = run_worker("libtracer_pairs_check2.so", tmp_path)
    assert set(product) == set(viol) == set(skipped) and len(product) >= 10
    for name in product:
        assert viol[name]["tri_tests"] == 0, (name, viol[name]["tri_tests"])              # never a hit behind a "certain miss"
        assert viol[name]["digest"] == product[name]["digest"], name                       # same bits with every pair tested
        assert skipped[name]["digest"] == product[name]["digest"], name
        assert skipped[name]["tri_tests"] > 0, name                                         # and the pre-test does reject pairs


# ---------------------------------------------------------------------------------------------------------------------
# scenes at the corners of the pair machinery, bit for bit against the oracle
# ---------------------------------------------------------------------------------------------------------------------
WALLS = """mesh wall plane
object wall
  scale 10 10 10
  translate 0 -5 0
object wall
  rotate 180 1 0 0
  scale 10 10 10
  translate 0 5 0
  color 1 1 1
  emit 2
object wall
  rotate 90 1 0 0
  scale 10 10 10
  translate 0 0 -5
  color 0.5 0.5 1
"""
CORNER_SCENES = {
    # no sphere, cube or mesh at all: the pair list stays empty
    "planes_only": "camera fov 30 zoom -13 focal 13 lens 0.3\n" + WALLS,
    # no plane at all; the camera sits INSIDE the big sphere, so all 64 rays of a wavefront keep their pair with it (a full
    # list), the mirror ball inside adds pairs that no longer fit: two passes on every trip
    "no_planes_nested_spheres": """camera fov 40 zoom -3 focal 3 lens 0.1
mesh ball sphere
mesh box cube
object ball
  scale 30 30 30
  color 0.9 0.9 0.9
  emit 1.5
object ball
  scale 2 2 2
  translate 0.5 0.2 0
  material mirr
object box
  scale 1 1 1
  rotate 30 0.3 1 0.2
  translate -1.5 -0.5 0.5
  color 0.9 0.5 0.3
""",
    # one object of each kind alone with a floor and a light
    "single_sphere": "camera fov 30 zoom -13 focal 13 lens 0\n" + WALLS + "mesh ball sphere\nobject ball\n  scale 3 3 3\n  material trans\n  ior 1.5\n",
    "single_cube": "camera fov 30 zoom -13 focal 13 lens 0\n" + WALLS + "mesh box cube\nobject box\n  scale 2 3 2\n  rotate 40 0 1 0\n  translate 1 -2 0\n",
    "single_mesh": "camera fov 30 zoom -13 focal 13 lens 0\n" + WALLS + "mesh ico obj %(data)s/icosahedron.obj\nobject ico\n  scale 2.5\n  rotate 20 1 0 0\n  material mirr\n",
    # transforms that are singular, tiny or huge: inverses with inf / NaN entries (no bound: every ray keeps its pair), products
    # that overflow or underflow in the object-space re-normalisation
    "degenerate_transforms": "camera fov 30 zoom -13 focal 13 lens 0.2\n" + WALLS + """mesh ball sphere
mesh box cube
mesh ico obj %(data)s/icosahedron.obj
object ball
  scale 2 0 2
  translate -2 0 0
object ball
  scale 1e-20 1e-20 1e-20
  translate 1 1 0
object ball
  scale 1e20 1e20 1e20
  material trans
  ior 1.5
object box
  scale 3 0 1
  rotate 30 0 1 1
  translate 2 -1 0
object ico
  scale 1e-19
  translate 0 -2 1
object ico
  scale 2 0 2
  rotate 45 1 0 0
  translate -1 2 -1
  material mirr
object box
  scale 1e19 1 1
  translate 0 3 0
""",
    # a MESH scaled below 5e-20: |m_matrixi d|^2 overflows, the object-space direction is the zero vector, and the slab test
    # of Ray.inl:161-167 passes for every origin that lies below bmax on all three axes -- here the camera and most of the
    # box: the reference reports the mesh "hit" at distance 0 with t = FLT_MAX and a NaN normal (Ray.inl:171-185), so it
    # wins every such ray. Its bounding sphere must not cull those pairs (bound.h).
    "tiny_mesh_overflow": "camera fov 30 zoom -13 focal 13 lens 0.2\n" + WALLS + """mesh ico obj %(data)s/icosahedron.obj
mesh ball sphere
object ico
  scale 2e-20
  translate 6 6 14
object ico
  scale 4e-20
  translate 3 -2 -1
  material mirr
object ball
  scale 3 3 3
  translate -1 -2 0
""",
    # two BIG meshes (1280 faces each: their brute-force pass runs ahead of the closest-hit kernel in the launch-per-bounce
    # pipelines and in mode 1, the pairs then read its keys with the owning lane's ray slot) next to a small one
    "two_big_meshes": "camera fov 30 zoom -13 focal 13 lens 0.2\n" + WALLS + """mesh ico obj %(data)s/icosahedron.obj
mesh ball3 icosphere ico 3
object ball3
  scale 1.6
  rotate 20 0.3 1 0.1
  translate -2 -1 0
  material mirr
object ball3
  scale 1.2 2 1.2
  translate 2 0 1
  material trans
  ior 1.5
object ico
  scale 1.5
  translate 0 -3 2
""",
    # eight meshes side by side (several meshes per pass) and nothing else but the floor and the light
    "many_meshes": "camera fov 30 zoom -13 focal 13 lens 0.2\n" + WALLS + "mesh ico obj %(data)s/icosahedron.obj\nmesh ball2 icosphere ico 1\n" + "".join(
        "object %s\n  scale %g\n  rotate %d 0.2 1 0.3\n  translate %g %g %g\n  material %s\n" % (("ico", "ball2")[k % 2], 0.8 + 0.2 * k, 25 * k,
                                                                                          -3.5 + k, -2.0 + 0.5 * (k % 3), -1.0 + 0.6 * k, ("diff", "mirr", "trans")[k % 3]) +
        ("  ior 1.5\n" if k % 3 == 2 else "") for k in range(8)),
}


@pytest.mark.parametrize("name", sorted(CORNER_SCENES))
def test_corner_scenes_every_pipeline(name, oracle_c, tmp_path):
    import numpy as np
    import tracer_amd as ta
    from oracle.pyoracle import Oracle, Camera as OCam
    data = os.path.join(REPO, "tracer_amd", "data")
    path = tmp_path / (name + ".scene")
    path.write_text(CORNER_SCENES[name] % {"data": data} if "%(data)s" in CORNER_SCENES[name] else CORNER_SCENES[name])
    o = oracle_c
    o.set_math_mode(Oracle.MATH_PORTABLE)
    try:
        w, h = 96, 64
        sc = ta.Scene(path=str(path), aspect=(w, h))
        s = o.scene_wrap(sc.objects_ptr, sc.size)
        ocam = OCam.from_array(sc.camera.as_array())
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        tr.raytrace1(sc.camera)
        hits = tr.read_hits(); ref = o.raytrace1(s, w, h, ocam, nthreads=8)
        assert np.array_equal(hits["ids"], ref["ids"])
        for k in ("ts", "pos", "nor"):
            assert hits[k].tobytes() == ref[k].tobytes(), k
        for rng in (0, 1):
            pref = o.pathtrace(s, w, h, ocam, 3, 1, 5, rng_mode=rng, nthreads=8)
            for pipe in (0, 1, 2, 3):
                tr.set_pipeline(pipe); tr.set_rng_mode(rng); tr.reset_counters()
                tr.pathtrace(sc.camera, 3, 1, 5)
                assert tr.read_film().tobytes() == pref["film"].tobytes(), (rng, pipe)
                assert tr.counters()["live_segments"] == pref["stats"]["live_segments"]
        tr.close()
    finally:
        o.set_math_mode(Oracle.MATH_LIBM)
