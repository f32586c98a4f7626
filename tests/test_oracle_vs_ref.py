"""Live comparison of the plain-C oracle with the reference-linked host build
(oracle/_ref/libtracer_ref.so) on inputs the fixtures do not cover. Skipped where the
reference build is absent."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("variant,param", [(0, 0), (1, 0), (2, 2), (2, 4)])
def test_scene_and_mode1(oracle_c, oracle_ref, variant, param):
    sr = oracle_ref.scene_create(variant, param); sc = oracle_c.scene_create(variant, param)
    assert oracle_ref.scene_size(sr) == oracle_c.scene_size(sc)
    for i in range(oracle_c.scene_size(sc)):
        a = oracle_ref.scene_object(sr, i); b = oracle_c.scene_object(sc, i)
        for k in a:
            assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), (i, k)
    for w, h, rx, ry in ((160, 120, 0.0, 0.0), (100, 100, 0.9, -0.35)):
        cr = oracle_ref.camera_default(w, h, rx, ry); cc = oracle_c.camera_default(w, h, rx, ry)
        assert bytes(cr) == bytes(cc)
        a = oracle_ref.raytrace1(sr, w, h, cr); b = oracle_c.raytrace1(sc, w, h, cc, nthreads=4)
        for k in a:
            assert a[k].tobytes() == b[k].tobytes(), (variant, w, h, k)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("rng", [0, 1])
def test_mode2(oracle_c, oracle_ref, variant, rng):
    sr = oracle_ref.scene_create(variant); sc = oracle_c.scene_create(variant)
    cam = oracle_ref.camera_default(96, 80, 0.2, 0.1)
    a = oracle_ref.pathtrace(sr, 96, 80, cam, 99, 1, 5, rng_mode=rng)
    b = oracle_c.pathtrace(sc, 96, 80, cam, 99, 1, 5, rng_mode=rng, nthreads=4)
    assert a["film"].tobytes() == b["film"].tobytes()
    assert np.array_equal(a["pbo"], b["pbo"])
    assert a["stats"] == b["stats"]
    # resume + stride
    a = oracle_ref.pathtrace(sr, 96, 80, cam, 99, 3, 4, iter_stride=3, rng_mode=rng, film=a["film"].copy())
    b = oracle_c.pathtrace(sc, 96, 80, cam, 99, 3, 4, iter_stride=3, rng_mode=rng, film=b["film"].copy(), nthreads=4)
    assert a["film"].tobytes() == b["film"].tobytes()


def test_scene_edits(oracle_c, oracle_ref):
    """Object::translate / rotate / scale recompute the inverse identically (interactive path)."""
    sr = oracle_ref.scene_create(0); sc = oracle_c.scene_create(0)
    for o, s in ((oracle_ref, sr), (oracle_c, sc)):
        o.scene_translate(s, 6, [0.3, 0.1, -0.2]); o.scene_rotate(s, 8, 33.0, [0.0, 1.0, 0.0]); o.scene_scale(s, 7, [1.1, 0.9, 1.0])
    for i in (6, 7, 8):
        a = oracle_ref.scene_object(sr, i, with_mesh=False); b = oracle_c.scene_object(sc, i, with_mesh=False)
        assert a["matrix"].tobytes() == b["matrix"].tobytes() and a["matrixi"].tobytes() == b["matrixi"].tobytes()
    cam = oracle_c.camera_default(64, 64)
    a = oracle_ref.raytrace1(sr, 64, 64, cam); b = oracle_c.raytrace1(sc, 64, 64, cam)
    assert a["ts"].tobytes() == b["ts"].tobytes() and np.array_equal(a["ids"], b["ids"])


@pytest.mark.parametrize("seed", [1, 2])
def test_random_cluttered_scene(oracle_c, oracle_ref, product_lib, tmp_path, seed):
    """Beyond the reference's one scene: 29 objects of every primitive type with non-uniform scales, arbitrary
    rotations, all materials, three meshes, built by the product's scene-file front end and handed to both CPU
    checkers as the Object[] it is (orc_scene_wrap). The C restatement must still equal the reference build."""
    import tracer_amd as ta
    from scene_gen import clutter_scene
    sc = ta.Scene(path=clutter_scene(str(tmp_path / "c.scene"), seed), aspect=(80, 60))
    assert sc.size == 6 + 12 + 6 + 3 + 3
    sr = oracle_ref.scene_wrap(sc.objects_ptr, sc.size); s = oracle_c.scene_wrap(sc.objects_ptr, sc.size)
    from oracle.pyoracle import Camera as OCam
    cam = OCam.from_array(sc.camera.as_array())
    a = oracle_ref.raytrace1(sr, 80, 60, cam); b = oracle_c.raytrace1(s, 80, 60, cam, nthreads=4)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
    assert len(np.unique(a["ids"])) > 15
    for rng in (0, 1):
        a = oracle_ref.pathtrace(sr, 80, 60, cam, 7, 1, 3, rng_mode=rng)
        b = oracle_c.pathtrace(s, 80, 60, cam, 7, 1, 3, rng_mode=rng, nthreads=4)
        assert a["film"].tobytes() == b["film"].tobytes() and np.array_equal(a["pbo"], b["pbo"])
        assert a["stats"] == b["stats"] and b["stats"]["trans"] > 0 and b["stats"]["mirr"] > 0


@pytest.mark.parametrize("lens", [0.0, 0.7])
def test_axis_aligned_scene(oracle_c, oracle_ref, product_lib, tmp_path, lens):
    """The scene of test_gpu_parity.py::test_axis_aligned_objects_every_pipeline (matrices full of exact zeros,
    zero translations, a sphere at the origin): the C restatement against the reference build, so that the
    GPU's short matrix products are checked against the reference's own arithmetic."""
    import tracer_amd as ta
    from scene_gen import aligned_scene
    from oracle.pyoracle import Camera as OCam
    sc = ta.Scene(path=aligned_scene(str(tmp_path / "a.scene"), lens), aspect=(64, 64))
    sr = oracle_ref.scene_wrap(sc.objects_ptr, sc.size); s = oracle_c.scene_wrap(sc.objects_ptr, sc.size)
    cam = OCam.from_array(sc.camera.as_array())
    a = oracle_ref.raytrace1(sr, 64, 64, cam); b = oracle_c.raytrace1(s, 64, 64, cam, nthreads=4)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
    for rng in (0, 1):
        a = oracle_ref.pathtrace(sr, 64, 64, cam, 5, 1, 4, rng_mode=rng)
        b = oracle_c.pathtrace(s, 64, 64, cam, 5, 1, 4, rng_mode=rng, nthreads=4)
        assert a["film"].tobytes() == b["film"].tobytes() and a["stats"] == b["stats"]
        assert b["stats"]["trans"] > 0 and b["stats"]["mirr"] > 0 and b["stats"]["light"] > 0


@pytest.mark.parametrize("variant", [0, 1])
def test_portable_math_reference_build(oracle_c, oracle_ref, variant):
    """oracle/_ref/libtracer_ref_pm.so = the reference's own sources with sinf / cosf / sincosf / acosf / powf bound
    to oracle/pmath.h (oracle/pm_interpose.c). The C restatement in portable-math mode -- the arithmetic the HIP
    kernels are bit-identical to -- must equal it on film, LDR bytes, statistics and the sampling helpers; the
    plain build must NOT equal it on the helpers (the interposition is real)."""
    from oracle import pyoracle
    if not pyoracle.available("ref_pm"):
        pytest.skip("oracle/_ref/libtracer_ref_pm.so not built")
    P = pyoracle.Oracle("ref_pm")
    assert P.name() == "reference-host-build+portable-math"
    sp = P.scene_create(variant); sc = oracle_c.scene_create(variant)
    cam = P.camera_default(112, 80, -0.4, 0.15)
    oracle_c.set_math_mode(pyoracle.Oracle.MATH_PORTABLE)
    try:
        for rng in (0, 1):
            a = P.pathtrace(sp, 112, 80, cam, 21, 1, 5, rng_mode=rng)
            b = oracle_c.pathtrace(sc, 112, 80, cam, 21, 1, 5, rng_mode=rng, nthreads=4)
            assert a["film"].tobytes() == b["film"].tobytes() and np.array_equal(a["pbo"], b["pbo"]) and a["stats"] == b["stats"]
        rs = np.random.RandomState(3)
        differs = 0
        for r in rs.uniform(0, 1, (200, 3)).astype(np.float32):
            v = P.rand_vector(float(r[0]), float(r[1]))
            assert v.tobytes() == oracle_c.rand_vector(float(r[0]), float(r[1])).tobytes()
            assert P.rand_point_disk(*[float(x) for x in r]).tobytes() == oracle_c.rand_point_disk(*[float(x) for x in r]).tobytes()
            differs += v.tobytes() != oracle_ref.rand_vector(float(r[0]), float(r[1])).tobytes()
        assert differs > 20
        films = rs.uniform(0, 20, (100, 3)).astype(np.float32)
        assert np.array_equal(P.tonemap(films, 7.0), oracle_c.tonemap(films, 7.0))
    finally:
        oracle_c.set_math_mode(pyoracle.Oracle.MATH_LIBM)


@pytest.mark.parametrize("name", ["degenerate_transforms", "no_planes_nested_spheres", "two_big_meshes", "stress"])
def test_corner_scenes_and_degenerate_rays(oracle_c, oracle_ref, product_lib, tmp_path, name):
    """The scenes that tests/test_gpu_pairs.py uses to corner the GPU's (ray, object) pairs -- singular / tiny / huge
    transforms, the camera inside a sphere, big meshes, objects at the limits of the bounding-sphere rules -- and rays with
    NaN, infinite, zero, denormal and huge coordinates through intersectScene: the C restatement (what the GPU is compared
    with) against the reference build, so that also these corners are pinned by the reference's own arithmetic (among them
    the mesh "hit" at t = FLT_MAX of Ray.inl:171-185 for a direction that overflows to the zero vector)."""
    import tracer_amd as ta
    from oracle.pyoracle import Camera as OCam
    from scene_gen import stress_scene
    from test_gpu_pairs import CORNER_SCENES
    data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tracer_amd", "data")
    if name == "stress":
        path = stress_scene(str(tmp_path / "s.scene"), 1)
    else:
        path = str(tmp_path / (name + ".scene"))
        text = CORNER_SCENES[name]
        open(path, "w").write(text % {"data": data} if "%(data)s" in text else text)
    w, h = 48, 32
    sc = ta.Scene(path=path, aspect=(w, h))
    sr = oracle_ref.scene_wrap(sc.objects_ptr, sc.size); s = oracle_c.scene_wrap(sc.objects_ptr, sc.size)
    cam = OCam.from_array(sc.camera.as_array())
    a = oracle_ref.raytrace1(sr, w, h, cam); b = oracle_c.raytrace1(s, w, h, cam, nthreads=4)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    a = oracle_ref.pathtrace(sr, w, h, cam, 3, 1, 2); b = oracle_c.pathtrace(s, w, h, cam, 3, 1, 2, nthreads=4)
    assert np.array_equal(a["film"], b["film"], equal_nan=True) and a["stats"] == b["stats"]
    rs = np.random.RandomState(5)
    n = 600
    rays = np.concatenate([rs.uniform(-5, 5, (n, 3)), rs.normal(size=(n, 3))], 1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, 1e-42, 3e38, -0.0], np.float32)
    rays[np.arange(n), rs.randint(0, 6, n)] = vals[rs.randint(0, len(vals), n)]
    rays[::7, 3:] = 0.0
    for i in range(n):
        ta_, ida, pa, na = oracle_ref.intersect_scene(sr, rays[i]); tb, idb, pb, nb = oracle_c.intersect_scene(s, rays[i])
        assert ida == idb and np.array_equal(np.float32(ta_), np.float32(tb), equal_nan=True), (i, rays[i])
        assert np.array_equal(pa, pb, equal_nan=True) and np.array_equal(na, nb, equal_nan=True), (i, rays[i])
