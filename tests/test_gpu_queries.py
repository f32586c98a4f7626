"""Ray queries through the C ABI (trc_intersect_rays / trc_intersect_rays_device / trc_pick): the interactive
path's picking (reference src/main.cpp:180-208 -> Ray::intersectScene, Ray.inl:66-84) on the GPU, against
known answers recorded from the REFERENCE host build (tests/golden/units.npz, picks.npz) and against the
live oracle. Everything is bit-exact. Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tracer_amd
    tracer_amd.load_library()
    return tracer_amd


def make_tracer(ta, variant=0, param=0, w=64, h=64):
    sc = ta.Scene(variant, param)
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    tr._scene = sc
    return tr


@pytest.mark.parametrize("key,variant", [("v0", 0), ("v1", 1)])
def test_reference_known_answer_rays_scene(ta, golden, key, variant):
    """The 400 rays of units.npz through Ray::intersectScene of the reference build: id, t, position, normal."""
    g = golden["units"]
    tr = make_tracer(ta, variant)
    r = tr.intersect_rays(g["rays"])
    assert np.array_equal(r["ids"], g[key + "_scene_id"])
    assert r["ts"].tobytes() == g[key + "_scene_t"].tobytes()
    assert r["pos"].tobytes() == g[key + "_scene_pos"].tobytes()
    assert r["nor"].tobytes() == g[key + "_scene_nor"].tobytes()
    assert (r["ids"] >= 0).sum() > 300 and len(set(r["ids"].tolist())) >= 8
    tr.close()


@pytest.mark.parametrize("key,variant", [("v0", 0), ("v1", 1)])
def test_reference_known_answer_rays_per_object(ta, golden, key, variant):
    """Ray::intersect(ray, obj) (Ray.inl:86-100) of the reference build, object by object: each of the nine objects
    uploaded as a scene of its own. intersectScene admits a hit iff its t > 0 and its world distance is a number
    (Ray.inl:72-75): that rules out the MESH's "AABB hit, no triangle" result, t = FLT_MAX with a NaN normal."""
    g = golden["units"]
    rays = g["rays"]
    sc = ta.Scene(variant)
    hits_seen = 0
    for ob in range(9):
        tr = ta.Tracer(8, 8)
        tr.upload_scene(sc.objects_ptr + 168 * ob, 1)
        r = tr.intersect_rays(rays)
        t_ref = g[key + "_obj_t"][:, ob]
        admitted = (t_ref > 0) & (t_ref < np.float32(3.0e38))
        assert np.array_equal(r["ids"], np.where(admitted, 0, -1).astype(np.int32)), ob
        assert r["ts"][admitted].tobytes() == t_ref[admitted].tobytes(), ob
        assert r["pos"][admitted].tobytes() == g[key + "_obj_pos"][:, ob][admitted].tobytes(), ob
        assert r["nor"][admitted].tobytes() == g[key + "_obj_nor"][:, ob][admitted].tobytes(), ob
        # a miss is a default Hit (Ray.inl:30): t = -1, zero vectors
        assert np.all(r["ts"][~admitted] == -1.0) and not r["pos"][~admitted].any() and not r["nor"][~admitted].any()
        hits_seen += int(admitted.sum())
        tr.close()
    assert hits_seen > 400


@pytest.mark.parametrize("name", ["v0", "v1_rot"])
def test_picks_vs_reference_fixture(ta, golden, name):
    """main.cpp:180-208 recorded from the reference build: every window position of the fixture through trc_pick
    (host-side ray construction) and the recorded rays through the batch entry."""
    g = golden["picks"]
    W, H = [int(x) for x in g["window"]]
    tr = make_tracer(ta, int(g[name + "_variant"][0]))
    cam = ta.Camera.from_array(g[name + "_cam"])
    for k, (x, y) in enumerate(g["xy"]):
        hid, t, pos, nor = tr.pick(cam, W, H, int(x), int(y))
        assert hid == int(g[name + "_id"][k]), (x, y)
        assert np.float32(t).tobytes() == g[name + "_t"][k].tobytes()
        assert pos.tobytes() == g[name + "_pos"][k].tobytes() and nor.tobytes() == g[name + "_nor"][k].tobytes()
    r = tr.intersect_rays(g[name + "_ray"])
    assert np.array_equal(r["ids"], g[name + "_id"]) and r["ts"].tobytes() == g[name + "_t"].tobytes()
    assert r["pos"].tobytes() == g[name + "_pos"].tobytes() and r["nor"].tobytes() == g[name + "_nor"].tobytes()
    tr.close()


def test_pick_equals_the_mode1_frame(ta):
    """A pick at window position (x, y) is the mode-1 pixel (x, H - y) (main.cpp:196 vs raytrace.cu:40): same id,
    t, position, normal as trc_read_hits of a whole frame, for a window that is not square."""
    w, h = 96, 64
    tr = make_tracer(ta, 0, 0, w, h)
    cam = ta.camera_default(w, h, 0.3, -0.2)
    tr.raytrace1(cam)
    hits = tr.read_hits()
    for x, y in [(0, 64), (1, 63), (48, 32), (95, 1), (17, 40), (60, 5)]:
        hid, t, pos, nor = tr.pick(cam, w, h, x, y)
        idx = (h - y) * w + x
        assert hid == hits["ids"][idx] and np.float32(t).tobytes() == hits["ts"][idx].tobytes()
        assert pos.tobytes() == hits["pos"][idx].tobytes() and nor.tobytes() == hits["nor"][idx].tobytes()
    tr.close()


@pytest.mark.parametrize("variant,param,n", [(0, 0, 10007), (2, 3, 5000), (2, 5, 3001)])
def test_random_ray_batches_vs_oracle(ta, oracle_c, variant, param, n):
    """Seeded random rays (not from the camera; un-normalised directions too, which intersectScene takes as they
    are) against the live oracle: ragged batch sizes, a small MESH inside closest_hit, a big MESH (20480 faces)
    through the brute-force pass with the ray index as its slot."""
    rs = np.random.RandomState(100 + variant + param)
    org = rs.uniform(-4.8, 4.8, (n, 3)).astype(np.float32)
    org[: n // 3] = np.array([0, 0, 13], np.float32)
    d = (rs.uniform(-5, 5, (n, 3)).astype(np.float32) - org)
    d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)
    rays = np.concatenate([org, d.astype(np.float32)], 1).astype(np.float32)
    tr = make_tracer(ta, variant, param)
    r = tr.intersect_rays(rays)
    s = oracle_c.scene_create(variant, param)
    step = 1 if n <= 5000 else 3
    for i in range(0, n, step):
        t, hid, pos, nor = oracle_c.intersect_scene(s, rays[i])
        assert hid == r["ids"][i], i
        assert np.float32(t).tobytes() == r["ts"][i].tobytes(), i
        assert pos.tobytes() == r["pos"][i].tobytes() and nor.tobytes() == r["nor"][i].tobytes(), i
    assert (r["ids"] >= 0).mean() > 0.5
    tr.close()


def test_device_resident_batch_and_edge_cases(ta, oracle_c):
    """trc_intersect_rays_device on torch tensors (no host copies, the context's stream), an empty batch, null
    outputs, and the error returns."""
    import torch
    g_rays = np.random.RandomState(7).uniform(-4, 4, (777, 6)).astype(np.float32)
    stream = torch.cuda.Stream()
    sc = ta.Scene(0)
    tr = ta.Tracer(32, 32, stream=stream.cuda_stream)
    with pytest.raises(ta.TracerError):
        tr.intersect_rays(g_rays)                       # no scene yet
    tr.upload_scene(sc.objects_ptr, sc.size)
    host = tr.intersect_rays(g_rays)
    with torch.cuda.stream(stream):
        rays_d = torch.from_numpy(g_rays).cuda()
        ids_d = torch.full((777,), -7, dtype=torch.int32, device="cuda")
        ts_d = torch.zeros(777, dtype=torch.float32, device="cuda")
        tr.intersect_rays_device(rays_d.data_ptr(), 777, ids_d.data_ptr(), ts_d.data_ptr())     # pos / nor not wanted
        tr.intersect_rays_device(rays_d.data_ptr(), 0, ids_d.data_ptr(), ts_d.data_ptr())       # empty batch: no-op
    tr.synchronize()
    assert np.array_equal(ids_d.cpu().numpy(), host["ids"]) and ts_d.cpu().numpy().tobytes() == host["ts"].tobytes()
    empty = tr.intersect_rays(np.zeros((0, 6), np.float32))
    assert empty["ids"].size == 0
    L = tr.L
    assert L.trc_intersect_rays(tr.ctx, None, 5, None, None, None, None) != 0
    assert L.trc_pick(tr.ctx, C.byref(ta.camera_default(32, 32)), 0, 32, 1, 1, None, None, None, None) != 0
    i = C.c_int32(99)
    assert L.trc_pick(tr.ctx, C.byref(ta.camera_default(32, 32)), 32, 32, 16, 16, C.byref(i), None, None, None) == 0
    assert i.value == oracle_c.pick(oracle_c.scene_create(0), oracle_c.camera_default(32, 32), 32, 32, 16, 16)[2]
    tr.close()


def test_degenerate_rays_vs_oracle(ta, oracle_c):
    """Rays with a NaN, an infinity, a zero, a denormal or a huge coordinate, and zero directions: what Ray::intersectScene
    makes of them (mostly "no hit", sometimes a hit with a NaN in it) is reproduced -- ids exactly, floats up to the
    payload of a NaN. The (ray, object) pre-test must keep every pair of such a ray (its comparisons are false for a NaN,
    and a direction that is not of unit length switches it off)."""
    rs = np.random.RandomState(11)
    n = 1536
    org = rs.uniform(-4.8, 4.8, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], 1).astype(np.float32)
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, 1e-42, 3e38, -0.0], np.float32)
    rays[np.arange(n), rs.randint(0, 6, n)] = vals[rs.randint(0, len(vals), n)]
    rays[::7, 3:] = 0.0
    for variant in (0, 2):
        tr = make_tracer(ta, variant, 2)
        r = tr.intersect_rays(rays)
        s = oracle_c.scene_create(variant, 2)
        for i in range(n):
            t, hid, pos, nor = oracle_c.intersect_scene(s, rays[i])
            assert hid == r["ids"][i], (variant, i, rays[i])
            assert np.array_equal(np.float32(t), r["ts"][i], equal_nan=True), (variant, i)
            assert np.array_equal(pos, r["pos"][i], equal_nan=True) and np.array_equal(nor, r["nor"][i], equal_nan=True), (variant, i)
        tr.close()
