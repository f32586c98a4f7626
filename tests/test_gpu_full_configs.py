"""BASELINE.json configs[2], [3], [4] at their FULL sizes on the GPU, checked through what does not
depend on the size: a few whole iterations bit for bit against the oracle, sub-sampled pixels of a
mode-1 frame against a small oracle frame with the same rays, resume == one call, pipelines equal,
and the sample partition that the multi-GPU run uses summing to the single-GPU film.
(The small-size, every-pixel parity of the same scenes is in test_gpu_parity.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tracer_amd
    tracer_amd.load_library()
    return tracer_amd


@pytest.fixture(scope="module")
def portable_oracle(oracle_c):
    from oracle.pyoracle import Oracle
    oracle_c.set_math_mode(Oracle.MATH_PORTABLE)
    yield oracle_c
    oracle_c.set_math_mode(Oracle.MATH_LIBM)


def make_tracer(ta, w, h, variant=0, param=0):
    sc = ta.Scene(variant, param)
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    tr._scene = sc
    return tr


def test_config3_glass_icosahedron_dof_1024x1024_1024spp(ta, portable_oracle):
    """configs[2]: glass (TRANS, n = 1.6) icosahedron, thin lens 1.0 / focal 13, 1024x1024, 1024 samples."""
    o = portable_oracle
    w = h = 1024
    tr = make_tracer(ta, w, h, 1, 0)
    cam = ta.camera_default(w, h)
    assert cam.lensRadius == 1.0 and cam.focalDist == 13.0
    # two whole iterations against the oracle: film and segment count bit-exact
    s = o.scene_create(1)
    tr.reset_counters()
    tr.pathtrace(cam, 1, 1, 2)
    ref = o.pathtrace(s, w, h, o.camera_default(w, h), 1, 1, 2, nthreads=16)
    assert tr.read_film().tobytes() == ref["film"].tobytes()
    assert tr.counters()["live_segments"] == ref["stats"]["live_segments"]
    assert ref["stats"]["trans"] > 0
    # the full 1024 samples: one call == two resumed calls, work queue == fused kernel
    tr.reset_counters()
    tr.pathtrace(cam, 1, 1, 1024)
    full = tr.read_film(); ldr = tr.read_ldr(); c = tr.counters()
    assert c["paths"] == 1024 * w * h and 3.0 < c["live_segments"] / c["paths"] < 4.5
    tr.pathtrace(cam, 1, 1, 512); tr.pathtrace(cam, 1, 513, 512)
    assert tr.read_film().tobytes() == full.tobytes() and np.array_equal(tr.read_ldr(), ldr)
    tr.set_pipeline(ta.PIPE_FUSED)
    tr.pathtrace(cam, 1, 1, 1024)
    assert tr.read_film().tobytes() == full.tobytes()
    assert np.isfinite(full).all() and full.min() >= 0.0
    mean = full.mean(0) / 1024.0
    assert (mean > 0.1).all() and (mean < 0.5).all()
    tr.close()


def test_config4_4k_sample_partition(ta, portable_oracle):
    """configs[3] at 3840x2160 (aspect 16:9): one whole iteration against the oracle, then the
    sample-index partition of tracer_amd.multi (rank r of G renders iterations r+1, r+1+G, ...) for
    G = 4 summed on the host equals the single-GPU film to float summation order."""
    from tracer_amd.multi import rank_iterations
    o = portable_oracle
    w, h = 3840, 2160
    tr = make_tracer(ta, w, h)
    cam = ta.camera_default(w, h)
    s = o.scene_create(0)
    tr.pathtrace(cam, 5, 1, 1)
    ref = o.pathtrace(s, w, h, o.camera_default(w, h), 5, 1, 1, nthreads=16)
    assert tr.read_film().tobytes() == ref["film"].tobytes()
    assert np.array_equal(tr.read_ldr(), ref["pbo"])
    total = 32
    tr.pathtrace(cam, 5, 1, total)
    full = tr.read_film()
    acc = np.zeros_like(full, dtype=np.float32)
    n = 0
    for r in range(4):
        first, stride, count = rank_iterations(total, r, 4)
        tr.reset_film()
        tr.pathtrace(cam, 5, first, count, iter_stride=stride)
        acc += tr.read_film(); n += count
    assert n == total
    assert np.abs(acc - full).max() <= 1e-5 * np.sqrt(total) * full.max()
    tr.close()


def test_config5_icosphere_1310720_triangles_1024x1024(ta, oracle_c):
    """configs[4]: brute-force closest hit over a 1 310 720-triangle MESH at 1024x1024. Pixel (64 i, 64 j)
    of that frame has exactly the ray of pixel (i, j) of a 16x16 frame (u = x / w is exact in both), which
    the oracle can afford: ids, t and normals there bit for bit. Then one path-traced iteration: the
    compacted pipeline with the brute-force pass (AUTO) against the wavefront one, same film."""
    level, w, h = 8, 1024, 1024
    tr = make_tracer(ta, w, h, 2, level)
    cam = ta.camera_default(w, h)
    tr.reset_counters()
    tr.raytrace1(cam)
    hits = tr.read_hits()
    ntri = 20 * 4 ** level
    tests = tr.counters()["tri_tests"]
    assert tests % ntri == 0 and tests // ntri > 0.2 * w * h      # rays that passed the AABB x all faces
    s = oracle_c.scene_create(2, level)
    ref = oracle_c.raytrace1(s, 16, 16, oracle_c.camera_default(16, 16), nthreads=16)
    sub = lambda a: a.reshape(h, w, -1)[::64, ::64].reshape(256, -1)
    assert np.array_equal(sub(hits["ids"]).ravel(), ref["ids"])
    assert sub(hits["ts"]).tobytes() == ref["ts"].reshape(256, -1).tobytes()
    assert sub(hits["nor"]).tobytes() == ref["nor"].reshape(256, -1).tobytes()
    assert sub(hits["pos"]).tobytes() == ref["pos"].reshape(256, -1).tobytes()
    assert (ref["ids"] == 6).sum() >= 20 and (hits["ids"] == 6).mean() > 0.1
    # the same frame through the BVH (TRC_OPT_MESH_ACCEL): every pixel identical to the exhaustive loop's
    tr.set_mesh_accel(True)
    tr.raytrace1(cam)
    accel = tr.read_hits()
    for k in ("ids", "ts", "pos", "nor"):
        assert accel[k].tobytes() == hits[k].tobytes(), k
    # and through the exhaustive loop without its bounding-sphere pre-test: again every pixel
    tr.set_mesh_accel(2)
    tr.raytrace1(cam)
    plain = tr.read_hits()
    for k in ("ids", "ts", "pos", "nor"):
        assert plain[k].tobytes() == hits[k].tobytes(), k
    tr.set_mesh_accel(False)
    tr.pathtrace(cam, 2, 1, 1)                       # AUTO -> compacted + brute-force pass
    a = tr.read_film()
    tr.set_pipeline(ta.PIPE_WAVEFRONT)
    tr.pathtrace(cam, 2, 1, 1)
    assert tr.read_film().tobytes() == a.tobytes()
    tr.set_pipeline(ta.PIPE_AUTO); tr.set_mesh_accel(True)
    tr.pathtrace(cam, 2, 1, 1)
    assert tr.read_film().tobytes() == a.tobytes()
    assert np.isfinite(a).all()
    tr.close()


@pytest.mark.parametrize("by_tile", [False, True])
@pytest.mark.parametrize("own_stream", [True, False])
def test_render_partitioned_orders_its_streams(ta, by_tile, own_stream):
    """tracer_amd.multi.render_partitioned (what bench.py and a multi-GPU caller run) with one rank: a Tracer on a
    PRIVATE stream (film.zero_() and the tracer's launches are then on different streams and the function must
    order them itself) and a Tracer on torch's current stream; sample split and image-tile split. The film is
    the plain trc_pathtrace film, every time, although it starts from garbage."""
    import torch
    from tracer_amd import multi
    w, h, iters = 512, 512, 24
    sc = ta.Scene(0)
    cam = ta.camera_default(w, h)
    with ta.Tracer(w, h) as ref:
        ref.upload_scene(sc.objects_ptr, sc.size)
        ref.pathtrace(cam, 4, 1, iters)
        want = ref.read_film(); want_ldr = ref.read_ldr()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        tr = ta.Tracer(w, h) if own_stream else ta.Tracer(w, h, stream=stream.cuda_stream)
        tr.upload_scene(sc.objects_ptr, sc.size)
        film = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
        tr.bind_film(film.data_ptr())
        for _ in range(3):
            film.fill_(7.0)                    # garbage that a render overtaking the zero_() would keep
            multi.render_partitioned(tr, cam, 4, iters, film, 0, 1, None, by_tile=by_tile)
            tr.synchronize(); stream.synchronize()
            assert film.cpu().numpy().tobytes() == want.tobytes()
            assert np.array_equal(tr.read_ldr(), want_ldr)
        tr.close()
