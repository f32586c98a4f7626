"""trc_pathtrace_static -- the one entry point that allocates nothing (include/tracer_abi.h): the work-queue
pipeline with its task counter and sample staging in the library's static device array. Checked through the C ABI
against trc_pathtrace (TRC_PIPE_QUEUE) bit for bit, on two calls in a row that share the GPU's array, the second
with a smaller input; plus the size limit, the band work-around, and that device memory in use does not move.
Collected last (test_zz_*): the round's end runs `pytest -x`. Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tracer_amd
    tracer_amd.load_library()
    return tracer_amd


def _reference_render(ta, w, h, variant, cam, seed, first, n, stride=1, tile=None):
    """The existing entry point: trc_pathtrace through the work queue, on a context of its own."""
    sc = ta.Scene(variant)
    with ta.Tracer(w, h) as tr:
        tr.upload_scene(sc.objects_ptr, sc.size)
        tr.set_pipeline(ta.PIPE_QUEUE)
        if tile:
            tr.set_tile(*tile)
        tr.pathtrace(cam, seed, first, n, iter_stride=stride)
        return tr.read_film(), tr.read_ldr()


def test_static_scratch_entry_point_equals_trc_pathtrace(ta):
    import torch
    # call 1: 1024x1024 x 8 iterations = 100 MB of staging -> two chunks of the 64 MiB static area
    # call 2: a smaller input (64x48, glass scene, strided iterations) on another context of the same GPU
    cases = [(1024, 1024, 0, 5, 1, 8, 1), (64, 48, 1, 9, 2, 3, 2)]
    expected = []
    for w, h, variant, seed, first, n, stride in cases:
        expected.append(_reference_render(ta, w, h, variant, ta.camera_default(w, h), seed, first, n, stride))
    ctxs = []
    for w, h, variant, *_ in cases:                    # everything that allocates happens before the calls under test
        sc = ta.Scene(variant)
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        tr._scene = sc
        ctxs.append(tr)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for tr, (w, h, variant, seed, first, n, stride), (film, ldr) in zip(ctxs, cases, expected):
        tr.pathtrace_static(ta.camera_default(w, h), seed, first, n, iter_stride=stride)
        got = tr.read_film()
        assert got.tobytes() == film.tobytes(), (w, h)
        assert np.array_equal(tr.read_ldr(), ldr), (w, h)
        assert got.max() > 0
    assert torch.cuda.mem_get_info()[0] == free0       # nothing was allocated (or freed) on the device
    # again, larger after smaller: the array keeps the last call's contents and must not care
    tr, (w, h, variant, seed, first, n, stride), (film, _) = ctxs[0], cases[0], expected[0]
    tr.pathtrace_static(ta.camera_default(w, h), seed, first, n, iter_stride=stride)
    assert tr.read_film().tobytes() == film.tobytes()
    for tr in ctxs:
        tr.close()


def test_static_scratch_size_limit_and_bands(ta):
    """3840x2160 = 8 294 400 pixels is beyond one iteration of staging (5 592 405 pixels): a clear error, nothing
    rendered; two bands of 1080 rows fit and assemble the frame that trc_pathtrace renders."""
    w, h = 3840, 2160
    cam = ta.camera_default(w, h)
    film, _ = _reference_render(ta, w, h, 0, cam, 3, 1, 2)
    sc = ta.Scene(0)
    with ta.Tracer(w, h) as tr:
        tr.upload_scene(sc.objects_ptr, sc.size)
        with pytest.raises(ta.TracerError, match="static scratch holds"):
            tr.pathtrace_static(cam, 3, 1, 2)
        assert not tr.read_film().any()
        for row0 in (0, 1080):
            tr.set_tile(row0, 1080)
            tr.pathtrace_static(cam, 3, 1, 2)
        tr.set_tile(0, 0)
        assert tr.read_film().tobytes() == film.tobytes()
        # argument checks as trc_pathtrace's
        with pytest.raises(ta.TracerError):
            tr.pathtrace_static(cam, 3, 0, 1)


def test_static_scratch_two_contexts_back_to_back_without_a_host_sync(ta):
    """Two contexts with private (non-blocking) streams on the one GPU, called alternately from one host thread with
    nothing in between: both use the GPU's one static array -- task counter and staging -- so without the per-device
    guard event the second call's memset and samples would land in the first call's. Every film must be
    the one trc_pathtrace renders alone."""
    cases = [(768, 512, 0, 21, 1, 12, 1), (512, 512, 1, 22, 1, 16, 1)]      # ~4.7 M and ~4.2 M paths: milliseconds each
    expected = [_reference_render(ta, w, h, v, ta.camera_default(w, h), seed, first, n, stride)
                for w, h, v, seed, first, n, stride in cases]
    ctxs = []
    for w, h, variant, *_ in cases:
        sc = ta.Scene(variant)
        tr = ta.Tracer(w, h)                      # stream=None: a private hipStreamNonBlocking stream each
        tr.upload_scene(sc.objects_ptr, sc.size)
        tr._scene = sc
        ctxs.append(tr)
    for rounds in range(3):                       # a, b, a, b, a, b -- enqueued faster than they run, no host wait
        for tr, (w, h, variant, seed, first, n, stride) in zip(ctxs, cases):
            tr.pathtrace_static(ta.camera_default(w, h), seed, first, n, iter_stride=stride)
    for tr, (film, ldr) in zip(ctxs, expected):
        assert tr.read_film().tobytes() == film.tobytes()
        assert np.array_equal(tr.read_ldr(), ldr)
    for tr in ctxs:
        tr.close()


def test_static_scratch_entry_point_under_graph_capture(ta):
    """The advertised use: ONE call captured into a HIP graph (the stream in global capture mode, where hipMalloc and
    its like are errors), replayed twice, film and LDR against trc_pathtrace's. Capture through torch.cuda.graph on
    the stream the context launches on."""
    import torch
    w, h, variant, seed, n = 640, 384, 1, 31, 6
    cam = ta.camera_default(w, h)
    film, ldr = _reference_render(ta, w, h, variant, cam, seed, 1, n)
    sc = ta.Scene(variant)
    s = torch.cuda.Stream()
    with ta.Tracer(w, h, stream=s.cuda_stream) as tr:
        tr.upload_scene(sc.objects_ptr, sc.size)
        tr.pathtrace_static(cam, seed, 1, 1)                  # warm-up outside the capture (code objects loaded)
        tr.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="global"):
            tr.pathtrace_static(cam, seed, 1, n)
        with pytest.raises(ta.TracerError, match="nothing rendered"):
            tr.last_kernel_ms()                               # no timing events were recorded into the capture
        for _ in range(2):
            tr.write_film(np.full((w * h, 3), 7.0, np.float32))   # iteration 1 restarts the film: the replay must overwrite this
            g.replay()
            torch.cuda.synchronize()
            assert tr.read_film().tobytes() == film.tobytes()
            assert np.array_equal(tr.read_ldr(), ldr)
        # an ordinary call afterwards still works and is timed again
        tr.pathtrace_static(cam, seed, 1, n)
        assert tr.read_film().tobytes() == film.tobytes()
        assert tr.last_kernel_ms() > 0
        del g
