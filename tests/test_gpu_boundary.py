"""The two other faces of the boundary, on a real GPU:
  * the reference's own launcher signatures `pathtrace` / `raytrace1` (raytrace.cu:189-238)
    fed with device buffers laid out exactly as the reference's Renderer.cpp builds them;
  * the C++ host mirror (Renderer / Camera / Object / Mesh classes) driven by a headless
    analogue of the reference's main.cpp (tests/cpp/headless_main.cpp)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tracer_amd
    tracer_amd.load_library()
    return tracer_amd


def _device_scene(torch, raw, n):
    """Deep-copy Object[] to the device as Renderer::initScene does (Renderer.cpp:112-151)."""
    keep = []
    objs = bytearray(raw)
    for i in range(n):
        meshp = struct.unpack("Q", raw[168 * i + 128:168 * i + 136])[0]
        mesh = bytearray(C.string_at(meshp, 72))
        nv, nn, nf = struct.unpack("III", mesh[0:12]); pv, pn, pf = struct.unpack("QQQ", mesh[16:40])
        ptrs = []
        for p, nbytes in ((pv, 12 * nv), (pn, 12 * nn), (pf, 24 * nf)):
            if nbytes:
                t = torch.frombuffer(bytearray(C.string_at(p, nbytes)), dtype=torch.uint8).cuda()
                keep.append(t); ptrs.append(t.data_ptr())
            else:
                ptrs.append(0)
        mesh[16:40] = struct.pack("QQQ", *ptrs)
        mt = torch.frombuffer(mesh, dtype=torch.uint8).cuda(); keep.append(mt)
        objs[168 * i + 128:168 * i + 136] = struct.pack("Q", mt.data_ptr())
    sd = torch.frombuffer(objs, dtype=torch.uint8).cuda(); keep.append(sd)
    return sd, keep


def test_reference_launcher_signatures(ta, oracle_c):
    import torch
    from oracle.pyoracle import Oracle
    L = ta.load_library()
    w, h = 96, 64
    n = w * h
    sc = ta.Scene(0)
    scene_d, keep = _device_scene(torch, sc.raw_objects(), sc.size)
    cam = ta.camera_default(w, h)
    f3 = C.c_float * 3
    args_cam = [f3(*cam.campos), f3(*cam.A), f3(*cam.B), f3(*cam.C)]
    pbo = torch.zeros(n, dtype=torch.int32, device="cuda")
    vp = C.c_void_p
    L.raytrace1.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, f3, f3, f3, f3, vp, C.c_uint32]
    L.raytrace1.restype = None
    L.pathtrace.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, f3, f3, f3, f3, C.c_float, C.c_float,
                            vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32]
    L.pathtrace.restype = None
    L.raytrace1(pbo.data_ptr(), w, h, 0.01, *args_cam, scene_d.data_ptr(), sc.size)
    torch.cuda.synchronize()
    s = oracle_c.scene_create(0)
    ocam = oracle_c.camera_default(w, h)
    ref = oracle_c.raytrace1(s, w, h, ocam)
    assert np.array_equal(pbo.cpu().numpy().view(np.uint32), ref["pbo"])

    # mode 2 exactly as Renderer::render drives it: ++filmIters, time += DELTA_T per frame
    rays = torch.zeros(n * 6, dtype=torch.float32, device="cuda"); col = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
    rnd = torch.zeros(n * 3, dtype=torch.float32, device="cuda"); idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    film = torch.full((n * 3,), 5.0, dtype=torch.float32, device="cuda")      # garbage: iteration 1 must zero it
    oracle_c.set_math_mode(Oracle.MATH_PORTABLE)
    try:
        ref_film = np.zeros((n, 3), np.float32)
        timer = np.float32(0.0)
        for it in (1, 2, 3):
            timer = np.float32(timer + np.float32(0.01))
            L.pathtrace(pbo.data_ptr(), w, h, float(timer), *args_cam, cam.lensRadius, cam.focalDist,
                        scene_d.data_ptr(), sc.size, rnd.data_ptr(), rays.data_ptr(), col.data_ptr(), idx.data_ptr(),
                        film.data_ptr(), it)
            seed = int(np.uint64(np.float32(timer * np.float32(100.0))))
            r = oracle_c.pathtrace(s, w, h, ocam, seed, it, 1, film=ref_film)
            ref_film = r["film"]
        torch.cuda.synchronize()
        assert film.cpu().numpy().tobytes() == ref_film.tobytes()
        assert np.array_equal(pbo.cpu().numpy().view(np.uint32), r["pbo"])
        # Renderer::updateScene edits scene_d in place (Renderer.cpp:221-225): must be honoured
        oracle_c.scene_translate(s, 6, [0.0, 1.0, 0.0])
        d = oracle_c.scene_object(s, 6, with_mesh=False)
        upd = torch.from_numpy(np.concatenate([d["matrix"], d["matrixi"]])).cuda()
        scene_d[168 * 6:168 * 6 + 128] = upd.view(torch.uint8)
        L.raytrace1(pbo.data_ptr(), w, h, 0.0, *args_cam, scene_d.data_ptr(), sc.size)
        torch.cuda.synchronize()
        assert np.array_equal(pbo.cpu().numpy().view(np.uint32), oracle_c.raytrace1(s, w, h, ocam)["pbo"])
    finally:
        oracle_c.set_math_mode(Oracle.MATH_LIBM)


def test_cpp_renderer_mirror(ta, tmp_path):
    """headless_main drives Renderer::render(camera, timer) frame by frame like main.cpp:125-131;
    the film must equal the same iterations issued through the C ABI."""
    exe = os.path.join(REPO, "tests", "cpp", "bin", "headless_main")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(REPO, "tests", "cpp")], check=True)
    data = os.path.join(REPO, "tracer_amd", "data")
    w, h, frames = 80, 60, 5
    out = str(tmp_path / "frame")
    r = subprocess.run([exe, data, str(w), str(h), str(frames), out, "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"iterations {frames}" in r.stdout
    # Renderer::pick (main.cpp:180-208) printed by the program vs trc_pick through ctypes
    import re
    m = re.search(r"pick centre (-?\d+), pick \(w/4, h/8\) (-?\d+)", r.stdout)
    assert m, r.stdout
    film = np.fromfile(out + ".film", np.float32).reshape(-1, 3)
    ldr = np.fromfile(out + ".ldr", np.uint32)
    sc = ta.Scene(0)
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    cam = ta.camera_default(w, h)
    timer = np.float32(0.0)
    for it in range(1, frames + 1):
        timer = np.float32(timer + np.float32(0.01))
        tr.pathtrace(cam, int(np.uint64(np.float32(timer * np.float32(100.0)))), it, 1)
    assert tr.read_film().tobytes() == film.tobytes()
    assert np.array_equal(tr.read_ldr(), ldr)
    assert int(m.group(1)) == tr.pick(cam, w, h, w // 2, h // 2)[0] and int(m.group(2)) == tr.pick(cam, w, h, w // 4, h // 8)[0]
    assert os.path.getsize(out + ".ppm") == len(f"P6\n{w} {h}\n255\n") + 3 * w * h
    # mode 1 through the class, and the batched extension
    r = subprocess.run([exe, data, str(w), str(h), "1", out + "1", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    tr.raytrace1(cam)
    assert np.array_equal(tr.read_ldr(), np.fromfile(out + "1.ldr", np.uint32))
    r = subprocess.run([exe, data, str(w), str(h), "4", out + "b", "2", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    tr.reset_film()
    tr.pathtrace(cam, 1, 1, 4)            # time = 0.01 -> seed (uint64)(0.01f*100) = 1
    assert tr.read_film().tobytes() == np.fromfile(out + "b.film", np.float32).tobytes()
    tr.close()


def test_legacy_launchers_notice_a_mesh_swapped_in_place(ta, oracle_c, tmp_path):
    """The shim behind the reference's signatures keeps a flattened copy of `scene_d`. A mesh whose arrays are
    rewritten in place -- what a freed scene and a new Mesh landing at the same addresses look like -- must be
    rendered as it is NOW: same pointers, same counts, other vertices. Also: an unchanged scene is not read back
    again (the digest path), trc_legacy_invalidate / trc_legacy_release work, and a frame through the legacy
    signature costs about what the fused frame costs."""
    import time
    import torch
    L = ta.load_library()
    w, h = 128, 96
    n = w * h
    # scene B = the Cornell box with a squashed icosahedron: same counts, other vertices and bounds
    src = open(os.path.join(REPO, "tracer_amd", "data", "icosahedron.obj")).read().splitlines()
    out = []
    for ln in src:
        if ln.startswith("v "):
            x, y, z = [float(t) for t in ln.split()[1:]]
            ln = "v %.6f %.6f %.6f" % (x * 1.25, y * 0.5, z * 0.75)
        out.append(ln)
    squashed = tmp_path / "squashed.obj"; squashed.write_text("\n".join(out) + "\n")
    sc_a = ta.Scene(0)
    sc_b = ta.Scene(0)
    k = sc_b.add_obj(str(squashed))
    raw_b = bytearray(sc_b.raw_objects())
    meshp_b = struct.unpack("Q", raw_b[168 * k + 128:168 * k + 136])[0]
    mesh_b = C.string_at(meshp_b, 72)
    scene_d, keep = _device_scene(torch, sc_a.raw_objects(), sc_a.size)
    cam = ta.camera_default(w, h)
    f3 = C.c_float * 3
    args_cam = [f3(*cam.campos), f3(*cam.A), f3(*cam.B), f3(*cam.C)]
    vp = C.c_void_p
    L.raytrace1.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, f3, f3, f3, f3, vp, C.c_uint32]
    L.raytrace1.restype = None
    pbo = torch.zeros(n, dtype=torch.int32, device="cuda")

    def frame():
        L.raytrace1(pbo.data_ptr(), w, h, 0.0, *args_cam, scene_d.data_ptr(), sc_a.size)
        torch.cuda.synchronize()
        return pbo.cpu().numpy().view(np.uint32)

    ocam = oracle_c.camera_default(w, h)
    ref_a = oracle_c.raytrace1(oracle_c.scene_create(0), w, h, ocam)["pbo"]
    assert np.array_equal(frame(), ref_a)
    assert np.array_equal(frame(), ref_a)            # second frame: the digest matches, nothing is read back
    # overwrite the icosahedron's device arrays and its header's bounds IN PLACE with scene B's mesh
    obj8 = sc_a.raw_objects()[168 * 8:168 * 9]
    mesh_d_ptr = struct.unpack("Q", bytes(scene_d[168 * 8 + 128:168 * 8 + 136].cpu().numpy()))[0]
    mesh_t = next(t for t in keep if t.data_ptr() == mesh_d_ptr)
    hdr = bytearray(mesh_t.cpu().numpy().tobytes())
    nv, nn, nf = struct.unpack("III", hdr[0:12]); pv, pn, pf = struct.unpack("QQQ", hdr[16:40])
    assert (nv, nn, nf) == struct.unpack("III", mesh_b[0:12])
    bv = struct.unpack("Q", mesh_b[16:24])[0]
    verts_t = next(t for t in keep if t.data_ptr() == pv)
    verts_t.copy_(torch.frombuffer(bytearray(C.string_at(bv, 12 * nv)), dtype=torch.uint8).cuda())
    hdr[40:64] = mesh_b[40:64]                        # m_bmin, m_bmax
    mesh_t.copy_(torch.frombuffer(hdr, dtype=torch.uint8).cuda())
    torch.cuda.synchronize()
    # the oracle's view of the edited scene: object 8 now uses scene B's mesh (host copy), everything else as before
    raw_mix = bytearray(sc_a.raw_objects())
    raw_mix[168 * 8 + 128:168 * 8 + 136] = struct.pack("Q", meshp_b)
    buf = C.create_string_buffer(bytes(raw_mix), len(raw_mix))
    s_mix = oracle_c.scene_wrap(C.addressof(buf), sc_a.size)
    ref_b = oracle_c.raytrace1(s_mix, w, h, ocam)["pbo"]
    assert not np.array_equal(ref_a, ref_b)
    assert np.array_equal(frame(), ref_b)
    assert obj8 == sc_a.raw_objects()[168 * 8:168 * 9]
    # housekeeping entry points
    assert L.trc_legacy_invalidate(scene_d.data_ptr()) == 0
    assert np.array_equal(frame(), ref_b)
    # cost of a frame through the legacy signature (digest + launch), reported, loosely bounded
    for _ in range(5):
        frame()
    t0 = time.perf_counter()
    for _ in range(50):
        L.raytrace1(pbo.data_ptr(), w, h, 0.0, *args_cam, scene_d.data_ptr(), sc_a.size)
    torch.cuda.synchronize()
    per_frame_ms = (time.perf_counter() - t0) / 50 * 1e3
    print(f"legacy raytrace1 frame {w}x{h}: {per_frame_ms:.3f} ms")
    assert per_frame_ms < 5.0
    assert L.trc_legacy_release() == 0
    assert np.array_equal(frame(), ref_b)             # and it comes back after a release
    assert L.trc_legacy_release() == 0


def test_multi_gpu_c_abi(ta, tmp_path):
    """include/tracer_multi.h driven by a plain C++ program (tests/cpp/multi_main.cpp; no Python, no torch in that
    process): one context per listed device, sample-index or row-band partition, one film sum onto GPU 0.
    On this one-GPU box: G = 1 through a real RCCL communicator and ncclReduce (TRC_MULTI_RCCL_SINGLE=1), and G = 2, 3
    as several contexts on the one GPU (the library's same-device mode, film[0] += film[r] in order) -- the partition,
    the per-GPU zeroing and the combine are the code an 8-GPU node runs."""
    exe = os.path.join(REPO, "tests", "cpp", "bin", "multi_main")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(REPO, "tests", "cpp")], check=True)
    data = os.path.join(REPO, "tracer_amd", "data")
    w, h, iters = 160, 96, 7
    sc = ta.Scene(0)
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    cam = ta.camera_default(w, h)
    tr.pathtrace(cam, 7, 1, iters)
    whole, whole_ldr = tr.read_film(), tr.read_ldr()

    def run(devices, by_tile, env=None):
        out = str(tmp_path / f"m{len(devices)}{by_tile}")
        r = subprocess.run([exe, data, str(w), str(h), str(iters), str(by_tile), out] + [str(d) for d in devices],
                           capture_output=True, text=True, timeout=180, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stderr + r.stdout
        assert f"gpus {len(devices)} iterations {iters}" in r.stdout
        return np.fromfile(out + ".film", np.float32).reshape(-1, 3), np.fromfile(out + ".ldr", np.uint32), r.stdout

    # G = 1: the same film; once plain, once with the RCCL communicator and the (one-rank) ncclReduce in the path
    for env in (None, {"TRC_MULTI_RCCL_SINGLE": "1"}):
        film, ldr, _ = run([0], 0, env)
        assert film.tobytes() == whole.tobytes() and np.array_equal(ldr, whole_ldr)
    # G = 2, 3 by sample index: exactly the sum of the strided partial films, in rank order; within 1e-5 sqrt(n) of the whole
    for G in (2, 3):
        film, ldr, text = run([0] * G, 0)
        parts = []
        for r in range(G):
            tr.reset_film(); tr.pathtrace(cam, 7, r + 1, len(range(r + 1, iters + 1, G)), iter_stride=G); parts.append(tr.read_film())
        expect = parts[0]
        for p in parts[1:]:
            expect = (expect + p).astype(np.float32)
        assert film.tobytes() == expect.tobytes(), G
        assert np.abs(film - whole).max() <= 1e-5 * np.sqrt(iters) * whole.max()
        tr.write_film(expect); tr.tonemap(iters)
        assert np.array_equal(ldr, tr.read_ldr())
    # G = 3 by bands of rows: disjoint tiles over zeroed films -> the whole film bit for bit
    film, ldr, _ = run([0, 0, 0], 1)
    assert film.tobytes() == whole.tobytes() and np.array_equal(ldr, whole_ldr)
    tr.close()


def test_multi_gpu_c_abi_at_4k_with_four_contexts(ta, tmp_path):
    """BASELINE.json configs[3]'s shape through the C/C++ multi-GPU layer: 3840x2160, 32 samples, G = 4 contexts (on this
    box all on the one GPU: the same partition, zeroing, combine and display transform an 8-GPU node runs, with the film sum
    as a same-device kernel instead of ncclReduce). By sample index: exactly the rank-ordered sum of the four strided partial
    films, and within 1e-5 sqrt(iterations) of the one-context film; by bands of rows: the one-context film
    bit for bit. The statistics line must account for the time: per-rank render, all-rendered and combine."""
    import re
    exe = os.path.join(REPO, "tests", "cpp", "bin", "multi_main")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(REPO, "tests", "cpp")], check=True)
    data = os.path.join(REPO, "tracer_amd", "data")
    w, h, iters, G = 3840, 2160, 32, 4
    sc = ta.Scene(0)
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    cam = ta.camera_default(w, h)
    tr.pathtrace(cam, 7, 1, iters)
    whole, whole_ldr = tr.read_film(), tr.read_ldr()
    expect = None
    for r in range(G):
        tr.reset_film(); tr.pathtrace(cam, 7, r + 1, iters // G, iter_stride=G)
        part = tr.read_film()
        expect = part if expect is None else (expect + part).astype(np.float32)
    for by_tile in (0, 1):
        out = str(tmp_path / f"k{by_tile}")
        r = subprocess.run([exe, data, str(w), str(h), str(iters), str(by_tile), out] + ["0"] * G,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        film = np.fromfile(out + ".film", np.float32).reshape(-1, 3)
        ldr = np.fromfile(out + ".ldr", np.uint32)
        if by_tile:
            assert film.tobytes() == whole.tobytes() and np.array_equal(ldr, whole_ldr)
        else:
            assert film.tobytes() == expect.tobytes()
            assert np.abs(film - whole).max() <= 1e-5 * np.sqrt(iters) * whole.max()
            tr.write_film(expect); tr.tonemap(iters)
            assert np.array_equal(ldr, tr.read_ldr())
        m = re.search(r"ms ([\d.]+) live_segments (\d+)", r.stdout)
        b = re.search(r"breakdown all_rendered_ms ([\d.]+) combine_ms ([\d.]+) render_ms((?: [\d.]+)+)", r.stdout)
        assert m and b, r.stdout
        total, all_rendered, combine = float(m.group(1)), float(b.group(1)), float(b.group(2))
        render = [float(x) for x in b.group(3).split()]
        assert len(render) == G and all(x > 0 for x in render)
        assert abs(all_rendered + combine - total) < 0.01 * total + 0.01
        assert max(render) <= all_rendered * 1.05 + 0.5          # a rank's GPU time fits inside the host's render span
        assert int(m.group(2)) > 3.5 * w * h * iters             # ~3.7 live segments per path, all ranks together
    tr.close()
