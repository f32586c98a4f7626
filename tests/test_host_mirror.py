"""The product's host side (tracer_amd/csrc/host/: Camera, Object/Mesh/Material API, scene
builder, .obj loader, image writers) and the C-ABI surface, checked WITHOUT a GPU:
  * the library loads and exports every symbol include/*.h declares;
  * the Cornell scene and camera frames are byte-identical to the oracle's (and, through the
    fixtures, to the reference host build's);
  * error behaviour of the boundary: bad arguments are reported, nothing exits, and with no GPU
    the compute entry points fail loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from conftest import REPO


def declared_symbols(header):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:trc_[a-z0-9_]+)|pathtrace|raytrace1)\s*\(", txt)))


def test_library_exports_every_declared_symbol(product_lib):
    from tracer_amd.api import ABI_SYMBOLS, HOST_SYMBOLS
    abi = declared_symbols("tracer_abi.h"); host = [s for s in declared_symbols("tracer_host.h") if s not in abi]
    assert sorted(abi) == sorted(ABI_SYMBOLS)
    assert sorted(host) == sorted(HOST_SYMBOLS)
    for name in abi + host:
        assert getattr(product_lib, name) is not None


def test_multi_gpu_library_exports_every_declared_symbol(product_lib):
    """include/tracer_multi.h -> tracer_amd/libtracer_amd_multi.so (context API + RCCL). Checked from its dynamic symbol
    table, without loading it: the library brings in the system's RCCL and HIP runtime, which must not be mixed
    into a process that also holds PyTorch's copies (One HIP runtime per process)."""
    import subprocess
    lib = os.path.join(REPO, "tracer_amd", "libtracer_amd_multi.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(REPO, "tracer_amd", "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    declared = [s for s in declared_symbols("tracer_multi.h") if s.startswith("trc_multi_")]
    assert len(declared) == 13, declared
    assert not [s for s in declared if s not in exported]
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    assert "librccl" in needed and "libtracer_amd.so" in needed
    # and libtracer_amd.so itself stays free of RCCL
    needed = subprocess.run(["readelf", "-d", os.path.join(REPO, "tracer_amd", "libtracer_amd.so")], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed


def test_no_oracle_on_the_product_path():
    """Nothing under tracer_amd/ (library sources or binding) or the C++ host layer may reach into oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "tracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                if "oracle" in open(os.path.join(root, f), errors="ignore").read().lower():
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def _flatten_objects(raw, n):
    out = []
    for i in range(n):
        ob = raw[168 * i:168 * (i + 1)]
        meshp = struct.unpack("Q", ob[128:136])[0]
        mesh = C.string_at(meshp, 72)
        nv, nn, nf = struct.unpack("III", mesh[0:12]); pv, pn, pf = struct.unpack("QQQ", mesh[16:40])
        out.append(dict(matrix=ob[0:64], matrixi=ob[64:128], material=ob[136:164], counts=(nv, nn, nf),
                        bmin=mesh[40:52], bmax=mesh[52:64], mesh_type=struct.unpack("i", mesh[64:68])[0],
                        verts=C.string_at(pv, 12 * nv) if nv else b"", norms=C.string_at(pn, 12 * nn) if nn else b"",
                        faces=C.string_at(pf, 24 * nf) if nf else b""))
    return out


@pytest.mark.parametrize("variant,param", [(0, 0), (1, 0), (2, 2), (2, 5)])
def test_cornell_scene_equals_oracle(product_lib, oracle_c, variant, param):
    import tracer_amd as ta
    sc = ta.Scene(variant, param); s = oracle_c.scene_create(variant, param)
    assert sc.size == oracle_c.scene_size(s)
    mine = _flatten_objects(sc.raw_objects(), sc.size)
    for i, p in enumerate(mine):
        d = oracle_c.scene_object(s, i)
        assert p["matrix"] == d["matrix"].tobytes() and p["matrixi"] == d["matrixi"].tobytes(), i
        assert p["bmin"] == d["bmin"].tobytes() and p["bmax"] == d["bmax"].tobytes()
        assert p["mesh_type"] == d["mesh_type"] and p["counts"] == tuple(int(x) for x in d["counts"])
        assert p["verts"] == d["verts"].tobytes() and p["norms"] == d["norms"].tobytes() and p["faces"] == d["faces"].tobytes()
        assert p["material"] == d["color"].tobytes() + np.array([d["emit"], d["brdf"], d["n"]], np.float32).tobytes() + struct.pack("i", d["mat_type"])


def test_scene_file_reproduces_the_hard_coded_scene(product_lib, tmp_path):
    """tracer_amd/data/cornell.scene lists main.cpp:274-396's calls; loading it gives the same
    Object[] (matrices, inverses, materials, meshes) and camera frame as the built-in scene, bit for bit."""
    import tracer_amd as ta
    built_in = ta.Scene(0)
    loaded = ta.Scene(path=os.path.join(REPO, "tracer_amd", "data", "cornell.scene"), aspect=(512, 512))
    assert loaded.size == built_in.size == 9
    a = _flatten_objects(built_in.raw_objects(), 9); b = _flatten_objects(loaded.raw_objects(), 9)
    for i in range(9):
        assert a[i] == b[i], i
    assert bytes(loaded.camera) == bytes(ta.camera_default(512, 512))
    # variants of the grammar, and errors with file:line
    p = tmp_path / "s.scene"
    p.write_text("camera zoom -13 focal 13 lens 0.5 rotate 0.2 0.1\nmesh base obj %s\nmesh ball icosphere base 2\n"
                 "object ball\n scale 3\n translate 0 -2 0\n color 1 1 0\n material trans\n ior 1.5\n"
                 % os.path.join(REPO, "tracer_amd", "data", "icosahedron.obj"))
    sc = ta.Scene(path=str(p), aspect=(64, 32))
    o = _flatten_objects(sc.raw_objects(), 1)[0]
    ball = ta.Scene(2, 2)
    ref = _flatten_objects(ball.raw_objects(), 7)[6]
    assert o["matrix"] == ref["matrix"] and o["verts"] == ref["verts"] and o["faces"] == ref["faces"] and o["bmax"] == ref["bmax"]
    assert abs(sc.camera.lensRadius - 0.5) < 1e-7
    for text, where in (("object nothing\n", ":1:"), ("mesh a plane\nobject a\n scale 1 2\n", ":3:"),
                        ("mesh a plane\n translate 0 0 0\n", ":2:"), ("mesh a obj missing.obj\n", ":1:"), ("", "no objects")):
        p.write_text(text)
        with pytest.raises(ta.TracerError, match=where):
            ta.Scene(path=str(p))


def test_scene_edits_equal_oracle(product_lib, oracle_c):
    import tracer_amd as ta
    sc = ta.Scene(0); s = oracle_c.scene_create(0)
    sc.translate(6, [0.3, 0.1, -0.2]); oracle_c.scene_translate(s, 6, [0.3, 0.1, -0.2])
    sc.rotate(8, 33.0, [0, 1, 0]); oracle_c.scene_rotate(s, 8, 33.0, [0, 1, 0])
    sc.scale(7, [1.1, 0.9, 1.0]); oracle_c.scene_scale(s, 7, [1.1, 0.9, 1.0])
    mine = _flatten_objects(sc.raw_objects(), sc.size)
    for i in (6, 7, 8):
        d = oracle_c.scene_object(s, i, with_mesh=False)
        assert mine[i]["matrix"] == d["matrix"].tobytes() and mine[i]["matrixi"] == d["matrixi"].tobytes()


def test_camera_frames_equal_reference_fixture(product_lib, golden):
    import tracer_amd as ta
    g = golden["scenes"]
    for args, expect in zip(g["camera_args"], g["cameras"]):
        cam = ta.camera_default(int(args[0]), int(args[1]), float(args[2]), float(args[3]))
        assert cam.as_array().tobytes() == expect.tobytes(), args


def test_product_icosahedron_is_the_closed_form_and_equals_the_fixture(product_lib, tmp_path):
    """tracer_amd/data/icosahedron.obj is an output of tools/make_icosahedron.py (golden-ratio closed form, six
    decimals), and it loads to the same vertices, normals, faces and AABB as the fixture tests/golden/icosahedron.obj
    (the scene input the reference-build fixtures were generated with)."""
    import subprocess
    import sys
    import tracer_amd as ta
    out = tmp_path / "ico.obj"
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_icosahedron.py"), str(out)], check=True,
                   stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(REPO, "tracer_amd", "data", "icosahedron.obj")).read()
    sc = ta.Scene(0)
    a = sc.add_obj(str(out))
    b = sc.add_obj(os.path.join(REPO, "tests", "golden", "icosahedron.obj"))
    objs = _flatten_objects(sc.raw_objects(), sc.size)
    for k in ("counts", "verts", "norms", "faces", "bmin", "bmax"):
        assert objs[a][k] == objs[b][k] == objs[8][k], k


def test_obj_loader_dialects_and_errors(product_lib, tmp_path):
    import tracer_amd as ta
    sc = ta.Scene(0)
    # the reference's own dialect (v / vn / f a//b) round-trips identically to the Cornell icosahedron
    i = sc.add_obj(os.path.join(REPO, "tracer_amd", "data", "icosahedron.obj"))
    objs = _flatten_objects(sc.raw_objects(), sc.size)
    assert objs[i]["verts"] == objs[8]["verts"] and objs[i]["faces"] == objs[8]["faces"] and objs[i]["bmin"] == objs[8]["bmin"]
    # "f a/b/c", quads (fanned) and files without normals (geometric normals synthesised)
    p = tmp_path / "quad.obj"
    p.write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nvt 0 0\nf 1/1 2/1 3/1 4/1\n")
    j = sc.add_obj(str(p))
    q = _flatten_objects(sc.raw_objects(), sc.size)[j]
    assert q["counts"] == (4, 2, 2)
    n = np.frombuffer(q["norms"], np.float32).reshape(-1, 3)
    assert np.allclose(np.abs(n), [[0, 1, 0], [0, 1, 0]])
    # errors are reported, not exit()ed
    bad = tmp_path / "bad.obj"; bad.write_text("v 0 0 0\nf 1//1 2//1 9//1\n")
    with pytest.raises(ta.TracerError):
        sc.add_obj(str(bad))
    with pytest.raises(ta.TracerError):
        sc.add_obj(str(tmp_path / "missing.obj"))
    sc.set_material(j, [0.2, 0.3, 0.4], emit=1.0, mat_type=2)
    with pytest.raises(ta.TracerError):
        sc.set_material(j, [0, 0, 0], mat_type=7)


def test_image_writers(product_lib, tmp_path):
    import tracer_amd as ta
    w, h = 5, 3
    ldr = (np.arange(w * h, dtype=np.uint32) * 0x010203) & 0xFFFFFF
    ta.write_ppm(str(tmp_path / "a.ppm"), ldr, w, h)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n5 3\n255\n")
    px = np.frombuffer(raw[len(b"P6\n5 3\n255\n"):], np.uint8).reshape(h, w, 3)
    # buffer row 0 is the bottom of the image
    assert px[h - 1, 1].tolist() == [(ldr[1] >> 16) & 255, (ldr[1] >> 8) & 255, ldr[1] & 255]
    film = np.arange(w * h * 3, dtype=np.float32)
    ta.write_pfm(str(tmp_path / "a.pfm"), film, w, h, 2.0)
    raw = (tmp_path / "a.pfm").read_bytes()
    hdr = b"PF\n5 3\n-1.0\n"
    assert raw.startswith(hdr) and np.allclose(np.frombuffer(raw[len(hdr):], np.float32), film / 2.0)


def test_boundary_rejects_bad_arguments(product_lib):
    L = product_lib
    ctx = C.c_void_p()
    assert L.trc_create(0, 0, 16, None, C.byref(ctx)) != 0          # zero width
    assert b"size" in L.trc_last_error_string()
    assert L.trc_create(0, 16, 16, None, None) != 0                  # null out
    assert L.trc_upload_scene(None, None, 0) != 0
    assert L.trc_read_film(None, None) != 0
    assert L.trc_destroy(None) == 0


def test_no_cpu_fallback_without_gpu(product_lib):
    """On a box without a GPU the compute boundary must fail with a HIP error, never render."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tracer_amd as ta
    with pytest.raises(ta.TracerError):
        ta.Tracer(16, 16)


def test_two_hip_runtimes_are_refused_not_crashed():
    """INTEGRATION.md, One HIP runtime per process: a process that loaded libtracer_amd.so on the system's libamdhip64 and
    then mapped PyTorch's bundled copy gets an error from trc_create (before any HIP call -- so this runs without a GPU)
    instead of a crash inside libhsa-runtime64 later; loaded through tracer_amd.load_library() there is one runtime and
    the check is silent."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.find_spec("torch")
    torch_hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec else None
    if not torch_hip or not os.path.exists(torch_hip):
        pytest.skip("no PyTorch with a bundled libamdhip64.so here")
    wrong = r'''
import ctypes, sys
L = ctypes.CDLL(sys.argv[1])
L.trc_last_error_string.restype = ctypes.c_char_p
ctypes.CDLL(sys.argv[2], mode=ctypes.RTLD_GLOBAL)
ctx = ctypes.c_void_p()
rc = L.trc_create(0, 64, 64, None, ctypes.byref(ctx))
print(rc, L.trc_last_error_string().decode())
'''
    p = subprocess.run([sys.executable, "-c", wrong, os.path.join(REPO, "tracer_amd", "libtracer_amd.so"), torch_hip],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith("1 ") and "two HIP runtimes" in p.stdout and "libamdhip64" in p.stdout, p.stdout
    right = r'''
import ctypes, sys
sys.path.insert(0, sys.argv[1])
import tracer_amd
L = tracer_amd.load_library()
ctx = ctypes.c_void_p()
rc = L.trc_create(0, 64, 64, None, ctypes.byref(ctx))
print(rc, L.trc_last_error_string().decode())
'''
    p = subprocess.run([sys.executable, "-c", right, REPO], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "two HIP runtimes" not in p.stdout, p.stdout          # (on a box without a GPU the call fails later, in hipGetDeviceCount)
