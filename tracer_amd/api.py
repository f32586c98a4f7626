"""ctypes binding of include/tracer_abi.h (libtracer_amd.so). Fails loudly if the library is absent."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RNG_REFERENCE, RNG_DECORRELATED = 0, 1
PIPE_WAVEFRONT, PIPE_COMPACTED, PIPE_FUSED, PIPE_QUEUE, PIPE_AUTO = 0, 1, 2, 3, 4
OPT_RNG_MODE, OPT_PIPELINE, OPT_RECORD_HITS, OPT_STAGE_MB, OPT_MESH_ACCEL, OPT_SKIP_TONEMAP = 0, 1, 2, 3, 4, 5

# every symbol include/tracer_abi.h declares
ABI_SYMBOLS = [
    "trc_create", "trc_destroy", "trc_set_option", "trc_upload_scene", "trc_update_object",
    "trc_raytrace1", "trc_pathtrace", "trc_reset_film", "trc_tonemap", "trc_read_film",
    "trc_write_film", "trc_read_ldr", "trc_read_hits", "trc_synchronize", "trc_film_device_ptr",
    "trc_bind_film", "trc_set_tile", "trc_get_counters", "trc_reset_counters", "trc_last_render_ms", "trc_last_kernel_ms",
    "trc_last_error_string", "trc_version", "pathtrace", "raytrace1",
    "trc_intersect_rays", "trc_intersect_rays_device", "trc_pick", "trc_pathtrace_static",
    "trc_legacy_invalidate", "trc_legacy_release",
]
# every symbol include/tracer_host.h declares
HOST_SYMBOLS = [
    "trc_scene_cornell", "trc_scene_load", "trc_scene_free", "trc_scene_size", "trc_scene_objects",
    "trc_scene_translate", "trc_scene_rotate", "trc_scene_scale", "trc_scene_add_obj",
    "trc_scene_set_material", "trc_camera_default", "trc_write_ppm", "trc_write_pfm",
]
DATA_DIR = os.path.join(_HERE, "data")


class TracerError(RuntimeError):
    pass


class Camera(C.Structure):
    """trc_camera: campos, A, B, C, lensRadius, focalDist (Renderer.cpp:156-171)."""
    _fields_ = [("campos", C.c_float * 3), ("A", C.c_float * 3), ("B", C.c_float * 3),
                ("C", C.c_float * 3), ("lensRadius", C.c_float), ("focalDist", C.c_float)]

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == 14
        return cls.from_buffer_copy(a.tobytes())

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class Counters(C.Structure):
    _fields_ = [("live_segments", C.c_uint64), ("paths", C.c_uint64), ("tri_tests", C.c_uint64),
                ("wave_cycles", C.c_uint64), ("wave_ref_ticks", C.c_uint64)]


def lib_path():
    # TRC_LIB: load another build of the same library (kernel tuning experiments)
    return os.environ.get("TRC_LIB") or os.path.join(_HERE, "libtracer_amd.so")


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same as the
    system ROCm's). A process must hold ONE HIP/HSA runtime: two copies fault inside
    libhsa-runtime64 as soon as both touch the GPU, and streams / device pointers handed from
    torch to this library only mean something inside torch's copy. If torch is installed but
    not imported yet, map its copy first so that our DT_NEEDED libamdhip64.so.7 binds to it and
    a later `import torch` finds it already loaded."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def load_library():
    """dlopen libtracer_amd.so and declare prototypes. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise TracerError(f"{p} not found: build it with `make -C tracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
    L.trc_create.argtypes = [i32, u32, u32, vp, C.POINTER(vp)]
    L.trc_destroy.argtypes = [vp]
    L.trc_set_option.argtypes = [vp, i32, i32]
    L.trc_upload_scene.argtypes = [vp, vp, u32]
    L.trc_update_object.argtypes = [vp, u32, vp, vp]
    L.trc_raytrace1.argtypes = [vp, C.POINTER(Camera)]
    L.trc_pathtrace.argtypes = [vp, C.POINTER(Camera), u64, u32, u32, u32]
    L.trc_pathtrace_static.argtypes = [vp, C.POINTER(Camera), u64, u32, u32, u32]
    L.trc_intersect_rays.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.trc_intersect_rays_device.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.trc_pick.argtypes = [vp, C.POINTER(Camera), u32, u32, i32, i32, C.POINTER(i32), C.POINTER(f32), vp, vp]
    L.trc_legacy_invalidate.argtypes = [vp]
    L.trc_legacy_release.argtypes = []
    L.trc_reset_film.argtypes = [vp]
    L.trc_tonemap.argtypes = [vp, f32]
    L.trc_read_film.argtypes = [vp, vp]
    L.trc_write_film.argtypes = [vp, vp]
    L.trc_read_ldr.argtypes = [vp, vp]
    L.trc_read_hits.argtypes = [vp, vp, vp, vp, vp]
    L.trc_synchronize.argtypes = [vp]
    L.trc_film_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.trc_bind_film.argtypes = [vp, vp]
    L.trc_set_tile.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.trc_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.trc_reset_counters.argtypes = [vp]
    L.trc_last_render_ms.argtypes = [vp, C.POINTER(f32)]
    L.trc_last_kernel_ms.argtypes = [vp, C.POINTER(f32)]
    L.trc_last_error_string.restype = C.c_char_p
    L.trc_version.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if name.startswith("trc_") and name not in ("trc_last_error_string", "trc_version"):
            fn.restype = i32
    # include/tracer_host.h
    for name in HOST_SYMBOLS:
        getattr(L, name).restype = i32
    L.trc_scene_cornell.argtypes = [i32, i32, C.c_char_p, C.POINTER(vp)]
    L.trc_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(Camera), C.c_char_p, C.c_uint32]
    L.trc_scene_free.argtypes = [vp]; L.trc_scene_free.restype = None
    L.trc_scene_size.argtypes = [vp]; L.trc_scene_size.restype = u32
    L.trc_scene_objects.argtypes = [vp]; L.trc_scene_objects.restype = vp
    L.trc_scene_translate.argtypes = [vp, u32, vp]
    L.trc_scene_rotate.argtypes = [vp, u32, f32, vp]
    L.trc_scene_scale.argtypes = [vp, u32, vp]
    L.trc_scene_add_obj.argtypes = [vp, C.c_char_p, C.POINTER(u32)]
    L.trc_scene_set_material.argtypes = [vp, u32, vp, f32, f32, f32, i32]
    L.trc_camera_default.argtypes = [u32, u32, f32, f32, C.POINTER(Camera)]
    L.trc_write_ppm.argtypes = [C.c_char_p, vp, u32, u32]
    L.trc_write_pfm.argtypes = [C.c_char_p, vp, u32, u32, f32]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Tracer:
    """One rendering context on one GPU (trc_ctx)."""

    def __init__(self, width, height, device=0, stream=None):
        self.L = load_library()
        self.w, self.h, self.npix = int(width), int(height), int(width) * int(height)
        self.ctx = C.c_void_p()
        self.stream_handle = int(stream) if stream else None     # the caller's hipStream_t, or None: a private stream
        self._check(self.L.trc_create(device, width, height, stream, C.byref(self.ctx)))

    def _check(self, rc):
        if rc != 0:
            raise TracerError(f"trc status {rc}: {self.L.trc_last_error_string().decode()}")

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.L.trc_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration -------------------------------------------------------------------
    def set_rng_mode(self, mode):
        self._check(self.L.trc_set_option(self.ctx, OPT_RNG_MODE, mode))

    def set_pipeline(self, pipe):
        self._check(self.L.trc_set_option(self.ctx, OPT_PIPELINE, pipe))

    def set_tile(self, row0, rows):
        """Render only image rows [row0, row0 + rows) from now on (rows == 0: the whole image)."""
        self._check(self.L.trc_set_tile(self.ctx, row0, rows))

    def set_mesh_accel(self, mode):
        """Big meshes: False / 0 = every face against every ray that passed the AABB (the reference's loop, default; a
        bounding-sphere pre-test skips the arithmetic of pairs that cannot hit), True / 1 = BVH in front of the same
        exact test, 2 = the loop without the pre-test (validation)."""
        self._check(self.L.trc_set_option(self.ctx, OPT_MESH_ACCEL, int(mode)))

    def set_stage_mb(self, mb):
        self._check(self.L.trc_set_option(self.ctx, OPT_STAGE_MB, int(mb)))

    def set_skip_tonemap(self, on):
        """pathtrace calls leave the LDR image alone (ranks of a multi-GPU render: the root tonemaps after the reduce)."""
        self._check(self.L.trc_set_option(self.ctx, OPT_SKIP_TONEMAP, int(bool(on))))

    def set_record_hits(self, on):
        self._check(self.L.trc_set_option(self.ctx, OPT_RECORD_HITS, int(bool(on))))

    # -- scene ---------------------------------------------------------------------------
    def upload_scene(self, objects_ptr, count):
        """objects_ptr: address of a host Object::Object[count] in the reference's layout."""
        self._check(self.L.trc_upload_scene(self.ctx, objects_ptr, count))

    def update_object(self, index, matrix, matrixi):
        m = np.ascontiguousarray(matrix, np.float32).reshape(16)
        mi = np.ascontiguousarray(matrixi, np.float32).reshape(16)
        self._check(self.L.trc_update_object(self.ctx, index, _ptr(m), _ptr(mi)))

    # -- rendering -------------------------------------------------------------------------
    def raytrace1(self, cam):
        self._check(self.L.trc_raytrace1(self.ctx, C.byref(cam)))

    def pathtrace(self, cam, seed, iter_first, n_iters, iter_stride=1):
        self._check(self.L.trc_pathtrace(self.ctx, C.byref(cam), seed, iter_first, iter_stride, n_iters))

    def pathtrace_static(self, cam, seed, iter_first, n_iters, iter_stride=1):
        """trc_pathtrace_static: the work-queue pipeline with its scratch in the library's static device array."""
        self._check(self.L.trc_pathtrace_static(self.ctx, C.byref(cam), seed, iter_first, iter_stride, n_iters))

    def intersect_rays(self, rays6):
        """Ray::intersectScene for an (n, 6) float32 array of {pos, dir}: dict(ids, ts, pos, nor)."""
        r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = r.shape[0]
        ids = np.empty(n, np.int32); ts = np.empty(n, np.float32)
        pos = np.empty((n, 3), np.float32); nor = np.empty((n, 3), np.float32)
        self._check(self.L.trc_intersect_rays(self.ctx, _ptr(r), n, _ptr(ids), _ptr(ts), _ptr(pos), _ptr(nor)))
        return dict(ids=ids, ts=ts, pos=pos, nor=nor)

    def intersect_rays_device(self, rays6_ptr, n, ids_ptr=None, ts_ptr=None, pos_ptr=None, nor_ptr=None):
        self._check(self.L.trc_intersect_rays_device(self.ctx, rays6_ptr, n, ids_ptr, ts_ptr, pos_ptr, nor_ptr))

    def pick(self, cam, window_w, window_h, x, y):
        """main.cpp:180-208: (id, t, pos, nor) under window position (x, y), y from the top."""
        i = C.c_int32(); t = C.c_float(); pos = np.empty(3, np.float32); nor = np.empty(3, np.float32)
        self._check(self.L.trc_pick(self.ctx, C.byref(cam), window_w, window_h, x, y, C.byref(i), C.byref(t), _ptr(pos), _ptr(nor)))
        return int(i.value), np.float32(t.value), pos, nor

    def reset_film(self):
        self._check(self.L.trc_reset_film(self.ctx))

    def tonemap(self, iters):
        self._check(self.L.trc_tonemap(self.ctx, float(iters)))

    def synchronize(self):
        self._check(self.L.trc_synchronize(self.ctx))

    # -- output ------------------------------------------------------------------------------
    def read_film(self):
        a = np.empty((self.npix, 3), np.float32)
        self._check(self.L.trc_read_film(self.ctx, _ptr(a)))
        return a

    def write_film(self, film):
        a = np.ascontiguousarray(film, np.float32)
        assert a.size == 3 * self.npix
        self._check(self.L.trc_write_film(self.ctx, _ptr(a)))

    def read_ldr(self):
        a = np.empty(self.npix, np.uint32)
        self._check(self.L.trc_read_ldr(self.ctx, _ptr(a)))
        return a

    def read_hits(self):
        ids = np.empty(self.npix, np.int32); ts = np.empty(self.npix, np.float32)
        pos = np.empty((self.npix, 3), np.float32); nor = np.empty((self.npix, 3), np.float32)
        self._check(self.L.trc_read_hits(self.ctx, _ptr(ids), _ptr(ts), _ptr(pos), _ptr(nor)))
        return dict(ids=ids, ts=ts, pos=pos, nor=nor)

    def film_device_ptr(self):
        p = C.c_void_p()
        self._check(self.L.trc_film_device_ptr(self.ctx, C.byref(p)))
        return p.value

    def bind_film(self, dev_ptr):
        self._check(self.L.trc_bind_film(self.ctx, dev_ptr))

    # -- measurement ---------------------------------------------------------------------------
    def counters(self):
        c = Counters()
        self._check(self.L.trc_get_counters(self.ctx, C.byref(c)))
        return dict(live_segments=int(c.live_segments), paths=int(c.paths), tri_tests=int(c.tri_tests),
                    wave_cycles=int(c.wave_cycles), wave_ref_ticks=int(c.wave_ref_ticks),
                    shader_mhz=(100.0 * c.wave_cycles / c.wave_ref_ticks) if c.wave_ref_ticks else None)

    def reset_counters(self):
        self._check(self.L.trc_reset_counters(self.ctx))

    def last_render_ms(self):
        ms = C.c_float()
        self._check(self.L.trc_last_render_ms(self.ctx, C.byref(ms)))
        return float(ms.value)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.L.trc_last_kernel_ms(self.ctx, C.byref(ms)))
        return float(ms.value)


class Scene:
    """Host-side scene (include/tracer_host.h): the reference's Cornell box built through the
    Object/Mesh API mirror. No GPU involved."""

    def __init__(self, variant=0, param=0, data_dir=None, path=None, aspect=(512, 512)):
        """Scene(variant, param): the built-in Cornell variants; Scene(path=..., aspect=(w, h)): a scene
        description file (include/tracer_host.h: trc_scene_load), whose camera is then in self.camera."""
        self.L = load_library()
        self.h = C.c_void_p()
        self.camera = None
        if path is not None:
            cam = Camera()
            err = C.create_string_buffer(512)
            rc = self.L.trc_scene_load(os.fsencode(path), aspect[0], aspect[1], C.byref(self.h), C.byref(cam), err, 512)
            if rc != 0:
                raise TracerError(f"trc_scene_load: {err.value.decode(errors='replace')}")
            self.camera = cam
            return
        rc = self.L.trc_scene_cornell(variant, param, (data_dir or DATA_DIR).encode(), C.byref(self.h))
        if rc != 0:
            raise TracerError(f"trc_scene_cornell failed ({rc})")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.trc_scene_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return int(self.L.trc_scene_size(self.h))

    @property
    def objects_ptr(self):
        return self.L.trc_scene_objects(self.h)

    def raw_objects(self):
        """The Object[] bytes (168 per object, including host pointers)."""
        return C.string_at(self.objects_ptr, 168 * self.size)

    def _chk(self, rc, what):
        if rc != 0:
            raise TracerError(f"{what} failed ({rc})")

    def translate(self, i, amount):
        a = np.asarray(amount, np.float32); self._chk(self.L.trc_scene_translate(self.h, i, _ptr(a)), "translate")

    def rotate(self, i, angle_deg, axis):
        a = np.asarray(axis, np.float32); self._chk(self.L.trc_scene_rotate(self.h, i, angle_deg, _ptr(a)), "rotate")

    def scale(self, i, amount):
        a = np.asarray(amount, np.float32); self._chk(self.L.trc_scene_scale(self.h, i, _ptr(a)), "scale")

    def add_obj(self, path):
        idx = C.c_uint32()
        self._chk(self.L.trc_scene_add_obj(self.h, path.encode(), C.byref(idx)), "add_obj")
        return int(idx.value)

    def set_material(self, i, color, emit=0.0, brdf=1.0, n=1.0, mat_type=0):
        c = np.asarray(color, np.float32)
        self._chk(self.L.trc_scene_set_material(self.h, i, _ptr(c), emit, brdf, n, mat_type), "set_material")


def camera_default(aspect_w=512, aspect_h=512, rot_x=0.0, rot_y=0.0):
    L = load_library()
    cam = Camera()
    rc = L.trc_camera_default(aspect_w, aspect_h, rot_x, rot_y, C.byref(cam))
    if rc != 0:
        raise TracerError("trc_camera_default failed")
    return cam


def write_ppm(path, ldr, w, h):
    L = load_library()
    a = np.ascontiguousarray(ldr, np.uint32)
    if L.trc_write_ppm(path.encode(), _ptr(a), w, h) != 0:
        raise TracerError("trc_write_ppm failed")


def write_pfm(path, film, w, h, iters):
    L = load_library()
    a = np.ascontiguousarray(film, np.float32)
    if L.trc_write_pfm(path.encode(), _ptr(a), w, h, float(iters)) != 0:
        raise TracerError("trc_write_pfm failed")
