"""oracle/pyoracle.py -- TEST INFRASTRUCTURE: ctypes binding of oracle/oracle_api.h.

Loads either CPU checker:
  Oracle("c")   -> oracle/libtracer_oracle.so      (plain-C restatement)
  Oracle("ref") -> oracle/_ref/libtracer_ref.so    (reference .inl + GLM host build)
  Oracle("ref_pm") -> oracle/_ref/libtracer_ref_pm.so  (the same with portable sin/cos/acos/pow interposed)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
DATA_DIR = os.path.join(REPO, "tracer_amd", "data")
REF_DIR = "/root/reference"

LIBS = {
    "c": os.path.join(HERE, "libtracer_oracle.so"),
    "ref": os.path.join(HERE, "_ref", "libtracer_ref.so"),
    # the same reference sources with sinf/cosf/acosf/powf bound to oracle/pmath.h (oracle/pm_interpose.c)
    "ref_pm": os.path.join(HERE, "_ref", "libtracer_ref_pm.so"),
}


class Camera(C.Structure):
    _fields_ = [("campos", C.c_float * 3), ("A", C.c_float * 3), ("B", C.c_float * 3),
                ("C", C.c_float * 3), ("lensRadius", C.c_float), ("focalDist", C.c_float)]

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == 14
        return cls.from_buffer_copy(a.tobytes())


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("live_segments", "miss", "light", "depthcap", "diff", "trans", "mirr", "paths")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def available(kind):
    return os.path.exists(LIBS[kind])


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    RNG_REF, RNG_DECORRELATED = 0, 1
    MATH_LIBM, MATH_PORTABLE = 0, 1

    def __init__(self, kind="c"):
        path = LIBS[kind]
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        self.kind = kind
        L = self.lib = C.CDLL(path)
        vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
        L.orc_name.restype = C.c_char_p
        L.orc_set_math_mode.argtypes = [i32]; L.orc_set_math_mode.restype = i32
        L.orc_scene_create.argtypes = [i32, i32, C.c_char_p]; L.orc_scene_create.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_wrap.argtypes = [vp, C.c_uint]; L.orc_scene_wrap.restype = vp
        L.orc_scene_size.argtypes = [vp]; L.orc_scene_size.restype = u32
        L.orc_scene_objects.argtypes = [vp]; L.orc_scene_objects.restype = vp
        L.orc_scene_object_info.argtypes = [vp, u32] + [vp] * 9
        L.orc_scene_object_mesh.argtypes = [vp, u32, vp, vp, vp]
        L.orc_scene_translate.argtypes = [vp, u32, vp]
        L.orc_scene_rotate.argtypes = [vp, u32, f32, vp]
        L.orc_scene_scale.argtypes = [vp, u32, vp]
        L.orc_camera_default.argtypes = [u32, u32, f32, f32, C.POINTER(Camera)]
        L.orc_raytrace1.argtypes = [vp, u32, u32, C.POINTER(Camera), vp, vp, vp, vp, vp, i32]
        L.orc_pathtrace.argtypes = [vp, u32, u32, C.POINTER(Camera), u64, u32, u32, u32, i32,
                                    vp, vp, C.POINTER(Stats), i32]
        L.orc_tonemap.argtypes = [vp, u32, f32, vp]
        L.orc_intersect_scene.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_intersect_object.argtypes = [vp, u32, vp, vp, vp, vp]
        L.orc_pick.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.orc_bounce.argtypes = [i32, f32, vp, vp, vp, vp]
        L.orc_reflectance.argtypes = [vp, vp, f32, f32]; L.orc_reflectance.restype = f32
        L.orc_rand_vector.argtypes = [f32, f32, vp]
        L.orc_rand_vector_hem.argtypes = [f32, f32, vp, vp]
        L.orc_rand_point_disk.argtypes = [f32, f32, f32, vp]
        L.orc_rand3_api.argtypes = [u64, u32, u32, u32, vp]
        L.orc_philox.argtypes = [vp, vp, vp]
        L.orc_rgb_to_int.argtypes = [f32, f32, f32]; L.orc_rgb_to_int.restype = u32
        if kind == "c":
            L.orc_mesh_exit_histogram.argtypes = [vp, u32, u32, C.POINTER(Camera), u32, u32, vp, i32]
            L.orc_pm_sincos.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
            L.orc_pm_acos.argtypes = [f32]; L.orc_pm_acos.restype = f32
            L.orc_pm_pow.argtypes = [f32, f32]; L.orc_pm_pow.restype = f32

    # -- configuration -------------------------------------------------------------
    def name(self):
        return self.lib.orc_name().decode()

    def set_math_mode(self, mode):
        rc = self.lib.orc_set_math_mode(mode)
        if rc != 0:
            raise RuntimeError(f"{self.name()}: math mode {mode} unsupported")

    # -- scenes ----------------------------------------------------------------------
    def scene_create(self, variant=0, param=0, data_dir=None):
        if data_dir is None:
            data_dir = REF_DIR if (self.kind.startswith("ref") and os.path.isdir(REF_DIR)) else DATA_DIR
        s = self.lib.orc_scene_create(variant, param, data_dir.encode())
        if not s:
            raise RuntimeError("orc_scene_create failed")
        return s

    def scene_wrap(self, objects_ptr, n):
        """A scene from an Object[] in the reference's host layout built elsewhere (records copied, meshes shared)."""
        s = self.lib.orc_scene_wrap(objects_ptr, n)
        if not s:
            raise RuntimeError("orc_scene_wrap failed")
        return s

    def scene_free(self, s):
        self.lib.orc_scene_free(s)

    def scene_size(self, s):
        return int(self.lib.orc_scene_size(s))

    def scene_objects(self, s):
        """Raw pointer to Object[] in the reference's host layout (feeds trc_upload_scene)."""
        return self.lib.orc_scene_objects(s)

    def scene_object(self, s, i, with_mesh=True):
        m = np.zeros(16, np.float32); mi = np.zeros(16, np.float32)
        mesh_type = C.c_int(); counts = np.zeros(3, np.uint32)
        bmin = np.zeros(3, np.float32); bmax = np.zeros(3, np.float32)
        color = np.zeros(3, np.float32); ebn = np.zeros(3, np.float32); mat_type = C.c_int()
        self.lib.orc_scene_object_info(s, i, _fp(m), _fp(mi), C.addressof(mesh_type), _fp(counts),
                                       _fp(bmin), _fp(bmax), _fp(color), _fp(ebn), C.addressof(mat_type))
        d = dict(matrix=m, matrixi=mi, mesh_type=int(mesh_type.value), counts=counts, bmin=bmin,
                 bmax=bmax, color=color, emit=float(ebn[0]), brdf=float(ebn[1]), n=float(ebn[2]),
                 mat_type=int(mat_type.value))
        if with_mesh:
            v = np.zeros((int(counts[0]), 3), np.float32)
            n = np.zeros((int(counts[1]), 3), np.float32)
            f = np.zeros((int(counts[2]), 6), np.uint32)
            self.lib.orc_scene_object_mesh(s, i, _fp(v), _fp(n), _fp(f))
            d.update(verts=v, norms=n, faces=f)
        return d

    def scene_translate(self, s, i, amount):
        a = np.asarray(amount, np.float32); self.lib.orc_scene_translate(s, i, _fp(a))

    def scene_rotate(self, s, i, angle_deg, axis):
        a = np.asarray(axis, np.float32); self.lib.orc_scene_rotate(s, i, angle_deg, _fp(a))

    def scene_scale(self, s, i, amount):
        a = np.asarray(amount, np.float32); self.lib.orc_scene_scale(s, i, _fp(a))

    # -- camera ----------------------------------------------------------------------
    def camera_default(self, aspect_w=512, aspect_h=512, rot_x=0.0, rot_y=0.0):
        cam = Camera()
        self.lib.orc_camera_default(aspect_w, aspect_h, rot_x, rot_y, C.byref(cam))
        return cam

    # -- kernels ---------------------------------------------------------------------
    def raytrace1(self, s, w, h, cam, nthreads=1):
        n = w * h
        pbo = np.zeros(n, np.uint32); ids = np.zeros(n, np.int32); ts = np.zeros(n, np.float32)
        pos = np.zeros((n, 3), np.float32); nor = np.zeros((n, 3), np.float32)
        self.lib.orc_raytrace1(s, w, h, C.byref(cam), _fp(pbo), _fp(ids), _fp(ts), _fp(pos), _fp(nor), nthreads)
        return dict(pbo=pbo, ids=ids, ts=ts, pos=pos, nor=nor)

    def pathtrace(self, s, w, h, cam, seed, iter_first, n_iters, iter_stride=1, rng_mode=0,
                  film=None, want_pbo=True, nthreads=1):
        n = w * h
        if film is None:
            film = np.zeros((n, 3), np.float32)
        assert film.dtype == np.float32 and film.size == 3 * n and film.flags.c_contiguous
        pbo = np.zeros(n, np.uint32) if want_pbo else None
        st = Stats()
        self.lib.orc_pathtrace(s, w, h, C.byref(cam), seed, iter_first, iter_stride, n_iters, rng_mode,
                               _fp(film), _fp(pbo), C.byref(st), nthreads)
        return dict(film=film, pbo=pbo, stats=st.as_dict())

    def tonemap(self, film, iters):
        film = np.ascontiguousarray(film, np.float32)
        n = film.size // 3
        pbo = np.zeros(n, np.uint32)
        self.lib.orc_tonemap(_fp(film), n, float(iters), _fp(pbo))
        return pbo

    def mesh_exit_histogram(self, s, w, h, cam, obj, step, nthreads=1):
        """(C restatement only) exit stages of Ray.inl:189-226 over the (ray, face) pairs of a sub-sampled mode-1 frame."""
        counts = np.zeros(6, np.uint64)
        self.lib.orc_mesh_exit_histogram(s, w, h, C.byref(cam), obj, step, _fp(counts), nthreads)
        return dict(det=int(counts[0]), bary_x=int(counts[1]), bary_y=int(counts[2]), full=int(counts[3]),
                    rays_past_aabb=int(counts[4]), rays=int(counts[5]))

    # -- unit-level ------------------------------------------------------------------
    def intersect_scene(self, s, ray6):
        r = np.asarray(ray6, np.float32)
        t = C.c_float(); i = C.c_int(); pos = np.zeros(3, np.float32); nor = np.zeros(3, np.float32)
        self.lib.orc_intersect_scene(s, _fp(r), C.addressof(t), C.addressof(i), _fp(pos), _fp(nor))
        return np.float32(t.value), int(i.value), pos, nor

    def intersect_object(self, s, obj, ray6):
        r = np.asarray(ray6, np.float32)
        t = C.c_float(); pos = np.zeros(3, np.float32); nor = np.zeros(3, np.float32)
        self.lib.orc_intersect_object(s, obj, _fp(r), C.addressof(t), _fp(pos), _fp(nor))
        return np.float32(t.value), pos, nor

    def pick(self, s, cam, window_w, window_h, x, y):
        """main.cpp:180-208: (ray6, t, id, pos, nor) of the picking ray through window position (x, y)."""
        ray = np.zeros(6, np.float32); t = C.c_float(); i = C.c_int()
        pos = np.zeros(3, np.float32); nor = np.zeros(3, np.float32)
        self.lib.orc_pick(s, C.byref(cam), window_w, window_h, x, y, _fp(ray), C.addressof(t), C.addressof(i), _fp(pos), _fp(nor))
        return ray, np.float32(t.value), int(i.value), pos, nor

    def bounce(self, mat_type, mat_n, rd, nor, rnd):
        rd = np.asarray(rd, np.float32); nor = np.asarray(nor, np.float32); rnd = np.asarray(rnd, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_bounce(mat_type, mat_n, _fp(rd), _fp(nor), _fp(rnd), _fp(out))
        return out

    def reflectance(self, inc, nor, n1, n2):
        inc = np.asarray(inc, np.float32); nor = np.asarray(nor, np.float32)
        return np.float32(self.lib.orc_reflectance(_fp(inc), _fp(nor), n1, n2))

    def rand_vector(self, r1, r2):
        out = np.zeros(3, np.float32); self.lib.orc_rand_vector(r1, r2, _fp(out)); return out

    def rand_vector_hem(self, r1, r2, nor):
        nor = np.asarray(nor, np.float32); out = np.zeros(3, np.float32)
        self.lib.orc_rand_vector_hem(r1, r2, _fp(nor), _fp(out)); return out

    def rand_point_disk(self, r1, r2, r3):
        out = np.zeros(2, np.float32); self.lib.orc_rand_point_disk(r1, r2, r3, _fp(out)); return out

    def rand3(self, seed, index, it, dim=0):
        out = np.zeros(3, np.float32); self.lib.orc_rand3_api(seed, index, it, dim, _fp(out)); return out

    def philox(self, ctr, key):
        ctr = np.asarray(ctr, np.uint32); key = np.asarray(key, np.uint32); out = np.zeros(4, np.uint32)
        self.lib.orc_philox(_fp(ctr), _fp(key), _fp(out)); return out

    def rgb_to_int(self, r, g, b):
        return int(self.lib.orc_rgb_to_int(r, g, b))

    # -- portable math (C restatement only) ---------------------------------------------
    def pm_sincos(self, x):
        s = C.c_float(); c = C.c_float()
        self.lib.orc_pm_sincos(x, C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)

    def pm_acos(self, x):
        return np.float32(self.lib.orc_pm_acos(x))

    def pm_pow(self, x, y):
        return np.float32(self.lib.orc_pm_pow(x, y))
