"""Seeded random scene descriptions (tracer_amd scene-file grammar) for tests that go beyond the
reference's one hard-coded scene: many objects of every primitive type, non-uniform scales,
arbitrary rotations, all three materials, several meshes."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tracer_amd", "data")


def clutter_scene(path, seed=1, n_spheres=12, n_panels=6, n_cubes=3, n_meshes=3):
    """Cornell walls + ceiling light + a cloud of spheres, small planes ("panels"), cubes and triangle meshes."""
    r = np.random.default_rng(seed)
    L = ["camera fov 30 zoom -13 focal 13 lens 0.7",
         "mesh wall plane", "mesh box cube", "mesh ball sphere",
         f"mesh ico obj {os.path.join(DATA, 'icosahedron.obj')}", "mesh ball2 icosphere ico 2"]
    walls = [(None, (0, -5, 0), (1, 1, 1)), ((180, 1, 0, 0), (0, 5, 0), (1, 1, 1)), ((90, 1, 0, 0), (0, 0, -5), (1, 1, 1)),
             ((90, 0, 0, 1), (5, 0, 0), (0.5, 0.5, 1)), ((-90, 0, 0, 1), (-5, 0, 0), (1, 0.5, 0.5))]
    for rot, at, col in walls:
        L.append("object wall")
        if rot:
            L.append("  rotate %g %g %g %g" % rot)
        L += ["  scale 10 10 10", "  translate %g %g %g" % at, "  color %g %g %g" % col]
    L += ["object box", "  scale 5 0.5 5", "  translate 0 5 0", "  color 1 1 1", "  emit 2"]
    mats = ["diff", "diff", "mirr", "trans"]

    def material():
        m = mats[int(r.integers(len(mats)))]
        out = ["  color %.3f %.3f %.3f" % tuple(r.uniform(0.4, 1.0, 3)), "  material " + m]
        if m == "trans":
            out.append("  ior %.3f" % r.uniform(1.2, 1.8))
        return out

    def place(lo=-4.0, hi=4.0):
        return "  translate %.3f %.3f %.3f" % tuple(r.uniform(lo, hi, 3))

    def rotate():
        ax = r.normal(size=3); ax /= np.linalg.norm(ax)
        return "  rotate %.2f %.4f %.4f %.4f" % (r.uniform(-180, 180), *ax)

    for _ in range(n_spheres):
        L += ["object ball", "  scale %.3f %.3f %.3f" % tuple(r.uniform(0.6, 2.2, 3)), rotate(), place()] + material()
    for _ in range(n_panels):
        L += ["object wall", "  scale %.3f 1 %.3f" % tuple(r.uniform(1.0, 3.0, 2)), rotate(), place()] + material()
    for k in range(n_cubes):
        L += ["object box", "  scale %.3f %.3f %.3f" % tuple(r.uniform(0.5, 1.8, 3)), rotate(), place()] + material()
        if k == 0:
            L.append("  emit 1.5")
    for k in range(n_meshes):
        L += ["object " + ("ball2" if k % 2 else "ico"), "  scale %.3f" % r.uniform(0.8, 1.8), rotate(), place()] + material()
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


def aligned_scene(path, lens=0.7):
    """Objects whose matrices have exact zeros (no rotation, or a rotation about one coordinate axis), several
    with zero translation components and one at the origin, next to arbitrarily rotated ones: every zero
    pattern the kernels specialise on (tracer_amd/csrc/sparse_xform.h) for every primitive type that has it."""
    L = ["camera fov 30 zoom -13 focal 13 lens %g" % lens,
         "mesh wall plane", "mesh box cube", "mesh ball sphere",
         f"mesh ico obj {os.path.join(DATA, 'icosahedron.obj')}"]
    walls = [(None, (0, -5, 0), (1, 1, 1)), ((180, 1, 0, 0), (0, 5, 0), (1, 1, 1)), ((90, 1, 0, 0), (0, 0, -5), (1, 1, 1)),
             ((90, 0, 0, 1), (5, 0, 0), (0.5, 0.5, 1)), ((-90, 0, 0, 1), (-5, 0, 0), (1, 0.5, 0.5))]
    for rot, at, col in walls:
        L.append("object wall")
        if rot:
            L.append("  rotate %g %g %g %g" % rot)
        L += ["  scale 10 10 10", "  translate %g %g %g" % at, "  color %g %g %g" % col]
    L += ["object box", "  scale 5 0.5 5", "  translate 0 5 0", "  color 1 1 1", "  emit 2"]
    L += ["object ball", "  scale 3 3 3", "  color 0.9 0.9 0.9", "  material trans", "  ior 1.5",           # at the origin
          "object ball", "  scale 1 2 1", "  translate 2.5 0 -1", "  color 0.8 0.8 0.8", "  material mirr",
          "object ball", "  scale 1.5 1 2", "  rotate 30 0 1 0", "  translate -3 -3 1", "  color 0.9 0.6 0.3",
          "object ball", "  scale 1.2 1.2 1.2", "  rotate 50 0.3 0.5 0.8", "  translate 3 -3.5 2", "  color 0.3 0.6 0.9",
          "object wall", "  scale 3 1 2", "  rotate 40 0 1 0", "  translate 0 -2.5 0", "  color 0.7 0.9 0.7",
          "object wall", "  scale 2 1 2", "  rotate 90 1 0 0", "  translate -3 0 0", "  color 0.9 0.7 0.9",
          "object wall", "  scale 2 1 3", "  rotate 35 0 0 1", "  translate 3 2 0", "  color 0.9 0.9 0.5", "  material mirr",
          "object wall", "  scale 2 1 2", "  rotate 25 0.6 0.2 0.7", "  translate 0 2 -2", "  color 0.6 0.6 0.9",
          "object box", "  scale 1 1 1", "  translate 0 0 2.5", "  color 0.9 0.5 0.5",
          "object box", "  scale 1 2 1", "  rotate 45 0 1 0", "  translate -2 -4 -2", "  color 0.5 0.9 0.5", "  material trans", "  ior 1.4",
          "object ico", "  scale 1.2", "  translate -2 -3.5 0", "  color 1 1 0.3"]
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


def icosphere_obj(path, level=3, offset=(0.0, 0.0, 0.0), stretch=(1.0, 1.0, 1.0), radius=1.0, spike=1.0):
    """A subdivided icosahedron written in the loader's .obj dialect (v / vn / f a//b), its vertices stretched per
    axis and then moved by `offset` in OBJECT space -- a big mesh that does not sit at its own origin."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        mid = {}

        def midpoint(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p)); mid[key] = len(v) - 1
            return mid[key]
        nf = []
        for a, b, c in f:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    s, o = np.array(stretch, float), np.array(offset, float)
    v[0] = v[0] * spike                      # spike > 1: a few faces far larger than the rest
    with open(path, "w") as out:
        for p in v:
            q = p * radius * s + o
            out.write("v %.9g %.9g %.9g\n" % tuple(q))
        for p in v:
            n = p / s; n = n / np.linalg.norm(n)
            out.write("vn %.9g %.9g %.9g\n" % tuple(n))
        for a, b, c in f:
            out.write("f %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))
    return path, len(f)


def offset_mesh_scene(path, obj_path, offset, scale=2.0, lens=0.5):
    """Cornell walls + light + one big mesh whose object-space geometry sits at `offset`; the object is moved back
    so that it hangs in the middle of the box."""
    L = ["camera fov 30 zoom -13 focal 13 lens %g" % lens, "mesh wall plane", "mesh box cube", f"mesh blob obj {obj_path}"]
    walls = [(None, (0, -5, 0), (1, 1, 1)), ((180, 1, 0, 0), (0, 5, 0), (1, 1, 1)), ((90, 1, 0, 0), (0, 0, -5), (1, 1, 1)),
             ((90, 0, 0, 1), (5, 0, 0), (0.5, 0.5, 1)), ((-90, 0, 0, 1), (-5, 0, 0), (1, 0.5, 0.5))]
    for rot, at, col in walls:
        L.append("object wall")
        if rot:
            L.append("  rotate %g %g %g %g" % rot)
        L += ["  scale 10 10 10", "  translate %g %g %g" % at, "  color %g %g %g" % col]
    L += ["object box", "  scale 5 0.5 5", "  translate 0 5 0", "  color 1 1 1", "  emit 2"]
    # Object::scale multiplies from the right (object space), Object::rotate from the left (about the world origin)
    L += ["object blob", "  scale %g" % scale, "  translate %.9g %.9g %.9g" % tuple(-scale * np.array(offset, float)), "  rotate 20 0.2 1 0.1",
          "  translate 0.3 -1 0.5", "  color 0.9 0.8 0.3"]
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path


def stress_scene(path, seed=1):
    """Transforms at the edges of what the (ray, object) pre-test of intersect.h accepts (bound.h): tiny and huge
    objects, objects far from the origin, scale ratios around the conditioning limit of 64 (below it the bounding sphere
    is used, above it every ray keeps its pair), skew from a rotation between two non-uniform scales, meshes of both
    kinds; no walls, so rays also leave the scene."""
    r = np.random.default_rng(seed)
    L = ["camera fov 40 zoom -30 focal 30 lens 0.3",
         "mesh box cube", "mesh ball sphere", "mesh wall plane",
         f"mesh ico obj {os.path.join(DATA, 'icosahedron.obj')}", "mesh ball2 icosphere ico 1"]
    mats = ["diff", "mirr", "trans"]

    def rotate():
        ax = r.normal(size=3); ax /= np.linalg.norm(ax)
        return "  rotate %.2f %.4f %.4f %.4f" % (r.uniform(-180, 180), *ax)

    def finish(k):
        out = ["  color %.3f %.3f %.3f" % tuple(r.uniform(0.4, 1.0, 3)), "  material " + mats[k % 3]]
        if mats[k % 3] == "trans":
            out.append("  ior 1.5")
        return out

    L += ["object wall", "  scale 60 60 60", "  translate 0 -12 0", "  color 0.8 0.8 0.8"]
    L += ["object box", "  scale 8 0.5 8", "  translate 0 14 0", "  color 1 1 1", "  emit 3"]
    kinds = ["ball", "box", "ico", "ball2"]
    specs = [  # (scale xyz, second scale or None, translate)
        ((0.004, 0.004, 0.004), None, (0.5, 0.2, 3.0)),            # tiny
        ((0.02, 0.02, 0.02), None, (-0.3, -0.1, 6.0)),
        ((9.0, 9.0, 9.0), None, (0.0, -2.0, -14.0)),               # big
        ((1.0, 1.0, 1.0), None, (2000.0, 900.0, -4000.0)),         # far away (never hit; the pre-test must cope with |c|)
        ((2.0, 2.0, 2.0), None, (6.0, 3.0, -3.0)),
        ((6.0, 0.1, 6.0), None, (-5.0, 4.0, -2.0)),                # ratio 60: inside the limit
        ((7.0, 0.1, 3.0), None, (5.0, -4.0, -1.0)),                # ratio 70: outside, no bound
        ((0.1, 5.0, 0.1), None, (-6.0, -3.0, 1.0)),                # needle, ratio 50
        ((3.0, 0.3, 1.0), (1.0, 4.0, 0.5), (0.0, 5.0, 2.0)),       # skew: scale, rotate, scale
        ((1.5, 1.5, 1.5), (3.0, 0.2, 1.0), (-2.0, 0.0, 4.0)),
    ]
    k = 0
    for sc, sc2, at in specs:
        for kind in kinds:
            jitter = r.uniform(-0.8, 0.8, 3) * min(1.0, max(sc))
            L += ["object " + kind, "  scale %.6g %.6g %.6g" % sc, rotate()]
            if sc2:
                L += ["  scale %.6g %.6g %.6g" % sc2, rotate()]
            L += ["  translate %.6g %.6g %.6g" % tuple(np.array(at) + jitter)] + finish(k)
            k += 1
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    return path
