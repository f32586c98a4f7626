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
