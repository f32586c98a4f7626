#!/usr/bin/env python3
"""tools/ab.py -- A/B timing of library builds on the headline workload (development aid). Each build runs in its own
process (TRC_LIB selects it), alternating A, B, A, B ... so that clock / thermal drift hits both; per build the minimum
and median k_queue duration (HIP events on the launch stream) over all its launches are printed.
    python3 tools/ab.py [--reps 12] [--rounds 3] [--pipe 3] name=path/to/lib.so ...      (path '-' = the in-tree build)"""
import argparse
import json
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import tracer_amd as ta
pipe, reps, variant, w, h, iters, clutter = [int(x) for x in sys.argv[1:8]]
if clutter:
    sys.path.insert(0, os.path.join(%r, "tests"))
    from scene_gen import clutter_scene
    sc = ta.Scene(path=clutter_scene("/tmp/ab_clutter_%%d.scene" %% clutter, clutter), aspect=(w, h)); cam = sc.camera
else:
    sc = ta.Scene(variant); cam = ta.camera_default(w, h)
tr = ta.Tracer(w, h); tr.set_pipeline(pipe); tr.upload_scene(sc.objects_ptr, sc.size)
ms = []
for r in range(reps + 3):
    tr.pathtrace(cam, 1, 1, iters)
    if r >= 3: ms.append(tr.last_kernel_ms())
tr.reset_counters(); tr.pathtrace(cam, 1, 1, iters); c = tr.counters()
print(json.dumps({"ms": ms, "segs": c["live_segments"], "mhz": c["shader_mhz"]}))
''' % (HERE, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--pipe", type=int, default=3)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--size", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--clutter", type=int, default=0, help="seed of a random 29-object scene (tests/scene_gen.py) instead of the Cornell box")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    libs = [x.split("=", 1) for x in a.libs]
    res = {n: [] for n, _ in libs}
    info = {}
    for _ in range(a.rounds):
        for name, path in libs:
            env = dict(os.environ)
            if path != "-":
                env["TRC_LIB"] = os.path.abspath(path)
            else:
                env.pop("TRC_LIB", None)
            out = subprocess.run([sys.executable, "-c", CHILD, str(a.pipe), str(a.reps), str(a.variant), str(a.size[0]), str(a.size[1]), str(a.iters), str(a.clutter)],
                                 env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(name, "FAILED", out.stderr[-2000:]); sys.exit(1)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[name] += d["ms"]; info[name] = d
    for name, _ in libs:
        v = res[name]
        print(f"{name:14s} min {min(v):.3f} ms  median {statistics.median(v):.3f} ms  max {max(v):.3f}  ({len(v)} launches, "
              f"{info[name]['segs'] / min(v) / 1e6:.2f} Gseg/s at the min, clock {info[name]['mhz'] or 0:.0f} MHz)", flush=True)


if __name__ == "__main__":
    main()
