"""oracle/cpu_worker.py -- TEST INFRASTRUCTURE: one CPU process of bench.py's `cpu_baseline` leg.

    python oracle/cpu_worker.py KIND W H SEED ITER_FIRST ITER_STRIDE N_ITERS START_AT

Renders iterations ITER_FIRST, ITER_FIRST + ITER_STRIDE, ... (N_ITERS of them) of the mode-2 Cornell
workload with oracle KIND ("ref" = oracle/_ref/libtracer_ref.so, the reference's own .inl + GLM host build,
single-threaded by construction; "c" = the plain-C restatement on one thread) and prints one JSON line with
the live segments it traced and the wall-clock window of the render. bench.py starts one such process per
host core on disjoint iteration strides, all released at the same START_AT (unix time), and divides the sum
of the segments by the span from START_AT to the last process's end: the reference's CPU path
(Ray.inl:66-84 under the restated kernel bodies of raytrace.cu:64-177) on all the cores of the box.

Touches no GPU: it imports numpy and ctypes only.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.pyoracle import Oracle  # noqa: E402


def main(argv):
    kind = argv[1]
    w, h, seed, first, stride, n = (int(x) for x in argv[2:8])
    start_at = float(argv[8])
    o = Oracle(kind)
    if kind == "c":
        o.set_math_mode(Oracle.MATH_LIBM)
    # the reference build reads data/icosahedron.obj below the directory it is given: the product's generated copy
    s = o.scene_create(0, 0, data_dir=os.path.join(os.path.dirname(HERE), "tracer_amd", "data"))
    cam = o.camera_default(w, h)
    o.pathtrace(s, 64, 64, o.camera_default(64, 64), seed, 1, 1, want_pbo=False)      # page the code in
    late = time.time() > start_at
    while time.time() < start_at:
        time.sleep(0.001)
    t0 = time.time()
    r = o.pathtrace(s, w, h, cam, seed, first, n, iter_stride=stride, want_pbo=False)
    t1 = time.time()
    print(json.dumps({"t0": t0, "t1": t1, "late": late, "live_segments": r["stats"]["live_segments"],
                      "paths": r["stats"]["paths"]}), flush=True)


if __name__ == "__main__":
    main(sys.argv)
