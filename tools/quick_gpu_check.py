"""Quick GPU parity probe (development aid; the real tests live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.pyoracle import Oracle
from tracer_amd import Tracer, Camera, PIPE_WAVEFRONT, PIPE_COMPACTED, PIPE_FUSED

o = Oracle("c")
s = o.scene_create(0)
ocam = o.camera_default()
cam = Camera.from_array(ocam.as_array())
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 256

tr = Tracer(W, H)
tr.upload_scene(o.scene_objects(s), o.scene_size(s))
tr.raytrace1(cam)
g = tr.read_hits(); ldr = tr.read_ldr()
r = o.raytrace1(s, W, H, ocam, nthreads=8)
print("mode1 ids equal:", np.array_equal(g['ids'], r['ids']), "mismatch", int((g['ids'] != r['ids']).sum()))
print("mode1 t bit-equal:", g['ts'].tobytes() == r['ts'].tobytes(), "nor:", g['nor'].tobytes() == r['nor'].tobytes(),
      "pos:", g['pos'].tobytes() == r['pos'].tobytes(), "ldr:", np.array_equal(ldr, r['pbo']))
print("mode1 ms", tr.last_render_ms())

for math_mode in (1, 0):
    o.set_math_mode(math_mode)
    for rng in (0, 1):
        ref = o.pathtrace(s, W, H, ocam, 5, 1, 3, rng_mode=rng, nthreads=8)
        for pipe in (0, 1, 2, 3):
            tr.set_pipeline(pipe); tr.set_rng_mode(rng); tr.reset_film(); tr.reset_counters()
            tr.pathtrace(cam, 5, 1, 3)
            film = tr.read_film(); l2 = tr.read_ldr(); c = tr.counters()
            d = np.abs(film - ref['film'])
            print(f"math={math_mode} rng={rng} pipe={pipe}: film bit-equal {film.tobytes()==ref['film'].tobytes()} "
                  f"maxabs {d.max():.3g} npix>1e-4 {(d.max(1)>1e-4).sum()} ldr_equal {np.array_equal(l2, ref['pbo'])} "
                  f"ldr maxdiff {np.abs(((l2[:,None]>>np.array([16,8,0]))&255).astype(int)-((ref['pbo'][:,None]>>np.array([16,8,0]))&255).astype(int)).max()} "
                  f"live {c['live_segments']} vs {ref['stats']['live_segments']} ms {tr.last_render_ms():.3f}")
