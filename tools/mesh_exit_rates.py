#!/usr/bin/env python3
"""tools/mesh_exit_rates.py [step=16] -- (CPU, oracle) where the reference's triangle test leaves on configs[4].

Ray.inl:189-226 returns early for almost every (ray, face) pair: at :196 (|det| < EPS), :203 (bary.x outside [0, 1]) or
:208 (bary.y, bary.x + bary.y). SURVEY.md 8(d) prices the four stages at 20 / 30 / 46 / 74 flop; a roofline that charges
every pair the full 74 (BENCH_r02: frac 1.48) prices arithmetic the reference never does. This script counts the exit
stages with the C oracle on the configs[4] frame -- Cornell walls + light + the 1 310 720-face icosphere, 1024x1024,
mode 1 -- for the primary rays of every `step`-th pixel in x and y (all 1 310 720 faces for each ray that passes the
AABB), and writes the exit rates and the weighted flop per pair to profiles/r03_mesh_exit_rates.json. bench.py reads
that file (data, not the oracle) for the configs[4] roofline.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.pyoracle import Oracle  # noqa: E402

STAGE_FLOP = {"det": 20, "bary_x": 30, "bary_y": 46, "full": 74}


def main():
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    level, w, h = 8, 1024, 1024
    o = Oracle("c")
    s = o.scene_create(2, level)
    mesh = [i for i in range(o.scene_size(s)) if o.scene_object(s, i, with_mesh=False)["mesh_type"] == 0]
    assert len(mesh) == 1
    faces = int(o.scene_object(s, mesh[0], with_mesh=False)["counts"][2])
    t = time.time()
    c = o.mesh_exit_histogram(s, w, h, o.camera_default(w, h), mesh[0], step, nthreads=os.cpu_count() or 1)
    dt = time.time() - t
    pairs = sum(c[k] for k in STAGE_FLOP)
    assert pairs == c["rays_past_aabb"] * faces
    rates = {k: c[k] / pairs for k in STAGE_FLOP}
    flop = sum(rates[k] * STAGE_FLOP[k] for k in STAGE_FLOP)
    out = {"workload": f"BASELINE.json configs[4]: icosphere level {level} ({faces} faces), {w}x{h}, mode 1, default camera",
           "sample": f"primary rays of every {step}th pixel in x and y: {c['rays']} rays, {c['rays_past_aabb']} past the AABB, "
                     f"{pairs} (ray, face) pairs, counted by oracle/libtracer_oracle.so (orc_mesh_exit_histogram) in {dt:.1f} s",
           "reference": "Ray.inl:189-226; exits at :196 / :203 / :208 / none", "stage_flop": STAGE_FLOP,
           "pairs_by_exit_stage": {k: c[k] for k in STAGE_FLOP}, "exit_rates": rates,
           "fraction_of_rays_past_aabb": c["rays_past_aabb"] / c["rays"],
           "flop_per_pair_weighted": flop}
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "r03_mesh_exit_rates.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
