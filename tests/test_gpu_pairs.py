"""The (ray, object) pre-test of intersect.h (bounding spheres from bound.h) must be CONSERVATIVE: it may only call a pair
a certain miss if the exact test of Ray.inl rejects it (and, for a mesh, if the ray fails the AABB, so that the count of
triangle tests stays the reference's). Checked with the library's checking builds (tracer_amd/variants, built by
__graft_entry__.build()): every pair is tested exactly and carries the pre-test's verdict.
  libtracer_pairs_check1.so  counts the pairs called a certain miss that hit after all        -> 0 in every scenario
  libtracer_pairs_check2.so  counts the pairs called a certain miss                           -> most pairs of most scenes
and every output (mode-1 hits, films of several pipelines, caller ray batches) of both equals the product build's bit for
bit. Scenarios: the Cornell variants, random cluttered scenes, the axis-aligned scene, and scenes with tiny, huge, far-away,
badly conditioned and skewed objects (tests/scene_gen.py: stress_scene); ray batches with non-unit directions, origins up
to 10^6 away, axis-aligned directions and origins on surfaces."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
VARIANTS = os.path.join(REPO, "tracer_amd", "variants")


def run_worker(lib, tmp):
    env = dict(os.environ)
    if lib:
        env["TRC_LIB"] = os.path.join(VARIANTS, lib)
        assert os.path.exists(env["TRC_LIB"]), "checking build missing: run __graft_entry__.build()"
    else:
        env.pop("TRC_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "pairs_check_worker.py"), str(tmp)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_pre_test_is_conservative(tmp_path):
    product = run_worker(None, tmp_path)
    viol = run_worker("libtracer_pairs_check1.so", tmp_path)
    skipped = run_worker("libtracer_pairs_check2.so", tmp_path)
    assert set(product) == set(viol) == set(skipped) and len(product) >= 10
    for name in product:
        assert viol[name]["tri_tests"] == 0, (name, viol[name]["tri_tests"])              # never a hit behind a "certain miss"
        assert viol[name]["digest"] == product[name]["digest"], name                       # same bits with every pair tested
        assert skipped[name]["digest"] == product[name]["digest"], name
        assert skipped[name]["tri_tests"] > 0, name                                         # and the pre-test does reject pairs
