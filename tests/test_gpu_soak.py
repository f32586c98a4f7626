"""A fixed slice of the parity soaks (tools/soak.py, tools/soak_mesh.py) as a test: 50 seeded random scenes through all four
pipelines against the C oracle, and 50 random big meshes through the big-mesh pass with and without its bounding-sphere
pre-test -- each once with the product library and once with the checking build of the conservative pre-tests
(-DTRC_PAIRS_CHECK=1: every (ray, object) pair is tested exactly, and a pair that the pre-test, or the mesh hull cull, called
a certain miss and that hits after all is counted). A regression in bound.h's or bvh.h's margins turns this red; the long
runs of the same tools are recorded in profiles/r03_soak*.json. Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(REPO, "tracer_amd", "variants", "libtracer_pairs_check1.so")


def run_tool(tool, first, n, tmp_path, lib=None):
    rec = str(tmp_path / (tool + ("_check" if lib else "") + ".json"))
    env = dict(os.environ)
    env.pop("TRC_LIB", None)
    if lib:
        env["TRC_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool + ".py"), str(first), str(n), "--json", rec],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(rec))


@pytest.mark.parametrize("checking", [False, True])
def test_soak_slice_random_scenes(tmp_path, checking):
    assert os.path.exists(CHECK), "checking build missing: make -C tracer_amd/csrc"
    rec = run_tool("soak", 7000, 50, tmp_path, CHECK if checking else None)
    assert rec["scenes"] == 50 and rec["films_compared"] == 200
    assert rec["mismatches"] == 0 and rec["pre_test_violations"] == 0
    assert rec["checking_build"] is checking and rec["segments_compared"] > 1_000_000


@pytest.mark.parametrize("checking", [False, True])
def test_soak_slice_big_meshes(tmp_path, checking):
    rec = run_tool("soak_mesh", 7000, 50, tmp_path, CHECK if checking else None)
    assert rec["meshes"] == 50 and rec["mismatches"] == 0 and rec["pre_test_violations"] == 0
    assert rec["checking_build"] is checking and rec["mesh_pixels_compared"] > 50_000
