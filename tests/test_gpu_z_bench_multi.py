"""bench.py's N > 1 control flow on ONE GPU: two ranks launched the way the driver launches them
(python -m torch.distributed.run), both on cuda:0, the film sum over gloo through the host
(TRC_BENCH_REHEARSAL=1 -- RCCL refuses two ranks on one device). Checks what the real multi-GPU run relies on:
the sample-index partition, the reduce to rank 0, the tonemap with the total sample count, the max-over-ranks
timing and the one JSON line; the numbers themselves mean nothing here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TRC_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=420)   # a cold first `import torch` on a fresh box can take minutes
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # rank 0 prints, once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and "rehearsal" in out
    assert out["film_check"]["ok"], out["film_check"]
    # configs[3]'s shape, a fixed total split over the ranks (strong scaling), through the same product function
    c4 = out["config4_strong"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["iterations_per_gpu"] * 2 == c4["total_iterations"] and c4["value"] > 0
    # both ranks rendered 300 iterations of 512x512 per step: the whole job's segments are twice one rank's
    per_gpu = out["config"]["live_segments_per_step_per_gpu"]
    assert abs(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 / (2 * per_gpu) - 1.0) < 0.01
