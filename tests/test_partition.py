"""Multi-GPU path on CPU: sample-index partition across ranks + ONE sum-reduce of the films
(tracer_amd/multi.py: render_partitioned, the function bench.py and the 8-GPU runs call), exercised
with world_size 2 and 3 over gloo. The per-rank compute is done by the oracle behind a stand-in for
the Tracer (no GPU in this container); what is under test is the partition / ordering / reduce logic."""
import os
import socket

import numpy as np
import pytest

from tracer_amd import multi


def test_rank_iterations_cover_exactly_once():
    for total in (0, 1, 5, 300, 4096):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, stride, count = multi.rank_iterations(total, r, world)
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(1, total + 1)), (total, world)
    with pytest.raises(ValueError):
        multi.rank_iterations(10, 2, 2)


def test_tile_rows_cover_exactly_once():
    for h in (1, 17, 512, 2160):
        for world in (1, 2, 4, 8):
            rows = []
            for r in range(world):
                lo, hi = multi.tile_rows(h, r, world, align=1)
                rows += list(range(lo, hi))
            assert rows == list(range(h))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleTracer:
    """Stands in for tracer_amd.Tracer on a machine without a GPU: the calls multi.render_partitioned makes
    (pathtrace into the bound film, set_tile, synchronize, tonemap), computed by the oracle into a CPU tensor.
    What is under test is the product function's partition / ordering / reduce logic, not this stub."""

    def __init__(self, oracle, w, h, film_tensor):
        self.o, self.w, self.h = oracle, w, h
        self.s = oracle.scene_create(0)
        self.film = film_tensor.numpy().reshape(w * h, 3)      # shares memory with the tensor
        self.tile = None
        self.stream_handle = None
        self.calls = []

    def set_tile(self, row0, rows):
        self.tile = (row0, rows) if rows else None
        self.calls.append(("tile", row0, rows))

    def set_skip_tonemap(self, on):
        self.skip_tonemap = bool(on)
        self.calls.append(("skip_tonemap", bool(on)))

    def pathtrace(self, cam, seed, iter_first, n_iters, iter_stride=1):
        self.calls.append(("pathtrace", iter_first, n_iters, iter_stride, getattr(self, "skip_tonemap", False)))
        if self.tile is None:
            self.o.pathtrace(self.s, self.w, self.h, cam, seed, iter_first, n_iters, iter_stride=iter_stride,
                             film=self.film, want_pbo=False)
        else:     # the oracle renders whole frames: cut this band out of one (trc_set_tile renders only the band)
            whole = np.zeros((self.w * self.h, 3), np.float32)
            self.o.pathtrace(self.s, self.w, self.h, cam, seed, iter_first, n_iters, iter_stride=iter_stride,
                             film=whole, want_pbo=False)
            y0, rows = self.tile
            self.film[y0 * self.w:(y0 + rows) * self.w] += whole[y0 * self.w:(y0 + rows) * self.w]

    def synchronize(self):
        self.calls.append(("sync",))

    def tonemap(self, iters):
        self.calls.append(("tonemap", iters))


def _worker(rank, world, port, total_iters, w, h, seed, out_dir, by_tile=False):
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("c")
    film = torch.full((w * h * 3,), 3.0, dtype=torch.float32)      # garbage: render_partitioned must zero it
    tr = _OracleTracer(o, w, h, film)
    multi.render_partitioned(tr, o.camera_default(w, h), seed, total_iters, film, rank, world, dist, by_tile=by_tile)
    names = [c[0] for c in tr.calls]
    assert names.index("sync") > names.index("pathtrace")           # the render is awaited before the reduce
    assert (("tonemap", total_iters) in tr.calls) == (rank == 0)
    assert tr.tile is None                                          # the tile is always reset
    # the ranks' own display transforms are off during the render (rank 0 tonemaps the REDUCED film) and back on afterwards
    assert all(c[4] for c in tr.calls if c[0] == "pathtrace") and tr.skip_tonemap is False
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), film.numpy().reshape(w * h, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_equals_single_rank(oracle_c, tmp_path):
    import torch.multiprocessing as mp
    w, h, total, seed = 48, 40, 7, 5
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, w, h, seed, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "film.npy")
    s = oracle_c.scene_create(0)
    cam = oracle_c.camera_default(w, h)
    ref = oracle_c.pathtrace(s, w, h, cam, seed, 1, total)["film"]
    # same samples, different summation order: (odd iterations) + (even iterations)
    tol = 1e-5 * np.sqrt(total) * max(1.0, float(ref.max()))
    assert np.abs(got - ref).max() <= tol
    # and exactly the sum of the two strided partial films
    a = oracle_c.pathtrace(s, w, h, cam, seed, 1, 4, iter_stride=2)["film"]
    b = oracle_c.pathtrace(s, w, h, cam, seed, 2, 3, iter_stride=2)["film"]
    assert got.tobytes() == (a + b).astype(np.float32).tobytes()


def test_three_rank_tile_partition_equals_single_rank(oracle_c, tmp_path):
    """Disjoint bands of rows over zeroed films, one sum-reduce: bit for bit the single-rank film."""
    import torch.multiprocessing as mp
    w, h, total, seed = 40, 31, 3, 8
    port = _free_port()
    mp.spawn(_worker, args=(3, port, total, w, h, seed, str(tmp_path), True), nprocs=3, join=True)
    got = np.load(tmp_path / "film.npy")
    s = oracle_c.scene_create(0)
    ref = oracle_c.pathtrace(s, w, h, oracle_c.camera_default(w, h), seed, 1, total)["film"]
    assert got.tobytes() == ref.tobytes()
