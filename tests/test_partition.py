"""Multi-GPU path on CPU: sample-index partition across ranks + ONE sum-reduce of the films
(tracer_amd/multi.py), exercised with world_size 2 over gloo. The per-rank compute is done by
the oracle here (no GPU in this container); what is under test is the partition/reduce logic
bench.py and the 8-GPU runs use."""
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


def _worker(rank, world, port, total_iters, w, h, seed, out_dir, by_tile=False):
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("c")
    s = o.scene_create(0)
    cam = o.camera_default(w, h)
    first, stride, count = multi.rank_iterations(total_iters, rank, world)
    film = np.zeros((w * h, 3), np.float32)
    if by_tile:
        # image-tile partition: this rank's band of rows of every iteration, zero elsewhere (the oracle
        # renders whole frames, so the band is cut out of one; on the GPU trc_set_tile renders only it)
        y0, y1 = multi.tile_rows(h, rank, world)
        whole = o.pathtrace(s, w, h, cam, seed, 1, total_iters, want_pbo=False)["film"].reshape(h, w, 3)
        film = film.reshape(h, w, 3); film[y0:y1] = whole[y0:y1]; film = film.reshape(w * h, 3)
    elif count:
        film = o.pathtrace(s, w, h, cam, seed, first, count, iter_stride=stride, film=film, want_pbo=False)["film"]
    t = torch.from_numpy(film)
    multi.reduce_film(t, dist, 0)
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), t.numpy())
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
