"""The N > 1 path on CPU: two gloo ranks each render their tile shard (with the CPU oracle standing in
for the device render — this test is about the sharding and the reduce, not the kernels), the film
is sum-reduced to rank 0 with the same helper bench.py uses, and must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libyafaray_amd import scenes
    from libyafaray_amd.parallel import reduce_film, shard_of_tile
    from oracle import pyoracle as po
    sc = scenes.cornell_soup(400, seed=21, res=(48, 40))
    rd = scenes.render_settings(48, 40, 8, bounces=2, tile_size=16, shard_index=rank, shard_count=world)
    film, st = po.OracleScene(sc).render(rd)
    # ownership: a rank's samples only come from its own tiles
    ntx = (48 + 15) // 16
    own = np.zeros((40, 48), bool)
    for t in range(ntx * ((40 + 15) // 16)):
        if shard_of_tile(t, world) == rank:
            tx, ty = t % ntx, t // ntx
            own[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    assert st.camera_samples == own.sum() * 8
    t = torch.from_numpy(film.copy())
    reduce_film(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_shard_reduce_equals_single_process(tmp_path):
    out = str(tmp_path / "film.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    merged = np.load(out)
    from libyafaray_amd import scenes
    from oracle import pyoracle as po
    sc = scenes.cornell_soup(400, seed=21, res=(48, 40))
    full, _ = po.OracleScene(sc).render(scenes.render_settings(48, 40, 8, bounces=2, tile_size=16))
    assert np.array_equal(merged[..., 4], full[..., 4])                 # weights: exact
    same = (merged == full).all(axis=-1)
    assert same.mean() > 0.97                                           # only border-splat pixels may differ in the last bit
    np.testing.assert_allclose(merged, full, rtol=3e-7, atol=1e-7)
