"""Sharded frames on the device whose ranks depend on each other: multi-pass (adaptive) anti-aliasing, and the reference's serial light
counter (estimateOneDirectLight's correlative_sample_number_, which runs on from tile to tile: a rank's tile starts with the calls of
every tile before it, other ranks' included — the ranks exchange their per-tile counts between the record pass and the final pass).
Two ranks share one GPU here (the pool's boxes have
one; gloo carries the collective, staged through the host — on a multi-GPU node the same code runs over RCCL), each renders its
tiles, between passes the plane exchange (yafaray_setPlaneExchange, libyafaray_amd.parallel.plane_exchange) gives both the
whole frame's film for the noise detection, and the summed films must equal the single-GPU render: the same pixels sampled
again (weights exact), colours to the rounding of one addition on tile borders."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, T = 96, 80, 16
AA = dict(AA_passes=3, AA_inc_samples=2, AA_threshold=0.02)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene(case):
    from libyafaray_amd import scenes
    if case == "aa":
        sc = scenes.cornell_soup(900, seed=29, res=(W, H))
        rd = scenes.render_settings(W, H, 3, bounces=2, tile_size=T, background=(0.05, 0.1, 0.2), **AA)
    else:
        # two area lights (the light counter picks one per estimate) and Russian roulette from the first bounce on (the per-tile
        # stream decides which paths make further estimates); "lights_aa" carries the counter over adaptive passes as well
        sc = scenes.cornell_soup(900, seed=31, res=(W, H), n_lights=1 if case == "lights_rr_only" else 2)
        rr = {} if case == "lights_lc_only" else dict(russian_roulette_min_bounces=1)
        rd = scenes.render_settings(W, H, 3, bounces=4, tile_size=128 if case == "lights_one_tile" else T, background=(0.05, 0.1, 0.2), **rr, **(AA if case == "lights_aa" else {}))
    return sc, rd


def _worker(rank, world, port, out_path, case):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["YAFGPU_PIPELINE"] = "wavefront"
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from libyafaray_amd import Interface, scenes
    from libyafaray_amd.parallel import plane_exchange, reduce_film
    sc, rd = _scene(case)
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.setShard(rank, world)
    yi.setPlaneExchange(plane_exchange())
    yi.render()
    film = torch.from_numpy(yi.getFilm(W, H).copy())
    reduce_film(film, dst=0)
    st = yi.getRenderStats()
    counts = torch.tensor([st.camera_samples, st.rays_closest, st.rays_shadow], dtype=torch.int64)
    dist.reduce(counts, dst=0)
    if rank == 0:
        np.savez(out_path, film=film.numpy(), counts=counts.numpy(), rand_state=np.array(yi.getRandState()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", ["aa", "lights_rr_only", "lights_lc_only", "lights", "lights_aa", "lights_chunks", "lights_one_tile"])
def test_sharded_render_with_cross_rank_state_equals_the_single_gpu_render(tmp_path, monkeypatch, case):
    monkeypatch.setenv("YAFGPU_PIPELINE", "wavefront")
    if case == "lights_chunks":      # several chunks per rank (a tile each): the count pass and the final pass record a chunk's events twice
        monkeypatch.setenv("YAFGPU_WF_CHUNK", "1500")
    from libyafaray_amd import Interface, scenes
    sc, rd = _scene(case)
    # the single-GPU render, in a fresh process like the ranks below: the reference's material / object constructors number
    # themselves from a process-wide counter and seed libc's rand() with it (material.cc:53-66), which the per-tile roulette
    # streams continue from — a process that has rendered before starts elsewhere
    one = str(tmp_path / "single.npz")
    mp.spawn(_worker, args=(1, _free_port(), one, case), nprocs=1, join=True)
    single = np.load(one)
    full, counts = single["film"], single["counts"].tolist()
    if not case.startswith("lights") or case == "lights_aa":
        assert len(np.unique(np.round(full[..., 4]))) >= 2, "the adaptive passes did not single out any pixels"
        # without the exchange a sharded multi-pass render is refused
        y2 = Interface(strict=False)
        scenes.load_scene(y2, sc, rd)
        y2.setShard(0, 2)
        assert not y2.render() and "exchange" in y2.getLastError()
    else:
        # the single-GPU render is the exact replay of the reference's serial state: it equals the single-threaded oracle
        from oracle import pyoracle as po
        from tests.test_gpu_parity import compare_films
        seed, skip = (int(x) for x in single["rand_state"])
        ofilm, ost = po.OracleScene(sc).render(dict(rd, oracle_threads=1, rand_srand=seed, rand_skip=skip))
        assert counts[1:] == [ost.rays_closest, ost.rays_shadow]
        compare_films(full, ofilm, "two lights + roulette, one GPU", exact_weights=True)
    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, case), nprocs=2, join=True)
    got = np.load(out)
    assert got["counts"].tolist() == counts
    assert np.array_equal(got["film"][..., 4], full[..., 4]), "the ranks sampled other pixels again than the single GPU"
    interior = np.ones((H, W), bool)
    if case != "lights_one_tile":        # (one tile: rank 1 has none — it still joins the exchange — and rank 0's film is the whole frame)
        interior[::T, :] = False; interior[:, ::T] = False
    assert np.array_equal(got["film"][interior], full[interior])
    np.testing.assert_allclose(got["film"], full, rtol=2.5e-7, atol=1e-7)
