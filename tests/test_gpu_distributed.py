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


def test_c4_as_stated_in_eight_shards_equals_the_unsharded_frame():
    """BASELINE.json configs[4] short of RCCL: the 1M-triangle glossy scene with its TWO area lights, 1024x1024 (2 spp), rendered as the
    eight shards of an 8-GPU node one after the other on this one GPU.  The ranks' only exchange on this scene — the per-tile call
    counts of the serial light counter, an all-reduce between the record pass and the final pass — is emulated in-process: a first
    round of renders collects every rank's contribution (a rank's own table depends on its own tiles only), a second round hands
    every rank the sum, exactly what ncclAllReduce returns to all of them.  The eight films, summed as yafaray_reduceFilm sums them,
    must be the unsharded device frame — which test_full_size_c4_two_lights_exact_replay pins to the single-threaded oracle:
    same ray counts, weights exact, interior pixels bit for bit, tile-border pixels to the rounding of one addition."""
    import torch
    import bench
    from libyafaray_amd import Interface, scenes
    from libyafaray_amd.parallel import _DeviceFloats
    WORLD = 8
    w, sc, rd = bench.make_workload("c4", spp=2)
    Wd, Hd, T = rd["width"], rd["height"], rd["tile_size"]
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.render()
    full, st_full = yi.getFilm(Wd, Hd).copy(), yi.getRenderStats()
    contrib = {}          # (rank, k-th exchange of its render) -> this rank's table
    state = {"rank": 0, "k": 0, "phase": 0}

    def exchange(ptr, n):
        t = torch.as_tensor(_DeviceFloats(ptr, n), device=torch.device("cuda", 0))
        key = state["k"]
        state["k"] += 1
        if state["phase"] == 0:
            contrib[(state["rank"], key)] = t.clone()
        else:
            total = sum(contrib[(r, key)] for r in range(WORLD))
            assert torch.equal(contrib[(state["rank"], key)], t), "a rank's own contribution changed between the rounds"
            t.copy_(total)
        torch.cuda.synchronize()

    yi.setPlaneExchange(exchange)
    parts = []
    for phase in (0, 1):
        state["phase"] = phase
        for r in range(WORLD):
            state["rank"], state["k"] = r, 0
            yi.setShard(r, WORLD)
            yi.render()
            if phase == 1:
                parts.append((yi.getFilm(Wd, Hd).copy(), yi.getRenderStats()))
    assert contrib, "the light-counter exchange never ran"
    assert sum(p[1].camera_samples for p in parts) == st_full.camera_samples
    assert (sum(p[1].rays_closest for p in parts), sum(p[1].rays_shadow for p in parts)) == (st_full.rays_closest, st_full.rays_shadow)
    total = np.zeros_like(full)
    for f, _ in parts:
        total = total + f
    assert np.array_equal(total[..., 4], full[..., 4]), "weights differ"
    interior = np.ones((Hd, Wd), bool)
    interior[::T, :] = False; interior[:, ::T] = False
    assert np.array_equal(total[interior], full[interior]), "interior pixels differ from the unsharded frame"
    np.testing.assert_allclose(total, full, rtol=2.5e-7, atol=1e-7)


def test_rccl_communicator_of_the_c_abi_world_one():
    """yafaray_commCreate / yafaray_reduceFilm / yafaray_allReduce (csrc/yafaray_reduce.cpp) on the one GPU a box has: a
    communicator of one rank through real RCCL calls (ncclCommInitRank, ncclReduce, ncclAllReduce in place) leaves the film as it is;
    attached to an interface (yafaray_setComm) it serves the exchange of a 'sharded' one-rank render."""
    import torch
    from libyafaray_amd import Interface, scenes
    from libyafaray_amd.parallel import FilmComm
    torch.cuda.set_device(0)
    comm = FilmComm(0, 1, 0)
    assert "rccl" in comm.backend
    film = torch.rand((64, 48, 5), device="cuda")
    keep = film.clone()
    comm.reduce_film(film, dst=0)
    comm.all_reduce(film)
    torch.cuda.synchronize()
    assert torch.equal(film, keep)
    sc = scenes.cornell_soup(600, seed=5, res=(48, 40), n_lights=2)
    rd = scenes.render_settings(48, 40, 2, bounces=3, tile_size=16)
    films = []
    for attach in (False, True):
        yi = Interface()
        scenes.load_scene(yi, sc, rd)
        yi.setRandState(3, 0)
        if attach:
            yi.setComm(comm)
        yi.render()
        films.append(yi.getFilm(48, 40).copy())
        yi.setComm(None)
    assert np.array_equal(films[0], films[1])
    comm.close()
