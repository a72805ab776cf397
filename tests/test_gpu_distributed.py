"""Multi-pass (adaptive) anti-aliasing on a SHARDED frame, on the device: two ranks share one GPU here (the pool's boxes have
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


def _scene():
    from libyafaray_amd import scenes
    sc = scenes.cornell_soup(900, seed=29, res=(W, H))
    rd = scenes.render_settings(W, H, 3, bounces=2, tile_size=T, background=(0.05, 0.1, 0.2), **AA)
    return sc, rd


def _worker(rank, world, port, out_path):
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
    sc, rd = _scene()
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
        np.savez(out_path, film=film.numpy(), counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_multi_pass_anti_aliasing_equals_the_single_gpu_render(tmp_path, monkeypatch):
    monkeypatch.setenv("YAFGPU_PIPELINE", "wavefront")
    from libyafaray_amd import Interface, scenes
    sc, rd = _scene()
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.render()
    full, st = yi.getFilm(W, H), yi.getRenderStats()
    assert len(np.unique(np.round(full[..., 4]))) >= 2, "the adaptive passes did not single out any pixels"
    # without the exchange a sharded multi-pass render is refused
    y2 = Interface(strict=False)
    scenes.load_scene(y2, sc, rd)
    y2.setShard(0, 2)
    assert not y2.render() and "exchange" in y2.getLastError()
    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got["counts"].tolist() == [st.camera_samples, st.rays_closest, st.rays_shadow]
    assert np.array_equal(got["film"][..., 4], full[..., 4]), "the ranks sampled other pixels again than the single GPU"
    interior = np.ones((H, W), bool)
    interior[::T, :] = False; interior[:, ::T] = False
    assert np.array_equal(got["film"][interior], full[interior])
    np.testing.assert_allclose(got["film"], full, rtol=2.5e-7, atol=1e-7)
