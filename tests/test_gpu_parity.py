"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from libyafaray_amd import Interface, scenes
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

RTOL = 1e-4     # BASELINE.json north_star: per-pixel RGB within 1e-4 relative
ABS_FLOOR = 1e-3


def compare_films(gpu_film, ora_film, what, max_outliers=0):
    a, b = po.film_to_rgb(gpu_film), po.film_to_rgb(ora_film)
    assert np.array_equal(gpu_film[..., 4], ora_film[..., 4]), f"{what}: film weights differ"
    rel = np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), ABS_FLOOR)
    worst = rel.max(axis=-1)
    n_bad = int((worst > RTOL).sum())
    exact = float((gpu_film == ora_film).all(axis=-1).mean())
    print(f"{what}: {n_bad}/{worst.size} pixels over {RTOL}, max rel {worst.max():.3g}, bit-exact pixels {exact:.4f}")
    assert np.array_equal(a[..., 3], b[..., 3]), f"{what}: alpha differs"
    assert n_bad <= max_outliers, f"{what}: {n_bad} pixels over tolerance (max rel {worst.max():.3g})"
    return n_bad, exact


def render_both(sc, rd):
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.render()
    film = yi.getFilm(rd["width"], rd["height"])
    st = yi.getRenderStats()
    osc = po.OracleScene(sc)
    ofilm, ost = osc.render(rd)
    return film, st, ofilm, ost


def test_ray_batches_match_oracle():
    sc = scenes.cornell_soup(3000, seed=11)
    yi = Interface()
    scenes.load_scene(yi, sc, scenes.render_settings(32, 32, 1))
    yi.prepareRender()
    osc = po.OracleScene(sc)
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform(-0.95, 0.95, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d.astype(np.float32), np.full((n, 1), 5e-5, np.float32), np.full((n, 1), -1.0, np.float32)], axis=1)
    rays[::7, 7] = rng.uniform(0.05, 1.0, size=rays[::7].shape[0])   # bounded rays too
    tri, t, bary = yi.intersectRays(rays)
    sh = yi.shadowRays(rays)
    mism = 0
    for i in range(n):
        h, oti, ot, ob = osc.intersect(rays[i, :3], rays[i, 3:6], float(rays[i, 6]), float(rays[i, 7]), use_tree=False)
        if (tri[i] >= 0) != bool(h) or (h and (tri[i] != oti or t[i] != ot or not np.array_equal(bary[i], ob))):
            mism += 1
        s = osc.is_shadowed(rays[i, :3], rays[i, 3:6], float(rays[i, 6]), float(rays[i, 7]), use_tree=False)
        if bool(s) != bool(sh[i]):
            mism += 1
    assert mism == 0, f"{mism} ray results differ from the brute-force oracle"


@pytest.mark.parametrize("n_tris,res,spp,bounces", [(12, 32, 4, 2), (500, 48, 16, 3), (5000, 64, 16, 3), (2000, 40, 64, 2), (800, 33, 3, 4)])
def test_render_matches_oracle(n_tris, res, spp, bounces):
    sc = scenes.cornell_soup(n_tris, seed=n_tris, res=(res, res))
    rd = scenes.render_settings(res, res, spp, bounces=bounces)
    film, st, ofilm, ost = render_both(sc, rd)
    assert st.camera_samples == res * res * spp
    assert st.rays_closest == ost.rays_closest and st.rays_shadow == ost.rays_shadow, "ray counts differ from the oracle"
    compare_films(film, ofilm, f"cornell {n_tris} tris {res}x{res} {spp}spp b{bounces}")
