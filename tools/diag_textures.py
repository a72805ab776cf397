"""diagnostic (GPU): where do textured renders differ from the oracle in the last bits?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YAFGPU_PIPELINE"] = "wavefront"
import numpy as np
from libyafaray_amd import Interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_textures import _textured_box

def run(sc, rd, what):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film = yi.getFilm(rd["width"], rd["height"]); st = yi.getRenderStats()
    seed, skip = yi.getRandState()
    ofilm, ost = po.OracleScene(sc).render(dict(rd, oracle_threads=1, rand_srand=seed, rand_skip=skip))
    diff = (film != ofilm).any(axis=-1)
    a, b = po.film_to_rgb(film), po.film_to_rgb(ofilm)
    rel = (np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), 1e-3)).max(axis=-1)
    print(f"{what}: rays {st.rays_closest}/{ost.rays_closest} {st.rays_shadow}/{ost.rays_shadow}; differing pixels {int(diff.sum())}/{diff.size}; max rel {rel.max():.3g}; over 1e-4: {int((rel > 1e-4).sum())}", flush=True)
    return film, ofilm

base = _textured_box()
plain = scenes.cornell_soup(400, seed=5, sigma=0.12, res=(48, 40))["materials"]
rd = scenes.render_settings(48, 40, 4, integrator="directlighting", transpShad=True, shadowDepth=3)
for keep in ([], [0], [1], [2], [4], [0, 1, 2, 4]):
    sc = copy.deepcopy(base)
    for k in (0, 1, 2, 4):
        if k not in keep:
            sc["materials"][k] = {kk: v for kk, v in base["materials"][k].items() if kk != "nodes" and not kk.endswith("_shader")}
    run(sc, rd, f"DL textured {keep}")
# material 1 and 2 slot by slot
for k in (1, 2, 4):
    m = base["materials"][k]
    slots = [kk for kk in m if kk.endswith("_shader")]
    for s in slots:
        sc = copy.deepcopy(base)
        for j in (0, 1, 2, 4):
            mm = {kk: v for kk, v in base["materials"][j].items() if not kk.endswith("_shader")}
            if j != k:
                mm.pop("nodes", None)
            else:
                mm[s] = m[s]
            sc["materials"][j] = mm
        run(sc, rd, f"DL material {k} slot {s}")
rd = scenes.render_settings(48, 40, 4, integrator="pathtracing", bounces=4, russian_roulette_min_bounces=1)
film, ofilm = run(base, rd, "PT rr=1")
ys, xs = np.nonzero((np.abs(film[..., :3] - ofilm[..., :3]) > 1e-4 * np.maximum(np.abs(ofilm[..., :3]), 1e-3)).any(axis=-1))
print("pixels beyond 1e-4:", list(zip(ys.tolist(), xs.tolist()))[:10])
for y, x in list(zip(ys.tolist(), xs.tolist()))[:4]:
    print(y, x, film[y, x], ofilm[y, x])
