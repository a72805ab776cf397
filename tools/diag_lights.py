import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YAFGPU_PIPELINE"] = "wavefront"
import numpy as np
from libyafaray_amd import Interface, scenes, interface
from oracle import pyoracle as po
sc = scenes.cornell_soup(120, seed=51, res=(24, 20), sigma=0.1)
rng = np.random.default_rng(7)
points = [{"type": "pointlight", "from": tuple(float(x) for x in rng.uniform(-0.8, 0.8, 3)), "color": (1.0, 0.9, 0.8), "power": 0.02} for _ in range(255)]
rd = scenes.render_settings(24, 20, 1, integrator="directlighting")
for nl in (254, 100, 30):
    s2 = dict(sc, lights=list(sc["lights"]) + points[:nl])
    yi = Interface(); scenes.load_scene(yi, s2, rd); yi.render(); film = yi.getFilm(24, 20); st = yi.getRenderStats()
    osc = po.OracleScene(s2); ofilm, ost = osc.render(rd)
    osc.set_tree(*interface.build_kdtree(s2["verts"], threads=4)[:3]); pfilm, _ = osc.render(rd)
    a, b, c = po.film_to_rgb(film), po.film_to_rgb(ofilm), po.film_to_rgb(pfilm)
    rel = (np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), 1e-3)).max(axis=-1)
    rel2 = (np.abs(a[..., :3] - c[..., :3]) / np.maximum(np.abs(c[..., :3]), 1e-3)).max(axis=-1)
    ys, xs = np.nonzero(rel > 1e-4)
    print(nl, "lights: bad vs own-tree oracle", len(ys), "vs product-tree oracle", int((rel2 > 1e-4).sum()), "shadow rays", st.rays_shadow, ost.rays_shadow)
    for y, x in zip(ys.tolist(), xs.tolist()):
        print("   ", y, x, a[y, x, :3], b[y, x, :3], c[y, x, :3])
# which single light makes the difference at the first bad pixel?
