"""Debugging aid (GPU box): push every closest-hit query the oracle makes on one case of the integrator fixture through the
device's yafaray_intersectRays and list the queries the two answer differently.
usage: python tests/tools/integrator_case_rays.py <case name>"""
import sys

import numpy as np

sys.path.insert(0, ".")
from libyafaray_amd import Interface, scenes          # noqa: E402
from oracle import pyoracle as po                     # noqa: E402
from tests.integrator_fixture import case_scene, film_from_samples, load      # noqa: E402

doc = load("ieee")
name = sys.argv[1]
cs = next(c for c in doc["cases"] if c["name"] == name)
sc, rd = case_scene(doc, cs)
osc = po.OracleScene(sc)
film_o, st_o, smp, rays, n_rays = osc.render_traced(rd, 1 << 16, 1 << 17)
yi = Interface()
scenes.load_scene(yi, sc, rd)
yi.setRandState(cs["srand"], 0)
yi.render()
film = yi.getFilm(rd["width"], rd["height"])
st = yi.getRenderStats()
print("device rays", st.rays_closest, st.rays_shadow, "oracle", st_o.rays_closest, st_o.rays_shadow, "reference", cs["n_closest"], cs["n_shadow"])
want = film_from_samples(doc, cs)
bad = np.argwhere((film != want).any(axis=-1))
print("pixels that differ from the reference's film:", len(bad), bad[:12].tolist())
q = np.ascontiguousarray(rays[:, :8])
tri, t, bary = yi.intersectRays(q)
otri = rays[:, 9].view(np.int32)
diff = np.nonzero((tri != otri) | ((tri >= 0) & (t != rays[:, 8])))[0]
print("closest-hit queries answered differently:", len(diff), "of", len(q))
for i in diff[:20]:
    print(i, "from", rays[i, :3], "dir", rays[i, 3:6], "tmin/tmax", rays[i, 6:8], "oracle tri/t", otri[i], rays[i, 8], "device tri/t", tri[i], t[i],
          "brute", osc.intersect(rays[i, :3], rays[i, 3:6], float(rays[i, 6]), float(rays[i, 7]), use_tree=False)[1:3])
