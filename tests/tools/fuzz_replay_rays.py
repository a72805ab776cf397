"""GPU-box helper: for one pixel of a fuzz scene, log the oracle's closest-hit queries and replay them on the GPU's ray API and brute force."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libyafaray_amd import Interface, interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_parity import _feature_mix
seed, px, py = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
sc, rd, w, h, base, kw = _feature_mix(seed)
rd = dict(rd, AA_passes=1, xstart=px, ystart=py, width=1, height=1, oracle_threads=1, **extra)
tree = interface.build_kdtree(sc["verts"], threads=4)[:3]
os.environ["YOR_LOG_RAYS"] = "/tmp/rays.bin"
osc = po.OracleScene(sc); osc.set_tree(*tree)
ofilm, ost = osc.render(rd)
os.environ["YOR_LOG_RAYS"] = ""
osc.render(dict(rd, AA_minsamples=1))         # closes the log
log = np.fromfile("/tmp/rays.bin", dtype=np.float32).reshape(-1, 10)
tri_o = log[:, 9].copy().view(np.int32)
yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
film, st = yi.getFilm(1, 1), yi.getRenderStats()
print("film gpu", film.reshape(-1), "oracle", ofilm.reshape(-1), "closest rays", st.rays_closest, ost.rays_closest, "logged", len(log))
rays = log[:, :8].copy()
tri, t, bary = yi.intersectRays(rays)
for i in range(len(log)):
    hb = osc.intersect(rays[i, :3], rays[i, 3:6], float(rays[i, 6]), float(rays[i, 7]), use_tree=False)
    flag = "" if (tri[i] == tri_o[i] and (tri[i] < 0 or t[i] == log[i, 8])) else "   <<<<<< GPU ray API differs from the oracle's tree walk"
    print(i, "tmin %.3g tmax %.3g" % (rays[i, 6], rays[i, 7]), "oracle tree:", tri_o[i], log[i, 8], "| gpu:", tri[i], t[i], "| brute:", (hb[1] if hb[0] else -1), (hb[2] if hb[0] else -1.0), flag)
