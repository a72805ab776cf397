import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libyafaray_amd import Interface, interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_parity import _feature_mix
seed, px, py = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc, rd, w, h, base, kw = _feature_mix(seed)
tree = interface.build_kdtree(sc["verts"], threads=4)[:3]
def both(rd):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film = yi.getFilm(rd["width"], rd["height"])
    osc = po.OracleScene(sc); osc.set_tree(*tree)
    ofilm, ost = osc.render(rd)
    return film, ofilm
def report(tag, rd):
    f, o = both(rd)
    b = np.argwhere(~(f == o).all(axis=-1))
    a_, b_ = po.film_to_rgb(f), po.film_to_rgb(o)
    rel = (np.abs(a_ - b_)[..., :3] / np.maximum(np.abs(b_[..., :3]), 1e-3)).max(axis=-1)
    big = np.argwhere(rel > 1e-4)
    print(tag, "differing:", len(b), "over tol:", [tuple(int(v) for v in q) for q in big[:5]], "worst", float(rel.max()))
    return f, o
report("threshold 0, 3 passes", dict(rd, AA_threshold=0.0))
crop = dict(rd, AA_threshold=0.0, xstart=px, ystart=py, width=1, height=1)
f, o = report("crop 1 px, threshold 0", crop)
print(f, o)
for v in [dict(path_samples=1), dict(bounces=1), dict(no_recursive=False), dict(raydepth=0), dict(transpShad=False), dict(AA_passes=2), dict(AA_inc_samples=1), dict(AA_minsamples=1)]:
    f, o = report(str(v), dict(crop, **v))
    print("   ", f.reshape(-1), o.reshape(-1))
