"""GPU-box helper: re-run one seed of tests/test_gpu_parity.py::test_random_feature_mixes and show where GPU and oracle differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libyafaray_amd import Interface, interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_parity import _feature_mix

seed = int(sys.argv[1])
sc, rd, w, h, base, kw = _feature_mix(seed)
print([m["type"] for m in sc["materials"]], kw, "spp", rd["AA_minsamples"], "lights", len(sc["lights"]))

def both(rd):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film = yi.getFilm(rd["width"], rd["height"])
    osc = po.OracleScene(sc); osc.set_tree(*interface.build_kdtree(sc["verts"], threads=4)[:3])
    ofilm, ost = osc.render(rd)
    return film, ofilm

film, ofilm = both(rd)
bad = np.argwhere(~(film == ofilm).all(axis=-1))
print("differing pixels:", len(bad))
for y, x in bad[:10]:
    print(y, x, film[y, x], ofilm[y, x])
# single pass variants to localise
for variant in [dict(AA_passes=1), dict(AA_passes=1, path_samples=1), dict(AA_passes=1, raydepth=0), dict(AA_passes=1, no_recursive=False), dict(AA_passes=1, transpShad=False), dict(AA_passes=1, bounces=1)]:
    r2 = dict(rd); r2.update(variant)
    f, o = both(r2)
    b = np.argwhere(~(f == o).all(axis=-1))
    worst = 0.0
    if len(b):
        a_, b_ = po.film_to_rgb(f), po.film_to_rgb(o)
        worst = float((np.abs(a_ - b_)[..., :3] / np.maximum(np.abs(b_[..., :3]), 1e-3)).max())
    print(variant, "differing:", len(b), "worst rel", worst, [tuple(v) for v in b[:4]])
