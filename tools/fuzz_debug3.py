import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libyafaray_amd import Interface, interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_parity import _feature_mix
seed = int(sys.argv[1])
sc, rd, w, h, base, kw = _feature_mix(seed)
print([m for m in sc["materials"][base:]]); print(kw, "spp", rd["AA_minsamples"], "cam", sc["camera"])
tree = interface.build_kdtree(sc["verts"], threads=4)[:3]
def both(rd):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film = yi.getFilm(rd["width"], rd["height"]); st = yi.getRenderStats()
    osc = po.OracleScene(sc); osc.set_tree(*tree)
    ofilm, ost = osc.render(rd)
    return film, ofilm, st, ost
def report(tag, rd):
    f, o, st, ost = both(rd)
    b = np.argwhere(~(f == o).all(axis=-1))
    print(tag, "rays", st.rays_closest, ost.rays_closest, st.rays_shadow, ost.rays_shadow, "differing px:", len(b))
    for y, x in b[:6]:
        print("   ", y, x, f[y, x], o[y, x])
    return b
b = report("full", rd)
one = dict(rd, AA_passes=1)
b1 = report("one pass", one)
for v in [dict(raydepth=0), dict(bounces=1), dict(transpShad=False), dict(bg_transp=False, bg_transp_refract=False)]:
    report(str(v), dict(one, **v))
if len(b1):
    y, x = b1[0]
    for spp in range(1, rd["AA_minsamples"] + 1):
        report(f"crop {x},{y} spp {spp}", dict(one, xstart=int(x), ystart=int(y), width=1, height=1, AA_minsamples=spp))
