"""diagnostic (GPU): feature mix N against the oracle, with features switched off one at a time"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YAFGPU_PIPELINE"] = "wavefront"
import numpy as np
from libyafaray_amd import Interface, scenes, interface
from oracle import pyoracle as po
import tests.test_gpu_parity as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 13
if seed % 2:
    os.environ["YAFGPU_WF_CHUNK"] = str([700, 1500, 4000][seed % 3])

def run(sc, rd, w, h, what):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film, st = yi.getFilm(w, h), yi.getRenderStats()
    osc = po.OracleScene(sc)
    osc.set_tree(*interface.build_kdtree(sc["verts"], threads=4)[:3])
    ofilm, ost = osc.render(rd)
    a, b = po.film_to_rgb(film), po.film_to_rgb(ofilm)
    rel = (np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), 1e-3)).max(axis=-1)
    bad = (rel > 1e-4) | ~np.isclose(film[..., 4], ofilm[..., 4], rtol=2e-6)
    print(f"{what}: rays {st.rays_closest}/{ost.rays_closest} {st.rays_shadow}/{ost.rays_shadow} samples {st.camera_samples}/{ost.camera_samples}; bad pixels {int(bad.sum())}; max rel {rel.max():.3g}", flush=True)
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:3]:
        print("   ", y, x, film[y, x], ofilm[y, x])

sc, rd, w, h, base, kw = T._feature_mix(seed)
print([ {k: v for k, v in m.items() if k in ("type", "as_diffuse", "anisotropic", "visibility", "receive_shadows")} for m in sc["materials"][base:]])
print(rd)
run(sc, rd, w, h, "as is")
rd1 = {k: v for k, v in rd.items() if not k.startswith("AA_") or k in ("AA_minsamples", "AA_pixelwidth")}
rd1["AA_passes"] = 1
run(sc, rd1, w, h, "one pass")
sc2 = copy.deepcopy(sc)
for m in sc2["materials"]:
    m.pop("anisotropic", None)
run(sc2, rd1, w, h, "one pass, no anisotropic")
sc3 = copy.deepcopy(sc2)
for m in sc3["materials"]:
    if "as_diffuse" in m: m["as_diffuse"] = True
run(sc3, rd1, w, h, "one pass, no anisotropic, as_diffuse")
run(sc2, dict(rd1, transpShad=False), w, h, "one pass, no aniso, no transpShad")
run(sc2, dict(rd1, bg_transp=False), w, h, "one pass, no aniso, no bg_transp")
run(sc2, dict(rd1, raydepth=1), w, h, "one pass, no aniso, raydepth 1")
