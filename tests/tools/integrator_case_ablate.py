"""Debugging aid (GPU box): one case of the integrator fixture, device against oracle, under ablations that switch single
features off — to localise what the two disagree on.  usage: python tests/tools/integrator_case_ablate.py <case name>"""
import copy
import sys

import numpy as np

sys.path.insert(0, ".")
from libyafaray_amd import Interface, scenes          # noqa: E402
from oracle import pyoracle as po                     # noqa: E402
from tests.integrator_fixture import case_scene, load      # noqa: E402

doc = load("ieee")
name = sys.argv[1]
cs = next(c for c in doc["cases"] if c["name"] == name)


def run(label, mut):
    sc, rd = case_scene(doc, cs)
    sc = copy.deepcopy(sc); rd = dict(rd)
    mut(sc, rd)
    osc = po.OracleScene(sc)
    fo, so = osc.render(dict(rd, oracle_threads=1))
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.setRandState(cs["srand"], 0)
    yi.render()
    f = yi.getFilm(rd["width"], rd["height"])
    st = yi.getRenderStats()
    rel = np.abs(f[..., :3] - fo[..., :3]) / np.maximum(np.abs(fo[..., :3]), 1e-3)
    print(f"{label:34s} rays dev {st.rays_closest}/{st.rays_shadow} ora {so.rays_closest}/{so.rays_shadow}  pixels differing {(f != fo).any(axis=-1).sum():4d}  max rel {rel.max():.3g}")


def plain(i):
    def m(sc, rd):
        sc["materials"][i] = {"type": "shinydiffusemat", "color": [0.75, 0.75, 0.75], "diffuse_reflect": 1.0}
    return m


run("as is", lambda sc, rd: None)
for i, m in enumerate(doc["materials"]):
    if i in set(cs["tri_mat"]) and m != doc["materials"][0] and m["type"] != "light_mat":
        run(f"material {i} ({m['type']}) -> white", plain(i))
run("path_samples 1", lambda sc, rd: rd.update(path_samples=1))
run("bounces 2", lambda sc, rd: rd.update(bounces=2, russian_roulette_min_bounces=max(2, rd.get("russian_roulette_min_bounces", 0))))
run("every light 1 sample", lambda sc, rd: [l.update(samples=1) for l in sc["lights"] if l["type"] == "arealight"])
run("first light only", lambda sc, rd: sc.update(lights=sc["lights"][:1]))
run("background black", lambda sc, rd: rd.update(background=[0.0, 0.0, 0.0]))
run("tile 32", lambda sc, rd: rd.update(tile_size=32))
run("no roulette", lambda sc, rd: rd.update(russian_roulette_min_bounces=rd.get("bounces", 3)))
run("1 spp", lambda sc, rd: rd.update(AA_minsamples=1, AA_passes=1))
