"""GPU-box helper: for fuzz seeds, drop one render option at a time and report how much of the frame matches the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libyafaray_amd import Interface, interface, scenes
from oracle import pyoracle as po
from tests.test_gpu_parity import _feature_mix
def frac(sc, rd, w, h):
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    f = yi.getFilm(w, h); st = yi.getRenderStats()
    osc = po.OracleScene(sc); osc.set_tree(*interface.build_kdtree(sc["verts"], threads=4)[:3])
    o, ost = osc.render(rd)
    a, b = po.film_to_rgb(f), po.film_to_rgb(o)
    rel = np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(b[..., :3]), 1e-3)
    ok = (rel.max(axis=-1) <= 1e-4) & np.isclose(f[..., 4], o[..., 4], rtol=2e-6)
    return float(ok.mean()), (st.rays_closest, ost.rays_closest, st.rays_shadow, ost.rays_shadow, st.camera_samples, ost.camera_samples)
for seed in [int(x) for x in sys.argv[1:]]:
    sc, rd, w, h, base, kw = _feature_mix(seed)
    full = frac(sc, rd, w, h)
    print("seed", seed, "full:", full, [m for m in sc["materials"][base:]], flush=True)
    defaults = scenes.render_settings(w, h, rd["AA_minsamples"])
    for k in list(kw):
        if k in ("bounces", "raydepth", "path_samples", "integrator", "background", "bg_transp", "bg_transp_refract", "shadowDepth", "no_recursive", "AA_inc_samples", "AA_threshold"): continue
        r2 = dict(rd); r2.pop(k)
        if k in defaults: r2[k] = defaults[k]
        print("   without", k, "=", kw[k], ":", frac(sc, r2, w, h), flush=True)
    for i, m in enumerate(sc["materials"][base:]):
        for k in ("visibility", "receive_shadows", "flat_material"):
            if k in m:
                sc2 = dict(sc); sc2["materials"] = [dict(x) for x in sc["materials"]]; sc2["materials"][base + i].pop(k)
                print("   without material", i, k, "=", m[k], ":", frac(sc2, rd, w, h), flush=True)
    for i, m in enumerate(sc["materials"][base:]):
        for k, v in m.items():
            if k == "type" or isinstance(v, (bool, str)): continue
            sc2 = dict(sc); sc2["materials"] = [dict(x) for x in sc["materials"]]
            sc2["materials"][base + i][k] = (0.5, 0.6, 0.7) if isinstance(v, tuple) else (1.5 if k == "IOR" else (50.0 if k == "exponent" else 0.5))
            fr = frac(sc2, rd, w, h)
            if fr[0] > full[0] + 0.01: print("   material", i, m["type"], k, "=", v, "-> mid value:", fr, flush=True)
    for i, l in enumerate(sc["lights"]):
        for k in ("samples", "cast_shadows"):
            if k in l:
                sc2 = dict(sc); sc2["lights"] = [dict(x) for x in sc["lights"]]; sc2["lights"][i].pop(k)
                print("   without light", i, k, "=", l[k], ":", frac(sc2, rd, w, h), flush=True)
