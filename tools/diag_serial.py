"""GPU-box diagnostic: one serial-state feature mix, device against the single-threaded oracle (both trees), where they part"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YAFGPU_PIPELINE"] = "wavefront"
from tests import test_gpu_parity as T

seed = int(sys.argv[1])
over = dict(a.split("=") for a in sys.argv[2:])
sc, rd, w, h, base, kw = T._feature_mix(seed, serial=True)
for k, v in over.items():
    rd[k] = type(rd.get(k, 0))(v) if k in rd else int(v)
print("lights", [l["type"] for l in sc["lights"]], {k: rd[k] for k in ("bounces", "raydepth", "path_samples", "russian_roulette_min_bounces", "tile_size") if k in rd})
print("materials", [(m["type"], {k: v for k, v in m.items() if k in ("as_diffuse", "specular_reflect", "transparency", "fresnel_effect")}) for m in sc["materials"]])
for same_tree in (True, False):
    film, st, ofilm, ost = T._render_with_rand_state(sc, rd, same_tree=same_tree)
    d = ~np.isclose(film, ofilm, rtol=1e-4, atol=1e-6).all(axis=-1)
    print("same_tree", same_tree, "counts", (st.camera_samples, st.rays_closest, st.rays_shadow), (ost.camera_samples, ost.rays_closest, ost.rays_shadow), "pixels differing", int(d.sum()), "of", d.size)
    ys, xs = np.nonzero(d)
    if len(ys):
        ts = rd.get("tile_size", 32)
        order = np.lexsort((xs, ys, xs // ts, ys // ts))
        print("  first differing pixels in tile order:", [(int(xs[i]), int(ys[i])) for i in order[:6]])
