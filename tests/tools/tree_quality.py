"""Host-only: walk the product's kd-tree with the oracle's (reference) traversal and report per-ray work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libyafaray_amd import scenes, interface
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
sc = scenes.cornell_soup(n, seed=1234, sigma=sigma, res=(64, 64))
rd = scenes.render_settings(64, 64, 4, bounces=1, oracle_threads=8)
osc = po.OracleScene(sc)
_, st = osc.render(rd)
r = st.rays_closest + st.rays_shadow
print(f"oracle tree : nodes {st.kd_nodes} refs {st.kd_leaf_refs} | per ray: interior {st.interior_steps/r:.1f} leaves {st.leaves/r:.1f} tests {st.tri_tests/r:.1f} | render {st.render_seconds:.2f}s")
t = time.time()
nodes, refs, bound, info = interface.build_kdtree(sc["verts"])
print(f"product tree: nodes {info.n_nodes} refs {info.n_leaf_refs} depth {info.max_depth} build {info.build_seconds:.2f}s")
osc.set_tree(nodes, refs, bound)
film2, st2 = osc.render(rd)
r2 = st2.rays_closest + st2.rays_shadow
print(f"              per ray: interior {st2.interior_steps/r2:.1f} leaves {st2.leaves/r2:.1f} tests {st2.tri_tests/r2:.1f} | render {st2.render_seconds:.2f}s  rays {r2} vs {r}")
