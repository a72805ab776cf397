"""GPU-box helper: build the kd-tree on the device, check it against brute force through the oracle, compare with the host builder."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libyafaray_amd import scenes, interface
from oracle import pyoracle as po

for n_tris, sigma, seed in [(12, 0.02, 1), (300, 0.05, 2), (5000, 0.02, 3), (100000, 0.02, 1234), (1000000, 0.01, 1)]:
    sc = scenes.cornell_soup(n_tris, seed=seed, sigma=sigma)
    t0 = time.time(); hn, hr, hb, hi = interface.build_kdtree(sc["verts"], threads=0); th = time.time() - t0
    t0 = time.time(); dn, dr, db, di = interface.build_kdtree(sc["verts"], device=True); td = time.time() - t0
    t0 = time.time(); dn2, dr2, db2, di2 = interface.build_kdtree(sc["verts"], device=True); td2 = time.time() - t0
    assert np.array_equal(dn, dn2) and np.array_equal(dr, dr2), "device build is not deterministic"
    print(f"n={n_tris}: host nodes {hi.n_nodes} refs {hi.n_leaf_refs} depth {hi.max_depth} {hi.build_seconds:.3f}s (wall {th:.3f}) | "
          f"device nodes {di.n_nodes} refs {di.n_leaf_refs} depth {di.max_depth} {di.build_seconds:.3f}s (wall {td:.3f}, again {td2:.3f})", flush=True)
    assert np.array_equal(hb, db), (hb, db)
    # structure
    flags = dn[:, 1]; leafm = (flags & 3) == 3
    right = (flags[~leafm] >> 2).astype(np.int64)
    assert np.all(right < len(dn)) and np.all(right > np.nonzero(~leafm)[0] + 1)
    first = dn[leafm, 0].astype(np.int64); cnt = (flags[leafm] >> 2).astype(np.int64)
    assert np.all(first + cnt <= len(dr)) and np.unique(dr).size == n_tris
    if n_tris <= 100000:
        osc = po.OracleScene(sc); osc.set_tree(dn, dr, db)
        rng = np.random.default_rng(seed)
        n_rays = 1500
        o = rng.uniform(-0.98, 0.98, size=(n_rays, 3)).astype(np.float32)
        d = rng.normal(size=(n_rays, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        d[::11] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=d[::11].shape[0])]
        bad = 0
        for i in range(n_rays):
            a = osc.intersect(o[i], d[i], 0.0, -1.0, use_tree=True); b = osc.intersect(o[i], d[i], 0.0, -1.0, use_tree=False)
            if a[0] != b[0] or (a[0] and (a[1] != b[1] or a[2] != b[2])): bad += 1
        print("   rays vs brute force: mismatches", bad, flush=True)
        assert bad == 0
