"""GPU-box probe: what ray ORDER is worth to the traversal.  Random rays in the metric scene (origins uniform in the box, directions uniform)
through yafaray_intersectRays / shadowRays, once in random order, once sorted by the Morton code of their origin cell at several grid sizes.
Run under `rocprofv3 --kernel-trace --stats`: the trace_kernel launches appear in call order (one closest-hit + one any-hit per ordering)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench                                         # noqa: E402
from libyafaray_amd import Interface, scenes        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
w, sc, rd = bench.make_workload("m1", res=64, spp=1)
yi = Interface()
scenes.load_scene(yi, sc, rd)
yi.prepareRender()
rng = np.random.default_rng(7)
o = rng.uniform(-0.98, 0.98, size=(n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d.astype(np.float32), np.full((n, 1), 5e-5, np.float32), np.full((n, 1), -1.0, np.float32)], axis=1)


def morton(g):
    c = np.clip(((o + 1.0) * 0.5 * g).astype(np.int64), 0, g - 1)
    key = np.zeros(n, dtype=np.int64)
    for b in range(int(np.log2(g))):
        for k in range(3):
            key |= ((c[:, k] >> b) & 1) << (3 * b + k)
    return key


orders = [("random", np.arange(n))]
for g in (4, 8, 16, 32):
    orders.append((f"morton{g}", np.argsort(morton(g), kind="stable")))
oct_key = morton(16) * 8 + ((d[:, 0] < 0) * 1 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0) * 4)
orders.append(("morton16+octant", np.argsort(oct_key, kind="stable")))
ref = None
for name, order in orders:
    r = np.ascontiguousarray(rays[order])
    t0 = time.time()
    tri, t, bary = yi.intersectRays(r)
    t1 = time.time()
    sh = yi.shadowRays(r)
    t2 = time.time()
    back = np.empty(n, dtype=np.int64); back[order] = np.arange(n)
    if ref is None:
        ref = (tri.copy(), t.copy(), sh.copy())
    else:
        assert np.array_equal(tri[back], ref[0]) and np.array_equal(t[back], ref[1]) and np.array_equal(sh[back], ref[2])
    print(f"{name:16s} closest call {t1 - t0:.3f} s, any-hit call {t2 - t1:.3f} s (host copies included; kernel times: rocprofv3 trace)", flush=True)
