"""GPU-box helper: what the detection step between adaptive passes costs (host) against the passes themselves"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libyafaray_amd import Interface, scenes
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenes.cornell_soup(100000, seed=1234, res=(W, H))
for passes, thr in ((1, 0.02), (4, 0.02), (4, 0.0)):
    rd = scenes.render_settings(W, H, 16, bounces=2, AA_passes=passes, AA_inc_samples=8, AA_threshold=thr, AA_variance_pixels=0)
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.prepareRender()
    yi.render()
    t0 = time.perf_counter(); yi.render(); dt = time.perf_counter() - t0
    st = yi.getRenderStats()
    print(f"{W}x{H} passes {passes} threshold {thr}: {dt * 1e3:.1f} ms, camera samples {st.camera_samples} ({st.camera_samples / (W * H):.1f} per pixel)", flush=True)
    yi.close()
