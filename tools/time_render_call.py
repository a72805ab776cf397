"""GPU-box helper: yafaray_render (host film) against the resident pass it wraps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libyafaray_amd import Interface, scenes
W = H = 1024
sc = scenes.cornell_soup(100000, seed=1234, res=(W, H))
rd = scenes.render_settings(W, H, 16, bounces=2)
yi = Interface(); scenes.load_scene(yi, sc, rd); yi.prepareRender()
dev = torch.device("cuda", 0)
planes = torch.zeros((4, H, W, 5), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    yi.renderPassDevice(planes.data_ptr(), 0, stream); torch.cuda.synchronize()
    print("renderPassDevice %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for k in range(3):
    t0 = time.perf_counter(); yi.render(); dt = time.perf_counter() - t0
    print("render() %.1f ms, render_seconds %.1f ms" % (dt * 1e3, yi.getRenderStats().render_seconds * 1e3), flush=True)
t0 = time.perf_counter(); f = yi.getFilm(W, H); print("getFilm %.1f ms" % ((time.perf_counter() - t0) * 1e3))
