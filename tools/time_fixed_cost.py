"""GPU-box helper: the fixed cost of one resident pass (tiny frame): launches, host bookkeeping"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libyafaray_amd import Interface, scenes
sc = scenes.cornell_soup(100000, seed=1234, res=(64, 64))
rd = scenes.render_settings(64, 64, 1, bounces=2)
yi = Interface(); scenes.load_scene(yi, sc, rd); yi.prepareRender()
dev = torch.device("cuda", 0)
planes = torch.zeros((4, 64, 64, 5), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for k in range(5):
    yi.renderPassDevice(planes.data_ptr(), 0, stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(100):
    yi.renderPassDevice(planes.data_ptr(), 0, stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("64x64x1: host time per pass %.3f ms, wall per pass %.3f ms" % ((t1 - t0) * 10, (t2 - t0) * 10))
