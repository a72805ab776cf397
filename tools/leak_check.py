"""GPU-box helper: device and host memory across many scene set-ups and render passes (nothing may grow)."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libyafaray_amd import Interface, scenes, interface as yi_mod
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 1e6
def rss_mb(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
print("start: free %.0f MB, rss %.0f MB" % (free_mb(), rss_mb()), flush=True)
for rnd in range(6):
    sc = scenes.cornell_soup(100000, seed=1234, res=(256, 256))
    rd = scenes.render_settings(256, 256, 16, bounces=2, raydepth=2)
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.prepareRender()
    planes = torch.zeros((4, 256, 256, 5), dtype=torch.float32, device=dev)
    for k in range(40):
        yi.renderPassDevice(planes.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    yi.render()
    yi.close(); del planes
    print("round %d: free %.0f MB, rss %.0f MB" % (rnd, free_mb(), rss_mb()), flush=True)
# round 2: textured scenes, the serial-state replay (two lights, roulette on) and glossy-recursive frames
import tests.test_gpu_parity as T
for rnd in range(6):
    for seed in (5, 9, 13, 21, 40, 44, 61):
        sc, rd, w, h, base, kw = T._feature_mix(seed)
        yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render(); yi.close()
    sc = scenes.cornell_soup(20000, seed=7, res=(128, 128), n_lights=2, glossy_fraction=0.3)
    rd = scenes.render_settings(128, 128, 8, bounces=4, russian_roulette_min_bounces=1)
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render(); yi.close()
    print("round2 %d: free %.0f MB, rss %.0f MB" % (rnd, free_mb(), rss_mb()), flush=True)
