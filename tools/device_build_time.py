import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libyafaray_amd import scenes, interface
sc = scenes.cornell_soup(1000000, seed=1, sigma=0.01)
interface.build_kdtree(sc["verts"][:3000], device=True)
os.environ["YAFGPU_BUILD_VERBOSE"] = "1"
interface.build_kdtree(sc["verts"], device=True)
