import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libyafaray_amd import Interface, scenes
from oracle import pyoracle as po
for (n_tris,res,spp,bounces) in [(12,32,4,2),(500,48,16,3),(5000,64,16,3)]:
    sc = scenes.cornell_soup(n_tris, seed=n_tris, res=(res,res))
    rd = scenes.render_settings(res,res,spp,bounces=bounces)
    yi = Interface(); scenes.load_scene(yi, sc, rd); yi.render()
    film = yi.getFilm(res,res); st = yi.getRenderStats()
    ofilm, ost = po.OracleScene(sc).render(rd)
    print(n_tris, "rays", st.rays_closest, ost.rays_closest, st.rays_shadow, ost.rays_shadow)
    a,b = po.film_to_rgb(film), po.film_to_rgb(ofilm)
    rel = np.abs(a[...,:3]-b[...,:3])/np.maximum(np.abs(b[...,:3]),1e-3)
    w = rel.max(axis=-1)
    print("  weights equal", np.array_equal(film[...,4],ofilm[...,4]), "bitexact px", (film==ofilm).all(axis=-1).mean(), "over1e-4", (w>1e-4).sum(), "over1e-6", (w>1e-6).sum(), "max", w.max())
    ys,xs = np.nonzero(w>1e-6)
    for y,x in list(zip(ys,xs))[:6]:
        print("   px",x,y,a[y,x,:3],b[y,x,:3])
