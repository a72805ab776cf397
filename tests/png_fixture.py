"""The one whole-path fixture the reference's own tests hold: tests/test01's expected render (committed as
tests/golden/test01_expected.png by tests/golden/make_test01_pt.py --expected).  480x340 RGBA: a 70-row parameters badge
on top of the 480x270 render of test01.xml as shipped — directlighting, one point light, 1 spp, gauss filter 1.5, sRGB,
8 bit.  The six cubes are textured (shader nodes); every pixel whose filter footprint only sees the
untextured floor material is a pure function of camera, traversal, direct lighting, film filter and output transform,
i.e. of the whole path, and is compared here.

What stands between the float film and the PNG (restated from the reference):
  ImageFilm::flush, imagefilm.cc:737-772: Pixel::normalized -> clampRgb0 -> Rgb::colorSpaceFromLinearRgb(sRGB)
      (color.h:359-364: v <= 0.0031308 ? 12.92 v : 1.055 fPow(v, 0.416667) - 0.055, with the polynomial fPow__)
  PngHandler::saveToFile, imagehandler_png.cc:119-141: clampRgba01, then (unsigned char)(c * 255.f) — truncation.
The reference was built with OpenCV when it made the file or it was not (test01.xml asks for denoise, which only exists
with OpenCV, imagehandler_png.cc:106): the comparison tolerates 2 levels and reports how many pixels are exact.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_expected():
    from PIL import Image
    meta = json.load(open(os.path.join(HERE, "golden", "test01_expected.json")))
    img = np.array(Image.open(os.path.join(HERE, "golden", "test01_expected.png")))
    assert img.shape == (meta["height"] + meta["badge_rows_on_top"], meta["width"], 4)
    return img[meta["badge_rows_on_top"]:, :, :3].astype(np.int32), meta


def film_to_8bit(film):
    """float film [H][W][5] -> the bytes the reference's PNG output would hold."""
    from oracle import pyoracle as po
    L = po.lib()
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = np.where(w != 0, film[..., :3] * (np.float32(1.0) / w).astype(np.float32), np.float32(0)).astype(np.float32)   # color.h:310-314
    rgb = np.maximum(rgb, np.float32(0))                                                                                     # clampRgb0
    flat = rgb.reshape(-1)
    out = np.empty_like(flat)
    lin = flat <= np.float32(0.0031308)
    out[lin] = flat[lin] * np.float32(12.92)
    idx = np.nonzero(~lin)[0]
    vals = flat[idx]
    uniq, inv = np.unique(vals, return_inverse=True)             # the polynomial fPow__ through the oracle, once per distinct value
    pw = np.array([L.yor_fpow(C.c_float(float(v)), C.c_float(0.416667)) for v in uniq], dtype=np.float32)
    out[idx] = (np.float32(1.055) * pw[inv] - np.float32(0.055)).astype(np.float32)
    out = np.clip(out, np.float32(0), np.float32(1))                                                                         # clampRgba01
    return np.floor(out * np.float32(255.0)).astype(np.int32).reshape(rgb.shape)                                             # (YByte_t)(c * 255.f)


def primary_hit_materials(sc, rd):
    """material index of the triangle the camera ray through every pixel centre hits (-1: background), via the oracle's
    camera and brute-force-free kd traversal (checker side only)."""
    from oracle import pyoracle as po
    L = po.lib()
    W, H = rd["width"], rd["height"]
    osc = po.OracleScene(sc)
    cam = po.camera_desc(sc["camera"])
    out9 = np.zeros(9, np.float32)
    mats = np.full((H, W), -1, np.int32)
    tri_mat = np.asarray(sc["tri_mat"], np.int32)
    for y in range(H):
        for x in range(W):
            L.yor_camera_shoot(C.byref(cam), C.c_float(x + 0.5), C.c_float(y + 0.5), po.fptr(out9))
            hit, tri, t, _ = osc.intersect(out9[0:3], out9[3:6], float(out9[6]), float(out9[7]))
            if hit:
                mats[y, x] = tri_mat[tri]
    osc.close()
    return mats


def comparable_mask(sc, rd, meta, margin=3):
    """pixels whose gauss-1.5 footprint (half-width 1.5 px; `margin` px to be safe) lies on untextured materials or background"""
    names = sc["material_names"]
    plain = {i for i, n in enumerate(names) if n in meta["untextured_materials"]} | {-1}
    mats = primary_hit_materials(sc, rd)
    ok = np.isin(mats, list(plain))
    H, W = ok.shape
    grown = ok.copy()
    for dy in range(-margin, margin + 1):
        for dx in range(-margin, margin + 1):
            sh = np.zeros_like(ok)
            ys, ye = max(0, dy), min(H, H + dy)
            xs, xe = max(0, dx), min(W, W + dx)
            sh[ys:ye, xs:xe] = ok[ys - dy:ye - dy, xs - dx:xe - dx]
            # pixels shifted out of the frame count as comparable (nothing textured there)
            edge = np.ones_like(ok); edge[ys:ye, xs:xe] = False
            grown &= (sh | edge)
    return grown


def compare(film, sc, rd, what):
    ref, meta = load_expected()
    got = film_to_8bit(film)
    mask = comparable_mask(sc, rd, meta)
    d = np.abs(got - ref).max(axis=-1)[mask]
    n = int(mask.sum())
    stats = {"pixels_compared": n, "fraction_of_frame": n / mask.size, "exact": int((d == 0).sum()), "within_1": int((d <= 1).sum()),
             "within_2": int((d <= 2).sum()), "max_levels": int(d.max()) if n else 0}
    print(f"{what} vs the reference's expected PNG: {stats}")
    return stats
