"""SURVEY row N2 — image textures and shader nodes of the ORACLE against the reference's own sources compiled here
(oracle/ref_harness/ref_textures.cc; fixtures tests/golden/ref_textures_{ieee,fast}.json.gz made by
tests/golden/make_golden.py textures): ImageTexture::getColor / getFloat over every clip mode, repeat / mirror / crop /
rot90, none and bilinear interpolation, 10-bit "optimized" buffers, adjustments; TextureMapperNode over every texco and
mapping, ValueNode, MixNode in every mode, LayerNode in every blend mode and flag set, evaluated as one graph."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


def golden(variant):
    with gzip.open(os.path.join(HERE, "golden", f"ref_textures_{variant}.json.gz"), "rt") as f:
        return json.load(f)


def f32(a):
    return np.array(a, dtype=np.uint32).view(np.float32)


IMAGE_CASES = {   # the harness's TexCase table (ref_textures.cc sec_image), restated as the texture parameters
    "img_repeat_bilinear": dict(w=7, h=5, interpolate="bilinear", clipping="repeat"),
    "img_repeat_none": dict(w=7, h=5, interpolate="none", clipping="repeat", xrepeat=3, yrepeat=2),
    "img_repeat_mirror": dict(w=6, h=6, interpolate="bilinear", clipping="repeat", xrepeat=2, yrepeat=3, mirror_x=True, mirror_y=True, color_space="LinearRGB"),
    "img_extend_crop_rot": dict(w=8, h=4, interpolate="bilinear", clipping="extend", rot90=True, cropmin_x=0.1, cropmin_y=0.2, cropmax_x=0.9, cropmax_y=0.7),
    "img_clip": dict(w=5, h=5, interpolate="bilinear", clipping="clip"),
    "img_clipcube": dict(w=5, h=5, interpolate="none", clipping="clipcube"),
    "img_checker": dict(w=4, h=4, interpolate="bilinear", clipping="checker", even_tiles=True, odd_tiles=False, checker_dist=0.3),
    "img_adjust": dict(w=7, h=5, interpolate="bilinear", clipping="repeat", adj_intensity=1.2, adj_contrast=0.8, adj_mult_factor_red=0.9,
                       adj_mult_factor_green=1.1, adj_mult_factor_blue=0.7, adj_clamp=True),
    "img_adjust_hsv": dict(w=7, h=5, interpolate="bilinear", clipping="repeat", adj_saturation=1.4, adj_hue=40.0, color_space="Raw_Manual_Gamma", gamma=2.2),
}


@pytest.mark.parametrize("name", sorted(IMAGE_CASES))
def test_image_texture_lookups(name):
    g = golden("ieee")
    c = dict(IMAGE_CASES[name])
    c.setdefault("color_space", "sRGB")               # the harness builds its ImageTexture with Srgb unless the case says otherwise
    w, h = c.pop("w"), c.pop("h")
    texels = f32(g[name + "_texels"]).reshape(h, w, 4)
    d = po.texture_desc(dict(c, texels=texels))
    pts = f32(g[name + "_in"]).reshape(-1, 3)
    want = f32(g[name + "_out"]).reshape(-1, 5)
    got = np.zeros_like(want)
    L = po.lib()
    for i in range(len(pts)):
        L.yor_texture_probe(C.byref(d), po.fptr(np.ascontiguousarray(pts[i])), po.fptr(got[i]))
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    assert len(bad) == 0, f"{name}: {len(bad)} of {len(pts)} lookups differ; first {pts[bad[0]]} -> {got[bad[0]]} vs {want[bad[0]]}"
    # the release build (-O3 -ffast-math) of the same reference sources agrees to rounding
    fast = f32(golden("fast")[name + "_out"]).reshape(-1, 5)
    np.testing.assert_allclose(got, fast, rtol=2e-5, atol=2e-6)


def _node_graph(g):
    """rebuild the harness's graph (ref_textures.cc sec_nodes) from its `nodes_desc` words"""
    d = np.array(g["nodes_desc"], dtype=np.uint32)
    fl = lambda k: float(d[k:k + 1].view(np.float32)[0])
    sint = lambda v: int(np.int32(np.uint32(v)))
    nodes, i = [], 0
    while i < len(d):
        t = int(d[i])
        if t == 0:
            texco = ["uv", "global", "orco", "transformed", "normal", "reflect", "window"][int(d[i + 1])]
            mapping = ["plain", "cube", "tube", "sphere"][int(d[i + 2])]
            nodes.append({"name": f"n{len(nodes)}", "type": "texture_mapper", "texture": "t", "texco": texco, "mapping": mapping,
                          "proj_x": int(d[i + 3]), "proj_y": int(d[i + 4]), "proj_z": int(d[i + 5]),
                          "scale": tuple(fl(i + 6 + k) for k in range(3)), "offset": tuple(fl(i + 9 + k) for k in range(3)),
                          "transform": np.array([fl(i + 12 + k) for k in range(16)], np.float32).reshape(4, 4), "do_scalar": bool(d[i + 28])})
            i += 29
        elif t == 1:
            nodes.append({"name": f"n{len(nodes)}", "type": "value", "color": (fl(i + 1), fl(i + 2), fl(i + 3)), "alpha": fl(i + 4), "scalar": fl(i + 5)})
            i += 6
        elif t == 2:
            n = {"name": f"n{len(nodes)}", "type": "mix", "mode": int(d[i + 1]), "value": fl(i + 2), "input1": f"n{int(d[i + 3])}",
                 "color2": (fl(i + 6), fl(i + 7), fl(i + 8), fl(i + 9))}
            if sint(d[i + 4]) >= 0:
                n["input2"] = f"n{sint(d[i + 4])}"
            if sint(d[i + 5]) >= 0:
                n["factor"] = f"n{sint(d[i + 5])}"
            nodes.append(n)
            i += 10
        else:
            n = {"name": f"n{len(nodes)}", "type": "layer", "mode": int(d[i + 1]), "input": f"n{int(d[i + 2])}", "def_col": (0.9, 0.4, 0.2), "colfac": 0.8,
                 "def_val": 0.7, "valfac": 0.9, "noRGB": bool(d[i + 4]), "stencil": bool(d[i + 5]), "negative": bool(d[i + 6]), "use_alpha": bool(d[i + 7]),
                 "do_color": bool(d[i + 8]), "do_scalar": bool(d[i + 9]), "color_input": bool(d[i + 10])}
            if sint(d[i + 3]) >= 0:
                n["upper_layer"] = f"n{sint(d[i + 3])}"
            else:
                n.update({"upper_color": (0.25, 0.5, 0.75, 1.0), "upper_value": 0.45})
            nodes.append(n)
            i += 11
    return nodes


def test_shader_node_graph():
    """112 nodes — 24 texture mappers (texco x mapping), a value node, 20 mix nodes (modes 0..9, constant and node-driven
    factor), 54 layers (9 blend modes x 6 flag sets, stacked through upper_layer) — on 40 surface points: every node's
    colour, alpha and scalar bit for bit.  tube / sphere mappings go through libm's atan2 / acos in double on both sides."""
    g = golden("ieee")
    nodes = _node_graph(g)
    tex = dict(name="t", texels=f32(g["nodes_texels"]).reshape(5, 6, 4), interpolate="bilinear", clipping="repeat", color_space="sRGB")
    arr, index = po.node_descs(nodes, {"t": 0})
    assert [index[n["name"]] for n in nodes] == list(range(len(nodes)))          # the harness built them in evaluation order
    td = po.texture_desc(tex)
    c = f32(g["nodes_camera"])
    cam = po.camera_desc({"from": c[0:3], "to": c[3:6], "up": c[6:9], "resx": int(g["nodes_camera"][9]), "resy": int(g["nodes_camera"][10]), "focal": float(c[11])})
    sps = f32(g["nodes_in"]).reshape(-1, 18)
    want = f32(g["nodes_out"]).reshape(len(sps), len(nodes), 5)
    L = po.lib()
    got = np.zeros_like(want)
    for k in range(len(sps)):
        L.yor_nodes_probe(len(nodes), arr, 1, C.byref(td), C.byref(cam), po.fptr(np.ascontiguousarray(sps[k])), po.fptr(got[k]))
    diff = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
    bad_nodes = sorted(set(np.nonzero(diff)[1].tolist()))
    msg = ""
    if bad_nodes:
        k = int(np.nonzero(diff[:, bad_nodes[0]])[0][0])
        msg = f"{len(bad_nodes)} nodes differ, first: {nodes[bad_nodes[0]]} at point {k}: {got[k, bad_nodes[0]]} vs {want[k, bad_nodes[0]]}"
    assert not bad_nodes, msg


def test_bump_derivatives_and_apply_bump():
    """evalDerivative of every node of the same graph (TextureMapperNode's UV branch on surface points with UVs, its other
    branch without them and for every other coordinate kind; LayerNode; the base class's zero for value / mix) and
    Material::applyBump with the last layer's derivative: bit for bit."""
    g = golden("ieee")
    nodes = _node_graph(g)
    tex = dict(name="t", texels=f32(g["nodes_texels"]).reshape(5, 6, 4), interpolate="bilinear", clipping="repeat", color_space="sRGB")
    arr, index = po.node_descs(nodes, {"t": 0})
    td = po.texture_desc(tex)
    c = f32(g["nodes_camera"])
    cam = po.camera_desc({"from": c[0:3], "to": c[3:6], "up": c[6:9], "resx": int(g["nodes_camera"][9]), "resy": int(g["nodes_camera"][10]), "focal": float(c[11])})
    sps = f32(g["bump_in"]).reshape(-1, 31)
    want = f32(g["bump_out"]).reshape(len(sps), len(nodes), 5)
    want9 = f32(g["bump_applied"]).reshape(len(sps), 9)
    L = po.lib()
    L.yor_nodes_probe_derivative.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.yor_nodes_probe_derivative.restype = None
    got = np.zeros_like(want); got9 = np.zeros_like(want9)
    for k in range(len(sps)):
        L.yor_nodes_probe_derivative(len(nodes), C.cast(arr, C.c_void_p), 1, C.cast(C.pointer(td), C.c_void_p), C.cast(C.pointer(cam), C.c_void_p),
                                     po.fptr(np.ascontiguousarray(sps[k])), 40.0, po.fptr(got[k]), po.fptr(got9[k]))
    diff = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
    bad = sorted(set(np.nonzero(diff)[1].tolist()))
    msg = ""
    if bad:
        k = int(np.nonzero(diff[:, bad[0]])[0][0])
        msg = f"{len(bad)} nodes differ, first: {nodes[bad[0]]} at point {k} (has_uv {sps[k, 30]}): {got[k, bad[0]]} vs {want[k, bad[0]]}"
    assert not bad, msg
    assert np.array_equal(got9.view(np.uint32), want9.view(np.uint32)), "applyBump differs"
    # the same texture flagged as a normal map: evalDerivative's other two branches, setup() without the / 100
    tdn = po.texture_desc(dict(tex, normalmap=True))
    wantn = f32(g["bump_normalmap_out"]).reshape(len(sps), len(nodes), 5)
    gotn = np.zeros_like(wantn)
    for k in range(len(sps)):
        L.yor_nodes_probe_derivative(len(nodes), C.cast(arr, C.c_void_p), 1, C.cast(C.pointer(tdn), C.c_void_p), C.cast(C.pointer(cam), C.c_void_p),
                                     po.fptr(np.ascontiguousarray(sps[k])), 1.0, po.fptr(gotn[k]), None)
    diff = (gotn.view(np.uint32) != wantn.view(np.uint32)).any(axis=2)
    bad = sorted(set(np.nonzero(diff)[1].tolist()))
    assert not bad, f"normal map: {len(bad)} nodes differ, first {nodes[bad[0]]}: {gotn[np.nonzero(diff[:, bad[0]])[0][0], bad[0]]} vs {wantn[np.nonzero(diff[:, bad[0]])[0][0], bad[0]]}"
