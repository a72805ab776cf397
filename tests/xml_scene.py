"""Independent (Python, xml.etree) reading of a YafaRay scene XML into the scene/render dicts the
oracle binding takes — used to cross-check the product's C++ XML loader (yafaray_loadXml)."""
import xml.etree.ElementTree as ET

import numpy as np


def _param(el, rgba=False):
    a = el.attrib
    if len(a) == 1:
        k, v = next(iter(a.items()))
        if k == "ival":
            return int(v)
        if k == "fval":
            return float(v)
        if k == "bval":
            return v == "true"
        if k == "sval":
            return v
    if "x" in a or "y" in a or "z" in a:
        return (float(a.get("x", 0)), float(a.get("y", 0)), float(a.get("z", 0)))
    if "r" in a or "g" in a or "b" in a:
        if rgba and "a" in a:         # shader nodes read some colours as Rgba (upper_color, color1 / color2)
            return (float(a.get("r", 0)), float(a.get("g", 0)), float(a.get("b", 0)), float(a["a"]))
        return (float(a.get("r", 0)), float(a.get("g", 0)), float(a.get("b", 0)))
    if any(len(k) == 3 and k[0] == "m" for k in a):
        m = np.zeros((4, 4), np.float32)
        for k, v in a.items():
            m[int(k[1]), int(k[2])] = float(v)
        return m
    return None


def _params(el, rgba=False):
    return {c.tag: _param(c, rgba) for c in el if c.tag != "list_element"}


def load(path, texels=None):
    """texels: name -> (h, w, 4) float32 decoded image for the scene's <texture> elements (the oracle has no file decoders; the
    product's are pinned on their own by tests/test_image_decoders.py)"""
    root = ET.parse(path).getroot()
    assert root.tag == "scene"
    mats, mat_index, lights, textures = [], {}, [], []
    camera = background = None
    integrators, meshes = {}, []
    render = {}
    for el in root:
        if el.tag == "material":
            mat_index[el.attrib["name"]] = len(mats)
            m = _params(el)
            nodes = [_params(le, rgba=True) for le in el.findall("list_element")]
            nodes = [n for n in nodes if n.get("element", "shader_node") == "shader_node"]
            if nodes:
                for n in nodes:
                    n.pop("element", None)
                m["nodes"] = nodes
            mats.append(m)
        elif el.tag == "texture":
            t = dict(_params(el), name=el.attrib["name"])
            if texels is not None:
                t["texels"] = texels(t["name"])
                if t.get("filename", "").lower().endswith((".hdr", ".pic", ".exr")):
                    t["color_space"] = "LinearRGB"      # ImageTexture::factory forces it for HDR files (texture_image.cc:600-606)
            textures.append(t)
        elif el.tag == "light":
            p = _params(el)
            if p.get("light_enabled", True):
                lights.append(p)
        elif el.tag == "camera":
            camera = _params(el)
        elif el.tag == "background":
            background = _params(el)
        elif el.tag == "integrator":
            integrators[el.attrib["name"]] = _params(el)
        elif el.tag == "mesh":
            pts, cur, tris, tmat, orcos = [], None, [], [], []
            has_orco = el.attrib.get("has_orco") == "true"
            for c in el:
                if c.tag == "p":
                    pts.append((float(c.attrib["x"]), float(c.attrib["y"]), float(c.attrib["z"])))
                    orcos.append((float(c.attrib.get("ox", 0)), float(c.attrib.get("oy", 0)), float(c.attrib.get("oz", 0))) if has_orco else (np.nan, 0.0, 0.0))
                elif c.tag == "set_material":
                    cur = mat_index[c.attrib["sval"]]
                elif c.tag == "f":
                    tris.append((int(c.attrib["a"]), int(c.attrib["b"]), int(c.attrib["c"])))
                    tmat.append(cur)
            meshes.append((int(el.attrib.get("id", len(meshes) + 1)), np.array(pts, np.float32), np.array(tris, np.int64), np.array(tmat, np.int32), np.array(orcos, np.float32)))
        elif el.tag == "render":
            render = _params(el)
    meshes.sort(key=lambda m: m[0])          # Scene::update walks its std::map in object-id order (scene.cc:797)
    verts = np.concatenate([m[1][m[2]] for m in meshes], axis=0).astype(np.float32)
    tri_mat = np.concatenate([m[3] for m in meshes]).astype(np.int32)
    scene = {"verts": verts, "tri_mat": tri_mat, "vnormals": None, "materials": mats, "lights": lights, "camera": camera,
             "material_names": sorted(mat_index, key=mat_index.get)}
    if textures and any("nodes" in m for m in mats):
        scene["textures"] = textures
        # per triangle corner; a mesh without orco marks its triangles with a NaN first word (has_orco_ is per mesh, triangle.cc:46-63)
        orco = np.concatenate([m[4][m[2]] for m in meshes], axis=0).astype(np.float32)
        orco[np.isnan(orco[:, 0, 0]), 1:, :] = 0
        scene["orco"] = orco
    integ = integrators[render["integrator_name"]]
    rd = dict(render)
    rd["integrator"] = integ["type"]
    for k in ("path_samples", "bounces", "russian_roulette_min_bounces", "no_recursive", "bg_transp", "bg_transp_refract", "raydepth",
              "transpShad", "shadowDepth"):
        if k in integ:
            rd[k] = integ[k]
    if background is not None and "background_name" in render:
        rd["background"] = tuple(np.float32(background.get("power", 1.0)) * np.float32(c) for c in background.get("color", (0, 0, 0)))
    return scene, rd
