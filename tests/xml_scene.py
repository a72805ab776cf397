"""Independent (Python, xml.etree) reading of a YafaRay scene XML into the scene/render dicts the
oracle binding takes — used to cross-check the product's C++ XML loader (yafaray_loadXml)."""
import xml.etree.ElementTree as ET

import numpy as np


def _param(el):
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
        return (float(a.get("r", 0)), float(a.get("g", 0)), float(a.get("b", 0)))
    return None


def _params(el):
    return {c.tag: _param(c) for c in el if c.tag != "list_element"}


def load(path):
    root = ET.parse(path).getroot()
    assert root.tag == "scene"
    mats, mat_index, lights = [], {}, []
    camera = background = None
    integrators, meshes = {}, []
    render = {}
    for el in root:
        if el.tag == "material":
            mat_index[el.attrib["name"]] = len(mats)
            mats.append(_params(el))
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
            pts, cur, tris, tmat = [], None, [], []
            for c in el:
                if c.tag == "p":
                    pts.append((float(c.attrib["x"]), float(c.attrib["y"]), float(c.attrib["z"])))
                elif c.tag == "set_material":
                    cur = mat_index[c.attrib["sval"]]
                elif c.tag == "f":
                    tris.append((int(c.attrib["a"]), int(c.attrib["b"]), int(c.attrib["c"])))
                    tmat.append(cur)
            meshes.append((int(el.attrib.get("id", len(meshes) + 1)), np.array(pts, np.float32), np.array(tris, np.int64), np.array(tmat, np.int32)))
        elif el.tag == "render":
            render = _params(el)
    meshes.sort(key=lambda m: m[0])          # Scene::update walks its std::map in object-id order (scene.cc:797)
    verts = np.concatenate([m[1][m[2]] for m in meshes], axis=0).astype(np.float32)
    tri_mat = np.concatenate([m[3] for m in meshes]).astype(np.int32)
    scene = {"verts": verts, "tri_mat": tri_mat, "vnormals": None, "materials": mats, "lights": lights, "camera": camera,
             "material_names": sorted(mat_index, key=mat_index.get)}
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
