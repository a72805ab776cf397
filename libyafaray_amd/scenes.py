"""Synthetic scenes of BASELINE.json's configs and a driver that feeds a scene
description through the Interface API exactly as an exporter would.

A scene description is a plain dict:
    verts      (N,3,3) float32   triangle corners a,b,c
    tri_mat    (N,)    int32     index into materials
    vnormals   None | (N,3,3)    per-corner shading normals
    materials  [ {type: ..., <factory parameters by their reference names>} ]
    lights     [ {type: ..., ...} ]
    camera     {from,to,up,resx,resy,focal,...}
and a render description is a dict of the reference's render/integrator parameter names.
"""
import numpy as np


def _quad(a, b, c, d):
    """two triangles (a,b,c), (a,c,d)"""
    return np.array([[a, b, c], [a, c, d]], dtype=np.float32)


def cornell_soup(n_tris, seed=1234, sigma=0.02, glossy_fraction=0.0, n_lights=1, light_power=15.0,
                 res=(512, 512), open_front=True):
    """Cornell-style box [-1,1]^3 open toward -y, filled with a soup of `n_tris` small random triangles
    (uniform centres, edge vectors ~ N(0, sigma^2)), one (or two) quad area lights under the ceiling on
    emissive geometry.  Materials: 0 white, 1 red, 2 green shinydiffuse; 3 light_mat; 4 glossy."""
    rng = np.random.default_rng(seed)
    quads = []
    mats = []
    # floor (z=-1, normal +z), ceiling (z=1, normal -z), back (y=1, normal -y), left (x=-1, +x), right (x=1, -x)
    quads.append(_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1))); mats += [0, 0]
    quads.append(_quad((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1))); mats += [0, 0]
    quads.append(_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1))); mats += [0, 0]
    quads.append(_quad((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1))); mats += [1, 1]
    quads.append(_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))); mats += [2, 2]
    if not open_front:
        quads.append(_quad((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1))); mats += [0, 0]
    lights = []
    light_rects = [(-0.25, -0.25, 0.25, 0.25)] if n_lights == 1 else [(-0.7, -0.25, -0.2, 0.25), (0.2, -0.25, 0.7, 0.25)][:n_lights]
    for (x0, y0, x1, y1) in light_rects:
        z = 0.99
        # emissive geometry facing down (-z); arealight: to_x = +y, to_y = +x so that fnormal = to_y x to_x = +z
        # (AreaLight::illumSample needs (p - sp)·fnormal > 0 for surfaces below, light_area.cc:79-81)
        quads.append(_quad((x0, y0, z), (x0, y1, z), (x1, y1, z), (x1, y0, z))); mats += [3, 3]
        lights.append({"type": "arealight", "corner": (x0, y0, z), "point1": (x0, y1, z), "point2": (x1, y0, z),
                       "color": (1.0, 1.0, 1.0), "power": light_power, "samples": 1})
    walls = np.concatenate(quads, axis=0)
    n_soup = max(0, n_tris - walls.shape[0])
    centres = rng.uniform(-0.95, 0.95, size=(n_soup, 1, 3)).astype(np.float32)
    centres[:, :, 2] = centres[:, :, 2] * 0.9 - 0.05   # keep the soup below the lights
    offs = rng.normal(0.0, sigma, size=(n_soup, 3, 3)).astype(np.float32)
    soup = (centres + offs).astype(np.float32)
    soup_mat = rng.integers(0, 3, size=n_soup).astype(np.int32)
    if glossy_fraction > 0:
        soup_mat = np.where(rng.random(n_soup) < glossy_fraction, 4, soup_mat).astype(np.int32)
    verts = np.concatenate([walls, soup], axis=0).astype(np.float32)
    tri_mat = np.concatenate([np.array(mats, dtype=np.int32), soup_mat])
    materials = [
        {"type": "shinydiffusemat", "color": (0.7, 0.7, 0.7), "diffuse_reflect": 1.0},
        {"type": "shinydiffusemat", "color": (0.7, 0.15, 0.15), "diffuse_reflect": 1.0},
        {"type": "shinydiffusemat", "color": (0.15, 0.7, 0.15), "diffuse_reflect": 1.0},
        {"type": "light_mat", "color": (1.0, 1.0, 1.0), "power": light_power},
        {"type": "glossy", "color": (0.9, 0.9, 0.9), "diffuse_color": (0.6, 0.6, 0.7), "diffuse_reflect": 0.4,
         "glossy_reflect": 0.6, "exponent": 50.0, "as_diffuse": True},
    ]
    camera = {"type": "perspective", "from": (0.0, -3.8, 0.0), "to": (0.0, 0.0, 0.0), "up": (0.0, -3.8, 1.0),
              "resx": res[0], "resy": res[1], "focal": 1.4}
    return {"verts": verts, "tri_mat": tri_mat, "vnormals": None, "materials": materials, "lights": lights, "camera": camera}


def render_settings(width, height, spp, bounces=2, path_samples=1, integrator="pathtracing", **kw):
    """The deterministic-parity regime of SURVEY §8c: one pass, box filter width 1, Russian roulette off."""
    r = {"integrator": integrator, "path_samples": path_samples, "bounces": bounces,
         "russian_roulette_min_bounces": bounces, "width": width, "height": height, "AA_passes": 1,
         "AA_minsamples": spp, "AA_pixelwidth": 1.0, "filter_type": "box", "tile_size": 32,
         "background": (0.0, 0.0, 0.0)}
    r.update(kw)
    return r


def _color(v):
    return ("color", float(v[0]), float(v[1]), float(v[2]), 1.0)


_COLOR_KEYS = {"color", "mirror_color", "diffuse_color", "filter_color", "absorption"}
_NODE_COLOR_KEYS = {"color", "color1", "color2", "def_col", "upper_color"}


def load_scene(yi, scene, render):
    """Drive the Interface the way the XML loader / an exporter does (loader_xml.cc:217-308):
    materials, lights, camera, background, integrators, geometry, then the render ParamMap."""
    yi.startScene(0)
    handles = []
    for i, t in enumerate(scene.get("textures") or []):
        # image textures: ImageTexture::factory's parameters, and either a file name or texels already in memory
        yi.paramsClearAll()
        params = {k: v for k, v in t.items() if k not in ("name", "texels")}
        params.setdefault("type", "image")
        yi.paramsSet(params)
        name = t.get("name", f"tex{i}")
        if "texels" in t:
            yi.createTextureFromMemory(name, t["texels"])
        else:
            yi.createTexture(name)
    for i, m in enumerate(scene["materials"]):
        yi.paramsClearAll()
        yi.paramsSet({k: (_color(v) if k in _COLOR_KEYS else v) for k, v in m.items() if k != "nodes"})
        for node in m.get("nodes") or []:
            # shader nodes: one <list_element> ParamMap each (import_xml.cc:683-690)
            yi.paramsPushList()
            yi.paramsSetString("element", "shader_node")
            for k, v in node.items():
                if k == "transform":
                    yi.paramsSetMatrix(k, np.asarray(v, dtype=np.float32).reshape(16))
                elif k in _NODE_COLOR_KEYS:
                    yi.paramsSetColor(k, *[float(c) for c in v])
                else:
                    yi.paramsSet({k: v})
            yi.paramsEndList()
        handles.append(yi.createMaterial(f"mat{i}"))
    for i, l in enumerate(scene["lights"]):
        yi.paramsClearAll()
        yi.paramsSet({k: (_color(v) if k in _COLOR_KEYS else v) for k, v in l.items()})
        yi.createLight(f"light{i}")
    yi.paramsClearAll()
    cam = dict(scene["camera"])
    cam.setdefault("type", "perspective")
    yi.paramsSet(cam)
    yi.createCamera("cam")
    bg = render.get("background")
    if bg is not None:
        yi.paramsClearAll()
        yi.paramsSet({"type": "constant", "color": _color(bg)})
        yi.createBackground("world_background")
    yi.paramsClearAll()
    integ = {"type": render.get("integrator", "pathtracing")}
    integ["raydepth"] = 5      # MonteCarloIntegrator's default r_depth_, spelled out so that oracle and device agree
    for k in ("path_samples", "bounces", "russian_roulette_min_bounces", "no_recursive", "bg_transp", "bg_transp_refract", "raydepth", "transpShad", "shadowDepth"):
        if k in render:
            integ[k] = render[k]
    # (render["caustic_type"]: "none" unless given; "path" is the reference's own default when the parameter is absent, integrator_path_tracer.cc:36)
    integ["caustic_type"] = render.get("caustic_type", "none")
    yi.paramsSet(integ)
    yi.createIntegrator("default")
    yi.paramsClearAll()
    yi.paramsSet({"type": "none"})
    yi.createIntegrator("volintegr")

    yi.startGeometry()
    verts = np.asarray(scene["verts"], dtype=np.float32).reshape(-1, 3, 3)
    tri_mat = np.asarray(scene["tri_mat"], dtype=np.int32)
    vn = scene.get("vnormals")
    n = verts.shape[0]
    uv, orco = scene.get("uv"), scene.get("orco")
    if uv is not None or orco is not None:
        # texture coordinates (shader nodes): per-vertex path with orco and UVs, one mesh, triangles keep their order
        uv = None if uv is None else np.asarray(uv, dtype=np.float32).reshape(-1, 3, 2)
        orco = None if orco is None else np.asarray(orco, dtype=np.float32).reshape(-1, 3, 3)
        vn = None if vn is None else np.asarray(vn, dtype=np.float32).reshape(-1, 3, 3)
        yi.startTriMesh(yi.getNextFreeId(), 3 * n, n, orco is not None, uv is not None, 0)
        for t in range(n):
            for c in range(3):
                if orco is not None:
                    yi.addVertexWithOrco(*[float(x) for x in verts[t, c]], *[float(x) for x in orco[t, c]])
                else:
                    yi.addVertex(*[float(x) for x in verts[t, c]])
                if vn is not None and np.any(vn[t, c] != 0):
                    yi.addNormal(*[float(x) for x in vn[t, c]])
                if uv is not None:
                    yi.addUv(float(uv[t, c, 0]), float(uv[t, c, 1]))
            if uv is not None:
                yi.addTriangleWithUv(3 * t, 3 * t + 1, 3 * t + 2, 3 * t, 3 * t + 1, 3 * t + 2, handles[int(tri_mat[t])])
            else:
                yi.addTriangle(3 * t, 3 * t + 1, 3 * t + 2, handles[int(tri_mat[t])])
        yi.endTriMesh()
    elif vn is not None:
        # per-vertex path so that addNormal is exercised (one mesh; triangles keep their order)
        vn = np.asarray(vn, dtype=np.float32).reshape(-1, 3, 3)
        yi.startTriMesh(yi.getNextFreeId(), 3 * n, n, False, False, 0)
        for t in range(n):
            for c in range(3):
                yi.addVertex(*[float(x) for x in verts[t, c]])
                if np.any(vn[t, c] != 0):
                    yi.addNormal(*[float(x) for x in vn[t, c]])
            yi.addTriangle(3 * t, 3 * t + 1, 3 * t + 2, handles[int(tri_mat[t])])
        yi.endTriMesh()
    else:
        # one mesh; runs of equal material go through the bulk entry point, order preserved
        yi.startTriMesh(yi.getNextFreeId(), 3 * n, n, False, False, 0)
        start = 0
        while start < n:
            end = start + 1
            while end < n and tri_mat[end] == tri_mat[start]:
                end += 1
            k = end - start
            idx = np.arange(3 * k, dtype=np.int32)
            yi.addTriangles(verts[start:end].reshape(-1, 3), idx, handles[int(tri_mat[start])])
            start = end
        yi.endTriMesh()
    if scene.get("smooth_angle") is not None:
        yi.smoothMesh(0, float(scene["smooth_angle"]))      # Interface::smoothMesh on the mesh just closed
    yi.endGeometry()
    set_render_params(yi, render)
    return handles


def set_render_params(yi, render):
    """the parameter map Interface::render reads (load_scene's last step; again after anything that cleared the map)"""
    bg = render.get("background")
    yi.paramsClearAll()
    rs = {"camera_name": "cam", "integrator_name": "default", "volintegrator_name": "volintegr"}
    if bg is not None:
        rs["background_name"] = "world_background"
    for k in ("width", "height", "xstart", "ystart", "AA_passes", "AA_minsamples", "filter_type", "tile_size",
              "adv_base_sampling_offset", "adv_computer_node", "adv_auto_shadow_bias_enabled",
              "adv_auto_min_raydist_enabled", "threads", "AA_inc_samples", "AA_detect_color_noise", "AA_dark_detection_type",
              "AA_variance_edge_size", "AA_variance_pixels"):
        if k in render:
            rs[k] = render[k]
    for k in ("AA_pixelwidth", "adv_shadow_bias_value", "adv_min_raydist_value", "AA_threshold", "AA_resampled_floor",
              "AA_sample_multiplier_factor", "AA_light_sample_multiplier_factor", "AA_indirect_sample_multiplier_factor",
              "AA_dark_threshold_factor", "AA_clamp_samples", "AA_clamp_indirect"):
        if k in render:
            rs[k] = float(render[k])
    yi.paramsSet(rs)
