"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (libyafaray_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

MAT_SHINYDIFFUSE, MAT_GLOSSY, MAT_LIGHT = 0, 1, 2
LIGHT_AREA, LIGHT_POINT = 0, 1
INTEGRATOR_PATH, INTEGRATOR_DIRECT = 0, 1
FILTER_BOX, FILTER_MITCHELL, FILTER_GAUSS, FILTER_LANCZOS = 0, 1, 2, 3

f3 = C.c_float * 3


class MaterialDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("visibility", C.c_int32), ("receive_shadows", C.c_int32), ("flat_material", C.c_int32),
        ("color", f3), ("mirror_color", f3),
        ("diffuse_reflect", C.c_float), ("specular_reflect", C.c_float), ("transparency", C.c_float),
        ("translucency", C.c_float), ("emit", C.c_float), ("ior", C.c_float),
        ("fresnel_effect", C.c_int32), ("transmit_filter", C.c_float), ("oren_nayar", C.c_int32), ("pad0", C.c_float),
        ("sigma", C.c_double),
        ("glossy_color", f3), ("diffuse_color", f3),
        ("glossy_reflect", C.c_float), ("glossy_diffuse_reflect", C.c_float), ("exponent", C.c_float),
        ("as_diffuse", C.c_int32),
        ("light_color", f3), ("light_power", C.c_float), ("double_sided", C.c_int32), ("anisotropic", C.c_int32),
        ("absorption", f3), ("has_absorption", C.c_int32), ("absorption_dist", C.c_double),
        ("n_nodes", C.c_int32),
        ("sh_diffuse", C.c_int32), ("sh_mirror_color", C.c_int32), ("sh_mirror", C.c_int32), ("sh_transparency", C.c_int32),
        ("sh_translucency", C.c_int32), ("sh_sigma_oren", C.c_int32), ("sh_diffuse_refl", C.c_int32), ("sh_ior", C.c_int32),
        ("pad2", C.c_int32), ("nodes", C.c_void_p),
        ("exp_u", C.c_float), ("exp_v", C.c_float),
        ("sh_glossy", C.c_int32), ("sh_glossy_reflect", C.c_int32), ("sh_exponent", C.c_int32), ("sh_filter_color", C.c_int32),
        ("additional_depth", C.c_int32), ("transp_bias_factor", C.c_float), ("transp_bias_mult", C.c_int32),
        ("n_bump_nodes", C.c_int32), ("sh_bump", C.c_int32), ("pad5", C.c_int32), ("bump_nodes", C.c_void_p),
        ("rough_alpha", C.c_float), ("pad6", C.c_int32),
    ]


f4 = C.c_float * 4


class TextureDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("texels", C.POINTER(C.c_float)),
        ("interpolate", C.c_int32), ("clip", C.c_int32),
        ("xrepeat", C.c_int32), ("yrepeat", C.c_int32), ("rot90", C.c_int32), ("mirror_x", C.c_int32), ("mirror_y", C.c_int32),
        ("checker_even", C.c_int32), ("checker_odd", C.c_int32), ("checker_dist", C.c_float),
        ("cropmin_x", C.c_float), ("cropmin_y", C.c_float), ("cropmax_x", C.c_float), ("cropmax_y", C.c_float),
        ("adj_intensity", C.c_float), ("adj_contrast", C.c_float), ("adj_saturation", C.c_float), ("adj_hue", C.c_float),
        ("adj_red", C.c_float), ("adj_green", C.c_float), ("adj_blue", C.c_float), ("adj_clamp", C.c_int32),
        ("color_space", C.c_int32), ("gamma", C.c_float), ("normalmap", C.c_int32),
    ]


class NodeDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("texture", C.c_int32), ("texco", C.c_int32), ("mapping", C.c_int32), ("proj", C.c_int32 * 3),
        ("scale", f3), ("offset", f3), ("mtx", C.c_float * 16), ("do_scalar", C.c_int32),
        ("color", f4), ("scalar", C.c_float),
        ("mode", C.c_int32), ("cfactor", C.c_float), ("input1", C.c_int32), ("input2", C.c_int32), ("factor", C.c_int32),
        ("col1", f4), ("col2", f4),
        ("input", C.c_int32), ("upper_layer", C.c_int32),
        ("no_rgb", C.c_int32), ("stencil", C.c_int32), ("negative", C.c_int32), ("use_alpha", C.c_int32), ("do_color", C.c_int32),
        ("do_scalar_l", C.c_int32), ("color_input", C.c_int32),
        ("colfac", C.c_float), ("valfac", C.c_float), ("def_val", C.c_float), ("def_col", f3), ("upper_col", f4), ("upper_val", C.c_float),
        ("bump_strength", C.c_float),
    ]


TEXCO = {"uv": 0, "global": 1, "orco": 2, "transformed": 3, "normal": 4, "reflect": 5, "window": 6, "stick": 7, "stress": 8, "tangent": 9}
MAPPING = {"plain": 0, "cube": 1, "tube": 2, "sphere": 3}
CLIPPING = {"extend": 0, "clip": 1, "clipcube": 2, "repeat": 3, "checker": 4}
COLOR_SPACE = {"sRGB": 0, "XYZ": 1, "LinearRGB": 2, "Raw_Manual_Gamma": 3}
SHADER_SLOTS = {"diffuse_shader": "sh_diffuse", "mirror_color_shader": "sh_mirror_color", "mirror_shader": "sh_mirror",
                "transparency_shader": "sh_transparency", "translucency_shader": "sh_translucency",
                "sigma_oren_shader": "sh_sigma_oren", "diffuse_refl_shader": "sh_diffuse_refl", "IOR_shader": "sh_ior",
                "glossy_shader": "sh_glossy", "glossy_reflect_shader": "sh_glossy_reflect", "exponent_shader": "sh_exponent", "filter_color_shader": "sh_filter_color"}


def texture_desc(t):
    """t: the reference's ImageTexture::factory parameters (texture_image.cc:545-700) + "texels" (h, w, 4) float32 as the
    image buffer returns them"""
    d = TextureDesc()
    px = np.ascontiguousarray(t["texels"], dtype=np.float32)
    d._keep = px
    d.height, d.width = px.shape[0], px.shape[1]
    d.texels = px.ctypes.data_as(C.POINTER(C.c_float))
    d.interpolate = {"none": 0, "bilinear": 1}[t.get("interpolate", "bilinear")]
    d.clip = CLIPPING.get(t.get("clipping", "repeat"), 3)                # string2Cliptype__: unknown -> repeat
    d.xrepeat, d.yrepeat = t.get("xrepeat", 1), t.get("yrepeat", 1)
    d.rot90, d.mirror_x, d.mirror_y = int(t.get("rot90", False)), int(t.get("mirror_x", False)), int(t.get("mirror_y", False))
    d.checker_even, d.checker_odd, d.checker_dist = int(t.get("even_tiles", False)), int(t.get("odd_tiles", True)), t.get("checker_dist", 0.0)
    d.cropmin_x, d.cropmin_y, d.cropmax_x, d.cropmax_y = t.get("cropmin_x", 0.0), t.get("cropmin_y", 0.0), t.get("cropmax_x", 1.0), t.get("cropmax_y", 1.0)
    d.adj_intensity, d.adj_contrast = t.get("adj_intensity", 1.0), t.get("adj_contrast", 1.0)
    d.adj_saturation, d.adj_hue = t.get("adj_saturation", 1.0), t.get("adj_hue", 0.0)
    d.adj_red, d.adj_green, d.adj_blue = t.get("adj_mult_factor_red", 1.0), t.get("adj_mult_factor_green", 1.0), t.get("adj_mult_factor_blue", 1.0)
    d.adj_clamp = int(t.get("adj_clamp", False))
    d.color_space = COLOR_SPACE.get(t.get("color_space", "Raw_Manual_Gamma"), 0)   # the factory's default (:552); unknown names read as sRGB (:613)
    d.gamma = t.get("gamma", 1.0)
    d.normalmap = int(t.get("normalmap", False))
    return d


def sort_nodes(nodes):
    """evaluation order (NodeMaterial::solveNodesOrder): every node after the nodes it reads; -> (sorted list, name -> index)"""
    by_name = {n["name"]: n for n in nodes}
    order, seen = [], set()

    def visit(n):
        if n["name"] in seen:
            return
        seen.add(n["name"])
        for k in ("input", "upper_layer", "input1", "input2", "factor"):
            if k in n and n[k] in by_name:
                visit(by_name[n[k]])
        order.append(n)
    for n in nodes:
        visit(n)
    return order, {n["name"]: i for i, n in enumerate(order)}


def node_descs(nodes, texture_index):
    """nodes: the material's list_element dicts (reference parameter names) -> (NodeDesc array in evaluation order, name -> index)"""
    order, index = sort_nodes(nodes)
    arr = (NodeDesc * max(1, len(order)))()
    for i, n in enumerate(order):
        d = arr[i]
        ref = lambda key: index.get(n.get(key), -1) if key in n else -1
        t = n["type"]
        d.texture = d.input1 = d.input2 = d.factor = d.input = d.upper_layer = -1
        if t == "texture_mapper":
            d.type = 0
            d.texture = texture_index[n["texture"]]
            d.texco = TEXCO.get(n.get("texco", "global"), 1)
            d.mapping = MAPPING.get(n.get("mapping", "plain"), 0)
            d.proj = (C.c_int32 * 3)(n.get("proj_x", 1), n.get("proj_y", 2), n.get("proj_z", 3))
            d.scale = f3(*n.get("scale", (1, 1, 1))); d.offset = f3(*n.get("offset", (0, 0, 0)))
            d.mtx = (C.c_float * 16)(*np.asarray(n.get("transform", np.eye(4)), np.float32).reshape(16))
            d.do_scalar = int(n.get("do_scalar", True))
            d.bump_strength = n.get("bump_strength", 1.0)
        elif t == "value":
            d.type = 1
            col = n.get("color", (1, 1, 1))
            d.color = f4(col[0], col[1], col[2], n.get("alpha", 1.0)); d.scalar = n.get("scalar", 1.0)
        elif t == "mix":
            d.type = 2
            d.mode = n.get("mode", 0); d.cfactor = n.get("value", n.get("cfactor", 0.5))
            d.input1, d.input2, d.factor = ref("input1"), ref("input2"), ref("factor")
            c1, c2 = n.get("color1", (0, 0, 0, 1)), n.get("color2", (0, 0, 0, 1))
            d.col1 = f4(*(tuple(c1) + (1.0,))[:4]); d.col2 = f4(*(tuple(c2) + (1.0,))[:4])
        elif t == "layer":
            d.type = 3
            d.mode = n.get("mode", 0)
            d.input, d.upper_layer = ref("input"), ref("upper_layer")
            d.no_rgb, d.stencil, d.negative = int(n.get("noRGB", False)), int(n.get("stencil", False)), int(n.get("negative", False))
            d.use_alpha, d.do_color, d.do_scalar_l = int(n.get("use_alpha", False)), int(n.get("do_color", True)), int(n.get("do_scalar", False))
            d.color_input = int(n.get("color_input", True))
            d.colfac, d.valfac, d.def_val = n.get("colfac", 1.0), n.get("valfac", 1.0), n.get("def_val", 1.0)
            d.def_col = f3(*n.get("def_col", (1, 1, 1))[:3])
            uc = n.get("upper_color", (0, 0, 0))
            # configInputs: upper_color is read into an Rgba (alpha from the parameter, 1 by default); absent -> Rgb(0) (alpha 1)
            d.upper_col = f4(*(tuple(uc) + (1.0,))[:4]); d.upper_val = n.get("upper_value", 0.0)
        else:
            raise ValueError(f"shader node type {t}")
    return arr, index


class LightDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("samples", C.c_int32), ("cast_shadows", C.c_int32), ("pad0", C.c_int32),
        ("corner", f3), ("point1", f3), ("point2", f3), ("color", f3), ("power", C.c_float), ("pad1", f3),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("from_", f3), ("to", f3), ("up", f3), ("resx", C.c_int32), ("resy", C.c_int32),
        ("focal", C.c_float), ("aspect_ratio", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
        ("aperture", C.c_float), ("pad0", C.c_float),
        ("dof_distance", C.c_float), ("bokeh_type", C.c_int32), ("bokeh_bias", C.c_int32), ("bokeh_rotation", C.c_float),
    ]


class RenderDesc(C.Structure):
    _fields_ = [
        ("integrator", C.c_int32), ("path_samples", C.c_int32), ("bounces", C.c_int32), ("rr_min_bounces", C.c_int32),
        ("no_recursive", C.c_int32), ("bg_transp", C.c_int32), ("bg_transp_refract", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32), ("xstart", C.c_int32), ("ystart", C.c_int32),
        ("aa_passes", C.c_int32), ("aa_minsamples", C.c_int32), ("aa_pixelwidth", C.c_float),
        ("filter_type", C.c_int32), ("tile_size", C.c_int32), ("base_sampling_offset", C.c_uint32),
        ("shadow_bias_auto", C.c_int32), ("shadow_bias", C.c_float), ("min_raydist_auto", C.c_int32),
        ("min_raydist", C.c_float), ("aa_light_sample_multiplier", C.c_float),
        ("background", f3), ("has_background", C.c_int32), ("tile_seed_rand", C.c_uint32),
        ("rand_srand", C.c_int32), ("rand_skip", C.c_int32),
        ("n_threads", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
        ("aa_inc_samples", C.c_int32), ("aa_threshold", C.c_float), ("aa_resampled_floor", C.c_float),
        ("aa_sample_multiplier_factor", C.c_float), ("aa_light_sample_multiplier_factor", C.c_float),
        ("aa_indirect_sample_multiplier_factor", C.c_float), ("aa_detect_color_noise", C.c_int32),
        ("aa_dark_detection_type", C.c_int32), ("aa_dark_threshold_factor", C.c_float),
        ("aa_variance_edge_size", C.c_int32), ("aa_variance_pixels", C.c_int32), ("aa_clamp_samples", C.c_float),
        ("transp_shad", C.c_int32), ("shadow_depth", C.c_int32), ("raydepth", C.c_int32), ("trace_caustics", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("interior_steps", C.c_uint64),
        ("leaves", C.c_uint64), ("tri_tests", C.c_uint64), ("camera_samples", C.c_uint64),
        ("kd_nodes", C.c_uint32), ("kd_leaf_refs", C.c_uint32), ("build_seconds", C.c_double), ("render_seconds", C.c_double),
    ]


def build(force=False):
    """Compile liboracle.so with the flags in oracle/Makefile (gcc, IEEE, no fast-math)."""
    src = os.path.join(HERE, "yaf_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "yaf_oracle.h"))):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, timeout=300,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.yor_scene_create.restype = C.c_void_p
    L.yor_scene_create.argtypes = [C.c_int32, fp, C.POINTER(C.c_int32), fp, C.c_int32, C.POINTER(MaterialDesc),
                                   C.c_int32, C.POINTER(LightDesc), C.POINTER(CameraDesc)]
    L.yor_scene_destroy.argtypes = [C.c_void_p]
    L.yor_scene_set_textures.argtypes = [C.c_void_p, C.c_int32, C.POINTER(TextureDesc)]
    L.yor_scene_set_texcoords.argtypes = [C.c_void_p, fp, fp]
    L.yor_texture_probe.argtypes = [C.POINTER(TextureDesc), fp, fp]
    L.yor_nodes_probe.argtypes = [C.c_int32, C.POINTER(NodeDesc), C.c_int32, C.POINTER(TextureDesc), C.POINTER(CameraDesc), fp, fp]
    L.yor_render.restype = C.c_int
    L.yor_render.argtypes = [C.c_void_p, C.POINTER(RenderDesc), fp, C.POINTER(Stats)]
    L.yor_scene_set_tree.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), fp]
    L.yor_intersect.restype = C.c_int
    L.yor_intersect.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float, C.POINTER(C.c_int32), fp, fp]
    L.yor_is_shadowed.restype = C.c_int
    L.yor_is_shadowed.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float]
    for name in ("yor_fsin", "yor_fcos", "yor_fexp2", "yor_flog2", "yor_fsqrt", "yor_facos"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.yor_fpow.restype = C.c_float
    L.yor_fpow.argtypes = [C.c_float, C.c_float]
    for name in ("yor_ri_vdc", "yor_ri_s", "yor_ri_lp"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_uint32, C.c_uint32]
    L.yor_fnv32a.restype = C.c_uint32
    L.yor_fnv32a.argtypes = [C.c_uint32]
    L.yor_scr_halton.restype = C.c_double
    L.yor_scr_halton.argtypes = [C.c_int, C.c_uint32]
    L.yor_halton_seq.argtypes = [C.c_uint32, C.c_uint32, C.c_int, fp]
    L.yor_mwc_seq.argtypes = [C.c_uint32, C.c_int, fp]
    L.yor_faure_perm.restype = C.POINTER(C.c_int)
    L.yor_faure_perm.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.yor_create_cs.argtypes = [fp, fp, fp]
    L.yor_sample_cos_hemisphere.argtypes = [fp, fp, fp, C.c_float, C.c_float, fp]
    L.yor_bound_cross.restype = C.c_int
    L.yor_bound_cross.argtypes = [fp, fp, fp, fp, C.c_float, fp, fp]
    L.yor_camera_shoot.argtypes = [C.POINTER(CameraDesc), C.c_float, C.c_float, fp]
    L.yor_camera_shoot_lens.argtypes = [C.POINTER(CameraDesc), C.c_float, C.c_float, C.c_float, C.c_float, fp]
    L.yor_arealight_illum_sample.restype = C.c_int
    L.yor_arealight_illum_sample.argtypes = [C.POINTER(LightDesc), fp, C.c_float, C.c_float, fp]
    L.yor_arealight_intersect.restype = C.c_int
    L.yor_arealight_intersect.argtypes = [C.POINTER(LightDesc), fp, fp, fp]
    L.yor_pointlight_illuminate.restype = C.c_int
    L.yor_pointlight_illuminate.argtypes = [C.POINTER(LightDesc), fp, fp]
    L.yor_material_transparency.argtypes = [C.POINTER(MaterialDesc), fp, fp]
    L.yor_material_specular.argtypes = [C.POINTER(MaterialDesc), fp, C.c_int32, C.POINTER(C.c_int32), fp, C.POINTER(C.c_float)]
    L.yor_material_probe.argtypes = [C.POINTER(MaterialDesc), fp, C.c_int32, C.POINTER(C.c_int32), fp, fp,
                                     C.POINTER(C.c_int32), fp]
    L.yor_lightmat_emit.argtypes = [C.POINTER(MaterialDesc), fp, fp, C.c_int, fp]
    L.yor_material_sample_two.argtypes = [C.POINTER(MaterialDesc), fp, C.c_int32, C.POINTER(C.c_int32), fp]
    L.yor_beer_transmittance.argtypes = [fp, C.c_double, C.c_float, C.POINTER(C.c_int32), fp]
    L.yor_set_trace.argtypes = [fp, C.c_uint64, fp, C.c_uint64, C.c_int]
    L.yor_trace_counts.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = L
    return L


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ------------------------------------------------------------------ descriptors from plain dicts
def material_desc(m):
    """m: dict with the reference's factory parameter names (material_shiny_diffuse.cc:627-646,
    material_glossy.cc:427-439, material_simple.cc:69-71) plus 'type'."""
    d = MaterialDesc()
    t = m["type"]
    # "visibility" / "receive_shadows" are read by the shinydiffuse, glossy, coated_glossy and glass factories only
    # (material_shiny_diffuse.cc:627-646, material_glossy.cc:427-439, material_coated_glossy.cc, material_glass.cc:340-360);
    # MirrorMaterial::factory and LightMaterial::factory (material_glass.cc:486-493, material_simple.cc:63-73) leave the
    # Material defaults (normal, receiving); "flat_material" belongs to shinydiffuse alone
    reads_flags = t in ("shinydiffusemat", "glossy", "coated_glossy", "glass", "rough_glass")
    vis = {"normal": 0, "no_shadows": 1, "shadow_only": 2, "invisible": 3}[m.get("visibility", "normal")]
    d.visibility = vis if reads_flags else 0
    d.receive_shadows = int(m.get("receive_shadows", True)) if reads_flags else 1
    d.flat_material = int(m.get("flat_material", False)) if t == "shinydiffusemat" else 0
    if t == "shinydiffusemat":
        d.type = MAT_SHINYDIFFUSE
        d.color = f3(*m.get("color", (1, 1, 1))[:3])
        d.mirror_color = f3(*m.get("mirror_color", (1, 1, 1))[:3])
        d.diffuse_reflect = m.get("diffuse_reflect", 1.0)
        d.specular_reflect = m.get("specular_reflect", 0.0)
        d.transparency = m.get("transparency", 0.0)
        d.translucency = m.get("translucency", 0.0)
        d.emit = m.get("emit", 0.0)
        d.ior = m.get("IOR", 1.33)
        d.fresnel_effect = int(m.get("fresnel_effect", False))
        d.transmit_filter = m.get("transmit_filter", 1.0)
        d.oren_nayar = int(m.get("diffuse_brdf", "") == "oren_nayar")
        d.sigma = m.get("sigma", 0.1)
    elif t == "glossy":
        d.type = MAT_GLOSSY
        d.glossy_color = f3(*m.get("color", (1, 1, 1))[:3])
        d.diffuse_color = f3(*m.get("diffuse_color", (1, 1, 1))[:3])
        d.glossy_diffuse_reflect = m.get("diffuse_reflect", 0.0)
        d.glossy_reflect = m.get("glossy_reflect", 1.0)
        d.exponent = m.get("exponent", 50.0)
        d.as_diffuse = int(m.get("as_diffuse", True))
        d.anisotropic, d.exp_u, d.exp_v = int(m.get("anisotropic", False)), m.get("exp_u", 50.0), m.get("exp_v", 50.0)
        d.oren_nayar = int(m.get("diffuse_brdf", "") == "Oren-Nayar")
        d.sigma = m.get("sigma", 0.1)
    elif t == "coated_glossy":
        d.type = 5
        d.glossy_color = f3(*m.get("color", (1, 1, 1))[:3])
        d.diffuse_color = f3(*m.get("diffuse_color", (1, 1, 1))[:3])
        d.mirror_color = f3(*m.get("mirror_color", (1, 1, 1))[:3])
        d.glossy_diffuse_reflect = m.get("diffuse_reflect", 0.0)
        d.glossy_reflect = m.get("glossy_reflect", 1.0)
        d.exponent = m.get("exponent", 50.0)
        d.as_diffuse = int(m.get("as_diffuse", True))
        d.anisotropic, d.exp_u, d.exp_v = int(m.get("anisotropic", False)), m.get("exp_u", 50.0), m.get("exp_v", 50.0)
        d.specular_reflect = m.get("specular_reflect", 1.0)
        ior = m.get("IOR", 1.4)
        d.ior = 1.0000001 if ior == 1.0 else ior
        d.oren_nayar = int(m.get("diffuse_brdf", "") == "Oren-Nayar")
        d.sigma = m.get("sigma", 0.1)
    elif t == "glass":
        d.type = 3
        d.color = f3(*m.get("filter_color", (1, 1, 1))[:3])
        d.mirror_color = f3(*m.get("mirror_color", (1, 1, 1))[:3])
        d.ior = m.get("IOR", 1.4)
        d.sigma = m.get("transmit_filter", 0.0)          # a double parameter: filt * filt_col + (1 - filt)
        d.fresnel_effect = int(m.get("fake_shadows", False))
        if m.get("dispersion_power", 0.0) not in (0, 0.0):
            raise ValueError("glass: dispersion is outside the restated path")
        if "absorption" in m:
            d.absorption = f3(*m["absorption"][:3])
            d.has_absorption = 1
            d.absorption_dist = m.get("absorption_dist", 1.0)
    elif t == "rough_glass":
        d.type = 6
        d.color = f3(*m.get("filter_color", (1, 1, 1))[:3])
        d.mirror_color = f3(*m.get("mirror_color", (1, 1, 1))[:3])
        d.ior = m.get("IOR", 1.4)
        d.transmit_filter = m.get("transmit_filter", 0.0)
        d.fresnel_effect = int(m.get("fake_shadows", False))
        d.rough_alpha = m.get("alpha", 0.5)
        if m.get("dispersion_power", 0.0) not in (0, 0.0):
            raise ValueError("rough_glass: dispersion is outside the restated path")
        if "absorption" in m:
            d.absorption = f3(*m["absorption"][:3])
            d.has_absorption = 1
            d.absorption_dist = m.get("absorption_dist", 1.0)
    elif t == "mirror":
        d.type = 4
        d.color = f3(*m.get("color", (1, 1, 1))[:3])
        d.specular_reflect = m.get("reflect", 1.0)
    elif t == "light_mat":
        d.type = MAT_LIGHT
        d.light_color = f3(*m.get("color", (1, 1, 1))[:3])
        d.light_power = m.get("power", 1.0)
        d.double_sided = int(m.get("double_sided", False))
    else:
        raise ValueError(t)
    d.sh_diffuse = d.sh_mirror_color = d.sh_mirror = d.sh_transparency = d.sh_translucency = d.sh_sigma_oren = d.sh_diffuse_refl = d.sh_ior = -1
    d.sh_glossy = d.sh_glossy_reflect = d.sh_exponent = d.sh_filter_color = -1
    if t in ("shinydiffusemat", "glossy", "coated_glossy", "glass", "rough_glass"):
        d.additional_depth = m.get("additionaldepth", 0)
    if t == "shinydiffusemat":
        d.transp_bias_factor = m.get("transparentbias_factor", 0.0)
        d.transp_bias_mult = int(m.get("transparentbias_multiply_raydepth", False))
    return d


def light_desc(l):
    d = LightDesc()
    d.cast_shadows = int(l.get("cast_shadows", True))
    d.color = f3(*l.get("color", (1, 1, 1))[:3])
    d.power = l.get("power", 1.0)
    if l["type"] == "arealight":
        d.type = LIGHT_AREA
        d.samples = l.get("samples", 4)
        d.corner = f3(*l.get("corner", (0, 0, 0)))
        d.point1 = f3(*l.get("point1", (0, 0, 0)))
        d.point2 = f3(*l.get("point2", (0, 0, 0)))
    elif l["type"] == "pointlight":
        d.type = LIGHT_POINT
        d.corner = f3(*l.get("from", (0, 0, 0)))
    else:
        raise ValueError(l["type"])
    return d


BOKEH_TYPES = {"disk1": 0, "disk2": 1, "triangle": 3, "square": 4, "pentagon": 5, "hexagon": 6, "ring": 7}   # BokehType


def camera_desc(c):
    d = CameraDesc()
    d.from_ = f3(*c.get("from", (0, 1, 0)))
    d.to = f3(*c.get("to", (0, 0, 0)))
    d.up = f3(*c.get("up", (0, 1, 1)))
    d.resx = c.get("resx", 320)
    d.resy = c.get("resy", 200)
    d.focal = c.get("focal", 1.0)
    d.aspect_ratio = c.get("aspect_ratio", 1.0)
    d.near_clip = c.get("nearClip", 0.0)
    d.far_clip = c.get("farClip", -1.0)
    d.aperture = c.get("aperture", 0.0)
    d.dof_distance = c.get("dof_distance", 0.0)
    d.bokeh_type = BOKEH_TYPES[c.get("bokeh_type", "disk1")]
    d.bokeh_bias = {"uniform": 0, "center": 1, "edge": 2}.get(c.get("bokeh_bias", "uniform"), 0)
    d.bokeh_rotation = c.get("bokeh_rotation", 0.0)
    return d


def render_desc(r):
    d = RenderDesc()
    d.integrator = {"pathtracing": INTEGRATOR_PATH, "directlighting": INTEGRATOR_DIRECT}[r.get("integrator", "pathtracing")]
    d.path_samples = r.get("path_samples", 32)
    d.bounces = r.get("bounces", 3)
    d.rr_min_bounces = r.get("russian_roulette_min_bounces", 0)
    d.no_recursive = int(r.get("no_recursive", False))
    # PathIntegrator::factory: "none" clears trace_caustics_, "both" / "photon" need the photon map (out of scope); anything else,
    # "path" included, leaves the constructor's Path (integrator_path_tracer.cc:36, :382-387).  These dicts default to "none".
    d.trace_caustics = int(r.get("caustic_type", "none") != "none")
    d.bg_transp = int(r.get("bg_transp", False))
    d.bg_transp_refract = int(r.get("bg_transp_refract", False))
    d.width = r["width"]
    d.height = r["height"]
    d.xstart = r.get("xstart", 0)
    d.ystart = r.get("ystart", 0)
    d.aa_passes = r.get("AA_passes", 1)
    d.aa_minsamples = r.get("AA_minsamples", 1)
    d.aa_pixelwidth = r.get("AA_pixelwidth", 1.5)
    d.filter_type = {"box": 0, "mitchell": 1, "gauss": 2, "lanczos": 3}[r.get("filter_type", "box")]
    d.tile_size = r.get("tile_size", 32)
    d.base_sampling_offset = r.get("adv_base_sampling_offset", 0) + 100000 * r.get("adv_computer_node", 0)
    d.shadow_bias_auto = int(r.get("adv_auto_shadow_bias_enabled", True))
    d.shadow_bias = r.get("adv_shadow_bias_value", 0.0005)
    d.min_raydist_auto = int(r.get("adv_auto_min_raydist_enabled", True))
    d.min_raydist = r.get("adv_min_raydist_value", 0.00005)
    d.aa_light_sample_multiplier = 1.0
    bg = r.get("background")
    if bg is not None:
        d.background = f3(*bg)
        d.has_background = 1
    d.tile_seed_rand = r.get("tile_seed_rand", 0)
    d.rand_srand = r.get("rand_srand", -1)
    d.rand_skip = r.get("rand_skip", 0)
    d.n_threads = r.get("oracle_threads", 1)
    d.shard_index = r.get("shard_index", 0)
    d.shard_count = r.get("shard_count", 1)
    # multi-pass anti-aliasing: defaults of environment.cc:682-695
    d.aa_inc_samples = r.get("AA_inc_samples", d.aa_minsamples)
    d.aa_threshold = r.get("AA_threshold", 0.05)
    d.aa_resampled_floor = r.get("AA_resampled_floor", 0.0)
    d.aa_sample_multiplier_factor = r.get("AA_sample_multiplier_factor", 1.0)
    d.aa_light_sample_multiplier_factor = r.get("AA_light_sample_multiplier_factor", 1.0)
    d.aa_indirect_sample_multiplier_factor = r.get("AA_indirect_sample_multiplier_factor", 1.0)
    d.aa_detect_color_noise = int(r.get("AA_detect_color_noise", False))
    d.aa_dark_detection_type = {"none": 0, "linear": 1, "curve": 2}[r.get("AA_dark_detection_type", "none")]
    d.aa_dark_threshold_factor = r.get("AA_dark_threshold_factor", 0.0)
    d.aa_variance_edge_size = r.get("AA_variance_edge_size", 10)
    d.aa_variance_pixels = r.get("AA_variance_pixels", 0)
    d.aa_clamp_samples = r.get("AA_clamp_samples", 0.0)
    d.transp_shad = int(r.get("transpShad", False))
    d.shadow_depth = r.get("shadowDepth", 5)  # integrator_path_tracer.cc:352
    d.raydepth = r.get("raydepth", 5)       # MonteCarloIntegrator default r_depth_
    return d


class OracleScene:
    """A scene description dict -> oracle scene.  scene = {verts: (N,3,3) f32, tri_mat: (N,) i32,
    vnormals: None | (N,3,3) f32, materials: [dict], lights: [dict], camera: dict}."""

    def __init__(self, scene):
        L = lib()
        self.verts = np.ascontiguousarray(scene["verts"], dtype=np.float32).reshape(-1, 9)
        self.tri_mat = np.ascontiguousarray(scene["tri_mat"], dtype=np.int32)
        n = self.verts.shape[0]
        vn = scene.get("vnormals")
        self.vn = None if vn is None else np.ascontiguousarray(vn, dtype=np.float32).reshape(-1, 9)
        textures = scene.get("textures") or []
        tex_index = {t.get("name", f"tex{i}"): i for i, t in enumerate(textures)}
        mdescs, self._node_arrays = [], []
        for m in scene["materials"]:
            d = material_desc(m)
            if m.get("nodes"):
                arr, index = node_descs(m["nodes"], tex_index)
                self._node_arrays.append(arr)
                d.n_nodes = len(m["nodes"]); d.nodes = C.cast(arr, C.c_void_p)
                for pname, field in SHADER_SLOTS.items():
                    if pname in m and m[pname] in index:
                        setattr(d, field, index[m[pname]])
                if m.get("bump_shader") in index:
                    # bump_nodes_: what the bump shader reaches, in evaluation order (NodeMaterial::getNodeList)
                    by_name = {n["name"]: n for n in m["nodes"]}
                    order, seen = [], set()

                    def visit(n):
                        if n["name"] in seen:
                            return
                        seen.add(n["name"])
                        for k in ("input1", "input2", "factor", "input", "upper_layer"):
                            if k in n and n[k] in by_name:
                                visit(by_name[n[k]])
                        order.append(n)
                    visit(by_name[m["bump_shader"]])
                    barr, bindex = node_descs(order, tex_index)
                    assert [bindex[n["name"]] for n in order] == list(range(len(order)))
                    self._node_arrays.append(barr)
                    d.n_bump_nodes = len(order); d.sh_bump = len(order) - 1; d.bump_nodes = C.cast(barr, C.c_void_p)
            mdescs.append(d)
        mats = (MaterialDesc * len(scene["materials"]))(*mdescs)
        lights = (LightDesc * max(1, len(scene["lights"])))(*[light_desc(l) for l in scene["lights"]])
        cam = camera_desc(scene["camera"])
        self.h = L.yor_scene_create(n, fptr(self.verts), self.tri_mat.ctypes.data_as(C.POINTER(C.c_int32)),
                                    None if self.vn is None else fptr(self.vn),
                                    len(scene["materials"]), mats, len(scene["lights"]), lights, C.byref(cam))
        if textures:
            self._tdescs = [texture_desc(t) for t in textures]
            L.yor_scene_set_textures(self.h, len(textures), (TextureDesc * len(textures))(*self._tdescs))
        uv, orco = scene.get("uv"), scene.get("orco")
        if uv is not None or orco is not None:
            self._uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 6)
            self._orco = None if orco is None else np.ascontiguousarray(orco, dtype=np.float32).reshape(-1, 9)
            L.yor_scene_set_texcoords(self.h, None if self._uv is None else fptr(self._uv), None if self._orco is None else fptr(self._orco))

    def render(self, render):
        L = lib()
        rd = render_desc(render)
        film = np.zeros((rd.height, rd.width, 5), dtype=np.float32)
        st = Stats()
        rc = L.yor_render(self.h, C.byref(rd), fptr(film), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"oracle: unsupported configuration (code {rc})")
        return film, st

    def render_traced(self, render, cap_samples, cap_rays, with_shadow=False):
        """single-threaded render with the per-sample / per-query trace on: (film, stats, samples (n, 8), rays (m, 12): from, dir,
        tmin, tmax, t or -1, the triangle index as int bits, the pixel — with_shadow: any-hit queries too, verdict in column 8 and -2
        in the triangle column —, total queries recorded)"""
        L = lib()
        samples = np.zeros((max(1, cap_samples), 8), dtype=np.float32)
        rays = np.zeros((max(1, cap_rays), 12), dtype=np.float32)
        L.yor_set_trace(fptr(samples), cap_samples, fptr(rays), cap_rays, int(with_shadow))
        try:
            film, st = self.render(dict(render, oracle_threads=1))
            ns, nr = C.c_uint64(), C.c_uint64()
            L.yor_trace_counts(C.byref(ns), C.byref(nr))
        finally:
            L.yor_set_trace(None, 0, None, 0, 0)
        return film, st, samples[:min(ns.value, cap_samples)], rays[:min(nr.value, cap_rays)], nr.value

    def set_tree(self, nodes, refs, bound6):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 2)
        refs = np.ascontiguousarray(refs, dtype=np.uint32)
        b = np.ascontiguousarray(bound6, dtype=np.float32)
        up = C.POINTER(C.c_uint32)
        lib().yor_scene_set_tree(self.h, nodes.shape[0], nodes.ctypes.data_as(up), refs.shape[0], refs.ctypes.data_as(up), fptr(b))

    def intersect(self, frm, dr, tmin=0.0, tmax=-1.0, use_tree=True):
        L = lib()
        f = np.asarray(frm, dtype=np.float32)
        d = np.asarray(dr, dtype=np.float32)
        tri = C.c_int32()
        t = C.c_float()
        bary = np.zeros(3, dtype=np.float32)
        hit = L.yor_intersect(self.h, int(use_tree), fptr(f), fptr(d), tmin, tmax, C.byref(tri), C.byref(t), fptr(bary))
        return hit, tri.value, t.value, bary

    def is_shadowed(self, frm, dr, tmin, tmax, use_tree=True):
        L = lib()
        f = np.asarray(frm, dtype=np.float32)
        d = np.asarray(dr, dtype=np.float32)
        return L.yor_is_shadowed(self.h, int(use_tree), fptr(f), fptr(d), tmin, tmax)

    def close(self):
        if self.h:
            lib().yor_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def film_to_rgb(film):
    """Pixel::normalized, util_image_buffers.h:39-43: col/weight when weight != 0."""
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(w != 0, film[..., :4] / w, 0.0)
    return out.astype(np.float32)
