"""Pins the CPU oracle (oracle/yaf_oracle.c) against golden vectors produced by the reference's own
sources (tests/golden/make_golden.py -> oracle/_ref component harness).

Two fixture sets:
  ieee : reference built -O2 -ffp-contract=off  -> the oracle must match BIT FOR BIT
  fast : reference built with its release flags (-O3 -ffast-math) -> within FAST_RTOL, the
         reassociation noise of -ffast-math measured between the two reference builds themselves.
"""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
FAST_RTOL = 2e-3   # fPow at exponent 500 amplifies 1-ulp input noise to ~4e-4 between the two reference builds
FAST_ATOL = 1e-6


def load(variant):
    with gzip.open(os.path.join(HERE, "golden", f"ref_components_{variant}.json.gz"), "rt") as f:
        d = json.load(f)
    return {k: np.array(v, dtype=np.uint32 if not k.endswith(("_ok", "_hit", "_flags", "_sflags_in", "_sflags_out")) else np.int64)
            for k, v in d.items()}


def f32(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def check(variant, got, want_u32, what):
    got = np.asarray(got, dtype=np.float32).ravel()
    want = f32(want_u32).ravel()
    assert got.shape == want.shape, what
    if variant == "ieee":
        bad = got.view(np.uint32) != want.view(np.uint32)
        # +0/-0 and NaN payloads are not distinguished
        bad &= ~((got == 0) & (want == 0))
        bad &= ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), f"{what}: {bad.sum()}/{bad.size} not bit-exact, first idx {np.argmax(bad)}: got {got[bad][:4]} want {want[bad][:4]}"
    else:
        ok = np.isclose(got, want, rtol=FAST_RTOL, atol=FAST_ATOL) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), f"{what}: {(~ok).sum()}/{ok.size} outside tolerance, e.g. got {got[~ok][:4]} want {want[~ok][:4]}"


VARIANTS = ["ieee", "fast"]


@pytest.fixture(scope="module", params=VARIANTS)
def gold(request):
    return request.param, load(request.param)


def test_fastmath(gold):
    variant, g = gold
    L = po.lib()
    x = f32(g["fm_x"])
    check(variant, [L.yor_fsin(v) for v in x], g["fm_sin"], "fSin__")
    check(variant, [L.yor_fcos(v) for v in x], g["fm_cos"], "fCos__")
    check(variant, [L.yor_fexp2(v) for v in x], g["fm_exp2"], "fExp2__")
    ax = (np.abs(x) + np.float32(1e-3)).astype(np.float32)
    check(variant, [L.yor_flog2(v) for v in ax], g["fm_log2_absx"], "fLog2__")
    check(variant, [L.yor_fsqrt(v) for v in ax], g["fm_sqrt_absx"], "fSqrt__")
    a, b = f32(g["fm_pow_a"]), f32(g["fm_pow_b"])
    check(variant, [L.yor_fpow(p, q) for p, q in zip(a, b)], g["fm_pow"], "fPow__")


def test_facos_libm(gold):
    # fAcos__ calls libm acos: the only libm transcendental among the helpers; not on the configs' path
    variant, g = gold
    L = po.lib()
    x = (f32(g["fm_x"]) * np.float32(0.2)).astype(np.float32)
    got = np.array([L.yor_facos(v) for v in x], dtype=np.float32)
    np.testing.assert_allclose(got, f32(g["fm_acos_02x"]), rtol=3e-7, atol=1e-7)


def test_qmc(gold):
    variant, g = gold
    L = po.lib()
    bits, r = g["q_bits"], g["q_r"]
    check(variant, [L.yor_ri_vdc(int(b), int(q)) for b, q in zip(bits, r)], g["q_vdc"], "riVdC__")
    check(variant, [L.yor_ri_s(int(b), int(q)) for b, q in zip(bits, r)], g["q_ris"], "riS__")
    check(variant, [L.yor_ri_lp(int(b), int(q)) for b, q in zip(bits, r)], g["q_rilp"], "riLp__")
    assert [L.yor_fnv32a(int(b)) for b in bits] == [int(v) for v in g["q_fnv"]]
    # integer / double QMC is exact in both builds
    got = np.array([L.yor_scr_halton(int(d), int(n)) for d, n in zip(g["sh_dim"], g["sh_n"])], dtype=np.float64)
    want = (g["sh_f64lo"].astype(np.uint64) | (g["sh_f64hi"].astype(np.uint64) << np.uint64(32))).view(np.float64)
    if variant == "ieee":
        assert (got.view(np.uint64) == want.view(np.uint64)).all()
    else:
        np.testing.assert_allclose(got, want, rtol=1e-14)
    check("fast" if variant == "fast" else "ieee", got.astype(np.float32), g["sh_f32"], "scrHalton__ f32")
    # every entry of every Faure permutation
    prims = [1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101,
             103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227]
    fd = [np.float32(L.yor_scr_halton(d, n)) for d in range(1, 50) for n in range(prims[d])]
    check(variant, fd, g["sh_firstdigit_f32"], "scrHalton__ first digits (Faure tables)")
    # incremental Halton
    out = np.zeros(6, dtype=np.float32)
    got = []
    for base, start in zip(g["h_base"], g["h_start"]):
        L.yor_halton_seq(int(base), int(start), 6, po.fptr(out))
        got.extend(out.tolist())
    check(variant, got, g["h_next6"], "Halton::setStart/getNext")
    out = np.zeros(8, dtype=np.float32)
    got = []
    for seed in g["rng_seed"]:
        L.yor_mwc_seq(int(seed), 8, po.fptr(out))
        got.extend(out.tolist())
    check(variant, got, g["rng_f32x8"], "Random (MWC)")


def test_cs_hemisphere_bound(gold):
    variant, g = gold
    L = po.lib()
    n = f32(g["g_n"]).reshape(-1, 3).copy()
    s12 = f32(g["g_s12"]).reshape(-1, 2)
    cs, hemi = [], []
    for i in range(n.shape[0]):
        u = np.zeros(3, np.float32); v = np.zeros(3, np.float32); w = np.zeros(3, np.float32)
        L.yor_create_cs(po.fptr(n[i]), po.fptr(u), po.fptr(v))
        L.yor_sample_cos_hemisphere(po.fptr(n[i]), po.fptr(u), po.fptr(v), s12[i, 0], s12[i, 1], po.fptr(w))
        cs.extend(u.tolist() + v.tolist()); hemi.extend(w.tolist())
    check(variant, cs, g["g_cs_uv"], "createCs__")
    check(variant, hemi, g["g_coshemi"], "sampleCosHemisphere__")
    a = np.array([-1.0, -0.5, -2.0], np.float32); gg = np.array([1.5, 0.75, 0.25], np.float32)
    inp = f32(g["bc_in7"]).reshape(-1, 7).copy()
    ab = f32(g["bc_ab"]).reshape(-1, 2)
    hits, outs, want = [], [], []
    for i in range(inp.shape[0]):
        e = C.c_float(-7.0); l = C.c_float(-7.0)
        h = L.yor_bound_cross(po.fptr(a), po.fptr(gg), po.fptr(inp[i, 0:3].copy()), po.fptr(inp[i, 3:6].copy()), inp[i, 6], C.byref(e), C.byref(l))
        hits.append(h)
        if h:
            outs.extend([e.value, l.value]); want.extend(ab[i].view(np.uint32).tolist())
    assert hits == [int(v) for v in g["bc_hit"]]
    check(variant, outs, np.array(want, dtype=np.uint32), "Bound::cross")


def test_camera_depth_of_field(gold):
    """PerspectiveCamera with aperture != 0: lens sampling for every bokeh shape and bias (camera_perspective.cc:75-156)"""
    variant, g = gold
    L = po.lib()
    types = ["disk1", "disk2", "triangle", "square", "pentagon", "hexagon", "ring"]
    biases = ["uniform", "center", "edge"]
    cfg = g["camd_cfg17"].reshape(-1, 17)
    in4 = f32(g["camd_in4"]).reshape(len(cfg), -1, 4)
    got = []
    for c in range(len(cfg)):
        fl = f32(cfg[c])
        cam = po.camera_desc({"from": fl[0:3], "to": fl[3:6], "up": fl[6:9], "resx": int(cfg[c][9]), "resy": int(cfg[c][10]),
                              "focal": float(fl[11]), "aperture": float(fl[12]), "dof_distance": float(fl[13]),
                              "bokeh_type": types[int(cfg[c][14])], "bokeh_bias": biases[int(cfg[c][15])], "bokeh_rotation": float(fl[16])})
        out = np.zeros(9, np.float32)
        for k in range(in4.shape[1]):
            L.yor_camera_shoot_lens(C.byref(cam), in4[c, k, 0], in4[c, k, 1], in4[c, k, 2], in4[c, k, 3], po.fptr(out))
            got.extend(out.tolist())
    check(variant, got, g["camd_ray9"], "PerspectiveCamera::shootRay (depth of field)")


def test_camera(gold):
    variant, g = gold
    L = po.lib()
    cfg = g["cam_cfg12"].reshape(-1, 12)
    pxy = f32(g["cam_pxy"]).reshape(len(cfg), -1, 2)
    got = []
    for c in range(len(cfg)):
        fl = f32(cfg[c])
        cam = po.camera_desc({"from": fl[0:3], "to": fl[3:6], "up": fl[6:9], "resx": int(cfg[c][9]), "resy": int(cfg[c][10]),
                              "focal": float(fl[11])})
        out = np.zeros(9, np.float32)
        for k in range(pxy.shape[1]):
            L.yor_camera_shoot(C.byref(cam), pxy[c, k, 0], pxy[c, k, 1], po.fptr(out))
            got.extend(out.tolist())
    check(variant, got, g["cam_ray9"], "PerspectiveCamera::shootRay")


def test_beer_volume_handler(gold):
    """BeerVolumeHandler(absorption colour, distance)::transmittance — the absorption of a glass material."""
    variant, g = gold
    L = po.lib()
    inp = f32(g["beer_in5"]).reshape(-1, 5).copy()
    ref = g["beer_out4"].reshape(-1, 4)
    got = []
    out = np.zeros(3, np.float32)
    ok = C.c_int32(0)
    for row in inp:
        L.yor_beer_transmittance(po.fptr(row[0:3].copy()), float(row[3]), float(row[4]), C.byref(ok), po.fptr(out))
        assert ok.value == 1
        got.extend(out.tolist())
    assert np.all(ref[:, 0] == 1)
    check(variant, got, np.ascontiguousarray(ref[:, 1:4]).reshape(-1), "BeerVolumeHandler::transmittance")


def test_lights(gold):
    variant, g = gold
    L = po.lib()
    c = f32(g["al_cfg13"])
    al = po.light_desc({"type": "arealight", "corner": c[0:3], "point1": c[3:6], "point2": c[6:9], "color": c[9:12],
                        "power": float(c[12]), "samples": 1})
    inp = f32(g["al_is_in5"]).reshape(-1, 5).copy()
    out = np.zeros(8, np.float32)
    oks, got = [], []
    for i in range(inp.shape[0]):
        oks.append(L.yor_arealight_illum_sample(C.byref(al), po.fptr(inp[i, 0:3].copy()), inp[i, 3], inp[i, 4], po.fptr(out)))
        got.extend(out.tolist())
    assert oks == [int(v) for v in g["al_is_ok"]]
    check(variant, got, g["al_is_out8"], "AreaLight::illumSample")
    inp = f32(g["al_ix_in6"]).reshape(-1, 6).copy()
    out = np.zeros(5, np.float32)
    oks, got = [], []
    for i in range(inp.shape[0]):
        oks.append(L.yor_arealight_intersect(C.byref(al), po.fptr(inp[i, 0:3].copy()), po.fptr(inp[i, 3:6].copy()), po.fptr(out)))
        got.extend(out.tolist())
    if variant == "ieee":
        assert oks == [int(v) for v in g["al_ix_ok"]]
        check(variant, got, g["al_ix_out5"], "AreaLight::intersect")
    else:
        same = np.array(oks) == g["al_ix_ok"]
        assert same.mean() > 0.98  # edge rays may flip under -ffast-math
        gotm = np.array(got, np.float32).reshape(-1, 5)[same]
        check(variant, gotm, g["al_ix_out5"].reshape(-1, 5)[same], "AreaLight::intersect")
    c = f32(g["pl_cfg7"])
    pl = po.light_desc({"type": "pointlight", "from": c[0:3], "color": c[3:6], "power": float(c[6])})
    inp = f32(g["pl_in3"]).reshape(-1, 3).copy()
    out = np.zeros(7, np.float32)
    got = []
    for i in range(inp.shape[0]):
        L.yor_pointlight_illuminate(C.byref(pl), po.fptr(inp[i].copy()), po.fptr(out))
        got.extend(out.tolist())
    check(variant, got, g["pl_out7"], "PointLight::illuminate")


# RenderState::raylevel_ when the golden getSpecular calls were made (GlassMaterial::getSpecular depends on it)
# rg2: alpha at the factory's clamp (1e-4): D() reaches 1e8 and the two reference builds differ by 5 % on it; pinned against the IEEE build only
FAST_SKIP = {"rg2"}
SPEC_RAYLEVEL = {"gg0d": 4, "gg1": 2, "cg1": 6, "cg2": 2}

MATERIALS = {
    "sd0": {"type": "shinydiffusemat", "color": (0.7, 0.6, 0.5), "diffuse_reflect": 0.9},
    "sd1": {"type": "shinydiffusemat", "color": (0.8, 0.3, 0.2), "mirror_color": (0.9, 0.95, 1.0), "diffuse_reflect": 0.8,
            "specular_reflect": 0.3, "transparency": 0.2, "translucency": 0.25, "fresnel_effect": True, "IOR": 1.45,
            "transmit_filter": 0.7, "emit": 0.1},
    "sd2": {"type": "shinydiffusemat", "color": (0.5, 0.7, 0.4), "diffuse_reflect": 1.0, "diffuse_brdf": "oren_nayar", "sigma": 0.35},
    "sd3": {"type": "shinydiffusemat", "color": (0.6, 0.6, 0.7), "mirror_color": (0.9, 0.8, 0.7), "diffuse_reflect": 0.7, "specular_reflect": 0.45},
    "sd4": {"type": "shinydiffusemat", "color": (0.5, 0.8, 0.6), "mirror_color": (1.0, 1.0, 1.0), "diffuse_reflect": 0.6,
            "specular_reflect": 0.5, "transparency": 0.6, "fresnel_effect": True, "IOR": 1.33, "transmit_filter": 0.4},
    "gg0": {"type": "glass", "IOR": 1.52, "filter_color": (0.6, 0.9, 0.7), "transmit_filter": 0.8, "mirror_color": (0.95, 0.9, 1.0)},
    "gg0d": {"type": "glass", "IOR": 1.52, "filter_color": (0.6, 0.9, 0.7), "transmit_filter": 0.8, "mirror_color": (0.95, 0.9, 1.0)},
    "gg1": {"type": "glass", "IOR": 2.1, "filter_color": (1.0, 0.5, 0.5), "transmit_filter": 0.3, "fake_shadows": True},
    "cg0": {"type": "coated_glossy", "color": (0.9, 0.8, 0.7), "diffuse_color": (0.3, 0.5, 0.7), "mirror_color": (1.0, 0.95, 0.9),
            "diffuse_reflect": 0.5, "glossy_reflect": 0.6, "exponent": 80.0, "specular_reflect": 0.8, "IOR": 1.6, "as_diffuse": True},
    "cg1": {"type": "coated_glossy", "color": (1, 1, 1), "glossy_reflect": 0.9, "exponent": 300.0, "specular_reflect": 1.0, "IOR": 1.0},
    "cg2": {"type": "coated_glossy", "color": (0.8, 0.8, 0.8), "diffuse_color": (0.7, 0.3, 0.2), "diffuse_reflect": 0.8, "glossy_reflect": 0.3,
            "exponent": 25.0, "specular_reflect": 0.5, "IOR": 1.8, "diffuse_brdf": "Oren-Nayar", "sigma": 0.3},
    "mi0": {"type": "mirror", "color": (0.9, 0.8, 0.6), "reflect": 0.85},
    "gl0": {"type": "glossy", "color": (0.9, 0.85, 0.8), "diffuse_color": (0.4, 0.5, 0.6), "diffuse_reflect": 0.4,
            "glossy_reflect": 0.6, "exponent": 50.0, "as_diffuse": True},
    "gl1": {"type": "glossy", "color": (1, 1, 1), "glossy_reflect": 0.8, "exponent": 500.0, "as_diffuse": True},
    "gl2": {"type": "glossy", "color": (0.9, 0.9, 0.9), "diffuse_color": (0.6, 0.2, 0.2), "diffuse_reflect": 0.7,
            "glossy_reflect": 0.3, "exponent": 20.0, "diffuse_brdf": "Oren-Nayar", "sigma": 0.25},
    "gl3": {"type": "glossy", "color": (0.9, 0.8, 0.85), "diffuse_color": (0.5, 0.4, 0.6), "diffuse_reflect": 0.5, "glossy_reflect": 0.5,
            "anisotropic": True, "exp_u": 400.0, "exp_v": 12.0},
    "gl4": {"type": "glossy", "color": (1, 1, 1), "glossy_reflect": 0.9, "anisotropic": True, "exp_u": 8.0, "exp_v": 900.0},
    "cg3": {"type": "coated_glossy", "color": (0.9, 0.9, 0.8), "diffuse_color": (0.2, 0.6, 0.5), "diffuse_reflect": 0.6, "glossy_reflect": 0.5,
            "specular_reflect": 0.7, "IOR": 1.5, "as_diffuse": True, "anisotropic": True, "exp_u": 30.0, "exp_v": 250.0},
    # rough glass (material_rough_glass.cc): the GGX lobe that reflects and transmits
    "rg0": {"type": "rough_glass", "IOR": 1.5, "filter_color": (0.7, 0.9, 0.8), "transmit_filter": 0.7, "mirror_color": (0.95, 0.9, 1.0), "alpha": 0.3},
    "rg1": {"type": "rough_glass", "IOR": 2.0, "filter_color": (1.0, 0.6, 0.5), "transmit_filter": 0.4, "alpha": 0.9, "fake_shadows": True},
    "rg2": {"type": "rough_glass", "IOR": 1.33, "alpha": 0.0001},
}


@pytest.mark.parametrize("name", sorted(MATERIALS))
def test_materials(gold, name):
    variant, g = gold
    if variant == "fast" and name in FAST_SKIP:
        pytest.skip("a near-singular lobe amplifies -ffast-math's reassociation to several percent between the reference's own two builds")
    L = po.lib()
    md = po.material_desc(MATERIALS[name])
    inp = f32(g[f"{name}_in14"]).reshape(-1, 14).copy()
    sfl = g[f"{name}_sflags_in"]
    ev, pd, sm, fl, sfo = [], [], [], [], []
    e = np.zeros(3, np.float32); s8 = np.zeros(8, np.float32)
    for i in range(inp.shape[0]):
        bf = C.c_int32(); p = C.c_float(); so = C.c_int32()
        L.yor_material_probe(C.byref(md), po.fptr(inp[i]), int(sfl[i]), C.byref(bf), po.fptr(e), C.byref(p), C.byref(so), po.fptr(s8))
        fl.append(bf.value); ev.extend(e.tolist()); pd.append(p.value); sfo.append(so.value); sm.extend(s8.tolist())
    assert fl == [int(v) for v in g[f"{name}_flags"]]
    check(variant, ev, g[f"{name}_eval3"], f"{name} eval")
    check(variant, pd, g[f"{name}_pdf"], f"{name} pdf")
    if variant == "ieee":
        assert sfo == [int(v) for v in g[f"{name}_sflags_out"]]
        check(variant, sm, g[f"{name}_sample8"], f"{name} sample")
    else:
        same = np.array(sfo) == g[f"{name}_sflags_out"]
        assert same.mean() > 0.98  # a lobe pick exactly on a threshold may flip under -ffast-math
        check(variant, np.array(sm, np.float32).reshape(-1, 8)[same], g[f"{name}_sample8"].reshape(-1, 8)[same], f"{name} sample")
    # Material::getSpecular (recursiveRaytrace's perfect reflection / filtered transmission) and getAlpha
    sf, sp12, al = [], [], []
    o12 = np.zeros(12, np.float32)
    for i in range(inp.shape[0]):
        f = C.c_int32(); a = C.c_float()
        L.yor_material_specular(C.byref(md), po.fptr(inp[i]), SPEC_RAYLEVEL.get(name, 1), C.byref(f), po.fptr(o12), C.byref(a))
        sf.append(f.value); sp12.extend(o12.tolist()); al.append(a.value)
    assert sf == [int(v) for v in g[f"{name}_specflags"]]
    check(variant, sp12, g[f"{name}_spec12"], f"{name} getSpecular")
    check(variant, al, g[f"{name}_alpha"], f"{name} getAlpha")
    tr = []
    t3 = np.zeros(3, np.float32)
    for i in range(inp.shape[0]):
        L.yor_material_transparency(C.byref(md), po.fptr(inp[i]), po.fptr(t3))
        tr.extend(t3.tolist())
    check(variant, tr, g[f"{name}_transp3"], f"{name} getTransparency")


@pytest.mark.parametrize("name", ["rg0", "rg1", "rg2"])
def test_rough_glass_two_direction_sample(gold, name):
    """the Material::sample overload recursiveRaytrace's glossy branch calls for a lobe that reflects AND transmits
    (integrator_montecarlo.cc:919-970; RoughGlassMaterial::sample, material_rough_glass.cc:165-286)"""
    variant, g = gold
    if variant == "fast" and name in FAST_SKIP:
        pytest.skip("see FAST_SKIP")
    L = po.lib()
    md = po.material_desc(MATERIALS[name])
    inp = f32(g[f"{name}_two_in14"]).reshape(-1, 14).copy()
    sfl = g[f"{name}_two_sflags_in"]
    out, sfo = [], []
    o15 = np.zeros(15, np.float32)
    for i in range(inp.shape[0]):
        so = C.c_int32()
        L.yor_material_sample_two(C.byref(md), po.fptr(inp[i]), int(sfl[i]), C.byref(so), po.fptr(o15))
        out.extend(o15.tolist()); sfo.append(so.value)
    assert sfo == [int(v) for v in g[f"{name}_two_sflags_out"]]
    check(variant, out, g[f"{name}_two_out15"], f"{name} two-direction sample")


def test_light_material(gold):
    variant, g = gold
    L = po.lib()
    md = po.material_desc({"type": "light_mat", "color": (1.0, 0.9, 0.8), "power": 17.5})
    inp = g["lm_in7"].reshape(-1, 7)
    out = np.zeros(3, np.float32)
    got = []
    for row in inp:
        fl = f32(row[:6]).copy()
        L.yor_lightmat_emit(C.byref(md), po.fptr(fl[0:3].copy()), po.fptr(fl[3:6].copy()), int(row[6]), po.fptr(out))
        got.extend(out.tolist())
    check(variant, got, g["lm_emit3"], "LightMaterial::emit")


@pytest.mark.container
def test_faure_tables_match_reference_file():
    """The oracle regenerates the Faure permutations and the 9-digit inverse primes instead of copying
    them; compare with the numbers in the reference's files (read as text, container only)."""
    import re
    L = po.lib()
    txt = open("/root/reference/src/common/faure_tables.cc").read()
    arrays = {int(m.group(1)): [int(v) for v in m.group(2).replace("\n", " ").split(",") if v.strip()]
              for m in re.finditer(r"int fp_(\d+)__\[\] = \{([^}]*)\}", txt)}
    order = re.search(r"faure__\[\] = \{([^}]*)\}", txt).group(1)
    names = [int(v) for v in re.findall(r"fp_(\d+)__", order)]
    assert len(names) == 51
    for dim in range(50):
        n = C.c_int()
        p = L.yor_faure_perm(dim, C.byref(n))
        assert [p[i] for i in range(n.value)] == arrays[names[dim]], dim


def test_whole_path_against_the_references_expected_png():
    """The only whole-path fixture the reference holds (tests/test01's expected render, see tests/png_fixture.py): the
    ORACLE's render of the shipped scene (textures stripped), pushed through the reference's output transform, against
    the PNG on every pixel whose filter footprint sees only untextured materials — 47 % of the frame.  This pins
    camera, traversal, direct lighting, shadow rays, film filter (gauss 1.5) and the sRGB / 8-bit output together.
    Observed: 83.6 % of those pixels equal, 97.2 % within one level, 99.67 % within two; the rest (205 pixels) sit on
    1-spp silhouette and shadow edges, where the reference image shows intermediate values (it was made by a later
    build, v3.1.1-beta per its badge, and test01.xml asks for the OpenCV denoiser)."""
    import os
    from tests import png_fixture, xml_scene
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test01_dl.xml")
    sc, rd = xml_scene.load(path)
    film, _ = po.OracleScene(sc).render(dict(rd, oracle_threads=4))
    st = png_fixture.compare(film, sc, rd, "oracle")
    assert st["fraction_of_frame"] > 0.45
    assert st["within_2"] >= 0.995 * st["pixels_compared"], st
    assert st["within_1"] >= 0.96 * st["pixels_compared"], st
    assert st["exact"] >= 0.80 * st["pixels_compared"], st


def test_glibc_rand_restatement_against_this_machines_libc():
    """The tile seeds of the Russian-roulette streams come from libc's rand() (integrator_tiled.cc:319), whose state the
    last Material / ObjectGeometric constructor set with srand() (material.cc:56, object_geom.cc:42).  glibc is a third
    party: its generator is restated (oracle yor_glibc_rand, host side libyafaray_amd) and pinned here against the
    libc this process runs on."""
    L = po.lib()
    L.yor_glibc_rand.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_int32)]
    try:
        libc = C.CDLL("libc.so.6")
    except OSError:
        pytest.skip("no glibc on this machine")
    for seed in (0, 1, 2, 3, 9, 74, 12345, 2 ** 31 - 1, 2 ** 31 + 5):
        out = (C.c_int32 * 400)()
        L.yor_glibc_rand(seed, 400, out)
        libc.srand(C.c_uint(seed))
        assert list(out) == [libc.rand() for _ in range(400)], f"seed {seed}"
