"""Host logic of the Interface-shaped API: state machine, strictly typed ParamMap, type-string factories,
XML loader — mirrored from the reference's behaviour (scene.cc:110-131, param.cc:49-53, the factory()
functions).  No GPU needed: nothing here launches a kernel."""
import os
import textwrap

import numpy as np
import pytest

from libyafaray_amd import Interface, YafaRayError, scenes


def fresh(strict=False):
    return Interface(strict=strict)


def test_geometry_state_machine():
    yi = fresh()
    assert not yi.startGeometry()                 # before startScene: wrong state (scene.cc:110-115)
    assert yi.startScene(0)
    assert not yi.startTriMesh(1, 3, 1, False)    # not inside geometry
    assert yi.startGeometry()
    assert not yi.startGeometry()
    assert yi.startTriMesh(1, 3, 1, False, False, 0)
    assert yi.addVertex(0, 0, 0) == 0 and yi.addVertex(1, 0, 0) == 1 and yi.addVertex(0, 1, 0) == 2
    yi.paramsClearAll(); yi.paramsSet({"type": "shinydiffusemat"})
    mat = yi.createMaterial("m")
    assert mat
    assert yi.addTriangle(0, 1, 2, mat)
    assert not yi.addTriangle(0, 1, 7, mat)       # index out of range
    assert "out of range" in yi.getLastError()
    assert not yi.endGeometry()                   # mesh still open
    assert yi.endTriMesh() and yi.endGeometry()
    assert not yi.startScene(1)                   # "universal" scenes are out of scope
    assert "triangle" in yi.getLastError()


def test_parammap_is_strictly_typed():
    """Parameter::getVal only matches the exact type (param.cc:49-53): a float given as int is ignored
    and the factory default applies — the trap SURVEY §5 warns about."""
    yi = fresh(strict=True)
    yi.startScene(0)
    yi.paramsClearAll()
    yi.paramsSet({"type": "pathtracing", "bounces": 7.0})       # wrong type: float instead of int
    yi.createIntegrator("a")
    yi.paramsClearAll()
    yi.paramsSet({"type": "pathtracing", "bounces": 7})
    yi.createIntegrator("b")
    # the effect is visible through validation: bounces > 12 is rejected only when it was actually read
    yi.paramsClearAll()
    yi.paramsSet({"type": "pathtracing", "bounces": 40.0})
    assert yi.createIntegrator("c")                              # 40.0 ignored -> default 3


def test_factories_fail_loudly_outside_scope():
    yi = fresh()
    yi.startScene(0)
    for kind, params, needle in [
        ("material", {"type": "blend_mat"}, "scope"), ("material", {"type": "glass", "dispersion_power": 0.2}, "dispersion"),
        ("material", {"type": "rough_glass", "dispersion_power": 0.2}, "dispersion"), ("material", {"type": "rough_glass", "roughness_shader": "n"}, "roughness_shader"),
        ("material", {"type": "rough_glass", "additionaldepth": 9}, "additionaldepth"), ("material", {"type": "shinydiffusemat", "wireframe_amount": 0.5}, "wireframe"),
        ("light", {"type": "spotlight"}, "scope"), ("light", {"type": "pointlight", "photon_only": True}, "photon_only"), ("camera", {"type": "orthographic"}, "scope"),
        ("background", {"type": "sunsky"}, "scope"), ("integrator", {"type": "photonmapping"}, "scope"),
        ("integrator", {"type": "pathtracing", "caustic_type": "photon"}, "photon"),
    ]:
        yi.paramsClearAll(); yi.paramsSet(params)
        r = getattr(yi, "create" + kind.capitalize())("x_" + needle)
        assert not r and needle in yi.getLastError(), (kind, params, yi.getLastError())
    yi.paramsClearAll(); yi.paramsSet({"type": "rough_glass", "IOR": 1.6, "alpha": 0.3, "transmit_filter": 0.5, "fake_shadows": True, "absorption": (0.5, 0.6, 0.7), "absorption_dist": 2.0})
    assert yi.createMaterial("rough"), yi.getLastError()
    yi.paramsClearAll()
    assert not yi.createMaterial("untyped") and "type" in yi.getLastError()
    strict = fresh(strict=True)
    strict.startScene(0)
    strict.paramsSet({"type": "blend_mat"})
    with pytest.raises(YafaRayError):
        strict.createMaterial("g")


def test_render_requires_names_and_supported_settings():
    yi = fresh()
    sc = scenes.cornell_soup(12, seed=1)
    scenes.load_scene(yi, sc, scenes.render_settings(8, 8, 1))
    yi.paramsSet({"AA_passes": 0})
    assert not yi.prepareRender() and "AA_passes" in yi.getLastError()
    yi.paramsSet({"AA_passes": 1, "premult": True})
    assert not yi.prepareRender() and "premult" in yi.getLastError()
    yi.paramsSet({"premult": False, "camera_name": "nope"})
    assert not yi.prepareRender() and "Camera" in yi.getLastError()


XML = """<?xml version="1.0"?>
<!-- a minimal scene in the reference's XML grammar (import_xml.cc) -->
<scene type="triangle">
<material name="white"><type sval="shinydiffusemat"/><color r="0.8" g="0.8" b="0.8" a="1"/><diffuse_reflect fval="1"/></material>
<material name="lamp"><type sval="light_mat"/><color r="1" g="1" b="1" a="1"/><power fval="10"/></material>
<light name="l0"><type sval="arealight"/><corner x="-0.2" y="-0.2" z="0.9"/><point1 x="-0.2" y="0.2" z="0.9"/>
  <point2 x="0.2" y="-0.2" z="0.9"/><color r="1" g="1" b="1" a="1"/><power fval="10"/><samples ival="1"/></light>
<camera name="cam"><type sval="perspective"/><from x="0" y="-3" z="0"/><to x="0" y="0" z="0"/><up x="0" y="-3" z="1"/>
  <resx ival="16"/><resy ival="16"/><focal fval="1.2"/></camera>
<background name="world_background"><type sval="constant"/><color r="0.1" g="0.2" b="0.3" a="1"/><power fval="1"/></background>
<integrator name="default"><type sval="pathtracing"/><bounces ival="2"/><path_samples ival="1"/>
  <russian_roulette_min_bounces ival="2"/><caustic_type sval="none"/></integrator>
<integrator name="volintegr"><type sval="none"/></integrator>
<mesh id="1" vertices="4" faces="2" has_orco="false" has_uv="false" type="0">
  <p x="-1" y="-1" z="-1"/><p x="1" y="-1" z="-1"/><p x="1" y="1" z="-1"/><p x="-1" y="1" z="-1"/>
  <set_material sval="white"/><f a="0" b="1" c="2"/><f a="0" b="2" c="3"/>
</mesh>
<render><camera_name sval="cam"/><integrator_name sval="default"/><volintegrator_name sval="volintegr"/>
  <background_name sval="world_background"/><width ival="16"/><height ival="16"/><AA_passes ival="1"/><AA_minsamples ival="2"/>
  <AA_pixelwidth fval="1"/><filter_type sval="box"/><tile_size ival="8"/><tiles_order sval="linear"/><threads ival="1"/></render>
</scene>
"""


def test_xml_loader_parses_reference_grammar(tmp_path):
    p = tmp_path / "scene.xml"
    p.write_text(XML)
    yi = fresh(strict=True)
    assert yi.loadXml(str(p))
    # the loader left the <render> ParamMap current: a second createCamera under the same name still works,
    # and a scene with a texture element is refused with a diagnostic
    bad = tmp_path / "bad.xml"
    bad.write_text(XML.replace('<material name="lamp">', '<texture name="t"><type sval="image"/></texture><material name="lamp">'))
    y2 = fresh(strict=False)
    assert not y2.loadXml(str(bad)) and "texture" in y2.getLastError()
    y3 = fresh(strict=False)
    assert not y3.loadXml(str(tmp_path / "missing.xml"))


def test_no_cpu_fallback_without_a_gpu():
    """The product path must fail loudly when no device is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    yi = fresh()
    scenes.load_scene(yi, scenes.cornell_soup(12, seed=1), scenes.render_settings(8, 8, 1))
    assert not yi.render()
    assert yi.getLastError() != ""


def _smooth_mesh_restated(points, tris, angle):
    """Independent restatement of Scene::smoothMesh (scene.cc:383-543) in float32 scalar arithmetic."""
    f = np.float32
    P = points.astype(np.float32)
    nt = len(tris)

    def sub(a, b): return np.array([f(a[0] - b[0]), f(a[1] - b[1]), f(a[2] - b[2])], np.float32)
    def cross(a, b): return np.array([f(f(a[1] * b[2]) - f(a[2] * b[1])), f(f(a[2] * b[0]) - f(a[0] * b[2])), f(f(a[0] * b[1]) - f(a[1] * b[0]))], np.float32)
    def dot(a, b): return f(f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2]))
    def length(a): return f(np.sqrt(dot(a, a)))
    def normalize(a):
        l = dot(a, a)
        if l != 0:
            inv = f(f(1.0) / f(np.sqrt(l)))
            a = np.array([f(a[0] * inv), f(a[1] * inv), f(a[2] * inv)], np.float32)
        return a
    def sin_from(a, b):
        div = f(f(f(length(a) * length(b)) * f(0.99999)) + f(0.00001))
        arg = f(f(length(cross(a, b)) / div) * f(0.99999))
        if arg > 1: arg = f(1)
        return f(np.arcsin(np.float64(arg)))
    fn = np.array([normalize(cross(sub(P[t[1]], P[t[0]]), sub(P[t[2]], P[t[0]]))) for t in tris], np.float32)
    orders = [(0, 1, 2), (1, 0, 2), (2, 0, 1)]
    alpha = np.array([[sin_from(sub(P[t[o[1]]], P[t[o[0]]]), sub(P[t[o[2]]], P[t[o[0]]])) for o in orders] for t in tris], np.float32)
    out = np.zeros((nt, 3, 3), np.float32)
    if angle >= 180:
        vn = np.zeros_like(P)
        for ti, t in enumerate(tris):
            for k in range(3):
                for c in range(3):
                    vn[t[k], c] = f(vn[t[k], c] + f(fn[ti, c] * alpha[ti, k]))
        vn = np.array([normalize(v) for v in vn], np.float32)
        for ti, t in enumerate(tris):
            for k in range(3):
                out[ti, k] = vn[t[k]]
        return out
    if not angle > 0.1:
        return out
    from oracle import pyoracle as po
    thresh = f(po.lib().yor_fcos(f(np.float64(f(angle)) * 0.01745329251994329576922)))
    vface = [[] for _ in P]
    for ti, t in enumerate(tris):
        for k in range(3):
            vface[t[k]].append((ti, alpha[ti, k]))
    for i, lst in enumerate(vface):
        found = []
        for (fi, a_j) in lst:
            vnorm = np.array([f(fn[fi, c] * a_j) for c in range(3)], np.float32)
            smooth = False
            for (f2, a_k) in lst:
                if f2 == fi:
                    continue
                if dot(fn[fi], fn[f2]) > thresh:
                    smooth = True
                    vnorm = np.array([f(vnorm[c] + f(fn[f2, c] * a_k)) for c in range(3)], np.float32)
            if not smooth:
                continue
            vnorm = normalize(vnorm)
            use = None
            for v in found:
                if np.float64(dot(vnorm, v)) > 0.999:
                    use = v
                    break
            if use is None:
                found.append(vnorm)
                use = vnorm
            corner = list(tris[fi]).index(i)
            out[fi, corner] = use
    return out


@pytest.mark.parametrize("angle", [181.0, 60.0, 25.0, 0.05])
def test_smooth_mesh(angle):
    """Interface::smoothMesh on a faceted cylinder + cap: the C++ host code against the Python restatement"""
    from oracle import pyoracle as po
    po.lib().yor_fcos.restype = __import__("ctypes").c_float
    po.lib().yor_fcos.argtypes = [__import__("ctypes").c_float]
    n = 10
    ring0 = [(np.cos(2 * np.pi * k / n), np.sin(2 * np.pi * k / n), 0.0) for k in range(n)]
    ring1 = [(np.cos(2 * np.pi * k / n), np.sin(2 * np.pi * k / n), 1.3) for k in range(n)]
    pts = np.array(ring0 + ring1 + [(0.0, 0.0, 1.3)], np.float32)
    tris = []
    for k in range(n):
        a, b = k, (k + 1) % n
        tris += [(a, b, n + a), (b, n + b, n + a), (n + a, n + b, 2 * n)]
    yi = fresh()
    yi.startScene(0)
    yi.paramsClearAll(); yi.paramsSet({"type": "shinydiffusemat"})
    mat = yi.createMaterial("m")
    yi.startGeometry()
    mid = yi.getNextFreeId()
    yi.startTriMesh(mid, len(pts), len(tris), False, False, 0)
    for p in pts:
        yi.addVertex(float(p[0]), float(p[1]), float(p[2]))
    for t in tris:
        yi.addTriangle(t[0], t[1], t[2], mat)
    yi.endTriMesh()
    assert yi.smoothMesh(0, angle)
    got = yi.getMeshCornerNormals(mid, len(tris))
    want = _smooth_mesh_restated(pts, tris, angle)
    assert np.array_equal((np.abs(got).sum(axis=-1) == 0), (np.abs(want).sum(axis=-1) == 0)), "which corners keep the geometric normal"
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
    if angle >= 180:
        assert np.all(np.abs(np.linalg.norm(got, axis=-1) - 1) < 1e-6)
    if angle == 25.0:     # 36 degrees between the cylinder's facets: nothing is smoothed across them, but each quad's two triangles are
        assert (np.abs(got).sum(axis=-1) == 0).any() and (np.abs(got).sum(axis=-1) != 0).any()
    assert yi.endGeometry()


def test_every_interface_method_has_a_function_and_refusals_are_loud():
    """The rest of yafaray4::Interface's surface (include/interface/interface.h:62-128): exporters call several of these
    unconditionally.  Honoured where the path is concerned, accepted where only logging / decoration is, refused with a
    getLastError() diagnostic where the request lies outside the scope — never silently dropped."""
    yi = fresh()
    assert yi.startScene(0)
    # refusals: each leaves a diagnostic
    for call, needle in [
        (lambda: yi.createObject("sphere1"), "createObject"), (lambda: yi.createVolumeRegion("v"), "createVolumeRegion"),
        (lambda: yi.createImageHandler("ih"), "createImageHandler"), (lambda: yi.startCurveMesh(5, 10), "startCurveMesh"),
        (lambda: yi.endCurveMesh(None, 0.1, 0.1, 0.0), "endCurveMesh"), (lambda: yi.addInstance(1, np.eye(4)), "addInstance"),
    ]:
        assert not call()
        assert needle in yi.getLastError(), yi.getLastError()
    # render passes: the combined pass alone is fine, anything else is refused
    yi.paramsClearAll(); yi.paramsSet({"pass_enable": False, "pass_Depth": "z-depth-norm"})
    assert yi.setupRenderPasses()
    yi.paramsClearAll(); yi.paramsSet({"pass_enable": True, "pass_Depth": "z-depth-norm", "pass_AO": "disabled"})
    assert not yi.setupRenderPasses() and "pass_Depth" in yi.getLastError()
    yi.paramsClearAll(); yi.paramsSet({"pass_enable": True, "pass_Depth": "disabled"})
    assert yi.setupRenderPasses()
    # accepted
    yi.paramsClearAll(); yi.paramsSet({"logging_paramsBadgePosition": "top", "logging_title": "t"})
    assert yi.setLoggingAndBadgeSettings()
    assert yi.setInteractive(True)
    yi.setConsoleVerbosityLevel("verbose"); yi.setLogVerbosityLevel("debug"); yi.setParamsBadgePosition("bottom")
    assert yi.getDrawParams() is False
    yi.printInfo("info line")
    # getRenderParameters: the current ParamMap
    yi.paramsClearAll(); yi.paramsSet({"width": 17, "filter_type": "gauss", "AA_pixelwidth": 1.5})
    yi.paramsSetMatrix("transform", np.arange(16, dtype=np.float32).reshape(4, 4), transpose=True)
    rp = yi.getRenderParameters()
    assert rp["width"] == "17" and rp["filter_type"] == "gauss" and float(rp["AA_pixelwidth"]) == 1.5
    assert [float(x) for x in rp["transform"].split()] == list(np.arange(16, dtype=np.float32).reshape(4, 4).T.reshape(-1))


def test_uv_orco_geometry_and_mesh_ptr():
    """startTriMeshPtr (the scene picks the id), addVertex with orco, addUv, the UV overload of addTriangle — the geometry
    calls of a textured export (interface.h:63,67,70,71)."""
    yi = fresh()
    yi.startScene(0)
    yi.paramsClearAll(); yi.paramsSet({"type": "shinydiffusemat"})
    mat = yi.createMaterial("m")
    yi.startGeometry()
    mid = yi.startTriMeshPtr(3, 1, True, True)
    assert mid >= 1
    assert yi.addVertex(0, 0, 0) == 0                     # plain vertices stay legal on an orco mesh
    assert yi.addVertexWithOrco(1, 0, 0, 0.5, 0, 0) == 1 and yi.addVertexWithOrco(0, 1, 0, 0, 0.5, 0) == 2
    assert yi.addUv(0.0, 0.0) == 0 and yi.addUv(1.0, 0.0) == 1 and yi.addUv(0.0, 1.0) == 2
    assert yi.addTriangleWithUv(0, 1, 2, 0, 1, 2, mat)
    assert yi.endTriMesh()
    # the reference never range-checks UV offsets when a face is added (exporters may list the UVs after the faces,
    # scene.cc:652-686); the mesh is checked as a whole when it is closed — bad offsets and the reference's own
    # "UV-offsets mismatch!" (scene.cc:319-326)
    assert yi.startTriMesh(yi.getNextFreeId(), 3, 1, False, True)
    for v in ((0, 0, 0), (1, 0, 0), (0, 1, 0)):
        yi.addVertex(*v)
    assert yi.addTriangleWithUv(0, 1, 2, 0, 1, 2, mat)                  # UVs still to come
    assert yi.addUv(0.0, 0.0) == 0 and yi.addUv(1.0, 0.0) == 1
    assert not yi.endTriMesh() and "UV index" in yi.getLastError()
    assert yi.addUv(0.0, 1.0) == 2
    assert yi.addTriangle(0, 1, 2, mat)                                 # a face without UVs on a UV mesh
    assert not yi.endTriMesh() and "mismatch" in yi.getLastError()
    assert yi.addTriangleWithUv(0, 1, 2, 2, 1, 0, mat)                  # (the counts now differ for good: 3 faces, 2 with UVs)
    assert not yi.endTriMesh()
    yi.clearAll(); yi.startScene(0)
    yi.paramsClearAll(); yi.paramsSet({"type": "shinydiffusemat"})
    mat = yi.createMaterial("m")
    yi.startGeometry()
    assert yi.startTriMesh(yi.getNextFreeId(), 3, 1, False, False)
    assert yi.addVertexWithOrco(0, 0, 0, 0, 0, 0) == -1 and "orco" in yi.getLastError()
    assert not yi.addTriangleWithUv(0, 0, 0, 0, 0, 0, mat) and "UV" in yi.getLastError()
    assert yi.endTriMesh() and yi.endGeometry()


def test_input_color_space_converts_colours_on_entry():
    """Interface::setInputColorSpace + paramsSetColor (interface.cc:247-252,292-301): colours are converted to linear RGB
    when they are set — sRGB by the piecewise curve with the polynomial pow, XYZ by the D65 matrix, raw by the gamma."""
    from oracle import pyoracle as po
    import ctypes as C
    L = po.lib()
    yi = fresh()
    yi.startScene(0)
    def colour_of(name):
        return [float(x) for x in yi.getRenderParameters()[name].split()]
    yi.paramsClearAll()
    yi.paramsSetColor("a", 0.5, 0.02, 1.0, 0.7)                          # default: raw, gamma 1 -> unchanged
    assert colour_of("a") == [0.5, np.float32(0.02), 1.0, np.float32(0.7)]
    yi.setInputColorSpace("sRGB", 1.0)
    yi.paramsSetColor("b", 0.5, 0.02, 1.0, 0.7)
    want = [float(L.yor_fpow(C.c_float((np.float32(0.5) + np.float32(0.055)) / np.float32(1.055)), C.c_float(2.4))), float(np.float32(0.02) / np.float32(12.92)),
            float(L.yor_fpow(C.c_float((np.float32(1.0) + np.float32(0.055)) / np.float32(1.055)), C.c_float(2.4))), float(np.float32(0.7))]
    np.testing.assert_allclose(colour_of("b"), want, rtol=1e-7)
    yi.setInputColorSpace("XYZ", 1.0)
    yi.paramsSetColor("c", 0.2, 0.3, 0.4)
    m = np.array([[3.2406255, -1.537208, -0.4986286], [-0.9689307, 1.8757561, 0.0415175], [0.0557101, -0.2040211, 1.0569959]], np.float32)
    np.testing.assert_allclose(colour_of("c")[:3], m @ np.array([0.2, 0.3, 0.4], np.float32), rtol=2e-6)
    yi.setInputColorSpace("Raw_Manual_Gamma", 2.2)
    yi.paramsSetColor("d", 0.25, 0.5, 0.75)
    np.testing.assert_allclose(colour_of("d")[:3], [float(L.yor_fpow(C.c_float(v), C.c_float(2.2))) for v in (0.25, 0.5, 0.75)], rtol=1e-7)
    yi.setInputColorSpace("LinearRGB", 2.2)
    yi.paramsSetColor("e", 0.25, 0.5, 0.75)
    assert colour_of("e")[:3] == [0.25, 0.5, 0.75]
