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
        ("material", {"type": "glass"}, "scope"), ("material", {"type": "shinydiffusemat", "wireframe_amount": 0.5}, "wireframe"),
        ("material", {"type": "glossy", "anisotropic": True}, "anisotropic"),
        ("light", {"type": "spotlight"}, "scope"), ("camera", {"type": "orthographic"}, "scope"),
        ("background", {"type": "sunsky"}, "scope"), ("integrator", {"type": "photonmapping"}, "scope"),
        ("integrator", {"type": "pathtracing", "transpShad": True}, "transparent shadows"),
        ("integrator", {"type": "pathtracing", "caustic_type": "photon"}, "photon"),
    ]:
        yi.paramsClearAll(); yi.paramsSet(params)
        r = getattr(yi, "create" + kind.capitalize())("x_" + needle)
        assert not r and needle in yi.getLastError(), (kind, params, yi.getLastError())
    yi.paramsClearAll()
    assert not yi.createMaterial("untyped") and "type" in yi.getLastError()
    strict = fresh(strict=True)
    strict.startScene(0)
    strict.paramsSet({"type": "glass"})
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
