"""The device path against the reference's OWN integrators (tests/golden/ref_integrator_ieee.json.gz: the reference's
TiledIntegrator::render / renderTile / PathIntegrator::integrate / doLightEstimation / recursiveRaytrace compiled from
/root/reference and run on the harness's scenes — see tests/integrator_fixture.py and oracle/ref_harness/ref_integrator.cc).

Each case goes through the C ABI like any exporter's scene, renders on the GPU, and must reproduce (a) the number of
Scene::intersect and Scene::isShadowed calls the reference made and (b) the film ImageFilm::addSample's box footprint
makes of the samples the reference produced — bit for bit where the device adds a pixel's samples in the reference's
order (every pixel of a one-pass render), within 1e-6 otherwise.  No oracle in this comparison."""
import numpy as np
import pytest

from libyafaray_amd import Interface, scenes
from tests.integrator_fixture import case_scene, film_from_samples, load

pytestmark = pytest.mark.gpu

CASES = ["pt_mis_paths", "pt_three_lights_rr", "pt_recursive", "pt_multipass", "pt_dof", "directlighting",
         "pt_transparent_shadows", "pt_no_recursive", "pt_absorption_aniso", "pt_depth_bias_visibility", "dl_fake_shadows_flat", "pt_degenerate_lobes_clip", "pt_rough_glass",
         "pt_caustics_default", "pt_caustics_path_no_recursive", "dl_rough_glass"]


@pytest.fixture(scope="module")
def doc():
    return load("ieee")


@pytest.mark.parametrize("name", CASES)
def test_device_equals_the_reference_integrators(doc, name):
    cs = next(c for c in doc["cases"] if c["name"] == name)
    sc, rd = case_scene(doc, cs)
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.setRandState(cs["srand"], 0)         # the harness called srand(cs["srand"]) right before render()
    yi.render()
    film = yi.getFilm(rd["width"], rd["height"])
    st = yi.getRenderStats()
    want = film_from_samples(doc, cs)
    assert (st.rays_closest, st.rays_shadow) == (cs["n_closest"], cs["n_shadow"]), "ray counts differ from the reference's"
    assert np.array_equal(film[..., 4], want[..., 4]), "film weights"
    exact = (film.view(np.uint32) == want.view(np.uint32)) | ((film == 0) & (want == 0))
    frac = float(exact.all(axis=-1).mean())
    err = np.abs(film[..., :4] - want[..., :4]) / np.maximum(np.abs(want[..., :4]), 1e-3)
    print(f"{name}: bit-exact pixels {frac:.4f}, max rel {err.max():.3g}")
    assert err.max() <= 1e-6, f"{name}: max rel {err.max():.3g}"
    # transparent shadows: the filter product's last bit follows the order occluders are met in, i.e. tree topology
    assert frac >= (0.7 if cs["integrator"].get("transpShad") else 0.99), f"{name}: only {frac:.4f} of the pixels bit-exact"
