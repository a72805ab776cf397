"""Pins the CPU oracle's CONTROL FLOW against the reference's own integrator
sources compiled in this container (oracle/Makefile target `ref`, driver oracle/ref_harness/ref_integrator.cc, fixtures
tests/golden/ref_integrator_{ieee,fast}.json.gz made by tests/golden/make_golden.py).

The reference's TiledIntegrator::render -> renderPass -> renderTile -> PathIntegrator::integrate (or
DirectLightIntegrator::integrate) -> estimateAllDirectLight / estimateOneDirectLight -> doLightEstimation ->
recursiveRaytrace ran on the reference's real camera, materials, lights, QMC and per-tile Random; the geometry query
(brute force) and the film (a recorder) are the harness's.  Eight cases: path samples > 1 with four bounces and MIS on
glossy / Oren-Nayar / emitting surfaces; three lights with Russian roulette from the second bounce and a base sampling
offset; mirror + glass + transparency + the glossy-recursive branch at raydepth 3 with refraction alpha; three adaptive
passes; depth of field; the direct-lighting integrator; transparent shadows; no_recursive.

  ieee : reference built -O2 -ffp-contract=off -> every sample the oracle hands to addSample, every recorded ray query
         and both ray counts must match BIT FOR BIT, in order.
  fast : the reference's release flags (-O3 -ffast-math): a discrete decision that flips (a hit on a grazing ray, a
         roulette draw) forks the serial state of everything after it — the two reference builds part the same way
          — so the comparison is: close until the first fork, and close throughout where there is no serial state.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

from tests.integrator_fixture import case_scene, closest_rays, film_from_samples, load, samples

CASES = ["pt_mis_paths", "pt_three_lights_rr", "pt_recursive", "pt_multipass", "pt_dof", "directlighting",
         "pt_transparent_shadows", "pt_no_recursive", "pt_absorption_aniso", "pt_depth_bias_visibility", "dl_fake_shadows_flat", "pt_degenerate_lobes_clip", "pt_rough_glass",
         "pt_caustics_default", "pt_caustics_path_no_recursive", "dl_rough_glass"]
# one light and roulette off: every sample is a pure function of (pixel, sample index)
NO_SERIAL_STATE = {"pt_mis_paths", "pt_dof"}


@pytest.fixture(scope="module")
def docs():
    return {v: load(v) for v in ("ieee", "fast")}


def _case(doc, name):
    return next(c for c in doc["cases"] if c["name"] == name)


def _run(doc, cs, **override):
    sc, rd = case_scene(doc, cs)
    rd.update(override)
    osc = po.OracleScene(sc)
    n = len(cs["sample_xy"]) // 2
    out = osc.render_traced(rd, n + 16, len(cs["closest_tri"]))
    osc.close()
    return out


def test_fixture_holds_every_case(docs):
    for v in ("ieee", "fast"):
        assert [c["name"] for c in docs[v]["cases"]] == CASES


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_the_reference_integrators_bit_for_bit(docs, name):
    doc = docs["ieee"]
    cs = _case(doc, name)
    film, st, smp, rays, n_rays = _run(doc, cs)
    xy, dxdy, rgba = samples(cs)
    # the ray counts: every Scene::intersect / Scene::isShadowed call the reference made
    assert (st.rays_closest, st.rays_shadow) == (cs["n_closest"], cs["n_shadow"])
    assert n_rays == cs["n_closest"]
    # the samples, in addSample order: pixel, sub-pixel position, colour and alpha
    assert smp.shape[0] == xy.shape[0]
    assert np.array_equal(smp[:, :2].astype(np.int64), xy)
    assert np.array_equal(smp[:, 2:4].view(np.uint32), dxdy.view(np.uint32)), "sub-pixel positions (renderTile)"
    same = smp[:, 4:].view(np.uint32) == rgba.view(np.uint32)
    same |= (smp[:, 4:] == 0) & (rgba == 0)
    assert same.all(), f"{(~same.all(axis=1)).sum()} of {len(xy)} samples differ, first at {np.argmax(~same.all(axis=1))}"
    # the first closest-hit queries: origin, direction, tmin, tmax, answer
    want, want_tri = closest_rays(cs)
    assert rays.shape[0] == want.shape[0]
    assert np.array_equal(rays[:, :9].view(np.uint32), want.view(np.uint32)), "closest-hit queries"
    assert np.array_equal(rays[:, 9].view(np.int32), want_tri)
    # and the oracle's film is what addSample's box footprint makes of those samples
    assert np.array_equal(film, film_from_samples(doc, cs))


def test_tiles_are_handed_out_in_linear_order(docs):
    doc = docs["ieee"]
    for cs in doc["cases"]:
        t = np.asarray(cs["tiles4"]).reshape(-1, 4)
        n_pass = cs["render"].get("AA_passes", 1)
        rd = case_scene(doc, cs)[1]
        ts, w, h, x0, y0 = rd["tile_size"], rd["width"], rd["height"], rd.get("xstart", 0), rd.get("ystart", 0)
        one = [(x0 + x, y0 + y, min(ts, w - x), min(ts, h - y)) for y in range(0, h, ts) for x in range(0, w, ts)]
        assert [tuple(r) for r in t] == one * n_pass


@pytest.mark.parametrize("name", CASES)
def test_oracle_against_the_release_flag_build(docs, name):
    doc = docs["fast"]
    cs = _case(doc, name)
    film, st, smp, rays, n_rays = _run(doc, cs)
    xy, dxdy, rgba = samples(cs)
    assert smp.shape[0] == xy.shape[0] and np.array_equal(smp[:, :2].astype(np.int64), xy)
    np.testing.assert_allclose(smp[:, 2:4], dxdy, rtol=0, atol=1e-6)
    rel = (np.abs(smp[:, 4:] - rgba) / np.maximum(np.abs(rgba), 1e-3)).max(axis=1)
    off = rel > 1e-3
    if name in NO_SERIAL_STATE:
        assert off.sum() <= 0.01 * len(off), f"{off.sum()} of {len(off)} samples over 1e-3"
    else:
        # the oracle parts from the release-flag build exactly where the reference's own IEEE build does
        _, _, rgba_ieee = samples(_case(docs["ieee"], name))
        off_ref = (np.abs(rgba_ieee - rgba) / np.maximum(np.abs(rgba), 1e-3)).max(axis=1) > 1e-3
        first, first_ref = (int(np.argmax(o)) if o.any() else len(o) for o in (off, off_ref))
        assert first == first_ref, f"parts from the release build at sample {first}, the IEEE build of the reference at {first_ref}"
    assert abs(int(st.rays_closest) - cs["n_closest"]) <= 0.01 * cs["n_closest"]
    assert abs(int(st.rays_shadow) - cs["n_shadow"]) <= 0.03 * cs["n_shadow"]


def test_the_fixture_sees_the_serial_state(docs):
    """negative controls: the roulette stream and the light counter really are on these paths"""
    doc = docs["ieee"]
    cs = _case(doc, "pt_three_lights_rr")
    _, _, rgba = samples(cs)
    _, _, smp, _, _ = _run(doc, cs, rand_srand=cs["srand"] + 1)
    assert (smp[:, 4:] != rgba).any(), "another libc seed must change the roulette decisions"
    _, _, smp, _, _ = _run(doc, cs, adv_base_sampling_offset=0)
    assert (smp[:, 4:] != rgba).any(), "the base sampling offset feeds the light choice and every QMC index"
    cs = _case(doc, "pt_multipass")
    _, _, rgba = samples(cs)
    _, _, smp, _, _ = _run(doc, cs, AA_light_sample_multiplier_factor=1.0)
    n0 = doc["width"] * doc["height"] * cs["render"]["AA_minsamples"]
    assert np.array_equal(smp[:n0, 4:], rgba[:n0]) and (smp[n0:, 4:] != rgba[n0:]).any(), "later passes take more light samples"
