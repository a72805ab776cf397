"""SURVEY row N2 — the host's image-file decoders (libyafaray_amd/csrc/yafaray_image.cpp) behind yafaray_createTexture, against the
reference's own handlers compiled from its sources (oracle/ref_harness/ref_textures.cc sec_files; fixtures
tests/golden/ref_textures_ieee.json.gz): TgaHandler and HdrHandler on the reference's test01 texture files (committed under
tests/golden/ as data), every texel's sum in double and every 97th texel bit for bit — decode, colour-space linearisation and
the image buffer's storage format (10-bit "optimized" / float).  PNG has no reference-side golden (the harness has no libpng):
the C++ decoder is checked against PIL's PNG reader, and its linearisation against the TGA path's.
No GPU needed."""
import gzip
import json
import os

import numpy as np
import pytest

from libyafaray_amd import Interface

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def golden():
    with gzip.open(os.path.join(GOLD, "ref_textures_ieee.json.gz"), "rt") as f:
        return json.load(f)


def decode(filename, **params):
    yi = Interface(strict=True)
    yi.startScene(0)
    yi.paramsClearAll()
    yi.paramsSet(dict({"type": "image", "filename": os.path.join(GOLD, filename)}, **params))
    assert yi.createTexture("t")
    return yi.getTextureImage("t")


CASES = {
    "file_tga_srgb_optimized": ("test01_tex.tga", dict(color_space="sRGB", texture_optimization="optimized")),
    "file_tga_linear_none": ("test01_tex.tga", dict(color_space="LinearRGB", texture_optimization="none")),
    "file_hdr": ("test01_tex.hdr", dict(color_space="sRGB", texture_optimization="optimized")),   # HDR forces LinearRGB / none (texture_image.cc:600-606)
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_decoder_against_the_references_handler(name):
    g = golden()
    fn, params = CASES[name]
    px = decode(fn, **params)
    meta = np.array(g[name + "_meta"], dtype=np.uint32)
    w, h = int(meta[0]), int(meta[1])
    assert px.shape == (h, w, 4)
    want_sum = meta[2:10].view(np.float64)
    got_sum = np.cumsum(px.reshape(-1, 4).astype(np.float64), axis=0)[-1]       # the harness adds texel by texel in double, row major
    s = np.array(g[name + "_samples"], dtype=np.uint32).reshape(-1, 6)
    xs, ys = s[:, 0].astype(int), s[:, 1].astype(int)
    got = px[ys, xs].view(np.uint32)
    bad = np.nonzero((got != s[:, 2:6]).any(axis=1))[0]
    assert len(bad) == 0, f"{name}: {len(bad)} of {len(s)} sampled texels differ, first ({xs[bad[0]]}, {ys[bad[0]]}): {px[ys[bad[0]], xs[bad[0]]]} vs {s[bad[0], 2:6].view(np.float32)}"
    assert np.array_equal(got_sum.view(np.uint64), want_sum.view(np.uint64)), f"{name}: channel sums {got_sum} vs {want_sum}"


def test_png_decoder_against_pil():
    from PIL import Image
    raw = np.array(Image.open(os.path.join(GOLD, "test01_tex.png")))     # (h, w, channels) uint8
    if raw.ndim == 2:
        raw = np.repeat(raw[..., None], 3, axis=2)
    px = decode("test01_tex.png", color_space="LinearRGB", texture_optimization="none")
    assert px.shape[:2] == raw.shape[:2]
    # PngHandler::loadFromFile: 8-bit channels * (1 / 255) in float (imagehandler_png.cc), LinearRGB leaves them alone
    want = raw[..., :3].astype(np.float32) * np.float32(1.0 / 255.0)
    np.testing.assert_array_equal(px[..., :3], want)
    if raw.shape[2] == 4:
        np.testing.assert_array_equal(px[..., 3], raw[..., 3].astype(np.float32) * np.float32(1.0 / 255.0))
    else:
        assert (px[..., 3] == 1.0).all()


def test_unknown_formats_and_missing_files_are_refused():
    yi = Interface(strict=False)
    yi.startScene(0)
    for fn, needle in [("test01_tex.jpg", "no decoder"), ("nope.tga", "cannot"), ("test01_tex.tif", "no decoder")]:
        yi.paramsClearAll()
        yi.paramsSet({"type": "image", "filename": os.path.join(GOLD, fn)})
        assert not yi.createTexture("x_" + fn)
        assert needle in yi.getLastError(), yi.getLastError()
    yi.paramsClearAll()
    yi.paramsSet({"type": "clouds"})
    assert not yi.createTexture("c")
    assert "scope" in yi.getLastError()
    yi.paramsClearAll()
    yi.paramsSet({"type": "image", "filename": os.path.join(GOLD, "test01_tex.png"), "interpolate": "mipmap_ewa"})
    assert not yi.createTexture("e")
    assert "mipmap_ewa" in yi.getLastError()
