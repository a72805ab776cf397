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


# ---- crafted files: headers that lie about the image they hold (SIGFPE on a 16-bit palette PNG, 34 GB allocations from
# a 65535 x 65535 header, std::length_error out of the HDR reader's atoi'd size, TGA allocating before its truncation check) ----------
CRAFTED_CHILD = r'''
import os, struct, sys, zlib
sys.path.insert(0, %(root)r)
from libyafaray_amd import Interface

def png(w, h, color_type, depth, idat=b"", plte=None):
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if plte is not None:
        out += chunk(b"PLTE", plte)
    return out + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

def tga(w, h, image_type, depth, desc, body):
    return bytes([0, 0, image_type, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, depth, desc]) + body

files = {
    "pal16.png": png(4, 4, 3, 16, zlib.compress(b"\0" * (4 * (1 + 8))), plte=b"\1\2\3" * 4),
    "pal16_big.png": png(300, 300, 3, 16, zlib.compress(b"\0" * 1000), plte=b"\1\2\3"),
    "huge.png": png(65535, 65535, 6, 16, zlib.compress(b"\0" * 64)),
    "huge8.png": png(60000, 60000, 2, 8, zlib.compress(b"\0" * 64)),
    "lying.png": png(4000, 4000, 2, 8, zlib.compress(b"\0" * 4000)),
    "huge.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n" + b"\2\2\0\10" * 8,
    "huge2.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 30000 +X 30000\n" + b"\1\1\1\1" * 64,
    "neg.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y -5 +X 70000\n",
    "huge_raw.tga": tga(65535, 65535, 2, 32, 8, b"\0" * 64),
    "huge_rle.tga": tga(65535, 65535, 10, 24, 0, b"\xff\1\2\3" * 16),
    "trunc.tga": tga(64, 64, 2, 24, 0, b"\7" * (64 * 64 * 3 - 5)),
    "header_only.tga": tga(16, 16, 2, 24, 0, b"")[:17],
}
tmp = sys.argv[1]
n = 0
for name, data in files.items():
    path = os.path.join(tmp, name)
    open(path, "wb").write(data)
    yi = Interface(strict=False)
    yi.startScene(0)
    yi.paramsClearAll()
    yi.paramsSet({"type": "image", "filename": path})
    t = yi.createTexture("t")
    err = yi.getLastError()
    assert not t, name + " was accepted"
    assert err, name + ": no diagnostic"
    print(name, "->", err[:90])
    n += 1
print("refused", n)
'''


def test_crafted_image_files_are_refused_before_anything_is_allocated(tmp_path):
    import resource
    import subprocess
    import sys
    root = os.path.dirname(HERE)

    def limit():        # a decoder that allocates what a lying header asks for dies here instead of being refused
        resource.setrlimit(resource.RLIMIT_AS, (12 << 30, 12 << 30))
    r = subprocess.run([sys.executable, "-c", CRAFTED_CHILD % {"root": root}, str(tmp_path)], capture_output=True, text=True, timeout=300, preexec_fn=limit)
    assert r.returncode == 0 and "refused 12" in r.stdout, f"rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}"
    # the same files against the host sources built under AddressSanitizer + UBSan (tests/asan; CPU build)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan) or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        return
    subprocess.run(["bash", os.path.join(root, "tests", "asan", "build.sh")], check=True, timeout=900, capture_output=True)
    env = dict(os.environ, YAFARAY_LIBRARY=os.path.join(root, "tests", "asan", "libyafaray_host_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CRAFTED_CHILD % {"root": root}, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "refused 12" in r.stdout and "runtime error" not in r.stderr, f"rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
