"""Random call sequences against the C API (CPU only, in a child process): whatever an exporter does in whatever order,
the library answers with a return value and yafaray_getLastError, never with a crash."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import sys, random
    sys.path.insert(0, %(root)r)
    import numpy as np
    from libyafaray_amd import Interface
    seed0, n_seq = int(sys.argv[1]), int(sys.argv[2])
    names = ["type", "color", "from", "to", "up", "resx", "resy", "focal", "power", "samples", "corner", "point1", "point2", "width", "height",
             "AA_passes", "AA_minsamples", "camera_name", "integrator_name", "bounces", "path_samples", "IOR", "exponent", "transparency",
             "filter_type", "AA_pixelwidth", "raydepth", "absorption", "visibility", "aperture", "bokeh_type", "tile_size", "xstart",
             "element", "name", "texture", "input", "upper_layer", "input1", "input2", "factor", "diffuse_shader", "mirror_shader", "IOR_shader",
             "texco", "mapping", "mode", "filename", "interpolate", "clipping", "xrepeat", "cropmin_x", "color1", "color2", "value", "bump_shader",
             "alpha", "transmit_filter", "fake_shadows", "roughness_shader", "caustic_type", "photon_only", "dispersion_power"]
    strings = ["shinydiffusemat", "glossy", "glass", "rough_glass", "mirror", "light_mat", "coated_glossy", "arealight", "pointlight", "perspective", "pathtracing", "path", "both",
               "directlighting", "constant", "none", "box", "gauss", "nonsense", "", "cam", "default", "blend_mat", "photonmapping", "sunlight",
               "shader_node", "texture_mapper", "layer", "mix", "value", "image", "t0", "n0", "n1", "uv", "orco", "cube", "sphere", "checker",
               %(root)r + "/tests/golden/test01_tex.png", "bilinear", "mipmap_ewa"]
    for seq in range(n_seq):
        rng = random.Random(seed0 + seq)
        yi = Interface()
        handles = [None]
        for step in range(rng.randint(5, 120)):
            op = rng.randrange(32)
            try:
                if op == 0: yi.startScene(rng.choice([0, 0, 1, -3]))
                elif op == 1: yi.startGeometry()
                elif op == 2: yi.endGeometry()
                elif op == 3: yi.startTriMesh(rng.choice([yi.getNextFreeId(), 0, 7, -1]), rng.choice([0, 3, 30, -5]), rng.choice([0, 1, 10, -2]), rng.random() < 0.2, rng.random() < 0.2, rng.choice([0, 0, 5]))
                elif op == 4: yi.endTriMesh()
                elif op == 5: yi.addVertex(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.choice([rng.uniform(-2, 2), float("nan"), float("inf"), 1e30]))
                elif op == 6: yi.addNormal(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1))
                elif op == 7: yi.addTriangle(rng.randint(-2, 40), rng.randint(-2, 40), rng.randint(-2, 40), rng.choice(handles))
                elif op == 8: yi.smoothMesh(rng.choice([0, 1, 7, -1]), rng.choice([0.0, 30.0, 181.0, -5.0]))
                elif op == 9: yi.paramsSetString(rng.choice(names), rng.choice(strings))
                elif op == 10: yi.paramsSetInt(rng.choice(names), rng.choice([0, 1, 2, 64, -1, 2**31 - 1, 100000]))
                elif op == 11: yi.paramsSetFloat(rng.choice(names), rng.choice([0.0, 1.0, -1.0, 1e30, float("nan"), 0.5]))
                elif op == 12: yi.paramsSetColor(rng.choice(names), rng.random(), rng.random(), rng.random())
                elif op == 13: yi.paramsSetPoint(rng.choice(names), rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-3, 3))
                elif op == 14: yi.paramsSetBool(rng.choice(names), rng.random() < 0.5)
                elif op == 15: yi.paramsClearAll()
                elif op == 16: handles.append(yi.createMaterial(rng.choice(["m0", "m1", ""])))
                elif op == 17: yi.createLight(rng.choice(["l0", "l1", ""]))
                elif op == 18: yi.createCamera(rng.choice(["cam", "c2", ""]))
                elif op == 19: yi.createBackground("world_background")
                elif op == 20: yi.createIntegrator(rng.choice(["default", "volintegr", "x"]))
                elif op == 21: yi.paramsStartList(); yi.paramsPushList(); yi.paramsEndList()
                elif op == 22: yi.clearAll()
                elif op == 23: yi.setShard(rng.randint(-1, 3), rng.randint(-1, 3))
                elif op == 24: yi.getRenderSize(); yi.getLastError(); yi.getVersion()
                elif op == 26: yi.createTexture(rng.choice(["t0", "t1", ""]))
                elif op == 27: yi.createTextureFromMemory(rng.choice(["t0", "t1"]), np.random.default_rng(seq).uniform(0, 1, (rng.choice([1, 3]), rng.choice([1, 4]), 4)).astype(np.float32))
                elif op == 28: yi.addVertexWithOrco(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1))
                elif op == 29: yi.addUv(rng.uniform(-2, 2), rng.choice([rng.uniform(-2, 2), float("nan")]))
                elif op == 30: yi.addTriangleWithUv(rng.randint(-2, 40), rng.randint(-2, 40), rng.randint(-2, 40), rng.randint(-2, 40), rng.randint(-2, 40), rng.randint(-2, 40), rng.choice(handles))
                elif op == 31: yi.paramsPushList(); yi.paramsSetString("element", "shader_node"); yi.paramsSetString("name", rng.choice(["n0", "n1", "n2"])); yi.paramsSetString("type", rng.choice(["layer", "mix", "value", "texture_mapper", "x"])); yi.paramsSetString(rng.choice(["input", "input1", "upper_layer", "factor", "texture"]), rng.choice(["n0", "n1", "n2", "t0"])); yi.paramsEndList() if rng.random() < 0.8 else None
                elif op == 25:
                    v = np.random.default_rng(seq).uniform(-1, 1, (rng.choice([0, 1, 5]), 3, 3)).astype(np.float32)
                    yi.addTriangles(v, None, rng.choice(handles)) if hasattr(yi, "addTriangles") else None
            except Exception:
                pass              # an error return surfaced as a Python exception: fine
        try:
            yi.close()
        except Exception:
            pass
    # mutated valid scripts: a correct export with calls dropped, doubled and swapped
    def script():
        return [("startScene", 0),
                ("paramsClearAll",), ("paramsSetString", "type", "shinydiffusemat"), ("paramsSetColor", "color", 0.8, 0.7, 0.6), ("createMaterial", "m0"),
                ("paramsClearAll",), ("paramsSetString", "type", "glass"), ("paramsSetFloat", "IOR", 1.5), ("createMaterial", "m1"),
                ("paramsClearAll",), ("paramsSetString", "type", "pointlight"), ("paramsSetPoint", "from", 0.0, 0.0, 0.9), ("paramsSetFloat", "power", 3.0), ("createLight", "l0"),
                ("paramsClearAll",), ("paramsSetString", "type", "perspective"), ("paramsSetPoint", "from", 0.0, -3.0, 0.0), ("paramsSetPoint", "to", 0.0, 0.0, 0.0),
                ("paramsSetPoint", "up", 0.0, -3.0, 1.0), ("paramsSetInt", "resx", 16), ("paramsSetInt", "resy", 12), ("createCamera", "cam"),
                ("paramsClearAll",), ("paramsSetString", "type", "pathtracing"), ("paramsSetInt", "bounces", 2), ("createIntegrator", "default"),
                ("paramsClearAll",), ("paramsSetString", "type", "none"), ("createIntegrator", "volintegr"),
                ("startGeometry",), ("startTriMesh", 1, 4, 2, False, False, 0), ("addVertex", -1.0, 0.0, -1.0), ("addVertex", 1.0, 0.0, -1.0),
                ("addVertex", 1.0, 0.0, 1.0), ("addVertex", -1.0, 0.0, 1.0), ("addTriangle", 0, 1, 2, "m0"), ("addTriangle", 0, 2, 3, "m1"), ("endTriMesh",),
                ("smoothMesh", 1, 60.0), ("endGeometry",),
                ("paramsClearAll",), ("paramsSetString", "camera_name", "cam"), ("paramsSetString", "integrator_name", "default"),
                ("paramsSetString", "volintegrator_name", "volintegr"), ("paramsSetInt", "width", 16), ("paramsSetInt", "height", 12),
                ("paramsSetInt", "AA_minsamples", 2), ("prepareRender",), ("getRenderSize",), ("render",)]
    done = 0
    for seq in range(n_seq):
        rng = random.Random(77 + seed0 + seq)
        calls = script()
        for _ in range(rng.randint(0, 6)):
            k = rng.randrange(3)
            i = rng.randrange(len(calls))
            if k == 0: del calls[i]
            elif k == 1: calls.insert(rng.randrange(len(calls)), calls[i])
            else:
                j = rng.randrange(len(calls)); calls[i], calls[j] = calls[j], calls[i]
        yi = Interface(); mats = {}
        for c in calls:
            try:
                if c[0] == "createMaterial": mats[c[1]] = yi.createMaterial(c[1])
                elif c[0] == "addTriangle": yi.addTriangle(c[1], c[2], c[3], mats.get(c[4]))
                else: getattr(yi, c[0])(*c[1:])
                done += 1
            except Exception:
                pass
        try: yi.close()
        except Exception: pass
    print("survived", n_seq, "calls that succeeded", done)
''')


def test_random_call_sequences_never_crash():
    for seed0 in (0, 10_000):
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(seed0), "150"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count("survived 150") == 1 and "calls that succeeded" in r.stdout, f"child died (rc {r.returncode}):\\n{r.stderr[-2000:]}"


import pytest


@pytest.mark.gpu
def test_mutated_exports_render_or_fail_cleanly_on_the_gpu():
    """The same sequences with a device behind them: prepareRender / render now run on whatever state the mutated
    export left (no light, no geometry, a material defined twice, triangles before their material ...)."""
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, "3", "120"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "calls that succeeded" in r.stdout, f"child died (rc {r.returncode}):\n{r.stderr[-2000:]}"


XML_CHILD = textwrap.dedent('''
    import sys, random, os, tempfile
    sys.path.insert(0, %(root)r)
    from libyafaray_amd import Interface
    src = open(os.path.join(%(root)r, "tests", "golden", sys.argv[2] if len(sys.argv) > 2 else "test01_dl.xml"), "rb").read()
    os.chdir(os.path.join(%(root)r, "tests", "golden"))      # texture file names in the scene are relative
    n = int(sys.argv[1])
    tmp = tempfile.mkdtemp()
    loaded = 0
    for i in range(n):
        rng = random.Random(i)
        data = bytearray(src)
        for _ in range(rng.randint(1, 8)):
            k = rng.randrange(5)
            p = rng.randrange(len(data))
            if k == 0: del data[p:p + rng.randint(1, 400)]                       # a hole
            elif k == 1: data[p] = rng.randrange(256)                            # a flipped byte
            elif k == 2: data = data[:p]                                         # truncated
            elif k == 3: data[p:p] = data[max(0, p - rng.randint(1, 300)):p]     # a repeated stretch
            else: data[p:p] = rng.choice([b"<", b">", b'"', b"&", b"\\x00", b"<mesh", b"</scene>", b"1e999", b"-nan"])
            if not data: data = bytearray(b"<")
        path = os.path.join(tmp, "m.xml")
        open(path, "wb").write(bytes(data))
        yi = Interface()
        try:
            yi.loadXml(path); loaded += 1
        except Exception:
            pass
        try: yi.close()
        except Exception: pass
    print("survived", n, "loaded", loaded)
''')


def test_damaged_scene_files_never_crash_the_xml_loader():
    r = subprocess.run([sys.executable, "-c", XML_CHILD % {"root": ROOT}, "400"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "survived 400" in r.stdout, f"child died (rc {r.returncode}):\n{r.stdout[-500:]}\n{r.stderr[-2000:]}"
    # the textured scene: <texture> elements, shader-node <list_element>s, orco coordinates
    r = subprocess.run([sys.executable, "-c", XML_CHILD % {"root": ROOT}, "150", "test01_tex.xml"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "survived 150" in r.stdout, f"child died (rc {r.returncode}):\n{r.stdout[-500:]}\n{r.stderr[-2000:]}"


IMG_CHILD = textwrap.dedent('''
    import sys, random, os, tempfile
    sys.path.insert(0, %(root)r)
    from libyafaray_amd import Interface
    n = int(sys.argv[1])
    tmp = tempfile.mkdtemp()
    srcs = {ext: open(os.path.join(%(root)r, "tests", "golden", "test01_tex." + ext), "rb").read() for ext in ("tga", "hdr", "png")}
    srcs["tga"] = srcs["tga"][:18 + 64 * 300 * 4]          # the first rows are enough to reach every branch; keeps the loop fast
    decoded = 0
    for i in range(n):
        rng = random.Random(i)
        ext = ("tga", "hdr", "png")[i %% 3]
        data = bytearray(srcs[ext])
        for _ in range(rng.randint(1, 6)):
            k = rng.randrange(5)
            p = rng.randrange(len(data)) if rng.random() < 0.5 else rng.randrange(min(len(data), 64))     # headers get half the damage
            if k == 0: del data[p:p + rng.randint(1, 2000)]
            elif k == 1: data[p] = rng.randrange(256)
            elif k == 2: data = data[:p]
            elif k == 3: data[p:p] = data[max(0, p - rng.randint(1, 300)):p]
            else: data[p:p + 2] = rng.choice([b"\\xff\\xff", b"\\x00\\x00", b"\\x7f\\xff", b"\\x80\\x00"])
            if not data: data = bytearray(b"\\x00")
        path = os.path.join(tmp, "m." + ext)
        open(path, "wb").write(bytes(data))
        yi = Interface(strict=False)
        yi.startScene(0)
        yi.paramsClearAll()
        yi.paramsSetString("type", "image"); yi.paramsSetString("filename", path)
        yi.paramsSetString("texture_optimization", rng.choice(["none", "optimized", "compressed"])); yi.paramsSetBool("img_grayscale", rng.random() < 0.3)
        if yi.createTexture("t"):
            decoded += 1
            yi.getTextureImage("t")
        yi.close()
    print("survived", n, "decoded", decoded)
''')


def test_damaged_image_files_never_crash_the_decoders():
    r = subprocess.run([sys.executable, "-c", IMG_CHILD % {"root": ROOT}, "240"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "survived 240" in r.stdout, f"child died (rc {r.returncode}):\n{r.stdout[-500:]}\n{r.stderr[-2000:]}"


def test_host_side_under_address_and_ub_sanitizers():
    """The host side of the C ABI (parameter maps, Interface state machine, geometry assembly, smoothMesh, XML loader,
    host kd builder) built with a stub for the device unit under AddressSanitizer + UBSan (tests/asan), driven by the
    same random call sequences, damaged scene files and damaged image files (TGA / HDR / PNG decoders).  GPU code cannot run under sanitizers on this pool."""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan) or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("no libasan / HIP host headers here")
    subprocess.run(["bash", os.path.join(ROOT, "tests", "asan", "build.sh")], check=True, timeout=900, capture_output=True)
    env = dict(os.environ, YAFARAY_LIBRARY=os.path.join(ROOT, "tests", "asan", "libyafaray_host_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, "500", "120"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "calls that succeeded" in r.stdout and "runtime error" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-c", XML_CHILD % {"root": ROOT}, "600"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "survived 600" in r.stdout and "runtime error" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-c", XML_CHILD % {"root": ROOT}, "200", "test01_tex.xml"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "survived 200" in r.stdout and "runtime error" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-c", IMG_CHILD % {"root": ROOT}, "300"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "survived 300" in r.stdout and "runtime error" not in r.stderr, r.stderr[-3000:]
