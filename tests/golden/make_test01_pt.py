#!/usr/bin/env python3
"""Derive BASELINE.json config 0 from the reference's own test scene (container only).

Reads /root/reference/tests/test01/test01.xml — a scene DATA file of the reference's tests — and writes
tests/golden/test01_pt.xml following SURVEY.md Appendix C: integrator directlighting -> pathtracing
(path_samples 1, bounces 3, Russian roulette off, no caustics), 256x256, 16 spp, box filter width 1,
linear tiles, one thread; textures / shader nodes / render passes / orco coordinates removed (no
textures on the device path: the six cubes keep their plain material colours).

With --shipped it writes tests/golden/test01_dl.xml instead: the same stripping of textures, but the
integrator and render settings the reference ships (directlighting, 480x270, 1 spp, gauss filter width 1.5) —
the configuration BASELINE.md's 0.9 s badge was rendered with.

With --expected it copies the one rendered image the reference's tests hold for that scene
(tests/test01/"test01 - expected render result.png": 480x340 RGBA = a 70-row parameters badge on top of the 480x270
render, sRGB, 8 bit) to tests/golden/test01_expected.png and writes tests/golden/test01_expected.json: the badge height
and the names of the materials that carry NO shader node in the shipped scene (their pixels can be compared with a
render of test01_dl.xml), and the names of the materials whose texture file is one this build decodes (TGA, HDR, PNG).

With --textured it writes tests/golden/test01_tex.xml: the shipped settings again, this time KEEPING the image textures, shader
nodes and orco coordinates of the cubes whose texture file has a decoder here (test01_tex.tga / .hdr / .png, committed next
to it as data); the cubes textured from TIFF, JPEG and OpenEXR files (libraries this image lacks) keep their plain colours."""
import os
import re
import sys

SRC = "/root/reference/tests/test01/test01.xml"
DECODABLE = ("tga", "hdr", "png")
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test01_pt.xml")


def expected():
    import json
    import shutil
    here = os.path.dirname(os.path.abspath(__file__))
    src_png = os.path.join(os.path.dirname(SRC), "test01 - expected render result.png")
    shutil.copyfile(src_png, os.path.join(here, "test01_expected.png"))
    x = re.sub(r"<!--.*?-->", "", open(SRC).read(), flags=re.S)
    plain = [m.group(1) for m in re.finditer(r'<material name="([^"]*)">(.*?)</material>', x, flags=re.S) if "<list_element>" not in m.group(2)]
    tex_file = {m.group(1): re.search(r'<filename sval="([^"]*)"/>', m.group(2)).group(1) for m in re.finditer(r'<texture name="([^"]*)">(.*?)</texture>', x, flags=re.S)}
    decodable = []
    for m in re.finditer(r'<material name="([^"]*)">(.*?)</material>', x, flags=re.S):
        t = re.search(r'<texture sval="([^"]*)"/>', m.group(2))
        if t and tex_file[t.group(1)].rsplit(".", 1)[-1].lower() in DECODABLE:
            decodable.append(m.group(1))
    meta = {"source": "tests/test01/test01 - expected render result.png of the reference", "badge_rows_on_top": 70, "width": 480, "height": 270,
            "color_space": "sRGB", "untextured_materials": plain, "decodable_textured_materials": decodable}
    json.dump(meta, open(os.path.join(here, "test01_expected.json"), "w"), indent=1)
    print(meta)


def textured():
    """test01.xml with the shipped settings; textures / nodes / orco kept where the texture file can be decoded here"""
    import shutil
    here = os.path.dirname(os.path.abspath(__file__))
    x = re.sub(r"<!--.*?-->", "", open(SRC).read(), flags=re.S)
    tex_file = {m.group(1): re.search(r'<filename sval="([^"]*)"/>', m.group(2)).group(1) for m in re.finditer(r'<texture name="([^"]*)">(.*?)</texture>', x, flags=re.S)}
    keep = {n for n, f in tex_file.items() if f.rsplit(".", 1)[-1].lower() in DECODABLE}
    for n in keep:
        shutil.copyfile(os.path.join(os.path.dirname(SRC), tex_file[n]), os.path.join(here, tex_file[n]))

    def texture_block(m):
        return m.group(0) if m.group(1) in keep else ""
    x = re.sub(r'<texture name="([^"]*)">.*?</texture>\s*', texture_block, x, flags=re.S)

    def material_block(m):
        body = m.group(0)
        t = re.search(r'<texture sval="([^"]*)"/>', body)
        if t and t.group(1) not in keep:
            body = re.sub(r"\s*<list_element>.*?</list_element>", "", body, flags=re.S)
            body = re.sub(r"\s*<\w+_shader sval=\"[^\"]*\"/>", "", body)
        return body
    x = re.sub(r'<material name="[^"]*">.*?</material>', material_block, x, flags=re.S)
    for el in ("render_passes", "logging_badge"):
        x = re.sub(rf"<{el} name=.*?</{el}>\s*", "", x, flags=re.S)

    def render_block(m):
        r = m.group(0)
        r = re.sub(r'<threads ival="[^"]*"/>', '<threads ival="1"/>', r)
        r = re.sub(r'<tiles_order sval="[^"]*"/>', '<tiles_order sval="linear"/>', r)
        return re.sub(r'<color_space sval="[^"]*"/>', '<color_space sval="LinearRGB"/>', r)
    x = re.sub(r"<render>.*?</render>", render_block, x, flags=re.S)
    x = re.sub(r"\n\s*\n+", "\n", x)
    x = re.sub(r"^<\?xml[^>]*\?>\s*", "", x)
    head = ("<?xml version=\"1.0\"?>\n<!-- derived from the reference's tests/test01/test01.xml by tests/golden/make_test01_pt.py, option textured "
            "(shipped settings; TGA / HDR / PNG textures kept, TIFF / JPEG / OpenEXR ones removed) -->\n")
    dst = os.path.join(here, "test01_tex.xml")
    open(dst, "w").write(head + x.strip() + "\n")
    print(dst, len(x), "bytes; textures kept:", sorted(keep))


def main():
    if "--expected" in sys.argv[1:]:
        return expected()
    if "--textured" in sys.argv[1:]:
        return textured()
    shipped = "--shipped" in sys.argv[1:]
    dst = DST.replace("test01_pt", "test01_dl") if shipped else DST
    if not os.path.exists(SRC):
        sys.exit("reference tree not present")
    x = open(SRC).read()
    x = re.sub(r"<!--.*?-->", "", x, flags=re.S)
    for el in ("texture", "render_passes", "logging_badge"):
        x = re.sub(rf"<{el} name=.*?</{el}>\s*", "", x, flags=re.S)
    x = re.sub(r"\s*<list_element>.*?</list_element>", "", x, flags=re.S)
    x = re.sub(r"\s*<\w+_shader sval=\"[^\"]*\"/>", "", x)
    x = x.replace('has_orco="true"', 'has_orco="false"')
    x = re.sub(r'\s+o[xyz]="[^"]*"', "", x)
    integ = """<integrator name="default">
	<bg_transp bval="false"/>
	<bg_transp_refract bval="false"/>
	<bounces ival="3"/>
	<caustic_type sval="none"/>
	<do_AO bval="false"/>
	<no_recursive bval="false"/>
	<path_samples ival="1"/>
	<raydepth ival="8"/>
	<russian_roulette_min_bounces ival="3"/>
	<transpShad bval="false"/>
	<type sval="pathtracing"/>
</integrator>"""
    if not shipped:
        x = re.sub(r'<integrator name="default">.*?</integrator>', integ, x, flags=re.S)
        for k, v in (("resx", 256), ("resy", 256), ("width", 256), ("height", 256), ("AA_minsamples", 16)):
            x = re.sub(rf'<{k} ival="[^"]*"/>', f'<{k} ival="{v}"/>', x)
        x = re.sub(r'<AA_pixelwidth fval="[^"]*"/>', '<AA_pixelwidth fval="1"/>', x)
        x = re.sub(r'<filter_type sval="[^"]*"/>', '<filter_type sval="box"/>', x)
    x = re.sub(r'<threads ival="[^"]*"/>', '<threads ival="1"/>', x)
    x = re.sub(r'<tiles_order sval="[^"]*"/>', '<tiles_order sval="linear"/>', x)
    x = re.sub(r'<color_space sval="[^"]*"/>', '<color_space sval="LinearRGB"/>', x)
    x = re.sub(r"\n\s*\n+", "\n", x)
    head = ("<?xml version=\"1.0\"?>\n<!-- derived from the reference's tests/test01/test01.xml by tests/golden/make_test01_pt.py "
            + ("(shipped settings: directlighting, gauss 1.5, 1 spp; textures removed) -->\n" if shipped else
               "(BASELINE.json config 0: path tracing, 256x256, 16 spp; textures removed) -->\n"))
    x = re.sub(r"^<\?xml[^>]*\?>\s*", "", x)
    open(dst, "w").write(head + x.strip() + "\n")
    print(dst, len(x), "bytes")


if __name__ == "__main__":
    main()
