#!/usr/bin/env python3
"""Derive BASELINE.json config 0 from the reference's own test scene (container only).

Reads /root/reference/tests/test01/test01.xml — a scene DATA file of the reference's tests — and writes
tests/golden/test01_pt.xml following SURVEY.md Appendix C: integrator directlighting -> pathtracing
(path_samples 1, bounces 3, Russian roulette off, no caustics), 256x256, 16 spp, box filter width 1,
linear tiles, one thread; textures / shader nodes / render passes / orco coordinates removed (no
textures on the device path: the six cubes keep their plain material colours)."""
import os
import re
import sys

SRC = "/root/reference/tests/test01/test01.xml"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test01_pt.xml")


def main():
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
    x = re.sub(r'<integrator name="default">.*?</integrator>', integ, x, flags=re.S)
    for k, v in (("resx", 256), ("resy", 256), ("width", 256), ("height", 256), ("AA_minsamples", 16), ("threads", 1)):
        x = re.sub(rf'<{k} ival="[^"]*"/>', f'<{k} ival="{v}"/>', x)
    x = re.sub(r'<AA_pixelwidth fval="[^"]*"/>', '<AA_pixelwidth fval="1"/>', x)
    x = re.sub(r'<filter_type sval="[^"]*"/>', '<filter_type sval="box"/>', x)
    x = re.sub(r'<tiles_order sval="[^"]*"/>', '<tiles_order sval="linear"/>', x)
    x = re.sub(r'<color_space sval="[^"]*"/>', '<color_space sval="LinearRGB"/>', x)
    x = re.sub(r"\n\s*\n+", "\n", x)
    head = ("<?xml version=\"1.0\"?>\n<!-- derived from the reference's tests/test01/test01.xml by tests/golden/make_test01_pt.py "
            "(BASELINE.json config 0: path tracing, 256x256, 16 spp; textures removed) -->\n")
    x = re.sub(r"^<\?xml[^>]*\?>\s*", "", x)
    open(DST, "w").write(head + x.strip() + "\n")
    print(DST, len(x), "bytes")


if __name__ == "__main__":
    main()
