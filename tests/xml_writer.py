"""Write a scene / render dict pair (the shape libyafaray_amd.scenes produces) as a YafaRay scene XML, so that the
C++ loader can be driven with every feature of the device path.  Test support only."""
import numpy as np

_COLORS = {"color", "mirror_color", "diffuse_color", "filter_color", "absorption"}
_POINTS = {"from", "to", "up", "corner", "point1", "point2"}


def _param(k, v):
    if isinstance(v, bool):
        return f'\t<{k} bval="{"true" if v else "false"}"/>'
    if isinstance(v, (int, np.integer)):
        return f'\t<{k} ival="{int(v)}"/>'
    if isinstance(v, (float, np.floating)):
        return f'\t<{k} fval="{float(v)!r}"/>'
    if isinstance(v, str):
        return f'\t<{k} sval="{v}"/>'
    v = [float(x) for x in v]
    if k in _COLORS:
        return f'\t<{k} r="{v[0]!r}" g="{v[1]!r}" b="{v[2]!r}" a="1"/>'
    return f'\t<{k} x="{v[0]!r}" y="{v[1]!r}" z="{v[2]!r}"/>'


def write(path, scene, render, integrator_extra=None):
    out = ['<?xml version="1.0"?>', '<scene type="triangle">']
    for i, m in enumerate(scene["materials"]):
        out.append(f'<material name="mat{i}">')
        out += [_param(k, v) for k, v in m.items()]
        out.append("</material>")
    for i, l in enumerate(scene["lights"]):
        out.append(f'<light name="light{i}">')
        out += [_param(k, v) for k, v in l.items()]
        out.append("</light>")
    out.append('<camera name="cam">')
    out += [_param(k, v) for k, v in dict(scene["camera"], type="perspective").items()]
    out.append("</camera>")
    bg = render.get("background")
    if bg is not None:
        out += ['<background name="world_background">', _param("color", bg), _param("type", "constant"), "</background>"]
    integ = {"type": render.get("integrator", "pathtracing"), "caustic_type": "none"}
    if render.get("caustic_type", "none") == "path":
        integ.pop("caustic_type")      # no element at all: the loader's integrator then has the reference's default, path caustics
    for k in ("path_samples", "bounces", "russian_roulette_min_bounces", "no_recursive", "bg_transp", "bg_transp_refract", "raydepth",
              "transpShad", "shadowDepth"):
        if k in render:
            integ[k] = render[k]
    integ.update(integrator_extra or {})
    out.append('<integrator name="default">')
    out += [_param(k, v) for k, v in integ.items()]
    out += ["</integrator>", '<integrator name="volintegr">', _param("type", "none"), "</integrator>"]
    verts = np.asarray(scene["verts"], np.float32).reshape(-1, 3, 3)
    tm = np.asarray(scene["tri_mat"], np.int32)
    out.append(f'<mesh id="1" vertices="{3 * len(verts)}" faces="{len(verts)}" has_orco="false" has_uv="false" type="0">')
    for t in verts:
        for p in t:
            out.append(f'\t<p x="{float(p[0])!r}" y="{float(p[1])!r}" z="{float(p[2])!r}"/>')
    cur = None
    for i, m in enumerate(tm):
        if m != cur:
            out.append(f'\t<set_material sval="mat{int(m)}"/>')
            cur = m
        out.append(f'\t<f a="{3 * i}" b="{3 * i + 1}" c="{3 * i + 2}"/>')
    out.append("</mesh>")
    rs = {"camera_name": "cam", "integrator_name": "default", "volintegrator_name": "volintegr"}
    if bg is not None:
        rs["background_name"] = "world_background"
    for k, v in render.items():
        if k in ("integrator", "background", "caustic_type") or k in integ:
            continue
        rs[k] = v
    out.append("<render>")
    out += [_param(k, v) for k, v in rs.items()]
    out += ["</render>", "</scene>"]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
