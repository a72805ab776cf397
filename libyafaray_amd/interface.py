"""Host-side mirror of yafaray4::Interface (include/interface/interface.h:48-139) over the C ABI.

Method names and argument meaning follow the reference class (and its SWIG module
src/bindings/yafaray4_interface.i), so a script written against the reference's Python bindings
ports by changing the import.  Error behaviour follows the reference too (False / None on failure)
with the diagnostic available from getLastError(); `strict=True` raises instead.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # YAFARAY_LIBRARY: test support (tests/asan builds the host side with a device stub under AddressSanitizer)
    return os.environ.get("YAFARAY_LIBRARY") or os.path.join(_HERE, "libyafaray_gpu.so")


class YafaRayError(RuntimeError):
    pass


class RenderStats(C.Structure):
    _fields_ = [
        ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("interior_steps", C.c_uint64), ("leaves", C.c_uint64),
        ("tri_tests", C.c_uint64), ("camera_samples", C.c_uint64), ("restarts", C.c_uint64),
        ("tree_build_seconds", C.c_double), ("upload_seconds", C.c_double), ("render_seconds", C.c_double),
        ("kd_nodes", C.c_uint32), ("kd_leaf_refs", C.c_uint32), ("kd_max_depth", C.c_uint32), ("n_triangles", C.c_uint32),
        ("scene_device_bytes", C.c_uint64),
    ]


PUTPIXEL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float)
FLUSH = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)      # yafaray_plane_exchange_t
AREA = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)


class Output(C.Structure):
    _fields_ = [("user", C.c_void_p), ("putPixel", PUTPIXEL), ("flush", FLUSH), ("flushArea", AREA), ("highlightArea", AREA)]


_lib = None

# every symbol include/yafaray_c_api.h and include/yafgpu.h declare (checked by tests/test_abi.py)
C_API_SYMBOLS = [
    "yafaray_createInterface", "yafaray_destroyInterface", "yafaray_getLastError", "yafaray_getVersion",
    "yafaray_startScene", "yafaray_startGeometry", "yafaray_endGeometry", "yafaray_getNextFreeId",
    "yafaray_startTriMesh", "yafaray_endTriMesh", "yafaray_addVertex", "yafaray_addNormal", "yafaray_addTriangle",
    "yafaray_smoothMesh", "yafaray_getMeshCornerNormals", "yafaray_addTriangles",
    "yafaray_startTriMeshPtr", "yafaray_addVertexWithOrco", "yafaray_addUv", "yafaray_addTriangleWithUv",
    "yafaray_startCurveMesh", "yafaray_endCurveMesh", "yafaray_addInstance",
    "yafaray_paramsSetColorArray", "yafaray_paramsSetMatrix", "yafaray_paramsSetMatrixD", "yafaray_setInputColorSpace",
    "yafaray_createObject", "yafaray_createVolumeRegion", "yafaray_createImageHandler",
    "yafaray_setLoggingAndBadgeSettings", "yafaray_setupRenderPasses", "yafaray_setInteractive", "yafaray_getRenderParameters",
    "yafaray_setConsoleVerbosityLevel", "yafaray_setLogVerbosityLevel", "yafaray_setParamsBadgePosition", "yafaray_getDrawParams",
    "yafaray_printDebug", "yafaray_printVerbose", "yafaray_printInfo", "yafaray_printParams", "yafaray_printWarning", "yafaray_printError",
    "yafaray_setOutput2",
    "yafaray_paramsSetPoint", "yafaray_paramsSetString", "yafaray_paramsSetBool", "yafaray_paramsSetInt",
    "yafaray_paramsSetFloat", "yafaray_paramsSetColor", "yafaray_paramsClearAll", "yafaray_paramsStartList",
    "yafaray_paramsPushList", "yafaray_paramsEndList",
    "yafaray_createTexture", "yafaray_createTextureFromMemory", "yafaray_getTextureImage",
    "yafaray_createLight", "yafaray_createMaterial", "yafaray_createCamera", "yafaray_createBackground",
    "yafaray_createIntegrator", "yafaray_clearAll", "yafaray_render", "yafaray_abort", "yafaray_getRenderedImage",
    "yafaray_getFilm", "yafaray_getRenderStats", "yafaray_setShard", "yafaray_setPlaneExchange", "yafaray_setSerialReplay", "yafaray_getRandState", "yafaray_setRandState", "yafaray_prepareRender",
    "yafaray_renderPassDevice", "yafaray_getRenderSize", "yafaray_loadXml", "yafaray_intersectRays", "yafaray_shadowRays", "yafaray_probe", "yafaray_setProfiling", "yafaray_getKernelProfile", "yafaray_setPassPipelining",
    "yafaray_commGetUniqueId", "yafaray_commCreate", "yafaray_commDestroy", "yafaray_commRank", "yafaray_commWorld", "yafaray_commLastError",
    "yafaray_commBackend", "yafaray_reduceFilm", "yafaray_allReduce", "yafaray_commExchange", "yafaray_setComm",
]
GPU_ABI_SYMBOLS = [
    "yafgpu_last_error", "yafgpu_device_count", "yafgpu_set_device", "yafgpu_scene_create", "yafgpu_scene_destroy",
    "yafgpu_scene_info", "yafgpu_planes_bytes", "yafgpu_render_tiles", "yafgpu_film_combine", "yafgpu_render_to_host", "yafgpu_render_passes_to_host",
    "yafgpu_trace_closest", "yafgpu_trace_shadow", "yafgpu_scene_get_tree", "yafgpu_probe",
    "yafgpu_kdtree_build", "yafgpu_kdtree_build_device", "yafgpu_kdtree_info", "yafgpu_kdtree_get", "yafgpu_kdtree_destroy",
    "yafgpu_set_profiling", "yafgpu_scene_set_pass_pipelining", "yafgpu_get_profile", "yafgpu_scene_set_abort_flag", "yafgpu_scene_set_exchange", "yafgpu_glibc_rand",
]


def load():
    """Load libyafaray_gpu.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise YafaRayError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(libyafaray_amd/csrc/build.sh); there is no CPU fallback")
    L = C.CDLL(p)
    vp, cp, ci, cd, cf = C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_float
    sig = {
        "yafaray_createInterface": (vp, []), "yafaray_destroyInterface": (None, [vp]),
        "yafaray_getLastError": (cp, [vp]), "yafaray_getVersion": (cp, []),
        "yafaray_startScene": (ci, [vp, ci]), "yafaray_startGeometry": (ci, [vp]), "yafaray_endGeometry": (ci, [vp]),
        "yafaray_getNextFreeId": (C.c_uint, [vp]),
        "yafaray_startTriMesh": (ci, [vp, C.c_uint, ci, ci, ci, ci, ci, ci]), "yafaray_endTriMesh": (ci, [vp]),
        "yafaray_addVertex": (ci, [vp, cd, cd, cd]), "yafaray_addNormal": (None, [vp, cd, cd, cd]),
        "yafaray_addTriangle": (ci, [vp, ci, ci, ci, vp]), "yafaray_smoothMesh": (ci, [vp, C.c_uint, cd]),
        "yafaray_getMeshCornerNormals": (ci, [vp, C.c_uint, C.POINTER(cf), ci]),
        "yafaray_addTriangles": (ci, [vp, ci, C.POINTER(cf), ci, C.POINTER(ci), vp]),
        "yafaray_startTriMeshPtr": (ci, [vp, C.POINTER(C.c_uint), ci, ci, ci, ci, ci, ci]),
        "yafaray_addVertexWithOrco": (ci, [vp, cd, cd, cd, cd, cd, cd]), "yafaray_addUv": (ci, [vp, cf, cf]),
        "yafaray_addTriangleWithUv": (ci, [vp, ci, ci, ci, ci, ci, ci, vp]),
        "yafaray_startCurveMesh": (ci, [vp, C.c_uint, ci, ci]), "yafaray_endCurveMesh": (ci, [vp, vp, cf, cf, cf]),
        "yafaray_addInstance": (ci, [vp, C.c_uint, C.POINTER(cf)]),
        "yafaray_paramsSetColorArray": (None, [vp, cp, C.POINTER(cf), ci]), "yafaray_paramsSetMatrix": (None, [vp, cp, C.POINTER(cf), ci]),
        "yafaray_paramsSetMatrixD": (None, [vp, cp, C.POINTER(cd), ci]), "yafaray_setInputColorSpace": (None, [vp, cp, cf]),
        "yafaray_createObject": (C.c_uint, [vp, cp]), "yafaray_createVolumeRegion": (vp, [vp, cp]), "yafaray_createImageHandler": (vp, [vp, cp, ci]),
        "yafaray_setLoggingAndBadgeSettings": (ci, [vp]), "yafaray_setupRenderPasses": (ci, [vp]), "yafaray_setInteractive": (ci, [vp, ci]),
        "yafaray_getRenderParameters": (ci, [vp, C.c_char_p, ci]),
        "yafaray_setConsoleVerbosityLevel": (None, [vp, cp]), "yafaray_setLogVerbosityLevel": (None, [vp, cp]),
        "yafaray_setParamsBadgePosition": (None, [vp, cp]), "yafaray_getDrawParams": (ci, [vp]),
        "yafaray_printDebug": (None, [vp, cp]), "yafaray_printVerbose": (None, [vp, cp]), "yafaray_printInfo": (None, [vp, cp]),
        "yafaray_printParams": (None, [vp, cp]), "yafaray_printWarning": (None, [vp, cp]), "yafaray_printError": (None, [vp, cp]),
        "yafaray_setOutput2": (None, [vp, C.POINTER(Output)]),
        "yafaray_paramsSetPoint": (None, [vp, cp, cd, cd, cd]), "yafaray_paramsSetString": (None, [vp, cp, cp]),
        "yafaray_paramsSetBool": (None, [vp, cp, ci]), "yafaray_paramsSetInt": (None, [vp, cp, ci]),
        "yafaray_paramsSetFloat": (None, [vp, cp, cd]), "yafaray_paramsSetColor": (None, [vp, cp, cf, cf, cf, cf]),
        "yafaray_paramsClearAll": (None, [vp]), "yafaray_paramsStartList": (None, [vp]),
        "yafaray_paramsPushList": (None, [vp]), "yafaray_paramsEndList": (None, [vp]),
        "yafaray_createLight": (vp, [vp, cp]), "yafaray_createMaterial": (vp, [vp, cp]),
        "yafaray_createTexture": (vp, [vp, cp]), "yafaray_createTextureFromMemory": (vp, [vp, cp, ci, ci, C.POINTER(C.c_float)]),
        "yafaray_getTextureImage": (ci, [vp, cp, C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_float), ci]),
        "yafaray_createCamera": (vp, [vp, cp]), "yafaray_createBackground": (vp, [vp, cp]),
        "yafaray_createIntegrator": (vp, [vp, cp]), "yafaray_clearAll": (None, [vp]),
        "yafaray_render": (ci, [vp, C.POINTER(Output), vp]), "yafaray_abort": (None, [vp]),
        "yafaray_getRenderedImage": (ci, [vp, ci, C.POINTER(Output)]),
        "yafaray_getFilm": (ci, [vp, C.POINTER(cf), ci, ci]), "yafaray_getRenderStats": (ci, [vp, C.POINTER(RenderStats)]),
        "yafaray_setShard": (None, [vp, ci, ci]), "yafaray_setPlaneExchange": (None, [vp, EXCHANGE, vp]), "yafaray_setSerialReplay": (None, [vp, ci]), "yafaray_prepareRender": (ci, [vp]),
        "yafaray_renderPassDevice": (ci, [vp, vp, vp, vp]), "yafaray_getRenderSize": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
        "yafaray_loadXml": (ci, [vp, cp]), "yafaray_getRandState": (None, [vp, C.POINTER(ci), C.POINTER(ci)]), "yafaray_setRandState": (None, [vp, ci, ci]),
        "yafaray_intersectRays": (ci, [vp, ci, C.POINTER(cf), C.POINTER(ci), C.POINTER(cf), C.POINTER(cf)]),
        "yafaray_shadowRays": (ci, [vp, ci, C.POINTER(cf), C.POINTER(ci)]),
        "yafaray_probe": (ci, [vp, ci, ci, C.POINTER(cf), ci, C.POINTER(cf), ci]),
        "yafaray_setProfiling": (ci, [vp, ci]),
        "yafaray_setPassPipelining": (ci, [vp, ci]),
        "yafaray_getKernelProfile": (ci, [vp, C.POINTER(cd), C.POINTER(C.c_uint64)]),
        "yafaray_commGetUniqueId": (ci, [C.c_char_p]), "yafaray_commCreate": (vp, [C.c_char_p, ci, ci, ci]), "yafaray_commDestroy": (None, [vp]),
        "yafaray_commRank": (ci, [vp]), "yafaray_commWorld": (ci, [vp]), "yafaray_commLastError": (cp, []), "yafaray_commBackend": (cp, []),
        "yafaray_reduceFilm": (ci, [vp, vp, C.c_uint64, ci, vp]), "yafaray_allReduce": (ci, [vp, vp, C.c_uint64, vp]),
        "yafaray_commExchange": (ci, [vp, vp, C.c_uint64]), "yafaray_setComm": (None, [vp, vp]),
        "yafgpu_last_error": (cp, []), "yafgpu_device_count": (ci, []), "yafgpu_set_device": (ci, [ci]),
        "yafgpu_planes_bytes": (C.c_uint64, [ci, ci]),
        "yafgpu_film_combine": (ci, [vp, vp, ci, ci, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _b(s):
    return s.encode() if isinstance(s, str) else s


class Interface:
    """yafaray4::Interface.  See include/yafaray_c_api.h for the per-method reference citations."""

    def __init__(self, strict=True):
        self._L = load()
        self._h = self._L.yafaray_createInterface()
        self.strict = strict
        self._keep = []

    # -- plumbing
    def close(self):
        if self._h:
            self._L.yafaray_destroyInterface(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def getLastError(self):
        return self._L.yafaray_getLastError(self._h).decode()

    def getVersion(self):
        return self._L.yafaray_getVersion().decode()

    def _ok(self, r, what):
        if not r and self.strict:
            raise YafaRayError(f"{what}: {self.getLastError()}")
        return bool(r)

    def _obj(self, r, what):
        if not r and self.strict:
            raise YafaRayError(f"{what}: {self.getLastError()}")
        return r

    # -- scene
    def startScene(self, type=0):
        return self._ok(self._L.yafaray_startScene(self._h, type), "startScene")

    def startGeometry(self):
        return self._ok(self._L.yafaray_startGeometry(self._h), "startGeometry")

    def endGeometry(self):
        return self._ok(self._L.yafaray_endGeometry(self._h), "endGeometry")

    def getNextFreeId(self):
        return self._L.yafaray_getNextFreeId(self._h)

    def startTriMesh(self, id, vertices, triangles, has_orco, has_uv=False, type=0, obj_pass_index=0):
        return self._ok(self._L.yafaray_startTriMesh(self._h, id, vertices, triangles, int(has_orco), int(has_uv), type,
                                                     obj_pass_index), "startTriMesh")

    def endTriMesh(self):
        return self._ok(self._L.yafaray_endTriMesh(self._h), "endTriMesh")

    def addVertex(self, x, y, z):
        return self._L.yafaray_addVertex(self._h, x, y, z)

    def addNormal(self, nx, ny, nz):
        self._L.yafaray_addNormal(self._h, nx, ny, nz)

    def addTriangle(self, a, b, c, mat):
        return self._ok(self._L.yafaray_addTriangle(self._h, a, b, c, mat), "addTriangle")

    def startTriMeshPtr(self, vertices, triangles, has_orco, has_uv=False, type=0, obj_pass_index=0):
        """-> the id the scene picked (the reference's SWIG typemap returns it the same way)"""
        mid = C.c_uint(0)
        ok = self._ok(self._L.yafaray_startTriMeshPtr(self._h, C.byref(mid), vertices, triangles, int(has_orco), int(has_uv), type, obj_pass_index), "startTriMeshPtr")
        return mid.value if ok else 0

    def addVertexWithOrco(self, x, y, z, ox, oy, oz):
        return self._L.yafaray_addVertexWithOrco(self._h, x, y, z, ox, oy, oz)

    def addUv(self, u, v):
        return self._L.yafaray_addUv(self._h, u, v)

    def addTriangleWithUv(self, a, b, c, uv_a, uv_b, uv_c, mat):
        return self._ok(self._L.yafaray_addTriangleWithUv(self._h, a, b, c, uv_a, uv_b, uv_c, mat), "addTriangle")

    def startCurveMesh(self, id, vertices, obj_pass_index=0):
        return self._ok(self._L.yafaray_startCurveMesh(self._h, id, vertices, obj_pass_index), "startCurveMesh")

    def endCurveMesh(self, mat, strand_start, strand_end, strand_shape):
        return self._ok(self._L.yafaray_endCurveMesh(self._h, mat, strand_start, strand_end, strand_shape), "endCurveMesh")

    def addInstance(self, base_object_id, obj_to_world):
        m = (C.c_float * 16)(*[float(x) for x in np.asarray(obj_to_world, np.float32).reshape(16)])
        return self._ok(self._L.yafaray_addInstance(self._h, base_object_id, m), "addInstance")

    def createObject(self, name):
        return self._obj(self._L.yafaray_createObject(self._h, _b(name)), "createObject")

    def createVolumeRegion(self, name):
        return self._obj(self._L.yafaray_createVolumeRegion(self._h, _b(name)), "createVolumeRegion")

    def createImageHandler(self, name, add_to_table=True):
        return self._obj(self._L.yafaray_createImageHandler(self._h, _b(name), int(add_to_table)), "createImageHandler")

    def setLoggingAndBadgeSettings(self):
        return self._ok(self._L.yafaray_setLoggingAndBadgeSettings(self._h), "setLoggingAndBadgeSettings")

    def setupRenderPasses(self):
        return self._ok(self._L.yafaray_setupRenderPasses(self._h), "setupRenderPasses")

    def setInteractive(self, interactive):
        return bool(self._L.yafaray_setInteractive(self._h, int(interactive)))

    def getRenderParameters(self):
        """the render ParamMap as a dict of strings"""
        n = self._L.yafaray_getRenderParameters(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self._L.yafaray_getRenderParameters(self._h, buf, n + 1)
        return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)

    def setConsoleVerbosityLevel(self, level):
        self._L.yafaray_setConsoleVerbosityLevel(self._h, _b(level))

    def setLogVerbosityLevel(self, level):
        self._L.yafaray_setLogVerbosityLevel(self._h, _b(level))

    def setParamsBadgePosition(self, position="none"):
        self._L.yafaray_setParamsBadgePosition(self._h, _b(position))

    def getDrawParams(self):
        return bool(self._L.yafaray_getDrawParams(self._h))

    def printInfo(self, msg):
        self._L.yafaray_printInfo(self._h, _b(msg))

    def printWarning(self, msg):
        self._L.yafaray_printWarning(self._h, _b(msg))

    def printError(self, msg):
        self._L.yafaray_printError(self._h, _b(msg))

    def setInputColorSpace(self, color_space_string, gamma_val):
        self._L.yafaray_setInputColorSpace(self._h, _b(color_space_string), gamma_val)

    def paramsSetMatrix(self, name, m, transpose=False):
        a = (C.c_float * 16)(*[float(x) for x in np.asarray(m, np.float32).reshape(16)])
        self._L.yafaray_paramsSetMatrix(self._h, _b(name), a, int(transpose))

    def paramsSetMemMatrix(self, name, m, transpose=False):
        self.paramsSetMatrix(name, m, transpose)

    def addTriangles(self, verts, indices, mat):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        return self._ok(self._L.yafaray_addTriangles(self._h, v.shape[0], v.ctypes.data_as(C.POINTER(C.c_float)), i.shape[0],
                                                     i.ctypes.data_as(C.POINTER(C.c_int)), mat), "addTriangles")

    def smoothMesh(self, id, angle):
        return self._ok(self._L.yafaray_smoothMesh(self._h, id, angle), "smoothMesh")

    def getMeshCornerNormals(self, id, n_tris):
        out = np.zeros((n_tris, 3, 3), dtype=np.float32)
        self._ok(self._L.yafaray_getMeshCornerNormals(self._h, id, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "getMeshCornerNormals")
        return out

    # -- params
    def paramsSetPoint(self, name, x, y, z):
        self._L.yafaray_paramsSetPoint(self._h, _b(name), x, y, z)

    def paramsSetString(self, name, s):
        self._L.yafaray_paramsSetString(self._h, _b(name), _b(s))

    def paramsSetBool(self, name, b):
        self._L.yafaray_paramsSetBool(self._h, _b(name), int(b))

    def paramsSetInt(self, name, i):
        self._L.yafaray_paramsSetInt(self._h, _b(name), int(i))

    def paramsSetFloat(self, name, f):
        self._L.yafaray_paramsSetFloat(self._h, _b(name), float(f))

    def paramsSetColor(self, name, r, g, b, a=1.0):
        self._L.yafaray_paramsSetColor(self._h, _b(name), r, g, b, a)

    def paramsClearAll(self):
        self._L.yafaray_paramsClearAll(self._h)

    def paramsStartList(self):
        self._L.yafaray_paramsStartList(self._h)

    def paramsPushList(self):
        self._L.yafaray_paramsPushList(self._h)

    def paramsEndList(self):
        self._L.yafaray_paramsEndList(self._h)

    def paramsSet(self, d):
        """Convenience: fill the ParamMap from a dict, typed like the XML grammar (import_xml.cc:273-318):
        str -> sval, bool -> bval, int -> ival, float -> fval, 3-tuple -> point, ('color', r,g,b[,a]) -> colour."""
        for k, v in d.items():
            if isinstance(v, str):
                self.paramsSetString(k, v)
            elif isinstance(v, bool):
                self.paramsSetBool(k, v)
            elif isinstance(v, (int, np.integer)):
                self.paramsSetInt(k, v)
            elif isinstance(v, (float, np.floating)):
                self.paramsSetFloat(k, v)
            elif isinstance(v, tuple) and len(v) in (4, 5) and v[0] == "color":
                self.paramsSetColor(k, *[float(t) for t in v[1:]])
            elif len(v) == 3:
                self.paramsSetPoint(k, float(v[0]), float(v[1]), float(v[2]))
            else:
                raise TypeError(f"parameter {k}: {v!r}")

    # -- factories
    def createLight(self, name):
        return self._obj(self._L.yafaray_createLight(self._h, _b(name)), "createLight")

    def createMaterial(self, name):
        return self._obj(self._L.yafaray_createMaterial(self._h, _b(name)), "createMaterial")

    def createTexture(self, name):
        return self._obj(self._L.yafaray_createTexture(self._h, _b(name)), "createTexture")

    def createTextureFromMemory(self, name, texels):
        """texels: (height, width, 4) float32, as the reference's ImageHandler::getPixel would return them"""
        import numpy as np
        px = np.ascontiguousarray(texels, dtype=np.float32)
        return self._obj(self._L.yafaray_createTextureFromMemory(self._h, _b(name), px.shape[1], px.shape[0], px.ctypes.data_as(C.POINTER(C.c_float))), "createTextureFromMemory")

    def getTextureImage(self, name):
        import numpy as np
        w, h = C.c_int(0), C.c_int(0)
        self._ok(self._L.yafaray_getTextureImage(self._h, _b(name), C.byref(w), C.byref(h), None, 0), "getTextureImage")
        px = np.zeros((h.value, w.value, 4), np.float32)
        self._ok(self._L.yafaray_getTextureImage(self._h, _b(name), C.byref(w), C.byref(h), px.ctypes.data_as(C.POINTER(C.c_float)), px.size), "getTextureImage")
        return px

    def createCamera(self, name):
        return self._obj(self._L.yafaray_createCamera(self._h, _b(name)), "createCamera")

    def createBackground(self, name):
        return self._obj(self._L.yafaray_createBackground(self._h, _b(name)), "createBackground")

    def createIntegrator(self, name):
        return self._obj(self._L.yafaray_createIntegrator(self._h, _b(name)), "createIntegrator")

    def clearAll(self):
        self._L.yafaray_clearAll(self._h)

    # -- render
    def render(self, output=None, progress=None):
        out = None
        if output is not None:
            out = Output()
            pp = PUTPIXEL(lambda u, v, x, y, r, g, b, a: int(bool(output.putPixel(v, x, y, (r, g, b, a)))))
            fl = FLUSH(lambda u, v: output.flush(v) if hasattr(output, "flush") else None)
            self._keep = [pp, fl]
            out.putPixel = pp
            out.flush = fl
        return self._ok(self._L.yafaray_render(self._h, C.byref(out) if out is not None else None, None), "render")

    def abort(self):
        self._L.yafaray_abort(self._h)

    def loadXml(self, path):
        return self._ok(self._L.yafaray_loadXml(self._h, _b(path)), "loadXml")

    # -- additions (measurement / multi-GPU)
    def setShard(self, index, count):
        self._L.yafaray_setShard(self._h, index, count)

    def setPlaneExchange(self, fn):
        """fn(device_pointer: int, n_floats: int): sum that float32 device array over all ranks in place (see
        libyafaray_amd.parallel.plane_exchange); None detaches it"""
        if fn is None:
            self._exchange = None
            self._L.yafaray_setPlaneExchange(self._h, EXCHANGE(0), None)
            return

        def thunk(user, ptr, n):
            try:
                fn(int(ptr or 0), int(n))
                return 0
            except Exception as e:            # an exception must not unwind through the C caller
                print(f"plane exchange failed: {e!r}", file=sys.stderr)
                return 1
        self._exchange = EXCHANGE(thunk)
        self._L.yafaray_setPlaneExchange(self._h, self._exchange, None)

    def setComm(self, comm):
        """attach a FilmComm (libyafaray_amd.parallel): plane / light-counter exchanges of a sharded render run over its RCCL
        communicator inside the library; None detaches"""
        self._comm = comm
        self._L.yafaray_setComm(self._h, comm.handle if comm is not None else None)

    def setSerialReplay(self, on):
        self._L.yafaray_setSerialReplay(self._h, int(bool(on)))

    def getRandState(self):
        """(srand seed, values consumed since) of the libc stream the tile seeds continue — what the oracle needs to
        render the same scene (`rand_srand`, `rand_skip`)"""
        a, b = C.c_int(), C.c_int()
        self._L.yafaray_getRandState(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def setRandState(self, srand_seed, skip=0):
        """the embedder called libc srand(seed) itself after building the scene (and drew `skip` values): tile seeds continue there"""
        self._L.yafaray_setRandState(self._h, int(srand_seed), int(skip))

    def prepareRender(self):
        return self._ok(self._L.yafaray_prepareRender(self._h), "prepareRender")

    def getRenderSize(self):
        w, h = C.c_int(), C.c_int()
        self._ok(self._L.yafaray_getRenderSize(self._h, C.byref(w), C.byref(h)), "getRenderSize")
        return w.value, h.value

    def renderPassDevice(self, d_planes, d_counters=0, stream=0):
        return self._ok(self._L.yafaray_renderPassDevice(self._h, C.c_void_p(d_planes), C.c_void_p(d_counters or None),
                                                         C.c_void_p(stream or None)), "renderPassDevice")

    def intersectRays(self, rays):
        """rays (n,8) float32 -> (tri (n,) int32, t (n,), bary (n,3))"""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        tri = np.zeros(n, np.int32); t = np.zeros(n, np.float32); bary = np.zeros((n, 3), np.float32)
        fp = C.POINTER(C.c_float)
        self._ok(self._L.yafaray_intersectRays(self._h, n, r.ctypes.data_as(fp), tri.ctypes.data_as(C.POINTER(C.c_int)),
                                               t.ctypes.data_as(fp), bary.ctypes.data_as(fp)), "intersectRays")
        return tri, t, bary

    def shadowRays(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        sh = np.zeros(r.shape[0], np.int32)
        self._ok(self._L.yafaray_shadowRays(self._h, r.shape[0], r.ctypes.data_as(C.POINTER(C.c_float)),
                                            sh.ctypes.data_as(C.POINTER(C.c_int))), "shadowRays")
        return sh

    def probe(self, op, inp, n_out):
        """device-side component probe; inp (n, n_in) float32 (integers as bit patterns) -> (n, n_out) float32"""
        x = np.ascontiguousarray(inp, dtype=np.float32)
        x = x.reshape(x.shape[0], -1)
        out = np.zeros((x.shape[0], n_out), np.float32)
        fp = C.POINTER(C.c_float)
        self._ok(self._L.yafaray_probe(self._h, op, x.shape[0], x.ctypes.data_as(fp), x.shape[1], out.ctypes.data_as(fp), n_out), "probe")
        return out

    def setPassPipelining(self, mode):
        """-1: by size (default), 0: off, 1: on — consecutive independent passes on two internal streams (include/yafgpu.h)"""
        return self._ok(self._L.yafaray_setPassPipelining(self._h, int(mode)), "setPassPipelining")

    def setProfiling(self, enable):
        return self._ok(self._L.yafaray_setProfiling(self._h, int(enable)), "setProfiling")

    def getKernelProfile(self):
        """-> {name: (total_ms, launches)} of the last profiled pass"""
        ms = (C.c_double * 4)(); n = (C.c_uint64 * 4)()
        self._ok(self._L.yafaray_getKernelProfile(self._h, ms, n), "getKernelProfile")
        names = ("trace_closest", "trace_shadow", "shade", "other")
        return {k: (ms[i], int(n[i])) for i, k in enumerate(names)}

    def getFilm(self, width, height):
        film = np.zeros((height, width, 5), dtype=np.float32)
        self._ok(self._L.yafaray_getFilm(self._h, film.ctypes.data_as(C.POINTER(C.c_float)), width, height), "getFilm")
        return film

    def getRenderStats(self):
        s = RenderStats()
        self._L.yafaray_getRenderStats(self._h, C.byref(s))
        return s


class TreeInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaf_refs", C.c_uint32), ("max_depth", C.c_uint32), ("n_tris", C.c_uint32),
                ("build_seconds", C.c_double), ("upload_seconds", C.c_double), ("device_bytes", C.c_uint64)]


def build_kdtree(verts, threads=0, device=False):
    """Build of the flattened kd-tree the kernels walk, on the host (no GPU needed) or on the device
    -> (nodes (n,2) u32, refs (m,) u32, bound (6,) f32, info)."""
    L = load()
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    L.yafgpu_kdtree_build.restype = C.c_void_p
    L.yafgpu_kdtree_build.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_int32]
    L.yafgpu_kdtree_build_device.restype = C.c_void_p
    L.yafgpu_kdtree_build_device.argtypes = [C.POINTER(C.c_float), C.c_int32]
    L.yafgpu_kdtree_info.argtypes = [C.c_void_p, C.POINTER(TreeInfo)]
    L.yafgpu_kdtree_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.yafgpu_kdtree_destroy.argtypes = [C.c_void_p]
    if device:
        h = L.yafgpu_kdtree_build_device(v.ctypes.data_as(C.POINTER(C.c_float)), v.shape[0])
        if not h:
            raise YafaRayError("device kd build: " + L.yafgpu_last_error().decode())
    else:
        h = L.yafgpu_kdtree_build(v.ctypes.data_as(C.POINTER(C.c_float)), v.shape[0], threads)
    info = TreeInfo()
    L.yafgpu_kdtree_info(h, C.byref(info))
    nodes = np.zeros((max(info.n_nodes, 1), 2), np.uint32)
    refs = np.zeros(max(info.n_leaf_refs, 1), np.uint32)
    bound = np.zeros(6, np.float32)
    up = C.POINTER(C.c_uint32)
    L.yafgpu_kdtree_get(h, nodes.ctypes.data_as(up), refs.ctypes.data_as(up), bound.ctypes.data_as(C.POINTER(C.c_float)))
    L.yafgpu_kdtree_destroy(h)
    return nodes[:info.n_nodes], refs[:info.n_leaf_refs], bound, info


def planes_bytes(width, height):
    return load().yafgpu_planes_bytes(width, height)


def film_combine(d_planes, d_film, width, height, stream=0):
    rc = load().yafgpu_film_combine(C.c_void_p(d_planes), C.c_void_p(d_film), width, height, C.c_void_p(stream or None))
    if rc:
        raise YafaRayError(load().yafgpu_last_error().decode())


def film_to_rgba(film):
    """Pixel::normalized (util_image_buffers.h:39-43)."""
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(w != 0, film[..., :4] / w, 0.0).astype(np.float32)
