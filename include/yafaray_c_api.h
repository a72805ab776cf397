/* yafaray_c_api.h — the drop-in boundary: libYafaRay's scene/render API as a C ABI.
 *
 * The reference snapshot has no C API; its public surface is the C++ class yafaray4::Interface
 * (include/interface/interface.h:48-139, src/interface/interface.cc) reached through the single
 * C-linkage factory getYafray__() (interface.cc:451-457).  This header flattens that class 1:1 —
 * one function per Interface method, same name, same argument meaning, the object as first
 * parameter — so a binding written against Interface ports by renaming.  Each declaration cites
 * the method it replaces.  Error convention of the reference is kept: bool results (0 = wrong
 * state / missing object), NULL from failed create*, no exceptions; additionally
 * yafaray_getLastError() returns the diagnostic the reference would have logged.
 *
 * Scope: scenes of type "triangle"; materials shinydiffusemat / glossy(as_diffuse) /
 * coated_glossy(as_diffuse) / glass / mirror / light_mat; lights arealight / pointlight; camera perspective (with depth of
 * field); background constant; integrators pathtracing / directlighting; volume integrator none.  Every method of
 * Interface has a function here: what lies outside the scope fails loudly (0 / NULL + yafaray_getLastError) at the call
 * or at render time — nothing falls back to a CPU path, and nothing is silently dropped.
 */
#ifndef YAFARAY_C_API_H
#define YAFARAY_C_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int yafaray_bool_t;
typedef struct yafaray_interface yafaray_interface_t;
/* borrowed handles, owned by the interface until clearAll (environment.h:85-95) */
typedef struct yafaray_material yafaray_material_t;
typedef struct yafaray_light yafaray_light_t;
typedef struct yafaray_texture yafaray_texture_t;
typedef struct yafaray_camera yafaray_camera_t;
typedef struct yafaray_background yafaray_background_t;
typedef struct yafaray_integrator yafaray_integrator_t;

/* ColorOutput (include/output/output.h:38-51) as a callback table; any member may be NULL.
 * putPixel receives linear RGBA of the "combined" pass, x/y relative to the render window. */
typedef struct yafaray_output
{
	void *user;
	yafaray_bool_t (*putPixel)(void *user, int num_view, int x, int y, float r, float g, float b, float a);
	void (*flush)(void *user, int num_view);
	void (*flushArea)(void *user, int num_view, int x0, int y0, int x1, int y1);
	void (*highlightArea)(void *user, int num_view, int x0, int y0, int x1, int y1);
} yafaray_output_t;

/* ProgressBar (include/common/monitor.h:29-44) */
typedef struct yafaray_progress
{
	void *user;
	void (*init)(void *user, int total_steps);
	void (*update)(void *user, int steps);
	void (*done)(void *user);
	void (*setTag)(void *user, const char *text);
} yafaray_progress_t;

/* Interface::Interface / ~Interface, interface.cc:85-103 ; getYafray__ :451-457 */
yafaray_interface_t *yafaray_createInterface(void);
void yafaray_destroyInterface(yafaray_interface_t *yi);
const char *yafaray_getLastError(const yafaray_interface_t *yi);
const char *yafaray_getVersion(void);                                            /* interface.h:117 */

/* scene state machine — Interface::startScene .. endGeometry, interface.cc:113-219 */
yafaray_bool_t yafaray_startScene(yafaray_interface_t *yi, int type);            /* interface.h:103 */
yafaray_bool_t yafaray_startGeometry(yafaray_interface_t *yi);                   /* :54 */
yafaray_bool_t yafaray_endGeometry(yafaray_interface_t *yi);                     /* :55 */
unsigned int yafaray_getNextFreeId(yafaray_interface_t *yi);                     /* :60 */
yafaray_bool_t yafaray_startTriMesh(yafaray_interface_t *yi, unsigned int id, int vertices, int triangles,
                                    yafaray_bool_t has_orco, yafaray_bool_t has_uv, int type, int obj_pass_index); /* :61 */
yafaray_bool_t yafaray_endTriMesh(yafaray_interface_t *yi);                      /* :64 */
int yafaray_addVertex(yafaray_interface_t *yi, double x, double y, double z);    /* :66 */
void yafaray_addNormal(yafaray_interface_t *yi, double nx, double ny, double nz);/* :68 */
yafaray_bool_t yafaray_addTriangle(yafaray_interface_t *yi, int a, int b, int c, const yafaray_material_t *mat); /* :69 */
yafaray_bool_t yafaray_smoothMesh(yafaray_interface_t *yi, unsigned int id, double angle); /* :72 */
yafaray_bool_t yafaray_startTriMeshPtr(yafaray_interface_t *yi, unsigned int *id, int vertices, int triangles,
                                       yafaray_bool_t has_orco, yafaray_bool_t has_uv, int type, int obj_pass_index); /* :63 */
int yafaray_addVertexWithOrco(yafaray_interface_t *yi, double x, double y, double z, double ox, double oy, double oz); /* :67, the orco overload of addVertex */
int yafaray_addUv(yafaray_interface_t *yi, float u, float v);                    /* :71 */
yafaray_bool_t yafaray_addTriangleWithUv(yafaray_interface_t *yi, int a, int b, int c, int uv_a, int uv_b, int uv_c,
                                         const yafaray_material_t *mat);          /* :70, the UV overload of addTriangle */
/* refused with a diagnostic (outside the path's scope): */
yafaray_bool_t yafaray_startCurveMesh(yafaray_interface_t *yi, unsigned int id, int vertices, int obj_pass_index);       /* :62 */
yafaray_bool_t yafaray_endCurveMesh(yafaray_interface_t *yi, const yafaray_material_t *mat, float strand_start, float strand_end, float strand_shape); /* :65 */
yafaray_bool_t yafaray_addInstance(yafaray_interface_t *yi, unsigned int base_object_id, const float *obj_to_world_16); /* :73 */
/* extension (test support): the per-triangle-corner normals smoothMesh computed, n_tris*9 floats; an all-zero triple = geometric normal */
yafaray_bool_t yafaray_getMeshCornerNormals(yafaray_interface_t *yi, unsigned int id, float *out, int n_floats);
/* extension (not in the reference): bulk form of addVertex/addTriangle for large meshes;
 * verts = n_verts*3 floats, indices = n_tris*3 ints, one material for all triangles */
yafaray_bool_t yafaray_addTriangles(yafaray_interface_t *yi, int n_verts, const float *verts, int n_tris, const int *indices,
                                    const yafaray_material_t *mat);

/* ParamMap builders — interface.cc:221-310 */
void yafaray_paramsSetPoint(yafaray_interface_t *yi, const char *name, double x, double y, double z);   /* :76 */
void yafaray_paramsSetString(yafaray_interface_t *yi, const char *name, const char *s);                 /* :77 */
void yafaray_paramsSetBool(yafaray_interface_t *yi, const char *name, yafaray_bool_t b);                /* :78 */
void yafaray_paramsSetInt(yafaray_interface_t *yi, const char *name, int i);                            /* :79 */
void yafaray_paramsSetFloat(yafaray_interface_t *yi, const char *name, double f);                       /* :80 */
void yafaray_paramsSetColor(yafaray_interface_t *yi, const char *name, float r, float g, float b, float a); /* :81 */
void yafaray_paramsSetColorArray(yafaray_interface_t *yi, const char *name, const float *rgb, yafaray_bool_t with_alpha); /* :82 */
void yafaray_paramsSetMatrix(yafaray_interface_t *yi, const char *name, const float *m16, yafaray_bool_t transpose);   /* :83, :85 (paramsSetMemMatrix): 16 floats, row major */
void yafaray_paramsSetMatrixD(yafaray_interface_t *yi, const char *name, const double *m16, yafaray_bool_t transpose); /* :84, :86 */
/* Interface::setInputColorSpace (:127; interface.cc:292-301): the colour space paramsSetColor reads its arguments in
 * ("sRGB" | "XYZ" | "LinearRGB" | "Raw_Manual_Gamma"; converted to linear RGB on entry, interface.cc:247-252) */
void yafaray_setInputColorSpace(yafaray_interface_t *yi, const char *color_space_string, float gamma_val);
void yafaray_paramsClearAll(yafaray_interface_t *yi);                                                   /* :87 */
void yafaray_paramsStartList(yafaray_interface_t *yi);                                                  /* :88 */
void yafaray_paramsPushList(yafaray_interface_t *yi);                                                   /* :89 */
void yafaray_paramsEndList(yafaray_interface_t *yi);                                                    /* :90 */

/* RenderEnvironment factories — interface.cc:312-372; dispatch on the "type" string exactly like
 * Material::factory (src/material/material.cc:36-52), Light::factory (src/light/light.cc:36-51),
 * Camera::factory (src/camera/camera.cc:34-44), Integrator::factory (src/integrator/integrator.cc:36-57) */
yafaray_light_t *yafaray_createLight(yafaray_interface_t *yi, const char *name);            /* :92 */
yafaray_texture_t *yafaray_createTexture(yafaray_interface_t *yi, const char *name);        /* :93: type "image" (TGA, HDR and PNG files; none /
                                                                                               bilinear interpolation); procedural types are refused */
yafaray_material_t *yafaray_createMaterial(yafaray_interface_t *yi, const char *name);      /* :94 */
yafaray_camera_t *yafaray_createCamera(yafaray_interface_t *yi, const char *name);          /* :95 */
yafaray_background_t *yafaray_createBackground(yafaray_interface_t *yi, const char *name);  /* :96 */
yafaray_integrator_t *yafaray_createIntegrator(yafaray_interface_t *yi, const char *name);  /* :97 */
void yafaray_clearAll(yafaray_interface_t *yi);                                             /* :101 */
/* not in the reference's Interface: an image texture over RGBA float texels the caller holds (row major, as the reference's
 * ImageHandler::getPixel(x, y) would return them), configured by the current ParamMap like createTexture; and the decoded image
 * behind a texture (the file decoders' test hook) */
yafaray_texture_t *yafaray_createTextureFromMemory(yafaray_interface_t *yi, const char *name, int width, int height, const float *rgba);
yafaray_bool_t yafaray_getTextureImage(yafaray_interface_t *yi, const char *name, int *width, int *height, float *rgba, int n_floats);
/* refused with a diagnostic (NULL / 0 and yafaray_getLastError): */
unsigned int yafaray_createObject(yafaray_interface_t *yi, const char *name);               /* :100 */
void *yafaray_createVolumeRegion(yafaray_interface_t *yi, const char *name);                /* :98 */
void *yafaray_createImageHandler(yafaray_interface_t *yi, const char *name, yafaray_bool_t add_to_table); /* :99 */
/* the calls exporters make around a render besides render() itself */
yafaray_bool_t yafaray_setLoggingAndBadgeSettings(yafaray_interface_t *yi);                 /* :104: accepted; no badge is drawn, no log file written */
yafaray_bool_t yafaray_setupRenderPasses(yafaray_interface_t *yi);                          /* :105: accepted for the combined pass alone; any other enabled pass is refused */
yafaray_bool_t yafaray_setInteractive(yafaray_interface_t *yi, yafaray_bool_t interactive); /* :106 */
int yafaray_getRenderParameters(yafaray_interface_t *yi, char *buf, int len);               /* :108: the render ParamMap as "name=value" lines; returns the bytes needed */
void yafaray_setConsoleVerbosityLevel(yafaray_interface_t *yi, const char *level);          /* :111 */
void yafaray_setLogVerbosityLevel(yafaray_interface_t *yi, const char *level);              /* :112 */
void yafaray_setParamsBadgePosition(yafaray_interface_t *yi, const char *badge_position);   /* :114 */
yafaray_bool_t yafaray_getDrawParams(yafaray_interface_t *yi);                              /* :115 */
void yafaray_printDebug(yafaray_interface_t *yi, const char *msg);                          /* :120-125 */
void yafaray_printVerbose(yafaray_interface_t *yi, const char *msg);
void yafaray_printInfo(yafaray_interface_t *yi, const char *msg);
void yafaray_printParams(yafaray_interface_t *yi, const char *msg);
void yafaray_printWarning(yafaray_interface_t *yi, const char *msg);
void yafaray_printError(yafaray_interface_t *yi, const char *msg);
void yafaray_setOutput2(yafaray_interface_t *yi, const yafaray_output_t *out_2);            /* :128: second output, in color_space2 / gamma2 */

/* Interface::render, interface.cc:411-418 = RenderEnvironment::setupScene (environment.cc:679-813)
 * + Scene::render (scene.cc:1037-1067).  Blocking.  Reads the render settings from the current
 * ParamMap (camera_name, integrator_name, volintegrator_name, background_name, width, height,
 * xstart, ystart, AA_*, filter_type, AA_pixelwidth, tile_size, tiles_order, threads, adv_*).
 * Returns void in the reference; here 1 on success, 0 on failure (yafaray_getLastError). */
yafaray_bool_t yafaray_render(yafaray_interface_t *yi, const yafaray_output_t *output, const yafaray_progress_t *progress);
void yafaray_abort(yafaray_interface_t *yi);                                                 /* :107 */
yafaray_bool_t yafaray_getRenderedImage(yafaray_interface_t *yi, int num_view, const yafaray_output_t *output); /* :109 */

/* ---- additions for measurement and multi-GPU drivers (no reference counterpart) ---- */
typedef struct yafaray_render_stats
{
	uint64_t rays_closest, rays_shadow, interior_steps, leaves, tri_tests, camera_samples, restarts;
	double tree_build_seconds, upload_seconds, render_seconds; /* render_seconds: device time of the pass */
	uint32_t kd_nodes, kd_leaf_refs, kd_max_depth, n_triangles;
	uint64_t scene_device_bytes;
} yafaray_render_stats_t;
/* The float film of the last render: height*width*5 floats {r,g,b,a,weight} — the payload of the
 * reference's film file (imagefilm.cc:1560-1657), i.e. what its ImageFilm holds before normalisation. */
yafaray_bool_t yafaray_getFilm(yafaray_interface_t *yi, float *film, int width, int height);
yafaray_bool_t yafaray_getRenderStats(yafaray_interface_t *yi, yafaray_render_stats_t *stats);
/* pixel-tile sharding: must be set before render; tile t belongs to shard t % count */
void yafaray_setShard(yafaray_interface_t *yi, int shard_index, int shard_count);
/* Multi-pass (adaptive) anti-aliasing on a sharded frame: between passes the noise detection (integrator_tiled.cc:136-258)
 * needs every rank's pixels.  `fn` receives a copy of this rank's splat planes in DEVICE memory and must sum it over all ranks
 * in place (an all-reduce; libyafaray_amd/parallel.py does it with torch.distributed = RCCL) and return 0.  Without it a sharded
 * multi-pass render is refused.  The film a rank returns stays its own share (sum them as for a one-pass render).
 * The same function carries the light counter of the serial replay below across ranks (a small table of per-tile call counts once per
 * pass): with it a sharded render of a scene with several lights makes the single-GPU render's light choices; without it, its own.
 * The exchange is a collective: a rank that fails or is aborted (yafaray_abort) before it leaves the others waiting in theirs, as with
 * any collective — abort a sharded render on EVERY rank (the flag is polled between chunks and passes, before each exchange). */
typedef int (*yafaray_plane_exchange_t)(void *user, float *d_values, uint64_t n_floats);
void yafaray_setPlaneExchange(yafaray_interface_t *yi, yafaray_plane_exchange_t fn, void *user);
/* ---- multi-GPU: one process per GPU, RCCL over xGMI, no Python on the data path (yafaray_reduce.cpp) -------------------------
 * The reference's only multi-machine mechanism sums film files (imagefilm.cc:1467-1557: col += , weight +=); here each rank
 * renders its tile shard into a full-frame [H][W][5] float film and ONE ncclReduce(sum) assembles the frame on `root`.
 * Sequence for N ranks: rank 0 yafaray_commGetUniqueId -> the host hands the 128 bytes to every rank (file,
 * environment, MPI, a socket) -> each rank yafaray_commCreate(id, rank, N, device) -> yafaray_setShard(yi, rank, N) ->
 * yafaray_setComm(yi, comm) -> render into device memory -> yafaray_reduceFilm.  RCCL is loaded at run time; without it these
 * functions fail with a message in yafaray_commLastError (there is no host-staged fallback). */
#define YAFARAY_COMM_ID_BYTES 128
typedef struct yafaray_comm yafaray_comm_t;
yafaray_bool_t yafaray_commGetUniqueId(char id[YAFARAY_COMM_ID_BYTES]);
yafaray_comm_t *yafaray_commCreate(const char id[YAFARAY_COMM_ID_BYTES], int rank, int world, int device);
void yafaray_commDestroy(yafaray_comm_t *comm);
int yafaray_commRank(const yafaray_comm_t *comm);
int yafaray_commWorld(const yafaray_comm_t *comm);
const char *yafaray_commLastError(void);
const char *yafaray_commBackend(void);      /* which librccl was bound ("" when none) */
/* sum of the ranks' device films onto `root`, in place, asynchronously on `stream` (ncclReduce); n_floats = height*width*5 */
yafaray_bool_t yafaray_reduceFilm(yafaray_comm_t *comm, float *d_film, uint64_t n_floats, int root, void *stream);
/* sum over all ranks, in place on every rank (ncclAllReduce) */
yafaray_bool_t yafaray_allReduce(yafaray_comm_t *comm, float *d_values, uint64_t n_floats, void *stream);
/* the communicator as a yafaray_plane_exchange_t (user = the communicator) */
int yafaray_commExchange(void *user, float *d_values, uint64_t n_floats);
/* attach a communicator to an interface: yafaray_setPlaneExchange(yi, yafaray_commExchange, comm); NULL detaches */
void yafaray_setComm(yafaray_interface_t *yi, yafaray_comm_t *comm);

/* Exact replay of the reference's serial render state (on by default): the per-tile Random that Russian roulette draws from
 * (integrator_tiled.cc:319, seeded from libc rand() as the last Material / ObjectGeometric constructor left it;
 * integrator_path_tracer.cc:282-288) and the estimateOneDirectLight counter (integrator_montecarlo.cc:62-76), both as a
 * single-threaded render with tiles_order = linear consumes them.  Costs a record pass of closest-hit rays per chunk when a
 * render consumes either (RR active, or more than one light).  Off: per-sample streams — the same estimator, other pixels. */
void yafaray_setSerialReplay(yafaray_interface_t *yi, yafaray_bool_t on);
/* srand() seed and values consumed since, of the libc stream the next render's tile seeds continue (-1: nothing created yet) */
void yafaray_getRandState(yafaray_interface_t *yi, int *srand_seed, int *skip);
/* An embedder that calls libc srand(seed) itself (and consumes `skip` values) between scene construction and render says so
 * here — the reference would simply continue from that state at integrator_tiled.cc:319.  Holds until the next material or
 * object is created (their constructors call srand again).  seed < 0: back to the constructors' state. */
void yafaray_setRandState(yafaray_interface_t *yi, int srand_seed, int skip);
/* Two-step render for drivers that own device memory and streams (bench.py, RCCL reduce):
 * prepare = setupScene + Scene::update (tree build, upload); renderPass launches one pass
 * asynchronously on `stream` into caller-owned device memory d_planes (yafgpu_planes_bytes). */
yafaray_bool_t yafaray_prepareRender(yafaray_interface_t *yi);
yafaray_bool_t yafaray_renderPassDevice(yafaray_interface_t *yi, float *d_planes, void *d_counters, void *stream);
yafaray_bool_t yafaray_getRenderSize(yafaray_interface_t *yi, int *width, int *height);
/* Scene::intersect / Scene::isShadowed (scene.cc:896-994) on host ray batches, after prepareRender:
 * rays = n*8 floats {from.xyz, dir.xyz, tmin, tmax (<0 = infinite)}; tri = -1 on a miss */
yafaray_bool_t yafaray_intersectRays(yafaray_interface_t *yi, int n, const float *rays, int *tri, float *t, float *bary);
yafaray_bool_t yafaray_shadowRays(yafaray_interface_t *yi, int n, const float *rays, int *shadowed);
/* per-kernel device timing of the next renderPassDevice (yafgpu_set_profiling / yafgpu_get_profile):
 * ms[4]/launches[4] = closest-hit traversal, any-hit traversal, shading, other */
yafaray_bool_t yafaray_setProfiling(yafaray_interface_t *yi, yafaray_bool_t enable);
yafaray_bool_t yafaray_getKernelProfile(yafaray_interface_t *yi, double ms[4], uint64_t launches[4]);
/* consecutive renderPassDevice calls of independent passes on two internal streams (yafgpu_scene_set_pass_pipelining, include/yafgpu.h):
 * -1 by size (default), 0 off, 1 on.  What the caller sees — planes, counters — is written on the caller's stream, in call order, either way. */
yafaray_bool_t yafaray_setPassPipelining(yafaray_interface_t *yi, int mode);
/* device-side component probe (yafgpu_probe) against the prepared scene's materials/lights/camera */
yafaray_bool_t yafaray_probe(yafaray_interface_t *yi, int op, int n, const float *in, int n_in, float *out, int n_out);

/* XML scene loader: src/loader_xml/loader_xml.cc + src/common/import_xml.cc drive the same calls
 * from a scene file.  Parses `path` and leaves the interface ready for yafaray_render. */
yafaray_bool_t yafaray_loadXml(yafaray_interface_t *yi, const char *path);

#ifdef __cplusplus
}
#endif
#endif
