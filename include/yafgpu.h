/* yafgpu.h — the narrow host<->device seam of the MI355X path-tracing core.
 *
 * This is the C ABI *inside* the drop-in ("second, narrower C ABI"): the wide,
 * Interface-shaped API in yafaray_c_api.h flattens a scene into the POD blocks below and calls
 * these entry points.  Plain pointers and sizes only; device buffers are raw HIP device pointers
 * so that a caller may own them (e.g. a torch tensor's data_ptr()) or let the library allocate.
 *
 * What each entry point replaces in the reference:
 *   yafgpu_scene_create   <- Scene::update()                (src/common/scene.cc:784-894): kd-tree
 *                            build (TriKdTree ctor, kdtree_triangle.cc:76-157) + Triangle cached
 *                            values (triangle.h:197-207) + Light::init
 *   yafgpu_render_tiles   <- TiledIntegrator::renderPass / renderTile
 *                            (src/integrator/integrator_tiled.cc:261-521) with
 *                            PathIntegrator::integrate (integrator_path_tracer.cc:112-347) and
 *                            ImageFilm::addSample (src/common/imagefilm.cc:925-1015) inside
 *   yafgpu_film_combine   <- the per-pixel sum that addSample performs across neighbouring samples
 *   yafgpu_trace_rays     <- Scene::intersect / Scene::isShadowed (scene.cc:896-994) on ray batches
 */
#ifndef YAFGPU_H
#define YAFGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YAFGPU_FILM_CHANNELS 5   /* r,g,b,a,weight : Pixel, include/utility/util_image_buffers.h:36-48 */
#define YAFGPU_FILM_PLANES   4   /* own, right, down, diagonal splat planes */

enum { YAFGPU_MAT_SHINYDIFFUSE = 0, YAFGPU_MAT_GLOSSY = 1, YAFGPU_MAT_LIGHT = 2, YAFGPU_MAT_GLASS = 3, YAFGPU_MAT_MIRROR = 4,
       YAFGPU_MAT_COATED_GLOSSY = 5, /* glossy's fields + mirror_color, mirror_strength, glass_ior = IOR, c_flags[0..2], n_bsdf */
       YAFGPU_MAT_ROUGH_GLASS = 6 /* glass's fields (glass_ior, filter_color, mirror_color, fake_shadow, beer_sigma) + rg_a2: RoughGlassMaterial, material_rough_glass.cc */ };
enum { YAFGPU_LIGHT_AREA = 0, YAFGPU_LIGHT_POINT = 1 };
enum { YAFGPU_INTEGRATOR_PATH = 0, YAFGPU_INTEGRATOR_DIRECT = 1 };
enum { YAFGPU_FILTER_BOX = 0, YAFGPU_FILTER_MITCHELL = 1, YAFGPU_FILTER_GAUSS = 2, YAFGPU_FILTER_LANCZOS = 3 };

/* A material after its factory()/config() ran on the host (material_shiny_diffuse.cc:46-92,
 * material_glossy.cc:32-50, material_simple.cc:36-39).  64 floats, 16-byte aligned. */
typedef struct yafgpu_material
{
	int32_t type, visibility, receive_shadows, flat;
	uint32_t bsdf_flags;
	int32_t n_bsdf;
	uint32_t c_flags[4];
	int32_t c_index[4];
	/* shinydiffuse */
	float diffuse_color[3], mirror_color[3], emit_color[3];
	float mirror_strength, transparency_strength, translucency_strength, diffuse_strength;
	float transmit_filter, ior_squared;
	int32_t is_mirror, is_transparent, is_translucent, is_diffuse, has_fresnel;
	int32_t use_oren;
	float oren_a, oren_b;
	/* glossy */
	float gloss_color[3], diff_color[3];
	float exponent, reflectivity, diffuse;
	int32_t as_diffuse, with_diffuse;
	int32_t anisotropic;           /* the Ashikhmin-Shirley lobe with exp_u / exp_v instead of Blinn (material_utils_microfacet.h:38-87) */
	float exp_u, exp_v;
	/* light material */
	float light_col[3];
	int32_t double_sided;
	/* glass (material_glass.cc:32-49; mirror_color = specular reflection colour) and mirror (mirror_color = colour * reflect) */
	float glass_ior;
	float filter_color[3];         /* transmit_filter * filter_color + (1 - transmit_filter) */
	int32_t fake_shadow;
	uint32_t tm_flags;             /* the transmission lobe: Filter|Transmit with fake shadows, else Specular|Transmit */
	int32_t has_vol_i;             /* glass "absorption": vol_i_ = BeerVolumeHandler (material_glass.cc:371-398), bsdf_flags has Volumetric */
	float beer_sigma[3];           /* its sigma_a_ = -log(absorption) / absorption_dist (volumehandler_beer.cc:28-35) */
	/* shader nodes: the material's nodes are nodes[node_first .. node_first + n_nodes) of the scene's node array,
	   in evaluation order (NodeMaterial::solveNodesOrder, material_node.cc:88-108); each shader slot names the node it reads
	   (index into that range) or -1.  shinydiffusemat slots, material_shiny_diffuse.cc:697-724. */
	int32_t node_first, n_nodes;
	int32_t sh_diffuse, sh_mirror_color, sh_mirror, sh_transparency, sh_translucency, sh_sigma_oren, sh_diffuse_refl, sh_ior;
	int32_t sh_glossy, sh_glossy_reflect, sh_exponent;      /* glossy / coated_glossy: glossy_shader, glossy_reflect_shader, exponent_shader (material_glossy.cc:504-511) */
	int32_t sh_filter_color;       /* glass: filter_color_shader (material_glass.cc:419-422) */
	int32_t bump_first, n_bump, sh_bump;   /* bump mapping: nodes[bump_first .. bump_first + n_bump) = what the bump shader reaches, in evaluation order
	                                          (NodeMaterial::evalBump, material_node.cc:132-139); sh_bump its index in that range; n_bump 0 = none */
	int32_t additional_depth;      /* Material::additional_depth_: recursiveRaytrace may go this much deeper below this material (integrator_montecarlo.cc:791) */
	float transp_bias_factor;      /* shinydiffusemat transparentbias_factor / transparentbias_multiply_raydepth (integrator_montecarlo.cc:1003-1011) */
	int32_t transp_bias_mult;
	float transp_ior;              /* glass: the index getTransparency's fresnel sees — ior_, or the IOR shader's value alone (material_glass.cc:223, sic) */
	float ior_base;                /* ior_ (the IOR shader adds to it, :258-262; coated_glossy: material_coated_glossy.cc:147) */
	float rg_a2;                   /* rough_glass: a_2_ = alpha^2 of its GGX lobe, alpha = max(1e-4, min(alpha / 2, 1)) (material_rough_glass.cc:33-36, :362) */
	float emit_strength;           /* emit_strength_ (emit() with a diffuse shader: colour * emit_strength_, :300) */
	int32_t has_diffuse_refl;      /* the fields below are only set in a resolved per-hit copy of the record (mat_resolve) */
	float diffuse_refl;
	int32_t oren_tex;              /* orenNayar with a texture's sigma: A and B in double (:230-235) */
	double oren_ad, oren_bd;
} yafgpu_material;

/* ImageTexture (src/texture/texture_image.cc) over float texels that hold what ImageHandler::getPixel returns (the host decoded
   the file, linearised its colours and pushed them through the image buffer's storage format) */
typedef struct yafgpu_texture
{
	int32_t width, height;
	uint32_t texel_first;          /* first texel in the scene's texel array (float4 units), row-major, row y = the handler's row y */
	int32_t interpolate;           /* 0 none, 1 bilinear */
	int32_t clip;                  /* TexClipMode: 0 extend, 1 clip, 2 clipcube, 3 repeat, 4 checker */
	int32_t xrepeat, yrepeat, rot90, mirror_x, mirror_y, checker_even, checker_odd;
	float checker_dist;
	int32_t cropx, cropy;
	float cropminx, cropmaxx, cropminy, cropmaxy;
	int32_t adj_set, adj_clamp;
	float adj_int, adj_con, adj_sat, adj_hue, adj_r, adj_g, adj_b;    /* adj_hue already divided by 60 (Texture::setAdjustments) */
	int32_t color_space;           /* for getRawColor / getFloat: 0 sRGB, 1 XYZ, 2 LinearRGB, 3 RawManualGamma */
	float gamma;
	int32_t normalmap;             /* texture_image.cc:705: a normal map (TextureMapperNode::evalDerivative reads the normal from its raw colour) */
} yafgpu_texture;

enum { YAFGPU_NODE_TEXTURE_MAPPER = 0, YAFGPU_NODE_VALUE = 1, YAFGPU_NODE_MIX = 2, YAFGPU_NODE_LAYER = 3 };
/* one shader node after its factory() / configInputs() ran on the host (shader_node_basic.cc, shader_node_layer.cc); node
   references are indices into the material's node range, -1 = not connected */
typedef struct yafgpu_node
{
	int32_t type;
	/* texture_mapper */
	int32_t texture, texco, mapping, map_x, map_y, map_z, do_scalar;
	float scale[3], offset[3];     /* offset already doubled (TextureMapperNode::factory :411) */
	float mtx[16];
	/* value */
	float color[4]; float value;
	/* mix */
	int32_t mode; float cfactor; int32_t input1, input2, factor;
	float col1[4], col2[4];
	/* layer */
	int32_t input, upper; uint32_t texflag;
	float colfac, valfac, def_val, upper_val;
	float def_col[4], upper_col[4];
	int32_t do_color, do_scalar_l, color_input, use_alpha;
	/* bump mapping (texture_mapper): what TextureMapperNode::setup leaves (shader_node_basic.cc:34-59): the texel steps 1 / width, 1 / height and
	   bump_strength / |scale| / 100 */
	float d_u, d_v, bump_str;
} yafgpu_node;

/* A light after its constructor ran on the host (light_area.cc:34-52, light_point.cc:28-36) */
typedef struct yafgpu_light
{
	int32_t type, samples, cast_shadows, pad0;
	float corner[3], c2[3], c3[3], c4[3], to_x[3], to_y[3], fnormal[3];
	float color[3];
	float area;
	float position[3];
	float pad1[2];
} yafgpu_light;

/* PerspectiveCamera after setAxis (camera_perspective.cc:60-74) */
typedef struct yafgpu_camera
{
	float position[3], vto[3], vup[3], vright[3];
	float near_p[3], near_n[3], far_p[3], far_n[3];
	int32_t resx, resy;
	/* depth of field (camera_perspective.cc:33-54,60-74): aperture 0 = pinhole */
	float aperture, dof_distance;
	int32_t bokeh_type;            /* BokehType: 0 disk1, 1 disk2, 3 triangle, 4 square, 5 pentagon, 6 hexagon, 7 ring */
	int32_t bokeh_bias;            /* 0 none, 1 center, 2 edge */
	float bokeh_rotation;          /* degrees */
	float dof_rt[3], dof_up[3];    /* aperture * cam_x, aperture * cam_y */
	float ls[16];                  /* polygon corner table; filled by yafgpu_scene_create from bokeh_type / bokeh_rotation */
	/* for the `window` / `normal` texture coordinates (PerspectiveCamera::screenproject, Camera::getAxis) */
	float cam_x[3], cam_y[3], cam_z[3];
	float focal_distance, aspect_ratio;
} yafgpu_camera;

typedef struct yafgpu_scene_desc
{
	int32_t n_tris;
	const float *verts;          /* n_tris*9: a,b,c */
	const int32_t *tri_mat;      /* n_tris */
	const float *vnormals;       /* NULL or n_tris*9; an all-zero triple means "use the geometric normal" */
	/* texture coordinates per triangle corner (Triangle::getSurface, triangle.cc:46-79): uv = n_tris*6 floats or NULL,
	   orco = n_tris*9 floats or NULL; only read when some material has shader nodes */
	const float *tri_uv, *tri_orco;
	int32_t n_textures; const yafgpu_texture *textures;
	uint64_t n_texels; const float *texels;      /* n_texels * 4 floats */
	int32_t n_nodes; const yafgpu_node *nodes;
	int32_t n_materials;
	const yafgpu_material *materials;
	int32_t n_lights;
	const yafgpu_light *lights;
	yafgpu_camera camera;
	int32_t build_threads;       /* host threads for the kd build; <=0: hardware concurrency */
	int32_t build_on_device;     /* kd-tree builder: 1 = on the GPU (kdtree_build_device.hip), -1 = host builder,
	                                0 = by size (GPU from 65 536 triangles on).  Both give the same query results.
	                                The environment variable YAFGPU_BUILD=device|host overrides it. */
} yafgpu_scene_desc;

typedef struct yafgpu_render_params
{
	int32_t integrator;            /* YAFGPU_INTEGRATOR_* */
	int32_t path_samples, bounces, rr_min_bounces, no_recursive, bg_transp, bg_transp_refract;
	int32_t width, height, xstart, ystart;
	int32_t aa_minsamples;
	float aa_pixelwidth;           /* reconstruction filter width in pixels (AA_pixelwidth) */
	int32_t filter_type;           /* YAFGPU_FILTER_*: box | mitchell | gauss | lanczos (ImageFilm::FilterType, imagefilm.cc:155-163) */
	int32_t tile_size;
	uint32_t base_sampling_offset;
	int32_t shadow_bias_auto; float shadow_bias;
	int32_t min_raydist_auto; float min_raydist;
	float aa_light_sample_multiplier;
	float background[3]; int32_t has_background;
	/* pixel-tile sharding across GPUs: tile t is rendered iff t % shard_count == shard_index */
	int32_t shard_index, shard_count;
	/* one pass of the multi-pass schedule (TiledIntegrator::renderPass, integrator_tiled.cc:261-307); all zero for a
	   single-pass render.  aa_minsamples is the sample count of THIS pass. */
	int32_t multi_pass;            /* AA_passes > 1: sub-pixel positions from riVdC / riS (integrator_tiled.cc:394-398) */
	uint32_t pass_offset;          /* samples per pixel taken by the earlier passes (renderPass's `offset`) */
	int32_t accumulate;            /* add to the planes instead of starting from zero */
	float aa_clamp_samples;        /* ImageFilm::addSample clampProportionalRgb (imagefilm.cc:975); 0 = off */
	int32_t transp_shad;           /* tr_shad_: shadow rays are filtered by transparent materials instead of blocked
	                                  (TriKdTree::intersectTs, kdtree_triangle.cc:983-1162) */
	int32_t shadow_depth;          /* s_depth_: more distinct transparent surfaces than this along a shadow ray block it; at most 8 */
	int32_t raydepth;              /* r_depth_ of recursiveRaytrace (integrator_montecarlo.cc:791): levels of perfect specular
	                                  reflection / filtered transmission followed from a camera hit; at most 7 */
	int32_t serial_replay;         /* 1: replay the reference's serial state as its single-threaded render consumes it — the per-tile
	                                  MWC stream of Russian roulette (integrator_tiled.cc:319, integrator_path_tracer.cc:282-288) and
	                                  the estimateOneDirectLight counter (integrator_montecarlo.cc:62-76) — with a record pass, a scan
	                                  in the reference's sample order and the final pass (yafgpu_wavefront.h, WfArgs::replay).
	                                  0: per-sample streams (same distribution, not the reference's pixels). */
	const int32_t *tile_rand;      /* HOST pointer, one value per tile of the frame (row-major tile order): the libc rand() value
	                                  TiledIntegrator::renderTile seeds that tile's Random with in THIS pass; NULL = 0 for all */
	const uint8_t *resample_mask;  /* HOST pointer, width*height bytes, row-major in window coordinates: the pixels that
	                                  get samples in this pass (ImageFilm::doMoreSamples, imagefilm.cc:917-920); NULL = all */
	int32_t trace_caustics;        /* PathIntegrator::trace_caustics_ (integrator_path_tracer.cc:85): caustic_type "path" — the factory's default when the
	                                  parameter is absent — or "both": after a bounce through a specular, glossy or filter lobe the next vertex shows its
	                                  lights and adds its emission (:252-253, :290).  0 = caustic_type "none" */
} yafgpu_render_params;

/* Scene::setAntialiasing (scene.cc:761-778; defaults environment.cc:682-695) */
typedef struct yafgpu_aa_schedule
{
	int32_t passes, inc_samples;
	float threshold, resampled_floor;
	float sample_multiplier_factor, light_sample_multiplier_factor;
	int32_t detect_color_noise;
	int32_t dark_detection_type;   /* 0 none, 1 linear, 2 curve */
	float dark_threshold_factor;
	int32_t variance_edge_size, variance_pixels;
	/* libc state behind the tile seeds (used with yafgpu_render_params::serial_replay): srand(rand_srand) by the last
	   Material / ObjectGeometric constructor (material.cc:56, object_geom.cc:42), rand_skip values consumed by its colour
	   loop since; then one rand() per tile per pass that runs.  rand_srand < 0: every tile draws 0. */
	int32_t rand_srand, rand_skip;
} yafgpu_aa_schedule;

typedef struct yafgpu_counters   /* device atomics, accumulated per launch */
{
	uint64_t rays_closest, rays_shadow, interior_steps, leaves, tri_tests, camera_samples, restarts;
	uint64_t wave_rounds;   /* YAFGPU_STATS only: node-step rounds executed by waves | triangle rounds << 32 (SIMT efficiency of the traversal) */
} yafgpu_counters;

typedef struct yafgpu_tree_info
{
	uint32_t n_nodes, n_leaf_refs, max_depth, n_tris;
	double build_seconds, upload_seconds;
	uint64_t device_bytes;
} yafgpu_tree_info;

typedef struct yafgpu_scene yafgpu_scene_t;

/* all functions return 0 on success, a negative code on failure; yafgpu_last_error() describes it */
const char *yafgpu_last_error(void);
int yafgpu_device_count(void);
int yafgpu_set_device(int device);

int yafgpu_scene_create(const yafgpu_scene_desc *desc, yafgpu_scene_t **out);
void yafgpu_scene_destroy(yafgpu_scene_t *scene);
int yafgpu_scene_info(const yafgpu_scene_t *scene, yafgpu_tree_info *info);

/* bytes of one plane set: YAFGPU_FILM_PLANES*height*width*YAFGPU_FILM_CHANNELS floats */
uint64_t yafgpu_planes_bytes(int32_t width, int32_t height);

/* Render every tile of this shard.  d_planes: device pointer to yafgpu_planes_bytes() bytes (zeroed by
 * the call); d_counters: device pointer to a yafgpu_counters (may be NULL); stream: hipStream_t or NULL.
 * Asynchronous on `stream`. */
int yafgpu_render_tiles(yafgpu_scene_t *scene, const yafgpu_render_params *rp, float *d_planes,
                        yafgpu_counters *d_counters, void *stream);

/* d_film[h][w][5] = own + right(x-1) + down(y-1) + diag(x-1,y-1), in that fixed order */
int yafgpu_film_combine(const float *d_planes, float *d_film, int32_t width, int32_t height, void *stream);

/* Convenience: allocate, render, combine, copy the film to host memory, synchronise. */
/* TiledIntegrator::render (integrator_tiled.cc:116-258): pass 0 with rp->aa_minsamples samples, then aa->passes - 1
   adaptive passes; between passes ImageFilm::nextPass (imagefilm.cc:270-480) picks the pixels to resample from the
   film so far.  aa == NULL or aa->passes <= 1: one pass.  resampled[] (optional, aa->passes entries) receives the
   number of pixels each pass sampled. */
int yafgpu_render_passes_to_host(yafgpu_scene_t *scene, const yafgpu_render_params *rp, const yafgpu_aa_schedule *aa,
                                 float *h_film, yafgpu_counters *h_counters, int32_t *resampled);
int yafgpu_render_to_host(yafgpu_scene_t *scene, const yafgpu_render_params *rp, float *h_film, yafgpu_counters *h_counters);

/* Ray batches (host pointers): rays = n*8 floats {from.xyz, dir.xyz, tmin, tmax}; closest-hit writes
 * tri (int32, -1 = miss), t, and barycentrics b0,b1,b2 (n*3); any-hit writes 0/1 into shadowed. */
int yafgpu_trace_closest(yafgpu_scene_t *scene, int32_t n, const float *rays, int32_t *tri, float *t, float *bary);
int yafgpu_trace_shadow(yafgpu_scene_t *scene, int32_t n, const float *rays, int32_t *shadowed);

/* Per-kernel device timing of the NEXT render pass (HIP events around every launch on the pass's own
 * stream; the pass then synchronises after each launch, so it is a measurement mode, not a fast path).
 * slots: 0 closest-hit traversal, 1 any-hit traversal, 2 shading, 3 other (ray generation, film). */
int yafgpu_set_profiling(yafgpu_scene_t *scene, int32_t enable);
/* Pass pipelining: consecutive yafgpu_render_tiles calls whose passes do not depend on each other's film (no resample mask, no serial-state
 * replay, no recursion, one chunk) run their path work on two internal streams with a buffer set each, so that one pass's launch tails are
 * filled by the other's launches; everything the caller sees (the planes, the counters) is still written on the caller's stream, in call
 * order.  mode -1 (default): on for passes of up to 12 Mi paths, where it pays (an eighth of the metric frame: +18 %); 0: off; 1: on. */
int yafgpu_scene_set_pass_pipelining(yafgpu_scene_t *scene, int32_t mode);
/* glibc's rand() after srand(seed) (TYPE_3 additive feedback generator, restated; pinned against libc in the tests):
 * out[k] = the k-th value.  The tile seeds of a render are drawn from it (integrator_tiled.cc:319). */
void yafgpu_glibc_rand(uint32_t seed, int32_t count, int32_t *out);
/* Scene::abort (scene.cc:75-89): the render entry points poll *flag between wavefront chunks and between passes and
 * return -30 ("aborted") once it is non-zero.  The flag stays owned by the caller; NULL detaches it. */
int yafgpu_scene_set_abort_flag(yafgpu_scene_t *scene, const volatile int32_t *flag);
/* Multi-pass (adaptive) anti-aliasing on a sharded frame: the noise detection between passes (integrator_tiled.cc:136-258) reads
 * every pixel, other ranks' tiles included.  With an exchange function attached, yafgpu_render_passes_to_host hands it a COPY of
 * this rank's four splat planes (device memory, n_floats values) after every pass that is followed by a detection step; the
 * function sums the copies over all ranks in place (an all-reduce: RCCL over xGMI) and returns 0.  Every plane element is
 * written by exactly one rank, so the sums are exact (x + 0) and every rank derives the single-GPU render's resample mask.
 * The rank's own planes, and the film it returns at the end, stay its own share.
 * The same function carries the reference's serial light counter across ranks (path tracing with more than one light,
 * estimateOneDirectLight's correlative_sample_number_, integrator_montecarlo.cc:62-76): once per pass every rank — with or without
 * tiles of its own — hands it a table of 2 x (tiles of the frame) floats holding its own tiles' call counts; see DESIGN.md §6. */
typedef int (*yafgpu_exchange_fn)(void *user, float *d_values, uint64_t n_floats);
int yafgpu_scene_set_exchange(yafgpu_scene_t *scene, yafgpu_exchange_fn fn, void *user);
int yafgpu_get_profile(const yafgpu_scene_t *scene, double ms[4], uint64_t launches[4]);

/* Component probe for tests: evaluates device-side leaf functions (fast-math, QMC, camera, lights,
 * material eval/pdf/sample) on n items of n_in floats each, writing n_out floats each; `op` selects the
 * function (see probe_kernel in yafgpu_device.hip).  Lets the device code be pinned against the
 * reference's own golden vectors independently of any render. */
int yafgpu_probe(yafgpu_scene_t *scene, int32_t op, int32_t n, const float *in, int32_t n_in, float *out, int32_t n_out);

/* Host-only kd-tree build (no device needed): the builder Scene::update would run (scene.cc:818 ->
 * TriKdTree ctor, kdtree_triangle.cc:76-157), exposed so that host tests can check the tree the
 * kernels will walk.  nodes = n_nodes*2 uint32 (kdtree_build.h layout), refs = n_leaf_refs uint32. */
typedef struct yafgpu_kdtree yafgpu_kdtree_t;
/* the same on the GPU (needs a device); NULL on failure, yafgpu_last_error() says why */
yafgpu_kdtree_t *yafgpu_kdtree_build_device(const float *verts, int32_t n_tris);
yafgpu_kdtree_t *yafgpu_kdtree_build(const float *verts, int32_t n_tris, int32_t threads);
void yafgpu_kdtree_info(const yafgpu_kdtree_t *tree, yafgpu_tree_info *info);
void yafgpu_kdtree_get(const yafgpu_kdtree_t *tree, uint32_t *nodes, uint32_t *refs, float bound6[6]);
void yafgpu_kdtree_destroy(yafgpu_kdtree_t *tree);

/* kd-tree built on the host, downloadable for inspection/tests: nodes = n_nodes*2 uint32, refs = n_leaf_refs uint32 */
int yafgpu_scene_get_tree(const yafgpu_scene_t *scene, uint32_t *nodes, uint32_t *refs, float bound6[6]);

#ifdef __cplusplus
}
#endif
#endif
