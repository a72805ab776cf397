/* TEST INFRASTRUCTURE — CPU oracle for the libYafaRay path-tracing hot path.
 *
 * This is a plain-C restatement of the reference's algorithm (file:line citations are in
 * yaf_oracle.c).  It is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libyafaray_amd) never links or calls it.
 *
 * Pinning status:
 *   - fast-math, QMC, createCs/sampleCosHemisphere, Bound::cross, perspective camera (pinhole and
 *     depth of field), area/point lights, shinydiffuse/glossy/coated-glossy/glass/rough-glass/mirror/light
 *     materials (eval, pdf, sample — rough glass also its two-direction sample —, getSpecular, getAlpha,
 *     getTransparency), BeerVolumeHandler (glass absorption), image textures, shader nodes and image decoders
 *     are pinned bit-for-bit against the reference's own sources compiled here (oracle/_ref, IEEE build) and
 *     to ~1e-4 against the reference's -ffast-math release flags (tests/golden/ref_components_*.json,
 *     ref_textures_*.json).
 *   - TiledIntegrator::render / renderPass / renderTile, PathIntegrator::integrate (path caustics included),
 *     DirectLightIntegrator::integrate, doLightEstimation, estimateAll / OneDirectLight, recursiveRaytrace
 *     (specular branch, both cases of the glossy branch) and the per-tile roulette stream are pinned bit-for-bit
 *     against the reference's own integrator sources compiled here (oracle/ref_harness/ref_integrator.cc:
 *     every sample handed to ImageFilm::addSample, every geometry query and both query counts of sixteen
 *     cases, tests/golden/ref_integrator_*.json).  The harness provides the bodies of the few Scene:: /
 *     ImageFilm:: members the integrators call (its header lists them).
 *   - kd traversal (intersect / intersectS / intersectTs), Triangle::intersect / getSurface and the film's
 *     filter table are restated from the source, but the reference's implementation of them is NOT buildable
 *     under this project's rules (cmake-generated header): for those rows parity is UNPINNED except through
 *     the reference's expected render of test01 (tests/golden/test01_expected.png).
 */
#ifndef YAF_ORACLE_H
#define YAF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { YOR_MAT_SHINYDIFFUSE = 0, YOR_MAT_GLOSSY = 1, YOR_MAT_LIGHT = 2,
       YOR_MAT_GLASS = 3,   /* color = filter_color, mirror_color, ior = IOR, sigma = transmit_filter (double), fresnel_effect = fake_shadows */
       YOR_MAT_MIRROR = 4,  /* color, specular_reflect = reflect */
       YOR_MAT_COATED_GLOSSY = 5, /* glossy's fields + mirror_color, specular_reflect = mirror strength, ior = IOR */
       YOR_MAT_ROUGH_GLASS = 6 /* color = filter_color, mirror_color, ior = IOR, transmit_filter (float), fresnel_effect = fake_shadows, rough_alpha = alpha, absorption */ };
enum { YOR_LIGHT_AREA = 0, YOR_LIGHT_POINT = 1 };
enum { YOR_INTEGRATOR_PATH = 0, YOR_INTEGRATOR_DIRECT = 1 };
enum { YOR_FILTER_BOX = 0, YOR_FILTER_MITCHELL = 1, YOR_FILTER_GAUSS = 2, YOR_FILTER_LANCZOS = 3 };

/* Parameter-level material description: the values a reference factory() would read from its
 * ParamMap (material_shiny_diffuse.cc:599-665, material_glossy.cc:407-472,
 * material_simple.cc:63-73).  Derived state (config(), flags) is computed inside. */
typedef struct yor_material_desc
{
	int32_t type;
	int32_t visibility;        /* 0 normal, 1 no_shadows, 2 shadow_only, 3 invisible */
	int32_t receive_shadows;
	int32_t flat_material;
	/* shinydiffusemat */
	float color[3];
	float mirror_color[3];
	float diffuse_reflect;
	float specular_reflect;
	float transparency;
	float translucency;
	float emit;
	float ior;
	int32_t fresnel_effect;
	float transmit_filter;
	int32_t oren_nayar;
	float pad0;
	double sigma;
	/* glossy */
	float glossy_color[3];
	float diffuse_color[3];
	float glossy_reflect;
	float glossy_diffuse_reflect;
	float exponent;
	int32_t as_diffuse;
	/* light_mat */
	float light_color[3];
	float light_power;
	int32_t double_sided;
	int32_t anisotropic;      /* glossy / coated_glossy "anisotropic": the Ashikhmin-Shirley lobe with exp_u / exp_v (below) instead of Blinn */
	/* glass "absorption" / "absorption_dist": a BeerVolumeHandler inside the material (material_glass.cc:371-398) */
	float absorption[3];
	int32_t has_absorption;
	double absorption_dist;
	/* shader nodes: the material's node list in EVALUATION order (NodeMaterial::solveNodesOrder,
	 * material_node.cc:88-108: dependencies before dependants) and the node each shader slot reads (-1: none).
	 * shinydiffusemat slots, material_shiny_diffuse.cc:697-724. */
	int32_t n_nodes;
	int32_t sh_diffuse, sh_mirror_color, sh_mirror, sh_transparency, sh_translucency, sh_sigma_oren, sh_diffuse_refl, sh_ior;
	int32_t pad2;
	const struct yor_node_desc *nodes;
	float exp_u, exp_v;       /* material_glossy.cc:464-472, material_coated_glossy.cc:529-537 */
	int32_t sh_glossy, sh_glossy_reflect, sh_exponent, sh_filter_color;      /* glossy / coated_glossy slots: glossy_shader, glossy_reflect_shader, exponent_shader; glass: filter_color_shader */
	int32_t additional_depth;  /* "additionaldepth": recursiveRaytrace may go this much deeper below this material (integrator_montecarlo.cc:791) */
	float transp_bias_factor;  /* shinydiffusemat "transparentbias_factor" / "transparentbias_multiply_raydepth" (:1003-1011) */
	int32_t transp_bias_mult;
	int32_t n_bump_nodes;      /* bump mapping: the nodes the bump shader reaches, in evaluation order, and its index among them (material_node.cc:132-139) */
	int32_t sh_bump, pad5;
	const struct yor_node_desc *bump_nodes;
	float rough_alpha;         /* rough_glass "alpha" as given (the factory halves and clamps it, material_rough_glass.cc:362) */
	int32_t pad6;
} yor_material_desc;

/* ImageTexture (texture_image.cc) over texels as ImageBuffer::getColor returns them (imagehandler.h:137-160): the loader has
 * already linearised the file's colours and pushed them through the buffer's storage format. */
typedef struct yor_texture_desc
{
	int32_t width, height;
	const float *texels;       /* height * width * 4 (rgba), row y = the image handler's row y */
	int32_t interpolate;       /* 0 none, 1 bilinear */
	int32_t clip;              /* TexClipMode: 0 extend, 1 clip, 2 clipcube, 3 repeat, 4 checker */
	int32_t xrepeat, yrepeat, rot90, mirror_x, mirror_y, checker_even, checker_odd;
	float checker_dist;
	float cropmin_x, cropmin_y, cropmax_x, cropmax_y;
	float adj_intensity, adj_contrast, adj_saturation, adj_hue, adj_red, adj_green, adj_blue;
	int32_t adj_clamp;
	int32_t color_space;       /* for getRawColor (texture_image.cc:90-104): 0 sRGB, 1 XYZ, 2 LinearRGB, 3 RawManualGamma */
	float gamma;
	int32_t normalmap;         /* texture_image.cc:705: the texture is a normal map (changes TextureMapperNode::evalDerivative and setup()) */
} yor_texture_desc;

enum { YOR_NODE_TEXTURE_MAPPER = 0, YOR_NODE_VALUE = 1, YOR_NODE_MIX = 2, YOR_NODE_LAYER = 3 };
/* one shader node, parameter level (the factory() arguments of shader_node_basic.cc / shader_node_layer.cc); node
 * references are indices into the material's node list, -1 = not connected */
typedef struct yor_node_desc
{
	int32_t type;
	/* texture_mapper (shader_node_basic.cc:344-413) */
	int32_t texture;           /* index into the scene's textures */
	int32_t texco;             /* Coords: 0 uv, 1 glob, 2 orco, 3 tran, 4 nor, 5 refl, 6 win, 7 stick, 8 stress, 9 tan */
	int32_t mapping;           /* Projection: 0 plain, 1 cube, 2 tube, 3 sphere */
	int32_t proj[3];
	float scale[3], offset[3]; /* as given (the factory doubles the offset) */
	float mtx[16];
	int32_t do_scalar;
	/* value (:429-438) */
	float color[4];            /* rgb + alpha */
	float scalar;
	/* mix (:480-530, :682-703) */
	int32_t mode;              /* MixModes, also the layer's blend mode */
	float cfactor;
	int32_t input1, input2, factor;
	float col1[4], col2[4];
	/* layer (shader_node_layer.cc:205-244, :148-187) */
	int32_t input, upper_layer;
	int32_t no_rgb, stencil, negative, use_alpha, do_color, do_scalar_l, color_input;
	float colfac, valfac, def_val;
	float def_col[3];
	float upper_col[4];
	float upper_val;
	float bump_strength;       /* texture_mapper "bump_strength" (factory default 1) */
} yor_node_desc;

typedef struct yor_light_desc
{
	int32_t type;
	int32_t samples;
	int32_t cast_shadows;
	int32_t pad0;
	float corner[3];   /* area: corner; point: position */
	float point1[3];
	float point2[3];
	float color[3];
	float power;
	float pad1[3];
} yor_light_desc;

typedef struct yor_camera_desc
{
	float from[3], to[3], up[3];
	int32_t resx, resy;
	float focal;
	float aspect_ratio;
	float near_clip, far_clip;
	float aperture;            /* 0: pinhole */
	float pad0;
	/* depth of field (PerspectiveCamera::factory, camera_perspective.cc:200-240) */
	float dof_distance;
	int32_t bokeh_type;        /* 0 disk1, 1 disk2, 3 triangle, 4 square, 5 pentagon, 6 hexagon, 7 ring (BokehType) */
	int32_t bokeh_bias;        /* 0 none ("uniform"), 1 center, 2 edge */
	float bokeh_rotation;      /* degrees */
} yor_camera_desc;

typedef struct yor_render_desc
{
	int32_t integrator;        /* YOR_INTEGRATOR_* */
	int32_t path_samples;
	int32_t bounces;
	int32_t rr_min_bounces;    /* russian_roulette_min_bounces */
	int32_t no_recursive;
	int32_t bg_transp;
	int32_t bg_transp_refract;
	int32_t width, height, xstart, ystart;
	int32_t aa_passes;         /* > 1: the adaptive multi-pass schedule of TiledIntegrator::render (fields at the end) */
	int32_t aa_minsamples;
	float aa_pixelwidth;
	int32_t filter_type;
	int32_t tile_size;
	uint32_t base_sampling_offset; /* adv_base_sampling_offset + 100000*adv_computer_node */
	int32_t shadow_bias_auto;
	float shadow_bias;
	int32_t min_raydist_auto;
	float min_raydist;
	float aa_light_sample_multiplier; /* 1 for a single pass */
	float background[3];       /* constant background colour*power; used on primary misses */
	int32_t has_background;
	uint32_t tile_seed_rand;   /* stands for libc rand() of integrator_tiled.cc:319 (every tile the same value); only used when RR is on */
	/* ... or the libc stream itself (glibc's rand(), restated: see yor_glibc_rand): srand(rand_srand) as the last Material /
	 * ObjectGeometric constructor left it (material.cc:56, object_geom.cc:42), rand_skip values consumed since (the
	 * constructor's colour loop), then one value per tile started, over all passes.  rand_srand < 0: tile_seed_rand. */
	int32_t rand_srand, rand_skip;
	int32_t n_threads;         /* oracle worker threads (1 = reference's single-thread linear order) */
	/* tile subset for sharded renders: tile t is rendered iff (t % shard_count) == shard_index */
	int32_t shard_index, shard_count;
	/* multi-pass anti-aliasing (Scene::setAntialiasing, scene.cc:761-778; defaults environment.cc:682-695) */
	int32_t aa_inc_samples;            /* <= 0: aa_minsamples */
	float aa_threshold;                /* 0: every pixel is resampled in every pass (imagefilm.cc:319,460) */
	float aa_resampled_floor;          /* percent of the image */
	float aa_sample_multiplier_factor, aa_light_sample_multiplier_factor, aa_indirect_sample_multiplier_factor;
	int32_t aa_detect_color_noise;
	int32_t aa_dark_detection_type;    /* 0 none, 1 linear, 2 curve */
	float aa_dark_threshold_factor;
	int32_t aa_variance_edge_size, aa_variance_pixels;
	float aa_clamp_samples;            /* ImageFilm::addSample clampProportionalRgb (imagefilm.cc:975) */
	int32_t transp_shad;               /* tr_shad_: shadow rays are filtered by transparent materials (TriKdTree::intersectTs) */
	int32_t shadow_depth;              /* s_depth_: more distinct transparent surfaces than this along a shadow ray = shadowed */
	int32_t raydepth;                  /* r_depth_ of recursiveRaytrace (integrator_montecarlo.cc:791); 0 behaves like "no recursion" */
	int32_t trace_caustics;            /* PathIntegrator::trace_caustics_: caustic_type "path" — the factory's default when the parameter is absent — or "both"
	                                    * (integrator_path_tracer.cc:36, :85, :382-387); "none" clears it */
} yor_render_desc;

typedef struct yor_stats
{
	uint64_t rays_closest;
	uint64_t rays_shadow;
	uint64_t interior_steps;
	uint64_t leaves;
	uint64_t tri_tests;
	uint64_t camera_samples;
	uint32_t kd_nodes, kd_leaf_refs;
	double build_seconds;
	double render_seconds;
} yor_stats;

typedef struct yor_scene yor_scene;

/* geometry: n_tris triangles, verts = n_tris*9 floats (a,b,c), tri_mat = material index per
 * triangle, vnormals = NULL or n_tris*9 floats of per-vertex shading normals (all-zero triple
 * = use the geometric normal for that corner) */
yor_scene *yor_scene_create(int32_t n_tris, const float *verts, const int32_t *tri_mat, const float *vnormals,
                            int32_t n_mats, const yor_material_desc *mats,
                            int32_t n_lights, const yor_light_desc *lights,
                            const yor_camera_desc *cam);
void yor_scene_destroy(yor_scene *s);
/* textures of the scene (referred to by texture_mapper nodes) and per-triangle texture coordinates: uv = n_tris * 6 floats
 * (u, v of the three corners) or NULL, orco = n_tris * 9 floats or NULL.  The material descriptions handed to
 * yor_scene_create may refer to textures set later, but before the first render. */
void yor_scene_set_textures(yor_scene *s, int32_t n_textures, const yor_texture_desc *textures);
void yor_scene_set_texcoords(yor_scene *s, const float *uv, const float *orco);
/* probes for the pins: one image-texture lookup (getColor rgba + getFloat), one material's node stack at a surface point */
void yor_texture_probe(const yor_texture_desc *t, const float p[3], float out5[5]);
void yor_nodes_probe_derivative(int32_t n_nodes, const yor_node_desc *nodes, int32_t n_textures, const yor_texture_desc *textures, const yor_camera_desc *cam,
                                const float sp31[31], float bump_scale, float *out, float *bump9);
void yor_nodes_probe(int32_t n_nodes, const yor_node_desc *nodes, int32_t n_textures, const yor_texture_desc *textures, const yor_camera_desc *cam,
                     const float sp18[18], float *out /* n_nodes * 5 */);

/* film = height*width*5 floats {r,g,b,a,weight} (the reference's Pixel, util_image_buffers.h:36-48),
 * zeroed by the callee.  Returns 0 on success, negative on unsupported configuration. */
int yor_render(yor_scene *s, const yor_render_desc *rd, float *film, yor_stats *stats);

/* test hook: record the samples renderTile hands to addSample (8 floats each: x, y, dx, dy, r, g, b, a) and every closest-hit query
 * (12 floats: from, dir, tmin, tmax, t or -1, triangle index as int bits, pixel x, y) in call order — with_shadow: plain any-hit queries
 * too (verdict in slot 8, triangle slot -2); single-threaded renders only; NULL = off */
void yor_set_trace(float *samples8, uint64_t cap_samples, float *rays12, uint64_t cap_rays, int with_shadow);
void yor_trace_counts(uint64_t *n_samples, uint64_t *n_rays);

/* Replace the oracle's own kd-tree by an externally built one in the same 8-byte node layout
 * (interior: {split, axis | right<<2}, leaf: {first_ref, 3 | count<<2}, near child = next node): lets
 * tests walk the PRODUCT's tree with the reference's traversal algorithm. */
void yor_scene_set_tree(yor_scene *s, uint32_t n_nodes, const uint32_t *nodes, uint32_t n_refs, const uint32_t *refs, const float bound6[6]);

/* ray-level entry points (kd traversal vs brute force cross-checks and GPU ray parity tests) */
int yor_intersect(const yor_scene *s, int use_tree, const float from[3], const float dir[3], float tmin, float tmax,
                  int32_t *tri, float *t, float bary[3]);
int yor_is_shadowed(const yor_scene *s, int use_tree, const float from[3], const float dir[3], float tmin, float tmax);

/* ---- component entry points, used to pin the restatement against tests/golden ---- */
float yor_fsin(float x);
float yor_fcos(float x);
float yor_fexp2(float x);
float yor_flog2(float x);
float yor_fpow(float a, float b);
/* glibc rand() after srand(seed) (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31], output >> 1): out[k] = k-th value */
void yor_glibc_rand(uint32_t seed, int count, int32_t *out);
float yor_fsqrt(float x);
float yor_facos(float x);
float yor_ri_vdc(uint32_t bits, uint32_t r);
float yor_ri_s(uint32_t i, uint32_t r);
float yor_ri_lp(uint32_t i, uint32_t r);
uint32_t yor_fnv32a(uint32_t v);
double yor_scr_halton(int dim, uint32_t n);
void yor_halton_seq(uint32_t base, uint32_t start, int count, float *out);
void yor_mwc_seq(uint32_t seed, int count, float *out);
const int *yor_faure_perm(int dim, int *len);
void yor_create_cs(const float n[3], float u[3], float v[3]);
void yor_sample_cos_hemisphere(const float n[3], const float ru[3], const float rv[3], float s1, float s2, float out[3]);
int yor_bound_cross(const float a[3], const float g[3], const float from[3], const float dir[3], float dist, float *enter, float *leave);
void yor_camera_shoot(const yor_camera_desc *cam, float px, float py, float out9[9]);
void yor_camera_shoot_lens(const yor_camera_desc *cam, float px, float py, float lu, float lv, float out9[9]);
int yor_arealight_illum_sample(const yor_light_desc *l, const float p[3], float s1, float s2, float out8[8]);
int yor_arealight_intersect(const yor_light_desc *l, const float from[3], const float dir[3], float out5[5]);
int yor_pointlight_illuminate(const yor_light_desc *l, const float p[3], float out7[7]);
/* in14 = n, ng, wo, wl, s1, s2 ; outputs as in the harness */
/* Material::getTransparency(sp, wo) */
void yor_material_transparency(const yor_material_desc *m, const float in14[14], float out3[3]);
/* Material::getSpecular + getAlpha: flags bit0 reflect, bit1 refract; out12 = dir0, col0, dir1, col1 */
void yor_material_specular(const yor_material_desc *m, const float in14[14], int32_t raylevel, int32_t *flags, float out12[12], float *alpha);
void yor_material_probe(const yor_material_desc *m, const float in14[14], int32_t sample_flags,
                        int32_t *bsdf_flags, float eval3[3], float *pdf, int32_t *sampled_flags, float sample8[8]);
/* the two-direction Material::sample recursiveRaytrace's glossy branch calls for a lobe that reflects and transmits (integrator_montecarlo.cc:919-970;
 * RoughGlassMaterial::sample, material_rough_glass.cc:165-286): out15 = dir[0], the returned colour, w[0], dir[1], tcol, w[1], s.pdf_ */
void yor_material_sample_two(const yor_material_desc *m, const float in14[14], int32_t sample_flags, int32_t *sampled_flags, float out15[15]);
/* BeerVolumeHandler(acol, dist)::transmittance over a ray of length tmax (volumehandler_beer.cc:28-48) */
void yor_beer_transmittance(const float acol[3], double dist, float tmax, int32_t *ok, float out3[3]);
void yor_lightmat_emit(const yor_material_desc *m, const float n[3], const float wo[3], int include_lights, float out3[3]);

#ifdef __cplusplus
}
#endif
#endif
