// TEST INFRASTRUCTURE — not product code.
//
// Integrator-level reference harness: pins the CONTROL FLOW of the path — the rows of SURVEY §8a that the
// component harness (ref_components.cc) cannot reach:
//
//   T1  TiledIntegrator::render / renderPass          src/integrator/integrator_tiled.cc:116-307
//   T2  TiledIntegrator::renderTile                   src/integrator/integrator_tiled.cc:309-521
//   I1  PathIntegrator::integrate (+ DirectLightIntegrator::integrate)
//                                                     src/integrator/integrator_path_tracer.cc:112-347, integrator_direct_light.cc:104-185
//   D1  MonteCarloIntegrator::doLightEstimation       src/integrator/integrator_montecarlo.cc:78-345
//   D2  estimateAllDirectLight / estimateOneDirectLight   :47-76
//   R1  recursiveRaytrace                             :782-1028
//   P1  the Russian-roulette stream (per-tile Random seeded from libc rand(), integrator_tiled.cc:319)
//
// This translation unit is OUR driver.  oracle/Makefile (target `ref`) compiles it against the reference's own
// headers and links it with the reference's own integrator_{tiled,montecarlo,path_tracer,direct_light,empty_volume}.cc,
// renderpasses.cc, imagesplitter.cc, timer.cc, background_constant.cc and the component objects (materials,
// lights, camera, QMC ...) compiled WHERE THEY LIE under /root/reference.  Nothing from the reference is copied.
//
// ---- what is NOT the reference here (read this before trusting a number) ------------------------------------
// Scene (src/common/scene.cc), Triangle (include/common/triangle.h), TriKdTree and ImageFilm (src/common/imagefilm.cc)
// include the cmake-generated yafaray_config.h and cannot be compiled under this project's rules.
// The integrators above call into them, so THIS FILE PROVIDES THE BODIES of the following reference-declared
// member functions — they are harness code, written for the harness, and are not evidence about the reference:
//
//   Scene::Scene, ~Scene, setCamera, setBackground, getBackground, setAntialiasing, getAaParameters, getSignals,
//   getRenderPasses, passEnabled                       trivial state holders
//   Scene::intersect (Ray / DiffRay), Scene::isShadowed (plain / transparent)
//                                                      the geometry query: BRUTE FORCE over the harness's triangle list
//                                                      with the arithmetic of Triangle::intersect / getSurface (flat
//                                                      triangles, no UV) and the wrapper semantics of scene.cc:896-1035
//   ImageFilm::ImageFilm, ~ImageFilm, init, setAaNoiseParams, nextArea, finishArea, nextPass, doMoreSamples,
//   getImagePassFromIntPassType                        tile hand-out through the reference's real ImageSplitter, every
//                                                      pixel resampled in every pass (the reference's behaviour at
//                                                      AA_threshold = 0, imagefilm.cc:319,460,917-920)
//   ImageFilm::addSample                               RECORDS the sample (x, y, dx, dy, rgba) — the harness's output
//   RenderEnvironment::createVolumeH                   GlassMaterial::factory creates the Beer handler of an absorbing glass through it
//                                                      (material_glass.cc:390-398; environment.cc:586-607 keeps a name table and dispatches
//                                                      on the type): here it hands the ParamMap straight to the reference's
//                                                      BeerVolumeHandler::factory
//
// So what the fixture pins is: given the same geometry answers, the reference's render() / renderTile() / integrate() /
// doLightEstimation() / estimateOneDirectLight() / recursiveRaytrace() — run on the reference's real materials, lights,
// camera, QMC and PRNG — produce these per-sample colours, in this order, with these ray queries.  Rows K1/K2/G1/G2/S1
// (traversal, triangle test, getSurface, scene wrappers) and F1 (the film filter) stay pinned only as DESIGN.md §2 says.
//
// Output: one JSON document on stdout: the scene (reference parameter names, so that tests can hand the same
// description to the oracle and to the device through the C API) and per case the recorded samples and the first
// ray queries as IEEE-754 bit patterns.  tests/golden/make_golden.py stores it as ref_integrator_{fast,ieee}.json.gz.

#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <list>
#include <map>
#include <limits>

#include "constants.h"
#include "common/vector.h"
#include "common/ray.h"
#include "common/color.h"
#include "common/param.h"
#include "common/surface.h"
#include "common/scene.h"
#include "common/imagefilm.h"
#include "common/imagesplitter.h"
#include "common/renderpasses.h"
#include "common/logging.h"
#include "common/session.h"
#include "camera/camera_perspective.h"
#include "background/background_constant.h"
#include "integrator/integrator_path_tracer.h"
#include "integrator/integrator_direct_light.h"
#include "integrator/integrator_empty_volume.h"
#include "material/material_glass.h"
#include "material/material_coated_glossy.h"
#include "material/material_shiny_diffuse.h"
#include "material/material_glossy.h"
#include "material/material_simple.h"
#include "material/material_rough_glass.h"
#include "light/light_area.h"
#include "light/light_point.h"
#include "volume/volumehandler_beer.h"
#include "common/environment.h"

using namespace yafaray4;

// The two compile-time constants of the generated config header (CMakeLists.txt:46-52 ->
// CMakeConfig/templates/yafaray_config.h.cmake:29-30), under the harness's own names
static const double H_MIN_RAYDIST = 0.00005;
static const double H_SHADOW_BIAS = 0.0005;

static uint32_t f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }

// ---------------------------------------------------------------- harness geometry
struct HTri
{
	Point3 a, b, c;
	Vec3 e1, e2, ng;
	float eps;
	const Material *mat;
};
static std::vector<HTri> g_tris;

struct RayRec { float from[3], dir[3], tmin, tmax_in; int tri; float t; };
static std::vector<RayRec> g_ray_log;
static size_t g_ray_log_cap = 0;
static uint64_t g_n_closest = 0, g_n_shadow = 0;

// arithmetic of Triangle::intersect (include/common/triangle.h:223-259), on the harness's own triangle record
static inline bool tri_hit(const HTri &tr, const Ray &ray, float &t, float &u, float &v)
{
	Vec3 pvec = ray.dir_ ^ tr.e2;
	float det = tr.e1 * pvec;
	float epsilon = tr.eps;
	if(det > -epsilon && det < epsilon) return false;
	float inv_det = 1.f / det;
	Vec3 tvec = ray.from_ - tr.a;
	u = (tvec * pvec) * inv_det;
	if(u < 0.f || u > 1.f) return false;
	Vec3 qvec = tvec ^ tr.e1;
	v = (ray.dir_ * qvec) * inv_det;
	if((v < 0.f) || ((u + v) > 1.f)) return false;
	t = tr.e2 * qvec * inv_det;
	if(t < epsilon) return false;
	return true;
}

// what Triangle::getSurface (src/common/triangle.cc:30-133) leaves for a flat triangle of a mesh without UVs / orco
static void fill_sp(SurfacePoint &sp, int ti, const Point3 &hit, float u, float v)
{
	const HTri &tr = g_tris[ti];
	sp.ng_ = tr.ng;
	sp.n_ = sp.ng_;
	sp.orco_p_ = hit; sp.has_orco_ = false; sp.orco_ng_ = sp.ng_;
	sp.dp_du_ = tr.b - tr.a;
	sp.dp_dv_ = tr.c - tr.b;
	sp.u_ = 0.f; sp.v_ = 0.f;
	sp.dp_du_abs_ = sp.dp_du_; sp.dp_dv_abs_ = sp.dp_dv_;
	sp.dp_du_.normalize(); sp.dp_dv_.normalize();
	sp.object_ = nullptr;
	sp.prim_num_ = ti;
	sp.material_ = tr.mat;
	sp.p_ = hit;
	createCs__(sp.n_, sp.nu_, sp.nv_);
	sp.ds_du_.x_ = sp.nu_ * sp.dp_du_; sp.ds_du_.y_ = sp.nv_ * sp.dp_du_; sp.ds_du_.z_ = sp.n_ * sp.dp_du_;
	sp.ds_dv_.x_ = sp.nu_ * sp.dp_dv_; sp.ds_dv_.y_ = sp.nv_ * sp.dp_dv_; sp.ds_dv_.z_ = sp.n_ * sp.dp_dv_;
	sp.light_ = nullptr;
	sp.has_uv_ = false;
	sp.data_.b_0_ = 1 - u - v; sp.data_.b_1_ = u; sp.data_.b_2_ = v;
	sp.data_.edge_1_ = &tr.e1; sp.data_.edge_2_ = &tr.e2;
}

static bool closest_hit(const Ray &ray, SurfacePoint &sp)
{
	float dis, z;
	if(ray.tmax_ < 0) dis = std::numeric_limits<float>::infinity();
	else dis = ray.tmax_;
	++g_n_closest;
	z = dis;
	int hit_tri = -1; float hu = 0.f, hv = 0.f;
	for(size_t i = 0; i < g_tris.size(); ++i)
	{
		float t, u, v;
		if(!tri_hit(g_tris[i], ray, t, u, v)) continue;
		// kdtree_triangle.cc:782-786: closer than the best so far, not before tmin, material visible to camera rays
		if(t < z && t >= ray.tmin_)
		{
			const Visibility vis = g_tris[i].mat->getVisibility();
			if(vis == NormalVisible || vis == VisibleNoShadows) { z = t; hit_tri = (int)i; hu = u; hv = v; }
		}
	}
	if(g_ray_log.size() < g_ray_log_cap)
	{
		RayRec r;
		r.from[0] = ray.from_.x_; r.from[1] = ray.from_.y_; r.from[2] = ray.from_.z_;
		r.dir[0] = ray.dir_.x_; r.dir[1] = ray.dir_.y_; r.dir[2] = ray.dir_.z_;
		r.tmin = ray.tmin_; r.tmax_in = ray.tmax_; r.tri = hit_tri; r.t = hit_tri >= 0 ? z : -1.f;
		g_ray_log.push_back(r);
	}
	if(hit_tri < 0) return false;
	Point3 h = ray.from_ + z * ray.dir_;
	fill_sp(sp, hit_tri, h, hu, hv);
	sp.origin_ = nullptr;
	ray.tmax_ = z;
	return true;
}

// ---------------------------------------------------------------- harness-provided bodies: Scene
static RenderPasses *g_passes = nullptr;
struct AaStore
{
	int samples, passes, inc_samples; float threshold, resampled_floor, sample_mult, light_mult, indirect_mult;
	bool detect_color_noise; DarkDetectionType dark_type; float dark_factor; int var_edge, var_pixels; float clamp_samples, clamp_indirect;
};
static AaStore g_aa;

BEGIN_YAFARAY

Scene::Scene(const RenderEnvironment *render_environment): vol_integrator_(nullptr), camera_(nullptr), image_film_(nullptr),
	tree_(nullptr), vtree_(nullptr), background_(nullptr), surf_integrator_(nullptr), nthreads_(1), nthreads_photons_(1), mode_(0), signals_(0),
	env_(render_environment)
{
	shadow_bias_ = 0.f; shadow_bias_auto_ = true; ray_min_dist_ = 0.f; ray_min_dist_auto_ = true;
}
Scene::~Scene() {}
void Scene::setCamera(Camera *cam) { camera_ = cam; }
void Scene::setBackground(Background *bg) { background_ = bg; }
Background *Scene::getBackground() const { return background_; }
int Scene::getSignals() const { return 0; }
const RenderPasses *Scene::getRenderPasses() const { return g_passes; }
bool Scene::passEnabled(IntPassTypes int_pass_type) const { return int_pass_type == PassIntCombined; }
void Scene::getAaParameters(int &samples, int &passes, int &inc_samples, float &threshold, float &resampled_floor, float &sample_multiplier_factor, float &light_sample_multiplier_factor, float &indirect_sample_multiplier_factor, bool &detect_color_noise, DarkDetectionType &dark_detection_type, float &dark_threshold_factor, int &variance_edge_size, int &variance_pixels, float &clamp_samples, float &clamp_indirect) const
{
	samples = g_aa.samples; passes = g_aa.passes; inc_samples = g_aa.inc_samples; threshold = g_aa.threshold; resampled_floor = g_aa.resampled_floor;
	sample_multiplier_factor = g_aa.sample_mult; light_sample_multiplier_factor = g_aa.light_mult; indirect_sample_multiplier_factor = g_aa.indirect_mult;
	detect_color_noise = g_aa.detect_color_noise; dark_detection_type = g_aa.dark_type; dark_threshold_factor = g_aa.dark_factor;
	variance_edge_size = g_aa.var_edge; variance_pixels = g_aa.var_pixels; clamp_samples = g_aa.clamp_samples; clamp_indirect = g_aa.clamp_indirect;
}

// wrapper semantics of scene.cc:896-960
bool Scene::intersect(const Ray &ray, SurfacePoint &sp) const
{
	if(!closest_hit(ray, sp)) return false;
	sp.ray_ = nullptr;
	return true;
}
bool Scene::intersect(const DiffRay &ray, SurfacePoint &sp) const
{
	if(!closest_hit(ray, sp)) return false;
	sp.ray_ = &ray;
	return true;
}

// scene.cc:962-994 + the acceptance test of TriKdTree::intersectS (kdtree_triangle.cc:936-945)
bool Scene::isShadowed(RenderState &state, const Ray &ray, float &obj_index, float &mat_index) const
{
	Ray sray(ray);
	sray.from_ += sray.dir_ * sray.tmin_;
	sray.time_ = state.time_;
	float dis;
	if(ray.tmax_ < 0) dis = std::numeric_limits<float>::infinity();
	else dis = sray.tmax_ - 2 * sray.tmin_;
	++g_n_shadow;
	for(size_t i = 0; i < g_tris.size(); ++i)
	{
		float t, u, v;
		if(!tri_hit(g_tris[i], sray, t, u, v)) continue;
		if(t < dis && t >= 0.f)
		{
			const Visibility vis = g_tris[i].mat->getVisibility();
			if(vis == NormalVisible || vis == InvisibleShadowsOnly) return true;
		}
	}
	return false;
}

// scene.cc:996-1035 + the acceptance / filtering of TriKdTree::intersectTs (kdtree_triangle.cc:1099-1125): hits from the
// shadow ray's tmin_ on; an opaque material blocks; a transparent triangle filters once; more than max_depth of them block.
// Triangles are visited in index order here (the reference: in kd order) — the filter PRODUCT's last bits depend on that
// order when a ray crosses several transparent triangles; the harness scenes keep at most one transparent sheet in a ray's way.
bool Scene::isShadowed(RenderState &state, const Ray &ray, int max_depth, Rgb &filt, float &obj_index, float &mat_index) const
{
	Ray sray(ray);
	sray.from_ += sray.dir_ * sray.tmin_;
	float dis;
	if(ray.tmax_ < 0) dis = std::numeric_limits<float>::infinity();
	else dis = sray.tmax_ - 2 * sray.tmin_;
	filt = Rgb(1.0);
	void *odat = state.userdata_;
	alignas(16) unsigned char userdata[USER_DATA_SIZE + 7];
	state.userdata_ = (void *)userdata;
	++g_n_shadow;
	bool isect = false;
	int depth = 0;
	for(size_t i = 0; i < g_tris.size() && !isect; ++i)
	{
		float t, u, v;
		if(!tri_hit(g_tris[i], sray, t, u, v)) continue;
		if(t < dis && t >= sray.tmin_)
		{
			const Material *mat = g_tris[i].mat;
			const Visibility vis = mat->getVisibility();
			if(!(vis == NormalVisible || vis == InvisibleShadowsOnly)) continue;
			if(!mat->isTransparent()) { isect = true; break; }
			if(depth >= max_depth) { isect = true; break; }
			Point3 h = sray.from_ + t * sray.dir_;
			SurfacePoint sp;
			fill_sp(sp, (int)i, h, u, v);
			filt *= mat->getTransparency(state, sp, sray.dir_);
			++depth;
		}
	}
	state.userdata_ = odat;
	return isect;
}

// ---------------------------------------------------------------- harness-provided body: RenderEnvironment::createVolumeH
VolumeHandler *RenderEnvironment::createVolumeH(const std::string &, const ParamMap &params) { return BeerVolumeHandler::factory(params, *this); }

// ---------------------------------------------------------------- harness-provided bodies: ImageFilm
struct SampleRec { int x, y; float dx, dy; float c[4]; };
static std::vector<SampleRec> g_samples;
static std::vector<int> g_tile_log;      // x, y, w, h of every tile handed out, in order

ImageFilm::ImageFilm(int width, int height, int xstart, int ystart, ColorOutput &out, float filter_size, FilterType filt,
					 RenderEnvironment *e, bool show_sam_mask, int t_size, ImageSplitter::TilesOrderType tiles_order_type, bool pm_a):
	density_image_(nullptr), dp_image_(nullptr), w_(width), h_(height), cx_0_(xstart), cx_1_(xstart + width), cy_0_(ystart), cy_1_(ystart + height),
	area_cnt_(0), completed_cnt_(0), next_area_(0), aa_thesh_(0.f), filterw_(filter_size * 0.5f), table_scale_(0.f), output_(&out), env_(e), n_pass_(1),
	show_mask_(show_sam_mask), tile_size_(t_size), tiles_order_(tiles_order_type), premult_alpha_(pm_a), n_passes_(1)
{
}
ImageFilm::~ImageFilm() { if(splitter_) delete splitter_; }
void ImageFilm::init(int num_passes)
{
	next_area_ = 0;
	if(splitter_) delete splitter_;
	splitter_ = new ImageSplitter(w_, h_, cx_0_, cy_0_, tile_size_, tiles_order_, 1);      // the reference's own splitter (imagefilm.cc:206-214)
	area_cnt_ = splitter_->size();
	n_pass_ = 1; n_passes_ = num_passes;
}
void ImageFilm::setAaNoiseParams(bool, const DarkDetectionType &, float, int, int, float) {}
bool ImageFilm::nextArea(int num_view, RenderArea &a)
{
	int n = next_area_++;
	if(!splitter_->getArea(n, a)) return false;
	int ifilterw = (int) ceil(filterw_);
	a.sx_0_ = a.x_ + ifilterw; a.sx_1_ = a.x_ + a.w_ - ifilterw; a.sy_0_ = a.y_ + ifilterw; a.sy_1_ = a.y_ + a.h_ - ifilterw;
	g_tile_log.push_back(a.x_); g_tile_log.push_back(a.y_); g_tile_log.push_back(a.w_); g_tile_log.push_back(a.h_);
	return true;
}
void ImageFilm::finishArea(int, RenderArea &) {}
int ImageFilm::nextPass(int, bool, std::string, bool skip_next_pass)
{
	next_area_ = 0;
	n_pass_++;
	if(skip_next_pass) return 0;
	return h_ * w_;      // aa_thesh_ == 0: every pixel (imagefilm.cc:458-461)
}
bool ImageFilm::doMoreSamples(int, int) const { return true; }      // aa_thesh_ == 0 (imagefilm.cc:917-920)
Rgba2DImageWeighed_t *ImageFilm::getImagePassFromIntPassType(int) { return nullptr; }
void ImageFilm::addSample(ColorPasses &color_passes, int x, int y, float dx, float dy, const RenderArea *, int, int, float)
{
	SampleRec r;
	r.x = x; r.y = y; r.dx = dx; r.dy = dy;
	const Rgba &c = color_passes(PassIntCombined);
	r.c[0] = c.r_; r.c[1] = c.g_; r.c[2] = c.b_; r.c[3] = c.a_;
	g_samples.push_back(r);
}

END_YAFARAY

// ---------------------------------------------------------------- parameter lists (emitted as JSON and turned into ParamMaps)
struct P
{
	enum Kind { I, B, F, S, V3 } kind;
	std::string name; int i; bool b; double f; std::string s; double v[3];
};
typedef std::vector<P> Params;
static P pi(const char *n, int v) { P p; p.kind = P::I; p.name = n; p.i = v; return p; }
static P pb(const char *n, bool v) { P p; p.kind = P::B; p.name = n; p.b = v; return p; }
static P pf(const char *n, double v) { P p; p.kind = P::F; p.name = n; p.f = v; return p; }
static P ps(const char *n, const char *v) { P p; p.kind = P::S; p.name = n; p.s = v; return p; }
static P pv(const char *n, double x, double y, double z) { P p; p.kind = P::V3; p.name = n; p.v[0] = x; p.v[1] = y; p.v[2] = z; return p; }

static bool is_point_key(const std::string &k) { return k == "from" || k == "to" || k == "up" || k == "corner" || k == "point1" || k == "point2"; }

static ParamMap to_map(const Params &ps_)
{
	ParamMap m;
	for(const P &p : ps_)
	{
		switch(p.kind)
		{
			case P::I: m[p.name] = Parameter(p.i); break;
			case P::B: m[p.name] = Parameter(p.b); break;
			case P::F: m[p.name] = Parameter(p.f); break;
			case P::S: m[p.name] = Parameter(p.s); break;
			case P::V3:
				if(is_point_key(p.name)) m[p.name] = Parameter(Point3((float)p.v[0], (float)p.v[1], (float)p.v[2]));
				else m[p.name] = Parameter(Rgba((float)p.v[0], (float)p.v[1], (float)p.v[2], 1.f));
				break;
		}
	}
	return m;
}

static std::string json_params(const Params &ps_)
{
	std::string s = "{";
	char b[128];
	bool first = true;
	for(const P &p : ps_)
	{
		if(!first) s += ", ";
		first = false;
		s += "\"" + p.name + "\": ";
		switch(p.kind)
		{
			case P::I: snprintf(b, sizeof b, "%d", p.i); s += b; break;
			case P::B: s += p.b ? "true" : "false"; break;
			case P::F:      // always with a decimal point: the ParamMap is strictly typed (param.cc:49-53) and a JSON reader makes "14" an int
				snprintf(b, sizeof b, "%.17g", p.f); s += b;
				if(!strpbrk(b, ".eEn")) s += ".0";
				break;
			case P::S: s += "\"" + p.s + "\""; break;
			case P::V3: snprintf(b, sizeof b, "[%.17g, %.17g, %.17g]", p.v[0], p.v[1], p.v[2]); s += b; break;
		}
	}
	s += "}";
	return s;
}

static const P *find(const Params &ps_, const char *n) { for(const P &p : ps_) if(p.name == n) return &p; return nullptr; }

// The factories take a RenderEnvironment& they never touch with these parameters (no shader nodes, no IBL); see ref_components.cc
alignas(64) static unsigned char fake_env_storage[1 << 16];
static RenderEnvironment &fake_env() { return *reinterpret_cast<RenderEnvironment *>(fake_env_storage); }

static Material *make_material(const Params &ps_)
{
	std::list<ParamMap> no_nodes;
	ParamMap m = to_map(ps_);
	m["name"] = std::string("material");      // RenderEnvironment::createMaterial, environment.cc:261 (GlassMaterial::factory creates its Beer handler only with it)
	const std::string t = find(ps_, "type")->s;
	if(t == "shinydiffusemat") return ShinyDiffuseMaterial::factory(m, no_nodes, fake_env());
	if(t == "glossy") return GlossyMaterial::factory(m, no_nodes, fake_env());
	if(t == "coated_glossy") return CoatedGlossyMaterial::factory(m, no_nodes, fake_env());
	if(t == "glass") return GlassMaterial::factory(m, no_nodes, fake_env());
	if(t == "mirror") return MirrorMaterial::factory(m, no_nodes, fake_env());
	if(t == "rough_glass") return RoughGlassMaterial::factory(m, no_nodes, fake_env());
	if(t == "light_mat") return LightMaterial::factory(m, no_nodes, fake_env());
	fprintf(stderr, "unknown material type %s\n", t.c_str());
	exit(2);
}
static Light *make_light(const Params &ps_)
{
	ParamMap m = to_map(ps_);
	const std::string t = find(ps_, "type")->s;
	if(t == "arealight") return AreaLight::factory(m, fake_env());
	if(t == "pointlight") return PointLight::factory(m, fake_env());
	fprintf(stderr, "unknown light type %s\n", t.c_str());
	exit(2);
}

// ---------------------------------------------------------------- the scene
// A Cornell-style room [-1,1]^3 open toward -y with four objects (slots A-D) whose materials a case chooses, two quad
// area lights on emissive geometry and a point light.  Coordinates are deliberately uneven so that no camera or
// bounce ray runs exactly into an edge two triangles share (ties are resolved by visiting order, which brute force
// and a kd-tree do not share).
struct Quad { double p[4][3]; int slot; };      // slot < 0: fixed material -slot-1; slot >= 0: object slot
static std::vector<Quad> g_quads;

static void add_quad(const double a[3], const double b[3], const double c[3], const double d[3], int slot)
{
	Quad q;
	for(int k = 0; k < 3; ++k) { q.p[0][k] = a[k]; q.p[1][k] = b[k]; q.p[2][k] = c[k]; q.p[3][k] = d[k]; }
	q.slot = slot;
	g_quads.push_back(q);
}
static void quad(double ax, double ay, double az, double bx, double by, double bz, double cx, double cy, double cz, double dx, double dy, double dz, int slot)
{
	const double a[3] = {ax, ay, az}, b[3] = {bx, by, bz}, c[3] = {cx, cy, cz}, d[3] = {dx, dy, dz};
	add_quad(a, b, c, d, slot);
}
// a box with centre c, half sizes h, rotated by `deg` about z; outward-facing quads
static void box(double cx, double cy, double cz, double hx, double hy, double hz, double deg, int slot)
{
	const double r = deg * 3.14159265358979323846 / 180.0, cs = std::cos(r), sn = std::sin(r);
	double v[8][3];
	for(int i = 0; i < 8; ++i)
	{
		const double x = (i & 1) ? hx : -hx, y = (i & 2) ? hy : -hy, z = (i & 4) ? hz : -hz;
		v[i][0] = (double)(float)(cx + cs * x - sn * y); v[i][1] = (double)(float)(cy + sn * x + cs * y); v[i][2] = (double)(float)(cz + z);
	}
	static const int f[6][4] = {{0, 2, 3, 1}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 4, 6, 2}, {1, 3, 7, 5}};
	for(int k = 0; k < 6; ++k) add_quad(v[f[k][0]], v[f[k][1]], v[f[k][2]], v[f[k][3]], slot);
}

enum { M_WHITE = 0, M_RED, M_GREEN_ON, M_LIGHT1, M_GLOSSY, M_SD_MIRROR_TRANSP, M_GLASS, M_COATED, M_GLOSSY_REC, M_MIRROR, M_SD_EMIT, M_LIGHT2, M_SD_TRANSP,
       M_GLASS_ABS, M_GLASS_FAKE, M_ANISO, M_COATED_REC, M_SD_TRANSL, M_SD_FLAT, M_SD_DEPTH, M_SD_NOSHADOW, M_SD_SHADOWONLY, M_SD_NORECV, M_SD_NOLOBE, M_SD_MIRRORONLY, M_ROUGH_GLASS, M_ROUGH_GLASS_ABS_FAKE, N_MATS };
enum { SLOT_A = 0, SLOT_B, SLOT_C, SLOT_D, N_SLOTS };

static std::vector<Params> g_mat_params;
static std::vector<Material *> g_mats;
static std::vector<Params> g_light_params;

static void build_catalogue()
{
	g_mat_params.resize(N_MATS);
	g_mat_params[M_WHITE] = {ps("type", "shinydiffusemat"), pv("color", 0.75, 0.75, 0.75), pf("diffuse_reflect", 1.0)};
	g_mat_params[M_RED] = {ps("type", "shinydiffusemat"), pv("color", 0.7, 0.15, 0.15), pf("diffuse_reflect", 0.9)};
	g_mat_params[M_GREEN_ON] = {ps("type", "shinydiffusemat"), pv("color", 0.15, 0.7, 0.15), pf("diffuse_reflect", 1.0), ps("diffuse_brdf", "oren_nayar"), pf("sigma", 0.3)};
	g_mat_params[M_LIGHT1] = {ps("type", "light_mat"), pv("color", 1.0, 0.95, 0.9), pf("power", 14.0)};
	g_mat_params[M_GLOSSY] = {ps("type", "glossy"), pv("color", 0.9, 0.85, 0.8), pv("diffuse_color", 0.5, 0.55, 0.7), pf("diffuse_reflect", 0.5),
	                          pf("glossy_reflect", 0.6), pf("exponent", 40.0), pb("as_diffuse", true)};
	g_mat_params[M_SD_MIRROR_TRANSP] = {ps("type", "shinydiffusemat"), pv("color", 0.8, 0.8, 0.9), pv("mirror_color", 0.9, 0.9, 1.0), pf("diffuse_reflect", 0.5),
	                                    pf("specular_reflect", 0.4), pf("transparency", 0.35), pf("IOR", 1.4), pb("fresnel_effect", true), pf("transmit_filter", 0.8)};
	g_mat_params[M_GLASS] = {ps("type", "glass"), pf("IOR", 1.5), pv("filter_color", 0.8, 0.95, 0.85), pf("transmit_filter", 0.7), pv("mirror_color", 1.0, 1.0, 1.0)};
	g_mat_params[M_COATED] = {ps("type", "coated_glossy"), pv("color", 0.9, 0.8, 0.7), pv("diffuse_color", 0.3, 0.5, 0.7), pv("mirror_color", 1.0, 0.95, 0.9),
	                          pf("diffuse_reflect", 0.5), pf("glossy_reflect", 0.6), pf("exponent", 80.0), pf("specular_reflect", 0.8), pf("IOR", 1.6), pb("as_diffuse", true)};
	g_mat_params[M_GLOSSY_REC] = {ps("type", "glossy"), pv("color", 0.85, 0.9, 0.8), pv("diffuse_color", 0.6, 0.4, 0.3), pf("diffuse_reflect", 0.3),
	                              pf("glossy_reflect", 0.8), pf("exponent", 200.0), pb("as_diffuse", false)};
	g_mat_params[M_MIRROR] = {ps("type", "mirror"), pv("color", 0.9, 0.9, 0.85), pf("reflect", 0.85)};
	g_mat_params[M_SD_EMIT] = {ps("type", "shinydiffusemat"), pv("color", 0.9, 0.6, 0.2), pf("diffuse_reflect", 0.8), pf("emit", 0.6)};
	g_mat_params[M_LIGHT2] = {ps("type", "light_mat"), pv("color", 0.6, 0.8, 1.0), pf("power", 9.0)};
	g_mat_params[M_SD_TRANSP] = {ps("type", "shinydiffusemat"), pv("color", 0.9, 0.5, 0.4), pf("diffuse_reflect", 0.6), pf("transparency", 0.7), pf("transmit_filter", 0.9)};
	g_mat_params[M_GLASS_ABS] = {ps("type", "glass"), pf("IOR", 1.45), pv("filter_color", 0.9, 0.9, 1.0), pf("transmit_filter", 0.5), pv("mirror_color", 0.95, 1.0, 0.95),
	                             pv("absorption", 0.4, 0.7, 0.9), pf("absorption_dist", 0.6)};
	g_mat_params[M_GLASS_FAKE] = {ps("type", "glass"), pf("IOR", 1.33), pv("filter_color", 0.6, 0.9, 0.7), pf("transmit_filter", 0.8), pb("fake_shadows", true)};
	g_mat_params[M_ANISO] = {ps("type", "glossy"), pv("color", 0.8, 0.85, 0.9), pv("diffuse_color", 0.7, 0.5, 0.3), pf("diffuse_reflect", 0.4), pf("glossy_reflect", 0.7),
	                         pb("anisotropic", true), pf("exp_u", 20.0), pf("exp_v", 300.0), pb("as_diffuse", true)};
	g_mat_params[M_COATED_REC] = {ps("type", "coated_glossy"), pv("color", 0.9, 0.9, 0.8), pv("diffuse_color", 0.4, 0.3, 0.6), pv("mirror_color", 1.0, 1.0, 1.0),
	                              pf("diffuse_reflect", 0.6), pf("glossy_reflect", 0.5), pf("exponent", 120.0), pf("specular_reflect", 0.7), pf("IOR", 1.5), pb("as_diffuse", false),
	                              ps("diffuse_brdf", "Oren-Nayar"), pf("sigma", 0.2)};
	g_mat_params[M_SD_TRANSL] = {ps("type", "shinydiffusemat"), pv("color", 0.6, 0.8, 0.5), pf("diffuse_reflect", 0.7), pf("translucency", 0.4), pf("transmit_filter", 0.6), pf("emit", 0.2)};
	g_mat_params[M_SD_FLAT] = {ps("type", "shinydiffusemat"), pv("color", 0.8, 0.7, 0.3), pf("diffuse_reflect", 0.9), pb("flat_material", true)};
	g_mat_params[M_SD_DEPTH] = {ps("type", "shinydiffusemat"), pv("color", 0.7, 0.7, 0.8), pv("mirror_color", 0.9, 0.95, 1.0), pf("diffuse_reflect", 0.6), pf("specular_reflect", 0.5),
	                            pf("transparency", 0.3), pf("transmit_filter", 0.7), pi("additionaldepth", 2), pf("transparentbias_factor", 0.01), pb("transparentbias_multiply_raydepth", true)};
	g_mat_params[M_SD_NOSHADOW] = {ps("type", "shinydiffusemat"), pv("color", 0.3, 0.6, 0.8), pf("diffuse_reflect", 0.9), ps("visibility", "no_shadows")};
	g_mat_params[M_SD_SHADOWONLY] = {ps("type", "shinydiffusemat"), pv("color", 0.8, 0.3, 0.6), pf("diffuse_reflect", 0.9), ps("visibility", "shadow_only")};
	g_mat_params[M_SD_NORECV] = {ps("type", "shinydiffusemat"), pv("color", 0.7, 0.7, 0.4), pf("diffuse_reflect", 0.9), pb("receive_shadows", false)};
	g_mat_params[M_SD_NOLOBE] = {ps("type", "shinydiffusemat"), pv("color", 0.9, 0.4, 0.1), pf("diffuse_reflect", 0.0), pf("emit", 0.4)};      // nothing to sample: sample() returns Rgb(1) and leaves wi, w alone
	g_mat_params[M_SD_MIRRORONLY] = {ps("type", "shinydiffusemat"), pv("color", 0.5, 0.5, 0.5), pv("mirror_color", 0.8, 0.9, 0.7), pf("diffuse_reflect", 0.0), pf("specular_reflect", 1.0)};
	g_mat_params[M_ROUGH_GLASS] = {ps("type", "rough_glass"), pf("IOR", 1.5), pv("filter_color", 0.75, 0.9, 0.8), pf("transmit_filter", 0.6), pv("mirror_color", 0.95, 0.9, 1.0), pf("alpha", 0.35)};
	g_mat_params[M_ROUGH_GLASS_ABS_FAKE] = {ps("type", "rough_glass"), pf("IOR", 1.33), pv("filter_color", 0.9, 0.7, 0.6), pf("transmit_filter", 0.5), pf("alpha", 0.15), pb("fake_shadows", true),
	                                        pv("absorption", 0.6, 0.8, 0.5), pf("absorption_dist", 0.4)};
	for(const Params &p : g_mat_params) g_mats.push_back(make_material(p));

	// room
	quad(-1, -1, -1, 1, -1, -1, 1, 1, -1, -1, 1, -1, -1 - M_WHITE);            // floor
	quad(-1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, 1, -1 - M_WHITE);                // ceiling
	quad(-1, 1, -1, 1, 1, -1, 1, 1, 1, -1, 1, 1, -1 - M_WHITE);                // back
	quad(-1, -1, -1, -1, 1, -1, -1, 1, 1, -1, -1, 1, -1 - M_RED);              // left
	quad(1, -1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1 - M_GREEN_ON);             // right
	// light 1: under the ceiling, facing down; arealight: to_x = point1 - corner, to_y = point2 - corner, fnormal = to_y x to_x must point at the room
	quad(-0.31, -0.22, 0.985, -0.31, 0.27, 0.985, 0.24, 0.27, 0.985, 0.24, -0.22, 0.985, -1 - M_LIGHT1);
	// light 2: on the left wall, facing +x
	quad(-0.985, -0.45, 0.1, -0.985, -0.45, 0.62, -0.985, 0.15, 0.62, -0.985, 0.15, 0.1, -1 - M_LIGHT2);
	// slot A: a tall box left of the centre, slot B: a low box on the right, slot C: a tilted free-standing sheet, slot D: a sheet hanging under light 1
	box(-0.37, 0.31, -0.42, 0.29, 0.27, 0.58, 17.0, SLOT_A);
	box(0.43, -0.21, -0.71, 0.31, 0.28, 0.29, -23.0, SLOT_B);
	quad(-0.15, -0.55, -0.93, 0.55, -0.35, -0.95, 0.47, -0.05, -0.18, -0.22, -0.27, -0.14, SLOT_C);
	quad(-0.52, -0.47, 0.41, 0.46, -0.43, 0.47, 0.49, 0.38, 0.52, -0.49, 0.41, 0.44, SLOT_D);

	g_light_params.push_back({ps("type", "arealight"), pv("corner", -0.31, -0.22, 0.985), pv("point1", -0.31, 0.27, 0.985), pv("point2", 0.24, -0.22, 0.985),
	                          pv("color", 1.0, 0.95, 0.9), pf("power", 14.0), pi("samples", 2)});
	g_light_params.push_back({ps("type", "arealight"), pv("corner", -0.985, -0.45, 0.1), pv("point1", -0.985, 0.15, 0.1), pv("point2", -0.985, -0.45, 0.62),
	                          pv("color", 0.6, 0.8, 1.0), pf("power", 9.0), pi("samples", 3)});
	g_light_params.push_back({ps("type", "pointlight"), pv("from", 0.55, -0.6, 0.35), pv("color", 1.0, 0.9, 0.7), pf("power", 1.6)});
}

// ---------------------------------------------------------------- cases
struct Case
{
	const char *name;
	int slot_mat[N_SLOTS];
	std::vector<int> lights;           // indices into g_light_params
	std::vector<P> light_override;     // e.g. samples of light 0
	Params camera;                     // extra camera parameters (depth of field)
	Params integrator;                 // the integrator's ParamMap ("type" included)
	Params render;                     // AA_* etc. by their reference names
	int srand_seed;                    // srand() before render(): the libc state integrator_tiled.cc:319 draws tile seeds from
	double background[3];
};

static const int W = 20, H = 16, TILE = 8;

struct Emit
{
	std::string s;
	void raw(const std::string &t) { s += t; }
	void arr_u32(const char *k, const std::vector<uint32_t> &v)
	{
		char b[32];
		s += "\""; s += k; s += "\": [";
		for(size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof b, "%s%u", i ? "," : "", v[i]); s += b; }
		s += "]";
	}
	void arr_i32(const char *k, const std::vector<int> &v)
	{
		char b[32];
		s += "\""; s += k; s += "\": [";
		for(size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof b, "%s%d", i ? "," : "", v[i]); s += b; }
		s += "]";
	}
};

static int pint(const Params &p, const char *n, int def) { const P *q = find(p, n); return q ? q->i : def; }
static double pflt(const Params &p, const char *n, double def) { const P *q = find(p, n); return q ? q->f : def; }
static bool pbool(const Params &p, const char *n, bool def) { const P *q = find(p, n); return q ? q->b : def; }

static void run_case(Emit &out, const Case &cs, bool first)
{
	// geometry with the case's slot materials
	g_tris.clear();
	std::vector<int> tri_mat;
	for(const Quad &q : g_quads)
	{
		const int mat = q.slot < 0 ? -q.slot - 1 : cs.slot_mat[q.slot];
		const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}};
		for(int k = 0; k < 2; ++k)
		{
			HTri t;
			t.a = Point3((float)q.p[idx[k][0]][0], (float)q.p[idx[k][0]][1], (float)q.p[idx[k][0]][2]);
			t.b = Point3((float)q.p[idx[k][1]][0], (float)q.p[idx[k][1]][1], (float)q.p[idx[k][1]][2]);
			t.c = Point3((float)q.p[idx[k][2]][0], (float)q.p[idx[k][2]][1], (float)q.p[idx[k][2]][2]);
			t.e1 = t.b - t.a; t.e2 = t.c - t.a;                                                        // triangle.h:203-204
			t.eps = 0.1f * H_MIN_RAYDIST * std::max(t.e1.length(), t.e2.length());                     // triangle.h:206
			t.ng = ((t.b - t.a) ^ (t.c - t.a)).normalize();                                            // recNormal, triangle.h:295-302
			t.mat = g_mats[mat];
			g_tris.push_back(t);
			tri_mat.push_back(mat);
		}
	}

	Scene scene(nullptr);
	scene.shadow_bias_auto_ = pbool(cs.render, "adv_auto_shadow_bias_enabled", true);
	scene.shadow_bias_ = scene.shadow_bias_auto_ ? (float)H_SHADOW_BIAS : (float)pflt(cs.render, "adv_shadow_bias_value", H_SHADOW_BIAS);      // scene.cc:825
	scene.ray_min_dist_auto_ = pbool(cs.render, "adv_auto_min_raydist_enabled", true);
	scene.ray_min_dist_ = scene.ray_min_dist_auto_ ? (float)H_MIN_RAYDIST : (float)pflt(cs.render, "adv_min_raydist_value", H_MIN_RAYDIST);   // scene.cc:826

	// lights
	std::vector<Params> lights;
	for(int li : cs.lights)
	{
		Params lp = g_light_params[li];
		for(const P &o : cs.light_override)      // "<light index>:<name>"
		{
			const size_t colon = o.name.find(':');
			if(std::atoi(o.name.substr(0, colon).c_str()) != li) continue;
			const std::string key = o.name.substr(colon + 1);
			bool done = false;
			for(P &q : lp) if(q.name == key) { P r = o; r.name = key; q = r; done = true; }
			if(!done) { P r = o; r.name = key; lp.push_back(r); }
		}
		lights.push_back(lp);
		scene.lights_.push_back(make_light(lp));
	}

	// camera
	Params cam = {ps("type", "perspective"), pv("from", 0.13, -3.8, 0.21), pv("to", 0.02, 0.0, -0.06), pv("up", 0.13, -3.8, 1.21), pi("resx", W), pi("resy", H), pf("focal", 1.35)};
	for(const P &p : cs.camera) cam.push_back(p);
	ParamMap cam_map = to_map(cam);
	Camera *camera = PerspectiveCamera::factory(cam_map, fake_env());
	scene.setCamera(camera);

	// background, volume integrator
	Params bg = {ps("type", "constant"), pv("color", cs.background[0], cs.background[1], cs.background[2])};
	ParamMap bg_map = to_map(bg);
	scene.setBackground(ConstantBackground::factory(bg_map, fake_env()));
	ParamMap none;
	scene.vol_integrator_ = static_cast<VolumeIntegrator *>(EmptyVolumeIntegrator::factory(none, fake_env()));

	// anti-aliasing parameters: Scene::setAntialiasing's job (scene.cc:761-778; defaults environment.cc:682-695)
	g_aa.samples = std::max(1, pint(cs.render, "AA_minsamples", 1));
	g_aa.passes = pint(cs.render, "AA_passes", 1);
	g_aa.inc_samples = pint(cs.render, "AA_inc_samples", 0) > 0 ? pint(cs.render, "AA_inc_samples", 0) : g_aa.samples;
	g_aa.threshold = (float)pflt(cs.render, "AA_threshold", 0.0);
	g_aa.resampled_floor = (float)pflt(cs.render, "AA_resampled_floor", 0.0);
	g_aa.sample_mult = (float)pflt(cs.render, "AA_sample_multiplier_factor", 1.0);
	g_aa.light_mult = (float)pflt(cs.render, "AA_light_sample_multiplier_factor", 1.0);
	g_aa.indirect_mult = (float)pflt(cs.render, "AA_indirect_sample_multiplier_factor", 1.0);
	g_aa.detect_color_noise = false; g_aa.dark_type = DarkDetectionType::None; g_aa.dark_factor = 0.f; g_aa.var_edge = 10; g_aa.var_pixels = 0;
	g_aa.clamp_samples = 0.f; g_aa.clamp_indirect = 0.f;

	// the integrator: the reference's factory, preprocess() and render()
	ParamMap integ_map = to_map(cs.integrator);
	const std::string itype = find(cs.integrator, "type")->s;
	Integrator *integ = itype == "directlighting" ? DirectLightIntegrator::factory(integ_map, fake_env()) : PathIntegrator::factory(integ_map, fake_env());
	SurfaceIntegrator *surf = static_cast<SurfaceIntegrator *>(integ);
	surf->setScene(&scene);
	surf->preprocess();

	alignas(64) static unsigned char fake_output[256];
	// the film window (render parameters width / height / xstart / ystart, environment.cc:747-774) and the tile size
	ImageFilm film(pint(cs.render, "width", W), pint(cs.render, "height", H), pint(cs.render, "xstart", 0), pint(cs.render, "ystart", 0),
	               *reinterpret_cast<ColorOutput *>(fake_output), 1.f, ImageFilm::FilterType::Box, nullptr, false, pint(cs.render, "tile_size", TILE), ImageSplitter::Linear, false);
	film.setBaseSamplingOffset((unsigned)pint(cs.render, "adv_base_sampling_offset", 0));

	g_samples.clear(); g_tile_log.clear(); g_ray_log.clear(); g_ray_log_cap = 1500; g_n_closest = g_n_shadow = 0;
	srand((unsigned)cs.srand_seed);
	surf->render(0, &film);

	// ---- emit
	char b[256];
	if(!first) out.raw(",\n");
	out.raw("{\"name\": \""); out.raw(cs.name); out.raw("\",\n");
	out.arr_i32("tri_mat", tri_mat); out.raw(",\n");
	out.raw("\"lights\": [");
	for(size_t i = 0; i < lights.size(); ++i) { if(i) out.raw(", "); out.raw(json_params(lights[i])); }
	out.raw("],\n\"camera\": "); out.raw(json_params(cam));
	out.raw(",\n\"integrator\": "); out.raw(json_params(cs.integrator));
	out.raw(",\n\"render\": "); out.raw(json_params(cs.render));
	snprintf(b, sizeof b, ",\n\"background\": [%.17g, %.17g, %.17g],\n\"srand\": %d, \"n_closest\": %llu, \"n_shadow\": %llu,\n",
	         cs.background[0], cs.background[1], cs.background[2], cs.srand_seed, (unsigned long long)g_n_closest, (unsigned long long)g_n_shadow);
	out.raw(b);
	out.arr_i32("tiles4", g_tile_log); out.raw(",\n");
	std::vector<int> xy; std::vector<uint32_t> sm;
	for(const SampleRec &r : g_samples)
	{
		xy.push_back(r.x); xy.push_back(r.y);
		sm.push_back(f2u(r.dx)); sm.push_back(f2u(r.dy));
		for(int k = 0; k < 4; ++k) sm.push_back(f2u(r.c[k]));
	}
	out.arr_i32("sample_xy", xy); out.raw(",\n");
	out.arr_u32("sample_dxdy_rgba", sm); out.raw(",\n");
	std::vector<uint32_t> rays; std::vector<int> ray_tri;
	for(const RayRec &r : g_ray_log)
	{
		for(int k = 0; k < 3; ++k) rays.push_back(f2u(r.from[k]));
		for(int k = 0; k < 3; ++k) rays.push_back(f2u(r.dir[k]));
		rays.push_back(f2u(r.tmin)); rays.push_back(f2u(r.tmax_in)); rays.push_back(f2u(r.t));
		ray_tri.push_back(r.tri);
	}
	out.arr_u32("closest_rays9", rays); out.raw(",\n");
	out.arr_i32("closest_tri", ray_tri);
	out.raw("}");
	fprintf(stderr, "case %-22s samples %6zu  closest %8llu  shadow %8llu\n", cs.name, g_samples.size(), (unsigned long long)g_n_closest, (unsigned long long)g_n_shadow);

	delete camera;
	for(Light *l : scene.lights_) delete l;
}

int main()
{
	logger__.setConsoleMasterVerbosity("mute");
	logger__.setLogMasterVerbosity("mute");
	g_passes = new RenderPasses();
	build_catalogue();

	std::vector<Case> cases;
	{	// path samples > 1, four bounces (QMC dimensions up to 4*3+4 = 16), one area light with two samples: light- and BSDF-sampling halves of the MIS estimate
		// on diffuse, Oren-Nayar and glossy (as_diffuse) surfaces; an emitting shinydiffuse sheet; roulette off
		Case c; c.name = "pt_mis_paths";
		c.slot_mat[SLOT_A] = M_GLOSSY; c.slot_mat[SLOT_B] = M_WHITE; c.slot_mat[SLOT_C] = M_SD_EMIT; c.slot_mat[SLOT_D] = M_GREEN_ON;
		c.lights = {0};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 3), pi("bounces", 4), pi("russian_roulette_min_bounces", 4), pi("raydepth", 2), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 4)};
		c.srand_seed = 1; c.background[0] = 0.05; c.background[1] = 0.07; c.background[2] = 0.1;
		cases.push_back(c);
	}
	{	// three lights (two area lights with different sample counts + a point light): the one-light counter; roulette from the second bounce on,
		// six bounces; a coated-glossy sheet (path-traced lobes + specular coat through recursiveRaytrace); a base sampling offset
		Case c; c.name = "pt_three_lights_rr";
		c.slot_mat[SLOT_A] = M_GLOSSY; c.slot_mat[SLOT_B] = M_RED; c.slot_mat[SLOT_C] = M_COATED; c.slot_mat[SLOT_D] = M_WHITE;
		c.lights = {0, 1, 2};
		c.light_override = {pi("0:samples", 1)};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 6), pi("russian_roulette_min_bounces", 1), pi("raydepth", 2), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 3), pi("adv_base_sampling_offset", 37)};
		c.srand_seed = 7; c.background[0] = 0.0; c.background[1] = 0.0; c.background[2] = 0.0;
		cases.push_back(c);
	}
	{	// recursiveRaytrace: glass box, shinydiffuse with mirror + transparency, a mirror sheet, a glossy-recursive (as_diffuse off) sheet; raydepth 3;
		// transparent background with refraction alpha
		Case c; c.name = "pt_recursive";
		c.slot_mat[SLOT_A] = M_GLASS; c.slot_mat[SLOT_B] = M_SD_MIRROR_TRANSP; c.slot_mat[SLOT_C] = M_MIRROR; c.slot_mat[SLOT_D] = M_GLOSSY_REC;
		c.lights = {0, 2};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 2), pi("bounces", 3), pi("russian_roulette_min_bounces", 3), pi("raydepth", 3), ps("caustic_type", "none"),
		                pb("bg_transp", true), pb("bg_transp_refract", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 3; c.background[0] = 0.2; c.background[1] = 0.25; c.background[2] = 0.3;
		cases.push_back(c);
	}
	{	// three adaptive passes (every pixel resampled): riVdC / riS sub-pixel positions, growing sample counts and light-sample multiplier,
		// two lights and roulette so that both serial states run on across passes
		Case c; c.name = "pt_multipass";
		c.slot_mat[SLOT_A] = M_GLOSSY; c.slot_mat[SLOT_B] = M_WHITE; c.slot_mat[SLOT_C] = M_RED; c.slot_mat[SLOT_D] = M_GREEN_ON;
		c.lights = {0, 1};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 4), pi("russian_roulette_min_bounces", 2), pi("raydepth", 2), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 3), pi("AA_minsamples", 2), pi("AA_inc_samples", 2), pf("AA_threshold", 0.0), pf("AA_sample_multiplier_factor", 1.5), pf("AA_light_sample_multiplier_factor", 1.5)};
		c.srand_seed = 11; c.background[0] = 0.0; c.background[1] = 0.0; c.background[2] = 0.0;
		cases.push_back(c);
	}
	{	// depth of field: the per-pixel lens streams of renderTile
		Case c; c.name = "pt_dof";
		c.slot_mat[SLOT_A] = M_WHITE; c.slot_mat[SLOT_B] = M_GLOSSY; c.slot_mat[SLOT_C] = M_GREEN_ON; c.slot_mat[SLOT_D] = M_RED;
		c.lights = {0};
		c.light_override = {pi("0:samples", 1)};
		c.camera = {pf("aperture", 0.06), pf("dof_distance", 3.6), ps("bokeh_type", "disk1")};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 2), pi("russian_roulette_min_bounces", 2), pi("raydepth", 2), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 5)};
		c.srand_seed = 5; c.background[0] = 0.1; c.background[1] = 0.1; c.background[2] = 0.1;
		cases.push_back(c);
	}
	{	// the direct-lighting integrator (tests/test01's) with all three lights and recursion through mirror and glass
		Case c; c.name = "directlighting";
		c.slot_mat[SLOT_A] = M_GLOSSY; c.slot_mat[SLOT_B] = M_SD_MIRROR_TRANSP; c.slot_mat[SLOT_C] = M_MIRROR; c.slot_mat[SLOT_D] = M_GLASS;
		c.lights = {0, 1, 2};
		c.integrator = {ps("type", "directlighting"), pi("raydepth", 2), pb("caustics", false), pb("do_AO", false)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 3)};
		c.srand_seed = 2; c.background[0] = 0.02; c.background[1] = 0.03; c.background[2] = 0.05;
		cases.push_back(c);
	}
	{	// transparent shadows: a transparent shinydiffuse sheet hangs between light 1 and the room
		Case c; c.name = "pt_transparent_shadows";
		c.slot_mat[SLOT_A] = M_WHITE; c.slot_mat[SLOT_B] = M_GLOSSY; c.slot_mat[SLOT_C] = M_RED; c.slot_mat[SLOT_D] = M_SD_TRANSP;
		c.lights = {0, 2};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 3), pi("russian_roulette_min_bounces", 3), pi("raydepth", 2), ps("caustic_type", "none"),
		                pb("transpShad", true), pi("shadowDepth", 3)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 4; c.background[0] = 0.0; c.background[1] = 0.0; c.background[2] = 0.0;
		cases.push_back(c);
	}
	{	// no_recursive: every lobe is path-traced (path_flags = BsdfAll), specular ones included
		Case c; c.name = "pt_no_recursive";
		c.slot_mat[SLOT_A] = M_MIRROR; c.slot_mat[SLOT_B] = M_SD_MIRROR_TRANSP; c.slot_mat[SLOT_C] = M_COATED; c.slot_mat[SLOT_D] = M_WHITE;
		c.lights = {0};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 2), pi("bounces", 4), pi("russian_roulette_min_bounces", 0), pi("raydepth", 2), ps("caustic_type", "none"),
		                pb("no_recursive", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2), pb("adv_auto_shadow_bias_enabled", false), pf("adv_shadow_bias_value", 0.001),
		            pb("adv_auto_min_raydist_enabled", false), pf("adv_min_raydist_value", 0.0001)};
		c.srand_seed = 9; c.background[0] = 0.0; c.background[1] = 0.0; c.background[2] = 0.0;
		cases.push_back(c);
	}

	{	// glass with absorption (a Beer volume inside the material: recursive rays and path segments that ran inside it), the anisotropic lobe,
		// a glossy-recursive coated material with an Oren-Nayar substrate, translucency + emission
		Case c; c.name = "pt_absorption_aniso";
		c.slot_mat[SLOT_A] = M_GLASS_ABS; c.slot_mat[SLOT_B] = M_ANISO; c.slot_mat[SLOT_C] = M_COATED_REC; c.slot_mat[SLOT_D] = M_SD_TRANSL;
		c.lights = {0, 2};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 2), pi("bounces", 4), pi("russian_roulette_min_bounces", 4), pi("raydepth", 3), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 6; c.background[0] = 0.1; c.background[1] = 0.12; c.background[2] = 0.15;
		if(const char *e = getenv("YAF_ABLATE")) for(int k = 0; k < N_SLOTS; ++k) if(k != atoi(e)) c.slot_mat[k] = M_WHITE;      // (debugging aid: keep one slot's material)
		cases.push_back(c);
	}
	{	// material switches the integrators read: additionaldepth (raydepth 1 + 2), the transparent bias, visibility no_shadows / shadow_only,
		// receive_shadows off, a light that casts no shadows; roulette on; a film window off the origin with an odd tile size
		Case c; c.name = "pt_depth_bias_visibility";
		c.slot_mat[SLOT_A] = M_SD_DEPTH; c.slot_mat[SLOT_B] = M_SD_NOSHADOW; c.slot_mat[SLOT_C] = M_SD_SHADOWONLY; c.slot_mat[SLOT_D] = M_SD_NORECV;
		c.lights = {0, 1};
		c.light_override = {pb("1:cast_shadows", false), pi("1:samples", 1)};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 4), pi("russian_roulette_min_bounces", 1), pi("raydepth", 1), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 3), pi("width", 14), pi("height", 11), pi("xstart", 3), pi("ystart", 2), pi("tile_size", 5)};
		c.srand_seed = 8; c.background[0] = 0.0; c.background[1] = 0.0; c.background[2] = 0.0;
		cases.push_back(c);
	}
	{	// direct lighting with transparent shadows through fake-shadow glass (a filter lobe), a flat material, the glossy-recursive branch under directlighting
		Case c; c.name = "dl_fake_shadows_flat";
		c.slot_mat[SLOT_A] = M_SD_FLAT; c.slot_mat[SLOT_B] = M_GLOSSY_REC; c.slot_mat[SLOT_C] = M_WHITE; c.slot_mat[SLOT_D] = M_GLASS_FAKE;
		c.lights = {0, 2};
		c.integrator = {ps("type", "directlighting"), pi("raydepth", 2), pb("caustics", false), pb("do_AO", false), pb("transpShad", true), pi("shadowDepth", 2), pb("bg_transp", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 10; c.background[0] = 0.3; c.background[1] = 0.2; c.background[2] = 0.1;
		cases.push_back(c);
	}
	{	// materials with nothing to sample (the path runs straight on with the previous direction and weight, integrator_path_tracer.cc:243-249) and with a
		// mirror lobe alone; clipping planes on the camera (camera rays with tmin / tmax); path samples 2, five bounces, roulette from depth 3
		Case c; c.name = "pt_degenerate_lobes_clip";
		c.slot_mat[SLOT_A] = M_SD_NOLOBE; c.slot_mat[SLOT_B] = M_SD_MIRRORONLY; c.slot_mat[SLOT_C] = M_GLOSSY; c.slot_mat[SLOT_D] = M_SD_NOLOBE;
		c.lights = {0, 1};
		c.camera = {pf("nearClip", 3.05), pf("farClip", 5.3)};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 2), pi("bounces", 5), pi("russian_roulette_min_bounces", 2), pi("raydepth", 2), ps("caustic_type", "none")};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 12; c.background[0] = 0.2; c.background[1] = 0.1; c.background[2] = 0.3;
		cases.push_back(c);
	}
	{	// rough glass: the reflect + transmit case of recursiveRaytrace's glossy branch (two trajectories per glossy sample), a box of it (rays inside:
		// total inner reflection about the half vector) and an absorbing, fake-shadow sheet under transparent shadows; as a path vertex it is sampled through BsdfAll
		Case c; c.name = "pt_rough_glass";
		c.slot_mat[SLOT_A] = M_ROUGH_GLASS; c.slot_mat[SLOT_B] = M_WHITE; c.slot_mat[SLOT_C] = M_GLOSSY; c.slot_mat[SLOT_D] = M_ROUGH_GLASS_ABS_FAKE;
		c.lights = {0, 2};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 3), pi("russian_roulette_min_bounces", 3), pi("raydepth", 2), ps("caustic_type", "none"),
		                pb("transpShad", true), pi("shadowDepth", 3), pb("bg_transp_refract", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 13; c.background[0] = 0.15; c.background[1] = 0.2; c.background[2] = 0.25;
		cases.push_back(c);
	}
	{	// path caustics, the factory's default (no caustic_type parameter: the constructor's Path stays, integrator_path_tracer.cc:36): after a bounce
		// through a specular, glossy or filter lobe the next vertex shows its lights (include_lights_, :252-253) and adds its emission after the
		// roulette test (:290); glass, a shinydiffuse with mirror and transparency, an emitting shinydiffuse, a glossy-recursive sheet; roulette from bounce 2
		Case c; c.name = "pt_caustics_default";
		c.slot_mat[SLOT_A] = M_GLASS; c.slot_mat[SLOT_B] = M_SD_MIRROR_TRANSP; c.slot_mat[SLOT_C] = M_SD_EMIT; c.slot_mat[SLOT_D] = M_GLOSSY_REC;
		c.lights = {0, 1};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 2), pi("bounces", 5), pi("russian_roulette_min_bounces", 1), pi("raydepth", 2)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 3)};
		c.srand_seed = 17; c.background[0] = 0.1; c.background[1] = 0.1; c.background[2] = 0.12;
		cases.push_back(c);
	}
	{	// ... and spelled out ("path" is no value the factory knows: the default stays), with no_recursive (every lobe path-traced from the camera hit on),
		// a mirror sheet, coated glossy and rough glass as path vertices, three lights, six bounces without roulette
		Case c; c.name = "pt_caustics_path_no_recursive";
		c.slot_mat[SLOT_A] = M_ROUGH_GLASS; c.slot_mat[SLOT_B] = M_COATED; c.slot_mat[SLOT_C] = M_MIRROR; c.slot_mat[SLOT_D] = M_SD_EMIT;
		c.lights = {0, 1, 2};
		c.integrator = {ps("type", "pathtracing"), pi("path_samples", 1), pi("bounces", 6), pi("russian_roulette_min_bounces", 6), pi("raydepth", 1), ps("caustic_type", "path"),
		                pb("no_recursive", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 19; c.background[0] = 0.0; c.background[1] = 0.02; c.background[2] = 0.05;
		cases.push_back(c);
	}
	{	// direct lighting through rough glass: DirectLightIntegrator::integrate -> recursiveRaytrace's reflect + transmit glossy case at raydepth 3 (nested:
		// a trajectory's children split into one trajectory each), all three lights estimated at every hit, transparent shadows through the fake-shadow sheet
		Case c; c.name = "dl_rough_glass";
		c.slot_mat[SLOT_A] = M_ROUGH_GLASS; c.slot_mat[SLOT_B] = M_SD_MIRROR_TRANSP; c.slot_mat[SLOT_C] = M_GLOSSY_REC; c.slot_mat[SLOT_D] = M_ROUGH_GLASS_ABS_FAKE;
		c.lights = {0, 1, 2};
		c.integrator = {ps("type", "directlighting"), pi("raydepth", 3), pb("caustics", false), pb("do_AO", false), pb("transpShad", true), pi("shadowDepth", 4),
		                pb("bg_transp_refract", true)};
		c.render = {pi("AA_passes", 1), pi("AA_minsamples", 2)};
		c.srand_seed = 23; c.background[0] = 0.3; c.background[1] = 0.2; c.background[2] = 0.1;
		cases.push_back(c);
	}

	Emit out;
	out.raw("{\n\"width\": "); out.raw(std::to_string(W)); out.raw(", \"height\": "); out.raw(std::to_string(H)); out.raw(", \"tile_size\": "); out.raw(std::to_string(TILE));
	out.raw(",\n\"materials\": [");
	for(size_t i = 0; i < g_mat_params.size(); ++i) { if(i) out.raw(",\n "); out.raw(json_params(g_mat_params[i])); }
	out.raw("],\n");
	std::vector<uint32_t> verts;
	for(const Quad &q : g_quads)
	{
		const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}};
		for(int k = 0; k < 2; ++k) for(int c = 0; c < 3; ++c) for(int a = 0; a < 3; ++a) verts.push_back(f2u((float)q.p[idx[k][c]][a]));
	}
	out.arr_u32("verts", verts);
	out.raw(",\n\"cases\": [\n");
	for(size_t i = 0; i < cases.size(); ++i) run_case(out, cases[i], i == 0);
	out.raw("\n]\n}\n");
	fwrite(out.s.data(), 1, out.s.size(), stdout);
	return 0;
}
