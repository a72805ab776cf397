// TEST INFRASTRUCTURE — not product code.
//
// Component-level reference harness.  This translation unit is *our* driver; it is
// compiled (by oracle/Makefile, target `ref`) against the reference's own headers and
// a handful of the reference's own .cc files *where they lie* under /root/reference.
// Nothing from the reference is copied; only the components that build from their own
// sources with plain g++ (no cmake-generated header, no external library) are used:
// fast-math, QMC, vector/bound utilities, perspective camera, area/point lights,
// shinydiffuse/glossy/light materials.  The kd-tree, Triangle, Scene and ImageFilm
// need the cmake-generated yafaray_config.h and are therefore NOT built.
//
// Output: one JSON document on stdout holding inputs and the reference's outputs as
// raw IEEE-754 bit patterns (u32 for float, two u32 for double) so nothing is lost in
// printing.  tests/golden/make_golden.py turns it into tests/golden/ref_components.json.

#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>
#include <list>

#include "constants.h"
#include "utility/util_math_optimizations.h"
#include "utility/util_mcqmc.h"
#include "common/vector.h"
#include "common/scr_halton.h"
#include "utility/util_sample.h"
#include "common/bound.h"
#include "common/ray.h"
#include "common/color.h"
#include "common/param.h"
#include "common/surface.h"
#include "common/scene.h"
#include "camera/camera_perspective.h"
#include "material/material_glass.h"
#include "material/material_coated_glossy.h"
#include "light/light_area.h"
#include "light/light_point.h"
#include "material/material_shiny_diffuse.h"
#include "material/material_glossy.h"
#include "material/material_simple.h"
#include "material/material_rough_glass.h"
#include "volume/volumehandler_beer.h"

using namespace yafaray4;

// ---------------------------------------------------------------- tiny helpers
static uint32_t lcg_state = 12345u;
static uint32_t lcg() { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state; }
static float urand() { return (float)((lcg() >> 8) * (1.0 / 16777216.0)); }          // [0,1)
static float srand11() { return 2.f * urand() - 1.f; }                                 // [-1,1)

static uint32_t f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }

struct Json
{
	std::string s;
	bool first = true;
	void key(const char *k) { if(!first) s += ",\n"; first = false; s += "\""; s += k; s += "\": "; }
	void arr_u32(const char *k, const std::vector<uint32_t> &v)
	{
		key(k); s += "[";
		char b[32];
		for(size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof b, "%s%u", i ? "," : "", v[i]); s += b; }
		s += "]";
	}
	void arr_i32(const char *k, const std::vector<int> &v)
	{
		key(k); s += "[";
		char b[32];
		for(size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof b, "%s%d", i ? "," : "", v[i]); s += b; }
		s += "]";
	}
};

static void pushv(std::vector<uint32_t> &o, const Vec3 &v) { o.push_back(f2u(v.x_)); o.push_back(f2u(v.y_)); o.push_back(f2u(v.z_)); }
static void pushp(std::vector<uint32_t> &o, const Point3 &v) { o.push_back(f2u(v.x_)); o.push_back(f2u(v.y_)); o.push_back(f2u(v.z_)); }
static void pushc(std::vector<uint32_t> &o, const Rgb &c) { o.push_back(f2u(c.r_)); o.push_back(f2u(c.g_)); o.push_back(f2u(c.b_)); }

static Vec3 rand_unit()
{
	for(;;)
	{
		Vec3 v(srand11(), srand11(), srand11());
		float l = v.lengthSqr();
		if(l > 0.01f && l < 1.f) { v.normalize(); return v; }
	}
}

// a surface point with shading frame, the way Triangle::getSurface leaves it (flat shading)
static void make_sp(SurfacePoint &sp, const Vec3 &n, const Point3 &p, bool tilt_ng)
{
	sp.n_ = n;
	sp.ng_ = n;
	if(tilt_ng)
	{
		Vec3 t = n + 0.2f * rand_unit();
		t.normalize();
		sp.ng_ = t;
	}
	sp.p_ = p;
	createCs__(sp.n_, sp.nu_, sp.nv_);
	sp.u_ = sp.v_ = 0.f;
	sp.has_uv_ = false; sp.has_orco_ = false;
	sp.material_ = nullptr; sp.light_ = nullptr; sp.object_ = nullptr; sp.origin_ = nullptr;
	sp.prim_num_ = 0;
	sp.dp_du_ = sp.nu_; sp.dp_dv_ = sp.nv_;
	sp.dp_du_abs_ = sp.nu_; sp.dp_dv_abs_ = sp.nv_;
	sp.ds_du_ = Vec3(1, 0, 0); sp.ds_dv_ = Vec3(0, 1, 0);
	sp.orco_p_ = p; sp.orco_ng_ = sp.ng_;
}

// The factories take a RenderEnvironment& that is never touched when the shader-node
// list is empty (material_node.cc:141-148 iterates an empty list).  RenderEnvironment
// itself cannot be built here (environment.cc needs the generated config header), so the
// reference is bound to raw storage that is never read.
alignas(64) static unsigned char fake_env_storage[1 << 16];
static RenderEnvironment &fake_env() { return *reinterpret_cast<RenderEnvironment *>(fake_env_storage); }

// ---------------------------------------------------------------- sections
static void sec_fastmath(Json &j)
{
	std::vector<uint32_t> x, fsin, fcos, fexp2, flog2, fsqrt, pa, pb, fpow, facos;
	for(int i = 0; i < 600; ++i)
	{
		float v;
		if(i < 200) v = srand11() * 3.2f;
		else if(i < 400) v = srand11() * 20.f;
		else v = srand11() * 7.f;
		if(i == 0) v = 0.f;
		if(i == 1) v = (float)M_PI;
		if(i == 2) v = -(float)M_PI;
		if(i == 3) v = (float)M_2PI;
		if(i == 4) v = (float)M_PI_2;
		x.push_back(f2u(v));
		fsin.push_back(f2u(fSin__(v)));
		fcos.push_back(f2u(fCos__(v)));
		fexp2.push_back(f2u(fExp2__(v)));
		float pv = std::fabs(v) + 1e-3f;
		flog2.push_back(f2u(fLog2__(pv)));
		fsqrt.push_back(f2u(fSqrt__(pv)));
		facos.push_back(f2u(fAcos__(v * 0.2f)));
	}
	for(int i = 0; i < 400; ++i)
	{
		float a = urand();
		float b = (i < 200) ? urand() * 200.f : urand() * 4.f;
		if(i == 0) { a = 1.f; b = 50.f; }
		if(i == 1) { a = 0.f; b = 50.f; }
		pa.push_back(f2u(a)); pb.push_back(f2u(b));
		fpow.push_back(f2u(fPow__(a, b)));
	}
	j.arr_u32("fm_x", x); j.arr_u32("fm_sin", fsin); j.arr_u32("fm_cos", fcos);
	j.arr_u32("fm_exp2", fexp2); j.arr_u32("fm_log2_absx", flog2); j.arr_u32("fm_sqrt_absx", fsqrt);
	j.arr_u32("fm_acos_02x", facos);
	j.arr_u32("fm_pow_a", pa); j.arr_u32("fm_pow_b", pb); j.arr_u32("fm_pow", fpow);
}

static void sec_qmc(Json &j)
{
	std::vector<uint32_t> bits, r, vdc, ris, rilp, fnv;
	for(int i = 0; i < 400; ++i)
	{
		uint32_t b = (i < 64) ? (uint32_t)i : lcg();
		uint32_t rr = (i % 3 == 0) ? 0u : lcg();
		bits.push_back(b); r.push_back(rr);
		vdc.push_back(f2u(riVdC__(b, rr)));
		ris.push_back(f2u(riS__(b, rr)));
		rilp.push_back(f2u(riLp__(b, rr)));
		fnv.push_back(fnv32ABuf__(b));
	}
	j.arr_u32("q_bits", bits); j.arr_u32("q_r", r); j.arr_u32("q_vdc", vdc); j.arr_u32("q_ris", ris);
	j.arr_u32("q_rilp", rilp); j.arr_u32("q_fnv", fnv);

	// scrambled Halton: every dimension 1..49, a spread of n; result narrowed to float as the
	// integrators do (float s = scrHalton__(...)), plus the raw double
	std::vector<uint32_t> sd, sn, sf, sdlo, sdhi;
	for(int dim = 1; dim < 50; ++dim) // dim 0 has base 1: scrHalton__ never terminates there (and is never used)
		for(int k = 0; k < 24; ++k)
		{
			uint32_t n = (k < 8) ? (uint32_t)k : ((k < 16) ? (lcg() >> 12) : lcg());
			double v = scrHalton__(dim, n);
			union { double d; uint32_t u[2]; } c; c.d = v;
			sd.push_back(dim); sn.push_back(n); sf.push_back(f2u((float)v)); sdlo.push_back(c.u[0]); sdhi.push_back(c.u[1]);
		}
	j.arr_u32("sh_dim", sd); j.arr_u32("sh_n", sn); j.arr_u32("sh_f32", sf); j.arr_u32("sh_f64lo", sdlo); j.arr_u32("sh_f64hi", sdhi);
	// every first digit of every dimension: exercises each entry of each Faure permutation
	std::vector<uint32_t> fd;
	for(int dim = 1; dim < 50; ++dim)
		for(int n = 0; n < prims__[dim]; ++n) fd.push_back(f2u((float)scrHalton__(dim, (unsigned)n)));
	j.arr_u32("sh_firstdigit_f32", fd);

	// incremental Halton: setStart(i) then 6 getNext()
	std::vector<uint32_t> hb, hs, hv;
	const int bases[3] = {2, 3, 5};
	for(int bi = 0; bi < 3; ++bi)
		for(int k = 0; k < 60; ++k)
		{
			uint32_t st = (k < 20) ? (uint32_t)k : ((k < 40) ? (lcg() >> 10) : lcg());
			if(k == 20) st = 0xFFFFFFFFu; // offs-1 wrap of doLightEstimation
			Halton h(bases[bi]);
			h.setStart(st);
			hb.push_back(bases[bi]); hs.push_back(st);
			for(int q = 0; q < 6; ++q) hv.push_back(f2u(h.getNext()));
		}
	j.arr_u32("h_base", hb); j.arr_u32("h_start", hs); j.arr_u32("h_next6", hv);

	// MWC PRNG
	std::vector<uint32_t> seeds, rv;
	for(int k = 0; k < 8; ++k)
	{
		uint32_t seed = (k == 0) ? 123u : lcg();
		Random prng(seed);
		seeds.push_back(seed);
		for(int q = 0; q < 8; ++q) rv.push_back(f2u((float)prng()));
	}
	j.arr_u32("rng_seed", seeds); j.arr_u32("rng_f32x8", rv);
}

static void sec_geom(Json &j)
{
	// createCs + sampleCosHemisphere
	std::vector<uint32_t> n_in, s_in, cs_out, hemi_out;
	for(int i = 0; i < 200; ++i)
	{
		Vec3 n = rand_unit();
		if(i == 0) n = Vec3(0, 0, 1);
		if(i == 1) n = Vec3(0, 0, -1);
		if(i == 2) n = Vec3(1, 0, 0);
		float s_1 = urand(), s_2 = urand();
		if(i == 3) s_1 = 1.f;
		Vec3 u, v;
		createCs__(n, u, v);
		Vec3 w = sampleCosHemisphere__(n, u, v, s_1, s_2);
		pushv(n_in, n); s_in.push_back(f2u(s_1)); s_in.push_back(f2u(s_2));
		pushv(cs_out, u); pushv(cs_out, v); pushv(hemi_out, w);
	}
	j.arr_u32("g_n", n_in); j.arr_u32("g_s12", s_in); j.arr_u32("g_cs_uv", cs_out); j.arr_u32("g_coshemi", hemi_out);

	// Bound::cross
	std::vector<uint32_t> b_in, b_out;
	std::vector<int> b_hit;
	Bound bnd(Point3(-1.f, -0.5f, -2.f), Point3(1.5f, 0.75f, 0.25f));
	for(int i = 0; i < 300; ++i)
	{
		Point3 from(srand11() * 3.f, srand11() * 3.f, srand11() * 3.f);
		Vec3 dir = rand_unit();
		if(i % 17 == 0) { dir = Vec3(0, 0, 1); }
		if(i % 19 == 0) { dir = Vec3(1, 0, 0); }
		float dist = (i % 5 == 0) ? urand() * 3.f : std::numeric_limits<float>::infinity();
		Ray ray(from, dir);
		float a = -7.f, b = -7.f;
		bool hit = bnd.cross(ray, a, b, dist);
		pushp(b_in, from); pushv(b_in, dir); b_in.push_back(f2u(dist));
		b_hit.push_back(hit ? 1 : 0); b_out.push_back(f2u(a)); b_out.push_back(f2u(b));
	}
	j.arr_u32("bc_in7", b_in); j.arr_i32("bc_hit", b_hit); j.arr_u32("bc_ab", b_out);
}

static void sec_camera(Json &j)
{
	std::vector<uint32_t> cam_in, pxy, out;
	for(int c = 0; c < 4; ++c)
	{
		ParamMap pm;
		Point3 from(0.f, -4.f, 1.f), to(0.f, 0.f, 0.f), up(0.f, -4.f, 2.f);
		int resx = 512, resy = 512;
		float focal = 1.1f;
		if(c == 1) { from = Point3(3.f, 2.f, 1.5f); to = Point3(0.1f, -0.2f, 0.3f); up = Point3(3.f, 2.f, 2.5f); resx = 480; resy = 270; focal = 1.09f; }
		if(c == 2) { from = Point3(-2.f, 5.f, 3.f); to = Point3(0.f, 0.f, 0.5f); up = Point3(-2.f, 5.f, 4.f); resx = 256; resy = 256; focal = 0.8f; }
		if(c == 3) { from = Point3(0.f, 0.f, 6.f); to = Point3(0.f, 0.f, 0.f); up = Point3(0.f, 1.f, 6.f); resx = 1024; resy = 768; focal = 2.f; }
		pm["from"] = Parameter(from); pm["to"] = Parameter(to); pm["up"] = Parameter(up);
		pm["resx"] = Parameter(resx); pm["resy"] = Parameter(resy); pm["focal"] = Parameter(focal);
		Camera *cam = PerspectiveCamera::factory(pm, fake_env());
		pushp(cam_in, from); pushp(cam_in, to); pushp(cam_in, up);
		cam_in.push_back((uint32_t)resx); cam_in.push_back((uint32_t)resy); cam_in.push_back(f2u(focal));
		for(int k = 0; k < 24; ++k)
		{
			float px = urand() * resx, py = urand() * resy;
			if(k == 0) { px = 0.5f; py = 0.5f; }
			float wt;
			Ray r = cam->shootRay(px, py, 0.5f, 0.5f, wt);
			pxy.push_back(f2u(px)); pxy.push_back(f2u(py));
			pushp(out, r.from_); pushv(out, r.dir_); out.push_back(f2u(r.tmin_)); out.push_back(f2u(r.tmax_)); out.push_back(f2u(wt));
		}
	}
	j.arr_u32("cam_cfg12", cam_in); j.arr_u32("cam_pxy", pxy); j.arr_u32("cam_ray9", out);
}

// depth of field: lens sampling for every bokeh shape and bias (camera_perspective.cc:75-156)
static void sec_camera_dof(Json &j)
{
	std::vector<uint32_t> cfg, in4, out;
	const char *types[7] = {"disk1", "disk2", "triangle", "square", "pentagon", "hexagon", "ring"};
	const char *biases[3] = {"uniform", "center", "edge"};
	for(int t = 0; t < 7; ++t)
		for(int b = 0; b < 3; ++b)
		{
			ParamMap pm;
			Point3 from(0.3f, -4.f, 1.f), to(0.f, 0.1f, 0.2f), up(0.3f, -4.f, 2.f);
			int resx = 320, resy = 200;
			float focal = 1.1f, apt = 0.05f + 0.01f * t, dofd = 3.5f + 0.25f * b, rot = 10.f * t + 3.f * b;
			pm["from"] = Parameter(from); pm["to"] = Parameter(to); pm["up"] = Parameter(up);
			pm["resx"] = Parameter(resx); pm["resy"] = Parameter(resy); pm["focal"] = Parameter(focal);
			pm["aperture"] = Parameter(apt); pm["dof_distance"] = Parameter(dofd);
			pm["bokeh_type"] = Parameter(std::string(types[t])); pm["bokeh_bias"] = Parameter(std::string(biases[b]));
			pm["bokeh_rotation"] = Parameter(rot);
			Camera *cam = PerspectiveCamera::factory(pm, fake_env());
			pushp(cfg, from); pushp(cfg, to); pushp(cfg, up);
			cfg.push_back((uint32_t)resx); cfg.push_back((uint32_t)resy); cfg.push_back(f2u(focal));
			cfg.push_back(f2u(apt)); cfg.push_back(f2u(dofd)); cfg.push_back((uint32_t)t); cfg.push_back((uint32_t)b); cfg.push_back(f2u(rot));
			for(int k = 0; k < 40; ++k)
			{
				float px = urand() * resx, py = urand() * resy, lu = urand(), lv = urand();
				if(k == 0) { lu = 0.5f; lv = 0.5f; }
				if(k == 1) { lu = 0.f; lv = 0.f; }
				if(k == 2) { lu = 0.999999f; lv = 0.25f; }
				float wt;
				Ray r = cam->shootRay(px, py, lu, lv, wt);
				in4.push_back(f2u(px)); in4.push_back(f2u(py)); in4.push_back(f2u(lu)); in4.push_back(f2u(lv));
				pushp(out, r.from_); pushv(out, r.dir_); out.push_back(f2u(r.tmin_)); out.push_back(f2u(r.tmax_)); out.push_back(f2u(wt));
			}
		}
	j.arr_u32("camd_cfg17", cfg); j.arr_u32("camd_in4", in4); j.arr_u32("camd_ray9", out);
}

static void sec_lights(Json &j)
{
	// one area light, many surface points
	ParamMap pm;
	Point3 corner(-0.25f, -0.25f, 0.99f), p1(0.25f, -0.25f, 0.99f), p2(-0.25f, 0.25f, 0.99f);
	pm["corner"] = Parameter(corner); pm["point1"] = Parameter(p1); pm["point2"] = Parameter(p2);
	pm["color"] = Parameter(Rgba(1.f, 0.9f, 0.8f, 1.f)); pm["power"] = Parameter(17.5f); pm["samples"] = Parameter(1);
	Light *al = AreaLight::factory(pm, fake_env());
	std::vector<uint32_t> cfg, in, out, iin, iout;
	std::vector<int> ok, iok;
	pushp(cfg, corner); pushp(cfg, p1); pushp(cfg, p2);
	cfg.push_back(f2u(1.f)); cfg.push_back(f2u(0.9f)); cfg.push_back(f2u(0.8f)); cfg.push_back(f2u(17.5f));
	for(int i = 0; i < 200; ++i)
	{
		SurfacePoint sp;
		Point3 p(srand11(), srand11(), (i % 7 == 0) ? 1.5f : srand11() * 0.98f);
		make_sp(sp, Vec3(0, 0, 1), p, false);
		LSample ls; ls.s_1_ = urand(); ls.s_2_ = urand(); ls.sp_ = nullptr;
		Ray wi;
		bool r = al->illumSample(sp, ls, wi);
		pushp(in, p); in.push_back(f2u(ls.s_1_)); in.push_back(f2u(ls.s_2_));
		ok.push_back(r ? 1 : 0);
		if(!r) { wi.dir_ = Vec3(0.f); wi.tmax_ = 0.f; ls.pdf_ = 0.f; ls.col_ = Rgb(0.f); }
		pushv(out, wi.dir_); out.push_back(f2u(wi.tmax_)); out.push_back(f2u(ls.pdf_)); pushc(out, ls.col_);
	}
	for(int i = 0; i < 200; ++i)
	{
		Point3 from(srand11(), srand11(), srand11() * 0.9f);
		Vec3 dir = (Point3(srand11() * 0.4f, srand11() * 0.4f, 0.99f) - from);
		dir.normalize();
		if(i % 9 == 0) dir = rand_unit();
		Ray ray(from, dir);
		float t = -1.f, ipdf = 0.f; Rgb col(0.f);
		bool r = al->intersect(ray, t, col, ipdf);
		pushp(iin, from); pushv(iin, dir);
		iok.push_back(r ? 1 : 0);
		if(!r) { t = 0.f; ipdf = 0.f; col = Rgb(0.f); }
		iout.push_back(f2u(t)); iout.push_back(f2u(ipdf)); pushc(iout, col);
	}
	j.arr_u32("al_cfg13", cfg); j.arr_u32("al_is_in5", in); j.arr_i32("al_is_ok", ok); j.arr_u32("al_is_out8", out);
	j.arr_u32("al_ix_in6", iin); j.arr_i32("al_ix_ok", iok); j.arr_u32("al_ix_out5", iout);

	// point light
	ParamMap pp;
	Point3 pos(0.3f, -0.2f, 2.f);
	pp["from"] = Parameter(pos); pp["color"] = Parameter(Rgba(1.f, 0.5f, 0.25f, 1.f)); pp["power"] = Parameter(12.f);
	Light *pl = PointLight::factory(pp, fake_env());
	std::vector<uint32_t> pin, pout;
	for(int i = 0; i < 64; ++i)
	{
		SurfacePoint sp;
		Point3 p(srand11() * 2.f, srand11() * 2.f, srand11());
		make_sp(sp, Vec3(0, 0, 1), p, false);
		Ray wi; Rgb col(0.f);
		pl->illuminate(sp, col, wi);
		pushp(pin, p);
		pushv(pout, wi.dir_); pout.push_back(f2u(wi.tmax_)); pushc(pout, col);
	}
	std::vector<uint32_t> pcfg; pushp(pcfg, pos); pcfg.push_back(f2u(1.f)); pcfg.push_back(f2u(0.5f)); pcfg.push_back(f2u(0.25f)); pcfg.push_back(f2u(12.f));
	j.arr_u32("pl_cfg7", pcfg); j.arr_u32("pl_in3", pin); j.arr_u32("pl_out7", pout);
}

struct MatCase { const char *name; ParamMap pm; };

static void run_material(Json &j, const char *prefix, Material *mat, int n_cases, bool with_flags_variants, int raylevel = 1)
{
	alignas(16) static unsigned char userdata[4096];
	RenderState state(nullptr);
	state.userdata_ = (void *)userdata;
	state.include_lights_ = true;
	state.raylevel_ = raylevel;      // recursiveRaytrace increments it before it asks for the specular directions
	std::vector<uint32_t> in, ev, sm, pd, spec, alph, transp;
	std::vector<int> flags_out, sflags_in, sflags_out, spec_flags;
	for(int i = 0; i < n_cases; ++i)
	{
		SurfacePoint sp;
		Vec3 n = rand_unit();
		make_sp(sp, n, Point3(srand11(), srand11(), srand11()), (i % 4 == 3));
		Vec3 wo = rand_unit();
		if(i % 8 != 7 && (wo * sp.ng_) < 0.f) wo = -wo; // mostly front-facing
		Vec3 wl = rand_unit();
		if(i % 6 != 5 && (wl * sp.ng_) < 0.f) wl = -wl;
		float s_1 = urand(), s_2 = urand();
		Bsdf_t bsdfs;
		sp.material_ = mat;
		mat->initBsdf(state, sp, bsdfs);
		flags_out.push_back((int)bsdfs);
		pushv(in, sp.n_); pushv(in, sp.ng_); pushv(in, wo); pushv(in, wl); in.push_back(f2u(s_1)); in.push_back(f2u(s_2));
		Rgb e = mat->eval(state, sp, wo, wl, BsdfAll);
		pushc(ev, e);
		float p = mat->pdf(state, sp, wo, wl, BsdfGlossy | BsdfDiffuse | BsdfDispersive | BsdfReflect | BsdfTransmit);
		pd.push_back(f2u(p));
		Bsdf_t sf = BsdfAll;
		if(with_flags_variants)
		{
			if(i % 3 == 1) sf = BsdfDiffuse | BsdfReflect | BsdfTransmit;
			if(i % 3 == 2) sf = BsdfGlossy | BsdfDiffuse | BsdfDispersive | BsdfReflect | BsdfTransmit;
		}
		Sample s(s_1, s_2, sf);
		Vec3 wi(0.f);
		float w = 0.f;
		Rgb sc = mat->sample(state, sp, wo, wi, s, w);
		sflags_in.push_back((int)sf);
		sflags_out.push_back((int)s.sampled_flags_);
		pushc(sm, sc); pushv(sm, wi); sm.push_back(f2u(s.pdf_)); sm.push_back(f2u(w));
		// perfect specular directions and weights for recursiveRaytrace, and the alpha of transparent materials
		bool refl = false, refr = false;
		Vec3 sdir[2] = {Vec3(0.f), Vec3(0.f)};
		Rgb scol[2] = {Rgb(0.f), Rgb(0.f)};
		mat->initBsdf(state, sp, bsdfs);
		mat->getSpecular(state, sp, wo, refl, refr, sdir, scol);
		spec_flags.push_back((refl ? 1 : 0) | (refr ? 2 : 0));
		if(!refl) { sdir[0] = Vec3(0.f); scol[0] = Rgb(0.f); }
		if(!refr) { sdir[1] = Vec3(0.f); scol[1] = Rgb(0.f); }
		pushv(spec, sdir[0]); pushc(spec, scol[0]); pushv(spec, sdir[1]); pushc(spec, scol[1]);
		alph.push_back(f2u(mat->getAlpha(state, sp, wo)));
		pushc(transp, mat->getTransparency(state, sp, wo));      // what a transparent-shadow ray along wo picks up (intersectTs)
	}
	std::string p(prefix);
	j.arr_u32((p + "_in14").c_str(), in); j.arr_i32((p + "_flags").c_str(), flags_out);
	j.arr_u32((p + "_eval3").c_str(), ev); j.arr_u32((p + "_pdf").c_str(), pd);
	j.arr_i32((p + "_sflags_in").c_str(), sflags_in); j.arr_i32((p + "_sflags_out").c_str(), sflags_out);
	j.arr_u32((p + "_sample8").c_str(), sm);
	j.arr_i32((p + "_specflags").c_str(), spec_flags); j.arr_u32((p + "_spec12").c_str(), spec); j.arr_u32((p + "_alpha").c_str(), alph);
	j.arr_u32((p + "_transp3").c_str(), transp);
}

static void sec_materials(Json &j)
{
	std::list<ParamMap> no_nodes;
	{	// sd0: plain lambert, the workhorse of every diffuse config
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.7f, 0.6f, 0.5f, 1.f)); pm["diffuse_reflect"] = Parameter(0.9f);
		Material *m = ShinyDiffuseMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "sd0", m, 160, true);
	}
	{	// sd1: all four components + fresnel
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.8f, 0.3f, 0.2f, 1.f)); pm["mirror_color"] = Parameter(Rgba(0.9f, 0.95f, 1.f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.8f); pm["specular_reflect"] = Parameter(0.3f);
		pm["transparency"] = Parameter(0.2f); pm["translucency"] = Parameter(0.25f);
		pm["fresnel_effect"] = Parameter(true); pm["IOR"] = Parameter(1.45f); pm["transmit_filter"] = Parameter(0.7f);
		pm["emit"] = Parameter(0.1f);
		Material *m = ShinyDiffuseMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "sd1", m, 240, true);
	}
	{	// sd3: mirror without fresnel over a diffuse base
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.6f, 0.6f, 0.7f, 1.f)); pm["mirror_color"] = Parameter(Rgba(0.9f, 0.8f, 0.7f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.7f); pm["specular_reflect"] = Parameter(0.45f);
		Material *m = ShinyDiffuseMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "sd3", m, 160, true);
	}
	{	// sd4: transparent + mirror with fresnel, no translucency
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.5f, 0.8f, 0.6f, 1.f)); pm["mirror_color"] = Parameter(Rgba(1.f, 1.f, 1.f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.6f); pm["specular_reflect"] = Parameter(0.5f); pm["transparency"] = Parameter(0.6f);
		pm["fresnel_effect"] = Parameter(true); pm["IOR"] = Parameter(1.33f); pm["transmit_filter"] = Parameter(0.4f);
		Material *m = ShinyDiffuseMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "sd4", m, 160, true);
	}
	{	// gg0: glass, filtered transmission, tinted reflection (entering and leaving rays; raylevel 1 and 4: the
		// reflection of a ray leaving the glass is only followed below level 3, material_glass.cc:327)
		ParamMap pm;
		pm["IOR"] = Parameter(1.52); pm["filter_color"] = Parameter(Rgba(0.6f, 0.9f, 0.7f, 1.f)); pm["transmit_filter"] = Parameter(0.8);
		pm["mirror_color"] = Parameter(Rgba(0.95f, 0.9f, 1.f, 1.f));
		Material *m = GlassMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gg0", m, 200, true, 1);
		run_material(j, "gg0d", m, 120, true, 4);
	}
	{	// gg1: glass with fake shadows (filter lobe instead of specular transmission), higher index
		ParamMap pm;
		pm["IOR"] = Parameter(2.1); pm["filter_color"] = Parameter(Rgba(1.f, 0.5f, 0.5f, 1.f)); pm["transmit_filter"] = Parameter(0.3);
		pm["fake_shadows"] = Parameter(true);
		Material *m = GlassMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gg1", m, 160, true, 2);
	}
	{	// cg0: coated glossy with a diffuse substrate
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.8f, 0.7f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.3f, 0.5f, 0.7f, 1.f));
		pm["mirror_color"] = Parameter(Rgba(1.f, 0.95f, 0.9f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.5f); pm["glossy_reflect"] = Parameter(0.6f); pm["exponent"] = Parameter(80.f);
		pm["specular_reflect"] = Parameter(0.8f); pm["IOR"] = Parameter(1.6); pm["as_diffuse"] = Parameter(true);
		Material *m = CoatedGlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "cg0", m, 240, true, 1);
	}
	{	// cg1: coated glossy without substrate, Oren-Nayar irrelevant, deep recursion level (getSpecular stops above 5)
		ParamMap pm;
		pm["color"] = Parameter(Rgba(1.f, 1.f, 1.f, 1.f)); pm["glossy_reflect"] = Parameter(0.9f); pm["exponent"] = Parameter(300.f);
		pm["specular_reflect"] = Parameter(1.f); pm["IOR"] = Parameter(1.0);
		Material *m = CoatedGlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "cg1", m, 120, true, 6);
	}
	{	// cg2: diffuse substrate with Oren-Nayar
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.8f, 0.8f, 0.8f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.7f, 0.3f, 0.2f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.8f); pm["glossy_reflect"] = Parameter(0.3f); pm["exponent"] = Parameter(25.f);
		pm["specular_reflect"] = Parameter(0.5f); pm["IOR"] = Parameter(1.8);
		pm["diffuse_brdf"] = Parameter(std::string("Oren-Nayar")); pm["sigma"] = Parameter(0.3);
		Material *m = CoatedGlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "cg2", m, 160, true, 2);
	}
	{	// mi0: mirror
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.8f, 0.6f, 1.f)); pm["reflect"] = Parameter(0.85f);
		Material *m = MirrorMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "mi0", m, 120, true, 1);
	}
	{	// sd2: oren-nayar
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.5f, 0.7f, 0.4f, 1.f)); pm["diffuse_reflect"] = Parameter(1.0f);
		pm["diffuse_brdf"] = Parameter(std::string("oren_nayar")); pm["sigma"] = Parameter(0.35);
		Material *m = ShinyDiffuseMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "sd2", m, 160, true);
	}
	{	// gl0: blinn glossy, as_diffuse (factory default), with diffuse substrate — the C4 material
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.85f, 0.8f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.4f, 0.5f, 0.6f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.4f); pm["glossy_reflect"] = Parameter(0.6f); pm["exponent"] = Parameter(50.f);
		pm["as_diffuse"] = Parameter(true);
		Material *m = GlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gl0", m, 240, true);
	}
	{	// gl1: glossy only (no diffuse), higher exponent
		ParamMap pm;
		pm["color"] = Parameter(Rgba(1.f, 1.f, 1.f, 1.f)); pm["glossy_reflect"] = Parameter(0.8f); pm["exponent"] = Parameter(500.f);
		pm["as_diffuse"] = Parameter(true);
		Material *m = GlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gl1", m, 160, true);
	}
	{	// gl2: glossy + oren-nayar substrate
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.9f, 0.9f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.6f, 0.2f, 0.2f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.7f); pm["glossy_reflect"] = Parameter(0.3f); pm["exponent"] = Parameter(20.f);
		pm["diffuse_brdf"] = Parameter(std::string("Oren-Nayar")); pm["sigma"] = Parameter(0.25);
		Material *m = GlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gl2", m, 160, true);
	}
	{	// gl3: the anisotropic Ashikhmin-Shirley lobe (material_utils_microfacet.h:38-87) over a diffuse substrate
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.8f, 0.85f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.5f, 0.4f, 0.6f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.5f); pm["glossy_reflect"] = Parameter(0.5f);
		pm["anisotropic"] = Parameter(true); pm["exp_u"] = Parameter(400.f); pm["exp_v"] = Parameter(12.f);
		Material *m = GlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gl3", m, 240, true);
	}
	{	// gl4: anisotropic lobe alone, exponents the other way round
		ParamMap pm;
		pm["color"] = Parameter(Rgba(1.f, 1.f, 1.f, 1.f)); pm["glossy_reflect"] = Parameter(0.9f);
		pm["anisotropic"] = Parameter(true); pm["exp_u"] = Parameter(8.f); pm["exp_v"] = Parameter(900.f);
		Material *m = GlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "gl4", m, 160, true);
	}
	{	// cg3: coated glossy with the anisotropic lobe
		ParamMap pm;
		pm["color"] = Parameter(Rgba(0.9f, 0.9f, 0.8f, 1.f)); pm["diffuse_color"] = Parameter(Rgba(0.2f, 0.6f, 0.5f, 1.f));
		pm["diffuse_reflect"] = Parameter(0.6f); pm["glossy_reflect"] = Parameter(0.5f);
		pm["specular_reflect"] = Parameter(0.7f); pm["IOR"] = Parameter(1.5); pm["as_diffuse"] = Parameter(true);
		pm["anisotropic"] = Parameter(true); pm["exp_u"] = Parameter(30.f); pm["exp_v"] = Parameter(250.f);
		Material *m = CoatedGlossyMaterial::factory(pm, no_nodes, fake_env());
		run_material(j, "cg3", m, 200, true, 1);
	}
	{	// light material emit
		ParamMap pm;
		pm["color"] = Parameter(Rgba(1.f, 0.9f, 0.8f, 1.f)); pm["power"] = Parameter(17.5);
		Material *m = LightMaterial::factory(pm, no_nodes, fake_env());
		RenderState state(nullptr);
		std::vector<uint32_t> in, out;
		for(int i = 0; i < 32; ++i)
		{
			SurfacePoint sp; make_sp(sp, rand_unit(), Point3(0.f, 0.f, 0.f), false);
			Vec3 wo = rand_unit();
			state.include_lights_ = (i % 4 != 3);
			Rgb e = m->emit(state, sp, wo);
			pushv(in, sp.n_); pushv(in, wo); in.push_back(state.include_lights_ ? 1u : 0u);
			pushc(out, e);
		}
		j.arr_u32("lm_in7", in); j.arr_u32("lm_emit3", out);
	}
}

// BeerVolumeHandler(absorption colour, distance) + transmittance(ray.tmax_): what a glass material with "absorption"
// multiplies onto the light that travelled through it (volumehandler_beer.cc:28-48, material_glass.cc:371-398)
static void sec_beer(Json &j)
{
	std::vector<uint32_t> in, out;
	RenderState state(nullptr);
	const float cols[6][3] = {{0.5f, 0.7f, 0.9f}, {0.95f, 0.2f, 0.01f}, {1.f, 1.f, 0.3f}, {0.f, 0.5f, 1e-39f}, {0.33f, 0.66f, 0.99f}, {0.8f, 0.8f, 0.8f}};
	const double dists[6] = {1.0, 0.35, 3.0, 1.0, 0.0, 12.5};
	for(int c = 0; c < 6; ++c)
	{
		BeerVolumeHandler beer(Rgb(cols[c][0], cols[c][1], cols[c][2]), dists[c]);
		const VolumeHandler *vol = &beer;
		for(int i = 0; i < 24; ++i)
		{
			Ray ray(Point3(0.f, 0.f, 0.f), Vec3(0.f, 0.f, 1.f), 0.f, -1.f);
			if(i == 0) ray.tmax_ = -1.f;
			else if(i == 1) ray.tmax_ = 2e30f;
			else if(i == 2) ray.tmax_ = 0.f;
			else ray.tmax_ = urand() * (i % 3 == 0 ? 40.f : 2.f);
			Rgb col(1.f);
			const bool ok = vol->transmittance(state, ray, col);
			in.push_back(f2u(cols[c][0])); in.push_back(f2u(cols[c][1])); in.push_back(f2u(cols[c][2])); in.push_back(f2u((float)dists[c])); in.push_back(f2u(ray.tmax_));
			out.push_back(ok ? 1u : 0u); pushc(out, col);
		}
	}
	j.arr_u32("beer_in5", in); j.arr_u32("beer_out4", out);
}

// RoughGlassMaterial (material_rough_glass.cc): the GGX lobe that both reflects and transmits — one-direction sample() for path
// segments, the two-direction sample() recursiveRaytrace's glossy branch calls (integrator_montecarlo.cc:919-970), getTransparency /
// getAlpha / isTransparent for fake shadows.  Appended after every other section so that their random inputs stay what they were.
static void run_rough_glass(Json &j, const char *prefix, Material *mat, int n_cases)
{
	run_material(j, prefix, mat, n_cases, true, 1);
	alignas(16) static unsigned char userdata[4096];
	RenderState state(nullptr);
	state.userdata_ = (void *)userdata;
	state.raylevel_ = 1;
	std::vector<uint32_t> in, out;
	std::vector<int> sflags_in, sflags_out;
	for(int i = 0; i < n_cases; ++i)
	{
		SurfacePoint sp;
		Vec3 n = rand_unit();
		make_sp(sp, n, Point3(srand11(), srand11(), srand11()), (i % 4 == 3));
		Vec3 wo = rand_unit();
		if(i % 3 != 2 && (wo * sp.ng_) < 0.f) wo = -wo;      // a third of the rays leave the glass
		float s_1 = urand(), s_2 = urand();
		Bsdf_t bsdfs;
		sp.material_ = mat;
		mat->initBsdf(state, sp, bsdfs);
		Bsdf_t sf = BsdfGlossy | BsdfAllGlossy;                // what the glossy branch asks for (:921)
		if(i % 5 == 3) sf = BsdfGlossy | BsdfReflect;
		if(i % 5 == 4) sf = BsdfGlossy | BsdfTransmit;
		Sample s(s_1, s_2, sf);
		Vec3 dir[2] = {Vec3(0.f), Vec3(0.f)};
		Rgb tcol(0.f);
		float w[2] = {0.f, 0.f};
		Rgb ret = mat->sample(state, sp, wo, dir, tcol, s, w);
		pushv(in, sp.n_); pushv(in, sp.ng_); pushv(in, wo); pushv(in, Vec3(0.f)); in.push_back(f2u(s_1)); in.push_back(f2u(s_2));
		sflags_in.push_back((int)sf); sflags_out.push_back((int)s.sampled_flags_);
		pushv(out, dir[0]); pushc(out, ret); out.push_back(f2u(w[0]));
		pushv(out, dir[1]); pushc(out, tcol); out.push_back(f2u(w[1]));
		out.push_back(f2u(s.pdf_));
	}
	std::string p(prefix);
	j.arr_u32((p + "_two_in14").c_str(), in); j.arr_i32((p + "_two_sflags_in").c_str(), sflags_in); j.arr_i32((p + "_two_sflags_out").c_str(), sflags_out);
	j.arr_u32((p + "_two_out15").c_str(), out);
}
static void sec_rough_glass(Json &j)
{
	std::list<ParamMap> no_nodes;
	{	// rg0: moderately rough, tinted transmission and reflection
		ParamMap pm;
		pm["IOR"] = Parameter(1.5); pm["filter_color"] = Parameter(Rgba(0.7f, 0.9f, 0.8f, 1.f)); pm["transmit_filter"] = Parameter(0.7);
		pm["mirror_color"] = Parameter(Rgba(0.95f, 0.9f, 1.f, 1.f)); pm["alpha"] = Parameter(0.3);
		run_rough_glass(j, "rg0", RoughGlassMaterial::factory(pm, no_nodes, fake_env()), 240);
	}
	{	// rg1: very rough, high index, fake shadows (a filter lobe for transparent shadows)
		ParamMap pm;
		pm["IOR"] = Parameter(2.0); pm["filter_color"] = Parameter(Rgba(1.f, 0.6f, 0.5f, 1.f)); pm["transmit_filter"] = Parameter(0.4);
		pm["alpha"] = Parameter(0.9); pm["fake_shadows"] = Parameter(true);
		run_rough_glass(j, "rg1", RoughGlassMaterial::factory(pm, no_nodes, fake_env()), 160);
	}
	{	// rg2: nearly smooth (alpha clamps at 1e-4 * ... : factory takes max(1e-4, min(alpha / 2, 1)))
		ParamMap pm;
		pm["IOR"] = Parameter(1.33); pm["alpha"] = Parameter(0.0001);
		run_rough_glass(j, "rg2", RoughGlassMaterial::factory(pm, no_nodes, fake_env()), 120);
	}
}

int main()
{
	Json j;
	j.s = "{\n";
	sec_fastmath(j);
	sec_qmc(j);
	sec_geom(j);
	sec_camera(j);
	sec_camera_dof(j);
	sec_lights(j);
	sec_materials(j);
	sec_beer(j);
	sec_rough_glass(j);
	j.s += "\n}\n";
	fputs(j.s.c_str(), stdout);
	return 0;
}
