// TEST INFRASTRUCTURE — not product code.
//
// Reference harness for SURVEY row N2: image textures and shader nodes.  Our driver, compiled (oracle/Makefile, target
// `ref`) against the reference's own headers and sources where they lie: texture/texture_image.cc,
// common/imagehandler.cc, imagehandler/imagehandler_tga.cc, imagehandler/imagehandler_hdr.cc,
// shader/shader_node*.cc.  Nothing of the reference is copied.
//
//   * image lookups: an ImageTexture over an in-memory ImageHandler (a subclass of the reference's abstract
//     ImageHandler written here, like any image plug-in) — clip modes, repeat / mirror / crop / rot90,
//     none / bilinear interpolation, the "optimized" 10-bit buffers, colour adjustments, getFloat;
//   * node graphs: TextureMapperNode (every texco / mapping), ValueNode, MixNode (every mode), LayerNode (every blend
//     mode and flag) evaluated on random surface points.  The nodes' private constructors / members are reached by
//     compiling this one translation unit with `private` and `protected` spelled `public` — a test's way in, which
//     changes no layout and no code of the reference;
//   * file decoders: the reference's own TGA and HDR handlers reading the texture files its tests hold
//     (tests/test01/test01_tex.tga, .hdr) — what the host-side decoders of this repository must reproduce.
//
// Output: one JSON document on stdout, floats as IEEE-754 bit patterns.  tests/golden/make_golden.py stores it as
// tests/golden/ref_textures_{ieee,fast}.json.gz.
// the standard library first: its headers must not see the two macros below
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#define private public
#define protected public
#include "common/param.h"
#include "texture/texture_image.h"
#include "imagehandler/imagehandler.h"
#include "imagehandler/imagehandler_tga.h"
#include "imagehandler/imagehandler_hdr.h"
#include "shader/shader_node.h"
#include "shader/shader_node_basic.h"
#include "shader/shader_node_layer.h"
#include "material/material_shiny_diffuse.h"
#undef private
#undef protected

#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>

#include "common/surface.h"
#include "common/scene.h"
#include "camera/camera_perspective.h"

using namespace yafaray4;

static uint32_t lcg_state = 777u;
static uint32_t lcg() { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state; }
static float urand() { return (float)((lcg() >> 8) * (1.0 / 16777216.0)); }
static float srand11() { return 2.f * urand() - 1.f; }
static uint32_t f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }

struct Json
{
	std::string s; bool first = true;
	void key(const char *k) { if(!first) s += ",\n"; first = false; s += "\""; s += k; s += "\": "; }
	void arr_u32(const std::string &k, const std::vector<uint32_t> &v)
	{
		key(k.c_str()); s += "[";
		char b[32];
		for(size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof b, "%s%u", i ? "," : "", v[i]); s += b; }
		s += "]";
	}
};

alignas(64) static unsigned char fake_env_storage[1 << 16];
static RenderEnvironment &fake_env() { return *reinterpret_cast<RenderEnvironment *>(fake_env_storage); }

// an image handler that holds pixels given to it (nothing read from or written to a file)
class MemHandler final : public ImageHandler
{
	public:
		MemHandler(int w, int h, int channels, TextureOptimization opt)
		{
			width_ = w; height_ = h; has_alpha_ = channels == 4; texture_optimization_ = opt;
			img_buffer_.push_back(new ImageBuffer(w, h, channels, opt));
		}
		bool loadFromFile(const std::string &) override { return false; }
		bool saveToFile(const std::string &, int) override { return false; }
};

static void push_rgba(std::vector<uint32_t> &o, const Rgba &c) { o.push_back(f2u(c.r_)); o.push_back(f2u(c.g_)); o.push_back(f2u(c.b_)); o.push_back(f2u(c.a_)); }

// ---- image lookups --------------------------------------------------------------------------------------------
struct TexCase { const char *name; int w, h, ch; TextureOptimization opt; InterpolationType it; ImageTexture::TexClipMode clip; int xrep, yrep; bool rot90, mx, my; float crop[4]; bool even, odd; float cdist; float adj[7]; bool clampc; ColorSpace cs; float gamma; };

static void sec_image(Json &j)
{
	const TexCase cases[] = {
		{"img_repeat_bilinear", 7, 5, 4, TextureOptimization::Optimized, InterpolationType::Bilinear, ImageTexture::TexClipMode::Repeat, 1, 1, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_repeat_none", 7, 5, 3, TextureOptimization::Optimized, InterpolationType::None, ImageTexture::TexClipMode::Repeat, 3, 2, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_repeat_mirror", 6, 6, 4, TextureOptimization::None, InterpolationType::Bilinear, ImageTexture::TexClipMode::Repeat, 2, 3, false, true, true, {0, 0, 1, 1}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, LinearRgb, 1.f},
		{"img_extend_crop_rot", 8, 4, 4, TextureOptimization::Optimized, InterpolationType::Bilinear, ImageTexture::TexClipMode::Extend, 1, 1, true, false, false, {0.1f, 0.2f, 0.9f, 0.7f}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_clip", 5, 5, 3, TextureOptimization::None, InterpolationType::Bilinear, ImageTexture::TexClipMode::Clip, 1, 1, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_clipcube", 5, 5, 4, TextureOptimization::Optimized, InterpolationType::None, ImageTexture::TexClipMode::ClipCube, 1, 1, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_checker", 4, 4, 4, TextureOptimization::Optimized, InterpolationType::Bilinear, ImageTexture::TexClipMode::Checker, 1, 1, false, false, false, {0, 0, 1, 1}, true, false, 0.3f, {1, 1, 1, 0, 1, 1, 1}, false, Srgb, 1.f},
		{"img_adjust", 7, 5, 4, TextureOptimization::Optimized, InterpolationType::Bilinear, ImageTexture::TexClipMode::Repeat, 1, 1, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1.2f, 0.8f, 1.f, 0.f, 0.9f, 1.1f, 0.7f}, true, Srgb, 1.f},
		{"img_adjust_hsv", 7, 5, 3, TextureOptimization::None, InterpolationType::Bilinear, ImageTexture::TexClipMode::Repeat, 1, 1, false, false, false, {0, 0, 1, 1}, false, true, 0.f, {1.f, 1.f, 1.4f, 40.f, 1.f, 1.f, 1.f}, false, RawManualGamma, 2.2f},
	};
	for(const TexCase &c : cases)
	{
		MemHandler *ih = new MemHandler(c.w, c.h, c.ch, c.opt);
		std::vector<uint32_t> px;
		for(int y = 0; y < c.h; ++y)
			for(int x = 0; x < c.w; ++x)
			{
				Rgba col(urand(), urand(), urand(), urand());
				ih->putPixel(x, y, col);
				push_rgba(px, ih->getPixel(x, y));        // what the buffer gives back (10-bit storage for "optimized")
			}
		ImageTexture *tex = new ImageTexture(ih, c.it, c.gamma, c.cs);
		tex->xrepeat_ = c.xrep; tex->yrepeat_ = c.yrep; tex->rot_90_ = c.rot90;
		tex->setCrop(c.crop[0], c.crop[1], c.crop[2], c.crop[3]);
		tex->use_alpha_ = true; tex->calc_alpha_ = false; tex->normalmap_ = false;
		tex->tex_clip_mode_ = c.clip; tex->checker_even_ = c.even; tex->checker_odd_ = c.odd; tex->checker_dist_ = c.cdist;
		tex->mirror_x_ = c.mx; tex->mirror_y_ = c.my;
		tex->setAdjustments(c.adj[0], c.adj[1], c.adj[2], c.adj[3], c.clampc, c.adj[4], c.adj[5], c.adj[6]);
		std::vector<uint32_t> in, out;
		for(int k = 0; k < 160; ++k)
		{
			Point3 p(srand11() * 2.2f, srand11() * 2.2f, srand11() * 1.3f);
			if(k % 16 == 0) p.x_ = (k % 32 == 0) ? 1.f : -1.f;      // borders
			if(k % 24 == 0) p.y_ = 0.f;
			in.push_back(f2u(p.x_)); in.push_back(f2u(p.y_)); in.push_back(f2u(p.z_));
			push_rgba(out, tex->getColor(p));
			out.push_back(f2u(tex->getFloat(p)));
		}
		j.arr_u32(std::string(c.name) + "_texels", px);
		j.arr_u32(std::string(c.name) + "_in", in);
		j.arr_u32(std::string(c.name) + "_out", out);
	}
}

// ---- the reference's TGA / HDR decoders on the reference's own test textures -------------------------------------
static void dump_handler(Json &j, const char *name, ImageHandler *ih)
{
	const int w = ih->getWidth(), h = ih->getHeight();
	std::vector<uint32_t> meta = {(uint32_t)w, (uint32_t)h};
	std::vector<uint32_t> samples;
	double sum[4] = {0, 0, 0, 0};
	for(int y = 0; y < h; ++y)
		for(int x = 0; x < w; ++x)
		{
			const Rgba c = ih->getPixel(x, y);
			sum[0] += c.r_; sum[1] += c.g_; sum[2] += c.b_; sum[3] += c.a_;
			if((x * 7 + y * 13) % 97 == 0) { samples.push_back((uint32_t)x); samples.push_back((uint32_t)y); push_rgba(samples, c); }
		}
	for(int k = 0; k < 4; ++k) { union { double d; uint32_t u[2]; } v; v.d = sum[k]; meta.push_back(v.u[0]); meta.push_back(v.u[1]); }
	j.arr_u32(std::string(name) + "_meta", meta);
	j.arr_u32(std::string(name) + "_samples", samples);
}
static void sec_files(Json &j, const std::string &dir)
{
	{
		TgaHandler *ih = new TgaHandler();
		ih->setColorSpace(Srgb, 1.f); ih->setTextureOptimization(TextureOptimization::Optimized);
		if(ih->loadFromFile(dir + "/test01_tex.tga")) dump_handler(j, "file_tga_srgb_optimized", ih);
		TgaHandler *ih2 = new TgaHandler();
		ih2->setColorSpace(LinearRgb, 1.f); ih2->setTextureOptimization(TextureOptimization::None);
		if(ih2->loadFromFile(dir + "/test01_tex.tga")) dump_handler(j, "file_tga_linear_none", ih2);
	}
	{
		HdrHandler *ih = new HdrHandler();
		ih->setColorSpace(LinearRgb, 1.f); ih->setTextureOptimization(TextureOptimization::None);
		if(ih->loadFromFile(dir + "/test01_tex.hdr")) dump_handler(j, "file_hdr", ih);
	}
}

// ---- node graphs ---------------------------------------------------------------------------------------------------
static void make_sp(SurfacePoint &sp)
{
	Vec3 n(srand11(), srand11(), srand11()); n.normalize();
	Vec3 ng = n + 0.3f * Vec3(srand11(), srand11(), srand11()); ng.normalize();
	sp.n_ = n; sp.ng_ = ng;
	sp.p_ = Point3(srand11() * 1.5f, srand11() * 1.5f, srand11() * 1.5f);
	createCs__(sp.n_, sp.nu_, sp.nv_);
	sp.u_ = urand() * 1.4f - 0.2f; sp.v_ = urand() * 1.4f - 0.2f;
	sp.has_uv_ = true; sp.has_orco_ = true;
	sp.orco_p_ = Point3(srand11(), srand11(), srand11());
	Vec3 ong(srand11(), srand11(), srand11()); ong.normalize();
	sp.orco_ng_ = ong;
	sp.material_ = nullptr; sp.light_ = nullptr; sp.object_ = nullptr; sp.origin_ = nullptr; sp.ray_ = nullptr;
	sp.prim_num_ = 0;
	sp.dp_du_ = sp.nu_; sp.dp_dv_ = sp.nv_; sp.dp_du_abs_ = sp.nu_; sp.dp_dv_abs_ = sp.nv_;
	sp.ds_du_ = Vec3(1, 0, 0); sp.ds_dv_ = Vec3(0, 1, 0);
}
static void push_sp(std::vector<uint32_t> &o, const SurfacePoint &sp)
{
	const float v[18] = {sp.p_.x_, sp.p_.y_, sp.p_.z_, sp.n_.x_, sp.n_.y_, sp.n_.z_, sp.ng_.x_, sp.ng_.y_, sp.ng_.z_,
	                     sp.orco_p_.x_, sp.orco_p_.y_, sp.orco_p_.z_, sp.orco_ng_.x_, sp.orco_ng_.y_, sp.orco_ng_.z_, sp.u_, sp.v_, 0.f};
	for(float f : v) o.push_back(f2u(f));
}

static void sec_nodes(Json &j)
{
	// one texture for every mapper: 6 x 5 RGBA, optimized storage, repeat, bilinear
	MemHandler *ih = new MemHandler(6, 5, 4, TextureOptimization::Optimized);
	std::vector<uint32_t> px;
	for(int y = 0; y < 5; ++y) for(int x = 0; x < 6; ++x) { ih->putPixel(x, y, Rgba(urand(), urand(), urand(), urand())); push_rgba(px, ih->getPixel(x, y)); }
	ImageTexture *tex = new ImageTexture(ih, InterpolationType::Bilinear, 1.f, Srgb);
	tex->xrepeat_ = 1; tex->yrepeat_ = 1; tex->rot_90_ = false; tex->setCrop(0, 0, 1, 1);
	tex->use_alpha_ = true; tex->calc_alpha_ = false; tex->normalmap_ = false;
	tex->tex_clip_mode_ = ImageTexture::TexClipMode::Repeat; tex->checker_even_ = false; tex->checker_odd_ = true; tex->checker_dist_ = 0.f;
	tex->mirror_x_ = false; tex->mirror_y_ = false;
	tex->setAdjustments(1, 1, 1, 0, false, 1, 1, 1);
	j.arr_u32("nodes_texels", px);

	// camera for the `window` and `normal` texture coordinates
	ParamMap cp;
	cp["from"] = Point3(0.5f, -4.f, 1.f); cp["to"] = Point3(0.f, 0.f, 0.2f); cp["up"] = Point3(0.5f, -4.f, 2.f);
	cp["resx"] = 64; cp["resy"] = 48; cp["focal"] = 1.3f;
	Camera *cam = PerspectiveCamera::factory(cp, fake_env());
	RenderState state(nullptr);
	state.cam_ = cam;

	// mappers: texco x mapping x axis permutations, scale / offset; then the graph
	//   0..N-1 mappers | value | mix of (mapper k, value) by every mode | layer over (mapper k) for every blend mode and flag set
	const int texcos[] = {TextureMapperNode::Uv, TextureMapperNode::Glob, TextureMapperNode::Orco, TextureMapperNode::Tran, TextureMapperNode::Win, TextureMapperNode::Nor};
	std::vector<ShaderNode *> nodes;
	std::vector<TextureMapperNode *> mappers;
	std::vector<uint32_t> desc;       // per node: type and its parameters, as the test rebuilds them (see tests/test_oracle_golden.py)
	for(int tc : texcos)
		for(int proj = 0; proj < 4; ++proj)
		{
			TextureMapperNode *tm = new TextureMapperNode(tex);
			tm->coords_ = (TextureMapperNode::Coords)tc; tm->projection_ = (TextureMapperNode::Projection)proj;
			tm->map_x_ = 1 + (int)(lcg() % 3); tm->map_y_ = 1 + (int)(lcg() % 3); tm->map_z_ = (int)(lcg() % 4);
			const float sc[3] = {0.5f + urand() * 2.f, 0.5f + urand() * 2.f, 0.5f + urand()}, of[3] = {srand11() * 0.5f, srand11() * 0.5f, srand11() * 0.5f};
			tm->scale_ = Vec3(sc[0], sc[1], sc[2]);
			tm->offset_ = Vec3(2 * Point3(of[0], of[1], of[2]));          // TextureMapperNode::factory doubles it (:411)
			tm->do_scalar_ = (proj % 2) == 0; tm->bump_str_ = 1.f;
			float m[4][4];
			for(int a = 0; a < 4; ++a) for(int b = 0; b < 4; ++b) m[a][b] = (a == 3) ? (b == 3 ? 1.f : 0.f) : srand11();
			tm->mtx_ = Matrix4(m);
			tm->setup();
			tm->id_ = (unsigned)nodes.size();
			nodes.push_back(tm);
			mappers.push_back(tm);
			desc.push_back(0u); desc.push_back((uint32_t)tc); desc.push_back((uint32_t)proj);
			desc.push_back((uint32_t)tm->map_x_); desc.push_back((uint32_t)tm->map_y_); desc.push_back((uint32_t)tm->map_z_);
			for(float f : sc) desc.push_back(f2u(f));
			for(float f : of) desc.push_back(f2u(f));
			for(int a = 0; a < 4; ++a) for(int b = 0; b < 4; ++b) desc.push_back(f2u(m[a][b]));
			desc.push_back(tm->do_scalar_ ? 1u : 0u);
		}
	const int n_mappers = (int)nodes.size();
	{
		ParamMap vp; vp["color"] = Rgba(0.3f, 0.8f, 0.55f, 1.f); vp["alpha"] = 0.6f; vp["scalar"] = 0.35f;
		ShaderNode *v = ValueNode::factory(vp, fake_env());
		v->id_ = (unsigned)nodes.size(); nodes.push_back(v);
		desc.push_back(1u); desc.push_back(f2u(0.3f)); desc.push_back(f2u(0.8f)); desc.push_back(f2u(0.55f)); desc.push_back(f2u(0.6f)); desc.push_back(f2u(0.35f));
	}
	const int value_id = n_mappers;
	for(int mode = 0; mode <= 9; ++mode)
		for(int variant = 0; variant < 2; ++variant)
		{
			ParamMap mp; mp["mode"] = mode; mp["cfactor"] = 0.4f;
			MixNode *mx = (MixNode *)MixNode::factory(mp, fake_env());
			const int in1 = (int)(lcg() % (unsigned)n_mappers);
			mx->input_1_ = nodes[(size_t)in1];
			int in2 = -1, fac = -1;
			if(variant == 0) { mx->input_2_ = nodes[(size_t)value_id]; in2 = value_id; mx->cfactor_ = 0.3f + 0.1f * (float)mode; }
			else { mx->col_2_ = Rgba(0.7f, 0.2f, 0.4f, 0.9f); mx->val_2_ = 0.f; fac = (int)(lcg() % (unsigned)n_mappers); mx->factor_ = nodes[(size_t)fac]; }
			mx->val_1_ = 0.f; if(variant == 0) mx->val_2_ = 0.f;
			mx->id_ = (unsigned)nodes.size(); nodes.push_back(mx);
			desc.push_back(2u); desc.push_back((uint32_t)mode); desc.push_back(f2u(mx->cfactor_)); desc.push_back((uint32_t)in1); desc.push_back((uint32_t)in2); desc.push_back((uint32_t)fac);
			desc.push_back(f2u(0.7f)); desc.push_back(f2u(0.2f)); desc.push_back(f2u(0.4f)); desc.push_back(f2u(0.9f));
		}
	int prev_layer = -1;
	for(int mode = 0; mode <= 8; ++mode)
		for(int fl = 0; fl < 6; ++fl)
		{
			// flag sets: plain colour layer | scalar from colour | noRGB | negative | stencil (colour input) | scalar input + stencil + use_alpha
			const bool no_rgb = fl == 2, negative = fl == 3 || fl == 5, stencil = fl == 4 || fl == 5, use_alpha = fl == 5 || fl == 1;
			const bool do_color = fl != 1, do_scalar = fl == 1 || fl == 2 || fl == 5, color_input = fl != 5;
			ParamMap lp;
			lp["mode"] = mode; lp["def_col"] = Rgb(0.9f, 0.4f, 0.2f); lp["colfac"] = Parameter(0.8); lp["def_val"] = Parameter(0.7); lp["valfac"] = Parameter(0.9);
			lp["do_color"] = do_color; lp["do_scalar"] = do_scalar; lp["color_input"] = color_input; lp["use_alpha"] = use_alpha;
			lp["noRGB"] = no_rgb; lp["stencil"] = stencil; lp["negative"] = negative;
			LayerNode *ln = (LayerNode *)LayerNode::factory(lp, fake_env());
			const int in = (int)(lcg() % (unsigned)n_mappers);
			ln->input_ = nodes[(size_t)in];
			int upper = -1;
			if(fl % 2 == 1 && prev_layer >= 0) { upper = prev_layer; ln->upper_layer_ = nodes[(size_t)upper]; }
			else { ln->upper_layer_ = nullptr; ln->upper_col_ = Rgba(0.25f, 0.5f, 0.75f, 1.f); ln->upper_val_ = 0.45f; }
			ln->id_ = (unsigned)nodes.size(); nodes.push_back(ln);
			prev_layer = (int)ln->id_;
			desc.push_back(3u); desc.push_back((uint32_t)mode); desc.push_back((uint32_t)in); desc.push_back((uint32_t)upper);
			desc.push_back(no_rgb); desc.push_back(stencil); desc.push_back(negative); desc.push_back(use_alpha); desc.push_back(do_color); desc.push_back(do_scalar); desc.push_back(color_input);
		}
	j.arr_u32("nodes_desc", desc);
	std::vector<uint32_t> cam_in = {f2u(0.5f), f2u(-4.f), f2u(1.f), f2u(0.f), f2u(0.f), f2u(0.2f), f2u(0.5f), f2u(-4.f), f2u(2.f), 64u, 48u, f2u(1.3f)};
	j.arr_u32("nodes_camera", cam_in);
	std::vector<uint32_t> in, out;
	std::vector<NodeResult> stack_mem(nodes.size());
	for(int k = 0; k < 40; ++k)
	{
		SurfacePoint sp; make_sp(sp);
		push_sp(in, sp);
		NodeStack stack(stack_mem.data());
		for(ShaderNode *n : nodes) n->eval(stack, state, sp);
		for(size_t i = 0; i < nodes.size(); ++i) { push_rgba(out, stack_mem[i].col_); out.push_back(f2u(stack_mem[i].f_)); }
	}
	j.arr_u32("nodes_in", in);
	j.arr_u32("nodes_out", out);

	// bump mapping: evalDerivative of every node (TextureMapperNode :232-343 — the UV branch on a discrete texture, the branch for
	// every other coordinate kind —, LayerNode :122-152, the base class's zero for value / mix) and Material::applyBump with the last
	// layer's derivative (material.cc:77-84).  Surface points as above plus the shading-space UV derivatives getSurface leaves.
	std::vector<uint32_t> bin, bout, bsp, bnout;
	ParamMap mp; mp["type"] = std::string("shinydiffusemat");
	std::list<ParamMap> no_nodes;
	Material *any_mat = ShinyDiffuseMaterial::factory(mp, no_nodes, fake_env());
	for(int k = 0; k < 40; ++k)
	{
		SurfacePoint sp; make_sp(sp);
		sp.has_uv_ = (k % 4) != 3;
		Vec3 a(srand11(), srand11(), srand11()), b(srand11(), srand11(), srand11());
		a.normalize(); b.normalize();
		sp.ds_du_ = a; sp.ds_dv_ = b;
		push_sp(bin, sp);
		const float extra[13] = {sp.ds_du_.x_, sp.ds_du_.y_, sp.ds_du_.z_, sp.ds_dv_.x_, sp.ds_dv_.y_, sp.ds_dv_.z_, sp.nu_.x_, sp.nu_.y_, sp.nu_.z_, sp.nv_.x_, sp.nv_.y_, sp.nv_.z_,
		                         sp.has_uv_ ? 1.f : 0.f};
		for(float f : extra) bin.push_back(f2u(f));
		NodeStack stack(stack_mem.data());
		// the same texture as a NORMAL MAP (texture_image.cc:705 normalmap_; evalDerivative's other two branches, setup() without the / 100)
		tex->normalmap_ = true;
		for(TextureMapperNode *tm : mappers) { tm->bump_str_ = 1.f; tm->setup(); }
		for(ShaderNode *n : nodes) n->evalDerivative(stack, state, sp);
		for(size_t i = 0; i < nodes.size(); ++i) { push_rgba(bnout, stack_mem[i].col_); bnout.push_back(f2u(stack_mem[i].f_)); }
		tex->normalmap_ = false;
		for(TextureMapperNode *tm : mappers) { tm->bump_str_ = 1.f; tm->setup(); }
		for(ShaderNode *n : nodes) n->evalDerivative(stack, state, sp);
		for(size_t i = 0; i < nodes.size(); ++i) { push_rgba(bout, stack_mem[i].col_); bout.push_back(f2u(stack_mem[i].f_)); }
		float du, dv;
		nodes.back()->getDerivative(stack, du, dv);
		any_mat->applyBump(sp, du * 40.f, dv * 40.f);             // (scaled up: the harness's bump strengths are those of a 1 % bump)
		const float res[9] = {sp.n_.x_, sp.n_.y_, sp.n_.z_, sp.nu_.x_, sp.nu_.y_, sp.nu_.z_, sp.nv_.x_, sp.nv_.y_, sp.nv_.z_};
		for(float f : res) bsp.push_back(f2u(f));
	}
	j.arr_u32("bump_in", bin);
	j.arr_u32("bump_out", bout);
	j.arr_u32("bump_applied", bsp);
	j.arr_u32("bump_normalmap_out", bnout);
}

int main(int argc, char **argv)
{
	Json j;
	sec_image(j);
	sec_nodes(j);
	sec_files(j, argc > 1 ? argv[1] : "/root/reference/tests/test01");
	printf("{\n%s\n}\n", j.s.c_str());
	return 0;
}
