// The Interface-shaped C ABI (include/yafaray_c_api.h) over the MI355X path-tracing core.
//
// Host-side mirror of the reference's control path for ONE render: ParamMap building
// (src/interface/interface.cc:221-310), the type-string factories of RenderEnvironment
// (src/common/environment.cc:367-454 -> Material::factory etc.), the Scene geometry state machine
// (src/common/scene.cc:110-131,283-338), RenderEnvironment::setupScene (environment.cc:679-813)
// and Scene::update (scene.cc:784-894).  Everything per-sample happens on the device (yafgpu.h).
#include "../../include/yafaray_c_api.h"
#include "../../include/yafgpu.h"
#include "yafaray_image.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846;

// ---- ParamMap (include/common/param.h:37-107): strictly typed values, as Parameter::getVal (param.cc:49-53)
struct Param
{
	enum Type { None, Int, Bool, Float, String, Point, Color, Matrix } type = None;
	int i = 0; bool b = false; double f = 0; std::string s; float v[4] = {0, 0, 0, 0};
	float m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // Matrix4, row major
};
struct ParamMap
{
	std::map<std::string, Param> dicc;
	bool get(const std::string &n, int &o) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Int) return false; o = it->second.i; return true; }
	bool get(const std::string &n, bool &o) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Bool) return false; o = it->second.b; return true; }
	bool get(const std::string &n, float &o) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Float) return false; o = (float)it->second.f; return true; }
	bool get(const std::string &n, double &o) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Float) return false; o = it->second.f; return true; }
	bool get(const std::string &n, std::string &o) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::String) return false; o = it->second.s; return true; }
	bool getPoint(const std::string &n, float o[3]) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Point) return false; o[0] = it->second.v[0]; o[1] = it->second.v[1]; o[2] = it->second.v[2]; return true; }
	bool getColor(const std::string &n, float o[3]) const { auto it = dicc.find(n); if(it == dicc.end() || it->second.type != Param::Color) return false; o[0] = it->second.v[0]; o[1] = it->second.v[1]; o[2] = it->second.v[2]; return true; }
};

struct Mesh
{
	std::vector<float> points;        // xyz
	std::vector<float> normals;       // xyz per vertex (addNormal) or empty
	std::vector<int> tri;             // a,b,c
	std::vector<int> tri_mat;
	bool normals_exported = false, smooth = false, visible = true, base = false;
	std::vector<float> smooth_normals; // per triangle corner, filled by smoothMesh
	// texture coordinates as the exporter gives them (TriangleObject::points_ interleaves orco, uv_values_ / uv_offsets_;
	// object_geom/object_geom_mesh.cc): kept for the shader nodes; untextured shading never reads them
	bool has_orco = false, has_uv = false;
	std::vector<float> orco;          // xyz per vertex
	std::vector<float> uv;            // u, v per addUv
	std::vector<int> tri_uv;          // uv_a, uv_b, uv_c per triangle (when has_uv)
};

struct IntegratorCfg
{
	std::string type;
	int path_samples = 32, bounces = 3, rr_min_bounces = 0, raydepth = 5;
	bool no_recursive = false, bg_transp = false, bg_transp_refract = false, transp_shad = false;
	bool trace_caustics = false;     // PathIntegrator: caustic_type_ is Path unless the parameter says "none" (integrator_path_tracer.cc:36, :85, :382-387)
	int shadow_depth = 5;            // integrator_path_tracer.cc:352, integrator_direct_light.cc:201
};

struct CameraCfg { yafgpu_camera cam; };
struct BackgroundCfg { float color[3]; };

} // namespace

struct yafaray_material { yafgpu_material m; int index; std::vector<yafgpu_node> nodes; };   // nodes: evaluation order, `texture` = index into texture_order
struct yafaray_texture { yafgpu_texture t; int index; std::vector<float> texels; };            // texels: height * width * 4, as ImageHandler::getPixel returns them
struct yafaray_light { yafgpu_light l; };
struct yafaray_camera { CameraCfg c; };
struct yafaray_background { BackgroundCfg b; };
struct yafaray_integrator { IntegratorCfg c; };

struct yafaray_interface
{
	std::string err;
	ParamMap params;
	std::list<ParamMap> eparams;
	ParamMap *cparams = &params;
	// registries, name -> object (environment.h:85-95)
	std::map<std::string, std::unique_ptr<yafaray_material>> materials;
	std::vector<yafaray_material *> material_order;
	std::map<std::string, std::unique_ptr<yafaray_texture>> textures;
	std::vector<yafaray_texture *> texture_order;
	std::string base_dir;                // where relative texture file names are looked up after the working directory (the XML file's directory)
	std::map<std::string, std::unique_ptr<yafaray_light>> lights;
	std::vector<yafaray_light *> light_order;
	std::map<std::string, std::unique_ptr<yafaray_camera>> cameras;
	std::map<std::string, std::unique_ptr<yafaray_background>> backgrounds;
	std::map<std::string, std::unique_ptr<yafaray_integrator>> integrators;
	// scene state (scene.cc:110-131): 0 ready, 1 geometry, 2 object
	int state = -1;
	std::map<unsigned int, Mesh> meshes;
	Mesh *cur = nullptr, *last = nullptr;   // last: Scene's cur_obj_ outlives endTriMesh (smoothMesh(0, angle))
	unsigned int next_id = 1;
	bool geometry_changed = true;
	// render
	yafgpu_scene_t *gpu = nullptr;
	yafgpu_render_params rp{};
	yafgpu_aa_schedule aa{};
	std::vector<int32_t> resampled;      // pixels sampled by each pass of the last render
	bool prepared = false;
	bool scene_dirty = true;            // something the device scene is made of changed since it was built (Scene::update's state_.changes_, scene.cc:784-790)
	yafgpu_camera scene_cam{}; int scene_threads = 0;
	int shard_index = 0, shard_count = 1;
	std::vector<float> film;
	yafaray_render_stats_t stats{};
	// libc state behind the tile seeds of the Russian-roulette streams (integrator_tiled.cc:319): the reference's last
	// Material / ObjectGeometric constructor calls srand(its running index) and then draws a random colour
	// (material.cc:53-66, object_geom.cc:39-51); whichever object was made last leaves the state rand() continues from
	uint32_t last_srand = 0u; bool have_srand = false;
	int pass_pipelining = -1;            // yafaray_setPassPipelining: -1 by size, 0 off, 1 on (yafgpu_scene_set_pass_pipelining)
	int user_srand = -1, user_skip = 0;  // yafaray_setRandState: the embedder's own srand() after the last constructor
	bool serial_replay = true;           // yafaray_setSerialReplay
	std::vector<int32_t> tile_rand0;     // the first pass's value per tile (yafaray_renderPassDevice)
	yafgpu_exchange_fn exchange = nullptr; void *exchange_user = nullptr;     // yafaray_setPlaneExchange
	volatile int32_t abort_flag = 0;     // Scene::abort: set by yafaray_abort (any thread), polled by the device side between chunks and passes
	std::string color_space = "Raw_Manual_Gamma"; float gamma = 1.f;
	std::string color_space2 = "Raw_Manual_Gamma"; float gamma2 = 1.f;
	// Interface::setInputColorSpace (interface.cc:292-301): how paramsSetColor reads its arguments; the ctor's default is
	// RawManualGamma with gamma 1 (interface.cc:73), i.e. values are taken as linear
	int input_color_space = 3; float input_gamma = 1.f;       // 0 sRGB, 1 XYZ (D65), 2 LinearRGB, 3 RawManualGamma
	yafaray_output_t output2{}; bool has_output2 = false;
	bool interactive = false; std::string badge_position = "none";
};

namespace {

bool fail(yafaray_interface *yi, const std::string &m) { yi->err = m; return false; }

// fPow__ = fExp2__(fLog2__(a) * b), util_math_optimizations.h:116-142,176-183 (FAST_MATH is on in the reference's build)
float host_fexp2(float x)
{
	x = std::min(x, 129.00000f);
	x = std::max(x, -126.99999f);
	const int ipart = (int)(x - 0.5f);
	const float p = (x - (float)ipart);
	const int bits = (int)((unsigned)(ipart + 127) << 23);
	float expi; std::memcpy(&expi, &bits, 4);
	const float poly = (p * (p * (p * (p * (p * 1.8775767e-3f + 8.9893397e-3f) + 5.5826318e-2f) + 2.4015361e-1f) + 6.9315308e-1f) + 9.9999994e-1f);
	return expi * poly;
}
float host_flog2(float x)
{
	int i; std::memcpy(&i, &x, 4);
	const float e = (float)(((i & 0x7F800000) >> 23) - 127);
	const int mi = (i & 0x7FFFFF) | 0x3F800000;
	float m; std::memcpy(&m, &mi, 4);
	const float a = m * -3.4436006e-2f + 3.1821337e-1f;
	const float b = m * a + -1.2315303f;
	const double c = (double)(m * b) + 2.5988452;
	const double d = (double)m * c + (double)-3.3241990f;
	const double ee = (double)m * d + (double)3.1157899f;
	return ((float)ee * (m - 1.0f) + e);
}
float host_fpow(float a, float b) { return host_fexp2(host_flog2(a) * b); }

// Material::material_index_auto_ / ObjectGeometric::object_index_auto_ (common/material.cc:33, object_geom.cc:29): static,
// process-wide, never reset — like the reference, one process is assumed to build its scenes one after the other
unsigned int g_material_index_auto = 0u, g_object_index_auto = 0u;
void note_srand(yafaray_interface *yi, unsigned int seed) { yi->last_srand = seed; yi->have_srand = true; yi->user_srand = -1; yi->user_skip = 0; }
// values the constructor's colour loop consumed after its srand(): do { r, g, b = rand() % 8 / 8 } while(r + g + b < 0.5)
int colour_loop_draws(uint32_t seed)
{
	int32_t v[3 * 64];
	yafgpu_glibc_rand(seed, 3 * 64, v);
	for(int k = 0; k < 64; ++k)
	{
		const float r = (float)(v[3 * k] % 8) / 8.f, g = (float)(v[3 * k + 1] % 8) / 8.f, b = (float)(v[3 * k + 2] % 8) / 8.f;
		if(!(r + g + b < 0.5f)) return 3 * (k + 1);
	}
	return 3 * 64;
}

inline void cross3(const float a[3], const float b[3], float o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline void normalize3(float v[3]) // Vec3::normalize, vector.h:227-238
{
	float len = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	if(len != 0.f) { len = 1.0f / std::sqrt(len); v[0] *= len; v[1] *= len; v[2] *= len; }
}

int visibility_from(const std::string &s)
{
	if(s == "no_shadows") return 1;
	if(s == "shadow_only") return 2;
	if(s == "invisible") return 3;
	return 0;
}


// ---- shader nodes -----------------------------------------------------------------------------------
inline void clear_shader_slots(yafgpu_material &m)
{
	m.node_first = 0; m.n_nodes = 0;
	m.sh_diffuse = m.sh_mirror_color = m.sh_mirror = m.sh_transparency = m.sh_translucency = m.sh_sigma_oren = m.sh_diffuse_refl = m.sh_ior = -1;
	m.sh_glossy = m.sh_glossy_reflect = m.sh_exponent = m.sh_filter_color = -1;
	m.bump_first = 0; m.n_bump = 0; m.sh_bump = -1;
}

constexpr int kMaxMaterialNodes = 16;    // yafgpu_texture.h kMaxNodes: the per-lane node stack of the shading kernels

struct LoadedNodes
{
	std::vector<yafgpu_node> nodes;                 // as listed (loadNodes order), references = indices into this list
	std::map<std::string, int> by_name;
};

// the four factories + configInputs (shader_node.cc:28-37; shader_node_basic.cc:345-409, :425-434, :482-531, :683-703;
// shader_node_layer.cc:162-201, :211-245).  -1 = the reference's own failure (it then builds the material without shaders,
// material_shiny_diffuse.cc:709-713), 0 = refused by the GPU path (err set), 1 = loaded
int load_nodes(yafaray_interface *yi, const std::list<ParamMap> &list, LoadedNodes &out)
{
	if(list.size() > 4096) { fail(yi, "shader nodes: more than 4096 nodes in one material's list"); return 0; }      // (sort_nodes walks the graph recursively)
	std::vector<const ParamMap *> maps;
	for(const ParamMap &pm : list)
	{	// NodeMaterial::loadNodes, material_node.cc:150-205
		std::string element, name, type;
		if(pm.get("element", element) && element != "shader_node") continue;
		if(!pm.get("name", name)) return -1;
		if(out.by_name.count(name)) return -1;
		if(!pm.get("type", type)) return -1;
		yafgpu_node n; std::memset(&n, 0, sizeof n);
		n.texture = n.input1 = n.input2 = n.factor = n.input = n.upper = -1;
		if(type == "texture_mapper")
		{
			n.type = YAFGPU_NODE_TEXTURE_MAPPER;
			std::string texname, option;
			if(!pm.get("texture", texname)) return -1;
			auto tex = yi->textures.find(texname);
			if(tex == yi->textures.end()) return -1;
			n.texture = tex->second->index;
			n.texco = 1;         // Coords { Uv, Glob, Orco, Tran, Nor, Refl, Win, Stick, Stress, Tan }, shader_node_basic.h
			if(pm.get("texco", option))
			{
				static const char *names[] = {"uv", "global", "orco", "transformed", "normal", "reflect", "window", "stick", "stress", "tangent"};
				for(int k = 0; k < 10; ++k) if(option == names[k]) n.texco = k;
			}
			n.mapping = 0;       // Projection { Plain, Cube, Tube, Sphere }; image textures are discrete()
			if(pm.get("mapping", option))
			{
				static const char *names[] = {"plain", "cube", "tube", "sphere"};
				for(int k = 0; k < 4; ++k) if(option == names[k]) n.mapping = k;
			}
			float scale[3] = {1, 1, 1}, offset[3] = {0, 0, 0}; bool scalar = true; int map[3] = {1, 2, 3};
			for(int k = 0; k < 16; ++k) n.mtx[k] = (k % 5 == 0) ? 1.f : 0.f;
			{ auto it = pm.dicc.find("transform"); if(it != pm.dicc.end() && it->second.type == Param::Matrix) std::memcpy(n.mtx, it->second.m, sizeof n.mtx); }
			pm.getPoint("scale", scale); pm.getPoint("offset", offset); pm.get("do_scalar", scalar);
			pm.get("proj_x", map[0]); pm.get("proj_y", map[1]); pm.get("proj_z", map[2]);
			for(int k = 0; k < 3; ++k) map[k] = std::min(3, std::max(0, map[k]));
			n.map_x = map[0]; n.map_y = map[1]; n.map_z = map[2];
			for(int k = 0; k < 3; ++k) { n.scale[k] = scale[k]; n.offset[k] = 2 * offset[k]; }
			n.do_scalar = scalar ? 1 : 0;
			{	// setup(), shader_node_basic.cc:34-59 (image textures are discrete; normal maps are refused at createTexture)
				float bump_str = 1.f; pm.get("bump_strength", bump_str);
				const yafgpu_texture &t = tex->second->t;
				n.d_u = 1.f / (float)t.width; n.d_v = 1.f / (float)t.height;
				bump_str /= std::sqrt(scale[0] * scale[0] + scale[1] * scale[1] + scale[2] * scale[2]);
				if(!t.normalmap) bump_str /= 100.0f;
				n.bump_str = bump_str;
			}
		}
		else if(type == "value")
		{
			n.type = YAFGPU_NODE_VALUE;
			float col[3] = {1, 1, 1}, alpha = 1.f, val = 1.f;
			pm.getColor("color", col); pm.get("alpha", alpha); pm.get("scalar", val);
			n.color[0] = col[0]; n.color[1] = col[1]; n.color[2] = col[2]; n.color[3] = alpha; n.value = val;
		}
		else if(type == "mix")
		{
			n.type = YAFGPU_NODE_MIX;
			float val = 0.5f; int mode = 0;
			pm.get("cfactor", val); pm.get("mode", mode);
			// MixNode::factory's switch has no case for MnDiv (5) and none above MnOverlay (9): those build a plain MixNode
			n.mode = (mode >= 0 && mode <= 9 && mode != 5) ? mode : 0;
			n.cfactor = (n.mode == 0) ? val : 0.f;       // MixNode(val) / the derived nodes' MixNode(): cfactor_ 0
			n.col1[3] = n.col2[3] = 1.f;
		}
		else if(type == "layer")
		{
			n.type = YAFGPU_NODE_LAYER;
			float def_col[3] = {1, 1, 1};
			bool do_color = true, do_scalar = false, color_input = true, use_alpha = false, stencil = false, no_rgb = false, negative = false;
			double def_val = 1.0, colfac = 1.0, valfac = 1.0; int mode = 0;
			pm.get("mode", mode); pm.getColor("def_col", def_col); pm.get("colfac", colfac); pm.get("def_val", def_val); pm.get("valfac", valfac);
			pm.get("do_color", do_color); pm.get("do_scalar", do_scalar); pm.get("color_input", color_input); pm.get("use_alpha", use_alpha);
			pm.get("noRGB", no_rgb); pm.get("stencil", stencil); pm.get("negative", negative);
			n.texflag = (no_rgb ? 1u : 0u) | (stencil ? 2u : 0u) | (negative ? 4u : 0u) | (use_alpha ? 8u : 0u);
			n.mode = mode; n.colfac = (float)colfac; n.valfac = (float)valfac; n.def_val = (float)def_val;
			n.def_col[0] = def_col[0]; n.def_col[1] = def_col[1]; n.def_col[2] = def_col[2]; n.def_col[3] = 1.f;
			n.do_color = do_color; n.do_scalar_l = do_scalar; n.color_input = color_input; n.use_alpha = use_alpha;
		}
		else return -1;      // ShaderNode::factory knows no other type: "No shader node was constructed by plugin"
		out.by_name[name] = (int)out.nodes.size();
		out.nodes.push_back(n);
		maps.push_back(&pm);
	}
	for(size_t k = 0; k < out.nodes.size(); ++k)
	{	// configInputs, material_node.cc:207-220
		yafgpu_node &n = out.nodes[k]; const ParamMap &pm = *maps[k];
		auto find = [&](const std::string &nm) { auto it = out.by_name.find(nm); return it == out.by_name.end() ? -1 : it->second; };
		auto rgba = [&](const char *key, float o[4]) { auto it = pm.dicc.find(key); if(it == pm.dicc.end() || it->second.type != Param::Color) return false; for(int c = 0; c < 4; ++c) o[c] = it->second.v[c]; return true; };
		std::string name;
		if(n.type == YAFGPU_NODE_MIX)
		{
			if(pm.get("input1", name)) { if((n.input1 = find(name)) < 0) return -1; }
			else if(!rgba("color1", n.col1)) return -1;
			if(pm.get("input2", name)) { if((n.input2 = find(name)) < 0) return -1; }
			else if(!rgba("color2", n.col2)) return -1;
			if(pm.get("factor", name)) { if((n.factor = find(name)) < 0) return -1; }
			else if(!pm.get("value", n.cfactor)) return -1;
		}
		else if(n.type == YAFGPU_NODE_LAYER)
		{
			if(!pm.get("input", name)) return -1;
			if((n.input = find(name)) < 0) return -1;
			if(pm.get("upper_layer", name)) { if((n.upper = find(name)) < 0) return -1; }
			else
			{
				n.upper_col[3] = 1.f;
				if(!rgba("upper_color", n.upper_col)) { n.upper_col[0] = n.upper_col[1] = n.upper_col[2] = 0.f; n.upper_col[3] = 1.f; }
				if(!pm.get("upper_value", n.upper_val)) n.upper_val = 0.f;
			}
		}
	}
	return 1;
}

// NodeMaterial::solveNodesOrder + getNodeList (material_node.cc:88-121): the nodes the material's shader slots reach, every node
// after the ones it reads.  slots: in = index into ld.nodes or -1, out = index into `sorted`.  false: a cycle, or more nodes
// than the shading kernels' stack holds (err set).
bool sort_nodes(yafaray_interface *yi, const LoadedNodes &ld, int *slots, int n_slots, std::vector<yafgpu_node> &sorted)
{
	const int n = (int)ld.nodes.size();
	std::vector<int> new_index((size_t)n, -1), mark((size_t)n, 0);
	std::vector<int> order;
	bool cycle = false;
	std::function<void(int)> visit = [&](int k)
	{
		if(k < 0 || mark[(size_t)k] == 2) return;
		if(mark[(size_t)k] == 1) { cycle = true; return; }
		mark[(size_t)k] = 1;
		const yafgpu_node &nd = ld.nodes[(size_t)k];
		visit(nd.input1); visit(nd.input2); visit(nd.factor); visit(nd.input); visit(nd.upper);
		mark[(size_t)k] = 2;
		new_index[(size_t)k] = (int)order.size(); order.push_back(k);
	};
	for(int s = 0; s < n_slots; ++s) visit(slots[s]);
	if(cycle) return fail(yi, "shader nodes: the node graph has a cycle");
	if((int)order.size() > kMaxMaterialNodes) return fail(yi, "shader nodes: more than 16 nodes reachable from one material's shader slots is not supported by the GPU path");
	auto remap = [&](int k) { return k < 0 ? -1 : new_index[(size_t)k]; };
	sorted.clear();
	for(int k : order)
	{
		yafgpu_node nd = ld.nodes[(size_t)k];
		nd.input1 = remap(nd.input1); nd.input2 = remap(nd.input2); nd.factor = remap(nd.factor); nd.input = remap(nd.input); nd.upper = remap(nd.upper);
		sorted.push_back(nd);
	}
	for(int s = 0; s < n_slots; ++s) slots[s] = remap(slots[s]);
	return true;
}

// NodeMaterial's bump_nodes_ (getNodeList(bump_shader_, bump_nodes_), material_shiny_diffuse.cc:746, material_glossy.cc:545, ...): what the
// bump shader reaches, in evaluation order, kept behind the material's colour nodes.  first / count / slot: relative to `nodes`.
struct BumpList { int first = 0, count = 0, slot = -1; };
bool append_bump_nodes(yafaray_interface *yi, const LoadedNodes &ld, int bump_slot, std::vector<yafgpu_node> &nodes, BumpList &out)
{
	out = BumpList();
	if(bump_slot < 0) return true;
	std::vector<yafgpu_node> bump; int slot[1] = {bump_slot};
	if(!sort_nodes(yi, ld, slot, 1, bump)) return false;
	out.first = (int)nodes.size(); out.count = (int)bump.size(); out.slot = slot[0];
	nodes.insert(nodes.end(), bump.begin(), bump.end());
	return true;
}

// ShinyDiffuseMaterial::factory + ctor + config, material_shiny_diffuse.cc:599-690, :26-36, :46-92
bool make_shinydiffuse(yafaray_interface *yi, const ParamMap &p, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	float color[3] = {1, 1, 1}, mirror_color[3] = {1, 1, 1};
	float diffuse = 1.f, transp = 0.f, transl = 0.f, mirror = 0.f, emit = 0.f, ior = 1.33f, wire = 0.f;
	bool fresnel = false, recv = true, flat = false;
	double transmit_filter = 1.0;
	std::string vis = "normal", brdf;
	p.getColor("color", color); p.getColor("mirror_color", mirror_color);
	p.get("transparency", transp); p.get("translucency", transl); p.get("diffuse_reflect", diffuse);
	p.get("specular_reflect", mirror); p.get("emit", emit); p.get("IOR", ior); p.get("fresnel_effect", fresnel);
	p.get("transmit_filter", transmit_filter); p.get("receive_shadows", recv); p.get("flat_material", flat);
	p.get("visibility", vis); p.get("wireframe_amount", wire);
	if(wire != 0.f) return fail(yi, "shinydiffusemat: wireframe shading is not supported by the GPU path");
	// recursiveRaytrace's per-material parameters (integrator_montecarlo.cc:791, :1003-1014).  (`samplingfactor` only feeds a debug
	// render pass, integrator_tiled.cc:597.)
	int add_depth = 0; float tb_factor = 0.f; bool tb_mult = false;
	p.get("additionaldepth", add_depth); p.get("transparentbias_factor", tb_factor); p.get("transparentbias_multiply_raydepth", tb_mult);
	if(add_depth < 0 || add_depth > 7) return fail(yi, "shinydiffusemat: additionaldepth outside [0, 7]: the device path keeps at most 7 recursion frames per sample");
	// shader nodes, material_shiny_diffuse.cc:692-747: slots in the order of yafgpu_material's sh_* fields
	enum { kDiffuse, kMirrorColor, kMirror, kTransparency, kTranslucency, kSigmaOren, kDiffuseRefl, kIor, kBump, kWireframe, kSlots };
	int slots[kSlots]; for(int &v : slots) v = -1;
	int n_color_nodes = 0; BumpList bump;
	nodes.clear();
	if(!yi->eparams.empty())
	{
		LoadedNodes ld;
		const int rc = load_nodes(yi, yi->eparams, ld);
		if(rc == 0) return false;
		if(rc > 0)
		{	// parseNodes, material_node.cc:227-247: a slot naming a node that does not exist stays empty
			static const char *names[kSlots] = {"diffuse_shader", "mirror_color_shader", "mirror_shader", "transparency_shader", "translucency_shader",
			                                    "sigma_oren_shader", "diffuse_refl_shader", "IOR_shader", "bump_shader", "wireframe_shader"};
			std::string node;
			for(int k = 0; k < kSlots; ++k) if(p.get(names[k], node)) { auto it = ld.by_name.find(node); if(it != ld.by_name.end()) slots[k] = it->second; }
			if(slots[kWireframe] >= 0) return fail(yi, "shinydiffusemat: wireframe_shader is not supported by the GPU path");
			const int bump_slot = slots[kBump]; slots[kBump] = -1;
			if(!sort_nodes(yi, ld, slots, kSlots, nodes)) return false;
			n_color_nodes = (int)nodes.size();
			if(!append_bump_nodes(yi, ld, bump_slot, nodes, bump)) return false;
		}
		else std::fprintf(stderr, "WARNING: ShinyDiffuse: Loading shader nodes failed! (the material is built without them, as the reference does)\n");
	}
	std::memset(&m, 0, sizeof m);
	clear_shader_slots(m);
	m.n_nodes = (int32_t)n_color_nodes; m.bump_first = bump.first; m.n_bump = bump.count; m.sh_bump = bump.slot;
	m.sh_diffuse = slots[kDiffuse]; m.sh_mirror_color = slots[kMirrorColor]; m.sh_mirror = slots[kMirror]; m.sh_transparency = slots[kTransparency];
	m.sh_translucency = slots[kTranslucency]; m.sh_sigma_oren = slots[kSigmaOren]; m.sh_diffuse_refl = slots[kDiffuseRefl]; m.sh_ior = slots[kIor];
	m.ior_base = ior; m.emit_strength = emit;
	m.additional_depth = add_depth; m.transp_bias_factor = tb_factor; m.transp_bias_mult = tb_mult ? 1 : 0;
	m.type = YAFGPU_MAT_SHINYDIFFUSE; m.visibility = visibility_from(vis); m.receive_shadows = recv; m.flat = flat;
	for(int k = 0; k < 3; ++k) { m.diffuse_color[k] = color[k]; m.mirror_color[k] = mirror_color[k]; m.emit_color[k] = emit * color[k]; }
	m.diffuse_strength = diffuse; m.transparency_strength = transp; m.translucency_strength = transl; m.mirror_strength = mirror;
	m.transmit_filter = (float)transmit_filter;
	m.bsdf_flags = 0u;
	if(emit > 0.f) m.bsdf_flags |= 0x80u;
	m.ior_squared = 1.f;
	if(fresnel) { m.ior_squared = ior * ior; m.has_fresnel = 1; }
	if(p.get("diffuse_brdf", brdf) && brdf == "oren_nayar")
	{	// initOrenNayar :190-196
		double sigma = 0.1; p.get("sigma", sigma);
		const double s2 = sigma * sigma;
		m.oren_a = (float)(1.0 - 0.5 * (s2 / (s2 + 0.33)));
		m.oren_b = (float)(0.45 * s2 / (s2 + 0.09));
		m.use_oren = 1;
	}
	float acc = 1.f;
	m.n_bsdf = 0;
	if(m.mirror_strength > 0.00001f || m.sh_mirror >= 0)
	{
		m.is_mirror = 1;
		if(m.sh_mirror >= 0) {}
		else if(!m.has_fresnel) acc = 1.f - m.mirror_strength;
		m.bsdf_flags |= 0x1u | 0x10u;
		m.c_flags[m.n_bsdf] = 0x1u | 0x10u; m.c_index[m.n_bsdf] = 0; ++m.n_bsdf;
	}
	if(m.transparency_strength * acc > 0.00001f || m.sh_transparency >= 0)
	{
		m.is_transparent = 1;
		if(m.sh_transparency < 0) acc *= 1.f - m.transparency_strength;
		m.bsdf_flags |= 0x20u | 0x40u;
		m.c_flags[m.n_bsdf] = 0x20u | 0x40u; m.c_index[m.n_bsdf] = 1; ++m.n_bsdf;
	}
	if(m.translucency_strength * acc > 0.00001f || m.sh_translucency >= 0)
	{
		m.is_translucent = 1;
		if(m.sh_translucency < 0) acc *= 1.f - m.transparency_strength; // sic, material_shiny_diffuse.cc:72
		m.bsdf_flags |= 0x4u | 0x20u;
		m.c_flags[m.n_bsdf] = 0x4u | 0x20u; m.c_index[m.n_bsdf] = 2; ++m.n_bsdf;
	}
	if(m.diffuse_strength * acc > 0.00001f)
	{
		m.is_diffuse = 1;
		m.bsdf_flags |= 0x4u | 0x10u;
		m.c_flags[m.n_bsdf] = 0x4u | 0x10u; m.c_index[m.n_bsdf] = 3; ++m.n_bsdf;
	}
	return true;
}

// shader slots of glossy / coated_glossy (material_glossy.cc:490-530, material_coated_glossy.cc:556-602): loads the node list, maps the
// named slots and keeps what they reach in evaluation order.  false: refused (err set).
bool glossy_nodes(yafaray_interface *yi, const ParamMap &p, const char *what, bool coated, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	enum { kDiffuse, kGlossy, kGlossyReflect, kSigmaOren, kExponent, kDiffuseRefl, kIor, kMirror, kMirrorColor, kBump, kWireframe, kSlots };
	static const char *names[kSlots] = {"diffuse_shader", "glossy_shader", "glossy_reflect_shader", "sigma_oren_shader", "exponent_shader", "diffuse_refl_shader",
	                                    "IOR_shader", "mirror_shader", "mirror_color_shader", "bump_shader", "wireframe_shader"};
	int slots[kSlots]; for(int &v : slots) v = -1;
	nodes.clear();
	clear_shader_slots(m);
	if(yi->eparams.empty()) return true;
	LoadedNodes ld;
	const int rc = load_nodes(yi, yi->eparams, ld);
	if(rc == 0) return false;
	if(rc < 0) { std::fprintf(stderr, "WARNING: %s: Loading shader nodes failed! (the material is built without them, as the reference does)\n", what); return true; }
	std::string node;
	for(int k = 0; k < kSlots; ++k)
	{
		if(!coated && (k == kIor || k == kMirror || k == kMirrorColor)) continue;       // glossy has no such slots
		if(p.get(names[k], node)) { auto it = ld.by_name.find(node); if(it != ld.by_name.end()) slots[k] = it->second; }
	}
	if(slots[kWireframe] >= 0) return fail(yi, std::string(what) + ": wireframe_shader is not supported by the GPU path");
	const int bump_slot = slots[kBump]; slots[kBump] = -1;
	if(!sort_nodes(yi, ld, slots, kSlots, nodes)) return false;
	m.n_nodes = (int32_t)nodes.size();
	BumpList bump;
	if(!append_bump_nodes(yi, ld, bump_slot, nodes, bump)) return false;
	m.bump_first = bump.first; m.n_bump = bump.count; m.sh_bump = bump.slot;
	m.sh_diffuse = slots[kDiffuse]; m.sh_glossy = slots[kGlossy]; m.sh_glossy_reflect = slots[kGlossyReflect]; m.sh_sigma_oren = slots[kSigmaOren];
	m.sh_exponent = slots[kExponent]; m.sh_diffuse_refl = slots[kDiffuseRefl]; m.sh_ior = slots[kIor]; m.sh_mirror = slots[kMirror]; m.sh_mirror_color = slots[kMirrorColor];
	return true;
}

// GlossyMaterial::factory + ctor, material_glossy.cc:407-472, :32-50
bool make_glossy(yafaray_interface *yi, const ParamMap &p, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	float col[3] = {1, 1, 1}, dcol[3] = {1, 1, 1};
	float refl = 1.f, diff = 0.f, exponent = 50.f, wire = 0.f;
	bool as_diff = true, aniso = false, recv = true;
	std::string vis = "normal", brdf;
	p.getColor("color", col); p.getColor("diffuse_color", dcol); p.get("diffuse_reflect", diff); p.get("glossy_reflect", refl);
	p.get("as_diffuse", as_diff); p.get("exponent", exponent); p.get("anisotropic", aniso);
	p.get("receive_shadows", recv); p.get("visibility", vis); p.get("wireframe_amount", wire);
	if(wire != 0.f) return fail(yi, "glossy: wireframe shading is not supported by the GPU path");
	int add_depth = 0; p.get("additionaldepth", add_depth);
	if(add_depth < 0 || add_depth > 7) return fail(yi, "glossy: additionaldepth outside [0, 7]: the device path keeps at most 7 recursion frames per sample");
	std::memset(&m, 0, sizeof m);
	if(!glossy_nodes(yi, p, "glossy", false, m, nodes)) return false;
	m.type = YAFGPU_MAT_GLOSSY; m.visibility = visibility_from(vis); m.receive_shadows = recv;
	m.additional_depth = add_depth;
	for(int k = 0; k < 3; ++k) { m.gloss_color[k] = col[k]; m.diff_color[k] = dcol[k]; }
	m.exponent = exponent; m.reflectivity = refl; m.diffuse = diff; m.as_diffuse = as_diff;
	if(aniso)
	{	// material_glossy.cc:464-472
		float e_u = 50.f, e_v = 50.f;
		p.get("exp_u", e_u); p.get("exp_v", e_v);
		m.anisotropic = 1; m.exp_u = e_u; m.exp_v = e_v;
	}
	m.bsdf_flags = 0u;
	if(diff > 0) { m.bsdf_flags = 0x4u | 0x10u; m.with_diffuse = 1; }
	m.bsdf_flags |= as_diff ? (0x4u | 0x10u) : (0x2u | 0x10u);
	if(p.get("diffuse_brdf", brdf) && brdf == "Oren-Nayar")
	{	// :66-72
		double sigma = 0.1; p.get("sigma", sigma);
		const double s2 = sigma * sigma;
		m.oren_a = (float)(1.0 - 0.5 * (s2 / (s2 + 0.33)));
		m.oren_b = (float)(0.45 * s2 / (s2 + 0.09));
		m.use_oren = 1;
	}
	return true;
}

// LightMaterial::factory, material_simple.cc:63-73
bool make_lightmat(const ParamMap &p, yafgpu_material &m)
{
	float col[3] = {1, 1, 1}; double power = 1.0; bool ds = false;
	p.getColor("color", col); p.get("power", power); p.get("double_sided", ds);
	std::memset(&m, 0, sizeof m);
	m.type = YAFGPU_MAT_LIGHT; m.receive_shadows = 1;
	for(int k = 0; k < 3; ++k) m.light_col[k] = (float)power * col[k];
	m.double_sided = ds;
	m.bsdf_flags = 0x80u;
	return true;
}

// CoatedGlossyMaterial::factory + ctor, material_coated_glossy.cc:464-560, :41-66 (Blinn lobe, as_diffuse only)
bool make_coated_glossy(yafaray_interface *yi, const ParamMap &p, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	float col[3] = {1, 1, 1}, dcol[3] = {1, 1, 1}, mcol[3] = {1, 1, 1};
	float refl = 1.f, diff = 0.f, exponent = 50.f, mirror = 1.f, wire = 0.f; double ior = 1.4, sigma = 0.1;
	bool as_diff = true, aniso = false, recv = true; std::string vis = "normal", brdf; int add_depth = 0;
	p.getColor("color", col); p.getColor("diffuse_color", dcol); p.get("diffuse_reflect", diff); p.get("glossy_reflect", refl);
	p.get("as_diffuse", as_diff); p.get("exponent", exponent); p.get("anisotropic", aniso); p.get("IOR", ior);
	p.getColor("mirror_color", mcol); p.get("specular_reflect", mirror);
	p.get("receive_shadows", recv); p.get("visibility", vis); p.get("additionaldepth", add_depth); p.get("wireframe_amount", wire);
	if(wire != 0.f) return fail(yi, "coated_glossy: wireframe shading is not supported by the GPU path");
	if(add_depth < 0 || add_depth > 7) return fail(yi, "coated_glossy: additionaldepth outside [0, 7]: the device path keeps at most 7 recursion frames per sample");
	if(ior == 1.0) ior = 1.0000001f;                                // :512
	std::memset(&m, 0, sizeof m);
	if(!glossy_nodes(yi, p, "coated_glossy", true, m, nodes)) return false;
	m.ior_base = (float)ior;
	m.type = YAFGPU_MAT_COATED_GLOSSY; m.visibility = visibility_from(vis); m.receive_shadows = recv;
	m.additional_depth = add_depth;
	for(int k = 0; k < 3; ++k) { m.gloss_color[k] = col[k]; m.diff_color[k] = dcol[k]; m.mirror_color[k] = mcol[k]; }
	m.mirror_strength = mirror; m.glass_ior = (float)ior; m.exponent = exponent; m.reflectivity = refl; m.diffuse = diff; m.as_diffuse = as_diff;
	if(aniso)
	{	// material_coated_glossy.cc:529-537
		float e_u = 50.f, e_v = 50.f;
		p.get("exp_u", e_u); p.get("exp_v", e_v);
		m.anisotropic = 1; m.exp_u = e_u; m.exp_v = e_v;
	}
	m.c_flags[0] = 0x1u | 0x10u;                                    // Specular | Reflect
	m.c_flags[1] = as_diff ? (0x4u | 0x10u) : (0x2u | 0x10u);         // :55: as_diffuse ? Diffuse | Reflect : Glossy | Reflect (recursiveRaytrace's glossy branch then samples it)
	if(diff > 0.f) { m.c_flags[2] = 0x4u | 0x10u; m.with_diffuse = 1; m.n_bsdf = 3; }
	else { m.c_flags[2] = 0u; m.n_bsdf = 2; }
	m.bsdf_flags = m.c_flags[0] | m.c_flags[1] | m.c_flags[2];
	if(p.get("diffuse_brdf", brdf) && brdf == "Oren-Nayar")
	{	// initOrenNayar :81-87
		p.get("sigma", sigma);
		const double s2 = sigma * sigma;
		m.oren_a = (float)(1.0 - 0.5 * (s2 / (s2 + 0.33))); m.oren_b = (float)(0.45 * s2 / (s2 + 0.09)); m.use_oren = 1;
	}
	return true;
}

// GlassMaterial::factory + ctor, material_glass.cc:340-443, :32-49 (no dispersion or shader nodes)
bool make_glass(yafaray_interface *yi, const ParamMap &p, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	double ior = 1.4, filt = 0.0, disp = 0.0; float fcol[3] = {1, 1, 1}, scol[3] = {1, 1, 1}, absorp[3] = {1, 1, 1}, wire = 0.f;
	bool fake = false, recv = true; std::string vis = "normal"; int add_depth = 0;
	p.get("IOR", ior); p.getColor("filter_color", fcol); p.get("transmit_filter", filt); p.getColor("mirror_color", scol);
	p.get("dispersion_power", disp); p.get("fake_shadows", fake); p.get("receive_shadows", recv); p.get("visibility", vis);
	p.get("additionaldepth", add_depth); p.get("wireframe_amount", wire); p.getColor("absorption", absorp);
	if(disp > 0.0) return fail(yi, "glass: dispersion is not supported by the GPU path (recursiveRaytrace's dispersive branch)");
	if(add_depth < 0 || add_depth > 7) return fail(yi, "glass: additionaldepth outside [0, 7]: the device path keeps at most 7 recursion frames per sample");
	if(wire != 0.f) return fail(yi, "glass: wireframe shading is not supported by the GPU path");
	std::memset(&m, 0, sizeof m);
	clear_shader_slots(m);
	nodes.clear();
	if(!yi->eparams.empty())
	{	// material_glass.cc:402-441: mirror_color_shader, filter_color_shader, IOR_shader, bump_shader (wireframe refused)
		enum { kMirrorColor, kFilterColor, kIor, kBump, kWireframe, kSlots };
		static const char *names[kSlots] = {"mirror_color_shader", "filter_color_shader", "IOR_shader", "bump_shader", "wireframe_shader"};
		int slots[kSlots]; for(int &v : slots) v = -1;
		LoadedNodes ld;
		const int rc = load_nodes(yi, yi->eparams, ld);
		if(rc == 0) return false;
		if(rc > 0)
		{
			std::string node;
			for(int k = 0; k < kSlots; ++k) if(p.get(names[k], node)) { auto it = ld.by_name.find(node); if(it != ld.by_name.end()) slots[k] = it->second; }
			if(slots[kWireframe] >= 0) return fail(yi, "glass: wireframe_shader is not supported by the GPU path");
			const int bump_slot = slots[kBump]; slots[kBump] = -1;
			if(!sort_nodes(yi, ld, slots, kSlots, nodes)) return false;
			m.n_nodes = (int32_t)nodes.size();
			BumpList bump;
			if(!append_bump_nodes(yi, ld, bump_slot, nodes, bump)) return false;
			m.bump_first = bump.first; m.n_bump = bump.count; m.sh_bump = bump.slot;
			m.sh_mirror_color = slots[kMirrorColor]; m.sh_filter_color = slots[kFilterColor]; m.sh_ior = slots[kIor];
		}
		else std::fprintf(stderr, "WARNING: Glass: Loading shader nodes failed! (the material is built without them, as the reference does)\n");
	}
	m.ior_base = (float)ior; m.transp_ior = (float)ior;
	m.type = YAFGPU_MAT_GLASS; m.receive_shadows = recv; m.visibility = visibility_from(vis);
	m.additional_depth = add_depth;
	m.glass_ior = (float)ior;
	const float ff = (float)filt, fc = (float)(1.f - filt);       // filt * filt_col + Rgb(1.f - filt)
	for(int k = 0; k < 3; ++k) { m.filter_color[k] = ff * fcol[k] + fc; m.mirror_color[k] = scol[k]; }
	m.fake_shadow = fake;
	m.bsdf_flags = 0x1u | 0x10u | 0x20u;                         // BsdfAllSpecular
	if(fake) m.bsdf_flags |= 0x40u;
	m.tm_flags = fake ? (0x40u | 0x20u) : (0x1u | 0x20u);
	if(absorp[0] < 1.f || absorp[1] < 1.f || absorp[2] < 1.f)
	{	// material_glass.cc:371-398: vol_i_ = BeerVolumeHandler(absorption, absorption_dist); volumehandler_beer.cc:28-35
		double dist = 1.0;
		p.get("absorption_dist", dist);
		const float maxlog = (float)std::log(1e38);
		for(int k = 0; k < 3; ++k)
		{
			m.beer_sigma[k] = (absorp[k] > 1e-38) ? (float)-std::log((double)absorp[k]) : maxlog;
			if(dist != 0.f) m.beer_sigma[k] = m.beer_sigma[k] * (float)(1.f / dist);
		}
		m.has_vol_i = 1;
		m.bsdf_flags |= 0x100u;                                   // BsdfVolumetric
	}
	return true;
}
// RoughGlassMaterial::factory + ctor, material_rough_glass.cc:320-455, :33-48 (no dispersion, no shader nodes)
bool make_rough_glass(yafaray_interface *yi, const ParamMap &p, yafgpu_material &m, std::vector<yafgpu_node> &nodes)
{
	float ior = 1.4f, filt = 0.f, alpha = 0.5f, disp = 0.f, fcol[3] = {1, 1, 1}, scol[3] = {1, 1, 1}, absorp[3] = {1, 1, 1}, wire = 0.f;
	bool fake = false, recv = true; std::string vis = "normal", node; int add_depth = 0;
	p.get("IOR", ior); p.getColor("filter_color", fcol); p.get("transmit_filter", filt); p.getColor("mirror_color", scol); p.get("alpha", alpha);
	p.get("dispersion_power", disp); p.get("fake_shadows", fake); p.get("receive_shadows", recv); p.get("visibility", vis);
	p.get("additionaldepth", add_depth); p.get("wireframe_amount", wire); p.getColor("absorption", absorp);
	if(disp > 0.f) return fail(yi, "rough_glass: dispersion is not supported by the GPU path (recursiveRaytrace's dispersive branch)");
	if(add_depth < 0 || add_depth > 7) return fail(yi, "rough_glass: additionaldepth outside [0, 7]: the device path keeps at most 7 recursion frames per sample");
	if(wire != 0.f) return fail(yi, "rough_glass: wireframe shading is not supported by the GPU path");
	for(const char *name : {"mirror_color_shader", "bump_shader", "filter_color_shader", "IOR_shader", "wireframe_shader", "roughness_shader"})
		if(p.get(name, node)) return fail(yi, std::string("rough_glass: ") + name + " is not supported by the GPU path (shader nodes on rough glass)");
	std::memset(&m, 0, sizeof m);
	clear_shader_slots(m);
	nodes.clear();
	alpha = std::max(1e-4f, std::min(alpha * 0.5f, 1.f));          // :375
	m.type = YAFGPU_MAT_ROUGH_GLASS; m.receive_shadows = recv; m.visibility = visibility_from(vis);
	m.additional_depth = add_depth;
	m.glass_ior = ior; m.ior_base = ior; m.transp_ior = ior;
	m.rg_a2 = alpha * alpha;
	const float fc = 1.f - filt;                                   // filt * filt_col + Rgb(1.f - filt), floats here (:323)
	for(int k = 0; k < 3; ++k) { m.filter_color[k] = filt * fcol[k] + fc; m.mirror_color[k] = scol[k]; }
	m.fake_shadow = fake;
	m.bsdf_flags = 0x2u | 0x10u | 0x20u;                           // BsdfAllGlossy
	if(fake) m.bsdf_flags |= 0x40u;
	if(absorp[0] < 1.f || absorp[1] < 1.f || absorp[2] < 1.f)
	{	// :389-412: vol_i_ = BeerVolumeHandler(absorption, absorption_dist); volumehandler_beer.cc:28-35
		double dist = 1.0;
		p.get("absorption_dist", dist);
		const float maxlog = (float)std::log(1e38);
		for(int k = 0; k < 3; ++k)
		{
			m.beer_sigma[k] = (absorp[k] > 1e-38) ? (float)-std::log((double)absorp[k]) : maxlog;
			if(dist != 0.f) m.beer_sigma[k] = m.beer_sigma[k] * (float)(1.f / dist);
		}
		m.has_vol_i = 1;
		m.bsdf_flags |= 0x100u;                                   // BsdfVolumetric
	}
	return true;
}
// MirrorMaterial::factory + ctor, material_glass.cc:486-493, material_glass.h:74-79
bool make_mirror(const ParamMap &p, yafgpu_material &m)
{
	float col[3] = {1, 1, 1}, refl = 1.f;
	p.getColor("color", col); p.get("reflect", refl);
	std::memset(&m, 0, sizeof m);
	m.type = YAFGPU_MAT_MIRROR; m.receive_shadows = 1;
	for(int k = 0; k < 3; ++k) m.mirror_color[k] = col[k] * refl;
	m.bsdf_flags = 0x1u;
	return true;
}

// AreaLight::factory + ctor, light_area.cc:169-205, :34-52
bool make_arealight(const ParamMap &p, yafgpu_light &l)
{
	float corner[3] = {0, 0, 0}, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, color[3] = {1, 1, 1};
	float power = 1.f; int samples = 4; bool enabled = true, cast = true;
	p.getPoint("corner", corner); p.getPoint("point1", p1); p.getPoint("point2", p2); p.getColor("color", color);
	p.get("power", power); p.get("samples", samples); p.get("light_enabled", enabled); p.get("cast_shadows", cast);
	std::memset(&l, 0, sizeof l);
	l.type = YAFGPU_LIGHT_AREA; l.samples = samples; l.cast_shadows = cast;
	float tx[3], ty[3];
	for(int k = 0; k < 3; ++k) { l.corner[k] = corner[k]; tx[k] = p1[k] - corner[k]; ty[k] = p2[k] - corner[k]; l.to_x[k] = tx[k]; l.to_y[k] = ty[k]; }
	float fn[3]; cross3(ty, tx, fn);
	for(int k = 0; k < 3; ++k) l.color[k] = (power * color[k]) * (float)kPi;   // col * inte * M_PI
	float vl = fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2];                 // normLen, vector.h:61-71
	if(vl != 0.f) { vl = std::sqrt(vl); const float d = (float)(1.0 / (double)vl); fn[0] *= d; fn[1] *= d; fn[2] *= d; }
	l.area = vl;
	for(int k = 0; k < 3; ++k)
	{
		l.fnormal[k] = fn[k];
		l.c2[k] = corner[k] + tx[k];
		l.c3[k] = corner[k] + (tx[k] + ty[k]);
		l.c4[k] = corner[k] + ty[k];
	}
	return enabled;
}
// PointLight::factory + ctor, light_point.cc:97-120, :28-36
bool make_pointlight(const ParamMap &p, yafgpu_light &l)
{
	float from[3] = {0, 0, 0}, color[3] = {1, 1, 1}; float power = 1.f; bool enabled = true, cast = true;
	p.getPoint("from", from); p.getColor("color", color); p.get("power", power); p.get("light_enabled", enabled); p.get("cast_shadows", cast);
	std::memset(&l, 0, sizeof l);
	l.type = YAFGPU_LIGHT_POINT; l.samples = 1; l.cast_shadows = cast;
	for(int k = 0; k < 3; ++k) { l.position[k] = from[k]; l.color[k] = power * color[k]; }
	return enabled;
}

// PerspectiveCamera::factory, Camera::Camera, setAxis — camera_perspective.cc:198-243, camera.cc:46-66, :60-74
bool make_camera(yafaray_interface *yi, const ParamMap &p, yafgpu_camera &c)
{
	float from[3] = {0, 1, 0}, to[3] = {0, 0, 0}, up[3] = {0, 1, 1};
	int resx = 320, resy = 200; float aspect = 1, dfocal = 1, apt = 0, near_clip = 0.f, far_clip = -1.f;
	p.getPoint("from", from); p.getPoint("to", to); p.getPoint("up", up); p.get("resx", resx); p.get("resy", resy);
	p.get("focal", dfocal); p.get("aperture", apt); p.get("aspect_ratio", aspect); p.get("nearClip", near_clip); p.get("farClip", far_clip);
	float dofd = 0.f, bkhrot = 0.f; std::string bkhtype = "disk1", bkhbias = "uniform";
	p.get("dof_distance", dofd); p.get("bokeh_type", bkhtype); p.get("bokeh_bias", bkhbias); p.get("bokeh_rotation", bkhrot);
	(void)yi;
	const float aspect_ratio = aspect * (float)resy / (float)resx;
	float cy[3], cz[3], cx[3];
	for(int k = 0; k < 3; ++k) { cy[k] = up[k] - from[k]; cz[k] = to[k] - from[k]; }
	cross3(cz, cy, cx);
	cross3(cz, cx, cy);
	normalize3(cx); normalize3(cy); normalize3(cz);
	std::memset(&c, 0, sizeof c);
	c.resx = resx; c.resy = resy;
	for(int k = 0; k < 3; ++k)
	{
		c.position[k] = from[k];
		c.near_n[k] = cz[k]; c.near_p[k] = from[k] + near_clip * cz[k];
		c.far_n[k] = cz[k]; c.far_p[k] = from[k] + far_clip * cz[k];
		const float vright = cx[k], vup = aspect_ratio * cy[k];
		c.vto[k] = (dfocal * cz[k]) - 0.5f * (vup + vright);
		c.vup[k] = vup / (float)resy;
		c.vright[k] = vright / (float)resx;
		c.dof_rt[k] = apt * cx[k]; c.dof_up[k] = apt * cy[k];         // setAxis, :66-67
		c.cam_x[k] = cx[k]; c.cam_y[k] = cy[k]; c.cam_z[k] = cz[k];
	}
	c.focal_distance = dfocal; c.aspect_ratio = aspect_ratio;
	c.aperture = apt; c.dof_distance = dofd; c.bokeh_rotation = bkhrot;
	c.bokeh_type = bkhtype == "disk2" ? 1 : bkhtype == "triangle" ? 3 : bkhtype == "square" ? 4 : bkhtype == "pentagon" ? 5
	             : bkhtype == "hexagon" ? 6 : bkhtype == "ring" ? 7 : 0;
	c.bokeh_bias = bkhbias == "center" ? 1 : bkhbias == "edge" ? 2 : 0;
	return true;
}

void set_param(yafaray_interface *yi, const char *name, const Param &v) { if(name) yi->cparams->dicc[name] = v; }

} // namespace

extern "C" {

yafaray_interface_t *yafaray_createInterface(void) { return new yafaray_interface(); }
void yafaray_destroyInterface(yafaray_interface_t *yi)
{
	if(!yi) return;
	if(yi->gpu) yafgpu_scene_destroy(yi->gpu);
	delete yi;
}
const char *yafaray_getLastError(const yafaray_interface_t *yi) { return yi ? yi->err.c_str() : "null interface"; }
const char *yafaray_getVersion(void) { return "yafgpu-0.1 (MI355X path-tracing core behind the libYafaRay v3 Interface API)"; }

yafaray_bool_t yafaray_startScene(yafaray_interface_t *yi, int type)
{
	if(type != 0) return fail(yi, "startScene: only scene type 0 (\"triangle\") is supported (import_xml.cc:339-347)");
	yi->state = 0; yi->meshes.clear(); yi->cur = nullptr; yi->last = nullptr; yi->geometry_changed = true; yi->prepared = false; yi->scene_dirty = true;
	return 1;
}
yafaray_bool_t yafaray_startGeometry(yafaray_interface_t *yi) { if(yi->state != 0) return fail(yi, "startGeometry: wrong state"); yi->state = 1; return 1; }
yafaray_bool_t yafaray_endGeometry(yafaray_interface_t *yi) { if(yi->state != 1) return fail(yi, "endGeometry: wrong state"); yi->state = 0; return 1; }
unsigned int yafaray_getNextFreeId(yafaray_interface_t *yi) { while(yi->meshes.count(yi->next_id)) ++yi->next_id; return yi->next_id++; }

yafaray_bool_t yafaray_startTriMesh(yafaray_interface_t *yi, unsigned int id, int vertices, int triangles, yafaray_bool_t has_orco, yafaray_bool_t has_uv, int type, int)
{
	if(yi->state != 1) return fail(yi, "startTriMesh: wrong state");
	if((type & 0xFF) != 0) return fail(yi, "startTriMesh: only TRIM meshes (type 0) are supported");
	note_srand(yi, ++g_object_index_auto);          // ObjectGeometric::ObjectGeometric, object_geom.cc:39-43
	Mesh &m = yi->meshes[id];
	m = Mesh();
	m.visible = !(type & 0x0100); m.base = (type & 0x0200) != 0;
	m.has_orco = has_orco != 0; m.has_uv = has_uv != 0;
	m.points.reserve((size_t)std::max(vertices, 0) * 3); m.tri.reserve((size_t)std::max(triangles, 0) * 3); m.tri_mat.reserve((size_t)std::max(triangles, 0));
	yi->cur = &m; yi->last = &m; yi->state = 2; yi->geometry_changed = true; yi->prepared = false; yi->scene_dirty = true;
	return 1;
}
yafaray_bool_t yafaray_endTriMesh(yafaray_interface_t *yi)
{	// Scene::endTriMesh, scene.cc:311-337: "UV-offsets mismatch!" when the triangle and UV-offset counts disagree; the reference never
	// range-checks the offsets themselves (exporters may list <uv> after the faces), so that check waits until here
	if(yi->state != 2) return fail(yi, "endTriMesh: wrong state");
	Mesh &m = *yi->cur;
	if(m.has_uv)
	{
		if(m.tri_uv.size() != m.tri.size()) return fail(yi, "endTriMesh: UV-offsets mismatch!");
		const int nuv = (int)(m.uv.size() / 2);
		for(int o : m.tri_uv) if(o < 0 || o >= nuv) return fail(yi, "endTriMesh: UV index out of range");
	}
	yi->state = 1; yi->cur = nullptr;
	return 1;
}
int yafaray_addVertex(yafaray_interface_t *yi, double x, double y, double z)
{
	if(yi->state != 2) { fail(yi, "addVertex: wrong state"); return -1; }
	Mesh &m = *yi->cur;
	m.points.push_back((float)x); m.points.push_back((float)y); m.points.push_back((float)z);
	return (int)(m.points.size() / 3) - 1;
}
void yafaray_addNormal(yafaray_interface_t *yi, double nx, double ny, double nz)
{	// Scene::addNormal, scene.cc:592-607: attaches to the last vertex
	if(yi->state != 2) { fail(yi, "addNormal: wrong state"); return; }
	Mesh &m = *yi->cur;
	const size_t nv = m.points.size() / 3;
	if(nv == 0) return;
	if(m.normals.size() < nv * 3) m.normals.resize(nv * 3, 0.f);
	m.normals[(nv - 1) * 3] = (float)nx; m.normals[(nv - 1) * 3 + 1] = (float)ny; m.normals[(nv - 1) * 3 + 2] = (float)nz;
	m.normals_exported = true;
}
yafaray_bool_t yafaray_addTriangle(yafaray_interface_t *yi, int a, int b, int c, const yafaray_material_t *mat)
{
	if(yi->state != 2) return fail(yi, "addTriangle: wrong state");
	if(!mat) return fail(yi, "addTriangle: null material");
	Mesh &m = *yi->cur;
	const int nv = (int)(m.points.size() / 3);
	if(a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) return fail(yi, "addTriangle: vertex index out of range");
	m.tri.push_back(a); m.tri.push_back(b); m.tri.push_back(c); m.tri_mat.push_back(mat->index);
	return 1;
}
int yafaray_addVertexWithOrco(yafaray_interface_t *yi, double x, double y, double z, double ox, double oy, double oz)
{	// Interface::addVertex(x, y, z, ox, oy, oz) -> Scene::addVertex(p, orco), scene.cc:567-590
	if(yi->state != 2) { fail(yi, "addVertex: wrong state"); return -1; }
	Mesh &m = *yi->cur;
	if(!m.has_orco) { fail(yi, "addVertex: the mesh was started without orco coordinates"); return -1; }
	m.points.push_back((float)x); m.points.push_back((float)y); m.points.push_back((float)z);
	m.orco.resize(m.points.size(), 0.f);
	const size_t k = m.points.size() - 3;
	m.orco[k] = (float)ox; m.orco[k + 1] = (float)oy; m.orco[k + 2] = (float)oz;
	return (int)(m.points.size() / 3) - 1;
}
int yafaray_addUv(yafaray_interface_t *yi, float u, float v)
{	// Scene::addUv, scene.cc:672-686
	if(yi->state != 2 || !yi->cur) { fail(yi, "addUv: wrong state"); return -1; }
	Mesh &m = *yi->cur;
	m.uv.push_back(u); m.uv.push_back(v);
	return (int)(m.uv.size() / 2) - 1;
}
yafaray_bool_t yafaray_addTriangleWithUv(yafaray_interface_t *yi, int a, int b, int c, int uv_a, int uv_b, int uv_c, const yafaray_material_t *mat)
{	// Interface::addTriangle(a, b, c, uv_a, uv_b, uv_c, mat) -> Scene::addTriangle, scene.cc:652-670
	if(yi->state != 2) return fail(yi, "addTriangle: wrong state");
	Mesh &m = *yi->cur;
	if(!m.has_uv) return fail(yi, "addTriangle: the mesh was started without UV coordinates");
	if(!yafaray_addTriangle(yi, a, b, c, mat)) return 0;
	m.tri_uv.push_back(uv_a); m.tri_uv.push_back(uv_b); m.tri_uv.push_back(uv_c);     // uv_offsets_.push_back x3
	return 1;
}
yafaray_bool_t yafaray_startTriMeshPtr(yafaray_interface_t *yi, unsigned int *id, int vertices, int triangles, yafaray_bool_t has_orco, yafaray_bool_t has_uv, int type, int obj_pass_index)
{	// interface.cc:153-160: the scene picks the id
	if(!id) return fail(yi, "startTriMeshPtr: null id");
	*id = yafaray_getNextFreeId(yi);
	return yafaray_startTriMesh(yi, *id, vertices, triangles, has_orco, has_uv, type, obj_pass_index);
}
// outside the hot path's scope: refused with a diagnostic, never routed anywhere else
yafaray_bool_t yafaray_startCurveMesh(yafaray_interface_t *yi, unsigned int, int, int) { return fail(yi, "startCurveMesh: curve (strand) meshes are outside the GPU path's scope"); }
yafaray_bool_t yafaray_endCurveMesh(yafaray_interface_t *yi, const yafaray_material_t *, float, float, float) { return fail(yi, "endCurveMesh: curve (strand) meshes are outside the GPU path's scope"); }
yafaray_bool_t yafaray_addInstance(yafaray_interface_t *yi, unsigned int, const float *) { return fail(yi, "addInstance: instanced base meshes are outside the GPU path's scope"); }
unsigned int yafaray_createObject(yafaray_interface_t *yi, const char *) { fail(yi, "createObject: parametric objects (sphere) are outside the GPU path's scope (triangle meshes only)"); return 0u; }
void *yafaray_createVolumeRegion(yafaray_interface_t *yi, const char *) { fail(yi, "createVolumeRegion: volume regions are outside the GPU path's scope (volume integrator \"none\" only)"); return nullptr; }
void *yafaray_createImageHandler(yafaray_interface_t *yi, const char *, yafaray_bool_t) { fail(yi, "createImageHandler: image files are written by the caller from the ColorOutput callbacks; no image handlers on the GPU path"); return nullptr; }

yafaray_bool_t yafaray_addTriangles(yafaray_interface_t *yi, int n_verts, const float *verts, int n_tris, const int *indices, const yafaray_material_t *mat)
{
	if(yi->state != 2) return fail(yi, "addTriangles: wrong state");
	if(!mat || !verts || !indices || n_verts < 0 || n_tris < 0) return fail(yi, "addTriangles: bad argument");
	Mesh &m = *yi->cur;
	const int base = (int)(m.points.size() / 3);
	m.points.insert(m.points.end(), verts, verts + (size_t)n_verts * 3);
	for(int i = 0; i < n_tris * 3; ++i)
	{
		if(indices[i] < 0 || indices[i] >= n_verts) return fail(yi, "addTriangles: vertex index out of range");
		m.tri.push_back(base + indices[i]);
	}
	m.tri_mat.insert(m.tri_mat.end(), (size_t)n_tris, mat->index);
	return 1;
}
// Scene::smoothMesh, scene.cc:383-543.  Vertex normals per triangle corner; a corner left without one (all-zero
// triple) uses the geometric normal, as Triangle::getSurface does for a negative normal index (triangle.cc:38-40).
// Parity unpinned: scene.cc does not build outside the reference's own build system; restated, and checked against an
// independent restatement in tests/test_host_api.py.
namespace {
inline void sub3(const float *a, const float *b, float *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
inline float len3(const float *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }   // Vec3::length, vector.h (fSqrt__ = sqrt)
inline float sin_from_vectors(const float *a, const float *b)   // Vec3::sinFromVectors, vector.h:263-270
{
	const float div = (len3(a) * len3(b)) * 0.99999f + 0.00001f;
	float c[3]; cross3(a, b, c);
	float arg = (len3(c) / div) * 0.99999f;
	if(arg > 1.f) arg = 1.f;
	return (float)std::asin((double)arg);
}
inline float host_fsin_poly(float x)   // fSin__ with FAST_TRIG, util_math_optimizations.h:219-244
{
	const double k2Pi = 6.28318530717958647692, kPi = 3.14159265358979323846;
	if((double)x > k2Pi || (double)x < -k2Pi) x -= ((int)(x * (float)0.15915494309189533577)) * (float)k2Pi;
	if((double)x < -kPi) x += (float)k2Pi;
	else if((double)x > kPi) x -= (float)k2Pi;
	x = ((float)1.27323954473516268615 * x) - ((float)0.40528473456935108578 * x * std::fabs(x));
	const float result = 0.225f * (x * std::fabs(x) - x) + x;
	if(result <= -1.0f) return -1.0f;
	if(result >= 1.0f) return 1.0f;
	return result;
}
}
yafaray_bool_t yafaray_smoothMesh(yafaray_interface_t *yi, unsigned int id, double angle_d)
{
	if(yi->state != 1) return fail(yi, "smoothMesh: wrong state (call it between endTriMesh and endGeometry)");
	Mesh *mp = nullptr;
	if(id) { auto it = yi->meshes.find(id); if(it == yi->meshes.end()) return fail(yi, "smoothMesh: no such mesh"); mp = &it->second; }
	else { mp = yi->last; if(!mp) return fail(yi, "smoothMesh: no current mesh"); }
	Mesh &m = *mp;
	const float angle = (float)angle_d;
	const size_t nv = m.points.size() / 3, nt = m.tri.size() / 3;
	yi->geometry_changed = true; yi->prepared = false; yi->scene_dirty = true;
	if(m.normals_exported && m.normals.size() == m.points.size()) { m.smooth = true; return 1; }      // :402-406
	m.smooth_normals.assign(nt * 9, 0.f);
	// face normals: Triangle::recNormal, triangle.h:295-302
	std::vector<float> fn(nt * 3);
	for(size_t t = 0; t < nt; ++t)
	{
		const float *a = &m.points[3 * (size_t)m.tri[3 * t]], *b = &m.points[3 * (size_t)m.tri[3 * t + 1]], *c = &m.points[3 * (size_t)m.tri[3 * t + 2]];
		float e1[3], e2[3]; sub3(b, a, e1); sub3(c, a, e2);
		cross3(e1, e2, &fn[3 * t]);
		normalize3(&fn[3 * t]);
	}
	auto corner_alpha = [&](size_t t, int q, int v1, int v2) {
		const float *pq = &m.points[3 * (size_t)m.tri[3 * t + (size_t)q]], *p1 = &m.points[3 * (size_t)m.tri[3 * t + (size_t)v1]], *p2 = &m.points[3 * (size_t)m.tri[3 * t + (size_t)v2]];
		float e1[3], e2[3]; sub3(p1, pq, e1); sub3(p2, pq, e2);      // PREPARE_EDGES, :383-384
		return sin_from_vectors(e1, e2);
	};
	auto normal_normalize = [](float *v) {   // Normal::normalize, vector.h:272-283
		float len = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
		if(len != 0.f) { len = (float)(1.0 / (double)std::sqrt(len)); v[0] *= len; v[1] *= len; v[2] *= len; }
	};
	if(angle >= 180.f)
	{	// :420-448
		std::vector<float> vn(nv * 3, 0.f);
		for(size_t t = 0; t < nt; ++t)
		{
			const int order[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
			for(int k = 0; k < 3; ++k)
			{
				const float alpha = corner_alpha(t, order[k][0], order[k][1], order[k][2]);
				float *dst = &vn[3 * (size_t)m.tri[3 * t + (size_t)k]];
				for(int c = 0; c < 3; ++c) dst[c] += fn[3 * t + (size_t)c] * alpha;
			}
		}
		for(size_t v = 0; v < nv; ++v) normal_normalize(&vn[3 * v]);
		for(size_t t = 0; t < nt; ++t)
			for(int k = 0; k < 3; ++k)
				for(int c = 0; c < 3; ++c) m.smooth_normals[9 * t + 3 * (size_t)k + (size_t)c] = vn[3 * (size_t)m.tri[3 * t + (size_t)k] + (size_t)c];
	}
	else if(angle > 0.1f)
	{	// :450-538 angle dependent smoothing
		const float thresh = host_fsin_poly((float)((double)angle * 0.01745329251994329576922) + (float)1.57079632679489661923);   // fCos__(DEG_TO_RAD(angle))
		std::vector<std::vector<uint32_t>> vface(nv);
		std::vector<std::vector<float>> alphas(nv);
		for(size_t t = 0; t < nt; ++t)
		{
			const int order[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
			for(int k = 0; k < 3; ++k)
			{
				const size_t v = (size_t)m.tri[3 * t + (size_t)k];
				alphas[v].push_back(corner_alpha(t, order[k][0], order[k][1], order[k][2]));
				vface[v].push_back((uint32_t)t);
			}
		}
		std::vector<float> vnormals;      // the distinct normals found so far at this vertex
		for(size_t i = 0; i < nv; ++i)
		{
			const std::vector<uint32_t> &tris = vface[i];
			vnormals.clear();
			for(size_t j = 0; j < tris.size(); ++j)
			{
				const uint32_t f = tris[j];
				bool smooth = false;
				float vnorm[3] = {fn[3 * f] * alphas[i][j], fn[3 * f + 1] * alphas[i][j], fn[3 * f + 2] * alphas[i][j]};
				for(size_t k = 0; k < tris.size(); ++k)
				{
					const uint32_t f2 = tris[k];
					if(f2 == f) continue;                                   // Triangle::operator== compares indices (triangle.h:76-79)
					const float *n2 = &fn[3 * f2];
					if((fn[3 * f] * n2[0] + fn[3 * f + 1] * n2[1] + fn[3 * f + 2] * n2[2]) > thresh)
					{
						smooth = true;
						for(int c = 0; c < 3; ++c) vnorm[c] += n2[c] * alphas[i][k];
					}
				}
				if(!smooth) continue;                                       // n_idx = -1: the corner keeps the geometric normal
				normalize3(vnorm);
				const float *use = nullptr;
				for(size_t l = 0; l + 2 < vnormals.size(); l += 3)
					if((double)(vnorm[0] * vnormals[l] + vnorm[1] * vnormals[l + 1] + vnorm[2] * vnormals[l + 2]) > 0.999) { use = &vnormals[l]; break; }
				float chosen[3];
				if(use) { chosen[0] = use[0]; chosen[1] = use[1]; chosen[2] = use[2]; }
				else { chosen[0] = vnorm[0]; chosen[1] = vnorm[1]; chosen[2] = vnorm[2]; vnormals.insert(vnormals.end(), vnorm, vnorm + 3); }
				// :526-528: the first corner of f that is vertex i
				int corner = -1;
				for(int c = 0; c < 3; ++c) if((size_t)m.tri[3 * (size_t)f + (size_t)c] == i) { corner = c; break; }
				if(corner < 0) return fail(yi, "smoothMesh: mesh smoothing error");
				for(int c = 0; c < 3; ++c) m.smooth_normals[9 * (size_t)f + 3 * (size_t)corner + (size_t)c] = chosen[c];
			}
		}
	}
	m.smooth = true;
	return 1;
}
yafaray_bool_t yafaray_getMeshCornerNormals(yafaray_interface_t *yi, unsigned int id, float *out, int n_floats)
{
	auto it = yi->meshes.find(id);
	if(it == yi->meshes.end()) return fail(yi, "getMeshCornerNormals: no such mesh");
	const Mesh &m = it->second;
	if(m.smooth_normals.empty() || (int)m.smooth_normals.size() != n_floats) return fail(yi, "getMeshCornerNormals: the mesh has no smoothed normals of that size");
	std::memcpy(out, m.smooth_normals.data(), m.smooth_normals.size() * sizeof(float));
	return 1;
}

void yafaray_paramsSetPoint(yafaray_interface_t *yi, const char *name, double x, double y, double z) { Param p; p.type = Param::Point; p.v[0] = (float)x; p.v[1] = (float)y; p.v[2] = (float)z; set_param(yi, name, p); }
void yafaray_paramsSetString(yafaray_interface_t *yi, const char *name, const char *s) { Param p; p.type = Param::String; p.s = s ? s : ""; set_param(yi, name, p); }
void yafaray_paramsSetBool(yafaray_interface_t *yi, const char *name, yafaray_bool_t b) { Param p; p.type = Param::Bool; p.b = b != 0; set_param(yi, name, p); }
void yafaray_paramsSetInt(yafaray_interface_t *yi, const char *name, int i) { Param p; p.type = Param::Int; p.i = i; set_param(yi, name, p); }
void yafaray_paramsSetFloat(yafaray_interface_t *yi, const char *name, double f) { Param p; p.type = Param::Float; p.f = f; set_param(yi, name, p); }
void yafaray_paramsSetColor(yafaray_interface_t *yi, const char *name, float r, float g, float b, float a)
{	// interface.cc:247-252: Rgba::linearRgbFromColorSpace(input_color_space_, input_gamma_) (color.h:366-386); alpha stays
	float c[3] = {r, g, b};
	if(yi->input_color_space == 0)
	{	// linearRgbFromSRgb, color.h:352-357
		for(float &v : c) v = (v <= 0.04045f) ? (v / 12.92f) : host_fpow(((v + 0.055f) / 1.055f), 2.4f);
	}
	else if(yi->input_color_space == 1)
	{	// XYZ (D65) -> linear RGB, color.h:337-342,375-381
		static const float m[3][3] = {{3.2406255f, -1.537208f, -0.4986286f}, {-0.9689307f, 1.8757561f, 0.0415175f}, {0.0557101f, -0.2040211f, 1.0569959f}};
		const float o[3] = {c[0], c[1], c[2]};
		for(int k = 0; k < 3; ++k) c[k] = m[k][0] * o[0] + m[k][1] * o[1] + m[k][2] * o[2];
	}
	else if(yi->input_color_space == 3 && yi->input_gamma != 1.f)
		for(float &v : c) v = host_fpow(v, yi->input_gamma);      // gammaAdjust, color.h:101
	Param p; p.type = Param::Color; p.v[0] = c[0]; p.v[1] = c[1]; p.v[2] = c[2]; p.v[3] = a; set_param(yi, name, p);
}
void yafaray_paramsSetColorArray(yafaray_interface_t *yi, const char *name, const float *rgb, yafaray_bool_t with_alpha)
{	// interface.cc:254-259
	if(rgb) yafaray_paramsSetColor(yi, name, rgb[0], rgb[1], rgb[2], with_alpha ? rgb[3] : 1.f);
}
void yafaray_paramsSetMatrix(yafaray_interface_t *yi, const char *name, const float *m16, yafaray_bool_t transpose)
{	// interface.cc:260-289 (paramsSetMatrix / paramsSetMemMatrix, float): 16 values, row major
	if(!m16) return;
	Param p; p.type = Param::Matrix;
	for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) p.m[4 * i + j] = transpose ? m16[4 * j + i] : m16[4 * i + j];
	set_param(yi, name, p);
}
void yafaray_paramsSetMatrixD(yafaray_interface_t *yi, const char *name, const double *m16, yafaray_bool_t transpose)
{	// the double overloads, interface.cc:266-289
	if(!m16) return;
	float f[16];
	for(int k = 0; k < 16; ++k) f[k] = (float)m16[k];
	yafaray_paramsSetMatrix(yi, name, f, transpose);
}
void yafaray_setInputColorSpace(yafaray_interface_t *yi, const char *color_space_string, float gamma_val)
{	// interface.cc:292-301
	const std::string cs = color_space_string ? color_space_string : "";
	if(cs == "sRGB") yi->input_color_space = 0;
	else if(cs == "XYZ") yi->input_color_space = 1;
	else if(cs == "LinearRGB") yi->input_color_space = 2;
	else if(cs == "Raw_Manual_Gamma") yi->input_color_space = 3;
	else yi->input_color_space = 0;
	yi->input_gamma = gamma_val;
}
void yafaray_paramsClearAll(yafaray_interface_t *yi) { yi->params.dicc.clear(); yi->eparams.clear(); yi->cparams = &yi->params; }
void yafaray_paramsStartList(yafaray_interface_t *yi) { yi->eparams.emplace_back(); yi->cparams = &yi->eparams.back(); }
void yafaray_paramsPushList(yafaray_interface_t *yi) { yi->eparams.emplace_back(); yi->cparams = &yi->eparams.back(); }
void yafaray_paramsEndList(yafaray_interface_t *yi) { yi->cparams = &yi->params; }

yafaray_material_t *yafaray_createMaterial(yafaray_interface_t *yi, const char *name)
{
	std::string type;
	if(!name) { fail(yi, "createMaterial: null name"); return nullptr; }
	if(yi->materials.count(name)) { fail(yi, std::string("createMaterial: \"") + name + "\" already defined"); return nullptr; }
	if(!yi->params.get("type", type)) { fail(yi, "createMaterial: type of material not specified"); return nullptr; }
	auto m = std::make_unique<yafaray_material>();
	bool ok;
	if(type == "shinydiffusemat") ok = make_shinydiffuse(yi, yi->params, m->m, m->nodes);
	else if(type == "glossy") ok = make_glossy(yi, yi->params, m->m, m->nodes);
	else if(type == "light_mat") ok = make_lightmat(yi->params, m->m);
	else if(type == "glass") ok = make_glass(yi, yi->params, m->m, m->nodes);
	else if(type == "rough_glass") ok = make_rough_glass(yi, yi->params, m->m, m->nodes);
	else if(type == "coated_glossy") ok = make_coated_glossy(yi, yi->params, m->m, m->nodes);
	else if(type == "mirror") ok = make_mirror(yi->params, m->m);
	else { fail(yi, "createMaterial: material type \"" + type + "\" is outside the GPU path's scope (shinydiffusemat, glossy, coated_glossy, glass, mirror, light_mat)"); return nullptr; }
	if(!ok) return nullptr;
	if(type != "shinydiffusemat" && type != "glossy" && type != "coated_glossy" && type != "glass") clear_shader_slots(m->m);
	note_srand(yi, ++g_material_index_auto);        // Material::Material, material.cc:53-57
	m->index = (int)yi->material_order.size();
	yafaray_material *raw = m.get();
	yi->material_order.push_back(raw);
	yi->materials[name] = std::move(m);
	yi->prepared = false; yi->scene_dirty = true;
	return raw;
}
// ImageTexture::factory's parameters other than the image itself (texture_image.cc:559-563, :657-716)
static bool texture_params(yafaray_interface *yi, const ParamMap &p, yafgpu_texture &t)
{
	std::string intp, clip;
	p.get("interpolate", intp);
	if(intp == "bicubic" || intp == "mipmap_trilinear" || intp == "mipmap_ewa")
		return fail(yi, "createTexture: interpolate \"" + intp + "\" is not supported by the GPU path (none and bilinear are)");
	bool normalmap = false; p.get("normalmap", normalmap);
	t.normalmap = normalmap ? 1 : 0;                 // :563, :705
	t.interpolate = intp == "none" ? 0 : 1;          // bilinear is the default (:575)
	bool rot90 = false, even = false, odd = true, mirror_x = false, mirror_y = false, clamp = false;
	int xrep = 1, yrep = 1; double minx = 0.0, miny = 0.0, maxx = 1.0, maxy = 1.0, cdist = 0.0;
	float intensity = 1.f, contrast = 1.f, saturation = 1.f, hue = 0.f, fr = 1.f, fg = 1.f, fb = 1.f;
	p.get("xrepeat", xrep); p.get("yrepeat", yrep);
	p.get("cropmin_x", minx); p.get("cropmin_y", miny); p.get("cropmax_x", maxx); p.get("cropmax_y", maxy);
	p.get("rot90", rot90); p.get("clipping", clip); p.get("even_tiles", even); p.get("odd_tiles", odd); p.get("checker_dist", cdist);
	p.get("mirror_x", mirror_x); p.get("mirror_y", mirror_y);
	p.get("adj_mult_factor_red", fr); p.get("adj_mult_factor_green", fg); p.get("adj_mult_factor_blue", fb);
	p.get("adj_intensity", intensity); p.get("adj_contrast", contrast); p.get("adj_saturation", saturation); p.get("adj_hue", hue); p.get("adj_clamp", clamp);
	t.xrepeat = xrep; t.yrepeat = yrep; t.rot90 = rot90; t.mirror_x = mirror_x; t.mirror_y = mirror_y;
	t.checker_even = even; t.checker_odd = odd; t.checker_dist = (float)cdist;
	// setCrop (:218-223): float members compared with double literals
	t.cropminx = (float)minx; t.cropmaxx = (float)maxx; t.cropminy = (float)miny; t.cropmaxy = (float)maxy;
	t.cropx = ((double)t.cropminx != 0.0) || ((double)t.cropmaxx != 1.0);
	t.cropy = ((double)t.cropminy != 0.0) || ((double)t.cropmaxy != 1.0);
	// string2Cliptype__ (:533-543): TexClipMode { Extend, Clip, ClipCube, Repeat, Checker }, anything unknown repeats
	t.clip = clip == "extend" ? 0 : clip == "clip" ? 1 : clip == "clipcube" ? 2 : clip == "checker" ? 4 : 3;
	// Texture::setAdjustments, texture.h:142-190
	t.adj_int = intensity; t.adj_con = contrast; t.adj_sat = saturation; t.adj_hue = hue / 60.f; t.adj_clamp = clamp;
	t.adj_r = fr; t.adj_g = fg; t.adj_b = fb;
	t.adj_set = (intensity != 1.f || contrast != 1.f || saturation != 1.f || hue != 0.f || clamp || fr != 1.f || fg != 1.f || fb != 1.f) ? 1 : 0;
	return true;
}
static int color_space_from(const std::string &cs)
{	// texture_image.cc:609-613
	if(cs == "sRGB") return yafimg::kSrgb;
	if(cs == "XYZ") return yafimg::kXyz;
	if(cs == "LinearRGB") return yafimg::kLinearRgb;
	if(cs == "Raw_Manual_Gamma") return yafimg::kRawManualGamma;
	return yafimg::kSrgb;
}
static yafaray_texture_t *register_texture(yafaray_interface *yi, const char *name, std::unique_ptr<yafaray_texture> t)
{
	t->index = (int)yi->texture_order.size();
	yafaray_texture *raw = t.get();
	yi->texture_order.push_back(raw);
	yi->textures[name] = std::move(t);
	yi->prepared = false; yi->scene_dirty = true;
	return raw;
}

// Interface::createTexture (interface.h:84) -> RenderEnvironment::createTexture (environment.cc:203-224) -> ImageTexture::factory
// (texture_image.cc:545-720).  Only type "image" is on the GPU path; the procedural textures are refused.
yafaray_texture_t *yafaray_createTexture(yafaray_interface_t *yi, const char *name)
{
	const ParamMap &p = yi->params;
	std::string type, file, cs = "Raw_Manual_Gamma", opt = "optimized";
	if(!name) { fail(yi, "createTexture: null name"); return nullptr; }
	if(yi->textures.count(name)) { fail(yi, std::string("createTexture: \"") + name + "\" already defined"); return nullptr; }
	if(!p.get("type", type)) { fail(yi, "createTexture: type of texture not specified"); return nullptr; }
	if(type != "image") { fail(yi, "createTexture: texture type \"" + type + "\" is outside the GPU path's scope (image textures only)"); return nullptr; }
	auto t = std::make_unique<yafaray_texture>();
	std::memset(&t->t, 0, sizeof t->t);
	if(!texture_params(yi, p, t->t)) return nullptr;
	double gamma = 1.0; bool gray = false;
	p.get("color_space", cs); p.get("gamma", gamma); p.get("filename", file); p.get("texture_optimization", opt); p.get("img_grayscale", gray);
	if(file.empty()) { fail(yi, "createTexture: required argument filename not found for image texture"); return nullptr; }
	yafimg::Image img;
	img.color_space = color_space_from(cs); img.gamma = (float)gamma; img.grayscale = gray;
	img.optimization = opt == "optimized" ? yafimg::kOptOptimized : (opt == "compressed" ? yafimg::kOptCompressed : yafimg::kOptNone);   // :615-618
	std::string err, path = file;
	{	// the reference opens the name as given (relative to the working directory); the XML loader adds the scene file's directory
		FILE *f = std::fopen(path.c_str(), "rb");
		if(f) std::fclose(f);
		else if(!yi->base_dir.empty() && !file.empty() && file[0] != '/') path = yi->base_dir + "/" + file;
	}
	if(!yafimg::load(path, img, err)) { fail(yi, "createTexture: " + err); return nullptr; }
	t->t.width = img.width; t->t.height = img.height;
	t->t.color_space = img.color_space; t->t.gamma = (float)gamma;      // ImageTexture(ih, interpolation, gamma, color_space) with HDR's forced LinearRgb (:600-606, :631)
	t->texels = std::move(img.texels);
	return register_texture(yi, name, std::move(t));
}

// Not in the reference's Interface: an image texture over texels the caller already holds (what its ImageHandler::getPixel would
// return, row major, RGBA float), with the current ParamMap's ImageTexture parameters.  For embedders with in-memory images and
// for the parity tests, which feed the reference harness's own buffers.
yafaray_texture_t *yafaray_createTextureFromMemory(yafaray_interface_t *yi, const char *name, int width, int height, const float *rgba)
{
	const ParamMap &p = yi->params;
	if(!name || !rgba || width <= 0 || height <= 0) { fail(yi, "createTextureFromMemory: bad arguments"); return nullptr; }
	if(yi->textures.count(name)) { fail(yi, std::string("createTexture: \"") + name + "\" already defined"); return nullptr; }
	auto t = std::make_unique<yafaray_texture>();
	std::memset(&t->t, 0, sizeof t->t);
	if(!texture_params(yi, p, t->t)) return nullptr;
	std::string cs = "Raw_Manual_Gamma"; double gamma = 1.0;
	p.get("color_space", cs); p.get("gamma", gamma);
	t->t.width = width; t->t.height = height; t->t.color_space = color_space_from(cs); t->t.gamma = (float)gamma;
	t->texels.assign(rgba, rgba + (size_t)width * (size_t)height * 4);
	return register_texture(yi, name, std::move(t));
}

// the decoded image behind a texture (test hook for the file decoders): width / height, and up to n_floats of its RGBA texels
yafaray_bool_t yafaray_getTextureImage(yafaray_interface_t *yi, const char *name, int *width, int *height, float *rgba, int n_floats)
{
	auto it = name ? yi->textures.find(name) : yi->textures.end();
	if(it == yi->textures.end()) return fail(yi, "getTextureImage: no such texture");
	if(width) *width = it->second->t.width;
	if(height) *height = it->second->t.height;
	if(rgba && n_floats > 0) std::memcpy(rgba, it->second->texels.data(), sizeof(float) * std::min((size_t)n_floats, it->second->texels.size()));
	return 1;
}
yafaray_light_t *yafaray_createLight(yafaray_interface_t *yi, const char *name)
{
	std::string type;
	if(!name) { fail(yi, "createLight: null name"); return nullptr; }
	if(yi->lights.count(name)) { fail(yi, std::string("createLight: \"") + name + "\" already defined"); return nullptr; }
	if(!yi->params.get("type", type)) { fail(yi, "createLight: type of light not specified"); return nullptr; }
	auto l = std::make_unique<yafaray_light>();
	bool enabled, photon_only = false;
	// light_area.cc:69, light_point.cc:40,60: a photon-only light gives illumSample / illuminate nothing (and still counts among the lights the
	// one-light estimator picks from): a photon-mapping setting, not taken over
	if(yi->params.get("photon_only", photon_only) && photon_only) { fail(yi, "createLight: photon_only lights are outside the GPU path's scope (photon mapping)"); return nullptr; }
	if(type == "arealight") enabled = make_arealight(yi->params, l->l);
	else if(type == "pointlight") enabled = make_pointlight(yi->params, l->l);
	else { fail(yi, "createLight: light type \"" + type + "\" is outside the GPU path's scope (arealight, pointlight)"); return nullptr; }
	yafaray_light *raw = l.get();
	if(enabled) yi->light_order.push_back(raw);   // Scene::addLight only sees enabled lights (environment.cc:230-233)
	yi->lights[name] = std::move(l);
	yi->prepared = false; yi->scene_dirty = true;
	return raw;
}
yafaray_camera_t *yafaray_createCamera(yafaray_interface_t *yi, const char *name)
{
	std::string type;
	if(!name) { fail(yi, "createCamera: null name"); return nullptr; }
	if(!yi->params.get("type", type)) { fail(yi, "createCamera: type of camera not specified"); return nullptr; }
	if(type != "perspective") { fail(yi, "createCamera: camera type \"" + type + "\" is outside the GPU path's scope (perspective)"); return nullptr; }
	auto c = std::make_unique<yafaray_camera>();
	if(!make_camera(yi, yi->params, c->c.cam)) return nullptr;
	yafaray_camera *raw = c.get();
	yi->cameras[name] = std::move(c);
	yi->prepared = false; yi->scene_dirty = true;
	return raw;
}
yafaray_background_t *yafaray_createBackground(yafaray_interface_t *yi, const char *name)
{
	std::string type;
	if(!name) { fail(yi, "createBackground: null name"); return nullptr; }
	if(!yi->params.get("type", type)) { fail(yi, "createBackground: type of background not specified"); return nullptr; }
	if(type != "constant") { fail(yi, "createBackground: background type \"" + type + "\" is outside the GPU path's scope (constant)"); return nullptr; }
	bool ibl = false; yi->params.get("ibl", ibl);
	if(ibl) { fail(yi, "createBackground: image-based lighting (ibl) is not supported by the GPU path"); return nullptr; }
	float col[3] = {0, 0, 0}; float power = 1.f;
	yi->params.getColor("color", col); yi->params.get("power", power);   // background_constant.cc:49-66
	auto b = std::make_unique<yafaray_background>();
	for(int k = 0; k < 3; ++k) b->b.color[k] = power * col[k];
	yafaray_background *raw = b.get();
	yi->backgrounds[name] = std::move(b);
	yi->prepared = false;
	return raw;
}
yafaray_integrator_t *yafaray_createIntegrator(yafaray_interface_t *yi, const char *name)
{
	std::string type;
	if(!name) { fail(yi, "createIntegrator: null name"); return nullptr; }
	if(!yi->params.get("type", type)) { fail(yi, "createIntegrator: type of integrator not specified"); return nullptr; }
	auto it = std::make_unique<yafaray_integrator>();
	IntegratorCfg &c = it->c;
	c.type = type;
	const ParamMap &p = yi->params;
	if(type == "pathtracing" || type == "directlighting")
	{
		bool transp_shad = false, do_ao = false, caustics = false;
		std::string c_method;
		p.get("raydepth", c.raydepth); p.get("shadowDepth", c.shadow_depth); p.get("transpShad", transp_shad); p.get("do_AO", do_ao);
		p.get("bg_transp", c.bg_transp); p.get("bg_transp_refract", c.bg_transp_refract);
		c.transp_shad = transp_shad;     // checked against the materials at render time (TriKdTree::intersectTs, row K3)
		if(do_ao) { fail(yi, "createIntegrator: ambient occlusion is not supported by the GPU path"); return nullptr; }
		if(type == "pathtracing")
		{	// PathIntegrator::factory, integrator_path_tracer.cc:349-422
			p.get("path_samples", c.path_samples); p.get("bounces", c.bounces);
			p.get("russian_roulette_min_bounces", c.rr_min_bounces); p.get("no_recursive", c.no_recursive);
			if(p.get("caustic_type", c_method) && (c_method == "photon" || c_method == "both"))
			{ fail(yi, "createIntegrator: photon caustics are not supported by the GPU path (use caustic_type=path or none)"); return nullptr; }
			c.trace_caustics = c_method != "none";      // absent or any other word: the constructor's Path stays (:36), preprocess() sets trace_caustics_ (:85)
		}
		else
		{
			p.get("caustics", caustics);
			if(caustics) { fail(yi, "createIntegrator: photon caustics are not supported by the GPU path"); return nullptr; }
		}
	}
	else if(type != "none")
	{
		fail(yi, "createIntegrator: integrator type \"" + type + "\" is outside the GPU path's scope (pathtracing, directlighting, none)");
		return nullptr;
	}
	yafaray_integrator *raw = it.get();
	yi->integrators[name] = std::move(it);
	yi->prepared = false; yi->scene_dirty = true;
	return raw;
}

void yafaray_clearAll(yafaray_interface_t *yi)
{
	if(yi->gpu) { yafgpu_scene_destroy(yi->gpu); yi->gpu = nullptr; }
	yi->materials.clear(); yi->material_order.clear(); yi->textures.clear(); yi->texture_order.clear(); yi->lights.clear(); yi->light_order.clear();
	yi->cameras.clear(); yi->backgrounds.clear(); yi->integrators.clear(); yi->meshes.clear();
	yi->params.dicc.clear(); yi->eparams.clear(); yi->cparams = &yi->params;
	yi->state = -1; yi->prepared = false; yi->scene_dirty = true; yi->geometry_changed = true; yi->film.clear();
}

void yafaray_getRandState(yafaray_interface_t *yi, int *srand_seed, int *skip)
{
	if(yi->user_srand >= 0) { if(srand_seed) *srand_seed = yi->user_srand; if(skip) *skip = yi->user_skip; return; }
	if(srand_seed) *srand_seed = yi->have_srand ? (int)(yi->last_srand & 0x7fffffffu) : -1;
	if(skip) *skip = yi->have_srand ? colour_loop_draws(yi->last_srand) : 0;
}
void yafaray_setRandState(yafaray_interface_t *yi, int srand_seed, int skip)
{
	yi->user_srand = srand_seed < 0 ? -1 : srand_seed;
	yi->user_skip = (srand_seed < 0 || skip < 0) ? 0 : skip;
	yi->prepared = false;
}
void yafaray_setSerialReplay(yafaray_interface_t *yi, yafaray_bool_t on) { yi->serial_replay = on != 0; yi->prepared = false; }
void yafaray_setPlaneExchange(yafaray_interface_t *yi, yafaray_plane_exchange_t fn, void *user)
{
	yi->exchange = fn; yi->exchange_user = user;
	if(yi->gpu) yafgpu_scene_set_exchange(yi->gpu, fn, user);
}
void yafaray_setComm(yafaray_interface_t *yi, yafaray_comm_t *comm)
{
	if(comm) yafaray_setPlaneExchange(yi, yafaray_commExchange, comm);
	else yafaray_setPlaneExchange(yi, nullptr, nullptr);
}
void yafaray_setShard(yafaray_interface_t *yi, int shard_index, int shard_count) { yi->shard_index = shard_index; yi->shard_count = std::max(1, shard_count); }

// RenderEnvironment::setupScene (environment.cc:679-813) + createImageFilm (:456-584) + Scene::update (scene.cc:784-894)
yafaray_bool_t yafaray_prepareRender(yafaray_interface_t *yi)
{
	const ParamMap &p = yi->params;
	std::string name;
	if(!p.get("camera_name", name)) return fail(yi, "Specify a Camera!!");
	auto cam = yi->cameras.find(name);
	if(cam == yi->cameras.end()) return fail(yi, "Specify an _existing_ Camera!!");
	if(!p.get("integrator_name", name)) return fail(yi, "Specify an Integrator!!");
	auto inte = yi->integrators.find(name);
	if(inte == yi->integrators.end()) return fail(yi, "Specify an _existing_ Integrator!!");
	if(inte->second->c.type == "none") return fail(yi, "Integrator is no surface integrator!");
	if(!p.get("volintegrator_name", name)) return fail(yi, "Specify a Volume Integrator!");
	auto vol = yi->integrators.find(name);
	if(vol == yi->integrators.end() || vol->second->c.type != "none") return fail(yi, "volume integrators other than type \"none\" are not supported by the GPU path");
	const BackgroundCfg *bg = nullptr;
	if(p.get("background_name", name))
	{
		auto b = yi->backgrounds.find(name);
		if(b == yi->backgrounds.end()) return fail(yi, "please specify an _existing_ Background!!");
		bg = &b->second->b;
	}
	int aa_passes = 1, aa_samples = 1, tile_size = 32, width = 320, height = 240, xstart = 0, ystart = 0;
	int base_offset = 0, node = 0;
	float filt_sz = 1.5f, shadow_bias = (float)0.0005, min_raydist = (float)0.00005, clamp_samples = 0.f, clamp_indirect = 0.f;
	bool auto_bias = true, auto_dist = true, premult = false;
	std::string filter = "box";
	p.get("AA_passes", aa_passes); p.get("AA_minsamples", aa_samples); p.get("AA_pixelwidth", filt_sz);
	p.get("width", width); p.get("height", height); p.get("xstart", xstart); p.get("ystart", ystart);
	p.get("filter_type", filter); p.get("tile_size", tile_size); p.get("premult", premult);
	p.get("AA_clamp_samples", clamp_samples); p.get("AA_clamp_indirect", clamp_indirect);
	p.get("adv_auto_shadow_bias_enabled", auto_bias); p.get("adv_shadow_bias_value", shadow_bias);
	p.get("adv_auto_min_raydist_enabled", auto_dist); p.get("adv_min_raydist_value", min_raydist);
	p.get("adv_base_sampling_offset", base_offset); p.get("adv_computer_node", node);
	p.get("color_space", yi->color_space); p.get("gamma", yi->gamma);
	yi->color_space2 = "Raw_Manual_Gamma"; yi->gamma2 = 1.f;
	p.get("color_space2", yi->color_space2); p.get("gamma2", yi->gamma2);
	// Scene::setAntialiasing (scene.cc:761-778), defaults of RenderEnvironment::setupScene (environment.cc:682-695,747-762)
	{
		yafgpu_aa_schedule &aa = yi->aa;
		aa = yafgpu_aa_schedule{};
		int inc = aa_samples, var_edge = 10, var_pix = 0; double threshold = 0.05;
		float floor_pct = 0.f, smf = 1.f, lmf = 1.f, dark_factor = 0.f; bool color_noise = false; std::string dark = "none";
		p.get("AA_inc_samples", inc); p.get("AA_threshold", threshold); p.get("AA_resampled_floor", floor_pct);
		p.get("AA_sample_multiplier_factor", smf); p.get("AA_light_sample_multiplier_factor", lmf);
		p.get("AA_detect_color_noise", color_noise); p.get("AA_dark_detection_type", dark); p.get("AA_dark_threshold_factor", dark_factor);
		p.get("AA_variance_edge_size", var_edge); p.get("AA_variance_pixels", var_pix);
		aa.passes = aa_passes; aa.inc_samples = inc; aa.threshold = (float)threshold; aa.resampled_floor = floor_pct;
		aa.sample_multiplier_factor = smf; aa.light_sample_multiplier_factor = lmf; aa.detect_color_noise = color_noise ? 1 : 0;
		aa.dark_detection_type = dark == "linear" ? 1 : (dark == "curve" ? 2 : 0);
		aa.dark_threshold_factor = dark_factor; aa.variance_edge_size = var_edge; aa.variance_pixels = var_pix;
		if(aa_passes < 1) return fail(yi, "render: AA_passes must be at least 1");
		aa.rand_srand = -1; aa.rand_skip = 0;
		if(yi->have_srand) { aa.rand_srand = (int32_t)(yi->last_srand & 0x7fffffffu); aa.rand_skip = colour_loop_draws(yi->last_srand); }
		if(yi->user_srand >= 0) { aa.rand_srand = yi->user_srand; aa.rand_skip = yi->user_skip; }      // yafaray_setRandState
	}
	int filter_type = YAFGPU_FILTER_BOX;      // RenderEnvironment::createImageFilm, environment.cc:537-541: unknown names default to box
	if(filter == "mitchell") filter_type = YAFGPU_FILTER_MITCHELL;
	else if(filter == "gauss") filter_type = YAFGPU_FILTER_GAUSS;
	else if(filter == "lanczos") filter_type = YAFGPU_FILTER_LANCZOS;
	if(premult) return fail(yi, "render: premultiplied alpha is not supported by the GPU path");
	(void)clamp_indirect;   // only clamps caustic-photon estimates (integrator_path_tracer.cc:160-165), which this path does not have
	const IntegratorCfg &ic = inte->second->c;
	yafgpu_render_params &rp = yi->rp;
	std::memset(&rp, 0, sizeof rp);
	rp.integrator = ic.type == "pathtracing" ? YAFGPU_INTEGRATOR_PATH : YAFGPU_INTEGRATOR_DIRECT;
	rp.path_samples = ic.path_samples; rp.bounces = ic.bounces; rp.rr_min_bounces = ic.rr_min_bounces;
	rp.no_recursive = ic.no_recursive; rp.bg_transp = ic.bg_transp; rp.bg_transp_refract = ic.bg_transp_refract;
	rp.trace_caustics = (ic.type == "pathtracing" && ic.trace_caustics) ? 1 : 0;
	rp.width = width; rp.height = height; rp.xstart = xstart; rp.ystart = ystart;
	rp.aa_minsamples = aa_samples; rp.aa_pixelwidth = filt_sz; rp.filter_type = filter_type; rp.tile_size = tile_size;
	rp.base_sampling_offset = (uint32_t)base_offset + (uint32_t)node * 100000u;   // imagefilm.h:124
	rp.shadow_bias_auto = auto_bias; rp.shadow_bias = shadow_bias; rp.min_raydist_auto = auto_dist; rp.min_raydist = min_raydist;
	rp.aa_light_sample_multiplier = 1.f;
	rp.aa_clamp_samples = clamp_samples;
	rp.raydepth = ic.raydepth;
	rp.transp_shad = ic.transp_shad ? 1 : 0; rp.shadow_depth = ic.shadow_depth;
	if(bg) { rp.has_background = 1; for(int k = 0; k < 3; ++k) rp.background[k] = bg->color[k]; }
	rp.shard_index = yi->shard_index; rp.shard_count = yi->shard_count;
	rp.serial_replay = yi->serial_replay ? 1 : 0;
	{	// the first pass's rand() per tile, for callers that run single passes themselves (yafaray_renderPassDevice)
		const int nt = ((width + tile_size - 1) / std::max(tile_size, 1)) * ((height + tile_size - 1) / std::max(tile_size, 1));
		yi->tile_rand0.assign((size_t)std::max(nt, 0), 0);
		if(yi->aa.rand_srand >= 0 && nt > 0)
		{
			std::vector<int32_t> v((size_t)yi->aa.rand_skip + (size_t)nt);
			yafgpu_glibc_rand((uint32_t)yi->aa.rand_srand, (int32_t)v.size(), v.data());
			std::copy(v.begin() + yi->aa.rand_skip, v.end(), yi->tile_rand0.begin());
		}
	}

	// Scene::update: flatten visible non-base meshes in object-id order (scene.cc:797-817)
	if(yi->state != 0) return fail(yi, "render: scene is not in the ready state (missing endGeometry?)");
	int threads = -1; p.get("threads", threads);
	if(yi->gpu && !yi->scene_dirty && std::memcmp(&yi->scene_cam, &cam->second->c.cam, sizeof yi->scene_cam) == 0 && yi->scene_threads == threads)
	{	// nothing the device scene is made of changed: keep it, tree and all (Scene::update rebuilds only on changes, scene.cc:784-790)
		yafgpu_scene_set_exchange(yi->gpu, yi->exchange, yi->exchange_user);
		yi->prepared = true;
		return 1;
	}
	if(yi->gpu) { yafgpu_scene_destroy(yi->gpu); yi->gpu = nullptr; }
	std::vector<float> verts; std::vector<int32_t> tri_mat; std::vector<float> vnormals; bool any_normals = false;
	for(auto &kv : yi->meshes) if(kv.second.normals_exported || (kv.second.smooth && !kv.second.smooth_normals.empty())) any_normals = true;
	// texture coordinates, only when some material evaluates shader nodes: per triangle corner UVs and orcos (yafgpu_scene_desc)
	bool any_nodes = false;
	for(auto *m : yi->material_order) if(!m->nodes.empty()) any_nodes = true;
	std::vector<float> tri_uv, tri_orco;
	const float kNoOrco = std::numeric_limits<float>::quiet_NaN();
	for(auto &kv : yi->meshes)
	{
		const Mesh &m = kv.second;
		if(!m.visible || m.base) continue;
		const size_t nt = m.tri_mat.size();
		for(size_t t = 0; t < nt; ++t)
		{
			for(int c = 0; c < 3; ++c)
			{
				const int vi = m.tri[3 * t + (size_t)c];
				verts.push_back(m.points[3 * (size_t)vi]); verts.push_back(m.points[3 * (size_t)vi + 1]); verts.push_back(m.points[3 * (size_t)vi + 2]);
				if(any_normals)
				{
					if(m.smooth && !m.smooth_normals.empty())
					{ for(int q = 0; q < 3; ++q) vnormals.push_back(m.smooth_normals[9 * t + 3 * (size_t)c + (size_t)q]); }
					else if(m.normals_exported && m.normals.size() >= 3 * ((size_t)vi + 1))
					{ vnormals.push_back(m.normals[3 * (size_t)vi]); vnormals.push_back(m.normals[3 * (size_t)vi + 1]); vnormals.push_back(m.normals[3 * (size_t)vi + 2]); }
					else { vnormals.push_back(0.f); vnormals.push_back(0.f); vnormals.push_back(0.f); }
				}
			}
			tri_mat.push_back(m.tri_mat[t]);
			if(any_nodes)
			{
				for(int c = 0; c < 3; ++c)
				{
					const size_t vi = (size_t)m.tri[3 * t + (size_t)c];
					if(m.has_uv && m.tri_uv.size() >= 3 * (t + 1))
					{ const size_t ui = (size_t)m.tri_uv[3 * t + (size_t)c]; tri_uv.push_back(m.uv[2 * ui]); tri_uv.push_back(m.uv[2 * ui + 1]); }
					else { tri_uv.push_back(c == 0 ? kNoOrco : 0.f); tri_uv.push_back(0.f); }            // has_uv_ false (NaN marker): sp.u_ = sp.v_ = 0, implicit dPdU / dPdV, triangle.cc:103-111
					if(m.has_orco && m.orco.size() >= 3 * (vi + 1))
					{ tri_orco.push_back(m.orco[3 * vi]); tri_orco.push_back(m.orco[3 * vi + 1]); tri_orco.push_back(m.orco[3 * vi + 2]); }
					else { tri_orco.push_back(c == 0 ? kNoOrco : 0.f); tri_orco.push_back(0.f); tri_orco.push_back(0.f); }   // has_orco_ false: orco = the hit point
				}
			}
		}
	}
	if(yi->material_order.empty()) return fail(yi, "render: no materials defined");
	std::vector<yafgpu_material> mats; std::vector<yafgpu_node> nodes;
	for(auto *m : yi->material_order)
	{
		yafgpu_material rec = m->m;
		rec.node_first = (int32_t)nodes.size();           // rec.n_nodes colour nodes, then the bump shader's list
		rec.bump_first += rec.node_first;
		nodes.insert(nodes.end(), m->nodes.begin(), m->nodes.end());
		mats.push_back(rec);
	}
	std::vector<yafgpu_texture> textures; std::vector<float> texels;
	if(any_nodes)
		for(auto *t : yi->texture_order)
		{
			if((texels.size() + t->texels.size()) / 4 > 0xffffffffull) return fail(yi, "render: more than 2^32 texels in the scene's image textures");
			yafgpu_texture rec = t->t;
			rec.texel_first = (uint32_t)(texels.size() / 4);
			texels.insert(texels.end(), t->texels.begin(), t->texels.end());
			textures.push_back(rec);
		}
	std::vector<yafgpu_light> lights; for(auto *l : yi->light_order) lights.push_back(l->l);
	yafgpu_scene_desc d{};
	d.n_tris = (int32_t)tri_mat.size(); d.verts = verts.data(); d.tri_mat = tri_mat.data();
	d.vnormals = any_normals ? vnormals.data() : nullptr;
	d.n_materials = (int32_t)mats.size(); d.materials = mats.data();
	d.n_lights = (int32_t)lights.size(); d.lights = lights.data();
	if(any_nodes)
	{
		d.tri_uv = tri_uv.data(); d.tri_orco = tri_orco.data();
		d.n_textures = (int32_t)textures.size(); d.textures = textures.data();
		d.n_texels = texels.size() / 4; d.texels = texels.data();
		d.n_nodes = (int32_t)nodes.size(); d.nodes = nodes.data();
	}
	d.camera = cam->second->c.cam;
	d.build_threads = 0;
	if(threads > 0) d.build_threads = threads;
	if(yafgpu_scene_create(&d, &yi->gpu)) return fail(yi, std::string("scene upload: ") + yafgpu_last_error());
	yi->scene_dirty = false; yi->scene_cam = cam->second->c.cam; yi->scene_threads = threads;
	yafgpu_scene_set_pass_pipelining(yi->gpu, yi->pass_pipelining);
	yafgpu_scene_set_abort_flag(yi->gpu, &yi->abort_flag);
	yafgpu_scene_set_exchange(yi->gpu, yi->exchange, yi->exchange_user);
	yafgpu_tree_info ti{};
	yafgpu_scene_info(yi->gpu, &ti);
	yi->stats = yafaray_render_stats_t{};
	yi->stats.tree_build_seconds = ti.build_seconds; yi->stats.upload_seconds = ti.upload_seconds;
	yi->stats.kd_nodes = ti.n_nodes; yi->stats.kd_leaf_refs = ti.n_leaf_refs; yi->stats.kd_max_depth = ti.max_depth; yi->stats.n_triangles = ti.n_tris;
	yi->stats.scene_device_bytes = ti.device_bytes;
	yi->prepared = true;
	return 1;
}

yafaray_bool_t yafaray_getRenderSize(yafaray_interface_t *yi, int *width, int *height)
{
	if(!yi->prepared) return fail(yi, "getRenderSize: call prepareRender first");
	if(width) *width = yi->rp.width;
	if(height) *height = yi->rp.height;
	return 1;
}

yafaray_bool_t yafaray_renderPassDevice(yafaray_interface_t *yi, float *d_planes, void *d_counters, void *stream)
{
	if(!yi->prepared) return fail(yi, "renderPassDevice: call prepareRender first");
	yi->rp.shard_index = yi->shard_index; yi->rp.shard_count = yi->shard_count;
	yi->rp.tile_rand = yi->tile_rand0.empty() ? nullptr : yi->tile_rand0.data();
	if(yafgpu_render_tiles(yi->gpu, &yi->rp, d_planes, (yafgpu_counters *)d_counters, stream)) return fail(yi, std::string("render: ") + yafgpu_last_error());
	return 1;
}

yafaray_bool_t yafaray_intersectRays(yafaray_interface_t *yi, int n, const float *rays, int *tri, float *t, float *bary)
{
	if(!yi->prepared) return fail(yi, "intersectRays: call prepareRender first");
	if(yafgpu_trace_closest(yi->gpu, n, rays, tri, t, bary)) return fail(yi, std::string("intersectRays: ") + yafgpu_last_error());
	return 1;
}
yafaray_bool_t yafaray_shadowRays(yafaray_interface_t *yi, int n, const float *rays, int *shadowed)
{
	if(!yi->prepared) return fail(yi, "shadowRays: call prepareRender first");
	if(yafgpu_trace_shadow(yi->gpu, n, rays, shadowed)) return fail(yi, std::string("shadowRays: ") + yafgpu_last_error());
	return 1;
}

yafaray_bool_t yafaray_setPassPipelining(yafaray_interface_t *yi, int mode)
{
	yi->pass_pipelining = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
	if(yi->gpu) yafgpu_scene_set_pass_pipelining(yi->gpu, yi->pass_pipelining);
	return 1;
}
yafaray_bool_t yafaray_setProfiling(yafaray_interface_t *yi, yafaray_bool_t enable)
{
	if(!yi->prepared) return fail(yi, "setProfiling: call prepareRender first");
	return yafgpu_set_profiling(yi->gpu, enable) == 0;
}
yafaray_bool_t yafaray_getKernelProfile(yafaray_interface_t *yi, double ms[4], uint64_t launches[4])
{
	if(!yi->prepared) return fail(yi, "getKernelProfile: call prepareRender first");
	return yafgpu_get_profile(yi->gpu, ms, launches) == 0;
}

yafaray_bool_t yafaray_probe(yafaray_interface_t *yi, int op, int n, const float *in, int n_in, float *out, int n_out)
{
	if(!yi->prepared) return fail(yi, "probe: call prepareRender first");
	if(yafgpu_probe(yi->gpu, op, n, in, n_in, out, n_out)) return fail(yi, std::string("probe: ") + yafgpu_last_error());
	return 1;
}

// ImageFilm::flush (imagefilm.cc:737-772), combined pass: Pixel::normalized -> clampRgb0 -> Rgb::colorSpaceFromLinearRgb
// (color.h:388-411: sRGB via the polynomial fPow__, XYZ D65 by matrix, RawManualGamma by gammaAdjust(1 / gamma)) -> alpha
// clamped to [0, 1]
static void to_output_space(const std::string &cs, float gamma, float c[4])
{
	if(cs == "sRGB") { for(int k = 0; k < 3; ++k) c[k] = (c[k] <= 0.0031308f) ? (c[k] * 12.92f) : ((1.055f * host_fpow(c[k], 0.416667f)) - 0.055f); }
	else if(cs == "XYZ")
	{
		static const float m[3][3] = {{0.412400f, 0.357600f, 0.180500f}, {0.212600f, 0.715200f, 0.072200f}, {0.019300f, 0.119200f, 0.950500f}};
		const float o[3] = {c[0], c[1], c[2]};
		for(int k = 0; k < 3; ++k) c[k] = m[k][0] * o[0] + m[k][1] * o[1] + m[k][2] * o[2];
	}
	else if(cs == "Raw_Manual_Gamma" && gamma != 1.f)
	{
		if(gamma <= 0.f) gamma = 1.0e-2f;
		const float inv = 1.f / gamma;
		for(int k = 0; k < 3; ++k) c[k] = host_fpow(c[k], inv);
	}
	if(c[3] < 0.f) c[3] = 0.f; else if(c[3] > 1.f) c[3] = 1.f;
}
static void deliver_to(yafaray_interface_t *yi, const yafaray_output_t *out, const std::string &cs, float gamma)
{
	if(!out) return;
	const int w = yi->rp.width, h = yi->rp.height;
	if(out->putPixel)
	{
		for(int y = 0; y < h; ++y)
			for(int x = 0; x < w; ++x)
			{
				const float *px = &yi->film[5 * ((size_t)y * (size_t)w + (size_t)x)];
				float c[4] = {0, 0, 0, 0};
				if(px[4] != 0.f) { const float f = (float)(1.0 / (double)px[4]); for(int k = 0; k < 4; ++k) c[k] = px[k] * f; }   // Pixel::normalized, Rgba / float (color.h:310-314)
				for(int k = 0; k < 3; ++k) c[k] = std::max(0.f, c[k]);              // clampRgb0
				to_output_space(cs, gamma, c);
				out->putPixel(out->user, 0, x, y, c[0], c[1], c[2], c[3]);
			}
	}
	if(out->flush) out->flush(out->user, 0);
}
static void deliver(yafaray_interface_t *yi, const yafaray_output_t *out)
{
	deliver_to(yi, out, yi->color_space, yi->gamma);
	if(yi->has_output2) deliver_to(yi, &yi->output2, yi->color_space2, yi->gamma2);      // Interface::setOutput2, interface.cc:420-423
}

// ---- the rest of Interface's surface (interface.h:62-128): honoured where it touches this path, accepted where it only
// concerns logging / image decoration, refused with a diagnostic where it asks for something outside the scope
yafaray_bool_t yafaray_setLoggingAndBadgeSettings(yafaray_interface_t *yi)
{	// interface.cc:131-135 -> RenderEnvironment::setupLoggingAndBadge: log files and the parameters badge drawn onto saved
	// images — decoration of the output, not of the film; the settings are read and dropped
	std::string pos;
	if(yi->params.get("logging_paramsBadgePosition", pos)) yi->badge_position = pos;
	return 1;
}
yafaray_bool_t yafaray_setupRenderPasses(yafaray_interface_t *yi)
{	// interface.cc:137-141 -> RenderEnvironment::setupRenderPasses (environment.cc:598-677): with pass_enable the external
	// passes named by pass_* are rendered beside the combined one.  The GPU path produces the combined pass only.
	bool enable = false;
	yi->params.get("pass_enable", enable);
	if(!enable) return 1;
	for(const auto &kv : yi->params.dicc)
		if(kv.first.compare(0, 5, "pass_") == 0 && kv.second.type == Param::String && kv.second.s != "disabled" && kv.second.s != "combined")
			return fail(yi, "setupRenderPasses: render pass \"" + kv.first + "\" = \"" + kv.second.s + "\": the GPU path renders the combined pass only");
	return 1;
}
yafaray_bool_t yafaray_setInteractive(yafaray_interface_t *yi, yafaray_bool_t interactive) { yi->interactive = interactive != 0; return 1; }   // interface.cc:143-151
void yafaray_setConsoleVerbosityLevel(yafaray_interface_t *, const char *) {}      // interface.cc:374-377: console log level
void yafaray_setLogVerbosityLevel(yafaray_interface_t *, const char *) {}          // :379-382
void yafaray_setParamsBadgePosition(yafaray_interface_t *yi, const char *badge_position) { yi->badge_position = badge_position ? badge_position : "none"; }   // :384-387
yafaray_bool_t yafaray_getDrawParams(yafaray_interface_t *) { return 0; }           // :389-396: no badge is drawn
static void print_line(const char *level, const char *msg) { std::fprintf(stderr, "%s: %s\n", level, msg ? msg : ""); }
void yafaray_printDebug(yafaray_interface_t *, const char *msg) { if(std::getenv("YAFGPU_VERBOSE")) print_line("DEBUG", msg); }     // :425-449
void yafaray_printVerbose(yafaray_interface_t *, const char *msg) { if(std::getenv("YAFGPU_VERBOSE")) print_line("VERB", msg); }
void yafaray_printInfo(yafaray_interface_t *, const char *msg) { if(std::getenv("YAFGPU_VERBOSE")) print_line("INFO", msg); }
void yafaray_printParams(yafaray_interface_t *, const char *msg) { if(std::getenv("YAFGPU_VERBOSE")) print_line("PARM", msg); }
void yafaray_printWarning(yafaray_interface_t *, const char *msg) { print_line("WARNING", msg); }
void yafaray_printError(yafaray_interface_t *, const char *msg) { print_line("ERROR", msg); }
void yafaray_setOutput2(yafaray_interface_t *yi, const yafaray_output_t *out_2)
{	// interface.cc:420-423: a second ColorOutput, fed in colour space `color_space2` / `gamma2`
	yi->has_output2 = out_2 != nullptr;
	if(out_2) yi->output2 = *out_2;
}
int yafaray_getRenderParameters(yafaray_interface_t *yi, char *buf, int len)
{	// Interface::getRenderParameters (interface.h:111) hands out the ParamMap; across a C ABI: "name=value" lines.
	// Returns the number of bytes needed (excluding the terminator); writes at most len - 1 of them.
	std::string out;
	for(const auto &kv : yi->params.dicc)
	{
		out += kv.first + "=";
		const Param &q = kv.second;
		char tmp[160];
		switch(q.type)
		{
			case Param::Int: std::snprintf(tmp, sizeof tmp, "%d", q.i); out += tmp; break;
			case Param::Bool: out += q.b ? "true" : "false"; break;
			case Param::Float: std::snprintf(tmp, sizeof tmp, "%.9g", q.f); out += tmp; break;
			case Param::String: out += q.s; break;
			case Param::Point: std::snprintf(tmp, sizeof tmp, "%.9g %.9g %.9g", q.v[0], q.v[1], q.v[2]); out += tmp; break;
			case Param::Color: std::snprintf(tmp, sizeof tmp, "%.9g %.9g %.9g %.9g", q.v[0], q.v[1], q.v[2], q.v[3]); out += tmp; break;
			case Param::Matrix: for(int k = 0; k < 16; ++k) { std::snprintf(tmp, sizeof tmp, k ? " %.9g" : "%.9g", q.m[k]); out += tmp; } break;
			default: break;
		}
		out += "\n";
	}
	if(buf && len > 0) { const size_t n = std::min(out.size(), (size_t)len - 1); std::memcpy(buf, out.data(), n); buf[n] = 0; }
	return (int)out.size();
}

yafaray_bool_t yafaray_render(yafaray_interface_t *yi, const yafaray_output_t *output, const yafaray_progress_t *progress)
{
	// an abort that arrived before this call belongs to an earlier render (Scene::render clears the flag too, scene.cc:1040)
	yi->abort_flag = 0;
	if(progress && progress->init) progress->init(progress->user, 100);
	if(progress && progress->setTag) progress->setTag(progress->user, "Rendering...");
	if(!yafaray_prepareRender(yi)) return 0;
	if(yi->abort_flag) return fail(yi, "aborted");
	const int w = yi->rp.width, h = yi->rp.height;
	yi->film.assign((size_t)w * (size_t)h * 5, 0.f);
	yafgpu_counters cn{};
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	(void)hipEventRecord(e0, nullptr);
	yi->resampled.assign((size_t)std::max(1, yi->aa.passes), 0);
	const int rc = yafgpu_render_passes_to_host(yi->gpu, &yi->rp, &yi->aa, yi->film.data(), &cn, yi->resampled.data());
	(void)hipEventRecord(e1, nullptr);
	(void)hipEventSynchronize(e1);
	float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	if(rc) return fail(yi, std::string("render: ") + yafgpu_last_error());
	yi->stats.rays_closest = cn.rays_closest; yi->stats.rays_shadow = cn.rays_shadow; yi->stats.interior_steps = cn.interior_steps;
	yi->stats.leaves = cn.leaves; yi->stats.tri_tests = cn.tri_tests; yi->stats.camera_samples = cn.camera_samples; yi->stats.restarts = cn.restarts;
	yi->stats.render_seconds = (double)ms * 1e-3;
	if(progress && progress->update) progress->update(progress->user, 100);
	if(progress && progress->done) progress->done(progress->user);
	deliver(yi, output);
	return 1;
}

void yafaray_abort(yafaray_interface_t *yi) { yi->abort_flag = 1; }
void yafaray_internal_set_error(yafaray_interface_t *yi, const char *msg) { if(yi) yi->err = msg ? msg : ""; }
void yafaray_internal_set_base_dir(yafaray_interface_t *yi, const char *dir) { if(yi) yi->base_dir = dir ? dir : ""; }

yafaray_bool_t yafaray_getRenderedImage(yafaray_interface_t *yi, int num_view, const yafaray_output_t *output)
{
	if(num_view != 0 || yi->film.empty()) return fail(yi, "getRenderedImage: nothing rendered");
	deliver(yi, output);
	return 1;
}
yafaray_bool_t yafaray_getFilm(yafaray_interface_t *yi, float *film, int width, int height)
{
	if(yi->film.empty() || width != yi->rp.width || height != yi->rp.height) return fail(yi, "getFilm: no film of that size");
	std::memcpy(film, yi->film.data(), yi->film.size() * sizeof(float));
	return 1;
}
yafaray_bool_t yafaray_getRenderStats(yafaray_interface_t *yi, yafaray_render_stats_t *stats)
{
	if(!stats) return 0;
	*stats = yi->stats;
	return 1;
}

} // extern "C"
