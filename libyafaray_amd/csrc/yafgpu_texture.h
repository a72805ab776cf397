// Image textures and shader nodes on the device.
//
//   ImageTexture            src/texture/texture_image.cc (getColor :75-88, getRawColor :90-104, doMapping :119-215,
//                           findTextureInterpolationCoordinates :224-289, noInterpolation / bilinearInterpolation :291-330),
//                           adjustments include/texture/texture.h:202-275
//   TextureMapperNode       src/shader/shader_node_basic.cc:127-229        ValueNode :423-427
//   MixNode and its modes   :446-680                                       LayerNode  src/shader/shader_node_layer.cc:29-117
//   blend functions         include/shader/shader_node.h:115-223
//
// Texels are float4 in HBM holding exactly what the reference's ImageHandler::getPixel returns (the host decodes the file,
// yafaray_image.cpp).  A textured material is evaluated once per surface point (every supported node is view independent:
// ShinyDiffuseMaterial::initBsdf, material_shiny_diffuse.cc:163-183) and what its functions would read through a shader
// slot is written into a per-lane copy of the material record (mat_resolve), so the material code itself is unchanged.
// The arithmetic follows the oracle's restatement, which is pinned bit for bit against the reference's compiled sources
// (tests/test_textures_golden.py); the device side is pinned against the same vectors through yafgpu_probe.
#pragma once

namespace yafgpu {

struct TexScene
{
	const yafgpu_texture *textures; const float4 *texels; const yafgpu_node *nodes;
	const float *tri_uv, *tri_orco;      // per triangle: u, v of the three corners (6 floats; first word NaN for a mesh without UVs) / orco of the three corners (9 floats; first word NaN for a mesh without orco), or nullptr
	const float *tri_e3;                 // bump mapping: vertex c - vertex b per triangle (3 floats), or nullptr
	int n_textures, has_bump;
};

struct Rgba4 { float r, g, b, a; };
YG_DEV Rgba4 ra4(float r, float g, float b, float a) { Rgba4 c; c.r = r; c.g = g; c.b = b; c.a = a; return c; }
YG_DEV float &ch(Rgba4 &c, int i) { return i == 0 ? c.r : (i == 1 ? c.g : (i == 2 ? c.b : c.a)); }

enum { kClipExtend = 0, kClipClip = 1, kClipCube = 2, kClipRepeat = 3, kClipChecker = 4 };
enum { kTxfRgbToInt = 1, kTxfStencil = 2, kTxfNegative = 4, kTxfAlphaMix = 8 };
enum { kMnMix = 0, kMnAdd, kMnMult, kMnSub, kMnScreen, kMnDiv, kMnDiff, kMnDark, kMnLight, kMnOverlay };
enum { kTcUv = 0, kTcGlob, kTcOrco, kTcTran, kTcNor, kTcRefl, kTcWin };

YG_DEV Rgba4 tex_pixel(const TexScene &ts, const yafgpu_texture &t, int x, int y)
{
	const float4 q = ts.texels[(size_t)t.texel_first + (size_t)y * (size_t)t.width + (size_t)x];
	return ra4(q.x, q.y, q.z, q.w);
}

// ImageTexture::doMapping; returns `outside`
YG_DEV bool tex_do_mapping(const yafgpu_texture &t, V3 &p)
{
	bool outside = false;
	p.x = 0.5f * p.x + 0.5f; p.y = 0.5f * p.y + 0.5f; p.z = 0.5f * p.z + 0.5f;
	if(t.clip == kClipRepeat)
	{
		if(t.xrepeat > 1) p.x *= (float)t.xrepeat;
		if(t.yrepeat > 1) p.y *= (float)t.yrepeat;
		if(t.mirror_x && (int)ceilf(p.x) % 2 == 0) p.x = -p.x;
		if(t.mirror_y && (int)ceilf(p.y) % 2 == 0) p.y = -p.y;
		if(p.x > 1.f) p.x -= (float)(int)p.x;
		else if(p.x < 0.f) p.x += (float)(1 - (int)p.x);
		if(p.y > 1.f) p.y -= (float)(int)p.y;
		else if(p.y < 0.f) p.y += (float)(1 - (int)p.y);
	}
	if(t.cropx) p.x = t.cropminx + p.x * (t.cropmaxx - t.cropminx);
	if(t.cropy) p.y = t.cropminy + p.y * (t.cropmaxy - t.cropminy);
	if(t.rot90) { const float tmp = p.x; p.x = p.y; p.y = tmp; }
	if(t.clip == kClipCube)
	{
		if((p.x < 0.f) || (p.x > 1.f) || (p.y < 0.f) || (p.y > 1.f) || (p.z < -1.f) || (p.z > 1.f)) outside = true;
	}
	else if(t.clip == kClipChecker || t.clip == kClipClip)
	{
		bool stop = false;
		if(t.clip == kClipChecker)
		{
			const int xs = (int)floor((double)p.x), ys = (int)floor((double)p.y);
			p.x -= (float)xs; p.y -= (float)ys;
			if(!t.checker_odd && !((xs + ys) & 1)) { outside = true; stop = true; }
			else if(!t.checker_even && ((xs + ys) & 1)) { outside = true; stop = true; }
			else if((double)t.checker_dist < 1.0)
			{
				p.x = (float)(((double)p.x - 0.5) / (1.0 - (double)t.checker_dist) + 0.5);
				p.y = (float)(((double)p.y - 0.5) / (1.0 - (double)t.checker_dist) + 0.5);
			}
		}
		if(!stop && ((p.x < 0.f) || (p.x > 1.f) || (p.y < 0.f) || (p.y > 1.f))) outside = true;
	}
	else if(t.clip == kClipExtend)
	{
		if(p.x > 0.99999f) p.x = 0.99999f; else if(p.x < 0.f) p.x = 0.f;
		if(p.y > 0.99999f) p.y = 0.99999f; else if(p.y < 0.f) p.y = 0.f;
	}
	return outside;
}

YG_DEV void tex_interp_coords(int &c0, int &c1, int &c2, int &c3, float &dec, float cf, int res, bool repeat, bool mirror)
{
	if(repeat)
	{
		c1 = ((int)cf) % res;
		if(mirror)
		{
			if(cf < 0.f) { c0 = 1 % res; c2 = c1; c3 = c0; dec = -cf; }
			else if(cf >= (float)res - 1.f) { c0 = (res + res - 1) % res; c2 = c1; c3 = c0; dec = cf - (float)((int)cf); }
			else
			{
				c0 = (res + c1 - 1) % res;
				c2 = c1 + 1; if(c2 >= res) c2 = (res + res - c2) % res;
				c3 = c1 + 2; if(c3 >= res) c3 = (res + res - c3) % res;
				dec = cf - (float)((int)cf);
			}
		}
		else
		{
			if(cf > 0.f) { c0 = (res + c1 - 1) % res; c2 = (c1 + 1) % res; c3 = (c1 + 2) % res; dec = cf - (float)((int)cf); }
			else { c0 = 1 % res; c2 = (res - 1) % res; c3 = (res - 2) % res; dec = -cf; }
		}
	}
	else
	{
		const int ci = (int)cf;
		c1 = ci < 0 ? 0 : (ci > res - 1 ? res - 1 : ci);
		if(cf > 0.f) c2 = (c1 + 1 < res - 1) ? c1 + 1 : res - 1; else c2 = 0;
		c0 = (c1 - 1 > 0) ? c1 - 1 : 0;
		c3 = (c2 + 1 < res - 1) ? c2 + 1 : res - 1;
		dec = (float)((double)cf - floor((double)cf));
	}
}

YG_DEV Rgba4 tex_interpolate(const TexScene &ts, const yafgpu_texture &t, V3 p)
{
	const int resx = t.width, resy = t.height;
	int x0, x1, x2, x3, y0, y1, y2, y3; float dx, dy;
	const bool rep = t.clip == kClipRepeat;
	// floor() in the reference's expressions is C's double floor: evaluated in double, narrowed once (pinned by the golden vectors)
	if(t.interpolate == 0)
	{
		const float xf = (float)((double)(float)resx * ((double)p.x - floor((double)p.x)));
		const float yf = (float)((double)(float)resy * ((double)p.y - floor((double)p.y)));
		tex_interp_coords(x0, x1, x2, x3, dx, xf, resx, rep, t.mirror_x != 0);
		tex_interp_coords(y0, y1, y2, y3, dy, yf, resy, rep, t.mirror_y != 0);
		return tex_pixel(ts, t, x1, y1);
	}
	const float xf = (float)((double)(float)resx * ((double)p.x - floor((double)p.x)) - (double)0.5f);
	const float yf = (float)((double)(float)resy * ((double)p.y - floor((double)p.y)) - (double)0.5f);
	tex_interp_coords(x0, x1, x2, x3, dx, xf, resx, rep, t.mirror_x != 0);
	tex_interp_coords(y0, y1, y2, y3, dy, yf, resy, rep, t.mirror_y != 0);
	const Rgba4 c11 = tex_pixel(ts, t, x1, y1), c21 = tex_pixel(ts, t, x2, y1), c12 = tex_pixel(ts, t, x1, y2), c22 = tex_pixel(ts, t, x2, y2);
	const float w11 = (1.f - dx) * (1.f - dy), w12 = (1.f - dx) * dy, w21 = dx * (1.f - dy), w22 = dx * dy;
	Rgba4 o;
	o.r = ((w11 * c11.r + w12 * c12.r) + w21 * c21.r) + w22 * c22.r;
	o.g = ((w11 * c11.g + w12 * c12.g) + w21 * c21.g) + w22 * c22.g;
	o.b = ((w11 * c11.b + w12 * c12.b) + w21 * c21.b) + w22 * c22.b;
	o.a = ((w11 * c11.a + w12 * c12.a) + w21 * c21.a) + w22 * c22.a;
	return o;
}

YG_DEV void rgb_to_hsv(Rgba4 c, float &h, float &s, float &v)
{
	const float r_1 = smax(c.r, 0.f), g_1 = smax(c.g, 0.f), b_1 = smax(c.b, 0.f);
	const float max_component = smax(smax(r_1, g_1), b_1), min_component = smin(smin(r_1, g_1), b_1);
	const float range = max_component - min_component;
	v = max_component;
	if(fabsf(range) < 1.0e-6f) { h = 0.f; s = 0.f; }
	else if(max_component == r_1) { h = fmodf((g_1 - b_1) / range, 6.f); s = range / smax(v, 1.0e-6f); }
	else if(max_component == g_1) { h = ((b_1 - r_1) / range) + 2.f; s = range / smax(v, 1.0e-6f); }
	else if(max_component == b_1) { h = ((r_1 - g_1) / range) + 4.f; s = range / smax(v, 1.0e-6f); }
	else { h = 0.f; s = 0.f; v = 0.f; }
	if(h < 0.f) h += 6.f;
}
YG_DEV void hsv_to_rgb(Rgba4 &o, float h, float s, float v)
{
	const float c = v * s;
	const float x = c * (1.f - fabsf(fmodf(h, 2.f) - 1.f));
	const float m = v - c;
	float r_1 = 0.f, g_1 = 0.f, b_1 = 0.f;
	if(h >= 0.f && h < 1.f) { r_1 = c; g_1 = x; b_1 = 0.f; }
	else if(h >= 1.f && h < 2.f) { r_1 = x; g_1 = c; b_1 = 0.f; }
	else if(h >= 2.f && h < 3.f) { r_1 = 0.f; g_1 = c; b_1 = x; }
	else if(h >= 3.f && h < 4.f) { r_1 = 0.f; g_1 = x; b_1 = c; }
	else if(h >= 4.f && h < 5.f) { r_1 = x; g_1 = 0.f; b_1 = c; }
	else if(h >= 5.f && h < 6.f) { r_1 = c; g_1 = 0.f; b_1 = x; }
	o.r = r_1 + m; o.g = g_1 + m; o.b = b_1 + m;
}
YG_DEV Rgba4 clamp_rgb0(Rgba4 c) { if(c.r < 0.f) c.r = 0.f; if(c.g < 0.f) c.g = 0.f; if(c.b < 0.f) c.b = 0.f; return c; }

YG_DEV Rgba4 tex_apply_adjustments(const yafgpu_texture &t, Rgba4 c)
{
	if(!t.adj_set) return c;
	Rgba4 ret = c;
	if(t.adj_int != 1.f || t.adj_con != 1.f)
	{
		ret.r = (c.r - 0.5f) * t.adj_con + t.adj_int - 0.5f;
		ret.g = (c.g - 0.5f) * t.adj_con + t.adj_int - 0.5f;
		ret.b = (c.b - 0.5f) * t.adj_con + t.adj_int - 0.5f;
	}
	if(t.adj_clamp) ret = clamp_rgb0(ret);
	if(t.adj_r != 1.f) ret.r *= t.adj_r;
	if(t.adj_g != 1.f) ret.g *= t.adj_g;
	if(t.adj_b != 1.f) ret.b *= t.adj_b;
	if(t.adj_clamp) ret = clamp_rgb0(ret);
	if(t.adj_sat != 1.f || t.adj_hue != 0.f)
	{
		float h = 0.f, sa = 0.f, v = 0.f;
		rgb_to_hsv(ret, h, sa, v);
		sa *= t.adj_sat;
		h += t.adj_hue;
		if(h < 0.f) h += 6.f; else if(h > 6.f) h -= 6.f;
		hsv_to_rgb(ret, h, sa, v);
		if(t.adj_clamp) ret = clamp_rgb0(ret);
	}
	return ret;
}

YG_DEV Rgba4 tex_get_color(const TexScene &ts, const yafgpu_texture &t, V3 p)
{
	V3 p_1 = mk(p.x, -p.y, p.z);
	if(tex_do_mapping(t, p_1)) return ra4(0.f, 0.f, 0.f, 0.f);
	return tex_apply_adjustments(t, tex_interpolate(ts, t, p_1));
}
// ImageTexture::getRawColor (texture_image.cc:90-104): getColor re-encoded into the texture's colour space
YG_DEV Rgba4 tex_get_raw_color(const TexScene &ts, const yafgpu_texture &t, V3 p)
{
	Rgba4 c = tex_get_color(ts, t, p);
	if(t.color_space == 0)
	{
		c.r = (c.r <= 0.0031308f) ? (c.r * 12.92f) : ((1.055f * f_pow(c.r, 0.416667f)) - 0.055f);
		c.g = (c.g <= 0.0031308f) ? (c.g * 12.92f) : ((1.055f * f_pow(c.g, 0.416667f)) - 0.055f);
		c.b = (c.b <= 0.0031308f) ? (c.b * 12.92f) : ((1.055f * f_pow(c.b, 0.416667f)) - 0.055f);
	}
	else if(t.color_space == 1)
	{
		const float r = c.r, g = c.g, b = c.b;
		c.r = 0.412400f * r + 0.357600f * g + 0.180500f * b;
		c.g = 0.212600f * r + 0.715200f * g + 0.072200f * b;
		c.b = 0.019300f * r + 0.119200f * g + 0.950500f * b;
	}
	else if(t.color_space == 3 && t.gamma != 1.f)
	{
		float gamma = t.gamma;
		if(gamma <= 0.f) gamma = 1.0e-2f;
		const float inv = 1.f / gamma;
		c.r = f_pow(c.r, inv); c.g = f_pow(c.g, inv); c.b = f_pow(c.b, inv);
	}
	return c;
}
YG_DEV float tex_get_float(const TexScene &ts, const yafgpu_texture &t, V3 p)
{	// Texture::getFloat = applyIntensityContrastAdjustments(getRawColor(p).col2Bri())
	const Rgba4 c = tex_get_raw_color(ts, t, p);
	float f = (0.2126f * c.r + 0.7152f * c.g + 0.0722f * c.b);
	if(!t.adj_set) return f;
	if(t.adj_int != 1.f || t.adj_con != 1.f) f = (f - 0.5f) * t.adj_con + t.adj_int - 0.5f;
	if(t.adj_clamp) { if(f < 0.f) f = 0.f; else if(f > 1.f) f = 1.f; }
	return f;
}

// fAcos__ util_math_optimizations.h:255-261
YG_DEV float f_acos(float x) { if((double)x <= -1.0) return (float)kPi; else if((double)x >= 1.0) return 0.0f; else return (float)acos((double)x); }

// TextureMapperNode::doMapping
YG_DEV V3 mapper_do_mapping(const yafgpu_node &n, V3 p, V3 ng)
{
	V3 texpt = p;
	if(n.texco == kTcUv) texpt = mk(2.0f * texpt.x - 1.0f, 2.0f * texpt.y - 1.0f, texpt.z);
	{
		const V3 q = texpt;
		texpt.x = n.map_x == 0 ? 0.f : comp(q, n.map_x - 1);
		texpt.y = n.map_y == 0 ? 0.f : comp(q, n.map_y - 1);
		texpt.z = n.map_z == 0 ? 0.f : comp(q, n.map_z - 1);
	}
	if(n.mapping == 2)
	{	// tubemap__
		V3 res; res.y = texpt.z;
		const float d = texpt.x * texpt.x + texpt.y * texpt.y;
		if(d > 0.f) { res.z = (float)(1.0 / (double)f_sqrt(d)); res.x = (float)(-atan2((double)texpt.x, (double)texpt.y) * k1Pi); }
		else { res.x = 0.f; res.z = 0.f; }
		texpt = res;
	}
	else if(n.mapping == 3)
	{	// spheremap__
		V3 res = mk(0.f, 0.f, 0.f);
		const float d = texpt.x * texpt.x + texpt.y * texpt.y + texpt.z * texpt.z;
		if(d > 0.f)
		{
			res.z = f_sqrt(d);
			if((texpt.x != 0.f) && (texpt.y != 0.f)) res.x = (float)(-atan2((double)texpt.x, (double)texpt.y) * k1Pi);
			res.y = (float)((double)1.0f - (double)2.0f * ((double)f_acos(texpt.z / res.z) * k1Pi));
		}
		texpt = res;
	}
	else if(n.mapping == 1)
	{	// cubemap__: ma = {{1,2,0},{0,2,1},{0,1,2}}
		int axis;
		if(fabsf(ng.z) >= fabsf(ng.x) && fabsf(ng.z) >= fabsf(ng.y)) axis = 2;
		else if(fabsf(ng.y) >= fabsf(ng.x) && fabsf(ng.y) >= fabsf(ng.z)) axis = 1;
		else axis = 0;
		const V3 q = texpt;
		if(axis == 0) texpt = mk(q.y, q.z, q.x);
		else if(axis == 1) texpt = mk(q.x, q.z, q.y);
	}
	return mk(texpt.x * n.scale[0] + n.offset[0], texpt.y * n.scale[1] + n.offset[1], texpt.z * n.scale[2] + n.offset[2]);
}

struct NodeResult { Rgba4 col; float f; };
struct TexPoint { V3 p, n, ng, orco_p, orco_ng; float u, v; bool has_uv; V3 nu, nv, ds_du, ds_dv; };      // what the nodes read of a SurfacePoint (nu .. ds_dv: bump mapping only)

constexpr int kMaxNodes = 16;       // nodes per material on the device (a layer stack of 8 textures; the host refuses more)

// NodeMaterial::evalNodes over nodes[0 .. n_nodes) in evaluation order; stack[k] = node k's result
YG_DEV void nodes_eval(const TexScene &ts, const yafgpu_node *nodes, int n_nodes, const yafgpu_camera &cam, const TexPoint &sp, NodeResult *stack)
{
	for(int k = 0; k < n_nodes; ++k)
	{
		const yafgpu_node &n = nodes[k];
		NodeResult res; res.col = ra4(0.f, 0.f, 0.f, 0.f); res.f = 0.f;
		if(n.type == YAFGPU_NODE_TEXTURE_MAPPER)
		{
			V3 texpt, ng;
			if(n.texco == kTcUv) { texpt = mk(sp.u, sp.v, 0.f); ng = sp.ng; }
			else if(n.texco == kTcOrco) { texpt = sp.orco_p; ng = sp.orco_ng; }
			else if(n.texco == kTcTran)
			{
				const float *m = n.mtx;
				texpt = mk(m[0] * sp.p.x + m[1] * sp.p.y + m[2] * sp.p.z + m[3], m[4] * sp.p.x + m[5] * sp.p.y + m[6] * sp.p.z + m[7], m[8] * sp.p.x + m[9] * sp.p.y + m[10] * sp.p.z + m[11]);
				ng = mk(m[0] * sp.ng.x + m[1] * sp.ng.y + m[2] * sp.ng.z, m[4] * sp.ng.x + m[5] * sp.ng.y + m[6] * sp.ng.z, m[8] * sp.ng.x + m[9] * sp.ng.y + m[10] * sp.ng.z);
			}
			else if(n.texco == kTcWin)
			{	// PerspectiveCamera::screenproject, camera_perspective.cc:158-173
				const V3 dir = sp.p - vec3(cam.position);
				const float dx = dot(dir, vec3(cam.cam_x)), dy = dot(dir, vec3(cam.cam_y)), dz = dot(dir, vec3(cam.cam_z));
				texpt = mk(2.0f * dx * cam.focal_distance / dz, -2.0f * dy * cam.focal_distance / (dz * cam.aspect_ratio), 0.f);
				ng = sp.ng;
			}
			else if(n.texco == kTcNor) { texpt = mk(dot(sp.n, vec3(cam.cam_x)), -dot(sp.n, vec3(cam.cam_y)), 0.f); ng = sp.ng; }
			else { texpt = sp.p; ng = sp.ng; }
			texpt = mapper_do_mapping(n, texpt, ng);
			if(n.texture >= 0 && n.texture < ts.n_textures)
			{
				const yafgpu_texture &t = ts.textures[n.texture];
				res.col = tex_get_color(ts, t, texpt);
				res.f = n.do_scalar ? tex_get_float(ts, t, texpt) : 0.f;
			}
		}
		else if(n.type == YAFGPU_NODE_VALUE) { res.col = ra4(n.color[0], n.color[1], n.color[2], n.color[3]); res.f = n.value; }
		else if(n.type == YAFGPU_NODE_MIX)
		{	// MixNode::getInputs; val_1_ / val_2_ are never set by the reference: 0
			const float f_2 = (n.factor >= 0) ? stack[n.factor].f : n.cfactor;
			Rgba4 c1, c2; float fin_1, fin_2;
			if(n.input1 >= 0) { c1 = stack[n.input1].col; fin_1 = stack[n.input1].f; } else { c1 = ra4(n.col1[0], n.col1[1], n.col1[2], n.col1[3]); fin_1 = 0.f; }
			if(n.input2 >= 0) { c2 = stack[n.input2].col; fin_2 = stack[n.input2].f; } else { c2 = ra4(n.col2[0], n.col2[1], n.col2[2], n.col2[3]); fin_2 = 0.f; }
			const float f_1 = 1.f - f_2;
			if(n.mode == kMnAdd) { for(int i = 0; i < 4; ++i) ch(c1, i) += f_2 * ch(c2, i); fin_1 += f_2 * fin_2; }
			else if(n.mode == kMnMult) { for(int i = 0; i < 4; ++i) ch(c1, i) *= f_1 + f_2 * ch(c2, i); }
			else if(n.mode == kMnSub) { for(int i = 0; i < 4; ++i) ch(c1, i) -= f_2 * ch(c2, i); fin_1 -= f_2 * fin_2; }
			else if(n.mode == kMnScreen)
			{
				for(int i = 0; i < 4; ++i) ch(c1, i) = 1.f - (f_1 + f_2 * (1.f - ch(c2, i))) * (1.f - ch(c1, i));
				fin_1 = (float)(1.0 - (double)((f_1 + f_2 * (1.f - fin_2)) * (1.f - fin_1)));
			}
			else if(n.mode == kMnDiff)
			{
				for(int i = 0; i < 4; ++i) ch(c1, i) = f_1 * ch(c1, i) + f_2 * fabsf(ch(c1, i) - ch(c2, i));
				fin_1 = f_1 * fin_1 + f_2 * fabsf(fin_1 - fin_2);
			}
			else if(n.mode == kMnDark)
			{
				for(int i = 0; i < 4; ++i) { ch(c2, i) *= f_2; if(ch(c2, i) < ch(c1, i)) ch(c1, i) = ch(c2, i); }
				fin_2 *= f_2; if(fin_2 < fin_1) fin_1 = fin_2;
			}
			else if(n.mode == kMnLight)
			{
				for(int i = 0; i < 4; ++i) { ch(c2, i) *= f_2; if(ch(c2, i) > ch(c1, i)) ch(c1, i) = ch(c2, i); }
				fin_2 *= f_2; if(fin_2 > fin_1) fin_1 = fin_2;
			}
			else if(n.mode == kMnOverlay)
			{
				Rgba4 o;
				for(int i = 0; i < 4; ++i)
				{
					const float a = ch(c1, i), b = ch(c2, i);
					ch(o, i) = (a < 0.5f) ? a * (f_1 + 2.0f * f_2 * b) : (float)(1.0 - ((double)f_1 + (double)(2.0f * f_2) * (1.0 - (double)b)) * (1.0 - (double)a));
				}
				fin_1 = (fin_1 < 0.5f) ? fin_1 * (f_1 + 2.0f * f_2 * fin_2) : (float)(1.0 - ((double)f_1 + (double)(2.0f * f_2) * (1.0 - (double)fin_2)) * (1.0 - (double)fin_1));
				c1 = o;
			}
			else
			{
				for(int i = 0; i < 4; ++i) ch(c1, i) = f_1 * ch(c1, i) + f_2 * ch(c2, i);
				fin_1 = f_1 * fin_1 + f_2 * fin_2;
			}
			res.col = c1; res.f = fin_1;
		}
		else if(n.type == YAFGPU_NODE_LAYER)
		{
			Rgba4 rcol, texcolor = ra4(0.f, 0.f, 0.f, 0.f);
			float rval, tin = 0.f, ta = 1.f, stencil_tin;
			rcol = (n.upper >= 0) ? stack[n.upper].col : ra4(n.upper_col[0], n.upper_col[1], n.upper_col[2], n.upper_col[3]);
			rval = (n.upper >= 0) ? stack[n.upper].f : n.upper_val;
			stencil_tin = rcol.a;
			bool tex_rgb = n.color_input != 0;
			if(n.color_input) { texcolor = stack[n.input].col; ta = texcolor.a; }
			else tin = stack[n.input].f;
			if(n.texflag & kTxfRgbToInt) { tin = (0.2126f * texcolor.r + 0.7152f * texcolor.g + 0.0722f * texcolor.b); tex_rgb = false; }
			if(n.texflag & kTxfNegative)
			{
				if(tex_rgb) texcolor = ra4(1.f - texcolor.r, 1.f - texcolor.g, 1.f - texcolor.b, 1.f - texcolor.a);
				tin = 1.f - tin;
			}
			if(n.texflag & kTxfStencil)
			{
				if(tex_rgb) { const float fact = ta; ta *= stencil_tin; stencil_tin *= fact; }
				else { const float fact = tin; tin *= stencil_tin; stencil_tin *= fact; }
			}
			if(n.do_color)
			{
				if(!tex_rgb) texcolor = ra4(n.def_col[0], n.def_col[1], n.def_col[2], 1.f); else tin = ta;
				const float tt = tin > 1.f ? 1.f : (tin < 0.f ? 0.f : tin);
				const float facg = stencil_tin * n.colfac;
				float f = tt;
				Rgba4 r = ra4(0.f, 0.f, 0.f, 1.f);
				for(int i = 0; i < 3; ++i)
				{
					const float tex = ch(texcolor, i), out = ch(rcol, i);
					float v;
					if(n.mode == kMnMult) { const float ff = f * facg; v = ((1.f - facg) + ff * tex) * out; }
					else if(n.mode == kMnScreen) { const float ff = f * facg; v = 1.0f - ((1.f - facg) + ff * (1.0f - tex)) * (1.0f - out); }
					else if(n.mode == kMnSub) { const float ff = (-f) * facg; v = ff * tex + out; }
					else if(n.mode == kMnAdd) { const float ff = f * facg; v = ff * tex + out; }
					else if(n.mode == kMnDiv) { const float ff = f * facg; const float it = (tex != 0.f) ? 1.f / tex : tex; v = (1.f - ff) * out + (ff * out) * it; }
					else if(n.mode == kMnDiff) { const float ff = f * facg; v = (1.f - ff) * out + ff * fabsf(tex - out); }
					else if(n.mode == kMnDark) { const float ff = f * facg; const float c = ff * tex; v = (out < c) ? out : c; }
					else if(n.mode == kMnLight) { const float ff = f * facg; const float c = ff * tex; v = (out > c) ? out : c; }
					else { const float ff = f * facg; v = ff * tex + (1.f - ff) * out; }
					ch(r, i) = v;
				}
				rcol = clamp_rgb0(r);
			}
			if(n.do_scalar_l)
			{
				if(tex_rgb)
				{
					if(n.use_alpha) { tin = ta; if(n.texflag & kTxfNegative) tin = 1.f - tin; }
					else tin = (0.2126f * texcolor.r + 0.7152f * texcolor.g + 0.0722f * texcolor.b);
				}
				const float facg = stencil_tin * n.valfac;
				float f = tin * facg, facm = 1.f - f;
				const float tex = n.def_val, out = rval;
				if(n.mode == kMnMult) { facm = 1.f - facg; rval = (facm + f * tex) * out; }
				else if(n.mode == kMnScreen) { facm = 1.f - facg; rval = 1.f - (facm + f * (1.f - tex)) * (1.f - out); }
				else if(n.mode == kMnSub) { f = -f; rval = f * tex + out; }
				else if(n.mode == kMnAdd) rval = f * tex + out;
				else if(n.mode == kMnDiv) rval = (tex == 0.f) ? 0.f : facm * out + f * out / tex;
				else if(n.mode == kMnDiff) rval = facm * out + f * fabsf(tex - out);
				else if(n.mode == kMnDark) { const float c = f * tex; rval = (c < out) ? c : out; }
				else if(n.mode == kMnLight) { const float c = f * tex; rval = (c > out) ? c : out; }
				else rval = f * tex + facm * out;
				if(rval < 0.f) rval = 0.f;
			}
			rcol.a = stencil_tin;
			res.col = rcol; res.f = rval;
		}
		stack[k] = res;
	}
}

// texture coordinates of a surface point: Triangle::getSurface, triangle.cc:46-79,103-111
YG_DEV void tex_point(const TexScene &ts, int tri, float bu, float bv, V3 p, V3 n, V3 ng, TexPoint &tp)
{
	tp.p = p; tp.n = n; tp.ng = ng;
	const float u = 1.f - bu - bv, v = bu, w = bv;
	const float *q_orco = ts.tri_orco != nullptr ? ts.tri_orco + 9 * (size_t)tri : nullptr;
	if(q_orco != nullptr && q_orco[0] == q_orco[0])        // a NaN first word marks a triangle of a mesh without orco (has_orco_ is per mesh)
	{
		const float *q = q_orco;
		const V3 p_0 = mk(q[0], q[1], q[2]), p_1 = mk(q[3], q[4], q[5]), p_2 = mk(q[6], q[7], q[8]);
		tp.orco_p = p_0 * u + p_1 * v + p_2 * w;
		tp.orco_ng = normalize(cross(p_1 - p_0, p_2 - p_0));
	}
	else { tp.orco_p = p; tp.orco_ng = ng; }
	const float *q_uv = ts.tri_uv != nullptr ? ts.tri_uv + 6 * (size_t)tri : nullptr;
	tp.has_uv = q_uv != nullptr && q_uv[0] == q_uv[0];     // a NaN first word: the triangle's mesh has no UVs (has_uv_ is per mesh)
	if(tp.has_uv)
	{
		const float *q = q_uv;
		tp.u = u * q[0] + v * q[2] + w * q[4];
		tp.v = u * q[1] + v * q[3] + w * q[5];
	}
	else { tp.u = 0.f; tp.v = 0.f; }
}

// dPdU / dPdV in shading space, Triangle::getSurface (triangle.cc:80-130): only bump mapping reads them.  a, e1 = b - a, e2 = c - a from the
// triangle record, e3 = c - b from its own array (the differences the reference forms, each rounded once)
YG_DEV void tex_point_derivatives(const TexScene &ts, int tri, V3 e1, V3 e2, V3 n, V3 nu, V3 nv, TexPoint &tp)
{
	const float *q3 = ts.tri_e3 + 3 * (size_t)tri;
	const V3 e3 = mk(q3[0], q3[1], q3[2]);
	V3 dp_du = e1, dp_dv = e3;                               // implicit mapping: p_1 - p_0, p_2 - p_1
	if(tp.has_uv)
	{
		const float *q = ts.tri_uv + 6 * (size_t)tri;
		const float du_1 = q[0] - q[4], du_2 = q[2] - q[4], dv_1 = q[1] - q[5], dv_2 = q[3] - q[5];
		const float det = du_1 * dv_2 - dv_1 * du_2;
		if(fabsf(det) > 1e-30f)
		{
			const float invdet = 1.f / det;
			const V3 dp_1 = -e2, dp_2 = -e3;                 // p_0 - p_2, p_1 - p_2
			dp_du = (dp_1 * dv_2 - dp_2 * dv_1) * invdet;
			dp_dv = (dp_2 * du_1 - dp_1 * du_2) * invdet;
		}
	}
	dp_du = normalize(dp_du); dp_dv = normalize(dp_dv);
	tp.nu = nu; tp.nv = nv;
	tp.ds_du = mk(dot(nu, dp_du), dot(nv, dp_du), dot(n, dp_du));
	tp.ds_dv = mk(dot(nu, dp_dv), dot(nv, dp_dv), dot(n, dp_dv));
}

// NodeMaterial::evalBump's node pass (material_node.cc:132-139): evalDerivative of every node in order — TextureMapperNode :232-343
// (image textures: discrete, no normal maps), LayerNode :122-152, the base class's zero for the others (shader_node.h:87-88)
YG_DEV V3 mapper_get_coords(const yafgpu_node &n, const yafgpu_camera &cam, const TexPoint &sp, V3 &ng)
{
	V3 texpt;
	if(n.texco == kTcUv) { texpt = mk(sp.u, sp.v, 0.f); ng = sp.ng; }
	else if(n.texco == kTcOrco) { texpt = sp.orco_p; ng = sp.orco_ng; }
	else if(n.texco == kTcTran)
	{
		const float *m = n.mtx;
		texpt = mk(m[0] * sp.p.x + m[1] * sp.p.y + m[2] * sp.p.z + m[3], m[4] * sp.p.x + m[5] * sp.p.y + m[6] * sp.p.z + m[7], m[8] * sp.p.x + m[9] * sp.p.y + m[10] * sp.p.z + m[11]);
		ng = mk(m[0] * sp.ng.x + m[1] * sp.ng.y + m[2] * sp.ng.z, m[4] * sp.ng.x + m[5] * sp.ng.y + m[6] * sp.ng.z, m[8] * sp.ng.x + m[9] * sp.ng.y + m[10] * sp.ng.z);
	}
	else if(n.texco == kTcWin)
	{
		const V3 dir = sp.p - vec3(cam.position);
		const float dx = dot(dir, vec3(cam.cam_x)), dy = dot(dir, vec3(cam.cam_y)), dz = dot(dir, vec3(cam.cam_z));
		texpt = mk(2.0f * dx * cam.focal_distance / dz, -2.0f * dy * cam.focal_distance / (dz * cam.aspect_ratio), 0.f);
		ng = sp.ng;
	}
	else if(n.texco == kTcNor) { texpt = mk(dot(sp.n, vec3(cam.cam_x)), -dot(sp.n, vec3(cam.cam_y)), 0.f); ng = sp.ng; }
	else { texpt = sp.p; ng = sp.ng; }
	return texpt;
}
YG_DEV void nodes_eval_derivative(const TexScene &ts, const yafgpu_node *nodes, int n_nodes, const yafgpu_camera &cam, const TexPoint &sp, NodeResult *stack)
{
	for(int k = 0; k < n_nodes; ++k)
	{
		const yafgpu_node &n = nodes[k];
		NodeResult res; res.col = ra4(0.f, 0.f, 0.f, 0.f); res.f = 0.f;
		if(n.type == YAFGPU_NODE_TEXTURE_MAPPER && n.texture >= 0 && n.texture < ts.n_textures)
		{
			const yafgpu_texture &t = ts.textures[n.texture];
			V3 ng;
			V3 texpt = mapper_get_coords(n, cam, sp, ng);
			float du = 0.0f, dv = 0.0f;
			if(t.normalmap)
			{	// :245-258 / :287-312: both branches read the normal from the texture's raw colour
				texpt = mapper_do_mapping(n, texpt, ng);
				const Rgba4 color = tex_get_raw_color(ts, t, texpt);
				const V3 norm = normalize(mk(2.f * color.r - 1.f, 2.f * color.g - 1.f, 2.f * color.b - 1.f));
				if(fabsf(norm.z) > 1e-30f)
				{
					const float nf = (float)(1.0 / (double)norm.z * (double)n.bump_str);
					du = norm.x * nf; dv = norm.y * nf;
				}
			}
			else if(sp.has_uv && n.texco == kTcUv)
			{
				texpt = mapper_do_mapping(n, texpt, ng);
				const V3 i_0 = mk(texpt.x - n.d_u, texpt.y - 0.f, texpt.z - 0.f), i_1 = mk(texpt.x + n.d_u, texpt.y + 0.f, texpt.z + 0.f);
				const V3 j_0 = mk(texpt.x - 0.f, texpt.y - n.d_v, texpt.z - 0.f), j_1 = mk(texpt.x + 0.f, texpt.y + n.d_v, texpt.z + 0.f);
				const float dfdu = (tex_get_float(ts, t, i_0) - tex_get_float(ts, t, i_1)) / n.d_u;
				const float dfdv = (tex_get_float(ts, t, j_0) - tex_get_float(ts, t, j_1)) / n.d_v;
				V3 vec_u = sp.ds_du, vec_v = sp.ds_dv;
				vec_u.z = dfdu; vec_v.z = dfdv;
				const V3 norm = normalize(cross(vec_u, vec_v));
				if(fabsf(norm.z) > 1e-30f)
				{
					const float nf = (float)(1.0 / (double)norm.z * (double)n.bump_str);
					du = norm.x * nf; dv = norm.y * nf;
				}
			}
			else
			{
				const V3 i_0 = mapper_do_mapping(n, texpt - sp.nu * n.d_u, ng), i_1 = mapper_do_mapping(n, texpt + sp.nu * n.d_u, ng);
				const V3 j_0 = mapper_do_mapping(n, texpt - sp.nv * n.d_v, ng), j_1 = mapper_do_mapping(n, texpt + sp.nv * n.d_v, ng);
				du = (tex_get_float(ts, t, i_0) - tex_get_float(ts, t, i_1)) / n.d_u;
				dv = (tex_get_float(ts, t, j_0) - tex_get_float(ts, t, j_1)) / n.d_v;
				du *= n.bump_str; dv *= n.bump_str;
				if(n.texco != kTcUv) { du = -du; dv = -dv; }
			}
			res.col = ra4(du, dv, 0.f, 0.f);
		}
		else if(n.type == YAFGPU_NODE_LAYER)
		{
			float rdu = 0.f, rdv = 0.f, stencil_tin = 1.f;
			if(n.upper >= 0) { rdu = stack[n.upper].col.r; rdv = stack[n.upper].col.g; stencil_tin = stack[n.upper].col.a; }
			float tdu = stack[n.input].col.r, tdv = stack[n.input].col.g;
			if(n.texflag & kTxfNegative) { tdu = -tdu; tdv = -tdv; }
			rdu += tdu; rdv += tdv;
			res.col = ra4(rdu, rdv, 0.f, stencil_tin);
		}
		stack[k] = res;
	}
}
// Material::applyBump, material.cc:77-84
YG_DEV void apply_bump(V3 &n, V3 &nu, V3 &nv, float df_dnu, float df_dnv)
{
	nu = nu + n * df_dnu;
	nv = nv + n * df_dnv;
	n = normalize(cross(nu, nv));
	nu = normalize(nu);
	nv = normalize(cross(n, nu));
}

// the material record as its functions see it at this surface point (see the header comment)
YG_DEV void mat_resolve(const TexScene &ts, const yafgpu_camera &cam, const yafgpu_material &m, const TexPoint &tp, yafgpu_material &out)
{
	NodeResult stack[kMaxNodes];
	const int nn = m.n_nodes < kMaxNodes ? m.n_nodes : kMaxNodes;
	nodes_eval(ts, ts.nodes + m.node_first, nn, cam, tp, stack);
	out = m;
	if(m.type == YAFGPU_MAT_GLASS)
	{	// material_glass.cc:87-95,109,121,143-190,223-224,262-300
		if(m.sh_mirror_color >= 0) { const Rgba4 c = stack[m.sh_mirror_color].col; out.mirror_color[0] = c.r; out.mirror_color[1] = c.g; out.mirror_color[2] = c.b; }
		if(m.sh_filter_color >= 0) { const Rgba4 c = stack[m.sh_filter_color].col; out.filter_color[0] = c.r; out.filter_color[1] = c.g; out.filter_color[2] = c.b; }
		if(m.sh_ior >= 0) { out.glass_ior = m.ior_base + stack[m.sh_ior].f; out.transp_ior = stack[m.sh_ior].f; }      // :223 sic: not added there
		return;
	}
	if(m.type != YAFGPU_MAT_SHINYDIFFUSE)
	{	// glossy / coated glossy: every use of a shader is `shader ? shader->get...(stack) : member` (material_glossy.cc:62,144-160,
		// material_coated_glossy.cc:78,147-174,253-256,448-451), so the members of the per-hit copy carry them
		if(m.sh_diffuse >= 0) { const Rgba4 c = stack[m.sh_diffuse].col; out.diff_color[0] = c.r; out.diff_color[1] = c.g; out.diff_color[2] = c.b; }
		if(m.sh_glossy >= 0) { const Rgba4 c = stack[m.sh_glossy].col; out.gloss_color[0] = c.r; out.gloss_color[1] = c.g; out.gloss_color[2] = c.b; }
		if(m.sh_glossy_reflect >= 0) out.reflectivity = stack[m.sh_glossy_reflect].f;
		if(m.sh_exponent >= 0) out.exponent = stack[m.sh_exponent].f;
		if(m.sh_sigma_oren >= 0)
		{
			const double sigma = (double)stack[m.sh_sigma_oren].f, s2 = sigma * sigma;
			out.oren_tex = 1; out.oren_ad = 1.0 - 0.5 * (s2 / (s2 + 0.33)); out.oren_bd = 0.45 * s2 / (s2 + 0.09);
		}
		if(m.sh_diffuse_refl >= 0) { out.has_diffuse_refl = 1; out.diffuse_refl = stack[m.sh_diffuse_refl].f; }
		if(m.sh_mirror_color >= 0) { const Rgba4 c = stack[m.sh_mirror_color].col; out.mirror_color[0] = c.r; out.mirror_color[1] = c.g; out.mirror_color[2] = c.b; }
		if(m.sh_mirror >= 0) out.mirror_strength = stack[m.sh_mirror].f;
		if(m.sh_ior >= 0) out.glass_ior = m.ior_base + stack[m.sh_ior].f;
		return;
	}
	if(m.sh_diffuse >= 0)
	{
		const Rgba4 c = stack[m.sh_diffuse].col;
		out.diffuse_color[0] = c.r; out.diffuse_color[1] = c.g; out.diffuse_color[2] = c.b;
		// emit(): diffuse_shader->getColor(stack) * emit_strength_ (material_shiny_diffuse.cc:300); emit_strength = emit_color / colour is
		// not recoverable from the record, so the host keeps it in ior_base's neighbour `emit_strength`
		out.emit_color[0] = c.r * m.emit_strength; out.emit_color[1] = c.g * m.emit_strength; out.emit_color[2] = c.b * m.emit_strength;
	}
	if(m.sh_mirror_color >= 0) { const Rgba4 c = stack[m.sh_mirror_color].col; out.mirror_color[0] = c.r; out.mirror_color[1] = c.g; out.mirror_color[2] = c.b; }
	if(m.sh_mirror >= 0) out.mirror_strength = stack[m.sh_mirror].f;
	if(m.sh_transparency >= 0) out.transparency_strength = stack[m.sh_transparency].f;
	if(m.sh_translucency >= 0) out.translucency_strength = stack[m.sh_translucency].f;
	if(m.sh_sigma_oren >= 0)
	{
		const double sigma = (double)stack[m.sh_sigma_oren].f, s2 = sigma * sigma;
		out.oren_tex = 1; out.oren_ad = 1.0 - 0.5 * (s2 / (s2 + 0.33)); out.oren_bd = 0.45 * s2 / (s2 + 0.09);
	}
	if(m.sh_diffuse_refl >= 0) { out.has_diffuse_refl = 1; out.diffuse_refl = stack[m.sh_diffuse_refl].f; }
	if(m.sh_ior >= 0) { const float cur = m.ior_base + stack[m.sh_ior].f; out.ior_squared = cur * cur; }
}

} // namespace yafgpu
