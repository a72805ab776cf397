// Device-side materials, lights and camera of the MI355X path-tracing core.
// Restated from the reference's virtual Material/Light/Camera implementations as switch-on-type
// inline functions over POD records (yafgpu.h) — no vtables, no per-hit scratch allocation.
#pragma once
#include "yafgpu_math.h"

// Material entry points are called from several places of the shading kernels; YG_MAT decides whether every
// call site gets its own inlined copy (YAFGPU_NOINLINE_MAT=0) or shares one body (=1, smaller instruction footprint).
#ifndef YAFGPU_NOINLINE_MAT
#define YAFGPU_NOINLINE_MAT 0
#endif
#if YAFGPU_NOINLINE_MAT
#define YG_MAT __device__ __attribute__((noinline))
#else
#define YG_MAT __device__ __forceinline__
#endif
#include "../../include/yafgpu.h"

namespace yafgpu {

// BsdfFlags, include/material/material.h:49-64
// Material types this compilation of the shading code handles: a kernel built for scenes without some material types
// (YAFGPU_MAT_MASK, one bit per YAFGPU_MAT_*) drops their code and the registers it would pin.
#ifndef YAFGPU_MAT_MASK
#define YAFGPU_MAT_MASK 0x7fu
#endif
#define YG_IS(m, T) ((((YAFGPU_MAT_MASK) >> (T)) & 1u) != 0u && (m).type == (T))

enum : uint32_t {
	kNone = 0, kSpecular = 1, kGlossy = 2, kDiffuse = 4, kDispersive = 8, kReflect = 0x10, kTransmit = 0x20,
	kFilter = 0x40, kEmit = 0x80, kVolumetric = 0x100,
	kAll = kSpecular | kGlossy | kDiffuse | kDispersive | kReflect | kTransmit | kFilter
};

struct SurfPt { V3 p, n, ng, nu, nv; int mat; };                     // the live part of SurfacePoint (surface.h:58-100)
struct BsdfDat { float c0, c1, c2, c3; float m_diffuse, m_glossy, p_diffuse; }; // SdDat / MDatT
struct BsdfSample { float s_1, s_2, pdf; uint32_t flags, sampled; }; // Sample, material.h:68-78

YG_DEV Col col3(const float *p) { return mkc(p[0], p[1], p[2]); }
YG_DEV V3 vec3(const float *p) { return mk(p[0], p[1], p[2]); }
YG_DEV V3 face_forward(V3 ng, V3 n, V3 i) { return (dot(ng, i) < 0.f) ? -n : n; } // material.h:33

// ShinyDiffuseMaterial::getFresnel, material_shiny_diffuse.cc:119-147
YG_DEV float sd_fresnel(const yafgpu_material &m, V3 wo, V3 n)
{
	if(!m.has_fresnel) return 1.f;
	const V3 N = (dot(wo, n) < 0.f) ? -n : n;
	const float c = dot(wo, N);
	float g = m.ior_squared + c * c - 1.f;
	if(g < 0.f) g = 0.f;
	else g = f_sqrt(g);
	const float aux = c * (g + c);
	return ((0.5f * (g - c) * (g - c)) / ((g + c) * (g + c))) * (1.f + ((aux - 1.f) * (aux - 1.f)) / ((aux + 1.f) * (aux + 1.f)));
}
// accumulate__, :152-161
YG_DEV void sd_accumulate(const BsdfDat &d, float kr, float acc_out[4])
{
	acc_out[0] = d.c0 * kr;
	float acc = 1.f - acc_out[0];
	acc_out[1] = d.c1 * acc;
	acc *= 1.f - d.c1;
	acc_out[2] = d.c2 * acc;
	acc *= 1.f - d.c2;
	acc_out[3] = d.c3 * acc;
}
// orenNayar, material_shiny_diffuse.cc:204-241 == material_glossy.cc:74-111
YG_DEV float oren_nayar(float oa, float ob, V3 wi, V3 wo, V3 n)
{
	const float cos_ti = smax(-1.f, smin(1.f, dot(n, wi)));
	const float cos_to = smax(-1.f, smin(1.f, dot(n, wo)));
	float maxcos_f = 0.f;
	if(cos_ti < 0.9999f && cos_to < 0.9999f)
	{
		const V3 v_1 = normalize(wi - n * cos_ti);
		const V3 v_2 = normalize(wo - n * cos_to);
		maxcos_f = smax(0.f, dot(v_1, v_2));
	}
	float sin_alpha, tan_beta;
	if(cos_to >= cos_ti)
	{
		sin_alpha = f_sqrt(1.f - cos_ti * cos_ti);
		tan_beta = f_sqrt(1.f - cos_to * cos_to) / ((cos_to == 0.f) ? 1e-8f : cos_to);
	}
	else
	{
		sin_alpha = f_sqrt(1.f - cos_to * cos_to);
		tan_beta = f_sqrt(1.f - cos_ti * cos_ti) / ((cos_ti == 0.f) ? 1e-8f : cos_ti);
	}
	return smin(1.f, smax(0.f, (oa + ob * maxcos_f * sin_alpha * tan_beta)));
}

// orenNayar with a texture's sigma (material_shiny_diffuse.cc:230-235): A and B, and the sum, in double
YG_DEV float oren_nayar_d(double oa, double ob, V3 wi, V3 wo, V3 n)
{
	const float cos_ti = smax(-1.f, smin(1.f, dot(n, wi)));
	const float cos_to = smax(-1.f, smin(1.f, dot(n, wo)));
	float maxcos_f = 0.f;
	if(cos_ti < 0.9999f && cos_to < 0.9999f)
	{
		const V3 v_1 = normalize(wi - n * cos_ti);
		const V3 v_2 = normalize(wo - n * cos_to);
		maxcos_f = smax(0.f, dot(v_1, v_2));
	}
	float sin_alpha, tan_beta;
	if(cos_to >= cos_ti)
	{
		sin_alpha = f_sqrt(1.f - cos_ti * cos_ti);
		tan_beta = f_sqrt(1.f - cos_to * cos_to) / ((cos_to == 0.f) ? 1e-8f : cos_to);
	}
	else
	{
		sin_alpha = f_sqrt(1.f - cos_to * cos_to);
		tan_beta = f_sqrt(1.f - cos_ti * cos_ti) / ((cos_ti == 0.f) ? 1e-8f : cos_ti);
	}
	return smin(1.f, smax(0.f, (float)(oa + ob * (double)maxcos_f * (double)sin_alpha * (double)tan_beta)));
}
#ifndef YAFGPU_FEAT_TEXTURE
#define YAFGPU_FEAT_TEXTURE 1      // shader nodes / image textures (a kernel built with 0 serves scenes without them)
#endif
YG_DEV float sd_oren(const yafgpu_material &m, V3 wi, V3 wo, V3 n)
{
	if(YAFGPU_FEAT_TEXTURE && m.oren_tex) return oren_nayar_d(m.oren_ad, m.oren_bd, wi, wo, n);
	return oren_nayar(m.oren_a, m.oren_b, wi, wo, n);
}

// initBsdf: material_shiny_diffuse.cc:163-183 (+getComponents :98-117), material_glossy.cc:51-64
YG_DEV uint32_t mat_init_bsdf(const yafgpu_material &m, BsdfDat &d)
{
	d.c0 = d.c1 = d.c2 = d.c3 = 0.f; d.m_diffuse = d.m_glossy = d.p_diffuse = 0.f;
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE))
	{
		if(m.is_mirror) d.c0 = m.mirror_strength;
		if(m.is_transparent) d.c1 = m.transparency_strength;
		if(m.is_translucent) d.c2 = m.translucency_strength;
		if(m.is_diffuse) d.c3 = m.diffuse_strength;
	}
	else if(YG_IS(m, YAFGPU_MAT_GLOSSY) || YG_IS(m, YAFGPU_MAT_COATED_GLOSSY))
	{
		d.m_diffuse = m.diffuse;
		d.m_glossy = m.reflectivity;
		d.p_diffuse = smin(0.6f, 1.f - (d.m_glossy / (d.m_glossy + (1.f - d.m_glossy) * d.m_diffuse)));
	}
	return m.bsdf_flags;
}

// material_utils_microfacet.h
YG_DEV float blinn_d(float cos_h, float e) { return (e + 1.f) * f_pow(cos_h, e); }                                   // :89-92
YG_DEV double pdf_divisor(float c) { return (double)8.f * kPi * (double)(c * 0.99f + 0.04f); }                        // :35
YG_DEV float blinn_pdf(float ct, float cwh, float e) { return (float)((double)blinn_d(ct, e) / pdf_divisor(cwh)); }   // :94-97
YG_DEV double as_divisor(float c1, float ci, float co) { return (double)8.f * kPi * (double)((c1 * smax(ci, co)) * 0.99f + 0.04f); } // :36
YG_DEV float schlick_fresnel(float ct, float r) // :188-193
{
	const float c_1 = (1.f - ct);
	const float c_2 = c_1 * c_1;
	return r + ((1.f - r) * c_1 * c_2 * c_2);
}
YG_DEV Col diffuse_reflect(float wi_n, float wo_n, float glossy, float diffuse, Col base) // :195-206
{
	float f_wi = (1.f - (0.5f * wi_n));
	float t = f_wi * f_wi;
	f_wi = t * t * f_wi;
	float f_wo = (1.f - (0.5f * wo_n));
	t = f_wo * f_wo;
	f_wo = t * t * f_wo;
	const double k = 0.387507688 * (double)diffuse * (double)(1.f - glossy) * (double)(1.f - f_wi) * (double)(1.f - f_wo);
	return base * (float)k;
}
YG_DEV V3 blinn_sample(float s_1, float s_2, float e) // :99-106
{
	const float cos_theta = f_pow(1.f - s_2, 1.f / (e + 1.f));
	const float sin_theta = f_sqrt(1.f - cos_theta * cos_theta);
	const float phi = (float)((double)s_1 * k2Pi);
	return mk(sin_theta * f_cos(phi), sin_theta * f_sin(phi), cos_theta);
}

// BeerVolumeHandler::transmittance over a ray that ended at tmax (volumehandler_beer.cc:37-48); fExp__(x) = fExp2__(M_LOG2E * x)
YG_DEV Col beer_transmittance(const float sigma[3], float tmax)
{
	if(tmax < 0.f || tmax > 1e30f) return mkc(0.f, 0.f, 0.f);
	const float l2e = (float)1.4426950408889634074;
	return mkc(f_exp2(l2e * (-tmax * sigma[0])), f_exp2(l2e * (-tmax * sigma[1])), f_exp2(l2e * (-tmax * sigma[2])));
}

// refract__, vector.cc:86-108
YG_DEV bool refract_dir(V3 n, V3 wi, V3 &wo, float ior)
{
	V3 N = n;
	float eta = ior;
	const V3 i = -wi;
	float cos_v_n = dot(wi, n);
	if(cos_v_n < 0.f) { N = -n; cos_v_n = -cos_v_n; }
	else eta = (float)(1.0 / (double)ior);
	const float k = 1.f - eta * eta * (1.f - cos_v_n * cos_v_n);
	if(k <= 0.f) return false;
	wo = normalize(i * eta + N * (eta * cos_v_n - f_sqrt(k)));
	return true;
}
// fresnel__, vector.cc:110-142 (kr is formed in double)
YG_DEV void fresnel_dielectric(V3 i, V3 n, float ior, float &kr, float &kt)
{
	const float eta = ior;
	const V3 N = (dot(i, n) < 0.f) ? -n : n;
	const float c = dot(i, N);
	float g = eta * eta + c * c - 1.f;
	g = (g <= 0.f) ? 0.f : f_sqrt(g);
	const float aux = c * (g + c);
	kr = (float)(((0.5 * (double)(g - c) * (double)(g - c)) / (double)((g + c) * (g + c))) *
	             (double)(1.f + ((aux - 1.f) * (aux - 1.f)) / ((aux + 1.f) * (aux + 1.f))));
	kt = (kr < 1.0f) ? 1.f - kr : 0.f;
}
// the shading normal the glass uses, material_glass.cc:77-80, 263-271
YG_DEV V3 glass_normal(const SurfPt &sp, V3 wo)
{
	const bool outside = dot(sp.ng, wo) > 0.f;
	const float cos_wo_n = dot(sp.n, wo);
	if(outside ? (cos_wo_n >= 0.f) : (cos_wo_n <= 0.f)) return sp.n;
	const float f = (float)(1.00001 * (double)cos_wo_n);
	return normalize(sp.n - wo * f);
}
YG_DEV V3 vec_reflect(V3 v, V3 n)      // Vec3::reflect, vector.h:291-298
{
	const float vn = 2.0f * (v.x * n.x + v.y * n.y + v.z * n.z);
	return mk(vn * n.x - v.x, vn * n.y - v.y, vn * n.z - v.z);
}
// the anisotropic Ashikhmin-Shirley lobe, material_utils_microfacet.h:38-87.  tanf: the reference calls libm's; here the double
// tan narrowed once, which agrees with a correctly rounded tanf except in double-rounding ties (glibc's is within 1 ulp)
YG_DEV V3 sample_quadrant_aniso(float s_1, float s_2, float e_u, float e_v) // :38-51
{
	const float t = (float)tan((double)(float)(1.57079632679489661923 * (double)s_1));
	const float phi = (float)atan((double)(f_sqrt((e_u + 1.f) / (e_v + 1.f)) * t));
	const float cos_phi = f_cos(phi), sin_phi = f_sin(phi);
	const float cos_phi_2 = cos_phi * cos_phi;
	const float sin_phi_2 = 1.f - cos_phi_2;
	const float cos_theta = f_pow(1.f - s_2, 1.f / (e_u * cos_phi_2 + e_v * sin_phi_2 + 1.f));
	const float sin_theta = f_sqrt(1.f - cos_theta * cos_theta);
	return mk(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
YG_DEV float as_aniso_d(V3 h, float e_u, float e_v) // :53-58
{
	if(h.z <= 0.f) return 0.f;
	const float exponent = (e_u * h.x * h.x + e_v * h.y * h.y) / (1.00001f - h.z * h.z);
	return f_sqrt((e_u + 1.f) * (e_v + 1.f)) * f_pow(smax(0.f, h.z), exponent);
}
YG_DEV V3 as_aniso_sample(float s_1, float s_2, float e_u, float e_v) // :65-87
{
	V3 h;
	if(s_1 < 0.25f) h = sample_quadrant_aniso(4.f * s_1, s_2, e_u, e_v);
	else if(s_1 < 0.5f) { h = sample_quadrant_aniso(1.f - 4.f * (0.5f - s_1), s_2, e_u, e_v); h.x = -h.x; }
	else if(s_1 < 0.75f) { h = sample_quadrant_aniso(4.f * (s_1 - 0.5f), s_2, e_u, e_v); h.x = -h.x; h.y = -h.y; }
	else { h = sample_quadrant_aniso(1.f - 4.f * (1.f - s_1), s_2, e_u, e_v); h.y = -h.y; }
	return h;
}
// the material's glossy lobe: Blinn on cos(n, h), or the anisotropic lobe on h in the shading frame (hs)
#ifndef YAFGPU_FEAT_ANISO
#define YAFGPU_FEAT_ANISO 1      // the specialised shading kernels are built without the anisotropic lobe (a scene that has one runs the general kernel)
#endif
YG_DEV float lobe_d(const yafgpu_material &m, V3 hs, float cos_n_h) { return (YAFGPU_FEAT_ANISO && m.anisotropic) ? as_aniso_d(hs, m.exp_u, m.exp_v) : blinn_d(cos_n_h, m.exponent); }
YG_DEV float lobe_pdf(const yafgpu_material &m, V3 hs, float cos_n_h, float cos_w_h) { return (float)((double)lobe_d(m, hs, cos_n_h) / pdf_divisor(cos_w_h)); }
YG_DEV V3 lobe_sample(const yafgpu_material &m, float s_1, float s_2) { return (YAFGPU_FEAT_ANISO && m.anisotropic) ? as_aniso_sample(s_1, s_2, m.exp_u, m.exp_v) : blinn_sample(s_1, s_2, m.exponent); }
YG_DEV V3 local_h(const yafgpu_material &m, const SurfPt &sp, V3 h, float cos_n_h) { return (YAFGPU_FEAT_ANISO && m.anisotropic) ? mk(dot(h, sp.nu), dot(h, sp.nv), cos_n_h) : mk(0.f, 0.f, cos_n_h); }

// Material::eval — material_shiny_diffuse.cc:244-293, material_glossy.cc:113-173
YG_MAT Col mat_eval(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo, V3 wl, uint32_t bsdfs)
{
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE))
	{
		const float cos_ng_wo = dot(sp.ng, wo), cos_ng_wl = dot(sp.ng, wl);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		if(!(bsdfs & m.bsdf_flags & kDiffuse)) return mkc(0.f, 0.f, 0.f);
		const float kr = sd_fresnel(m, wo, n);
		const float m_t = (1.f - kr * d.c0) * (1.f - d.c1);
		if((cos_ng_wo * cos_ng_wl) < 0.f)
		{
			if(m.is_translucent) return col3(m.diffuse_color) * (d.c2 * m_t);
		}
		if(dot(n, wl) < 0.f && !m.flat) return mkc(0.f, 0.f, 0.f);
		float m_d = m_t * (1.f - d.c2) * d.c3;
		if(m.use_oren) m_d *= sd_oren(m, wo, wl, n);
		if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) m_d *= m.diffuse_refl;            // diffuse_refl_shader_, :285
		return col3(m.diffuse_color) * m_d;
	}
	if(YG_IS(m, YAFGPU_MAT_GLOSSY))
	{
		if(!(bsdfs & kDiffuse) || (dot(sp.ng, wl) * dot(sp.ng, wo)) < 0.f) return mkc(0.f, 0.f, 0.f);
		Col col = mkc(0.f, 0.f, 0.f);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		const float wi_n = fabsf(dot(wl, n)), wo_n = fabsf(dot(wo, n));
		if(m.as_diffuse || (bsdfs & kGlossy))
		{
			const V3 h = normalize(wo + wl);
			const float cos_wi_h = smax(0.f, dot(wl, h));
			const float cos_n_h = dot(h, n);
			const float glossy = (float)((double)(lobe_d(m, local_h(m, sp, h, cos_n_h), cos_n_h) * schlick_fresnel(cos_wi_h, d.m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
			col = col3(m.gloss_color) * glossy;
		}
		if(m.with_diffuse)
		{
			Col add = col3(m.diff_color) * (d.m_diffuse * (1.f - d.m_glossy));
			if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) add = add * m.diffuse_refl;      // diffuse_reflection_shader_
			if(m.use_oren) add = add * sd_oren(m, wl, wo, n);
			col = col + add;
		}
		return col;
	}
	if(YG_IS(m, YAFGPU_MAT_COATED_GLOSSY))
	{	// material_coated_glossy.cc:130-186
		Col col = mkc(0.f, 0.f, 0.f);
		const bool diffuse_flag = (bsdfs & kDiffuse) != 0u;
		if(!diffuse_flag || (dot(sp.ng, wl) * dot(sp.ng, wo)) < 0.f) return col;
		const V3 n = face_forward(sp.ng, sp.n, wo);
		float kr, kt;
		const float wi_n = fabsf(dot(wl, n)), wo_n = fabsf(dot(wo, n));
		fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
		if((m.as_diffuse && diffuse_flag) || (!m.as_diffuse && (bsdfs & kGlossy)))
		{
			const V3 h = normalize(wo + wl);
			const float cos_wi_h = dot(wl, h);
			const float cos_n_h = dot(h, n);
			const float glossy = (float)((double)(kt * lobe_d(m, local_h(m, sp, h, cos_n_h), cos_n_h) * schlick_fresnel(cos_wi_h, d.m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
			col = col3(m.gloss_color) * glossy;
		}
		if(m.with_diffuse && diffuse_flag)
		{
			Col add = (col3(m.diff_color) * (d.m_diffuse * (1.f - d.m_glossy))) * kt;
			if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) add = add * m.diffuse_refl;      // diffuse_reflection_shader_
			if(m.use_oren) add = add * sd_oren(m, wl, wo, n);
			col = col + add;
		}
		return col;
	}
	return mkc(0.f, 0.f, 0.f);
}

// Material::pdf — material_shiny_diffuse.cc:410-460, material_glossy.cc:359-405
YG_MAT float mat_pdf(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo, V3 wi, uint32_t bsdfs)
{
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE))
	{
		if(!(bsdfs & kDiffuse)) return 0.f;
		float pdf = 0.f, acc[4];
		const float cos_ng_wo = dot(sp.ng, wo);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		sd_accumulate(d, sd_fresnel(m, wo, n), acc);
		float sum = 0.f;
		int n_match = 0;
		for(int i = 0; i < m.n_bsdf; ++i)
		{
			if(bsdfs & m.c_flags[i])
			{
				const float width = acc[m.c_index[i]];
				sum += width;
				if(m.c_flags[i] == (kDiffuse | kTransmit))
				{
					if(cos_ng_wo * dot(sp.ng, wi) < 0.f) pdf += fabsf(dot(wi, n)) * width;
				}
				else if(m.c_flags[i] == (kDiffuse | kReflect)) pdf += fabsf(dot(wi, n)) * width;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) return 0.f;
		return pdf / sum;
	}
	if(YG_IS(m, YAFGPU_MAT_GLOSSY))
	{
		if(dot(sp.ng, wo) * dot(sp.ng, wi) < 0.f) return 0.f;
		const V3 n = face_forward(sp.ng, sp.n, wo);
		float pdf = 0.f;
		const bool use_glossy = m.as_diffuse ? (bsdfs & kDiffuse) != 0 : (bsdfs & kGlossy) != 0;
		const bool use_diffuse = m.with_diffuse && (bsdfs & kDiffuse);
		if(use_diffuse)
		{
			pdf = fabsf(dot(wi, n));
			if(use_glossy)
			{
				const V3 h = normalize(wi + wo);
				const float cos_n_h = dot(n, h);
				pdf = pdf * d.p_diffuse + lobe_pdf(m, local_h(m, sp, h, cos_n_h), cos_n_h, dot(wo, h)) * (1.f - d.p_diffuse);
			}
			return pdf;
		}
		if(use_glossy)
		{
			const V3 h = normalize(wi + wo);
			const float cos_n_h = dot(n, h);
			pdf = lobe_pdf(m, local_h(m, sp, h, cos_n_h), cos_n_h, dot(wo, h));
		}
		return pdf;
	}
	if(YG_IS(m, YAFGPU_MAT_COATED_GLOSSY))
	{	// material_coated_glossy.cc:376-424
		if((dot(sp.ng, wo) * dot(sp.ng, wi)) < 0.f) return 0.f;
		const V3 n = face_forward(sp.ng, sp.n, wo);
		float pdf = 0.f, kr, kt;
		fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
		const float acc[3] = {kr, kt * (1.f - d.p_diffuse), kt * d.p_diffuse};
		float sum = 0.f;
		int n_match = 0;
		for(int i = 0; i < m.n_bsdf; ++i)
		{
			if((bsdfs & m.c_flags[i]) == m.c_flags[i])
			{
				const float width = acc[i];
				sum += width;
				if(i == 1)
				{
					const V3 h = normalize(wi + wo);
					const float cos_n_h = dot(n, h);
					pdf += lobe_pdf(m, local_h(m, sp, h, cos_n_h), cos_n_h, dot(wo, h)) * width;
				}
				else if(i == 2) pdf += fabsf(dot(wi, n)) * width;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) return 0.f;
		return pdf / sum;
	}
	return 0.f;
}

// ShinyDiffuseMaterial::getAlpha, :568-597
YG_DEV float sd_alpha(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo)
{
	if(!m.is_transparent) return 1.f;
	const V3 n = face_forward(sp.ng, sp.n, wo);
	const float kr = sd_fresnel(m, wo, n);
	return 1.f - (1.f - d.c0 * kr) * d.c1;
}

// Material::isTransparent / getTransparency: ShinyDiffuse (material_shiny_diffuse.h:53, .cc:530-566), Glass with fake
// shadows (material_glass.cc:217-228); Material's default: opaque
YG_DEV bool mat_is_transparent(const yafgpu_material &m)
{
	return (YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE) && m.is_transparent) || ((YG_IS(m, YAFGPU_MAT_GLASS) || YG_IS(m, YAFGPU_MAT_ROUGH_GLASS)) && m.fake_shadow);
}
YG_DEV Col mat_transparency(const yafgpu_material &m, const SurfPt &sp, V3 wo)
{
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE))
	{
		if(!m.is_transparent) return mkc(0.f, 0.f, 0.f);
		float accum = 1.f;
		const V3 n = face_forward(sp.ng, sp.n, wo);
		const float kr = sd_fresnel(m, wo, n);
		if(m.is_mirror) accum = 1.f - kr * m.mirror_strength;
		accum *= m.transparency_strength * accum;                     // sic, :557
		const float f = m.transmit_filter;
		const Col tcol = col3(m.diffuse_color) * f + mkc(1.f - f, 1.f - f, 1.f - f);
		return tcol * accum;
	}
	if(YG_IS(m, YAFGPU_MAT_GLASS) || YG_IS(m, YAFGPU_MAT_ROUGH_GLASS))      // material_glass.cc:217-228, material_rough_glass.cc:288-299
	{
		const V3 n = face_forward(sp.ng, sp.n, wo);
		float kr, kt;
		fresnel_dielectric(wo, n, m.transp_ior, kr, kt);
		return col3(m.filter_color) * kt;
	}
	return mkc(0.f, 0.f, 0.f);
}

// Material::getAlpha: ShinyDiffuse (:568-597), Glass (material_glass.cc:217-240), everything else 1
YG_DEV float mat_alpha(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo)
{
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE)) return sd_alpha(m, d, sp, wo);
	if(YG_IS(m, YAFGPU_MAT_GLASS))
	{
		const V3 n = face_forward(sp.ng, sp.n, wo);
		float kr, kt;
		fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
		const Col t = col3(m.filter_color) * kt;
		const float alpha = (float)(1.0 - (double)((t.r + t.g + t.b) * 0.333333f));
		return alpha < 0.f ? 0.f : alpha;
	}
	if(YG_IS(m, YAFGPU_MAT_ROUGH_GLASS))
	{	// material_rough_glass.cc:301-310: max(0, min(1, 1 - getTransparency().energy()))
		const Col t = mat_transparency(m, sp, wo);
		return smax(0.f, smin(1.f, 1.f - (t.r + t.g + t.b) * 0.333333f));
	}
	return 1.f;
}

// ShinyDiffuseMaterial::getSpecular, material_shiny_diffuse.cc:474-528 (no shader nodes, no wireframe): the perfect
// reflection and the filtered straight-through transmission recursiveRaytrace follows.  Every other material of this
// path keeps Material::getSpecular's default (neither).
// GlassMaterial::getSpecular (material_glass.cc:242-340, no dispersion) and MirrorMaterial::getSpecular (:475-484) too.
// raylevel: RenderState::raylevel_ as getSpecular sees it — recursiveRaytrace has already incremented it.
YG_DEV void mat_get_specular(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo, int raylevel, bool &do_reflect, bool &do_refract,
                             V3 &dir_reflect, Col &col_reflect, V3 &dir_refract, Col &col_refract)
{
	do_reflect = false; do_refract = false;
	dir_reflect = mk(0.f, 0.f, 0.f); dir_refract = dir_reflect; col_reflect = mkc(0.f, 0.f, 0.f); col_refract = col_reflect;
	if(YG_IS(m, YAFGPU_MAT_GLASS))
	{
		const bool outside = dot(sp.ng, wo) > 0.f;
		const V3 n = glass_normal(sp, wo);
		V3 refdir;
		if(refract_dir(n, wo, refdir, m.glass_ior))
		{
			float kr, kt;
			fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
			col_refract = col3(m.filter_color) * kt; dir_refract = refdir; do_refract = true;
			// the reflection of a ray leaving the glass is only followed near the top of the recursion (:324-332)
			if(outside || raylevel < 3) { dir_reflect = vec_reflect(wo, n); col_reflect = col3(m.mirror_color) * kr; do_reflect = true; }
		}
		else { col_reflect = col3(m.mirror_color); dir_reflect = vec_reflect(wo, n); do_reflect = true; }      // total inner reflection
		return;
	}
	if(YG_IS(m, YAFGPU_MAT_COATED_GLOSSY))
	{	// CoatedGlossyMaterial::getSpecular, material_coated_glossy.cc:426-462
		const bool outside = dot(sp.ng, wo) >= 0.f;
		const float cos_wo_n = dot(sp.n, wo);
		const V3 ng = outside ? sp.ng : -sp.ng;
		V3 n = sp.n;
		if(!(outside ? (cos_wo_n >= 0.f) : (cos_wo_n <= 0.f))) n = normalize(sp.n - wo * (float)(1.00001 * (double)cos_wo_n));
		float kr, kt;
		fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
		if(raylevel > 5) return;
		V3 r = vec_reflect(wo, n);
		col_reflect = (col3(m.mirror_color) * kr) * m.mirror_strength;
		const float cos_wi_ng = dot(r, ng);
		if((double)cos_wi_ng < 0.01) r = normalize(r + ng * (float)(0.01 - (double)cos_wi_ng));
		dir_reflect = r;
		do_reflect = true;
		return;
	}
	if(YG_IS(m, YAFGPU_MAT_MIRROR))
	{
		col_reflect = col3(m.mirror_color);
		dir_reflect = reflect_dir(face_forward(sp.ng, sp.n, wo), wo);
		do_reflect = true;
		return;
	}
	if(m.type != YAFGPU_MAT_SHINYDIFFUSE) return;
	const bool backface = dot(wo, sp.ng) < 0.f;
	const V3 n = backface ? -sp.n : sp.n;
	const V3 ng = backface ? -sp.ng : sp.ng;
	const float kr = sd_fresnel(m, wo, n);
	if(m.is_transparent)
	{
		do_refract = true;
		dir_refract = -wo;
		const float f = m.transmit_filter;
		const Col tcol = col3(m.diffuse_color) * f + mkc(1.f - f, 1.f - f, 1.f - f);
		col_refract = tcol * ((1.f - d.c0 * kr) * d.c1);
	}
	if(m.is_mirror)
	{
		do_reflect = true;
		const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);        // Vec3::reflect, vector.h:291-298
		V3 r = mk(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
		const float cos_wi_ng = dot(r, ng);
		if((double)cos_wi_ng < 0.01)
		{
			const float k = (float)(0.01 - (double)cos_wi_ng);
			r = normalize(r + ng * k);
		}
		dir_reflect = r;
		col_reflect = col3(m.mirror_color) * (d.c0 * kr);
	}
}

// GGX microfacet helpers, material_utils_microfacet.h:108-185
YG_DEV V3 ggx_sample(float alpha_2, float s_1, float s_2)                                // :111-121
{
	const float tan_theta_2 = alpha_2 * (s_1 / (1.00001f - s_1));
	const float cos_theta = 1.f / f_sqrt(1.f + tan_theta_2);
	const float sin_theta = f_sqrt(1.00001f - (cos_theta * cos_theta));
	const float phi = (float)(k2Pi * (double)s_2);
	return mk(sin_theta * f_cos(phi), sin_theta * f_sin(phi), cos_theta);
}
YG_DEV float ggx_d(float alpha_2, float cos_theta_2, float tan_theta_2)                 // :123-129: M_PI makes the divisor a double product
{
	const float cos_theta_4 = cos_theta_2 * cos_theta_2;
	const float a_tan = alpha_2 + tan_theta_2;
	const float div = (float)(((kPi * (double)cos_theta_4) * (double)a_tan) * (double)a_tan);
	return alpha_2 / div;
}
YG_DEV float ggx_g(float alpha_2, float wo_n, float wi_n)                                // :131-143
{
	const float wo_n_2 = wo_n * wo_n, wi_n_2 = wi_n * wi_n;
	const float sqr_term_1 = f_sqrt(1.f + alpha_2 * ((1.f - wo_n_2) / wo_n_2));
	const float sqr_term_2 = f_sqrt(1.f + alpha_2 * ((1.f - wi_n_2) / wi_n_2));
	const float g_1_wo = 2.f / (1.f + (sqr_term_1));
	const float g_1_wi = 2.f / (1.f + (sqr_term_2));
	return g_1_wo * g_1_wi;
}
YG_DEV float ggx_pdf(float d, float cos_theta, float jacobian) { return d * cos_theta * jacobian; }      // :145-148
YG_DEV float microfacet_fresnel(float wo_h, float ior)                                   // :150-162
{
	const float c = fabsf(wo_h);
	float g = ior * ior - 1.f + c * c;
	if(g > 0.f)
	{
		g = f_sqrt(g);
		const float a = (g - c) / (g + c);
		const float b = (c * (g + c) - 1.f) / (c * (g - c) + 1.f);
		return 0.5f * a * a * (1.f + b * b);
	}
	return 1.0f;
}
YG_DEV bool refract_microfacet(float eta, V3 wo, V3 &wi, V3 h, float wo_h, float &kr, float &kt)      // :164-179
{
	wi = mk(0.f, 0.f, 0.f);
	const float c = dot(-wo, h);
	const float sign = (c > 0.f) ? 1.f : -1.f;
	const float t_1 = 1.f - (eta * eta * (1.f - c * c));
	if(t_1 < 0.f) return false;
	wi = wo * eta + h * (eta * c - sign * f_sqrt(t_1));
	wi = -wi;
	kr = 0.f; kt = 0.f;
	kr = microfacet_fresnel(wo_h, 1.f / eta);
	if(kr == 1.f) return false;
	kt = 1.f - kr;
	return true;
}
YG_DEV V3 reflect_microfacet(V3 wo, V3 h) { const V3 wi = wo + h * (2.f * dot(h, -wo)); return -wi; }      // :181-185
// RoughGlassMaterial::sample, both forms (material_rough_glass.cc:62-163 one direction, :165-286 two): the half vector of the GGX lobe,
// refraction and reflection about it.  two == false: the lobe is picked by s_1 against kt, (dir0, w0) out.  two == true: both directions —
// the reference writes the TRANSMITTED one to dir[0] / w[0] with the returned colour and the REFLECTED one to dir[1] / w[1] with tcol, which
// recursiveRaytrace then reads the other way round (integrator_montecarlo.cc:925-958): restated as it is.
YG_MAT Col rough_glass_sample(const yafgpu_material &m, const SurfPt &sp, V3 wo, BsdfSample &s, bool two, V3 &dir0, float &w0, V3 &dir1, Col &tcol, float &w1)
{
	const V3 n = face_forward(sp.ng, sp.n, wo);
	const bool outside = dot(sp.ng, wo) > 0.f;
	s.pdf = 1.f;
	const float alpha_2 = m.rg_a2;
	V3 h = ggx_sample(alpha_2, s.s_1, s.s_2);
	h = (sp.nu * h.x + sp.nv * h.y) + n * h.z;
	h = normalize(h);
	const float cur_ior = m.glass_ior;
	float glossy, glossy_d = 0.f, glossy_g = 0.f, wi_n, wi_h, jacobian = 0.f;
	const float cos_theta = dot(h, n);
	const float cos_theta_2 = cos_theta * cos_theta;
	const float tan_theta_2 = (1.f - cos_theta_2) / smax(1.0e-8f, cos_theta_2);
	if(cos_theta > 0.f) glossy_d = ggx_d(alpha_2, cos_theta_2, tan_theta_2);
	const float wo_h = dot(wo, h), wo_n = dot(wo, n);
	float kr, kt;
	Col ret = mkc(0.f, 0.f, 0.f);
	V3 wi;
	if(two) s.sampled = 0u;
	if(refract_microfacet(outside ? 1.f / cur_ior : cur_ior, wo, wi, h, wo_h, kr, kt))
	{
		const bool take_t = two ? (s.flags & kTransmit) != 0u : (s.s_1 < kt && (s.flags & kTransmit) != 0u);
		if(take_t)
		{
			wi_n = dot(wi, n); wi_h = dot(wi, h);
			if((wi_h * wi_n) > 0.f && (wo_h * wo_n) > 0.f) glossy_g = ggx_g(alpha_2, wi_n, wo_n);
			float ior_wi = 1.f, ior_wo = 1.f;
			if(outside) ior_wi = cur_ior; else ior_wo = cur_ior;
			const float ht = ior_wo * wo_h + ior_wi * wi_h;
			jacobian = (ior_wi * ior_wi) / smax(1.0e-8f, ht * ht);
			glossy = fabsf((wo_h * wi_h) / (wi_n * wo_n)) * kt * glossy_g * glossy_d * jacobian;
			s.pdf = ggx_pdf(glossy_d, cos_theta, jacobian * fabsf(wi_h));
			s.sampled = kGlossy | kTransmit;
			ret = col3(m.filter_color) * glossy;
			w0 = fabsf(wi_n) / smax(0.1f, s.pdf);
			dir0 = wi;
		}
		if(two ? (s.flags & kReflect) != 0u : (!take_t && (s.flags & kReflect) != 0u))
		{
			wi = reflect_microfacet(wo, h);
			wi_n = dot(wi, n); wi_h = dot(wi, h);
			glossy_g = ggx_g(alpha_2, wi_n, wo_n);
			jacobian = 1.f / smax(1.0e-8f, (4.f * fabsf(wi_h)));
			glossy = (kr * glossy_g * glossy_d) / smax(1.0e-8f, (4.f * fabsf(wo_n * wi_n)));
			s.pdf = ggx_pdf(glossy_d, cos_theta, jacobian);
			if(two) s.sampled |= kGlossy | kReflect; else s.sampled = kGlossy | kReflect;
			const Col rc = col3(m.mirror_color) * glossy;
			const float ww = fabsf(wi_n) / smax(0.1f, s.pdf);
			if(two) { tcol = rc; w1 = ww; dir1 = wi; }
			else { ret = rc; w0 = ww; dir0 = wi; }
		}
		else if(!two && !take_t) dir0 = wi;      // neither lobe asked for: wi is what refractMicrofacet__ left, w untouched
	}
	else
	{	// total inner reflection about the half vector
		wi = vec_reflect(wo, h);
		if(two) s.sampled |= kGlossy | kReflect; else s.sampled = kGlossy | kReflect;
		dir0 = wi;
		ret = mkc(1.f, 1.f, 1.f);
		w0 = 1.f;
	}
	return ret;
}

// Material::sample — material_shiny_diffuse.cc:308-408, material_glossy.cc:176-357 (Blinn branch),
// material_simple.cc:41-46
YG_MAT Col mat_sample(const yafgpu_material &m, const BsdfDat &d, const SurfPt &sp, V3 wo, V3 &wi, BsdfSample &s, float &w)
{
	if(YG_IS(m, YAFGPU_MAT_ROUGH_GLASS))
	{
		V3 d1 = wi; Col tc = mkc(0.f, 0.f, 0.f); float w1 = w;
		return rough_glass_sample(m, sp, wo, s, false, wi, w, d1, tc, w1);
	}
	if(YG_IS(m, YAFGPU_MAT_GLASS))
	{	// GlassMaterial::sample, material_glass.cc:65-215, the branch without dispersion (:143-213)
		if(!(s.flags & kSpecular)) { s.pdf = 0.f; return mkc(0.f, 0.f, 0.f); }
		const V3 n = glass_normal(sp, wo);
		V3 refdir;
		s.pdf = 1.f;
		const uint32_t spec_refl = kSpecular | kReflect;
		if(refract_dir(n, wo, refdir, m.glass_ior))
		{
			float kr, kt;
			fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
			const float p_kr = (float)(0.01 + 0.99 * (double)kr), p_kt = (float)(0.01 + 0.99 * (double)kt);
			if(s.s_1 < p_kt && ((s.flags & m.tm_flags) == m.tm_flags))
			{
				wi = refdir; s.pdf = p_kt; s.sampled = m.tm_flags; w = 1.f;
				return col3(m.filter_color);
			}
			else if((s.flags & spec_refl) == spec_refl)
			{
				wi = vec_reflect(wo, n); s.pdf = p_kr; s.sampled = spec_refl; w = 1.f;
				return col3(m.mirror_color);
			}
		}
		else if((s.flags & spec_refl) == spec_refl)
		{	// total inner reflection
			wi = vec_reflect(wo, n); s.sampled = spec_refl; w = 1.f;
			return mkc(1.f, 1.f, 1.f);
		}
		s.pdf = 0.f;
		return mkc(0.f, 0.f, 0.f);
	}
	if(YG_IS(m, YAFGPU_MAT_COATED_GLOSSY))
	{	// material_coated_glossy.cc:188-374, Blinn lobe
		const float cos_ng_wo = dot(sp.ng, wo);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		V3 hs = mk(0.f, 0.f, 0.f);
		s.pdf = 0.f;
		float kr, kt;
		fresnel_dielectric(wo, n, m.glass_ior, kr, kt);
		const float acc[3] = {kr, kt * (1.f - d.p_diffuse), kt * d.p_diffuse};
		bool use1 = false, use2 = false;
		float sum = 0.f, val[3], width[3];
		int c_index[3] = {0, 0, 0}, rc1 = 0, rc2 = 0;
		int n_match = 0, pick = -1;
		for(int i = 0; i < m.n_bsdf; ++i)
		{
			if((s.flags & m.c_flags[i]) == m.c_flags[i])
			{
				if(i == 1) { use1 = true; rc1 = n_match; }
				if(i == 2) { use2 = true; rc2 = n_match; }
				width[n_match] = acc[i];
				c_index[n_match] = i;
				sum += width[n_match];
				val[n_match] = sum;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) { wi = reflect_dir(n, wo); return mkc(0.f, 0.f, 0.f); }
		else if(n_match == 1) { pick = 0; width[0] = 1.f; }
		else
		{
			const float inv_sum = 1.f / sum;
			for(int i = 0; i < n_match; ++i)
			{
				val[i] *= inv_sum;
				width[i] *= inv_sum;
				if((s.s_1 <= val[i]) && (pick < 0)) pick = i;
			}
		}
		if(pick < 0) pick = n_match - 1;
		float w_pick = width[0], v_prev = 0.f; int ci = c_index[0];
		if(pick == 1) { w_pick = width[1]; v_prev = val[0]; ci = c_index[1]; }
		if(pick == 2) { w_pick = width[2]; v_prev = val[1]; ci = c_index[2]; }
		const float s_1 = (pick > 0) ? (s.s_1 - v_prev) / w_pick : s.s_1 / w_pick;
		const float w_glossy = (rc1 == 0) ? width[0] : (rc1 == 1 ? width[1] : width[2]);
		const float w_diffuse = (rc2 == 0) ? width[0] : (rc2 == 1 ? width[1] : width[2]);
		Col scolor = mkc(0.f, 0.f, 0.f);
		float cos_ng_wi;
		if(ci == 0)
		{
			wi = reflect_dir(n, wo);
			scolor = (col3(m.mirror_color) * kr) * m.mirror_strength;
			s.pdf = w_pick;
		}
		else if(ci == 1) hs = lobe_sample(m, s_1, s.s_2);
		else
		{
			wi = sample_cos_hemisphere(n, sp.nu, sp.nv, s_1, s.s_2);
			cos_ng_wi = dot(sp.ng, wi);
			if(cos_ng_wo * cos_ng_wi < 0.f) return mkc(0.f, 0.f, 0.f);
		}
		float wi_n = fabsf(dot(wi, n));
		const float wo_n = fabsf(dot(wo, n));
		if(ci != 0)
		{
			if(use1)
			{
				float cos_wo_h;
				V3 h;
				if(ci != 1)
				{
					h = normalize(wi + wo);
					hs = local_h(m, sp, h, dot(h, n));
					cos_wo_h = dot(wo, h);
				}
				else
				{
					h = sp.nu * hs.x + sp.nv * hs.y + n * hs.z;
					cos_wo_h = dot(wo, h);
					if(cos_wo_h < 0.f) { h = vec_reflect(h, n); cos_wo_h = dot(wo, h); }
					wi = reflect_dir(h, wo);
					cos_ng_wi = dot(sp.ng, wi);
					if(cos_ng_wo * cos_ng_wi < 0.f) return mkc(0.f, 0.f, 0.f);
				}
				wi_n = fabsf(dot(wi, n));
				const float cos_hn = dot(h, n);
				s.pdf += lobe_pdf(m, hs, cos_hn, cos_wo_h) * w_glossy;
				const float glossy = (float)((double)(lobe_d(m, hs, cos_hn) * schlick_fresnel(cos_wo_h, d.m_glossy)) / as_divisor(cos_wo_h, wo_n, wi_n));
				scolor = col3(m.gloss_color) * (glossy * kt);
			}
			if(use2)
			{
				Col add = diffuse_reflect(wi_n, wo_n, d.m_glossy, d.m_diffuse, col3(m.diff_color)) * kt;
				if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) add = add * m.diffuse_refl;
			if(m.use_oren) add = add * sd_oren(m, wi, wo, n);
				scolor = scolor + add;
				s.pdf += wi_n * w_diffuse;
			}
			w = wi_n / (s.pdf * 0.99f + 0.01f);
		}
		else w = 1.f;
		s.sampled = m.c_flags[ci];
		return scolor;
	}
	if(YG_IS(m, YAFGPU_MAT_MIRROR))
	{	// MirrorMaterial::sample, material_glass.cc:467-473: flags ignored, pdf left at Sample's initial 0
		wi = reflect_dir(sp.n, wo);
		s.sampled = kSpecular | kReflect;
		w = 1.f;
		return col3(m.mirror_color) * (1.f / fabsf(dot(sp.n, wi)));
	}
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE))
	{
		float acc[4];
		const float cos_ng_wo = dot(sp.ng, wo);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		sd_accumulate(d, sd_fresnel(m, wo, n), acc);
		float sum = 0.f, val[4], width[4];
		uint32_t choice[4];
		int n_match = 0, pick = -1;
		for(int i = 0; i < m.n_bsdf; ++i)
		{
			if((s.flags & m.c_flags[i]) == m.c_flags[i])
			{
				width[n_match] = acc[m.c_index[i]];
				sum += width[n_match];
				choice[n_match] = m.c_flags[i];
				val[n_match] = sum;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) { s.sampled = kNone; s.pdf = 0.f; return mkc(1.f, 1.f, 1.f); }
		const float inv_sum = 1.f / sum;
		for(int i = 0; i < n_match; ++i)
		{
			val[i] *= inv_sum;
			width[i] *= inv_sum;
			if((s.s_1 <= val[i]) && (pick < 0)) pick = i;
		}
		if(pick < 0) pick = n_match - 1;
		float s_1;
		if(pick > 0) s_1 = (s.s_1 - val[pick - 1]) / width[pick];
		else s_1 = s.s_1 / width[pick];
		Col scolor = mkc(0.f, 0.f, 0.f);
		const uint32_t ch = choice[pick];
		if(ch == (kSpecular | kReflect))
		{
			wi = reflect_dir(n, wo);
			s.pdf = width[pick];
			scolor = col3(m.mirror_color) * acc[0];
			scolor = scolor * (1.f / smax(fabsf(dot(sp.n, wi)), 1.0e-6f));
		}
		else if(ch == (kTransmit | kFilter))
		{
			wi = -wo;
			const float omf = 1.f - m.transmit_filter;
			scolor = (col3(m.diffuse_color) * m.transmit_filter + mkc(omf, omf, omf)) * acc[1];
			const float cos_n = fabsf(dot(wi, n));
			s.pdf = ((double)cos_n < 1e-6) ? 0.f : width[pick];
		}
		else if(ch == (kDiffuse | kTransmit))
		{
			wi = sample_cos_hemisphere(-n, sp.nu, sp.nv, s_1, s.s_2);
			if(cos_ng_wo * dot(sp.ng, wi) < 0.f) scolor = col3(m.diffuse_color) * acc[2];
			s.pdf = fabsf(dot(wi, n)) * width[pick];
		}
		else
		{
			wi = sample_cos_hemisphere(n, sp.nu, sp.nv, s_1, s.s_2);
			if(cos_ng_wo * dot(sp.ng, wi) > 0.f) scolor = col3(m.diffuse_color) * acc[3];
			if(m.use_oren) scolor = scolor * sd_oren(m, wo, wi, n);
			s.pdf = fabsf(dot(wi, n)) * width[pick];
		}
		s.sampled = ch;
		w = (fabsf(dot(wi, sp.n))) / (s.pdf * 0.99f + 0.01f);
		const float alpha = sd_alpha(m, d, sp, wo);
		w = w * (alpha) + 1.f * (1.f - alpha);
		return scolor;
	}
	if(YG_IS(m, YAFGPU_MAT_GLOSSY))
	{
		const float cos_ng_wo = dot(sp.ng, wo);
		const V3 n = face_forward(sp.ng, sp.n, wo);
		s.pdf = 0.f;
		float wi_n = 0.f;
		const float wo_n = fabsf(dot(wo, n));
		Col scolor = mkc(0.f, 0.f, 0.f);
		float s_1 = s.s_1;
		const float cur_p = d.p_diffuse;
		const bool use_glossy = m.as_diffuse ? (s.flags & kDiffuse) != 0 : (s.flags & kGlossy) != 0;
		const bool use_diffuse = m.with_diffuse && (s.flags & kDiffuse);
		float glossy = 0.f;
		if(use_diffuse)
		{
			const float s_p_diffuse = use_glossy ? cur_p : 1.f;
			if(s_1 < s_p_diffuse)
			{
				s_1 /= s_p_diffuse;
				wi = sample_cos_hemisphere(n, sp.nu, sp.nv, s_1, s.s_2);
				if(dot(sp.ng, wi) * cos_ng_wo < 0.f) return scolor;
				wi_n = fabsf(dot(wi, n));
				s.pdf = wi_n;
				if(use_glossy)
				{
					const V3 h = normalize(wi + wo);
					const float cos_wo_h = dot(wo, h);
					const float cos_wi_h = fabsf(dot(wi, h));
					const float cos_n_h = dot(n, h);
					const V3 hl = local_h(m, sp, h, cos_n_h);
					s.pdf = s.pdf * cur_p + lobe_pdf(m, hl, cos_n_h, cos_wo_h) * (1.f - cur_p);
					glossy = (float)((double)(lobe_d(m, hl, cos_n_h) * schlick_fresnel(cos_wi_h, d.m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
				}
				s.sampled = kDiffuse | kReflect;
				if(!(s.flags & kReflect)) return mkc(0.f, 0.f, 0.f);
				scolor = col3(m.gloss_color) * glossy;
				Col add = diffuse_reflect(wi_n, wo_n, d.m_glossy, d.m_diffuse, col3(m.diff_color));
				if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) add = add * m.diffuse_refl;
			if(m.use_oren) add = add * sd_oren(m, wi, wo, n);
				scolor = scolor + add;
				w = wi_n / (s.pdf * 0.99f + 0.01f);
				return scolor;
			}
			s_1 -= cur_p;
			s_1 /= (1.f - cur_p);
		}
		if(use_glossy)
		{
			const V3 hs = lobe_sample(m, s_1, s.s_2);
			V3 h = sp.nu * hs.x + sp.nv * hs.y + n * hs.z;
			float cos_wo_h = dot(wo, h);
			if(cos_wo_h < 0.f)
			{	// Vec3::reflect, vector.h:265-272
				const float vn = 2.0f * (h.x * n.x + h.y * n.y + h.z * n.z);
				h = mk(vn * n.x - h.x, vn * n.y - h.y, vn * n.z - h.z);
				cos_wo_h = dot(wo, h);
			}
			wi = reflect_dir(h, wo);
			if(cos_ng_wo * dot(sp.ng, wi) < 0.f) return mkc(0.f, 0.f, 0.f);
			wi_n = fabsf(dot(wi, n));
			const float cos_hn = dot(h, n);
			// the anisotropic branch keeps the sampled Hs (:299-300), Blinn takes h * n of the (possibly reflected) h (:325-328)
			s.pdf = lobe_pdf(m, hs, cos_hn, cos_wo_h);
			glossy = (float)((double)(lobe_d(m, hs, cos_hn) * schlick_fresnel(cos_wo_h, d.m_glossy)) / as_divisor(cos_wo_h, wo_n, wi_n));
			scolor = col3(m.gloss_color) * glossy;
			s.sampled = m.as_diffuse ? (kDiffuse | kReflect) : (kGlossy | kReflect);
		}
		if(use_diffuse)
		{
			Col add = diffuse_reflect(wi_n, wo_n, d.m_glossy, d.m_diffuse, col3(m.diff_color));
			if(YAFGPU_FEAT_TEXTURE && m.has_diffuse_refl) add = add * m.diffuse_refl;
			if(m.use_oren) add = add * sd_oren(m, wi, wo, n);
			s.pdf = wi_n * cur_p + s.pdf * (1.f - cur_p);
			scolor = scolor + add;
		}
		w = wi_n / (s.pdf * 0.99f + 0.01f);
		return scolor;
	}
	s.pdf = 0.f; w = 0.f;
	return mkc(0.f, 0.f, 0.f);
}

// Material::emit — material_shiny_diffuse.cc:295-306, material_simple.cc:50-57
YG_DEV Col mat_emit(const yafgpu_material &m, const SurfPt &sp, V3 wo, bool include_lights)
{
	if(YG_IS(m, YAFGPU_MAT_SHINYDIFFUSE)) return col3(m.emit_color);
	if(YG_IS(m, YAFGPU_MAT_LIGHT))
	{
		if(!include_lights) return mkc(0.f, 0.f, 0.f);
		if(m.double_sided) return col3(m.light_col);
		return (dot(wo, sp.n) > 0.f) ? col3(m.light_col) : mkc(0.f, 0.f, 0.f);
	}
	return mkc(0.f, 0.f, 0.f);
}

// AreaLight::illumSample, light_area.cc:67-97
YG_DEV bool arealight_illum_sample(const yafgpu_light &l, V3 sp_p, float s_1, float s_2, V3 &wi_dir, float &wi_tmax, float &pdf)
{
	const V3 p = vec3(l.corner) + vec3(l.to_x) * s_1 + vec3(l.to_y) * s_2;
	V3 ldir = p - sp_p;
	const float dist_sqr = ldir.x * ldir.x + ldir.y * ldir.y + ldir.z * ldir.z;
	const float dist = f_sqrt(dist_sqr);
	if(dist <= 0.f) return false;
	const float inv = 1.f / dist;
	ldir.x *= inv; ldir.y *= inv; ldir.z *= inv;
	const float cos_angle = dot(ldir, vec3(l.fnormal));
	if(cos_angle <= 0.f) return false;
	wi_tmax = dist;
	wi_dir = ldir;
	pdf = (float)((double)dist_sqr * kPi / (double)(l.area * cos_angle));
	return true;
}
// triIntersect__, light_area.cc:118-137
YG_DEV bool tri_intersect_plain(V3 a, V3 b, V3 c, V3 from, V3 dir, float &t)
{
	const V3 edge_1 = b - a, edge_2 = c - a;
	const V3 pvec = cross(dir, edge_2);
	const float det = dot(edge_1, pvec);
	if(det == 0.f) return false;
	const float inv_det = 1.0f / det;
	const V3 tvec = from - a;
	const float u = dot(tvec, pvec) * inv_det;
	if(u < 0.f || u > 1.f) return false;
	const V3 qvec = cross(tvec, edge_1);
	const float v = dot(dir, qvec) * inv_det;
	if((v < 0.f) || ((u + v) > 1.f)) return false;
	t = dot(edge_2, qvec) * inv_det;
	return true;
}
// AreaLight::intersect, light_area.cc:139-155 (returns the INVERSE pdf)
YG_DEV bool arealight_intersect(const yafgpu_light &l, V3 from, V3 dir, float &t, float &ipdf)
{
	const float cos_angle = dot(dir, vec3(l.fnormal));
	if(cos_angle <= 0.f) return false;
	if(!tri_intersect_plain(vec3(l.corner), vec3(l.c2), vec3(l.c3), from, dir, t))
	{
		if(!tri_intersect_plain(vec3(l.corner), vec3(l.c3), vec3(l.c4), from, dir, t)) return false;
	}
	if(!(t > 1.0e-10f)) return false;
	ipdf = (float)((double)(1.f / (t * t) * l.area * cos_angle) * k1Pi);
	return true;
}
// PointLight::illuminate, light_point.cc:38-56
YG_DEV bool pointlight_illuminate(const yafgpu_light &l, V3 sp_p, Col &col, V3 &wi_dir, float &wi_tmax)
{
	V3 ldir = vec3(l.position) - sp_p;
	const float dist_sqr = ldir.x * ldir.x + ldir.y * ldir.y + ldir.z * ldir.z;
	const float dist = f_sqrt(dist_sqr);
	if(dist == 0.f) return false;
	const float idist_sqr = 1.f / (dist_sqr);
	const float inv = 1.f / dist;
	ldir.x *= inv; ldir.y *= inv; ldir.z *= inv;
	wi_tmax = dist;
	wi_dir = ldir;
	col = col3(l.color) * idist_sqr;
	return true;
}

// PerspectiveCamera::biasDist, camera_perspective.cc:75-89
YG_DEV float camera_bias_dist(const yafgpu_camera &c, float r)
{
	if(c.bokeh_bias == 1) return f_sqrt(f_sqrt(r) * r);
	if(c.bokeh_bias == 2) return f_sqrt(1.0f - r * r);
	return f_sqrt(r);
}
// shirleyDisk__, vector.cc:155-190 (the angle is formed in double and narrowed)
YG_DEV void shirley_disk(float r_1, float r_2, float &u, float &v)
{
	constexpr double kPi4 = 0.78539816339744830962;
	float phi = 0.f, r = 0.f;
	const float a = 2.f * r_1 - 1.f, b = 2.f * r_2 - 1.f;
	if(a > -b)
	{
		if(a > b) { r = a; phi = (float)(kPi4 * (double)(b / a)); }
		else { r = b; phi = (float)(kPi4 * (double)(2.f - a / b)); }
	}
	else
	{
		if(a < b) { r = -a; phi = (float)(kPi4 * (double)(4.f + b / a)); }
		else
		{
			r = -b;
			phi = (b != 0.f) ? (float)(kPi4 * (double)(6.f - a / b)) : 0.f;
		}
	}
	u = r * f_cos(phi);
	v = r * f_sin(phi);
}
// PerspectiveCamera::getLensUv / sampleTsd, camera_perspective.cc:91-131
YG_DEV void camera_lens_uv(const yafgpu_camera &c, float r_1, float r_2, float &u, float &v)
{
	const int bt = c.bokeh_type;
	if(bt >= 3 && bt <= 6)
	{
		const float fn = (float)bt;
		int idx = (int)(r_1 * fn);
		r_1 = (r_1 - ((float)idx) / fn) * fn;
		r_1 = camera_bias_dist(c, r_1);
		const float b_1 = r_1 * r_2;
		const float b_0 = r_1 - b_1;
		idx <<= 1;
		u = c.ls[idx] * b_0 + c.ls[idx + 2] * b_1;
		v = c.ls[idx + 1] * b_0 + c.ls[idx + 3] * b_1;
	}
	else if(bt == 1 || bt == 7)
	{
		const float w = (float)6.28318530717958647692 * r_2;
		if(bt == 7) r_1 = f_sqrt((float)0.707106781 + (float)0.292893218);
		else r_1 = camera_bias_dist(c, r_1);
		u = r_1 * f_cos(w);
		v = r_1 * f_sin(w);
	}
	else shirley_disk(r_1, r_2, u, v);
}

// PerspectiveCamera::shootRay, camera_perspective.cc:133-156; rayPlaneIntersection__ util_geometry.h:34-37
YG_DEV void camera_shoot(const yafgpu_camera &c, float px, float py, float lu, float lv, V3 &from, V3 &dir, float &tmin, float &tmax)
{
	from = vec3(c.position);
	dir = normalize(vec3(c.vright) * px + vec3(c.vup) * py + vec3(c.vto));
	tmin = dot(vec3(c.near_n), vec3(c.near_p) - from) / dot(dir, vec3(c.near_n));
	tmax = dot(vec3(c.far_n), vec3(c.far_p) - from) / dot(dir, vec3(c.far_n));
	if(c.aperture != 0.f)
	{
		float u, v;
		camera_lens_uv(c, lu, lv, u, v);
		const V3 li = vec3(c.dof_rt) * u + vec3(c.dof_up) * v;
		from = from + li;
		dir = normalize(dir * c.dof_distance - li);
	}
}
YG_DEV void camera_shoot(const yafgpu_camera &c, float px, float py, V3 &from, V3 &dir, float &tmin, float &tmax)
{
	camera_shoot(c, px, py, 0.5f, 0.5f, from, dir, tmin, tmax);
}

} // namespace yafgpu
