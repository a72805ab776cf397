/* TEST INFRASTRUCTURE — CPU oracle (plain C restatement of the reference's path-tracing hot
 * path).  Never linked into or called from the product; see yaf_oracle.h for who may use it
 * and for the pinning status of each part.
 *
 * All citations are file:line in /root/reference.  Arithmetic types follow the reference
 * expression by expression: where the C++ source mixes float with double literals the
 * double intermediate is kept, because the result differs in the last bits otherwise.
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "yaf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <pthread.h>

/* ------------------------------------------------------------------ constants */
#define Y_M_PI     3.14159265358979323846
#define Y_M_PI_2   1.57079632679489661923
#define Y_M_PI_4   0.78539816339744830962
#define Y_M_1_PI   0.31830988618379067154
#define Y_M_2PI    6.28318530717958647692  /* util_math_optimizations.h:84 */
#define Y_M_1_2PI  0.15915494309189533577  /* :86 */
#define Y_M_4_PI   1.27323954473516268615  /* :87 */
#define Y_M_4_PI2  0.40528473456935108578  /* :88 */
#define MIN_RAYDIST     0.00005            /* CMakeLists.txt:46-48 */
#define YAF_SHADOW_BIAS 0.0005             /* CMakeLists.txt:50-52 */
#define KD_MAX_STACK 64                    /* kdtree_triangle.cc:43 */

/* BsdfFlags, material.h:49-64 */
enum {
	BSDF_NONE = 0, BSDF_SPECULAR = 1, BSDF_GLOSSY = 2, BSDF_DIFFUSE = 4, BSDF_DISPERSIVE = 8,
	BSDF_REFLECT = 0x10, BSDF_TRANSMIT = 0x20, BSDF_FILTER = 0x40, BSDF_EMIT = 0x80, BSDF_VOLUMETRIC = 0x100,
	BSDF_ALL = BSDF_SPECULAR | BSDF_GLOSSY | BSDF_DIFFUSE | BSDF_DISPERSIVE | BSDF_REFLECT | BSDF_TRANSMIT | BSDF_FILTER
};
enum { VIS_NORMAL = 0, VIS_NO_SHADOWS = 1, VIS_SHADOW_ONLY = 2, VIS_INVISIBLE = 3 };

typedef struct { float x, y, z; } v3;
typedef struct { float r, g, b; } rgb;

static inline v3 V(float x, float y, float z) { v3 v = {x, y, z}; return v; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* vector.h:159-167: both Vec3*float overloads compute f*b.x_ */
static inline v3 vmul(v3 b, float f) { return V(f * b.x, f * b.y, f * b.z); }
/* vector.h:154 */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }
/* vector.h:194 */
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vcomp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline rgb C(float r, float g, float b) { rgb c = {r, g, b}; return c; }
static inline rgb cadd(rgb a, rgb b) { return C(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline rgb cmul(rgb a, rgb b) { return C(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline rgb cscale(rgb b, float f) { return C(f * b.r, f * b.g, f * b.b); }   /* color.h:261-269 */
static inline rgb cdiv(rgb b, float f) { return C(b.r / f, b.g / f, b.b / f); }       /* color.h:271 */
static inline int cblack(rgb c) { return c.r == 0 && c.g == 0 && c.b == 0; }
static inline float fmaxf_(float a, float b) { return a < b ? b : a; } /* std::max(a,b) */
static inline float fminf_(float a, float b) { return b < a ? b : a; } /* std::min(a,b) */

/* ------------------------------------------------------------------ fast math
 * util_math_optimizations.h:90-253 (FAST_MATH and FAST_TRIG are ON by default, CMakeLists.txt:22-23) */
typedef union { int32_t i; float f; } bit_twiddler;

float yor_fexp2(float x) /* :116-129 */
{
	bit_twiddler ipart, fpart, expipart;
	x = fminf_(x, 129.00000f);
	x = fmaxf_(x, -126.99999f);
	ipart.i = (int32_t)(x - 0.5f);
	fpart.f = (x - (float)(ipart.i));
	expipart.i = (int32_t)((uint32_t)(ipart.i + 127) << 23);
	{
		float p = fpart.f; /* POLYEXP :93, all-float */
		float poly = (float)(p * (p * (p * (p * (p * 1.8775767e-3f + 8.9893397e-3f) + 5.5826318e-2f) + 2.4015361e-1f) + 6.9315308e-1f) + 9.9999994e-1f);
		return (expipart.f * poly);
	}
}

float yor_flog2(float x) /* :131-142 */
{
	bit_twiddler one, i, m, e;
	one.f = 1.0f;
	i.f = x;
	e.f = (float)(((i.i & 0x7F800000) >> 23) - 127);
	m.i = ((i.i & 0x7FFFFF) | one.i);
	{
		/* POLYLOG :94 — the constant 2.5988452 has no f suffix, so from there on the Horner
		 * chain runs in double and is narrowed once at the end */
		float xx = m.f;
		float a = xx * -3.4436006e-2f + 3.1821337e-1f;
		float b = xx * a + -1.2315303f;
		double c = (double)(xx * b) + 2.5988452;
		double d = (double)xx * c + (double)-3.3241990f;
		double ee = (double)xx * d + (double)3.1157899f;
		float poly = (float)ee;
		return (poly * (m.f - one.f) + e.f);
	}
}

float yor_fpow(float a, float b) { return yor_fexp2(yor_flog2(a) * b); } /* :176-183 */
float yor_fsqrt(float a) { return sqrtf(a); }                              /* :203-210, :168 */

float yor_fsin(float x) /* :222-244 */
{
	if((double)x > Y_M_2PI || (double)x < -Y_M_2PI) x -= ((int)(x * (float)Y_M_1_2PI)) * (float)Y_M_2PI;
	if((double)x < -Y_M_PI) x += (float)Y_M_2PI;
	else if((double)x > Y_M_PI) x -= (float)Y_M_2PI;
	x = ((float)Y_M_4_PI * x) - ((float)Y_M_4_PI2 * x * fabsf(x));
	{
		float result = 0.225f * (x * fabsf(x) - x) + x;
		if((double)result <= -1.0) return -1.0f;
		else if((double)result >= 1.0) return 1.0f;
		else return result;
	}
}
float yor_fcos(float x) { return yor_fsin(x + (float)Y_M_PI_2); } /* :246-253 */
float yor_facos(float x) /* :255-261 */
{
	if((double)x <= -1.0) return (float)Y_M_PI;
	else if((double)x >= 1.0) return 0.0f;
	else return acosf(x);
}

/* vector.h:227-238 */
static inline v3 vnormalize(v3 v)
{
	float len = v.x * v.x + v.y * v.y + v.z * v.z;
	if(len != 0)
	{
		len = (float)(1.0 / (double)yor_fsqrt(len));
		v.x *= len; v.y *= len; v.z *= len;
	}
	return v;
}
static inline float vlength(v3 v) { return yor_fsqrt(v.x * v.x + v.y * v.y + v.z * v.z); } /* vector.h:222 */

/* vector.h:319-337 */
static void create_cs(v3 n, v3 *u, v3 *v)
{
	if((n.x == 0) && (n.y == 0))
	{
		if(n.z < 0) *u = V(-1, 0, 0);
		else *u = V(1, 0, 0);
		*v = V(0, 1, 0);
	}
	else
	{
		const float d = (float)(1.0 / (double)yor_fsqrt(n.y * n.y + n.x * n.x));
		*u = V(n.y * d, -n.x * d, 0);
		*v = vcross(n, *u);
	}
}

/* vector.h:273-278 */
static inline v3 reflect_dir(v3 n, v3 v)
{
	const float vn = vdot(v, n);
	if(vn < 0) return vneg(v);
	/* 2 * vn * n - v : (2*vn) float, Vec3 scale, minus */
	return vsub(vmul(n, 2 * vn), v);
}

/* ------------------------------------------------------------------ QMC
 * util_mcqmc.h, scr_halton.h */
#define MULT_RATIO 0.00000000023283064365386962890625 /* util_mcqmc.h:91 */

static inline float clamp01f(float v) { return fmaxf_(0.f, fminf_(1.f, v)); }

float yor_ri_vdc(uint32_t bits, uint32_t r) /* util_mcqmc.h:93-101 */
{
	bits = (bits << 16) | (bits >> 16);
	bits = ((bits & 0x00ff00ff) << 8) | ((bits & 0xff00ff00) >> 8);
	bits = ((bits & 0x0f0f0f0f) << 4) | ((bits & 0xf0f0f0f0) >> 4);
	bits = ((bits & 0x33333333) << 2) | ((bits & 0xcccccccc) >> 2);
	bits = ((bits & 0x55555555) << 1) | ((bits & 0xaaaaaaaa) >> 1);
	return clamp01f((float)((double)(bits ^ r) * MULT_RATIO));
}
float yor_ri_s(uint32_t i, uint32_t r) /* :103-108 */
{
	for(uint32_t v = 1u << 31; i; i >>= 1, v ^= v >> 1)
		if(i & 1) r ^= v;
	return clamp01f((float)((double)r * MULT_RATIO));
}
float yor_ri_lp(uint32_t i, uint32_t r) /* :110-115 */
{
	for(uint32_t v = 1u << 31; i; i >>= 1, v |= v >> 1)
		if(i & 1) r ^= v;
	return clamp01f((float)((double)r * MULT_RATIO));
}
uint32_t yor_fnv32a(uint32_t value) /* :147-163 (little-endian byte order of the union) */
{
	uint32_t hash = 0x811c9dc5u;
	for(int i = 0; i < 4; i++)
	{
		hash ^= (value >> (8 * i)) & 0xffu;
		hash *= 0x01000193u;
	}
	return hash;
}

/* Halton, util_mcqmc.h:28-87 */
typedef struct { uint32_t base; double inv_base, value; } halton_t;
static void halton_init(halton_t *h, int base) { h->base = (uint32_t)base; h->inv_base = 1.0 / (double)base; h->value = 0; }
static void halton_set_start(halton_t *h, uint32_t i)
{
	double factor = h->inv_base;
	h->value = 0.0;
	while(i > 0)
	{
		h->value += (double)(i % h->base) * factor;
		i /= h->base;
		factor *= h->inv_base;
	}
}
static float halton_next(halton_t *h)
{
	double r = 0.9999999999 - h->value;
	if(h->inv_base < r) h->value += h->inv_base;
	else
	{
		double hh = 0.0, hv = h->inv_base;
		while(hv >= r) { hh = hv; hv *= h->inv_base; }
		h->value += hh + hv - 1.0;
	}
	return fmaxf_(0.f, fminf_(1.f, (float)h->value));
}
void yor_halton_seq(uint32_t base, uint32_t start, int count, float *out)
{
	halton_t h; halton_init(&h, (int)base); halton_set_start(&h, start);
	for(int i = 0; i < count; ++i) out[i] = halton_next(&h);
}

/* MWC PRNG, util_mcqmc.h:173-192 */
typedef struct { uint32_t x, c; } mwc_t;
static void mwc_init(mwc_t *p, uint32_t seed) { p->x = 30903; p->c = seed; }
static double mwc_next(mwc_t *p)
{
	const uint32_t ya = 1791398085u, yah = ya >> 16, yal = ya & 65535u;
	const uint32_t xh = p->x >> 16, xl = p->x & 65535u;
	p->x = p->x * ya + p->c;
	p->c = xh * yah + ((xh * yal) >> 16) + ((xl * yah) >> 16);
	if(xl * yal >= ~p->c + 1) p->c++;
	return (double)p->x * MULT_RATIO;
}
/* glibc's rand() / srand() (stdlib/random_r.c, the default TYPE_3 state of 31 words): third-party arithmetic the
 * reference's tile seeds are drawn from (integrator_tiled.cc:319).  Restated from its published algorithm and pinned
 * against this machine's libc in tests/test_oracle_golden.py. */
typedef struct { int32_t r[34]; uint32_t *buf; int n, pos; } grand_t;
void yor_glibc_rand(uint32_t seed, int count, int32_t *out)
{
	if(count <= 0) return;
	size_t total = (size_t)count + 344;
	uint32_t *r = (uint32_t *)malloc(total * sizeof(uint32_t));
	if(seed == 0) seed = 1;
	r[0] = seed;
	for(int i = 1; i < 31; ++i)
	{	/* 16807 * r[i-1] % 2147483647 by Schrage's method, as random_r.c does it */
		int32_t word = (int32_t)r[i - 1];
		long hi = word / 127773, lo = word % 127773;
		word = (int32_t)(16807 * lo - 2836 * hi);
		if(word < 0) word += 2147483647;
		r[i] = (uint32_t)word;
	}
	for(int i = 31; i < 34; ++i) r[i] = r[i - 31];
	for(size_t i = 34; i < total; ++i) r[i] = r[i - 31] + r[i - 3];
	for(int k = 0; k < count; ++k) out[k] = (int32_t)(r[(size_t)k + 344] >> 1);
	free(r);
}

void yor_mwc_seq(uint32_t seed, int count, float *out)
{
	mwc_t p; mwc_init(&p, seed);
	for(int i = 0; i < count; ++i) out[i] = (float)mwc_next(&p);
}

/* scr_halton.h:29-47.  The Faure permutations (faure_tables.cc) are not copied: they are
 * regenerated with Faure's construction and a test compares them with the reference file.
 * Dimensions 0,1,2 all use the length-3 identity (faure_tables.cc:437). */
static const int prims[50] = {1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67,
                              71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167,
                              173, 179, 181, 191, 193, 197, 199, 211, 223, 227};
/* the reference's 9-digit truncated inverses, scr_halton.h:37-47 (these are data: 1/p printed to 9 decimals) */
static double inv_prims[50];
static int *faure_tab[50];
static int faure_ready = 0;
static pthread_once_t faure_once = PTHREAD_ONCE_INIT;

static void faure_build(int b, int *out) /* Faure 1992: sigma_b from sigma_{b/2} (even) or sigma_{b-1} (odd) */
{
	if(b == 2) { out[0] = 0; out[1] = 1; return; }
	if(b % 2 == 0)
	{
		int h = b / 2;
		int *half = (int *)malloc(sizeof(int) * (size_t)h);
		faure_build(h, half);
		for(int i = 0; i < h; ++i) { out[i] = 2 * half[i]; out[h + i] = 2 * half[i] + 1; }
		free(half);
	}
	else
	{
		int m = (b - 1) / 2;
		int *prev = (int *)malloc(sizeof(int) * (size_t)(b - 1));
		faure_build(b - 1, prev);
		for(int i = 0; i < b - 1; ++i) if(prev[i] >= m) prev[i]++;
		for(int i = 0; i < m; ++i) out[i] = prev[i];
		out[m] = m;
		for(int i = m; i < b - 1; ++i) out[i + 1] = prev[i];
		free(prev);
	}
}
static void faure_init(void)
{
	for(int d = 0; d < 50; ++d)
	{
		/* 1/p rounded to 9 decimals, as printed in scr_halton.h:37-47 */
		double q = floor(1e9 / (double)prims[d] + 0.5);
		char buf[32];
		snprintf(buf, sizeof buf, "%.9f", q / 1e9);
		inv_prims[d] = strtod(buf, NULL);
		int b = d < 3 ? 3 : prims[d];
		faure_tab[d] = (int *)malloc(sizeof(int) * (size_t)b);
		if(d < 3) { faure_tab[d][0] = 0; faure_tab[d][1] = 1; faure_tab[d][2] = 2; }
		else faure_build(b, faure_tab[d]);
	}
	faure_ready = 1;
}
const int *yor_faure_perm(int dim, int *len)
{
	pthread_once(&faure_once, faure_init);
	*len = dim < 3 ? 3 : prims[dim];
	return faure_tab[dim];
}

double yor_scr_halton(int dim, uint32_t n) /* scr_halton.h:52-75 */
{
	double value = 0.0;
	pthread_once(&faure_once, faure_init);
	if(dim < 50)
	{
		const int *sigma = faure_tab[dim];
		uint32_t base = (uint32_t)prims[dim];
		double f, factor, dn = (double)n;
		f = factor = inv_prims[dim];
		while(n > 0)
		{
			value += (double)(sigma[n % base]) * factor;
			dn *= f;
			n = (uint32_t)dn;
			factor *= f;
		}
	}
	else
	{
		/* dim >= 50 falls back to a global racy LCG in the reference (scr_halton.h:70-73); the
		 * path tracer reaches it only for bounces > 12, which yor_render rejects */
		value = 0.5;
	}
	return fmax(1.0e-36, fmin(1.0, value));
}

static inline float add_mod1(float a, float b) { float s = a + b; return s > 1 ? s - 1.f : s; } /* util_sample.h:183-187 */

/* util_sample.h:45-55 */
static v3 sample_cos_hemisphere(v3 n, v3 ru, v3 rv, float s_1, float s_2)
{
	if(s_1 >= 1.0f) return n;
	else
	{
		float z_1 = s_1;
		float z_2 = (float)((double)s_2 * Y_M_2PI);
		v3 a = vadd(vmul(ru, yor_fcos(z_2)), vmul(rv, yor_fsin(z_2)));
		return vadd(vmul(a, yor_fsqrt((float)(1.0 - (double)z_1))), vmul(n, yor_fsqrt(z_1)));
	}
}

/* ------------------------------------------------------------------ bound.h:144-212 */
static int bound_cross(v3 a_0, v3 a_1, v3 from, v3 dir, float *enter, float *leave, float dist)
{
	v3 p = vsub(from, a_0);
	float lmin = (float)-1e38, lmax = (float)1e38, ltmin, ltmax;
	if(dir.x != 0)
	{
		float invrx = (float)(1. / (double)dir.x);
		if(invrx > 0) { lmin = -p.x * invrx; lmax = ((a_1.x - a_0.x) - p.x) * invrx; }
		else { lmin = ((a_1.x - a_0.x) - p.x) * invrx; lmax = -p.x * invrx; }
		if((lmax < 0) || (lmin > dist)) return 0;
	}
	if(dir.y != 0)
	{
		float invry = (float)(1. / (double)dir.y);
		if(invry > 0) { ltmin = -p.y * invry; ltmax = ((a_1.y - a_0.y) - p.y) * invry; }
		else { ltmin = ((a_1.y - a_0.y) - p.y) * invry; ltmax = -p.y * invry; }
		lmin = fmaxf_(ltmin, lmin);
		lmax = fminf_(ltmax, lmax);
		if((lmax < 0) || (lmin > dist)) return 0;
	}
	if(dir.z != 0)
	{
		float invrz = (float)(1. / (double)dir.z);
		if(invrz > 0) { ltmin = -p.z * invrz; ltmax = ((a_1.z - a_0.z) - p.z) * invrz; }
		else { ltmin = ((a_1.z - a_0.z) - p.z) * invrz; ltmax = -p.z * invrz; }
		lmin = fmaxf_(ltmin, lmin);
		lmax = fminf_(ltmax, lmax);
		if((lmax < 0) || (lmin > dist)) return 0;
	}
	if((lmin <= lmax) && (lmax >= 0) && (lmin <= dist))
	{
		*enter = lmin;
		*leave = lmax;
		return 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ scene data */
typedef struct
{
	v3 a, e1, e2;      /* vertex a and cached edges, triangle.h:197-205 */
	float eps;         /* intersection_bias_factor_, triangle.h:206 */
	v3 ng;             /* recNormal, triangle.h:295-302 */
	int mat;
	int smooth;        /* has per-vertex normals */
	v3 na, nb, nc;
	v3 b, c;
} tri_t;

typedef struct
{
	int type, visibility, receive_shadows, flat;
	unsigned flags;
	/* shinydiffuse state after config(), material_shiny_diffuse.cc:46-92 */
	rgb diffuse_color, mirror_color, emit_color;
	float mirror_strength, transparency_strength, translucency_strength, diffuse_strength, emit_strength, transmit_filter;
	int has_vol_i; rgb beer_sigma;   /* vol_i_: BeerVolumeHandler of an absorbing glass */
	int is_mirror, is_transparent, is_translucent, is_diffuse, has_fresnel;
	float ior_squared;
	int use_oren; float oren_a, oren_b;
	int n_bsdf; unsigned c_flags[4]; int c_index[4];
	/* glossy */
	rgb gloss_color, diff_color;
	float exponent, reflectivity, diffuse;
	int as_diffuse, with_diffuse;
	int anisotropic; float exp_u, exp_v;
	/* light mat */
	rgb light_col; int double_sided;
	/* glass (material_glass.cc:32-49) and mirror (material_glass.h:74-79) */
	float ior; rgb filter_color, spec_refl_color; int fake_shadow; unsigned tm_flags;
	float rg_a2;                          /* rough glass: a_2_ = alpha^2 of the GGX lobe (material_rough_glass.cc:33-36) */
	rgb ref_col;
	/* shader nodes: the list in evaluation order and the node each slot reads (-1: none) */
	int n_nodes; struct node_s *nodes;
	int n_bump, sh_bump; struct node_s *bump_nodes;      /* bump_nodes_ in evaluation order and the bump shader's place in it (NodeMaterial::evalBump) */
	int sh_diffuse, sh_mirror_color, sh_mirror, sh_transparency, sh_translucency, sh_sigma_oren, sh_diffuse_refl, sh_ior;
	int sh_glossy, sh_glossy_reflect, sh_exponent;      /* glossy / coated_glossy: glossy_shader, glossy_reflect_shader, exponent_shader */
	int additional_depth; float transp_bias_factor; int transp_bias_mult;      /* Material::additional_depth_, transparent_bias_* (material.h) */
	int sh_filter_color; float transp_ior;             /* glass: filter_color_shader; the index getTransparency's fresnel sees (:223: the IOR shader's value alone) */
	float ior_plain;                                   /* coated_glossy: ior_ before the IOR shader's offset */
	float ior_base;                       /* ior_, for the IOR shader (material_shiny_diffuse.cc:258-262) */
	/* values a resolved copy carries (mat_resolve): orenNayar with a texture sigma computes A and B in double (:230-235) */
	int oren_tex; double oren_ad, oren_bd;
	int has_diffuse_refl; float diffuse_refl;
} mat_t;

typedef struct
{
	int type, samples, cast_shadows;
	/* area, light_area.cc:34-52 */
	v3 corner, c2, c3, c4, to_x, to_y, normal, fnormal;
	rgb color;
	float area, inv_area;
	/* point, light_point.cc:28-36 */
	v3 position;
} light_t;

typedef struct
{
	v3 position, cam_x, cam_y, cam_z, vto, vup, vright;
	v3 near_p, near_n, far_p, far_n;
	int resx, resy;
	float aspect_ratio, focal, aperture;
	/* depth of field */
	float dof_distance; int bkhtype, bkhbias; v3 dof_rt, dof_up; float ls[16];
} camera_t;

typedef struct { union { float split; uint32_t first; } u; uint32_t flags; } kdnode_t; /* kdtree_triangle.h:48-86, pointers -> indices */

struct yor_scene
{
	int n_tris; tri_t *tris;
	int n_mats; mat_t *mats;
	int n_lights; light_t *lights;
	camera_t cam;
	/* kd tree */
	kdnode_t *nodes; uint32_t n_nodes, cap_nodes;
	uint32_t *leaf_refs; uint32_t n_refs, cap_refs;
	v3 tb_a, tb_g; /* tree bound */
	double build_seconds;
	/* image textures + per-triangle texture coordinates (row N2) */
	int n_tex; struct tex_s *tex;
	float *tri_uv, *tri_orco;
};

/* per-thread counters */
typedef struct { uint64_t rays_closest, rays_shadow, interior, leaves, tests; } counters_t;

/* ------------------------------------------------------------------ triangle.h:223-259 */
static inline int tri_intersect(const tri_t *tr, v3 from, v3 dir, float *t, float *u_out, float *v_out)
{
	v3 pvec = vcross(dir, tr->e2);
	float det = vdot(tr->e1, pvec);
	float epsilon = tr->eps;
	if(det > -epsilon && det < epsilon) return 0;
	{
		float inv_det = 1.f / det;
		v3 tvec = vsub(from, tr->a);
		float u = vdot(tvec, pvec) * inv_det;
		if(u < 0.f || u > 1.f) return 0;
		{
			v3 qvec = vcross(tvec, tr->e1);
			float v = vdot(dir, qvec) * inv_det;
			if((v < 0.f) || ((u + v) > 1.f)) return 0;
			*t = vdot(tr->e2, qvec) * inv_det;
			if(*t < epsilon) return 0;
			*u_out = u; *v_out = v;
			return 1;
		}
	}
}

static inline int mat_visible_primary(const mat_t *m) { return m->visibility == VIS_NORMAL || m->visibility == VIS_NO_SHADOWS; }  /* kdtree_triangle.cc:786 */
static inline int mat_visible_shadow(const mat_t *m) { return m->visibility == VIS_NORMAL || m->visibility == VIS_SHADOW_ONLY; }   /* :938 */

/* ------------------------------------------------------------------ kd-tree build (our own; K4 is not mirrored)
 * Only the *results* of intersect/intersectS are contractual.  Parameters follow
 * scene.cc:818 and kdtree_triangle.cc:89-100: max depth 7+1.66 ln N capped at 64, leaf size 1,
 * cost ratio 0.8 (+ penalty above 65536 prims), empty bonus 0.33.  Split search is a 32-bin SAH
 * over triangle bounds; a triangle is referenced by every leaf its bounding box overlaps. */
typedef struct { v3 lo, hi; } aabb_t;
typedef struct
{
	yor_scene *s;
	const aabb_t *tb;
	int max_depth;
	float cost_ratio, e_bonus;
} build_ctx;

static uint32_t kd_alloc_node(yor_scene *s)
{
	if(s->n_nodes == s->cap_nodes)
	{
		s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 1024;
		s->nodes = (kdnode_t *)realloc(s->nodes, sizeof(kdnode_t) * s->cap_nodes);
	}
	return s->n_nodes++;
}
static void kd_make_leaf(yor_scene *s, uint32_t node, const uint32_t *prims_idx, uint32_t np)
{
	if(s->n_refs + np > s->cap_refs)
	{
		while(s->n_refs + np > s->cap_refs) s->cap_refs = s->cap_refs ? s->cap_refs * 2 : 4096;
		s->leaf_refs = (uint32_t *)realloc(s->leaf_refs, sizeof(uint32_t) * s->cap_refs);
	}
	s->nodes[node].u.first = s->n_refs;
	s->nodes[node].flags = (np << 2) | 3u;
	memcpy(s->leaf_refs + s->n_refs, prims_idx, sizeof(uint32_t) * np);
	s->n_refs += np;
}

#define KD_BINS 32
static void kd_build(build_ctx *c, uint32_t node, aabb_t box, uint32_t *prims_idx, uint32_t np, int depth, int bad_refines)
{
	yor_scene *s = c->s;
	if(np <= 1 || depth >= c->max_depth) { kd_make_leaf(s, node, prims_idx, np); return; }
	float d[3] = {box.hi.x - box.lo.x, box.hi.y - box.lo.y, box.hi.z - box.lo.z};
	float lo[3] = {box.lo.x, box.lo.y, box.lo.z};
	float total_sa = d[0] * d[1] + d[0] * d[2] + d[1] * d[2];
	float inv_total_sa = total_sa > 0 ? 1.f / total_sa : 0.f;
	float best_cost = INFINITY, old_cost = (float)np;
	int best_axis = -1; float best_pos = 0;
	uint32_t best_nl = 0, best_nr = 0;
	for(int axis = 0; axis < 3; ++axis)
	{
		if(!(d[axis] > 0)) continue;
		uint32_t cnt_lo[KD_BINS + 1], cnt_hi[KD_BINS + 1];
		memset(cnt_lo, 0, sizeof cnt_lo); memset(cnt_hi, 0, sizeof cnt_hi);
		float scale = (float)KD_BINS / d[axis];
		for(uint32_t i = 0; i < np; ++i)
		{
			const aabb_t *b = &c->tb[prims_idx[i]];
			float bl = vcomp(b->lo, axis), bh = vcomp(b->hi, axis);
			int il = (int)floorf((bl - lo[axis]) * scale); if(il < 0) il = 0; if(il > KD_BINS) il = KD_BINS;
			int ih = (int)ceilf((bh - lo[axis]) * scale); if(ih < 0) ih = 0; if(ih > KD_BINS) ih = KD_BINS;
			/* il: first plane index k (plane k at lo + k/scale) with plane > bl is il+1 -> counted left of planes > il
			 * ih: prim ends at or before plane ih */
			cnt_lo[il]++; cnt_hi[ih]++;
		}
		/* plane k (1..KD_BINS-1): n_left = #prims with il < k ; n_right = #prims with ih > k */
		uint32_t nl = 0, nr = np;
		int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
		for(int k = 1; k < KD_BINS; ++k)
		{
			nl += cnt_lo[k - 1];
			nr -= cnt_hi[k];
			/* prims ending exactly at plane k were removed from the right; those with ih==k and il<k stay left */
			float pos = lo[axis] + (float)k / scale;
			float l1 = pos - lo[axis], l2 = d[axis] - l1;
			float below_sa = d[a1] * d[a2] + l1 * (d[a1] + d[a2]);
			float above_sa = d[a1] * d[a2] + l2 * (d[a1] + d[a2]);
			float eb = (nl == 0 || nr == 0) ? c->e_bonus : 0.f;
			float cost = c->cost_ratio + inv_total_sa * (below_sa * (float)nl + above_sa * (float)nr) * (1.f - eb);
			if(cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = pos; best_nl = nl; best_nr = nr; }
		}
	}
	if(best_axis < 0) { kd_make_leaf(s, node, prims_idx, np); return; }
	if(best_cost > old_cost) ++bad_refines;
	if((best_cost > 1.6f * old_cost && np < 16) || bad_refines >= 2) { kd_make_leaf(s, node, prims_idx, np); return; }
	(void)best_nl; (void)best_nr;
	/* partition: conservative on the bounds (<= / >= so that a prim lying in the plane goes to both) */
	uint32_t *left = (uint32_t *)malloc(sizeof(uint32_t) * np), *right = (uint32_t *)malloc(sizeof(uint32_t) * np);
	uint32_t nl = 0, nr = 0;
	for(uint32_t i = 0; i < np; ++i)
	{
		const aabb_t *b = &c->tb[prims_idx[i]];
		float bl = vcomp(b->lo, best_axis), bh = vcomp(b->hi, best_axis);
		if(bl <= best_pos) left[nl++] = prims_idx[i];
		if(bh >= best_pos) right[nr++] = prims_idx[i];
	}
	if(nl == np && nr == np) { free(left); free(right); kd_make_leaf(s, node, prims_idx, np); return; }
	s->nodes[node].u.split = best_pos;
	s->nodes[node].flags = (uint32_t)best_axis;
	aabb_t lb = box, rb = box;
	if(best_axis == 0) { lb.hi.x = best_pos; rb.lo.x = best_pos; }
	else if(best_axis == 1) { lb.hi.y = best_pos; rb.lo.y = best_pos; }
	else { lb.hi.z = best_pos; rb.lo.z = best_pos; }
	uint32_t lchild = kd_alloc_node(s); /* == node + 1 */
	kd_build(c, lchild, lb, left, nl, depth + 1, bad_refines);
	free(left);
	uint32_t rchild = kd_alloc_node(s);
	s->nodes[node].flags = (s->nodes[node].flags & 3u) | (rchild << 2);
	kd_build(c, rchild, rb, right, nr, depth + 1, bad_refines);
	free(right);
}

static void build_tree(yor_scene *s)
{
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	int np = s->n_tris;
	s->n_nodes = 0; s->n_refs = 0;
	if(np <= 0) { s->build_seconds = 0; return; }
	aabb_t *tb = (aabb_t *)malloc(sizeof(aabb_t) * (size_t)np);
	aabb_t all;
	for(int i = 0; i < np; ++i)
	{
		const tri_t *t = &s->tris[i];
		tb[i].lo = V(fminf(fminf(t->a.x, t->b.x), t->c.x), fminf(fminf(t->a.y, t->b.y), t->c.y), fminf(fminf(t->a.z, t->b.z), t->c.z));
		tb[i].hi = V(fmaxf(fmaxf(t->a.x, t->b.x), t->c.x), fmaxf(fmaxf(t->a.y, t->b.y), t->c.y), fmaxf(fmaxf(t->a.z, t->b.z), t->c.z));
		if(i)
		{
			all.lo = V(fminf(all.lo.x, tb[i].lo.x), fminf(all.lo.y, tb[i].lo.y), fminf(all.lo.z, tb[i].lo.z));
			all.hi = V(fmaxf(all.hi.x, tb[i].hi.x), fmaxf(all.hi.y, tb[i].hi.y), fmaxf(all.hi.z, tb[i].hi.z));
		}
		else all = tb[i];
	}
	/* kdtree_triangle.cc:110-116: grow the tree bound by 0.1 % per side */
	{
		float *lo = &all.lo.x, *hi = &all.hi.x;
		for(int i = 0; i < 3; ++i)
		{
			double foo = (double)(hi[i] - lo[i]) * 0.001;
			lo[i] = (float)((double)lo[i] - foo); hi[i] = (float)((double)hi[i] + foo);
		}
	}
	s->tb_a = all.lo; s->tb_g = all.hi;
	build_ctx c;
	c.s = s; c.tb = tb;
	c.max_depth = (int)(7.0f + 1.66f * logf((float)np)); /* kdtree_triangle.cc:89 */
	if(c.max_depth > KD_MAX_STACK) c.max_depth = KD_MAX_STACK;
	c.cost_ratio = 0.8f; c.e_bonus = 0.33f;              /* scene.cc:818 */
	{
		double log_leaves = 1.442695f * log((double)np);   /* kdtree_triangle.cc:90,100 */
		if(log_leaves > 16.0) c.cost_ratio += (float)(0.25 * (log_leaves - 16.0));
	}
	uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np);
	for(int i = 0; i < np; ++i) idx[i] = (uint32_t)i;
	uint32_t root = kd_alloc_node(s);
	kd_build(&c, root, all, idx, (uint32_t)np, 0, 0);
	free(idx); free(tb);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	s->build_seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------ kd traversal
 * TriKdTree::intersect, kdtree_triangle.cc:684-837.  Same control flow (Havran entry/exit
 * stack); nodes are indices instead of pointers.  Tie rule: strictly `t_hit < z`, so the
 * first triangle visited wins (:782,806); leaf order here is ascending triangle index. */
typedef struct { uint32_t node; float t; v3 pb; int prev; } kdstack_t;
#define KD_NONE 0xFFFFFFFFu

static int kd_intersect(const yor_scene *s, v3 from, v3 dir, float tmin, float dist, int *tri_out, float *z_out, float *bu, float *bv, counters_t *cn)
{
	float z = dist;
	float a, b, t, t_hit;
	int hit = 0;
	*z_out = z;
	if(s->n_nodes == 0) return 0;
	if(!bound_cross(s->tb_a, s->tb_g, from, dir, &a, &b, dist)) return 0;
	v3 inv_dir = V((float)(1.0 / (double)dir.x), (float)(1.0 / (double)dir.y), (float)(1.0 / (double)dir.z));
	kdstack_t stack[KD_MAX_STACK];
	uint32_t far_child, curr = 0;
	int en_pt = 0;
	stack[en_pt].t = a;
	if(a >= 0.0) stack[en_pt].pb = vadd(from, vmul(dir, a));
	else stack[en_pt].pb = from;
	int ex_pt = 1;
	stack[ex_pt].t = b;
	stack[ex_pt].pb = vadd(from, vmul(dir, b));
	stack[ex_pt].node = KD_NONE;
	float cur_u = 0, cur_v = 0;
	while(curr != KD_NONE)
	{
		if(dist < stack[en_pt].t) break;
		while((s->nodes[curr].flags & 3u) != 3u)
		{
			int axis = (int)(s->nodes[curr].flags & 3u);
			float split_val = s->nodes[curr].u.split;
			uint32_t right = s->nodes[curr].flags >> 2;
			if(cn) cn->interior++;
			if(vcomp(stack[en_pt].pb, axis) <= split_val)
			{
				if(vcomp(stack[ex_pt].pb, axis) <= split_val) { curr++; continue; }
				if(vcomp(stack[ex_pt].pb, axis) == split_val) { curr = right; continue; }
				far_child = right;
				curr++;
			}
			else
			{
				if(split_val < vcomp(stack[ex_pt].pb, axis)) { curr = right; continue; }
				far_child = curr + 1;
				curr = right;
			}
			t = (split_val - vcomp(from, axis)) * vcomp(inv_dir, axis);
			int tmp = ex_pt;
			ex_pt++;
			if(ex_pt == en_pt) ex_pt++;
			{
				static const int np_axis[2][3] = {{1, 2, 0}, {2, 0, 1}};
				int next_axis = np_axis[0][axis], prev_axis = np_axis[1][axis];
				float pbv[3];
				stack[ex_pt].prev = tmp;
				stack[ex_pt].t = t;
				stack[ex_pt].node = far_child;
				pbv[axis] = split_val;
				pbv[next_axis] = vcomp(from, next_axis) + t * vcomp(dir, next_axis);
				pbv[prev_axis] = vcomp(from, prev_axis) + t * vcomp(dir, prev_axis);
				stack[ex_pt].pb = V(pbv[0], pbv[1], pbv[2]);
			}
		}
		{
			uint32_t n_primitives = s->nodes[curr].flags >> 2;
			uint32_t first = s->nodes[curr].u.first;
			if(cn) cn->leaves++;
			for(uint32_t i = 0; i < n_primitives; ++i)
			{
				uint32_t ti = s->leaf_refs[first + i];
				const tri_t *mp = &s->tris[ti];
				float uu, vv;
				if(cn) cn->tests++;
				if(tri_intersect(mp, from, dir, &t_hit, &uu, &vv))
				{
					if(t_hit < z && t_hit >= tmin)
					{
						if(mat_visible_primary(&s->mats[mp->mat]))
						{
							z = t_hit; *tri_out = (int)ti; cur_u = uu; cur_v = vv; hit = 1;
						}
					}
				}
			}
		}
		if(hit && z <= stack[ex_pt].t) { *z_out = z; *bu = cur_u; *bv = cur_v; return 1; }
		en_pt = ex_pt;
		curr = stack[ex_pt].node;
		ex_pt = stack[en_pt].prev;
	}
	*z_out = z; *bu = cur_u; *bv = cur_v;
	return hit;
}

/* TriKdTree::intersectS, kdtree_triangle.cc:840-977 */
/* forward declarations for TriKdTree::intersectTs, which asks materials about their transparency */
struct mat_s; struct sp_s;
static int mat_is_transparent_idx(const yor_scene *s, int mat);
static rgb mat_transparency_at(const yor_scene *s, int ti, v3 hit, float bu, float bv, v3 dir);

/* ts == NULL: TriKdTree::intersectS (:840-977).  ts != NULL: TriKdTree::intersectTs (:983-1162): transparent
 * triangles (each once: the std::set `filtered`) multiply their transparency into ts->filt instead of blocking the
 * ray; more than ts->max_depth of them block it. */
typedef struct { float ray_tmin; int max_depth; rgb filt; int depth; int n_seen; int seen[64]; } ts_state;
static int kd_intersect_s_impl(const yor_scene *s, v3 from, v3 dir, float dist, counters_t *cn, ts_state *ts)
{
	float a, b, t, t_hit;
	if(s->n_nodes == 0) return 0;
	if(!bound_cross(s->tb_a, s->tb_g, from, dir, &a, &b, dist)) return 0;
	v3 inv_dir = V(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
	kdstack_t stack[KD_MAX_STACK];
	uint32_t far_child, curr = 0;
	int en_pt = 0;
	stack[en_pt].t = a;
	if(a >= 0.0) stack[en_pt].pb = vadd(from, vmul(dir, a));
	else stack[en_pt].pb = from;
	int ex_pt = 1;
	stack[ex_pt].t = b;
	stack[ex_pt].pb = vadd(from, vmul(dir, b));
	stack[ex_pt].node = KD_NONE;
	while(curr != KD_NONE)
	{
		if(dist < stack[en_pt].t) break;
		while((s->nodes[curr].flags & 3u) != 3u)
		{
			int axis = (int)(s->nodes[curr].flags & 3u);
			float split_val = s->nodes[curr].u.split;
			uint32_t right = s->nodes[curr].flags >> 2;
			if(cn) cn->interior++;
			if(vcomp(stack[en_pt].pb, axis) <= split_val)
			{
				if(vcomp(stack[ex_pt].pb, axis) <= split_val) { curr++; continue; }
				if(vcomp(stack[ex_pt].pb, axis) == split_val) { curr = right; continue; }
				far_child = right;
				curr++;
			}
			else
			{
				if(split_val < vcomp(stack[ex_pt].pb, axis)) { curr = right; continue; }
				far_child = curr + 1;
				curr = right;
			}
			t = (split_val - vcomp(from, axis)) * vcomp(inv_dir, axis);
			int tmp = ex_pt;
			ex_pt++;
			if(ex_pt == en_pt) ex_pt++;
			{
				static const int np_axis[2][3] = {{1, 2, 0}, {2, 0, 1}};
				int next_axis = np_axis[0][axis], prev_axis = np_axis[1][axis];
				float pbv[3];
				stack[ex_pt].prev = tmp;
				stack[ex_pt].t = t;
				stack[ex_pt].node = far_child;
				pbv[axis] = split_val;
				pbv[next_axis] = vcomp(from, next_axis) + t * vcomp(dir, next_axis);
				pbv[prev_axis] = vcomp(from, prev_axis) + t * vcomp(dir, prev_axis);
				stack[ex_pt].pb = V(pbv[0], pbv[1], pbv[2]);
			}
		}
		{
			uint32_t n_primitives = s->nodes[curr].flags >> 2;
			uint32_t first = s->nodes[curr].u.first;
			if(cn) cn->leaves++;
			for(uint32_t i = 0; i < n_primitives; ++i)
			{
				const tri_t *mp = &s->tris[s->leaf_refs[first + i]];
				float uu, vv;
				if(cn) cn->tests++;
				if(tri_intersect(mp, from, dir, &t_hit, &uu, &vv))
				{
					if(t_hit < dist && t_hit >= (ts ? ts->ray_tmin : 0.f))
					{
						if(mat_visible_shadow(&s->mats[mp->mat]))
						{
							if(!ts || !mat_is_transparent_idx(s, mp->mat)) return 1;
							int ti = (int)s->leaf_refs[first + i], known = 0;
							for(int k = 0; k < ts->n_seen; ++k) if(ts->seen[k] == ti) { known = 1; break; }
							if(!known)
							{
								if(ts->n_seen < 64) ts->seen[ts->n_seen++] = ti;
								if(ts->depth >= ts->max_depth) return 1;
								v3 h = vadd(from, vmul(dir, t_hit));
								ts->filt = cmul(ts->filt, mat_transparency_at(s, ti, h, uu, vv, dir));
								++ts->depth;
							}
						}
					}
				}
			}
		}
		en_pt = ex_pt;
		curr = stack[ex_pt].node;
		ex_pt = stack[en_pt].prev;
	}
	return 0;
}
static int kd_intersect_s(const yor_scene *s, v3 from, v3 dir, float dist, counters_t *cn) { return kd_intersect_s_impl(s, from, dir, dist, cn, NULL); }

/* brute-force versions: the traversal-independent definition of the same queries */
static int brute_intersect(const yor_scene *s, v3 from, v3 dir, float tmin, float dist, int *tri_out, float *z_out, float *bu, float *bv)
{
	float z = dist, t_hit, uu, vv; int hit = 0;
	for(int i = 0; i < s->n_tris; ++i)
	{
		const tri_t *mp = &s->tris[i];
		if(tri_intersect(mp, from, dir, &t_hit, &uu, &vv))
			if(t_hit < z && t_hit >= tmin && mat_visible_primary(&s->mats[mp->mat])) { z = t_hit; *tri_out = i; *bu = uu; *bv = vv; hit = 1; }
	}
	*z_out = z;
	return hit;
}
static int brute_intersect_s(const yor_scene *s, v3 from, v3 dir, float dist)
{
	float t_hit, uu, vv;
	for(int i = 0; i < s->n_tris; ++i)
	{
		const tri_t *mp = &s->tris[i];
		if(tri_intersect(mp, from, dir, &t_hit, &uu, &vv))
			if(t_hit < dist && t_hit >= 0.f && mat_visible_shadow(&s->mats[mp->mat])) return 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ surface point
 * the subset of SurfacePoint (surface.h:58-100) the path uses */
typedef struct { v3 p, n, ng, nu, nv; int mat; int tri; float u, v; v3 orco_p, orco_ng; int has_uv, has_orco; v3 ds_du, ds_dv; } sp_t;   /* ds_du / ds_dv: dPdU / dPdV in shading space (triangle.cc:112-130), filled for bump mapping only */

/* Triangle::getSurface, triangle.cc:30-133 (dPdU / dPdV are only used by bump mapping: not restated) */
static void get_surface(const yor_scene *s, int ti, v3 hit, float bu, float bv, sp_t *sp)
{
	const tri_t *tr = &s->tris[ti];
	sp->ng = tr->ng;
	/* data.b_0_ = 1-u-v, b_1_ = u, b_2_ = v (triangle.h:252-254); getSurface names them u,v,w (:34) */
	float u = 1 - bu - bv, v = bu, w = bv;
	if(tr->smooth)
	{
		sp->n = vadd(vadd(vmul(tr->na, u), vmul(tr->nb, v)), vmul(tr->nc, w));
		sp->n = vnormalize(sp->n);
	}
	else sp->n = sp->ng;
	sp->tri = ti;
	if(s->tri_orco && s->tri_orco[9 * (size_t)ti] == s->tri_orco[9 * (size_t)ti])   /* has_orco_ is per mesh: a NaN first word = "this triangle's mesh has none" */
	{	/* :46-57 */
		const float *q = s->tri_orco + 9 * (size_t)ti;
		v3 p_0 = V(q[0], q[1], q[2]), p_1 = V(q[3], q[4], q[5]), p_2 = V(q[6], q[7], q[8]);
		sp->orco_p = vadd(vadd(vmul(p_0, u), vmul(p_1, v)), vmul(p_2, w));
		sp->orco_ng = vnormalize(vcross(vsub(p_1, p_0), vsub(p_2, p_0)));
		sp->has_orco = 1;
	}
	else { sp->orco_p = hit; sp->has_orco = 0; sp->orco_ng = sp->ng; }      /* :58-63 */
	v3 dp_du, dp_dv;
	const v3 p_0 = tr->a, p_1 = tr->b, p_2 = tr->c;
	if(s->tri_uv && s->tri_uv[6 * (size_t)ti] == s->tri_uv[6 * (size_t)ti])   /* has_uv_ is per mesh: a NaN first word = "this triangle's mesh has none" */
	{	/* :69-101 */
		const float *q = s->tri_uv + 6 * (size_t)ti;
		sp->u = u * q[0] + v * q[2] + w * q[4];
		sp->v = u * q[1] + v * q[3] + w * q[5];
		sp->has_uv = 1;
		float du_1 = q[0] - q[4], du_2 = q[2] - q[4], dv_1 = q[1] - q[5], dv_2 = q[3] - q[5];
		float det = du_1 * dv_2 - dv_1 * du_2;
		if(fabsf(det) > 1e-30f)
		{
			float invdet = 1.f / det;
			v3 dp_1 = vsub(p_0, p_2), dp_2 = vsub(p_1, p_2);
			dp_du = vmul(vsub(vmul(dp_1, dv_2), vmul(dp_2, dv_1)), invdet);
			dp_dv = vmul(vsub(vmul(dp_2, du_1), vmul(dp_1, du_2)), invdet);
		}
		else { dp_du = vsub(p_1, p_0); dp_dv = vsub(p_2, p_1); }
	}
	else { sp->u = 0.f; sp->v = 0.f; sp->has_uv = 0; dp_du = vsub(p_1, p_0); dp_dv = vsub(p_2, p_1); }                      /* :103-111 */
	dp_du = vnormalize(dp_du); dp_dv = vnormalize(dp_dv);                    /* :116-117 */
	sp->mat = tr->mat;
	sp->p = hit;
	create_cs(sp->n, &sp->nu, &sp->nv);
	sp->ds_du = V(vdot(sp->nu, dp_du), vdot(sp->nv, dp_du), vdot(sp->n, dp_du));      /* :124-130 */
	sp->ds_dv = V(vdot(sp->nu, dp_dv), vdot(sp->nv, dp_dv), vdot(sp->n, dp_dv));
}

/* Scene::intersect, scene.cc:896-927 */
static FILE *g_ray_log = NULL;
/* test hook (yor_set_trace): the samples renderTile hands to addSample and the closest-hit queries, in call order (single-threaded renders) */
static float *g_trace_samples = NULL, *g_trace_rays = NULL;
static uint64_t g_trace_samples_cap = 0, g_trace_rays_cap = 0, g_trace_n_samples = 0, g_trace_n_rays = 0;
static float g_trace_px = 0.f, g_trace_py = 0.f;      /* the pixel renderTile is at */
static int g_trace_shadow = 0;
static int scene_intersect(const yor_scene *s, v3 from, v3 dir, float tmin, float *tmax, sp_t *sp, counters_t *cn)
{
	float dis, z, bu = 0, bv = 0; int ti = -1;
	if(*tmax < 0) dis = INFINITY;
	else dis = *tmax;
	if(cn) cn->rays_closest++;
	const int got = kd_intersect(s, from, dir, tmin, dis, &ti, &z, &bu, &bv, cn);
	if(g_ray_log)      /* debugging aid (YOR_LOG_RAYS=<file>, single-threaded renders): every closest-hit query and its answer */
	{
		float rec[10] = {from.x, from.y, from.z, dir.x, dir.y, dir.z, tmin, *tmax, got ? z : -1.f, 0.f};
		int32_t tri = got ? ti : -1; memcpy(&rec[9], &tri, 4);
		fwrite(rec, sizeof rec, 1, g_ray_log);
	}
	if(g_trace_rays)
	{
		if(g_trace_n_rays < g_trace_rays_cap)
		{
			float *rec = g_trace_rays + 12 * g_trace_n_rays;
			rec[0] = from.x; rec[1] = from.y; rec[2] = from.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z; rec[6] = tmin; rec[7] = *tmax; rec[8] = got ? z : -1.f;
			int32_t tri = got ? ti : -1; memcpy(&rec[9], &tri, 4);
			rec[10] = g_trace_px; rec[11] = g_trace_py;
		}
		++g_trace_n_rays;
	}
	if(!got) return 0;
	v3 h = vadd(from, vmul(dir, z)); /* ray.from_ + z * ray.dir_ */
	get_surface(s, ti, h, bu, bv, sp);
	*tmax = z;
	return 1;
}

/* Scene::isShadowed, scene.cc:962-994 */
static int scene_is_shadowed(const yor_scene *s, v3 from, v3 dir, float tmin, float tmax, counters_t *cn)
{
	v3 sfrom = vadd(from, vmul(dir, tmin));
	float dis;
	if(tmax < 0) dis = INFINITY;
	else dis = tmax - 2 * tmin;
	if(cn) cn->rays_shadow++;
	const int verdict = kd_intersect_s(s, sfrom, dir, dis, cn);
	if(g_trace_rays && g_trace_shadow)
	{	/* any-hit queries in the same stream: the caller's ray, the verdict in slot 8, -2 in the triangle slot */
		if(g_trace_n_rays < g_trace_rays_cap)
		{
			float *rec = g_trace_rays + 12 * g_trace_n_rays;
			rec[0] = from.x; rec[1] = from.y; rec[2] = from.z; rec[3] = dir.x; rec[4] = dir.y; rec[5] = dir.z; rec[6] = tmin; rec[7] = tmax; rec[8] = (float)verdict;
			int32_t tri = -2; memcpy(&rec[9], &tri, 4);
			rec[10] = g_trace_px; rec[11] = g_trace_py;
		}
		++g_trace_n_rays;
	}
	return verdict;
}

/* Scene::isShadowed with transparent shadows, scene.cc:996-1035: filt = product of the transparencies passed */
static int scene_is_shadowed_ts(const yor_scene *s, v3 from, v3 dir, float tmin, float tmax, int max_depth, rgb *filt, counters_t *cn)
{
	v3 sfrom = vadd(from, vmul(dir, tmin));
	float dis;
	if(tmax < 0) dis = INFINITY;
	else dis = tmax - 2 * tmin;
	if(cn) cn->rays_shadow++;
	ts_state ts; memset(&ts, 0, sizeof ts);
	ts.ray_tmin = tmin; ts.max_depth = max_depth; ts.filt = C(1.f, 1.f, 1.f);      /* sray keeps ray.tmin_ (:998-999) */
	int r = kd_intersect_s_impl(s, sfrom, dir, dis, cn, &ts);
	*filt = ts.filt;
	return r;
}

/* ------------------------------------------------------------------ image textures (row N2)
 * ImageTexture, src/texture/texture_image.cc; adjustments include/texture/texture.h:202-275 */
typedef struct tex_s
{
	int w, h; float *px;
	int interp, clip, xrepeat, yrepeat, rot90, mirror_x, mirror_y, checker_even, checker_odd;
	float checker_dist;
	int cropx, cropy; float cropminx, cropmaxx, cropminy, cropmaxy;
	int adj_set, adj_clamp; float adj_int, adj_con, adj_sat, adj_hue, adj_r, adj_g, adj_b;
	int color_space; float gamma;
	int normalmap;
} tex_t;
typedef struct { float r, g, b, a; } rgba_t;
static inline rgba_t RA(float r, float g, float b, float a) { rgba_t c = {r, g, b, a}; return c; }
enum { TCL_EXTEND = 0, TCL_CLIP = 1, TCL_CLIPCUBE = 2, TCL_REPEAT = 3, TCL_CHECKER = 4 };

static void tex_configure(tex_t *t, const yor_texture_desc *d)
{
	memset(t, 0, sizeof *t);
	t->w = d->width; t->h = d->height;
	size_t n = (size_t)(d->width > 0 ? d->width : 0) * (size_t)(d->height > 0 ? d->height : 0) * 4;
	t->px = (float *)malloc((n ? n : 1) * sizeof(float));
	if(n) memcpy(t->px, d->texels, n * sizeof(float));
	t->interp = d->interpolate; t->clip = d->clip; t->xrepeat = d->xrepeat; t->yrepeat = d->yrepeat; t->rot90 = d->rot90;
	t->mirror_x = d->mirror_x; t->mirror_y = d->mirror_y; t->checker_even = d->checker_even; t->checker_odd = d->checker_odd;
	t->checker_dist = d->checker_dist;
	/* setCrop :217-222 */
	t->cropminx = d->cropmin_x; t->cropmaxx = d->cropmax_x; t->cropminy = d->cropmin_y; t->cropmaxy = d->cropmax_y;
	t->cropx = ((t->cropminx != 0.0) || (t->cropmaxx != 1.0));
	t->cropy = ((t->cropminy != 0.0) || (t->cropmaxy != 1.0));
	/* setAdjustments texture.h:142-200 */
	t->adj_int = d->adj_intensity; t->adj_con = d->adj_contrast; t->adj_sat = d->adj_saturation; t->adj_hue = d->adj_hue / 60.f;
	t->adj_clamp = d->adj_clamp; t->adj_r = d->adj_red; t->adj_g = d->adj_green; t->adj_b = d->adj_blue;
	t->adj_set = d->adj_intensity != 1.f || d->adj_contrast != 1.f || d->adj_saturation != 1.f || d->adj_hue != 0.f ||
	             d->adj_red != 1.f || d->adj_green != 1.f || d->adj_blue != 1.f || d->adj_clamp;
	t->color_space = d->color_space; t->gamma = d->gamma; t->normalmap = d->normalmap;
}

static inline rgba_t tex_pixel(const tex_t *t, int x, int y)
{
	const float *q = t->px + 4 * ((size_t)y * (size_t)t->w + (size_t)x);
	return RA(q[0], q[1], q[2], q[3]);
}

/* ImageTexture::doMapping :119-215; returns `outside` */
static int tex_do_mapping(const tex_t *t, v3 *texpt)
{
	int outside = 0;
	texpt->x = 0.5f * texpt->x + 0.5f; texpt->y = 0.5f * texpt->y + 0.5f; texpt->z = 0.5f * texpt->z + 0.5f;
	if(t->clip == TCL_REPEAT)
	{
		if(t->xrepeat > 1) texpt->x *= (float)t->xrepeat;
		if(t->yrepeat > 1) texpt->y *= (float)t->yrepeat;
		if(t->mirror_x && (int)ceilf(texpt->x) % 2 == 0) texpt->x = -texpt->x;
		if(t->mirror_y && (int)ceilf(texpt->y) % 2 == 0) texpt->y = -texpt->y;
		if(texpt->x > 1.f) texpt->x -= (int)texpt->x;
		else if(texpt->x < 0.f) texpt->x += 1 - (int)texpt->x;
		if(texpt->y > 1.f) texpt->y -= (int)texpt->y;
		else if(texpt->y < 0.f) texpt->y += 1 - (int)texpt->y;
	}
	if(t->cropx) texpt->x = t->cropminx + texpt->x * (t->cropmaxx - t->cropminx);
	if(t->cropy) texpt->y = t->cropminy + texpt->y * (t->cropmaxy - t->cropminy);
	if(t->rot90) { float tmp = texpt->x; texpt->x = texpt->y; texpt->y = tmp; }
	switch(t->clip)
	{
		case TCL_CLIPCUBE:
			if((texpt->x < 0) || (texpt->x > 1) || (texpt->y < 0) || (texpt->y > 1) || (texpt->z < -1) || (texpt->z > 1)) outside = 1;
			break;
		case TCL_CHECKER:
		{
			int xs = (int)floor(texpt->x), ys = (int)floor(texpt->y);
			texpt->x -= xs; texpt->y -= ys;
			if(!t->checker_odd && !((xs + ys) & 1)) { outside = 1; break; }
			if(!t->checker_even && ((xs + ys) & 1)) { outside = 1; break; }
			if(t->checker_dist < 1.0)
			{
				texpt->x = (float)(((double)texpt->x - 0.5) / (1.0 - (double)t->checker_dist) + 0.5);
				texpt->y = (float)(((double)texpt->y - 0.5) / (1.0 - (double)t->checker_dist) + 0.5);
			}
		}	/* falls through to clip */
		/* fall through */
		case TCL_CLIP:
			if((texpt->x < 0) || (texpt->x > 1) || (texpt->y < 0) || (texpt->y > 1)) outside = 1;
			break;
		case TCL_EXTEND:
			if(texpt->x > 0.99999f) texpt->x = 0.99999f; else if(texpt->x < 0) texpt->x = 0;
			if(texpt->y > 0.99999f) texpt->y = 0.99999f; else if(texpt->y < 0) texpt->y = 0;
			/* falls through */
		default: outside = 0;
	}
	return outside;
}

/* findTextureInterpolationCoordinates :224-289 */
static void tex_interp_coords(int *c0, int *c1, int *c2, int *c3, float *dec, float cf, int res, int repeat, int mirror)
{
	if(repeat)
	{
		*c1 = ((int)cf) % res;
		if(mirror)
		{
			if(cf < 0.f) { *c0 = 1 % res; *c2 = *c1; *c3 = *c0; *dec = -cf; }
			else if(cf >= res - 1.f) { *c0 = (res + res - 1) % res; *c2 = *c1; *c3 = *c0; *dec = cf - ((int)cf); }
			else
			{
				*c0 = (res + *c1 - 1) % res;
				*c2 = *c1 + 1; if(*c2 >= res) *c2 = (res + res - *c2) % res;
				*c3 = *c1 + 2; if(*c3 >= res) *c3 = (res + res - *c3) % res;
				*dec = cf - ((int)cf);
			}
		}
		else
		{
			if(cf > 0.f) { *c0 = (res + *c1 - 1) % res; *c2 = (*c1 + 1) % res; *c3 = (*c1 + 2) % res; *dec = cf - ((int)cf); }
			else { *c0 = 1 % res; *c2 = (res - 1) % res; *c3 = (res - 2) % res; *dec = -cf; }
		}
	}
	else
	{
		int ci = (int)cf;
		*c1 = ci < 0 ? 0 : (ci > res - 1 ? res - 1 : ci);
		if(cf > 0.f) *c2 = (*c1 + 1 < res - 1) ? *c1 + 1 : res - 1; else *c2 = 0;
		*c0 = (*c1 - 1 > 0) ? *c1 - 1 : 0;
		*c3 = (*c2 + 1 < res - 1) ? *c2 + 1 : res - 1;
		*dec = (float)((double)cf - floor((double)cf));
	}
}

/* noInterpolation :291-305, bilinearInterpolation :307-330 */
static rgba_t tex_interpolate(const tex_t *t, v3 p)
{
	int resx = t->w, resy = t->h;
	int x0, x1, x2, x3, y0, y1, y2, y3; float dx, dy;
	const int rep = t->clip == TCL_REPEAT;
	if(t->interp == 0)
	{
		float xf = (float)((double)(float)resx * ((double)p.x - floor((double)p.x)));
		float yf = (float)((double)(float)resy * ((double)p.y - floor((double)p.y)));
		tex_interp_coords(&x0, &x1, &x2, &x3, &dx, xf, resx, rep, t->mirror_x);
		tex_interp_coords(&y0, &y1, &y2, &y3, &dy, yf, resy, rep, t->mirror_y);
		return tex_pixel(t, x1, y1);
	}
	/* floor() here is C's double floor: the expression is evaluated in double and narrowed once (pinned by the golden vectors) */
	float xf = (float)((double)(float)resx * ((double)p.x - floor((double)p.x)) - (double)0.5f);
	float yf = (float)((double)(float)resy * ((double)p.y - floor((double)p.y)) - (double)0.5f);
	tex_interp_coords(&x0, &x1, &x2, &x3, &dx, xf, resx, rep, t->mirror_x);
	tex_interp_coords(&y0, &y1, &y2, &y3, &dy, yf, resy, rep, t->mirror_y);
	rgba_t c11 = tex_pixel(t, x1, y1), c21 = tex_pixel(t, x2, y1), c12 = tex_pixel(t, x1, y2), c22 = tex_pixel(t, x2, y2);
	float w11 = (1 - dx) * (1 - dy), w12 = (1 - dx) * dy, w21 = dx * (1 - dy), w22 = dx * dy;
	/* (w_11 * c_11) + (w_12 * c_12) + (w_21 * c_21) + (w_22 * c_22), left to right */
	rgba_t o;
	o.r = ((w11 * c11.r + w12 * c12.r) + w21 * c21.r) + w22 * c22.r;
	o.g = ((w11 * c11.g + w12 * c12.g) + w21 * c21.g) + w22 * c22.g;
	o.b = ((w11 * c11.b + w12 * c12.b) + w21 * c21.b) + w22 * c22.b;
	o.a = ((w11 * c11.a + w12 * c12.a) + w21 * c21.a) + w22 * c22.a;
	return o;
}

/* Rgb::rgbToHsv / hsvToRgb, color.h (used by the saturation / hue adjustments) */
static void rgb_to_hsv(rgba_t c, float *h, float *s, float *v)
{
	float r_1 = fmaxf_(c.r, 0.f), g_1 = fmaxf_(c.g, 0.f), b_1 = fmaxf_(c.b, 0.f);
	float max_component = fmaxf_(fmaxf_(r_1, g_1), b_1), min_component = fminf_(fminf_(r_1, g_1), b_1);
	float range = max_component - min_component;
	*v = max_component;
	if(fabsf(range) < 1.0e-6f) { *h = 0.f; *s = 0.f; }
	else if(max_component == r_1) { *h = fmodf((g_1 - b_1) / range, 6.f); *s = range / fmaxf_(*v, 1.0e-6f); }
	else if(max_component == g_1) { *h = ((b_1 - r_1) / range) + 2.f; *s = range / fmaxf_(*v, 1.0e-6f); }
	else if(max_component == b_1) { *h = ((r_1 - g_1) / range) + 4.f; *s = range / fmaxf_(*v, 1.0e-6f); }
	else { *h = 0.f; *s = 0.f; *v = 0.f; }
	if(*h < 0.f) *h += 6.f;
}
static void hsv_to_rgb(rgba_t *o, float h, float s, float v)
{
	float c = v * s;
	float x = c * (1.f - fabsf(fmodf(h, 2.f) - 1.f));
	float m = v - c;
	float r_1 = 0.f, g_1 = 0.f, b_1 = 0.f;
	if(h >= 0.f && h < 1.f) { r_1 = c; g_1 = x; b_1 = 0.f; }
	else if(h >= 1.f && h < 2.f) { r_1 = x; g_1 = c; b_1 = 0.f; }
	else if(h >= 2.f && h < 3.f) { r_1 = 0.f; g_1 = c; b_1 = x; }
	else if(h >= 3.f && h < 4.f) { r_1 = 0.f; g_1 = x; b_1 = c; }
	else if(h >= 4.f && h < 5.f) { r_1 = x; g_1 = 0.f; b_1 = c; }
	else if(h >= 5.f && h < 6.f) { r_1 = c; g_1 = 0.f; b_1 = x; }
	o->r = r_1 + m; o->g = g_1 + m; o->b = b_1 + m;
}

static rgba_t tex_clamp_rgb0(rgba_t c) { if(c.r < 0.f) c.r = 0.f; if(c.g < 0.f) c.g = 0.f; if(c.b < 0.f) c.b = 0.f; return c; }
/* Texture::applyAdjustments texture.h:202-255 */
static rgba_t tex_apply_adjustments(const tex_t *t, rgba_t c)
{
	if(!t->adj_set) return c;
	rgba_t ret = c;
	if(t->adj_int != 1.f || t->adj_con != 1.f)
	{
		ret.r = (c.r - 0.5f) * t->adj_con + t->adj_int - 0.5f;
		ret.g = (c.g - 0.5f) * t->adj_con + t->adj_int - 0.5f;
		ret.b = (c.b - 0.5f) * t->adj_con + t->adj_int - 0.5f;
	}
	if(t->adj_clamp) ret = tex_clamp_rgb0(ret);
	/* applyColorAdjustments */
	if(t->adj_r != 1.f) ret.r *= t->adj_r;
	if(t->adj_g != 1.f) ret.g *= t->adj_g;
	if(t->adj_b != 1.f) ret.b *= t->adj_b;
	if(t->adj_clamp) ret = tex_clamp_rgb0(ret);
	if(t->adj_sat != 1.f || t->adj_hue != 0.f)
	{
		float h = 0.f, sa = 0.f, v = 0.f;
		rgb_to_hsv(ret, &h, &sa, &v);
		sa *= t->adj_sat;
		h += t->adj_hue;
		if(h < 0.f) h += 6.f; else if(h > 6.f) h -= 6.f;
		hsv_to_rgb(&ret, h, sa, v);
		if(t->adj_clamp) ret = tex_clamp_rgb0(ret);
	}
	return ret;
}
static float tex_apply_ic_float(const tex_t *t, float f)
{	/* applyIntensityContrastAdjustments(float) :257-274 */
	if(!t->adj_set) return f;
	float ret = f;
	if(t->adj_int != 1.f || t->adj_con != 1.f) ret = (f - 0.5f) * t->adj_con + t->adj_int - 0.5f;
	if(t->adj_clamp) { if(ret < 0.f) ret = 0.f; else if(ret > 1.f) ret = 1.f; }
	return ret;
}

/* ImageTexture::getColor :75-88 */
static rgba_t tex_get_color(const tex_t *t, v3 p)
{
	v3 p_1 = V(p.x, -p.y, p.z);
	if(tex_do_mapping(t, &p_1)) return RA(0.f, 0.f, 0.f, 0.f);
	return tex_apply_adjustments(t, tex_interpolate(t, p_1));
}
/* sRgbFromLinearRgb color.h:359-364, colorSpaceFromLinearRgb :388-411 */
static rgba_t color_space_from_linear(rgba_t c, int color_space, float gamma)
{
	if(color_space == 0)
	{
		c.r = (c.r <= 0.0031308f) ? (c.r * 12.92f) : ((1.055f * yor_fpow(c.r, 0.416667f)) - 0.055f);
		c.g = (c.g <= 0.0031308f) ? (c.g * 12.92f) : ((1.055f * yor_fpow(c.g, 0.416667f)) - 0.055f);
		c.b = (c.b <= 0.0031308f) ? (c.b * 12.92f) : ((1.055f * yor_fpow(c.b, 0.416667f)) - 0.055f);
	}
	else if(color_space == 1)
	{
		float r = c.r, g = c.g, b = c.b;
		c.r = 0.412400f * r + 0.357600f * g + 0.180500f * b;
		c.g = 0.212600f * r + 0.715200f * g + 0.072200f * b;
		c.b = 0.019300f * r + 0.119200f * g + 0.950500f * b;
	}
	else if(color_space == 3 && gamma != 1.f)
	{
		if(gamma <= 0.f) gamma = 1.0e-2f;
		float inv = 1.f / gamma;
		c.r = yor_fpow(c.r, inv); c.g = yor_fpow(c.g, inv); c.b = yor_fpow(c.b, inv);
	}
	return c;
}
/* Texture::getFloat texture.h:51 = applyIntensityContrastAdjustments(getRawColor(p).col2Bri()); getRawColor :90-104 */
static float tex_get_float(const tex_t *t, v3 p)
{
	rgba_t c = color_space_from_linear(tex_get_color(t, p), t->color_space, t->gamma);
	return tex_apply_ic_float(t, (0.2126f * c.r + 0.7152f * c.g + 0.0722f * c.b));
}

/* ------------------------------------------------------------------ shader nodes (row N2)
 * TextureMapperNode / ValueNode / MixNode (shader_node_basic.cc), LayerNode (shader_node_layer.cc),
 * textureRgbBlend__ / textureValueBlend__ (shader_node.h:115-223) */
typedef struct node_s
{
	int type;
	int tex, texco, mapping, map_x, map_y, map_z; v3 scale, offset; float mtx[16]; int do_scalar;
	rgba_t color; float value;
	int mode; float cfactor; int input1, input2, factor; rgba_t col1, col2;
	int input, upper; unsigned texflag; float colfac, valfac, def_val; rgba_t def_col, upper_col; float upper_val;
	int do_color, do_scalar_l, color_input, use_alpha;
	float bump_strength;      /* texture_mapper "bump_strength" as given; setup() (:34-59) turns it into bump_str_ once the texture's size is known */
} node_t;
typedef struct { rgba_t col; float f; } node_result_t;
enum { TXF_RGBTOINT = 1, TXF_STENCIL = 2, TXF_NEGATIVE = 4, TXF_ALPHAMIX = 8 };
enum { MN_MIX = 0, MN_ADD, MN_MULT, MN_SUB, MN_SCREEN, MN_DIV, MN_DIFF, MN_DARK, MN_LIGHT, MN_OVERLAY };
enum { TC_UV = 0, TC_GLOB, TC_ORCO, TC_TRAN, TC_NOR, TC_REFL, TC_WIN, TC_STICK, TC_STRESS, TC_TAN };

static void node_configure(node_t *n, const yor_node_desc *d)
{
	memset(n, 0, sizeof *n);
	n->type = d->type;
	n->tex = d->texture; n->texco = d->texco; n->mapping = d->mapping;
	int map[3];
	for(int i = 0; i < 3; ++i) { map[i] = d->proj[i]; if(map[i] < 0) map[i] = 0; if(map[i] > 3) map[i] = 3; }   /* :406 */
	n->map_x = map[0]; n->map_y = map[1]; n->map_z = map[2];
	n->scale = V(d->scale[0], d->scale[1], d->scale[2]);
	n->offset = V(2 * d->offset[0], 2 * d->offset[1], 2 * d->offset[2]);                                       /* :411 */
	memcpy(n->mtx, d->mtx, sizeof n->mtx);
	n->do_scalar = d->do_scalar;
	n->color = RA(d->color[0], d->color[1], d->color[2], d->color[3]); n->value = d->scalar;
	n->mode = d->mode; n->cfactor = d->cfactor; n->input1 = d->input1; n->input2 = d->input2; n->factor = d->factor;
	n->col1 = RA(d->col1[0], d->col1[1], d->col1[2], d->col1[3]); n->col2 = RA(d->col2[0], d->col2[1], d->col2[2], d->col2[3]);
	n->input = d->input; n->upper = d->upper_layer;
	n->texflag = (d->no_rgb ? TXF_RGBTOINT : 0) | (d->stencil ? TXF_STENCIL : 0) | (d->negative ? TXF_NEGATIVE : 0) | (d->use_alpha ? TXF_ALPHAMIX : 0);
	n->colfac = d->colfac; n->valfac = d->valfac; n->def_val = d->def_val;
	n->def_col = RA(d->def_col[0], d->def_col[1], d->def_col[2], 1.f);
	n->upper_col = RA(d->upper_col[0], d->upper_col[1], d->upper_col[2], d->upper_col[3]); n->upper_val = d->upper_val;
	n->do_color = d->do_color; n->do_scalar_l = d->do_scalar_l; n->color_input = d->color_input; n->use_alpha = d->use_alpha;
	n->bump_strength = d->bump_strength;
}

/* fAcos__ util_math_optimizations.h:255-261 */
static float f_acos(float x) { if(x <= -1.0) return (float)Y_M_PI; else if(x >= 1.0) return 0.0f; else return (float)acos((double)x); }
#define Y_M_1_PI 0.31830988618379067154

/* TextureMapperNode::doMapping :127-155 */
static v3 mapper_do_mapping(const node_t *n, v3 p, v3 ng)
{
	v3 texpt = p;
	if(n->texco == TC_UV) texpt = V(2.0f * texpt.x - 1.0f, 2.0f * texpt.y - 1.0f, texpt.z);
	float texmap[4] = {0, texpt.x, texpt.y, texpt.z};
	texpt.x = texmap[n->map_x]; texpt.y = texmap[n->map_y]; texpt.z = texmap[n->map_z];
	switch(n->mapping)
	{
		case 2:
		{	/* tubemap__ :62-74 */
			v3 res; res.y = texpt.z;
			float d = texpt.x * texpt.x + texpt.y * texpt.y;
			if(d > 0) { res.z = (float)(1.0 / (double)yor_fsqrt(d)); res.x = (float)(-atan2((double)texpt.x, (double)texpt.y) * Y_M_1_PI); }
			else res.x = res.z = 0;
			texpt = res; break;
		}
		case 3:
		{	/* spheremap__ :77-88 */
			v3 res = V(0.f, 0.f, 0.f);
			float d = texpt.x * texpt.x + texpt.y * texpt.y + texpt.z * texpt.z;
			if(d > 0)
			{
				res.z = yor_fsqrt(d);
				if((texpt.x != 0) && (texpt.y != 0)) res.x = (float)(-atan2((double)texpt.x, (double)texpt.y) * Y_M_1_PI);
				res.y = (float)((double)1.0f - (double)2.0f * ((double)f_acos(texpt.z / res.z) * Y_M_1_PI));
			}
			texpt = res; break;
		}
		case 1:
		{	/* cubemap__ :91-115 */
			static const int ma[3][3] = {{1, 2, 0}, {0, 2, 1}, {0, 1, 2}};
			int axis;
			if(fabsf(ng.z) >= fabsf(ng.x) && fabsf(ng.z) >= fabsf(ng.y)) axis = 2;
			else if(fabsf(ng.y) >= fabsf(ng.x) && fabsf(ng.y) >= fabsf(ng.z)) axis = 1;
			else axis = 0;
			float pc[3] = {texpt.x, texpt.y, texpt.z};
			texpt = V(pc[ma[axis][0]], pc[ma[axis][1]], pc[ma[axis][2]]);
			break;
		}
		default: break;
	}
	texpt = V(texpt.x * n->scale.x + n->offset.x, texpt.y * n->scale.y + n->offset.y, texpt.z * n->scale.z + n->offset.z);
	return texpt;
}

static v3 mtx_point(const float *m, v3 p)      /* Matrix4 * Point3: with translation */
{
	return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static v3 mtx_vec(const float *m, v3 v)        /* Matrix4 * Vec3: without */
{
	return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

/* the texture coordinates a mapper starts from: TextureMapperNode::getCoords :163-190 */
static void mapper_get_coords(const node_t *n, const camera_t *cam, const sp_t *sp, v3 *texpt, v3 *ng)
{
	switch(n->texco)
	{
		case TC_UV: *texpt = V(sp->u, sp->v, 0.f); *ng = sp->ng; break;
		case TC_ORCO: *texpt = sp->orco_p; *ng = sp->orco_ng; break;
		case TC_TRAN: *texpt = mtx_point(n->mtx, sp->p); *ng = mtx_vec(n->mtx, sp->ng); break;
		case TC_WIN:
		{
			v3 dir = vsub(sp->p, cam->position);
			float dx = vdot(dir, cam->cam_x), dy = vdot(dir, cam->cam_y), dz = vdot(dir, cam->cam_z);
			*texpt = V(2.0f * dx * cam->focal / dz, -2.0f * dy * cam->focal / (dz * cam->aspect_ratio), 0.f);
			*ng = sp->ng; break;
		}
		case TC_NOR: *texpt = V(vdot(sp->n, cam->cam_x), -vdot(sp->n, cam->cam_y), 0.f); *ng = sp->ng; break;
		default: *texpt = sp->p; *ng = sp->ng; break;
	}
}
/* NodeMaterial::evalBump's node pass (material_node.cc:132-139): evalDerivative of every node in order — TextureMapperNode
 * :232-343 (image textures: discrete, never normal maps here), LayerNode :122-152, the base class's zero for the others
 * (shader_node.h:87-88).  stack[k].col = (du, dv, 0, stencil alpha). */
static void nodes_eval_derivative(const node_t *nodes, int n_nodes, const tex_t *tex, int n_tex, const camera_t *cam, const sp_t *sp, node_result_t *stack)
{
	for(int k = 0; k < n_nodes; ++k)
	{
		const node_t *n = &nodes[k];
		node_result_t res; res.col = RA(0.f, 0.f, 0.f, 0.f); res.f = 0.f;
		if(n->type == YOR_NODE_TEXTURE_MAPPER && n->tex >= 0 && n->tex < n_tex)
		{
			const tex_t *t = &tex[n->tex];
			/* setup() :34-59 */
			const float d_u = 1.f / (float)t->w, d_v = 1.f / (float)t->h;
			float bump_str = n->bump_strength;
			bump_str /= vlength(n->scale);
			if(!t->normalmap) bump_str /= 100.0f;
			v3 texpt, ng;
			float du = 0.0f, dv = 0.0f;
			mapper_get_coords(n, cam, sp, &texpt, &ng);
			if(t->normalmap)
			{	/* :245-258 / :287-312: both branches read the normal from the texture's raw colour */
				texpt = mapper_do_mapping(n, texpt, ng);
				rgba_t color = color_space_from_linear(tex_get_color(t, texpt), t->color_space, t->gamma);      /* getRawColor */
				v3 norm = V(2.f * color.r - 1.f, 2.f * color.g - 1.f, 2.f * color.b - 1.f);
				norm = vnormalize(norm);
				if(fabsf(norm.z) > 1e-30f)
				{
					float nf = (float)(1.0 / (double)norm.z * (double)bump_str);
					du = norm.x * nf; dv = norm.y * nf;
				}
				else du = dv = 0.f;
			}
			else if(sp->has_uv && n->texco == TC_UV)
			{
				texpt = mapper_do_mapping(n, texpt, ng);
				v3 i_0 = V(texpt.x - d_u, texpt.y - 0.f, texpt.z - 0.f), i_1 = V(texpt.x + d_u, texpt.y + 0.f, texpt.z + 0.f);
				v3 j_0 = V(texpt.x - 0.f, texpt.y - d_v, texpt.z - 0.f), j_1 = V(texpt.x + 0.f, texpt.y + d_v, texpt.z + 0.f);
				float dfdu = (tex_get_float(t, i_0) - tex_get_float(t, i_1)) / d_u;
				float dfdv = (tex_get_float(t, j_0) - tex_get_float(t, j_1)) / d_v;
				v3 vec_u = sp->ds_du, vec_v = sp->ds_dv;
				vec_u.z = dfdu; vec_v.z = dfdv;
				v3 norm = vnormalize(vcross(vec_u, vec_v));
				if(fabsf(norm.z) > 1e-30f)
				{
					float nf = (float)(1.0 / (double)norm.z * (double)bump_str);
					du = norm.x * nf; dv = norm.y * nf;
				}
				else du = dv = 0.f;
			}
			else
			{
				v3 i_0 = mapper_do_mapping(n, vsub(texpt, vmul(sp->nu, d_u)), ng), i_1 = mapper_do_mapping(n, vadd(texpt, vmul(sp->nu, d_u)), ng);
				v3 j_0 = mapper_do_mapping(n, vsub(texpt, vmul(sp->nv, d_v)), ng), j_1 = mapper_do_mapping(n, vadd(texpt, vmul(sp->nv, d_v)), ng);
				du = (tex_get_float(t, i_0) - tex_get_float(t, i_1)) / d_u;
				dv = (tex_get_float(t, j_0) - tex_get_float(t, j_1)) / d_v;
				du *= bump_str; dv *= bump_str;
				if(n->texco != TC_UV) { du = -du; dv = -dv; }
			}
			res.col = RA(du, dv, 0.f, 0.f);
		}
		else if(n->type == YOR_NODE_LAYER)
		{
			float rdu = 0.f, rdv = 0.f, stencil_tin = 1.f;
			if(n->upper >= 0) { rdu = stack[n->upper].col.r; rdv = stack[n->upper].col.g; stencil_tin = stack[n->upper].col.a; }
			float tdu = stack[n->input].col.r, tdv = stack[n->input].col.g;
			if(n->texflag & TXF_NEGATIVE) { tdu = -tdu; tdv = -tdv; }
			rdu += tdu; rdv += tdv;
			res.col = RA(rdu, rdv, 0.f, stencil_tin);
		}
		stack[k] = res;
	}
}
/* Material::applyBump, material.cc:77-84 */
static void apply_bump(sp_t *sp, float df_dnu, float df_dnv)
{
	sp->nu = vadd(sp->nu, vmul(sp->n, df_dnu));
	sp->nv = vadd(sp->nv, vmul(sp->n, df_dnv));
	sp->n = vnormalize(vcross(sp->nu, sp->nv));
	sp->nu = vnormalize(sp->nu);
	sp->nv = vnormalize(vcross(sp->n, sp->nu));
}

/* one pass over a material's nodes in evaluation order (NodeMaterial::evalNodes, material_node.cc:83-86) */
static void nodes_eval(const node_t *nodes, int n_nodes, const tex_t *tex, int n_tex, const camera_t *cam, const sp_t *sp, node_result_t *stack)
{
	for(int k = 0; k < n_nodes; ++k)
	{
		const node_t *n = &nodes[k];
		node_result_t res; res.col = RA(0.f, 0.f, 0.f, 0.f); res.f = 0.f;
		if(n->type == YOR_NODE_TEXTURE_MAPPER)
		{	/* getCoords :163-190, eval :193-229 (no mipmaps) */
			v3 texpt, ng;
			switch(n->texco)
			{
				case TC_UV: texpt = V(sp->u, sp->v, 0.f); ng = sp->ng; break;
				case TC_ORCO: texpt = sp->orco_p; ng = sp->orco_ng; break;
				case TC_TRAN: texpt = mtx_point(n->mtx, sp->p); ng = mtx_vec(n->mtx, sp->ng); break;
				case TC_WIN:
				{	/* PerspectiveCamera::screenproject camera_perspective.cc:158-173 */
					v3 dir = vsub(sp->p, cam->position);
					float dx = vdot(dir, cam->cam_x), dy = vdot(dir, cam->cam_y), dz = vdot(dir, cam->cam_z);
					texpt = V(2.0f * dx * cam->focal / dz, -2.0f * dy * cam->focal / (dz * cam->aspect_ratio), 0.f);
					ng = sp->ng; break;
				}
				case TC_NOR: texpt = V(vdot(sp->n, cam->cam_x), -vdot(sp->n, cam->cam_y), 0.f); ng = sp->ng; break;
				default: texpt = sp->p; ng = sp->ng; break;
			}
			texpt = mapper_do_mapping(n, texpt, ng);
			if(n->tex >= 0 && n->tex < n_tex)
			{
				res.col = tex_get_color(&tex[n->tex], texpt);
				res.f = n->do_scalar ? tex_get_float(&tex[n->tex], texpt) : 0.f;
			}
		}
		else if(n->type == YOR_NODE_VALUE) { res.col = n->color; res.f = n->value; }
		else if(n->type == YOR_NODE_MIX)
		{	/* MixNode::getInputs shader_node_basic.h:98-124 (val_1_ / val_2_ are never set by the reference: 0 here) */
			float f_2 = (n->factor >= 0) ? stack[n->factor].f : n->cfactor;
			rgba_t c1, c2; float fin_1, fin_2;
			if(n->input1 >= 0) { c1 = stack[n->input1].col; fin_1 = stack[n->input1].f; } else { c1 = n->col1; fin_1 = 0.f; }
			if(n->input2 >= 0) { c2 = stack[n->input2].col; fin_2 = stack[n->input2].f; } else { c2 = n->col2; fin_2 = 0.f; }
			float f_1 = 1.f - f_2;
			float *a = &c1.r, *b = &c2.r;
			switch(n->mode)
			{
				case MN_ADD: for(int i = 0; i < 4; ++i) a[i] += f_2 * b[i]; fin_1 += f_2 * fin_2; break;
				case MN_MULT: for(int i = 0; i < 4; ++i) a[i] *= f_1 + f_2 * b[i]; break;      /* fin_1 is left as it is (:563-566) */
				case MN_SUB: for(int i = 0; i < 4; ++i) a[i] -= f_2 * b[i]; fin_1 -= f_2 * fin_2; break;
				case MN_SCREEN:
					for(int i = 0; i < 4; ++i) a[i] = 1.f - (f_1 + f_2 * (1.f - b[i])) * (1.f - a[i]);
					fin_1 = (float)(1.0 - (double)((f_1 + f_2 * (1.f - fin_2)) * (1.f - fin_1)));
					break;
				case MN_DIFF:
					for(int i = 0; i < 4; ++i) a[i] = f_1 * a[i] + f_2 * fabsf(a[i] - b[i]);
					fin_1 = f_1 * fin_1 + f_2 * fabsf(fin_1 - fin_2);
					break;
				case MN_DARK:
					for(int i = 0; i < 4; ++i) { b[i] *= f_2; if(b[i] < a[i]) a[i] = b[i]; }
					fin_2 *= f_2; if(fin_2 < fin_1) fin_1 = fin_2;
					break;
				case MN_LIGHT:
					for(int i = 0; i < 4; ++i) { b[i] *= f_2; if(b[i] > a[i]) a[i] = b[i]; }
					fin_2 *= f_2; if(fin_2 > fin_1) fin_1 = fin_2;
					break;
				case MN_OVERLAY:
				{
					rgba_t o; float *oo = &o.r;
					for(int i = 0; i < 4; ++i)
						oo[i] = (a[i] < 0.5f) ? a[i] * (f_1 + 2.0f * f_2 * b[i]) : (float)(1.0 - ((double)f_1 + (double)(2.0f * f_2) * (1.0 - (double)b[i])) * (1.0 - (double)a[i]));
					fin_1 = (fin_1 < 0.5f) ? fin_1 * (f_1 + 2.0f * f_2 * fin_2) : (float)(1.0 - ((double)f_1 + (double)(2.0f * f_2) * (1.0 - (double)fin_2)) * (1.0 - (double)fin_1));
					c1 = o;
					break;
				}
				default:       /* MnMix, and MnDiv (no class of its own: MixNode::factory :697-703) */
					for(int i = 0; i < 4; ++i) a[i] = f_1 * a[i] + f_2 * b[i];
					fin_1 = f_1 * fin_1 + f_2 * fin_2;
					break;
			}
			res.col = c1; res.f = fin_1;
		}
		else if(n->type == YOR_NODE_LAYER)
		{	/* LayerNode::eval shader_node_layer.cc:29-117 */
			rgba_t rcol, texcolor = RA(0.f, 0.f, 0.f, 0.f);
			float rval, tin = 0.f, ta = 1.f, stencil_tin;
			rcol = (n->upper >= 0) ? stack[n->upper].col : n->upper_col;
			rval = (n->upper >= 0) ? stack[n->upper].f : n->upper_val;
			stencil_tin = rcol.a;
			int tex_rgb = n->color_input;
			if(n->color_input) { texcolor = stack[n->input].col; ta = texcolor.a; }
			else tin = stack[n->input].f;
			if(n->texflag & TXF_RGBTOINT) { tin = (0.2126f * texcolor.r + 0.7152f * texcolor.g + 0.0722f * texcolor.b); tex_rgb = 0; }
			if(n->texflag & TXF_NEGATIVE)
			{
				if(tex_rgb) texcolor = RA(1.f - texcolor.r, 1.f - texcolor.g, 1.f - texcolor.b, 1.f - texcolor.a);
				tin = 1.f - tin;
			}
			float fact;
			if(n->texflag & TXF_STENCIL)
			{
				if(tex_rgb) { fact = ta; ta *= stencil_tin; stencil_tin *= fact; }
				else { fact = tin; tin *= stencil_tin; stencil_tin *= fact; }
			}
			if(n->do_color)
			{
				if(!tex_rgb) texcolor = n->def_col; else tin = ta;
				float tt = tin > 1.f ? 1.f : (tin < 0.f ? 0.f : tin);
				/* textureRgbBlend__(texcolor, rcol, tt, stencil_tin * colfac_, mode_) on Rgb */
				float facg = stencil_tin * n->colfac, f = tt;
				float tex[3] = {texcolor.r, texcolor.g, texcolor.b}, out[3] = {rcol.r, rcol.g, rcol.b}, r[3];
				switch(n->mode)
				{
					case MN_MULT: f *= facg; for(int i = 0; i < 3; ++i) r[i] = ((1.f - facg) + f * tex[i]) * out[i]; break;
					case MN_SCREEN: f *= facg; for(int i = 0; i < 3; ++i) r[i] = 1.0f - ((1.f - facg) + f * (1.0f - tex[i])) * (1.0f - out[i]); break;
					case MN_SUB: f = -f; f *= facg; for(int i = 0; i < 3; ++i) r[i] = f * tex[i] + out[i]; break;
					case MN_ADD: f *= facg; for(int i = 0; i < 3; ++i) r[i] = f * tex[i] + out[i]; break;
					case MN_DIV:
						f *= facg;
						for(int i = 0; i < 3; ++i) { float it = (tex[i] != 0.f) ? 1.f / tex[i] : tex[i]; r[i] = (1.f - f) * out[i] + (f * out[i]) * it; }
						break;
					case MN_DIFF: f *= facg; for(int i = 0; i < 3; ++i) r[i] = (1.f - f) * out[i] + f * fabsf(tex[i] - out[i]); break;
					case MN_DARK: f *= facg; for(int i = 0; i < 3; ++i) { float c = f * tex[i]; r[i] = (out[i] < c) ? out[i] : c; } break;
					case MN_LIGHT: f *= facg; for(int i = 0; i < 3; ++i) { float c = f * tex[i]; r[i] = (out[i] > c) ? out[i] : c; } break;
					default: f *= facg; for(int i = 0; i < 3; ++i) r[i] = f * tex[i] + (1.f - f) * out[i]; break;
				}
				rcol = RA(r[0], r[1], r[2], 1.f);
				rcol = tex_clamp_rgb0(rcol);
			}
			if(n->do_scalar_l)
			{
				if(tex_rgb)
				{
					if(n->use_alpha) { tin = ta; if(n->texflag & TXF_NEGATIVE) tin = 1.f - tin; }
					else tin = (0.2126f * texcolor.r + 0.7152f * texcolor.g + 0.0722f * texcolor.b);
				}
				/* textureValueBlend__(default_val_, rval, tin, stencil_tin * valfac_, mode_) */
				float facg = stencil_tin * n->valfac, f = tin * facg, facm = 1.f - f, tex = n->def_val, out = rval;
				switch(n->mode)
				{
					case MN_MULT: facm = 1.f - facg; rval = (facm + f * tex) * out; break;
					case MN_SCREEN: facm = 1.f - facg; rval = 1.f - (facm + f * (1.f - tex)) * (1.f - out); break;
					case MN_SUB: f = -f; rval = f * tex + out; break;
					case MN_ADD: rval = f * tex + out; break;
					case MN_DIV: rval = (tex == 0.f) ? 0.f : facm * out + f * out / tex; break;
					case MN_DIFF: rval = facm * out + f * fabsf(tex - out); break;
					case MN_DARK: { float c = f * tex; rval = (c < out) ? c : out; break; }
					case MN_LIGHT: { float c = f * tex; rval = (c > out) ? c : out; break; }
					default: rval = f * tex + facm * out; break;
				}
				if(rval < 0.f) rval = 0.f;
			}
			rcol.a = stencil_tin;
			res.col = rcol; res.f = rval;
		}
		stack[k] = res;
	}
}

/* ------------------------------------------------------------------ materials */
typedef struct { float component[4]; float m_diffuse, m_glossy, p_diffuse; } bsdf_dat; /* SdDat / MDatT */
typedef struct { float s_1, s_2, pdf; unsigned flags, sampled_flags; } sample_t;         /* material.h:68-78 */


/* BeerVolumeHandler::BeerVolumeHandler(acol, dist), volumehandler_beer.cc:28-35 */
static rgb beer_sigma(const float acol[3], double dist)
{
	const float maxlog = (float)log(1e38);
	rgb s;
	s.r = (acol[0] > 1e-38) ? (float)-log(acol[0]) : maxlog;
	s.g = (acol[1] > 1e-38) ? (float)-log(acol[1]) : maxlog;
	s.b = (acol[2] > 1e-38) ? (float)-log(acol[2]) : maxlog;
	if(dist != 0.f) s = cscale(s, (float)(1.f / dist));
	return s;
}
/* BeerVolumeHandler::transmittance, volumehandler_beer.cc:37-48; fExp__(x) = fExp2__(M_LOG2E * x) (util_math_optimizations.h) */
static rgb beer_transmittance(rgb sigma, float tmax)
{
	if(tmax < 0.f || tmax > 1e30f) return C(0.f, 0.f, 0.f);
	const float dist = tmax;
	const float l2e = (float)1.4426950408889634074;
	return C(yor_fexp2(l2e * (-dist * sigma.r)), yor_fexp2(l2e * (-dist * sigma.g)), yor_fexp2(l2e * (-dist * sigma.b)));
}

static void mat_copy_nodes(mat_t *m, const yor_material_desc *d)      /* glossy / coated glossy shader slots (material_glossy.cc:504-511, material_coated_glossy.cc:572-582) */
{
	m->sh_diffuse = m->sh_mirror_color = m->sh_mirror = m->sh_transparency = m->sh_translucency = m->sh_sigma_oren = m->sh_diffuse_refl = m->sh_ior = -1;
	m->sh_glossy = m->sh_glossy_reflect = m->sh_exponent = m->sh_filter_color = -1;
	if(d->n_nodes > 0 && d->nodes)
	{
		m->n_nodes = d->n_nodes;
		m->nodes = (node_t *)calloc((size_t)d->n_nodes, sizeof(node_t));
		for(int k = 0; k < d->n_nodes; ++k) node_configure(&m->nodes[k], &d->nodes[k]);
		m->sh_diffuse = d->sh_diffuse; m->sh_sigma_oren = d->sh_sigma_oren; m->sh_diffuse_refl = d->sh_diffuse_refl;
		m->sh_glossy = d->sh_glossy; m->sh_glossy_reflect = d->sh_glossy_reflect; m->sh_exponent = d->sh_exponent;
		if(d->type == YOR_MAT_COATED_GLOSSY) { m->sh_mirror_color = d->sh_mirror_color; m->sh_mirror = d->sh_mirror; m->sh_ior = d->sh_ior; }
		if(d->type == YOR_MAT_GLASS) { m->sh_diffuse = m->sh_sigma_oren = m->sh_diffuse_refl = m->sh_glossy = m->sh_glossy_reflect = m->sh_exponent = -1;
		                               m->sh_mirror_color = d->sh_mirror_color; m->sh_filter_color = d->sh_filter_color; m->sh_ior = d->sh_ior; }   /* material_glass.cc:419-422 */
	}
}
static void mat_configure_(mat_t *m, const yor_material_desc *d);
static void mat_configure(mat_t *m, const yor_material_desc *d)
{
	mat_configure_(m, d);
	m->n_bump = 0; m->sh_bump = -1; m->bump_nodes = NULL;
	if(d->n_bump_nodes > 0 && d->bump_nodes && d->sh_bump >= 0 && d->sh_bump < d->n_bump_nodes)
	{
		m->n_bump = d->n_bump_nodes; m->sh_bump = d->sh_bump;
		m->bump_nodes = (node_t *)calloc((size_t)d->n_bump_nodes, sizeof(node_t));
		for(int k = 0; k < d->n_bump_nodes; ++k) node_configure(&m->bump_nodes[k], &d->bump_nodes[k]);
	}
	m->additional_depth = d->additional_depth; m->transp_bias_factor = d->transp_bias_factor; m->transp_bias_mult = d->transp_bias_mult;
}
static void mat_configure_(mat_t *m, const yor_material_desc *d)
{
	memset(m, 0, sizeof *m);
	m->type = d->type; m->visibility = d->visibility; m->receive_shadows = d->receive_shadows; m->flat = d->flat_material;
	if(d->type == YOR_MAT_SHINYDIFFUSE)
	{
		/* ctor material_shiny_diffuse.cc:26-36 and factory :599-690 */
		m->diffuse_color = C(d->color[0], d->color[1], d->color[2]);
		m->mirror_color = C(d->mirror_color[0], d->mirror_color[1], d->mirror_color[2]);
		m->diffuse_strength = d->diffuse_reflect; m->mirror_strength = d->specular_reflect;
		m->transparency_strength = d->transparency; m->translucency_strength = d->translucency;
		m->emit_strength = d->emit; m->transmit_filter = d->transmit_filter;
		m->emit_color = cscale(m->diffuse_color, d->emit);
		m->flags = BSDF_NONE;
		if(m->emit_strength > 0.f) m->flags |= BSDF_EMIT;
		m->ior_squared = 1.f;
		if(d->fresnel_effect) { m->ior_squared = d->ior * d->ior; m->has_fresnel = 1; }
		if(d->oren_nayar)
		{	/* initOrenNayar :190-196 */
			double sigma_squared = d->sigma * d->sigma;
			m->oren_a = (float)(1.0 - 0.5 * (sigma_squared / (sigma_squared + 0.33)));
			m->oren_b = (float)(0.45 * sigma_squared / (sigma_squared + 0.09));
			m->use_oren = 1;
		}
		/* shader slots (factory :692-752); a slot with a node makes its component present whatever the strength (config :52-85) */
		m->sh_diffuse = m->sh_mirror_color = m->sh_mirror = m->sh_transparency = m->sh_translucency = m->sh_sigma_oren = m->sh_diffuse_refl = m->sh_ior = -1;
		m->ior_base = d->ior;
		if(d->n_nodes > 0 && d->nodes)
		{
			m->n_nodes = d->n_nodes;
			m->nodes = (node_t *)calloc((size_t)d->n_nodes, sizeof(node_t));
			for(int k = 0; k < d->n_nodes; ++k) node_configure(&m->nodes[k], &d->nodes[k]);
			m->sh_diffuse = d->sh_diffuse; m->sh_mirror_color = d->sh_mirror_color; m->sh_mirror = d->sh_mirror; m->sh_transparency = d->sh_transparency;
			m->sh_translucency = d->sh_translucency; m->sh_sigma_oren = d->sh_sigma_oren; m->sh_diffuse_refl = d->sh_diffuse_refl; m->sh_ior = d->sh_ior;
		}
		/* config() :46-92 */
		float acc = 1.f;
		m->n_bsdf = 0;
		if(m->mirror_strength > 0.00001f || m->sh_mirror >= 0)
		{
			m->is_mirror = 1;
			if(m->sh_mirror >= 0) { /* the node's value is not known here: acc stays (:55) */ }
			else if(!m->has_fresnel) acc = 1.f - m->mirror_strength;
			m->flags |= BSDF_SPECULAR | BSDF_REFLECT;
			m->c_flags[m->n_bsdf] = BSDF_SPECULAR | BSDF_REFLECT; m->c_index[m->n_bsdf] = 0; ++m->n_bsdf;
		}
		if(m->transparency_strength * acc > 0.00001f || m->sh_transparency >= 0)
		{
			m->is_transparent = 1;
			if(m->sh_transparency < 0) acc *= 1.f - m->transparency_strength;
			m->flags |= BSDF_TRANSMIT | BSDF_FILTER;
			m->c_flags[m->n_bsdf] = BSDF_TRANSMIT | BSDF_FILTER; m->c_index[m->n_bsdf] = 1; ++m->n_bsdf;
		}
		if(m->translucency_strength * acc > 0.00001f || m->sh_translucency >= 0)
		{
			m->is_translucent = 1;
			if(m->sh_translucency < 0) acc *= 1.f - m->transparency_strength; /* sic: the reference multiplies by transparency here (:76) */
			m->flags |= BSDF_DIFFUSE | BSDF_TRANSMIT;
			m->c_flags[m->n_bsdf] = BSDF_DIFFUSE | BSDF_TRANSMIT; m->c_index[m->n_bsdf] = 2; ++m->n_bsdf;
		}
		if(m->diffuse_strength * acc > 0.00001f)
		{
			m->is_diffuse = 1;
			m->flags |= BSDF_DIFFUSE | BSDF_REFLECT;
			m->c_flags[m->n_bsdf] = BSDF_DIFFUSE | BSDF_REFLECT; m->c_index[m->n_bsdf] = 3; ++m->n_bsdf;
		}
	}
	else if(d->type == YOR_MAT_GLOSSY)
	{
		/* material_glossy.cc:32-50 */
		m->gloss_color = C(d->glossy_color[0], d->glossy_color[1], d->glossy_color[2]);
		m->diff_color = C(d->diffuse_color[0], d->diffuse_color[1], d->diffuse_color[2]);
		m->exponent = d->exponent; m->reflectivity = d->glossy_reflect; m->diffuse = d->glossy_diffuse_reflect;
		m->as_diffuse = d->as_diffuse;
		m->anisotropic = d->anisotropic; m->exp_u = d->exp_u; m->exp_v = d->exp_v;
		mat_copy_nodes(m, d);
		m->flags = BSDF_NONE;
		if(m->diffuse > 0) { m->flags = BSDF_DIFFUSE | BSDF_REFLECT; m->with_diffuse = 1; }
		m->flags |= m->as_diffuse ? (BSDF_DIFFUSE | BSDF_REFLECT) : (BSDF_GLOSSY | BSDF_REFLECT);
		if(d->oren_nayar)
		{	/* :66-72 */
			double sigma_2 = d->sigma * d->sigma;
			m->oren_a = (float)(1.0 - 0.5 * (sigma_2 / (sigma_2 + 0.33)));
			m->oren_b = (float)(0.45 * sigma_2 / (sigma_2 + 0.09));
			m->use_oren = 1;
		}
	}
	else if(d->type == YOR_MAT_COATED_GLOSSY)
	{	/* CoatedGlossyMaterial::factory + ctor, material_coated_glossy.cc:41-66, 464-560 (Blinn lobe, no nodes) */
		m->gloss_color = C(d->glossy_color[0], d->glossy_color[1], d->glossy_color[2]);
		m->diff_color = C(d->diffuse_color[0], d->diffuse_color[1], d->diffuse_color[2]);
		m->mirror_color = C(d->mirror_color[0], d->mirror_color[1], d->mirror_color[2]);
		m->mirror_strength = d->specular_reflect;
		m->ior = d->ior;
		m->exponent = d->exponent; m->reflectivity = d->glossy_reflect; m->diffuse = d->glossy_diffuse_reflect;
		m->as_diffuse = d->as_diffuse;
		m->anisotropic = d->anisotropic; m->exp_u = d->exp_u; m->exp_v = d->exp_v;
		mat_copy_nodes(m, d); m->ior_plain = d->ior;
		m->c_flags[0] = BSDF_SPECULAR | BSDF_REFLECT;
		m->c_flags[1] = m->as_diffuse ? (BSDF_DIFFUSE | BSDF_REFLECT) : (BSDF_GLOSSY | BSDF_REFLECT);
		if(m->diffuse > 0) { m->c_flags[2] = BSDF_DIFFUSE | BSDF_REFLECT; m->with_diffuse = 1; m->n_bsdf = 3; }
		else { m->c_flags[2] = BSDF_NONE; m->n_bsdf = 2; }
		m->flags = m->c_flags[0] | m->c_flags[1] | m->c_flags[2];
		if(d->oren_nayar)
		{	/* initOrenNayar :81-87 */
			double sigma_2 = d->sigma * d->sigma;
			m->oren_a = (float)(1.0 - 0.5 * (sigma_2 / (sigma_2 + 0.33)));
			m->oren_b = (float)(0.45 * sigma_2 / (sigma_2 + 0.09));
			m->use_oren = 1;
		}
	}
	else if(d->type == YOR_MAT_GLASS)
	{	/* GlassMaterial::factory + ctor, material_glass.cc:32-49, 340-388 (no dispersion, no absorption, no nodes) */
		m->ior = d->ior; m->transp_ior = d->ior; m->ior_plain = d->ior;
		mat_copy_nodes(m, d);
		const double filt = d->sigma;                                    /* transmit_filter, a double parameter */
		const float ff = (float)filt, fc = (float)(1.f - filt);          /* filt * filt_col + Rgb(1.f - filt) */
		m->filter_color = C(ff * d->color[0] + fc, ff * d->color[1] + fc, ff * d->color[2] + fc);
		m->spec_refl_color = C(d->mirror_color[0], d->mirror_color[1], d->mirror_color[2]);
		m->fake_shadow = d->fresnel_effect;
		m->flags = BSDF_SPECULAR | BSDF_REFLECT | BSDF_TRANSMIT;          /* BsdfAllSpecular */
		if(m->fake_shadow) m->flags |= BSDF_FILTER;
		m->tm_flags = m->fake_shadow ? (BSDF_FILTER | BSDF_TRANSMIT) : (BSDF_SPECULAR | BSDF_TRANSMIT);
		/* material_glass.cc:371-398: any channel of "absorption" below 1 -> bsdf_flags_ |= BsdfVolumetric and
		 * vol_i_ = BeerVolumeHandler(absorption, absorption_dist (default 1)) */
		if(d->has_absorption && (d->absorption[0] < 1.f || d->absorption[1] < 1.f || d->absorption[2] < 1.f))
		{
			m->flags |= BSDF_VOLUMETRIC;
			m->has_vol_i = 1;
			m->beer_sigma = beer_sigma(d->absorption, d->absorption_dist);
		}
	}
	else if(d->type == YOR_MAT_ROUGH_GLASS)
	{	/* RoughGlassMaterial::factory + ctor, material_rough_glass.cc:33-48, 322-400 (no dispersion, no nodes) */
		m->ior = d->ior; m->transp_ior = d->ior;
		m->sh_diffuse = m->sh_mirror_color = m->sh_mirror = m->sh_transparency = m->sh_translucency = m->sh_sigma_oren = m->sh_diffuse_refl = m->sh_ior = -1;
		m->sh_glossy = m->sh_glossy_reflect = m->sh_exponent = m->sh_filter_color = -1;
		const float filt = d->transmit_filter;                            /* a float parameter here (:324); glass reads a double */
		const float fc = 1.f - filt;                                      /* filt * filt_col + Rgb(1.f - filt) */
		m->filter_color = C(filt * d->color[0] + fc, filt * d->color[1] + fc, filt * d->color[2] + fc);
		m->spec_refl_color = C(d->mirror_color[0], d->mirror_color[1], d->mirror_color[2]);
		m->fake_shadow = d->fresnel_effect;
		const float alpha = fmaxf_(1e-4f, fminf_(d->rough_alpha * 0.5f, 1.f));      /* :362 */
		m->rg_a2 = alpha * alpha;
		m->flags = BSDF_GLOSSY | BSDF_REFLECT | BSDF_TRANSMIT;            /* BsdfAllGlossy */
		if(m->fake_shadow) m->flags |= BSDF_FILTER;
		if(d->has_absorption && (d->absorption[0] < 1.f || d->absorption[1] < 1.f || d->absorption[2] < 1.f))
		{	/* :370-398, as for glass */
			m->flags |= BSDF_VOLUMETRIC;
			m->has_vol_i = 1;
			m->beer_sigma = beer_sigma(d->absorption, d->absorption_dist);
		}
	}
	else if(d->type == YOR_MAT_MIRROR)
	{	/* MirrorMaterial, material_glass.h:74-79, material_glass.cc:486-493 */
		m->ref_col = cscale(C(d->color[0], d->color[1], d->color[2]), d->specular_reflect);
		m->flags = BSDF_SPECULAR;
	}
	else
	{
		/* material_simple.cc:36-39,63-73: col * (float)power */
		m->light_col = cscale(C(d->light_color[0], d->light_color[1], d->light_color[2]), d->light_power);
		m->double_sided = d->double_sided;
		m->flags = BSDF_EMIT;
	}
}

static inline v3 face_forward(v3 ng, v3 n, v3 i) { return (vdot(ng, i) < 0) ? vneg(n) : n; } /* material.h:33 */

/* getFresnel, material_shiny_diffuse.cc:119-147 */
static float sd_fresnel(const mat_t *m, v3 wo, v3 n)
{
	if(m->has_fresnel)
	{
		v3 N = (vdot(wo, n) < 0.f) ? vneg(n) : n;
		float c = vdot(wo, N);
		float g = m->ior_squared + c * c - 1.f;
		if(g < 0.f) g = 0.f;
		else g = yor_fsqrt(g);
		float aux = c * (g + c);
		return ((0.5f * (g - c) * (g - c)) / ((g + c) * (g + c))) *
		       (1.f + ((aux - 1) * (aux - 1)) / ((aux + 1) * (aux + 1)));
	}
	return 1.f;
}
/* accumulate__, :152-161 */
static void sd_accumulate(const float *component, float *accum, float kr)
{
	accum[0] = component[0] * kr;
	float acc = 1.f - accum[0];
	accum[1] = component[1] * acc;
	acc *= 1.f - component[1];
	accum[2] = component[2] * acc;
	acc *= 1.f - component[2];
	accum[3] = component[3] * acc;
}
/* orenNayar, :204-241 and material_glossy.cc:74-111 (identical bodies); a,b are float members */
static float oren_nayar(float oren_a, float oren_b, v3 wi, v3 wo, v3 n)
{
	float cos_ti = fmaxf_(-1.f, fminf_(1.f, vdot(n, wi)));
	float cos_to = fmaxf_(-1.f, fminf_(1.f, vdot(n, wo)));
	float maxcos_f = 0.f;
	if(cos_ti < 0.9999f && cos_to < 0.9999f)
	{
		v3 v_1 = vnormalize(vsub(wi, vmul(n, cos_ti)));
		v3 v_2 = vnormalize(vsub(wo, vmul(n, cos_to)));
		maxcos_f = fmaxf_(0.f, vdot(v_1, v_2));
	}
	float sin_alpha, tan_beta;
	if(cos_to >= cos_ti)
	{
		sin_alpha = yor_fsqrt(1.f - cos_ti * cos_ti);
		tan_beta = yor_fsqrt(1.f - cos_to * cos_to) / ((cos_to == 0.f) ? 1e-8f : cos_to);
	}
	else
	{
		sin_alpha = yor_fsqrt(1.f - cos_to * cos_to);
		tan_beta = yor_fsqrt(1.f - cos_ti * cos_ti) / ((cos_ti == 0.f) ? 1e-8f : cos_ti);
	}
	return fminf_(1.f, fmaxf_(0.f, (float)(oren_a + oren_b * maxcos_f * sin_alpha * tan_beta)));
}

/* the same with A and B computed from a texture's sigma, in double (material_shiny_diffuse.cc:230-235) */
static float oren_nayar_d(double oren_a, double oren_b, v3 wi, v3 wo, v3 n)
{
	float cos_ti = fmaxf_(-1.f, fminf_(1.f, vdot(n, wi)));
	float cos_to = fmaxf_(-1.f, fminf_(1.f, vdot(n, wo)));
	float maxcos_f = 0.f;
	if(cos_ti < 0.9999f && cos_to < 0.9999f)
	{
		v3 v_1 = vnormalize(vsub(wi, vmul(n, cos_ti)));
		v3 v_2 = vnormalize(vsub(wo, vmul(n, cos_to)));
		maxcos_f = fmaxf_(0.f, vdot(v_1, v_2));
	}
	float sin_alpha, tan_beta;
	if(cos_to >= cos_ti)
	{
		sin_alpha = yor_fsqrt(1.f - cos_ti * cos_ti);
		tan_beta = yor_fsqrt(1.f - cos_to * cos_to) / ((cos_to == 0.f) ? 1e-8f : cos_to);
	}
	else
	{
		sin_alpha = yor_fsqrt(1.f - cos_to * cos_to);
		tan_beta = yor_fsqrt(1.f - cos_ti * cos_ti) / ((cos_ti == 0.f) ? 1e-8f : cos_ti);
	}
	return fminf_(1.f, fmaxf_(0.f, (float)(oren_a + oren_b * (double)maxcos_f * (double)sin_alpha * (double)tan_beta)));
}
static float sd_oren(const mat_t *m, v3 wi, v3 wo, v3 n)
{
	return m->oren_tex ? oren_nayar_d(m->oren_ad, m->oren_bd, wi, wo, n) : oren_nayar(m->oren_a, m->oren_b, wi, wo, n);
}

/* refract__, vector.cc:86-108 */
static int refract_dir(v3 n, v3 wi, v3 *wo, float ior)
{
	v3 N = n, i;
	float eta = ior;
	i = vneg(wi);
	float cos_v_n = vdot(wi, n);
	if((cos_v_n) < 0)
	{
		N = vneg(n);
		cos_v_n = -cos_v_n;
	}
	else eta = (float)(1.0 / (double)ior);
	float k = 1 - eta * eta * (1 - cos_v_n * cos_v_n);
	if(k <= 0.f) return 0;
	*wo = vadd(vmul(i, eta), vmul(N, eta * cos_v_n - yor_fsqrt(k)));
	*wo = vnormalize(*wo);
	return 1;
}
/* fresnel__, vector.cc:110-142 (kr is formed in double) */
static void fresnel_dielectric(v3 i, v3 n, float ior, float *kr, float *kt)
{
	float eta = ior;
	v3 N = (vdot(i, n) < 0) ? vneg(n) : n;
	float c = vdot(i, N);
	float g = eta * eta + c * c - 1;
	if(g <= 0) g = 0;
	else g = yor_fsqrt(g);
	float aux = c * (g + c);
	*kr = (float)(((0.5 * (double)(g - c) * (double)(g - c)) / (double)((g + c) * (g + c))) *
	              (double)(1 + ((aux - 1) * (aux - 1)) / ((aux + 1) * (aux + 1))));
	if(*kr < 1.0) *kt = 1 - *kr;
	else *kt = 0;
}
/* the shading normal the glass uses, material_glass.cc:77-80, 263-271 */
static v3 glass_normal(const sp_t *sp, v3 wo)
{
	int outside = vdot(sp->ng, wo) > 0;
	float cos_wo_n = vdot(sp->n, wo);
	if(outside ? (cos_wo_n >= 0) : (cos_wo_n <= 0)) return sp->n;
	float f = (float)(1.00001 * (double)cos_wo_n);
	return vnormalize(vsub(sp->n, vmul(wo, f)));
}

/* A textured material at one surface point: the node list is evaluated once (ShinyDiffuseMaterial::initBsdf :163-183 —
 * every supported node is view independent) and what the material's functions read through a shader slot is written
 * into a copy of the material record: `slot ? slot->getColor / getScalar(stack) : member` becomes the member of the copy.
 * Untextured materials are returned as they are. */
#define YOR_MAX_NODES 128
static const mat_t *mat_resolve(const yor_scene *s, const sp_t *sp, mat_t *out)
{
	const mat_t *m = &s->mats[sp->mat];
	if(m->n_nodes <= 0 || (m->type != YOR_MAT_SHINYDIFFUSE && m->type != YOR_MAT_GLOSSY && m->type != YOR_MAT_COATED_GLOSSY && m->type != YOR_MAT_GLASS)) return m;
	node_result_t stack[YOR_MAX_NODES];
	nodes_eval(m->nodes, m->n_nodes < YOR_MAX_NODES ? m->n_nodes : YOR_MAX_NODES, s->tex, s->n_tex, &s->cam, sp, stack);
	*out = *m;
	if(m->type == YOR_MAT_GLASS)
	{	/* material_glass.cc:87-95,109,121,143-190,223-224,262-300 */
		if(m->sh_mirror_color >= 0) out->spec_refl_color = C(stack[m->sh_mirror_color].col.r, stack[m->sh_mirror_color].col.g, stack[m->sh_mirror_color].col.b);
		if(m->sh_filter_color >= 0) out->filter_color = C(stack[m->sh_filter_color].col.r, stack[m->sh_filter_color].col.g, stack[m->sh_filter_color].col.b);
		if(m->sh_ior >= 0) { out->ior = m->ior_plain + stack[m->sh_ior].f; out->transp_ior = stack[m->sh_ior].f; }      /* :223 sic: not added there */
		return out;
	}
	if(m->type != YOR_MAT_SHINYDIFFUSE)
	{	/* glossy / coated glossy: every use of a shader is `shader ? shader->get…(stack) : member` (material_glossy.cc:62,144-160,
		 * material_coated_glossy.cc:78,147-174,253-256,448-451), so the members of a per-hit copy carry them */
		if(m->sh_diffuse >= 0) out->diff_color = C(stack[m->sh_diffuse].col.r, stack[m->sh_diffuse].col.g, stack[m->sh_diffuse].col.b);
		if(m->sh_glossy >= 0) out->gloss_color = C(stack[m->sh_glossy].col.r, stack[m->sh_glossy].col.g, stack[m->sh_glossy].col.b);
		if(m->sh_glossy_reflect >= 0) out->reflectivity = stack[m->sh_glossy_reflect].f;
		if(m->sh_exponent >= 0) out->exponent = stack[m->sh_exponent].f;
		if(m->sh_sigma_oren >= 0)
		{
			double sigma = (double)stack[m->sh_sigma_oren].f, s2 = sigma * sigma;
			out->oren_tex = 1; out->oren_ad = 1.0 - 0.5 * (s2 / (s2 + 0.33)); out->oren_bd = 0.45 * s2 / (s2 + 0.09);
		}
		if(m->sh_diffuse_refl >= 0) { out->has_diffuse_refl = 1; out->diffuse_refl = stack[m->sh_diffuse_refl].f; }
		if(m->sh_mirror_color >= 0) out->mirror_color = C(stack[m->sh_mirror_color].col.r, stack[m->sh_mirror_color].col.g, stack[m->sh_mirror_color].col.b);
		if(m->sh_mirror >= 0) out->mirror_strength = stack[m->sh_mirror].f;
		if(m->sh_ior >= 0) out->ior = m->ior_plain + stack[m->sh_ior].f;
		return out;
	}
	if(m->sh_diffuse >= 0)
	{
		out->diffuse_color = C(stack[m->sh_diffuse].col.r, stack[m->sh_diffuse].col.g, stack[m->sh_diffuse].col.b);
		out->emit_color = cscale(out->diffuse_color, m->emit_strength);             /* emit :300 */
	}
	if(m->sh_mirror_color >= 0) out->mirror_color = C(stack[m->sh_mirror_color].col.r, stack[m->sh_mirror_color].col.g, stack[m->sh_mirror_color].col.b);
	if(m->sh_mirror >= 0) out->mirror_strength = stack[m->sh_mirror].f;             /* getComponents :98-117 */
	if(m->sh_transparency >= 0) out->transparency_strength = stack[m->sh_transparency].f;
	if(m->sh_translucency >= 0) out->translucency_strength = stack[m->sh_translucency].f;
	if(m->sh_sigma_oren >= 0)
	{	/* orenNayar :230-235 */
		double sigma = (double)stack[m->sh_sigma_oren].f, s2 = sigma * sigma;
		out->oren_tex = 1; out->oren_ad = 1.0 - 0.5 * (s2 / (s2 + 0.33)); out->oren_bd = 0.45 * s2 / (s2 + 0.09);
	}
	if(m->sh_diffuse_refl >= 0) { out->has_diffuse_refl = 1; out->diffuse_refl = stack[m->sh_diffuse_refl].f; }
	if(m->sh_ior >= 0) { float cur = m->ior_base + stack[m->sh_ior].f; out->ior_squared = cur * cur; }     /* :258-262 */
	return out;
}

/* initBsdf's first act on a material with a bump shader (material_shiny_diffuse.cc:171-175, material_glossy.cc:56, ...): evalBump moves
 * the surface point's shading frame; everything after it at this vertex — the colour nodes, the light estimate, the samplers — sees that */
static const mat_t *mat_at(const yor_scene *s, sp_t *sp, mat_t *out)
{
	const mat_t *m = &s->mats[sp->mat];
	if(m->n_bump > 0)
	{
		node_result_t stack[YOR_MAX_NODES];
		const int nb = m->n_bump < YOR_MAX_NODES ? m->n_bump : YOR_MAX_NODES;
		nodes_eval_derivative(m->bump_nodes, nb, s->tex, s->n_tex, &s->cam, sp, stack);
		apply_bump(sp, stack[m->sh_bump].col.r, stack[m->sh_bump].col.g);
	}
	return mat_resolve(s, sp, out);
}

static void mat_init_bsdf(const mat_t *m, bsdf_dat *dat, unsigned *bsdf_types)
{
	memset(dat, 0, sizeof *dat);
	*bsdf_types = m->flags;
	if(m->type == YOR_MAT_SHINYDIFFUSE)
	{	/* initBsdf :163-183 + getComponents :98-117 */
		if(m->is_mirror) dat->component[0] = m->mirror_strength;
		if(m->is_transparent) dat->component[1] = m->transparency_strength;
		if(m->is_translucent) dat->component[2] = m->translucency_strength;
		if(m->is_diffuse) dat->component[3] = m->diffuse_strength;
	}
	else if(m->type == YOR_MAT_GLOSSY || m->type == YOR_MAT_COATED_GLOSSY)
	{	/* material_glossy.cc:51-64, material_coated_glossy.cc:68-81 */
		dat->m_diffuse = m->diffuse;
		dat->m_glossy = m->reflectivity;
		dat->p_diffuse = fminf_(0.6f, 1.f - (dat->m_glossy / (dat->m_glossy + (1.f - dat->m_glossy) * dat->m_diffuse)));
	}
}

/* microfacet helpers, material_utils_microfacet.h */
static inline float blinn_d(float cos_h, float e) { return (e + 1.f) * yor_fpow(cos_h, e); }             /* :89-92 */
static inline double pdf_divisor(float c) { return (double)8.f * Y_M_PI * (double)(c * 0.99f + 0.04f); } /* :35 */
static inline float blinn_pdf(float costheta, float cos_w_h, float e) { return (float)((double)blinn_d(costheta, e) / pdf_divisor(cos_w_h)); } /* :94-97 */
static inline double as_divisor(float cos1, float cos_i, float cos_o) { return (double)8.f * Y_M_PI * (double)((cos1 * fmaxf_(cos_i, cos_o)) * 0.99f + 0.04f); } /* :36 */
static inline float schlick_fresnel(float costheta, float r) /* :188-193 */
{
	float c_1 = (1.f - costheta);
	float c_2 = c_1 * c_1;
	return r + ((1.f - r) * c_1 * c_2 * c_2);
}
static inline rgb diffuse_reflect(float wi_n, float wo_n, float glossy, float diffuse, rgb diff_base) /* :195-206 */
{
	float temp = 0.f;
	float f_wi = (1.f - (0.5f * wi_n));
	temp = f_wi * f_wi;
	f_wi = temp * temp * f_wi;
	float f_wo = (1.f - (0.5f * wo_n));
	temp = f_wo * f_wo;
	f_wo = temp * temp * f_wo;
	/* DIFFUSE_RATIO is a double literal: the scalar chain runs in double, then narrows for Rgb*float */
	double k = 0.387507688 * (double)diffuse * (double)(1.f - glossy) * (double)(1.f - f_wi) * (double)(1.f - f_wo);
	return cscale(diff_base, (float)k);
}
static inline v3 blinn_sample(float s_1, float s_2, float exponent) /* :99-106 */
{
	float cos_theta = yor_fpow(1.f - s_2, 1.f / (exponent + 1.f));
	float sin_theta = yor_fsqrt(1.f - cos_theta * cos_theta);
	float phi = (float)((double)s_1 * Y_M_2PI);
	return V(sin_theta * yor_fcos(phi), sin_theta * yor_fsin(phi), cos_theta);
}

/* the anisotropic Ashikhmin-Shirley lobe, material_utils_microfacet.h:38-87 */
static inline v3 sample_quadrant_aniso(float s_1, float s_2, float e_u, float e_v) /* :38-51 */
{
	float phi = (float)atan((double)(yor_fsqrt((e_u + 1.f) / (e_v + 1.f)) * tanf((float)(Y_M_PI_2 * (double)s_1))));
	float cos_phi = yor_fcos(phi);
	float sin_phi = yor_fsin(phi);
	float cos_theta, sin_theta;
	float cos_phi_2 = cos_phi * cos_phi;
	float sin_phi_2 = 1.f - cos_phi_2;
	cos_theta = yor_fpow(1.f - s_2, 1.f / (e_u * cos_phi_2 + e_v * sin_phi_2 + 1.f));
	sin_theta = yor_fsqrt(1.f - cos_theta * cos_theta);
	return V(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
static inline float as_aniso_d(v3 h, float e_u, float e_v) /* :53-58 */
{
	if(h.z <= 0.f) return 0.f;
	float exponent = (e_u * h.x * h.x + e_v * h.y * h.y) / (1.00001f - h.z * h.z);
	return yor_fsqrt((e_u + 1.f) * (e_v + 1.f)) * yor_fpow(fmaxf_(0.f, h.z), exponent);
}
static inline float as_aniso_pdf(v3 h, float cos_w_h, float e_u, float e_v) { return (float)((double)as_aniso_d(h, e_u, e_v) / pdf_divisor(cos_w_h)); } /* :60-63 */
static inline v3 as_aniso_sample(float s_1, float s_2, float e_u, float e_v) /* :65-87 */
{
	v3 h;
	if(s_1 < 0.25f) h = sample_quadrant_aniso(4.f * s_1, s_2, e_u, e_v);
	else if(s_1 < 0.5f) { h = sample_quadrant_aniso(1.f - 4.f * (0.5f - s_1), s_2, e_u, e_v); h.x = -h.x; }
	else if(s_1 < 0.75f) { h = sample_quadrant_aniso(4.f * (s_1 - 0.5f), s_2, e_u, e_v); h.x = -h.x; h.y = -h.y; }
	else { h = sample_quadrant_aniso(1.f - 4.f * (1.f - s_1), s_2, e_u, e_v); h.y = -h.y; }
	return h;
}
/* the material's glossy lobe: Blinn on cos(n, h), or the anisotropic lobe on h in the shading frame (hs) */
static inline float lobe_d(const mat_t *m, v3 hs, float cos_n_h) { return m->anisotropic ? as_aniso_d(hs, m->exp_u, m->exp_v) : blinn_d(cos_n_h, m->exponent); }
static inline float lobe_pdf(const mat_t *m, v3 hs, float cos_n_h, float cos_w_h) { return m->anisotropic ? as_aniso_pdf(hs, cos_w_h, m->exp_u, m->exp_v) : blinn_pdf(cos_n_h, cos_w_h, m->exponent); }
static inline v3 lobe_sample(const mat_t *m, float s_1, float s_2) { return m->anisotropic ? as_aniso_sample(s_1, s_2, m->exp_u, m->exp_v) : blinn_sample(s_1, s_2, m->exponent); }
static inline v3 local_h(const sp_t *sp, v3 h, float cos_n_h) { return V(vdot(h, sp->nu), vdot(h, sp->nv), cos_n_h); }

static rgb mat_eval(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo, v3 wl, unsigned bsdfs)
{
	if(m->type == YOR_MAT_SHINYDIFFUSE)
	{	/* material_shiny_diffuse.cc:244-293 */
		float cos_ng_wo = vdot(sp->ng, wo);
		float cos_ng_wl = vdot(sp->ng, wl);
		v3 n = face_forward(sp->ng, sp->n, wo);
		if(!(bsdfs & m->flags & BSDF_DIFFUSE)) return C(0, 0, 0);
		float kr = sd_fresnel(m, wo, n);
		float m_t = (1.f - kr * dat->component[0]) * (1.f - dat->component[1]);
		int transmit = (cos_ng_wo * cos_ng_wl) < 0.f;
		if(transmit)
		{
			if(m->is_translucent) return cscale(m->diffuse_color, dat->component[2] * m_t);
		}
		if(vdot(n, wl) < 0.0 && !m->flat) return C(0, 0, 0);
		float m_d = m_t * (1.f - dat->component[2]) * dat->component[3];
		if(m->use_oren) m_d *= sd_oren(m, wo, wl, n);
		if(m->has_diffuse_refl) m_d *= m->diffuse_refl;                      /* :285 */
		return cscale(m->diffuse_color, m_d);
	}
	else if(m->type == YOR_MAT_GLOSSY)
	{	/* material_glossy.cc:113-173 */
		if(!(bsdfs & BSDF_DIFFUSE) || (vdot(sp->ng, wl) * vdot(sp->ng, wo)) < 0.f) return C(0, 0, 0);
		rgb col = C(0, 0, 0);
		int diffuse_flag = (bsdfs & BSDF_DIFFUSE) != 0;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float wi_n = fabsf(vdot(wl, n));
		float wo_n = fabsf(vdot(wo, n));
		if((m->as_diffuse && diffuse_flag) || (!m->as_diffuse && (bsdfs & BSDF_GLOSSY)))
		{
			v3 h = vnormalize(vadd(wo, wl));
			float cos_wi_h = fmaxf_(0.f, vdot(wl, h));
			float cos_n_h = vdot(h, n);
			float glossy = (float)((double)(lobe_d(m, local_h(sp, h, cos_n_h), cos_n_h) * schlick_fresnel(cos_wi_h, dat->m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
			col = cscale(m->gloss_color, glossy);
		}
		if(m->with_diffuse && diffuse_flag)
		{
			rgb add_col = cscale(m->diff_color, dat->m_diffuse * (1.f - dat->m_glossy));
			if(m->has_diffuse_refl) add_col = cscale(add_col, m->diffuse_refl);      /* diffuse_reflection_shader_ */
			if(m->use_oren) add_col = cscale(add_col, sd_oren(m, wl, wo, n));
			col = cadd(col, add_col);
		}
		return col;
	}
	if(m->type == YOR_MAT_COATED_GLOSSY)
	{	/* material_coated_glossy.cc:130-186 */
		rgb col = C(0, 0, 0);
		int diffuse_flag = (bsdfs & BSDF_DIFFUSE) != 0;
		if(!diffuse_flag || (vdot(sp->ng, wl) * vdot(sp->ng, wo)) < 0.f) return col;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr, kt;
		float wi_n = fabsf(vdot(wl, n));
		float wo_n = fabsf(vdot(wo, n));
		fresnel_dielectric(wo, n, m->ior, &kr, &kt);
		if((m->as_diffuse && diffuse_flag) || (!m->as_diffuse && (bsdfs & BSDF_GLOSSY)))
		{
			v3 h = vnormalize(vadd(wo, wl));
			float cos_wi_h = vdot(wl, h);
			float cos_n_h = vdot(h, n);
			float glossy = (float)((double)(kt * lobe_d(m, local_h(sp, h, cos_n_h), cos_n_h) * schlick_fresnel(cos_wi_h, dat->m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
			col = cscale(m->gloss_color, glossy);
		}
		if(m->with_diffuse && diffuse_flag)
		{
			rgb add_col = cscale(cscale(m->diff_color, dat->m_diffuse * (1.f - dat->m_glossy)), kt);
			if(m->has_diffuse_refl) add_col = cscale(add_col, m->diffuse_refl);      /* diffuse_reflection_shader_ */
			if(m->use_oren) add_col = cscale(add_col, sd_oren(m, wl, wo, n));
			col = cadd(col, add_col);
		}
		return col;
	}
	return C(0, 0, 0); /* LightMaterial::eval, material_simple.h */
}

static float mat_pdf(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo, v3 wi, unsigned bsdfs)
{
	if(m->type == YOR_MAT_SHINYDIFFUSE)
	{	/* material_shiny_diffuse.cc:410-460 */
		if(!(bsdfs & BSDF_DIFFUSE)) return 0.f;
		float pdf = 0.f, accum_c[4];
		float cos_ng_wo = vdot(sp->ng, wo), cos_ng_wi;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr = sd_fresnel(m, wo, n);
		sd_accumulate(dat->component, accum_c, kr);
		float sum = 0.f, width;
		int n_match = 0;
		for(int i = 0; i < m->n_bsdf; ++i)
		{
			if((bsdfs & m->c_flags[i]))
			{
				width = accum_c[m->c_index[i]];
				sum += width;
				if(m->c_flags[i] == (BSDF_DIFFUSE | BSDF_TRANSMIT))
				{
					cos_ng_wi = vdot(sp->ng, wi);
					if(cos_ng_wo * cos_ng_wi < 0) pdf += fabsf(vdot(wi, n)) * width;
				}
				else if(m->c_flags[i] == (BSDF_DIFFUSE | BSDF_REFLECT))
				{
					pdf += fabsf(vdot(wi, n)) * width;
				}
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) return 0.f;
		return pdf / sum;
	}
	else if(m->type == YOR_MAT_COATED_GLOSSY)
	{	/* material_coated_glossy.cc:376-424 */
		if((vdot(sp->ng, wo) * vdot(sp->ng, wi)) < 0.f) return 0.f;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float pdf = 0.f, kr, kt;
		fresnel_dielectric(wo, n, m->ior, &kr, &kt);
		float accum_c[3], sum = 0.f, width;
		accum_c[0] = kr;
		accum_c[1] = kt * (1.f - dat->p_diffuse);
		accum_c[2] = kt * (dat->p_diffuse);
		int n_match = 0;
		for(int i = 0; i < m->n_bsdf; ++i)
		{
			if((bsdfs & m->c_flags[i]) == m->c_flags[i])
			{
				width = accum_c[i];
				sum += width;
				if(i == 1)
				{
					v3 h = vnormalize(vadd(wi, wo));
					float cos_wo_h = vdot(wo, h);
					float cos_n_h = vdot(n, h);
					pdf += lobe_pdf(m, local_h(sp, h, cos_n_h), cos_n_h, cos_wo_h) * width;
				}
				else if(i == 2) pdf += fabsf(vdot(wi, n)) * width;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) return 0.f;
		return pdf / sum;
	}
	else if(m->type == YOR_MAT_GLOSSY)
	{	/* material_glossy.cc:359-405 */
		if(vdot(sp->ng, wo) * vdot(sp->ng, wi) < 0.f) return 0.f;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float pdf = 0.f;
		float cur_p_diffuse = dat->p_diffuse;
		int use_glossy = m->as_diffuse ? (bsdfs & BSDF_DIFFUSE) != 0 : (bsdfs & BSDF_GLOSSY) != 0;
		int use_diffuse = m->with_diffuse && (bsdfs & BSDF_DIFFUSE);
		if(use_diffuse)
		{
			pdf = fabsf(vdot(wi, n));
			if(use_glossy)
			{
				v3 h = vnormalize(vadd(wi, wo));
				float cos_wo_h = vdot(wo, h);
				float cos_n_h = vdot(n, h);
				pdf = pdf * cur_p_diffuse + lobe_pdf(m, local_h(sp, h, cos_n_h), cos_n_h, cos_wo_h) * (1.f - cur_p_diffuse);
			}
			return pdf;
		}
		if(use_glossy)
		{
			v3 h = vnormalize(vadd(wi, wo));
			float cos_wo_h = vdot(wo, h);
			float cos_n_h = vdot(n, h);
			pdf = lobe_pdf(m, local_h(sp, h, cos_n_h), cos_n_h, cos_wo_h);
		}
		return pdf;
	}
	return 0.f;
}

/* getAlpha, material_shiny_diffuse.cc:568-597 */
static float sd_alpha(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo)
{
	if(m->is_transparent)
	{
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr = sd_fresnel(m, wo, n);
		float refl = (1.f - dat->component[0] * kr) * dat->component[1];
		return 1.f - refl;
	}
	return 1.f;
}

/* ShinyDiffuseMaterial::getSpecular, material_shiny_diffuse.cc:474-528 (no shader nodes, no wireframe);
 * every other material of this path keeps Material::getSpecular's default: neither */
/* Material::getTransparency: ShinyDiffuse (material_shiny_diffuse.cc:530-566), Glass (material_glass.cc:217-228);
 * Material's default is black */
static rgb mat_transparency(const mat_t *m, const sp_t *sp, v3 wo)
{
	if(m->type == YOR_MAT_SHINYDIFFUSE)
	{
		if(!m->is_transparent) return C(0, 0, 0);
		float accum = 1.f;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr = sd_fresnel(m, wo, n);
		if(m->is_mirror) accum = 1.f - kr * m->mirror_strength;
		accum *= m->transparency_strength * accum;       /* sic, :557 */
		float f = m->transmit_filter;
		rgb tcol = cadd(cscale(m->diffuse_color, f), C(1.f - f, 1.f - f, 1.f - f));
		return cscale(tcol, accum);
	}
	if(m->type == YOR_MAT_GLASS || m->type == YOR_MAT_ROUGH_GLASS)      /* material_glass.cc:217-228, material_rough_glass.cc:288-299 */
	{
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr, kt;
		fresnel_dielectric(wo, n, m->transp_ior, &kr, &kt);
		return cscale(m->filter_color, kt);
	}
	return C(0, 0, 0);
}
/* Material::isTransparent: ShinyDiffuse m_is_transparent_ (material_shiny_diffuse.h:53), Glass fake_shadow_ (material_glass.h) */
static int mat_is_transparent(const mat_t *m)
{
	if(m->type == YOR_MAT_ROUGH_GLASS) return m->fake_shadow;      /* material_rough_glass.h:38 */
	return (m->type == YOR_MAT_SHINYDIFFUSE && m->is_transparent) || (m->type == YOR_MAT_GLASS && m->fake_shadow);
}
static int mat_is_transparent_idx(const yor_scene *s, int mat) { return mat_is_transparent(&s->mats[mat]); }
static rgb mat_transparency_at(const yor_scene *s, int ti, v3 hit, float bu, float bv, v3 dir)
{
	sp_t sp;
	get_surface(s, ti, hit, bu, bv, &sp);
	mat_t mloc;
	return mat_transparency(mat_resolve(s, &sp, &mloc), &sp, dir);
}

/* GlassMaterial::getTransparency / getAlpha, material_glass.cc:217-240 */
static float glass_alpha(const mat_t *m, const sp_t *sp, v3 wo)
{
	v3 n = face_forward(sp->ng, sp->n, wo);
	float kr, kt;
	fresnel_dielectric(wo, n, m->ior, &kr, &kt);
	rgb t = cscale(m->filter_color, kt);
	float alpha = (float)(1.0 - (double)((t.r + t.g + t.b) * 0.333333f));
	if(alpha < 0.0f) alpha = 0.0f;
	return alpha;
}
/* Material::getAlpha of the materials on this path */
static float mat_alpha(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo)
{
	if(m->type == YOR_MAT_SHINYDIFFUSE) return sd_alpha(m, dat, sp, wo);
	if(m->type == YOR_MAT_GLASS) return glass_alpha(m, sp, wo);
	if(m->type == YOR_MAT_ROUGH_GLASS)
	{	/* material_rough_glass.cc:301-310: max(0, min(1, 1 - getTransparency().energy())) */
		const rgb t = mat_transparency(m, sp, wo);
		return fmaxf_(0.f, fminf_(1.f, 1.f - (t.r + t.g + t.b) * 0.333333f));
	}
	return 1.f;
}
/* raylevel: RenderState::raylevel_ as getSpecular sees it (recursiveRaytrace has already incremented it) */
static void mat_get_specular(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo, int raylevel, int *do_reflect, int *do_refract, v3 wi[2], rgb col[2])
{
	*do_reflect = 0; *do_refract = 0;
	if(m->type == YOR_MAT_GLASS)
	{	/* GlassMaterial::getSpecular, material_glass.cc:242-340 (no dispersion) */
		int outside = vdot(sp->ng, wo) > 0;
		v3 n = glass_normal(sp, wo), refdir;
		const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);
		const v3 refl = V(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
		if(refract_dir(n, wo, &refdir, m->ior))
		{
			float kr, kt;
			fresnel_dielectric(wo, n, m->ior, &kr, &kt);
			col[1] = cscale(m->filter_color, kt); wi[1] = refdir; *do_refract = 1;
			if(outside || raylevel < 3) { wi[0] = refl; col[0] = cscale(m->spec_refl_color, kr); *do_reflect = 1; }
		}
		else { col[0] = m->spec_refl_color; wi[0] = refl; *do_reflect = 1; }
		return;
	}
	if(m->type == YOR_MAT_COATED_GLOSSY)
	{	/* CoatedGlossyMaterial::getSpecular, material_coated_glossy.cc:426-462 */
		int outside = vdot(sp->ng, wo) >= 0;
		float cos_wo_n = vdot(sp->n, wo);
		v3 n, ng = outside ? sp->ng : vneg(sp->ng);
		if(outside ? (cos_wo_n >= 0) : (cos_wo_n <= 0)) n = sp->n;
		else { float f = (float)(1.00001 * (double)cos_wo_n); n = vnormalize(vsub(sp->n, vmul(wo, f))); }
		float kr, kt;
		fresnel_dielectric(wo, n, m->ior, &kr, &kt);
		if(raylevel > 5) return;
		const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);
		v3 r = V(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
		col[0] = cscale(cscale(m->mirror_color, kr), m->mirror_strength);
		float cos_wi_ng = vdot(r, ng);
		if((double)cos_wi_ng < 0.01)
		{
			float k = (float)(0.01 - (double)cos_wi_ng);
			r = vnormalize(vadd(r, vmul(ng, k)));
		}
		wi[0] = r;
		*do_reflect = 1;
		return;
	}
	if(m->type == YOR_MAT_MIRROR)
	{	/* MirrorMaterial::getSpecular, material_glass.cc:475-484 */
		col[0] = m->ref_col;
		wi[0] = reflect_dir(face_forward(sp->ng, sp->n, wo), wo);
		*do_reflect = 1;
		return;
	}
	if(m->type != YOR_MAT_SHINYDIFFUSE) return;
	const int backface = vdot(wo, sp->ng) < 0.f;
	const v3 n = backface ? vneg(sp->n) : sp->n;
	const v3 ng = backface ? vneg(sp->ng) : sp->ng;
	float kr = sd_fresnel(m, wo, n);
	if(m->is_transparent)
	{
		*do_refract = 1;
		wi[1] = vneg(wo);
		float f = m->transmit_filter;
		rgb tcol = cadd(cscale(m->diffuse_color, f), C(1.f - f, 1.f - f, 1.f - f));
		col[1] = cscale(tcol, (1.f - dat->component[0] * kr) * dat->component[1]);
	}
	if(m->is_mirror)
	{
		*do_reflect = 1;
		/* Vec3::reflect, vector.h:291-298 */
		const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);
		v3 r = V(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
		float cos_wi_ng = vdot(r, ng);
		if((double)cos_wi_ng < 0.01)
		{
			float k = (float)(0.01 - (double)cos_wi_ng);
			r = vadd(r, vmul(ng, k));
			r = vnormalize(r);
		}
		wi[0] = r;
		col[0] = cscale(m->mirror_color, dat->component[0] * kr);
	}
}

/* GGX microfacet helpers, material_utils_microfacet.h:108-185 */
static inline v3 ggx_sample(float alpha_2, float s_1, float s_2)                       /* :111-121 */
{
	const float tan_theta_2 = alpha_2 * (s_1 / (1.00001f - s_1));
	const float cos_theta = 1.f / yor_fsqrt(1.f + tan_theta_2);
	const float sin_theta = yor_fsqrt(1.00001f - (cos_theta * cos_theta));
	const float phi = (float)(Y_M_2PI * (double)s_2);
	return V(sin_theta * yor_fcos(phi), sin_theta * yor_fsin(phi), cos_theta);
}
static inline float ggx_d(float alpha_2, float cos_theta_2, float tan_theta_2)        /* :123-129: M_PI makes the divisor a double product */
{
	const float cos_theta_4 = cos_theta_2 * cos_theta_2;
	const float a_tan = alpha_2 + tan_theta_2;
	const float div = (float)(((Y_M_PI * (double)cos_theta_4) * (double)a_tan) * (double)a_tan);
	return alpha_2 / div;
}
static inline float ggx_g(float alpha_2, float wo_n, float wi_n)                       /* :131-143 */
{
	const float wo_n_2 = wo_n * wo_n, wi_n_2 = wi_n * wi_n;
	const float sqr_term_1 = yor_fsqrt(1.f + alpha_2 * ((1.f - wo_n_2) / wo_n_2));
	const float sqr_term_2 = yor_fsqrt(1.f + alpha_2 * ((1.f - wi_n_2) / wi_n_2));
	const float g_1_wo = 2.f / (1.f + (sqr_term_1));
	const float g_1_wi = 2.f / (1.f + (sqr_term_2));
	return g_1_wo * g_1_wi;
}
static inline float ggx_pdf(float d, float cos_theta, float jacobian) { return d * cos_theta * jacobian; }      /* :145-148 */
static inline float microfacet_fresnel(float wo_h, float ior)                          /* :150-162 */
{
	const float c = fabsf(wo_h);
	float g = ior * ior - 1 + c * c;
	if(g > 0)
	{
		g = yor_fsqrt(g);
		const float a = (g - c) / (g + c);
		const float b = (c * (g + c) - 1) / (c * (g - c) + 1);
		return 0.5f * a * a * (1 + b * b);
	}
	return 1.0f;
}
static inline int refract_microfacet(float eta, v3 wo, v3 *wi, v3 h, float wo_h, float *kr, float *kt)      /* :164-179 */
{
	*wi = V(0, 0, 0);
	const float c = vdot(vneg(wo), h);
	const float sign = (c > 0.f) ? 1 : -1;
	const float t_1 = 1 - (eta * eta * (1 - c * c));
	if(t_1 < 0.f) return 0;
	*wi = vadd(vmul(wo, eta), vmul(h, eta * c - sign * yor_fsqrt(t_1)));
	*wi = vneg(*wi);
	*kr = 0.f; *kt = 0.f;
	*kr = microfacet_fresnel(wo_h, 1.f / eta);
	if(*kr == 1.f) return 0;
	*kt = 1 - *kr;
	return 1;
}
static inline v3 reflect_microfacet(v3 wo, v3 h)                                       /* :181-185 */
{
	const v3 wi = vadd(wo, vmul(h, 2.f * vdot(h, vneg(wo))));
	return vneg(wi);
}
static inline v3 vreflect(v3 v, v3 n)                                                  /* Vec3::reflect, vector.h:291-298 */
{
	const float vn = 2.0f * (v.x * n.x + v.y * n.y + v.z * n.z);
	return V(vn * n.x - v.x, vn * n.y - v.y, vn * n.z - v.z);
}
/* RoughGlassMaterial::sample, both forms (material_rough_glass.cc:62-163 one direction, :165-286 two): the half vector of the GGX
 * lobe, refraction and reflection about it.  two == 0: the lobe is picked by s_1 against kt, (wi, w) out.  two != 0: both
 * directions — the reference writes the TRANSMITTED one to dir[0] / w[0] with the returned colour and the REFLECTED one to dir[1] /
 * w[1] with tcol, which recursiveRaytrace then reads the other way round (:925-958): restated as it is. */
static rgb rough_glass_sample(const mat_t *m, const sp_t *sp, v3 wo, sample_t *s, int two, v3 dir[2], rgb *tcol, float w[2])
{
	const v3 n = face_forward(sp->ng, sp->n, wo);
	const int outside = vdot(sp->ng, wo) > 0.f;
	s->pdf = 1.f;
	const float alpha_2 = m->rg_a2;
	v3 h = ggx_sample(alpha_2, s->s_1, s->s_2);
	h = vadd(vadd(vmul(sp->nu, h.x), vmul(sp->nv, h.y)), vmul(n, h.z));
	h = vnormalize(h);
	const float cur_ior = m->ior;
	float glossy, glossy_d = 0.f, glossy_g = 0.f, wi_n, wi_h, jacobian = 0.f;
	const float cos_theta = vdot(h, n);
	const float cos_theta_2 = cos_theta * cos_theta;
	const float tan_theta_2 = (1.f - cos_theta_2) / fmaxf_(1.0e-8f, cos_theta_2);
	if(cos_theta > 0.f) glossy_d = ggx_d(alpha_2, cos_theta_2, tan_theta_2);
	const float wo_h = vdot(wo, h), wo_n = vdot(wo, n);
	float kr, kt;
	rgb ret = C(0, 0, 0);
	v3 wi;
	if(two) s->sampled_flags = 0;
	if(refract_microfacet(outside ? 1.f / cur_ior : cur_ior, wo, &wi, h, wo_h, &kr, &kt))
	{
		const int take_t = two ? (s->flags & BSDF_TRANSMIT) != 0 : (s->s_1 < kt && (s->flags & BSDF_TRANSMIT));
		if(take_t)
		{
			wi_n = vdot(wi, n); wi_h = vdot(wi, h);
			if((wi_h * wi_n) > 0.f && (wo_h * wo_n) > 0.f) glossy_g = ggx_g(alpha_2, wi_n, wo_n);
			float ior_wi = 1.f, ior_wo = 1.f;
			if(outside) ior_wi = cur_ior; else ior_wo = cur_ior;
			const float ht = ior_wo * wo_h + ior_wi * wi_h;
			jacobian = (ior_wi * ior_wi) / fmaxf_(1.0e-8f, ht * ht);
			glossy = fabsf((wo_h * wi_h) / (wi_n * wo_n)) * kt * glossy_g * glossy_d * jacobian;
			s->pdf = ggx_pdf(glossy_d, cos_theta, jacobian * fabsf(wi_h));
			s->sampled_flags = BSDF_GLOSSY | BSDF_TRANSMIT;
			ret = cscale(m->filter_color, glossy);
			w[0] = fabsf(wi_n) / fmaxf_(0.1f, s->pdf);
			dir[0] = wi;
		}
		if(two ? (s->flags & BSDF_REFLECT) != 0 : (!take_t && (s->flags & BSDF_REFLECT)))
		{
			wi = reflect_microfacet(wo, h);
			wi_n = vdot(wi, n); wi_h = vdot(wi, h);
			glossy_g = ggx_g(alpha_2, wi_n, wo_n);
			jacobian = 1.f / fmaxf_(1.0e-8f, (4.f * fabsf(wi_h)));
			glossy = (kr * glossy_g * glossy_d) / fmaxf_(1.0e-8f, (4.f * fabsf(wo_n * wi_n)));
			s->pdf = ggx_pdf(glossy_d, cos_theta, jacobian);
			if(two) s->sampled_flags |= BSDF_GLOSSY | BSDF_REFLECT; else s->sampled_flags = BSDF_GLOSSY | BSDF_REFLECT;
			const rgb rc = cscale(m->spec_refl_color, glossy);
			const float ww = fabsf(wi_n) / fmaxf_(0.1f, s->pdf);
			if(two) { *tcol = rc; w[1] = ww; dir[1] = wi; }
			else { ret = rc; w[0] = ww; dir[0] = wi; }
		}
		else if(!two && !take_t) dir[0] = wi;      /* neither lobe asked for: wi is what refractMicrofacet__ left, w untouched */
	}
	else
	{	/* total inner reflection about the half vector */
		wi = vreflect(wo, h);
		if(two) s->sampled_flags |= BSDF_GLOSSY | BSDF_REFLECT; else s->sampled_flags = BSDF_GLOSSY | BSDF_REFLECT;
		dir[0] = wi;
		ret = C(1.f, 1.f, 1.f);
		w[0] = 1.f;
	}
	return ret;
}

static rgb mat_sample(const mat_t *m, const bsdf_dat *dat, const sp_t *sp, v3 wo, v3 *wi, sample_t *s, float *w)
{
	if(m->type == YOR_MAT_ROUGH_GLASS)
	{
		v3 dir[2] = {*wi, *wi}; rgb tcol; float ww[2] = {*w, *w};
		const rgb ret = rough_glass_sample(m, sp, wo, s, 0, dir, &tcol, ww);
		*wi = dir[0]; *w = ww[0];
		return ret;
	}
	if(m->type == YOR_MAT_GLASS)
	{	/* GlassMaterial::sample, material_glass.cc:65-215, the branch without dispersion (:143-213) */
		if(!(s->flags & BSDF_SPECULAR)) { s->pdf = 0.f; return C(0, 0, 0); }
		v3 refdir, n = glass_normal(sp, wo);
		s->pdf = 1.f;
		const unsigned spec_refl = BSDF_SPECULAR | BSDF_REFLECT;
		if(refract_dir(n, wo, &refdir, m->ior))
		{
			float kr, kt;
			fresnel_dielectric(wo, n, m->ior, &kr, &kt);
			float p_kr = (float)(0.01 + 0.99 * (double)kr), p_kt = (float)(0.01 + 0.99 * (double)kt);
			if(s->s_1 < p_kt && ((s->flags & m->tm_flags) == m->tm_flags))
			{
				*wi = refdir; s->pdf = p_kt; s->sampled_flags = m->tm_flags; *w = 1.f;
				return m->filter_color;
			}
			else if((s->flags & spec_refl) == spec_refl)
			{
				const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);
				*wi = V(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
				s->pdf = p_kr; s->sampled_flags = spec_refl; *w = 1.f;
				return m->spec_refl_color;
			}
		}
		else if((s->flags & spec_refl) == spec_refl)
		{	/* total inner reflection */
			const float vn = 2.0f * (wo.x * n.x + wo.y * n.y + wo.z * n.z);
			*wi = V(vn * n.x - wo.x, vn * n.y - wo.y, vn * n.z - wo.z);
			s->sampled_flags = spec_refl; *w = 1.f;
			return C(1.f, 1.f, 1.f);
		}
		s->pdf = 0.f;
		return C(0, 0, 0);
	}
	if(m->type == YOR_MAT_MIRROR)
	{	/* MirrorMaterial::sample, material_glass.cc:467-473: flags ignored, pdf left at Sample's initial 0 */
		*wi = reflect_dir(sp->n, wo);
		s->sampled_flags = BSDF_SPECULAR | BSDF_REFLECT;
		*w = 1.f;
		return cscale(m->ref_col, 1.f / fabsf(vdot(sp->n, *wi)));
	}
	if(m->type == YOR_MAT_SHINYDIFFUSE)
	{	/* material_shiny_diffuse.cc:308-408 */
		float accum_c[4];
		float cos_ng_wo = vdot(sp->ng, wo), cos_ng_wi, cos_n;
		v3 n = face_forward(sp->ng, sp->n, wo);
		float kr = sd_fresnel(m, wo, n);
		sd_accumulate(dat->component, accum_c, kr);
		float sum = 0.f, val[4], width[4];
		unsigned choice[4];
		int n_match = 0, pick = -1;
		for(int i = 0; i < m->n_bsdf; ++i)
		{
			if((s->flags & m->c_flags[i]) == m->c_flags[i])
			{
				width[n_match] = accum_c[m->c_index[i]];
				sum += width[n_match];
				choice[n_match] = m->c_flags[i];
				val[n_match] = sum;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001) { s->sampled_flags = BSDF_NONE; s->pdf = 0.f; return C(1, 1, 1); }
		float inv_sum = 1.f / sum;
		for(int i = 0; i < n_match; ++i)
		{
			val[i] *= inv_sum;
			width[i] *= inv_sum;
			if((s->s_1 <= val[i]) && (pick < 0)) pick = i;
		}
		if(pick < 0) pick = n_match - 1;
		float s_1;
		if(pick > 0) s_1 = (s->s_1 - val[pick - 1]) / width[pick];
		else s_1 = s->s_1 / width[pick];
		rgb scolor = C(0, 0, 0);
		switch(choice[pick])
		{
			case(BSDF_SPECULAR | BSDF_REFLECT):
				*wi = reflect_dir(n, wo);
				s->pdf = width[pick];
				scolor = cscale(m->mirror_color, accum_c[0]);
				scolor = cscale(scolor, 1.f / fmaxf_(fabsf(vdot(sp->n, *wi)), 1.0e-6f));
				break;
			case(BSDF_TRANSMIT | BSDF_FILTER):
				*wi = vneg(wo);
				scolor = cscale(cadd(cscale(m->diffuse_color, m->transmit_filter), C(1.f - m->transmit_filter, 1.f - m->transmit_filter, 1.f - m->transmit_filter)), accum_c[1]);
				cos_n = fabsf(vdot(*wi, n));
				if((double)cos_n < 1e-6) s->pdf = 0.f;
				else s->pdf = width[pick];
				break;
			case(BSDF_DIFFUSE | BSDF_TRANSMIT):
				*wi = sample_cos_hemisphere(vneg(n), sp->nu, sp->nv, s_1, s->s_2);
				cos_ng_wi = vdot(sp->ng, *wi);
				if(cos_ng_wo * cos_ng_wi < 0) scolor = cscale(m->diffuse_color, accum_c[2]);
				s->pdf = fabsf(vdot(*wi, n)) * width[pick]; break;
			case(BSDF_DIFFUSE | BSDF_REFLECT):
			default:
				*wi = sample_cos_hemisphere(n, sp->nu, sp->nv, s_1, s->s_2);
				cos_ng_wi = vdot(sp->ng, *wi);
				if(cos_ng_wo * cos_ng_wi > 0) scolor = cscale(m->diffuse_color, accum_c[3]);
				if(m->use_oren) scolor = cscale(scolor, sd_oren(m, wo, *wi, n));
				s->pdf = fabsf(vdot(*wi, n)) * width[pick]; break;
		}
		s->sampled_flags = choice[pick];
		*w = (fabsf(vdot(*wi, sp->n))) / (s->pdf * 0.99f + 0.01f);
		{
			const float alpha = sd_alpha(m, dat, sp, wo);
			*w = *w * (alpha) + 1.f * (1.f - alpha);
		}
		return scolor;
	}
	else if(m->type == YOR_MAT_GLOSSY)
	{	/* material_glossy.cc:176-357, isotropic (Blinn) branch */
		float cos_ng_wo = vdot(sp->ng, wo), cos_ng_wi;
		v3 n = face_forward(sp->ng, sp->n, wo);
		s->pdf = 0.f;
		float wi_n = 0.f;
		float wo_n = fabsf(vdot(wo, n));
		float cos_wo_h = 0.f;
		rgb scolor = C(0, 0, 0);
		float s_1 = s->s_1;
		float cur_p_diffuse = dat->p_diffuse;
		int use_glossy = m->as_diffuse ? (s->flags & BSDF_DIFFUSE) != 0 : (s->flags & BSDF_GLOSSY) != 0;
		int use_diffuse = m->with_diffuse && (s->flags & BSDF_DIFFUSE);
		float glossy = 0.f;
		if(use_diffuse)
		{
			float s_p_diffuse = use_glossy ? cur_p_diffuse : 1.f;
			if(s_1 < s_p_diffuse)
			{
				s_1 /= s_p_diffuse;
				*wi = sample_cos_hemisphere(n, sp->nu, sp->nv, s_1, s->s_2);
				cos_ng_wi = vdot(sp->ng, *wi);
				if(cos_ng_wi * cos_ng_wo < 0.f) return scolor;
				wi_n = fabsf(vdot(*wi, n));
				s->pdf = wi_n;
				if(use_glossy)
				{
					v3 h = vnormalize(vadd(*wi, wo));
					cos_wo_h = vdot(wo, h);
					float cos_wi_h = fabsf(vdot(*wi, h));
					float cos_n_h = vdot(n, h);
					v3 hl = local_h(sp, h, cos_n_h);
					s->pdf = s->pdf * cur_p_diffuse + lobe_pdf(m, hl, cos_n_h, cos_wo_h) * (1.f - cur_p_diffuse);
					glossy = (float)((double)(lobe_d(m, hl, cos_n_h) * schlick_fresnel(cos_wi_h, dat->m_glossy)) / as_divisor(cos_wi_h, wo_n, wi_n));
				}
				s->sampled_flags = BSDF_DIFFUSE | BSDF_REFLECT;
				if(!(s->flags & BSDF_REFLECT)) return C(0, 0, 0);
				scolor = cscale(m->gloss_color, glossy);
				{
					rgb add_col = diffuse_reflect(wi_n, wo_n, dat->m_glossy, dat->m_diffuse, m->diff_color);
					if(m->has_diffuse_refl) add_col = cscale(add_col, m->diffuse_refl);
					if(m->use_oren) add_col = cscale(add_col, sd_oren(m, *wi, wo, n));
					scolor = cadd(scolor, add_col);
				}
				*w = wi_n / (s->pdf * 0.99f + 0.01f);
				return scolor;
			}
			s_1 -= cur_p_diffuse;
			s_1 /= (1.f - cur_p_diffuse);
		}
		if(use_glossy)
		{
			v3 hs = lobe_sample(m, s_1, s->s_2);
			v3 h = vadd(vadd(vmul(sp->nu, hs.x), vmul(sp->nv, hs.y)), vmul(n, hs.z));
			cos_wo_h = vdot(wo, h);
			if(cos_wo_h < 0.f)
			{	/* h.reflect(n), vector.h:265-272 */
				const float vn = 2.0f * (h.x * n.x + h.y * n.y + h.z * n.z);
				h = V(vn * n.x - h.x, vn * n.y - h.y, vn * n.z - h.z);
				cos_wo_h = vdot(wo, h);
			}
			*wi = reflect_dir(h, wo);
			cos_ng_wi = vdot(sp->ng, *wi);
			if(cos_ng_wo * cos_ng_wi < 0.f) return C(0, 0, 0);
			wi_n = fabsf(vdot(*wi, n));
			{
				float cos_hn = vdot(h, n);
				/* the anisotropic branch keeps the sampled Hs (:299-300), Blinn takes h * n of the (possibly reflected) h (:325-328) */
				s->pdf = lobe_pdf(m, hs, cos_hn, cos_wo_h);
				glossy = (float)((double)(lobe_d(m, hs, cos_hn) * schlick_fresnel(cos_wo_h, dat->m_glossy)) / as_divisor(cos_wo_h, wo_n, wi_n));
			}
			scolor = cscale(m->gloss_color, glossy);
			s->sampled_flags = m->as_diffuse ? (BSDF_DIFFUSE | BSDF_REFLECT) : (BSDF_GLOSSY | BSDF_REFLECT);
		}
		if(use_diffuse)
		{
			rgb add_col = diffuse_reflect(wi_n, wo_n, dat->m_glossy, dat->m_diffuse, m->diff_color);
			if(m->has_diffuse_refl) add_col = cscale(add_col, m->diffuse_refl);
			if(m->use_oren) add_col = cscale(add_col, sd_oren(m, *wi, wo, n));
			s->pdf = wi_n * cur_p_diffuse + s->pdf * (1.f - cur_p_diffuse);
			scolor = cadd(scolor, add_col);
		}
		*w = wi_n / (s->pdf * 0.99f + 0.01f);
		return scolor;
	}
	if(m->type == YOR_MAT_COATED_GLOSSY)
	{	/* material_coated_glossy.cc:188-374, Blinn lobe */
		float cos_ng_wo = vdot(sp->ng, wo), cos_ng_wi;
		v3 n = face_forward(sp->ng, sp->n, wo);
		v3 hs = V(0, 0, 0);
		s->pdf = 0.f;
		float kr, kt, wi_n = 0.f, wo_n = 0.f;
		fresnel_dielectric(wo, n, m->ior, &kr, &kt);
		int use[3] = {0, 0, 0};
		float sum = 0.f, accum_c[3], val[3], width[3];
		int c_index[3], rc_index[3] = {0, 0, 0};
		accum_c[0] = kr;
		accum_c[1] = kt * (1.f - dat->p_diffuse);
		accum_c[2] = kt * (dat->p_diffuse);
		int n_match = 0, pick = -1;
		for(int i = 0; i < m->n_bsdf; ++i)
		{
			if((s->flags & m->c_flags[i]) == m->c_flags[i])
			{
				use[i] = 1;
				width[n_match] = accum_c[i];
				c_index[n_match] = i;
				rc_index[i] = n_match;
				sum += width[n_match];
				val[n_match] = sum;
				++n_match;
			}
		}
		if(!n_match || (double)sum < 0.00001)
		{
			*wi = reflect_dir(n, wo);
			return C(0, 0, 0);
		}
		else if(n_match == 1) { pick = 0; width[0] = 1.f; }
		else
		{
			float inv_sum = 1.f / sum;
			for(int i = 0; i < n_match; ++i)
			{
				val[i] *= inv_sum;
				width[i] *= inv_sum;
				if((s->s_1 <= val[i]) && (pick < 0)) pick = i;
			}
		}
		if(pick < 0) pick = n_match - 1;
		float s_1;
		if(pick > 0) s_1 = (s->s_1 - val[pick - 1]) / width[pick];
		else s_1 = s->s_1 / width[pick];
		rgb scolor = C(0, 0, 0);
		switch(c_index[pick])
		{
			case 0:
				*wi = reflect_dir(n, wo);
				scolor = cscale(cscale(m->mirror_color, kr), m->mirror_strength);
				s->pdf = width[pick];
				break;
			case 1:
				hs = lobe_sample(m, s_1, s->s_2);
				break;
			default:
				*wi = sample_cos_hemisphere(n, sp->nu, sp->nv, s_1, s->s_2);
				cos_ng_wi = vdot(sp->ng, *wi);
				if(cos_ng_wo * cos_ng_wi < 0) return C(0, 0, 0);
		}
		wi_n = fabsf(vdot(*wi, n));
		wo_n = fabsf(vdot(wo, n));
		if(c_index[pick] != 0)
		{
			if(use[1])
			{
				float glossy, cos_wo_h;
				v3 h;
				if(c_index[pick] != 1)
				{
					h = vnormalize(vadd(*wi, wo));
					hs = V(vdot(h, sp->nu), vdot(h, sp->nv), vdot(h, n));
					cos_wo_h = vdot(wo, h);
				}
				else
				{
					h = vadd(vadd(vmul(sp->nu, hs.x), vmul(sp->nv, hs.y)), vmul(n, hs.z));
					cos_wo_h = vdot(wo, h);
					if(cos_wo_h < 0.f)
					{
						const float vn = 2.0f * (h.x * n.x + h.y * n.y + h.z * n.z);
						h = V(vn * n.x - h.x, vn * n.y - h.y, vn * n.z - h.z);
						cos_wo_h = vdot(wo, h);
					}
					*wi = reflect_dir(h, wo);
					cos_ng_wi = vdot(sp->ng, *wi);
					if(cos_ng_wo * cos_ng_wi < 0) return C(0, 0, 0);
				}
				wi_n = fabsf(vdot(*wi, n));
				{
					float cos_hn = vdot(h, n);
					s->pdf += lobe_pdf(m, hs, cos_hn, cos_wo_h) * width[rc_index[1]];
					glossy = (float)((double)(lobe_d(m, hs, cos_hn) * schlick_fresnel(cos_wo_h, dat->m_glossy)) / as_divisor(cos_wo_h, wo_n, wi_n));
				}
				scolor = cscale(m->gloss_color, glossy * kt);
			}
			if(use[2])
			{
				rgb add_col = cscale(diffuse_reflect(wi_n, wo_n, dat->m_glossy, dat->m_diffuse, m->diff_color), kt);
				if(m->has_diffuse_refl) add_col = cscale(add_col, m->diffuse_refl);
				if(m->use_oren) add_col = cscale(add_col, sd_oren(m, *wi, wo, n));
				scolor = cadd(scolor, add_col);
				s->pdf += wi_n * width[rc_index[2]];
			}
			*w = wi_n / (s->pdf * 0.99f + 0.01f);
		}
		else *w = 1.f;
		s->sampled_flags = m->c_flags[c_index[pick]];
		return scolor;
	}
	/* LightMaterial::sample, material_simple.cc:41-46 */
	s->pdf = 0.f; *w = 0.f;
	return C(0, 0, 0);
}

static rgb mat_emit(const mat_t *m, const sp_t *sp, v3 wo, int include_lights)
{
	if(m->type == YOR_MAT_SHINYDIFFUSE) return m->emit_color; /* material_shiny_diffuse.cc:295-306 */
	if(m->type == YOR_MAT_LIGHT)
	{	/* material_simple.cc:50-57 */
		if(!include_lights) return C(0, 0, 0);
		if(m->double_sided) return m->light_col;
		float angle = vdot(wo, sp->n);
		return (angle > 0) ? m->light_col : C(0, 0, 0);
	}
	return C(0, 0, 0);
}

/* ------------------------------------------------------------------ lights */
static void light_configure(light_t *l, const yor_light_desc *d)
{
	memset(l, 0, sizeof *l);
	l->type = d->type; l->samples = d->samples; l->cast_shadows = d->cast_shadows;
	rgb col = C(d->color[0], d->color[1], d->color[2]);
	if(d->type == YOR_LIGHT_AREA)
	{	/* factory light_area.cc:197: (corner, p1 - corner, p2 - corner); ctor :34-52 */
		l->corner = V(d->corner[0], d->corner[1], d->corner[2]);
		l->to_x = vsub(V(d->point1[0], d->point1[1], d->point1[2]), l->corner);
		l->to_y = vsub(V(d->point2[0], d->point2[1], d->point2[2]), l->corner);
		l->fnormal = vcross(l->to_y, l->to_x);
		l->color = cscale(cscale(col, d->power), (float)Y_M_PI);
		{	/* normLen, vector.h:61-71 */
			float vl = l->fnormal.x * l->fnormal.x + l->fnormal.y * l->fnormal.y + l->fnormal.z * l->fnormal.z;
			if(vl != 0.0)
			{
				vl = yor_fsqrt(vl);
				const float dd = (float)(1.0 / (double)vl);
				l->fnormal.x *= dd; l->fnormal.y *= dd; l->fnormal.z *= dd;
			}
			l->area = vl;
		}
		l->inv_area = (float)(1.0 / (double)l->area);
		l->normal = vneg(l->fnormal);
		l->c2 = vadd(l->corner, l->to_x);
		l->c3 = vadd(l->corner, vadd(l->to_x, l->to_y));
		l->c4 = vadd(l->corner, l->to_y);
	}
	else
	{	/* light_point.cc:28-36 */
		l->position = V(d->corner[0], d->corner[1], d->corner[2]);
		l->color = cscale(col, d->power);
	}
}

/* AreaLight::illumSample, light_area.cc:67-97 */
static int arealight_illum_sample(const light_t *l, v3 sp_p, float s_1, float s_2, v3 *wi_dir, float *wi_tmax, float *pdf, rgb *col)
{
	v3 p = vadd(vadd(l->corner, vmul(l->to_x, s_1)), vmul(l->to_y, s_2));
	v3 ldir = vsub(p, sp_p);
	float dist_sqr = ldir.x * ldir.x + ldir.y * ldir.y + ldir.z * ldir.z;
	float dist = yor_fsqrt(dist_sqr);
	if(dist <= 0.0) return 0;
	{
		float inv = 1.f / dist;
		ldir.x *= inv; ldir.y *= inv; ldir.z *= inv;
	}
	float cos_angle = vdot(ldir, l->fnormal);
	if(cos_angle <= 0) return 0;
	*wi_tmax = dist;
	*wi_dir = ldir;
	*col = l->color;
	*pdf = (float)((double)dist_sqr * Y_M_PI / (double)(l->area * cos_angle));
	return 1;
}
/* triIntersect__, light_area.cc:118-137 */
static int tri_intersect_plain(v3 a, v3 b, v3 c, v3 from, v3 dir, float *t)
{
	v3 edge_1 = vsub(b, a), edge_2 = vsub(c, a);
	v3 pvec = vcross(dir, edge_2);
	float det = vdot(edge_1, pvec);
	if(det == 0.0) return 0;
	float inv_det = (float)(1.0 / (double)det);
	v3 tvec = vsub(from, a);
	float u = vdot(tvec, pvec) * inv_det;
	if(u < 0.0 || u > 1.0) return 0;
	v3 qvec = vcross(tvec, edge_1);
	float v = vdot(dir, qvec) * inv_det;
	if((v < 0.0) || ((u + v) > 1.0)) return 0;
	*t = vdot(edge_2, qvec) * inv_det;
	return 1;
}
/* AreaLight::intersect, light_area.cc:139-155 */
static int arealight_intersect(const light_t *l, v3 from, v3 dir, float *t, rgb *col, float *ipdf)
{
	float cos_angle = vdot(dir, l->fnormal);
	if(cos_angle <= 0) return 0;
	if(!tri_intersect_plain(l->corner, l->c2, l->c3, from, dir, t))
	{
		if(!tri_intersect_plain(l->corner, l->c3, l->c4, from, dir, t)) return 0;
	}
	if(!(*t > 1.0e-10f)) return 0;
	*col = l->color;
	*ipdf = (float)((double)(1.f / (*t * *t) * l->area * cos_angle) * Y_M_1_PI);
	return 1;
}
/* PointLight::illuminate, light_point.cc:38-56 */
static int pointlight_illuminate(const light_t *l, v3 sp_p, rgb *col, v3 *wi_dir, float *wi_tmax)
{
	v3 ldir = vsub(l->position, sp_p);
	float dist_sqr = ldir.x * ldir.x + ldir.y * ldir.y + ldir.z * ldir.z;
	float dist = yor_fsqrt(dist_sqr);
	if(dist == 0.0) return 0;
	float idist_sqr = 1.f / (dist_sqr);
	{
		float inv = 1.f / dist;
		ldir.x *= inv; ldir.y *= inv; ldir.z *= inv;
	}
	*wi_tmax = dist;
	*wi_dir = ldir;
	*col = cscale(l->color, (float)idist_sqr);
	return 1;
}

/* ------------------------------------------------------------------ camera
 * Camera::Camera camera.cc:46-66, PerspectiveCamera ctor + setAxis camera_perspective.cc:28-74 */
static void camera_configure(camera_t *c, const yor_camera_desc *d)
{
	v3 pos = V(d->from[0], d->from[1], d->from[2]), look = V(d->to[0], d->to[1], d->to[2]), up = V(d->up[0], d->up[1], d->up[2]);
	c->position = pos; c->resx = d->resx; c->resy = d->resy;
	c->aspect_ratio = d->aspect_ratio * (float)d->resy / (float)d->resx;
	c->cam_y = vsub(up, pos);
	c->cam_z = vsub(look, pos);
	c->cam_x = vcross(c->cam_z, c->cam_y);
	c->cam_y = vcross(c->cam_z, c->cam_x);
	c->cam_x = vnormalize(c->cam_x);
	c->cam_y = vnormalize(c->cam_y);
	c->cam_z = vnormalize(c->cam_z);
	c->near_n = c->cam_z; c->near_p = vadd(pos, vmul(c->cam_z, d->near_clip));
	c->far_n = c->cam_z; c->far_p = vadd(pos, vmul(c->cam_z, d->far_clip));
	c->focal = d->focal; c->aperture = d->aperture;
	c->vright = c->cam_x;
	c->vup = vmul(c->cam_y, c->aspect_ratio);
	c->vto = vsub(vmul(c->cam_z, c->focal), vmul(vadd(c->vup, c->vright), 0.5f));
	c->vup = V(c->vup.x / (float)c->resy, c->vup.y / (float)c->resy, c->vup.z / (float)c->resy);
	c->vright = V(c->vright.x / (float)c->resx, c->vright.y / (float)c->resx, c->vright.z / (float)c->resx);
	/* :42-54, :66-67 */
	c->dof_distance = d->dof_distance; c->bkhtype = d->bokeh_type; c->bkhbias = d->bokeh_bias;
	c->dof_rt = vmul(c->cam_x, c->aperture);
	c->dof_up = vmul(c->cam_y, c->aperture);
	memset(c->ls, 0, sizeof c->ls);
	int ns = c->bkhtype;
	if((ns >= 3) && (ns <= 6))
	{
		float w = (float)((double)d->bokeh_rotation * 0.01745329251994329576922), wi = (float)(6.28318530717958647692 / (double)(float)ns);
		ns = (ns + 2) * 2;
		for(int i = 0; i < ns; i += 2)
		{
			c->ls[i] = yor_fcos(w);
			c->ls[i + 1] = yor_fsin(w);
			w += wi;
		}
	}
}
/* PerspectiveCamera::biasDist, :75-89 */
static void camera_bias_dist(const camera_t *c, float *r)
{
	switch(c->bkhbias)
	{
		case 1: *r = yor_fsqrt(yor_fsqrt(*r) * *r); break;
		case 2: *r = yor_fsqrt((float)1.0 - *r * *r); break;
		default: *r = yor_fsqrt(*r);
	}
}
/* shirleyDisk__, vector.cc:155-190 */
static void shirley_disk(float r_1, float r_2, float *u, float *v)
{
	float phi = 0, r = 0, a = 2 * r_1 - 1, b = 2 * r_2 - 1;
	if(a > -b)
	{
		if(a > b) { r = a; phi = (float)(Y_M_PI_4 * (double)(b / a)); }
		else { r = b; phi = (float)(Y_M_PI_4 * (double)(2 - a / b)); }
	}
	else
	{
		if(a < b) { r = -a; phi = (float)(Y_M_PI_4 * (double)(4 + b / a)); }
		else
		{
			r = -b;
			if(b != 0) phi = (float)(Y_M_PI_4 * (double)(6 - a / b));
			else phi = 0;
		}
	}
	*u = r * yor_fcos(phi);
	*v = r * yor_fsin(phi);
}
/* PerspectiveCamera::getLensUv / sampleTsd, :91-131 */
static void camera_lens_uv(const camera_t *c, float r_1, float r_2, float *u, float *v)
{
	switch(c->bkhtype)
	{
		case 3: case 4: case 5: case 6:
		{
			float fn = (float)c->bkhtype;
			int idx = (int)(r_1 * fn);
			r_1 = (r_1 - ((float)idx) / fn) * fn;
			camera_bias_dist(c, &r_1);
			float b_1 = r_1 * r_2;
			float b_0 = r_1 - b_1;
			idx <<= 1;
			*u = c->ls[idx] * b_0 + c->ls[idx + 2] * b_1;
			*v = c->ls[idx + 1] * b_0 + c->ls[idx + 3] * b_1;
			break;
		}
		case 1: case 7:
		{
			float w = (float)6.28318530717958647692 * r_2;
			if(c->bkhtype == 7) r_1 = yor_fsqrt((float)0.707106781 + (float)0.292893218);
			else camera_bias_dist(c, &r_1);
			*u = r_1 * yor_fcos(w);
			*v = r_1 * yor_fsin(w);
			break;
		}
		default: shirley_disk(r_1, r_2, u, v);
	}
}
/* rayPlaneIntersection__, util_geometry.h:34-37 */
static inline float ray_plane(v3 from, v3 dir, v3 pp, v3 pn) { return vdot(pn, vsub(pp, from)) / vdot(dir, pn); }
/* shootRay, camera_perspective.cc:133-156 */
static void camera_shoot_lens(const camera_t *c, float px, float py, float lu, float lv, v3 *from, v3 *dir, float *tmin, float *tmax, float *wt)
{
	*wt = 1;
	*from = c->position;
	*dir = vadd(vadd(vmul(c->vright, px), vmul(c->vup, py)), c->vto);
	*dir = vnormalize(*dir);
	*tmin = ray_plane(*from, *dir, c->near_p, c->near_n);
	*tmax = ray_plane(*from, *dir, c->far_p, c->far_n);
	if(c->aperture != 0)
	{
		float u, v;
		camera_lens_uv(c, lu, lv, &u, &v);
		v3 li = vadd(vmul(c->dof_rt, u), vmul(c->dof_up, v));
		*from = vadd(*from, li);
		*dir = vsub(vmul(*dir, c->dof_distance), li);
		*dir = vnormalize(*dir);
	}
}
static void camera_shoot(const camera_t *c, float px, float py, v3 *from, v3 *dir, float *tmin, float *tmax, float *wt)
{
	camera_shoot_lens(c, px, py, 0.5f, 0.5f, from, dir, tmin, tmax, wt);
}

/* ------------------------------------------------------------------ scene create */
yor_scene *yor_scene_create(int32_t n_tris, const float *verts, const int32_t *tri_mat, const float *vnormals,
                            int32_t n_mats, const yor_material_desc *mats,
                            int32_t n_lights, const yor_light_desc *lights,
                            const yor_camera_desc *cam)
{
	yor_scene *s = (yor_scene *)calloc(1, sizeof *s);
	s->n_tris = n_tris;
	s->tris = (tri_t *)calloc((size_t)(n_tris > 0 ? n_tris : 1), sizeof(tri_t));
	for(int i = 0; i < n_tris; ++i)
	{
		tri_t *t = &s->tris[i];
		const float *p = verts + 9 * (size_t)i;
		t->a = V(p[0], p[1], p[2]); t->b = V(p[3], p[4], p[5]); t->c = V(p[6], p[7], p[8]);
		t->e1 = vsub(t->b, t->a); t->e2 = vsub(t->c, t->a);              /* triangle.h:203-204 */
		/* triangle.h:206: 0.1f * MIN_RAYDIST * max(|e1|,|e2|) — MIN_RAYDIST is a double literal */
		t->eps = (float)((double)0.1f * MIN_RAYDIST * (double)fmaxf_(vlength(t->e1), vlength(t->e2)));
		t->ng = vnormalize(vcross(vsub(t->b, t->a), vsub(t->c, t->a))); /* recNormal triangle.h:295-302 */
		t->mat = tri_mat[i];
		t->smooth = 0;
		if(vnormals)
		{
			const float *q = vnormals + 9 * (size_t)i;
			int any = 0;
			for(int k = 0; k < 9; ++k) if(q[k] != 0.f) any = 1;
			if(any)
			{
				t->smooth = 1;
				t->na = V(q[0], q[1], q[2]); t->nb = V(q[3], q[4], q[5]); t->nc = V(q[6], q[7], q[8]);
				if(t->na.x == 0 && t->na.y == 0 && t->na.z == 0) t->na = t->ng;
				if(t->nb.x == 0 && t->nb.y == 0 && t->nb.z == 0) t->nb = t->ng;
				if(t->nc.x == 0 && t->nc.y == 0 && t->nc.z == 0) t->nc = t->ng;
			}
		}
	}
	s->n_mats = n_mats;
	s->mats = (mat_t *)calloc((size_t)(n_mats > 0 ? n_mats : 1), sizeof(mat_t));
	for(int i = 0; i < n_mats; ++i) mat_configure(&s->mats[i], &mats[i]);
	s->n_lights = n_lights;
	s->lights = (light_t *)calloc((size_t)(n_lights > 0 ? n_lights : 1), sizeof(light_t));
	for(int i = 0; i < n_lights; ++i) light_configure(&s->lights[i], &lights[i]);
	camera_configure(&s->cam, cam);
	build_tree(s);
	return s;
}
void yor_scene_destroy(yor_scene *s)
{
	if(!s) return;
	for(int i = 0; i < s->n_mats; ++i) { free(s->mats[i].nodes); free(s->mats[i].bump_nodes); }
	for(int i = 0; i < s->n_tex; ++i) free(s->tex[i].px);
	free(s->tex); free(s->tri_uv); free(s->tri_orco);
	free(s->tris); free(s->mats); free(s->lights); free(s->nodes); free(s->leaf_refs); free(s);
}

void yor_scene_set_textures(yor_scene *s, int32_t n_textures, const yor_texture_desc *textures)
{
	for(int i = 0; i < s->n_tex; ++i) free(s->tex[i].px);
	free(s->tex);
	s->n_tex = n_textures > 0 ? n_textures : 0;
	s->tex = (tex_t *)calloc((size_t)(s->n_tex > 0 ? s->n_tex : 1), sizeof(tex_t));
	for(int i = 0; i < s->n_tex; ++i) tex_configure(&s->tex[i], &textures[i]);
}
void yor_scene_set_texcoords(yor_scene *s, const float *uv, const float *orco)
{
	free(s->tri_uv); free(s->tri_orco); s->tri_uv = NULL; s->tri_orco = NULL;
	if(uv && s->n_tris > 0) { s->tri_uv = (float *)malloc((size_t)s->n_tris * 6 * sizeof(float)); memcpy(s->tri_uv, uv, (size_t)s->n_tris * 6 * sizeof(float)); }
	if(orco && s->n_tris > 0) { s->tri_orco = (float *)malloc((size_t)s->n_tris * 9 * sizeof(float)); memcpy(s->tri_orco, orco, (size_t)s->n_tris * 9 * sizeof(float)); }
}
void yor_texture_probe(const yor_texture_desc *d, const float p[3], float out5[5])
{
	tex_t t; tex_configure(&t, d);
	rgba_t c = tex_get_color(&t, V(p[0], p[1], p[2]));
	out5[0] = c.r; out5[1] = c.g; out5[2] = c.b; out5[3] = c.a; out5[4] = tex_get_float(&t, V(p[0], p[1], p[2]));
	free(t.px);
}
/* evalDerivative of every node at a surface point given as 31 floats: the 18 of yor_nodes_probe, ds_du (3), ds_dv (3), nu (3), nv (3),
 * has_uv; out = n_nodes x (du, dv, 0, alpha, f).  bump9 (may be NULL): n, nu, nv after Material::applyBump with the LAST node's
 * derivative times `bump_scale`. */
void yor_nodes_probe_derivative(int32_t n_nodes, const yor_node_desc *nodes, int32_t n_textures, const yor_texture_desc *textures, const yor_camera_desc *cam,
                                const float sp31[31], float bump_scale, float *out, float *bump9)
{
	node_t nd[YOR_MAX_NODES]; node_result_t stack[YOR_MAX_NODES];
	if(n_nodes > YOR_MAX_NODES) n_nodes = YOR_MAX_NODES;
	for(int k = 0; k < n_nodes; ++k) node_configure(&nd[k], &nodes[k]);
	tex_t *tx = (tex_t *)calloc((size_t)(n_textures > 0 ? n_textures : 1), sizeof(tex_t));
	for(int i = 0; i < n_textures; ++i) tex_configure(&tx[i], &textures[i]);
	camera_t c; memset(&c, 0, sizeof c);
	if(cam) camera_configure(&c, cam);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.p = V(sp31[0], sp31[1], sp31[2]); sp.n = V(sp31[3], sp31[4], sp31[5]); sp.ng = V(sp31[6], sp31[7], sp31[8]);
	sp.orco_p = V(sp31[9], sp31[10], sp31[11]); sp.orco_ng = V(sp31[12], sp31[13], sp31[14]); sp.u = sp31[15]; sp.v = sp31[16];
	sp.ds_du = V(sp31[18], sp31[19], sp31[20]); sp.ds_dv = V(sp31[21], sp31[22], sp31[23]);
	sp.nu = V(sp31[24], sp31[25], sp31[26]); sp.nv = V(sp31[27], sp31[28], sp31[29]); sp.has_uv = sp31[30] != 0.f;
	nodes_eval_derivative(nd, n_nodes, tx, n_textures, &c, &sp, stack);
	for(int k = 0; k < n_nodes; ++k) { out[5 * k] = stack[k].col.r; out[5 * k + 1] = stack[k].col.g; out[5 * k + 2] = stack[k].col.b; out[5 * k + 3] = stack[k].col.a; out[5 * k + 4] = stack[k].f; }
	if(bump9 && n_nodes > 0)
	{
		apply_bump(&sp, stack[n_nodes - 1].col.r * bump_scale, stack[n_nodes - 1].col.g * bump_scale);
		const float r[9] = {sp.n.x, sp.n.y, sp.n.z, sp.nu.x, sp.nu.y, sp.nu.z, sp.nv.x, sp.nv.y, sp.nv.z};
		memcpy(bump9, r, sizeof r);
	}
	for(int i = 0; i < n_textures; ++i) free(tx[i].px);
	free(tx);
}
void yor_nodes_probe(int32_t n_nodes, const yor_node_desc *nodes, int32_t n_textures, const yor_texture_desc *textures, const yor_camera_desc *cam,
                     const float sp18[18], float *out)
{
	node_t nd[YOR_MAX_NODES]; node_result_t stack[YOR_MAX_NODES];
	if(n_nodes > YOR_MAX_NODES) n_nodes = YOR_MAX_NODES;
	for(int k = 0; k < n_nodes; ++k) node_configure(&nd[k], &nodes[k]);
	tex_t *tx = (tex_t *)calloc((size_t)(n_textures > 0 ? n_textures : 1), sizeof(tex_t));
	for(int i = 0; i < n_textures; ++i) tex_configure(&tx[i], &textures[i]);
	camera_t c; memset(&c, 0, sizeof c);
	if(cam) camera_configure(&c, cam);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.p = V(sp18[0], sp18[1], sp18[2]); sp.n = V(sp18[3], sp18[4], sp18[5]); sp.ng = V(sp18[6], sp18[7], sp18[8]);
	sp.orco_p = V(sp18[9], sp18[10], sp18[11]); sp.orco_ng = V(sp18[12], sp18[13], sp18[14]); sp.u = sp18[15]; sp.v = sp18[16];
	nodes_eval(nd, n_nodes, tx, n_textures, &c, &sp, stack);
	for(int k = 0; k < n_nodes; ++k) { out[5 * k] = stack[k].col.r; out[5 * k + 1] = stack[k].col.g; out[5 * k + 2] = stack[k].col.b; out[5 * k + 3] = stack[k].col.a; out[5 * k + 4] = stack[k].f; }
	for(int i = 0; i < n_textures; ++i) free(tx[i].px);
	free(tx);
}

void yor_scene_set_tree(yor_scene *s, uint32_t n_nodes, const uint32_t *nodes, uint32_t n_refs, const uint32_t *refs, const float bound6[6])
{
	free(s->nodes); free(s->leaf_refs);
	s->nodes = (kdnode_t *)malloc(sizeof(kdnode_t) * (n_nodes ? n_nodes : 1));
	s->leaf_refs = (uint32_t *)malloc(sizeof(uint32_t) * (n_refs ? n_refs : 1));
	memcpy(s->nodes, nodes, sizeof(kdnode_t) * n_nodes);
	memcpy(s->leaf_refs, refs, sizeof(uint32_t) * n_refs);
	s->n_nodes = s->cap_nodes = n_nodes; s->n_refs = s->cap_refs = n_refs;
	s->tb_a = V(bound6[0], bound6[1], bound6[2]); s->tb_g = V(bound6[3], bound6[4], bound6[5]);
}

/* ------------------------------------------------------------------ integrator */
typedef struct
{
	const yor_scene *s;
	const yor_render_desc *rd;
	float shadow_bias, ray_min_dist;
	counters_t cn;
	/* RenderState subset, scene.h:74-118 */
	unsigned pixel_sample;
	unsigned sampling_offs;
	int include_lights;
	mwc_t *prng;
	unsigned correlative_sample_number; /* integrator_tiled.h:91, per thread */
	float light_mult;                   /* aa_light_sample_multiplier_ of the current pass (integrator_tiled.cc:139,215) */
	/* trajectory splitting (scene.h:88-91): set by recursiveRaytrace's glossy branch, read by every sampler below it */
	int ray_division, ray_offset; float dc_1, dc_2;
} rstate_t;
static inline float add_mod_1(float a, float b) { float s = a + b; return s > 1 ? s - 1.f : s; } /* util_sample.h:183-187 */

/* MonteCarloIntegrator::doLightEstimation, integrator_montecarlo.cc:78-345 (render passes disabled,
 * tr_shad_ false, volume integrator = identity) */
static rgb do_light_estimation(rstate_t *st, const light_t *light, const sp_t *sp, const mat_t *material, const bsdf_dat *dat, v3 wo, unsigned loffs)
{
	const yor_scene *s = st->s;
	rgb col = C(0, 0, 0);
	int shadowed;
	unsigned l_offs = loffs * 4567u; /* LOFFS_DELTA :45 */
	v3 lr_dir = V(0, 0, 0); float lr_tmin = 0.f, lr_tmax = -1.f;
	rgb lcol = C(0, 0, 0);
	int cast_shadows = light->cast_shadows && material->receive_shadows;
	if(light->type == YOR_LIGHT_POINT)
	{	/* :94-148 */
		if(pointlight_illuminate(light, sp->p, &lcol, &lr_dir, &lr_tmax))
		{
			if(st->rd->shadow_bias_auto) lr_tmin = st->shadow_bias * fmaxf_(1.f, vlength(sp->p));
			else lr_tmin = st->shadow_bias;
			rgb scol = C(1.f, 1.f, 1.f);
			const int tr_shad = st->rd->transp_shad;
			if(cast_shadows) shadowed = tr_shad ? scene_is_shadowed_ts(s, sp->p, lr_dir, lr_tmin, lr_tmax, st->rd->shadow_depth, &scol, &st->cn)
			                                     : scene_is_shadowed(s, sp->p, lr_dir, lr_tmin, lr_tmax, &st->cn);
			else shadowed = 0;
			float angle_light_normal = (material->flat ? 1.f : fabsf(vdot(sp->n, lr_dir)));
			if(!shadowed)
			{
				if(tr_shad && cast_shadows) lcol = cmul(lcol, scol);                 /* :114 */
				rgb surf_col = mat_eval(material, dat, sp, wo, lr_dir, BSDF_ALL);
				col = cadd(col, cscale(cmul(surf_col, lcol), angle_light_normal));
			}
		}
	}
	else
	{	/* :149-342 */
		halton_t hal_2, hal_3;
		halton_init(&hal_2, 2); halton_init(&hal_3, 3);
		int n = (int)ceilf((float)light->samples * st->light_mult);
		if(st->ray_division > 1) { n = n / st->ray_division; if(n < 1) n = 1; }      /* :154 */
		float inv_ns = 1.f / (float)n;
		unsigned offs = (unsigned)n * st->pixel_sample + st->sampling_offs + l_offs;
		rgb ccol = C(0, 0, 0);
		halton_set_start(&hal_2, offs - 1);
		halton_set_start(&hal_3, offs - 1);
		for(int i = 0; i < n; ++i)
		{
			float ls_s1 = halton_next(&hal_2);
			float ls_s2 = halton_next(&hal_3);
			float ls_pdf; rgb ls_col;
			if(arealight_illum_sample(light, sp->p, ls_s1, ls_s2, &lr_dir, &lr_tmax, &ls_pdf, &ls_col))
			{
				if(st->rd->shadow_bias_auto) lr_tmin = st->shadow_bias * fmaxf_(1.f, vlength(sp->p));
				else lr_tmin = st->shadow_bias;
				rgb scol = C(1.f, 1.f, 1.f);
				const int tr_shad = st->rd->transp_shad;
				if(cast_shadows) shadowed = tr_shad ? scene_is_shadowed_ts(s, sp->p, lr_dir, lr_tmin, lr_tmax, st->rd->shadow_depth, &scol, &st->cn)
				                                     : scene_is_shadowed(s, sp->p, lr_dir, lr_tmin, lr_tmax, &st->cn);
				else shadowed = 0;
				if(!shadowed && ls_pdf > 1e-6f)
				{
					if(tr_shad && cast_shadows) ls_col = cmul(ls_col, scol);          /* :182 */
					rgb surf_col = mat_eval(material, dat, sp, wo, lr_dir, BSDF_ALL);
					float angle_light_normal = (material->flat ? 1.f : fabsf(vdot(sp->n, lr_dir)));
					/* canIntersect() is true for area lights (light_area.h) */
					float m_pdf = mat_pdf(material, dat, sp, wo, lr_dir, BSDF_GLOSSY | BSDF_DIFFUSE | BSDF_DISPERSIVE | BSDF_REFLECT | BSDF_TRANSMIT);
					if(m_pdf > 1e-6f)
					{
						float l_2 = ls_pdf * ls_pdf;
						float m_2 = m_pdf * m_pdf;
						float w = l_2 / (l_2 + m_2);
						ccol = cadd(ccol, cdiv(cscale(cscale(cmul(surf_col, ls_col), angle_light_normal), w), ls_pdf));
					}
					else
					{
						ccol = cadd(ccol, cdiv(cscale(cmul(surf_col, ls_col), angle_light_normal), ls_pdf));
					}
				}
			}
		}
		col = cadd(col, cscale(ccol, inv_ns));
		{	/* BSDF-sampling half of MIS :285-333 */
			rgb ccol_2 = C(0, 0, 0);
			halton_set_start(&hal_2, offs - 1);
			halton_set_start(&hal_3, offs - 1);
			for(int i = 0; i < n; ++i)
			{
				float b_tmin, b_tmax = -1.f; v3 b_dir = V(0, 0, 0);
				if(st->rd->min_raydist_auto) b_tmin = st->ray_min_dist * fmaxf_(1.f, vlength(sp->p));
				else b_tmin = st->ray_min_dist;
				float s_1 = halton_next(&hal_2);
				float s_2 = halton_next(&hal_3);
				float W = 0.f;
				sample_t sm; sm.s_1 = s_1; sm.s_2 = s_2; sm.pdf = 0.f; sm.sampled_flags = BSDF_NONE;
				sm.flags = BSDF_GLOSSY | BSDF_DIFFUSE | BSDF_DISPERSIVE | BSDF_REFLECT | BSDF_TRANSMIT;
				rgb surf_col = mat_sample(material, dat, sp, wo, &b_dir, &sm, &W);
				float light_pdf;
				if(sm.pdf > 1e-6f && arealight_intersect(light, sp->p, b_dir, &b_tmax, &lcol, &light_pdf))
				{
					rgb scol = C(1.f, 1.f, 1.f);
					const int tr_shad = st->rd->transp_shad;
					if(cast_shadows) shadowed = tr_shad ? scene_is_shadowed_ts(s, sp->p, b_dir, b_tmin, b_tmax, st->rd->shadow_depth, &scol, &st->cn)
					                                     : scene_is_shadowed(s, sp->p, b_dir, b_tmin, b_tmax, &st->cn);
					else shadowed = 0;
					if(!shadowed && light_pdf > 1e-6f)
					{
						if(tr_shad && cast_shadows) lcol = cmul(lcol, scol);          /* :309 */
						float l_pdf = 1.f / light_pdf;
						float l_2 = l_pdf * l_pdf;
						float m_2 = sm.pdf * sm.pdf;
						float w = m_2 / (l_2 + m_2);
						ccol_2 = cadd(ccol_2, cscale(cscale(cmul(surf_col, lcol), w), W));
					}
				}
			}
			col = cadd(col, cscale(ccol_2, inv_ns));
		}
	}
	return col;
}

/* estimateAllDirectLight, integrator_montecarlo.cc:47-60 */
static rgb estimate_all_direct_light(rstate_t *st, const sp_t *sp, const mat_t *material, const bsdf_dat *dat, v3 wo)
{
	rgb col = C(0, 0, 0);
	for(int l = 0; l < st->s->n_lights; ++l)
		col = cadd(col, do_light_estimation(st, &st->s->lights[l], sp, material, dat, wo, (unsigned)l));
	return col;
}
/* estimateOneDirectLight, :62-76.  With one light lnum is 0 whatever the counter holds. */
static rgb estimate_one_direct_light(rstate_t *st, const sp_t *sp, const mat_t *material, const bsdf_dat *dat, v3 wo)
{
	int light_num = st->s->n_lights;
	if(light_num == 0) return C(0, 0, 0);
	halton_t hal_2; halton_init(&hal_2, 2);
	halton_set_start(&hal_2, st->rd->base_sampling_offset + st->correlative_sample_number - 1);
	int lnum = (int)(halton_next(&hal_2) * (float)light_num);
	if(lnum > light_num - 1) lnum = light_num - 1;
	++st->correlative_sample_number;
	return cscale(do_light_estimation(st, &st->s->lights[lnum], sp, material, dat, wo, (unsigned)lnum), (float)light_num);
}

/* PathIntegrator::integrate, integrator_path_tracer.cc:112-347 (raylevel 0, no caustics, no
 * recursive raytrace: materials with specular/glossy/filter lobes are rejected by yor_render) */
/* ray_tmax_out (may be NULL): the ray's tmax_ after the call, i.e. the distance to the first hit or the caller's tmax on a miss */
static void integrate_d(rstate_t *st, v3 from, v3 dir, float tmin, float tmax, int raylevel, int additional_depth, float out_rgba[4], float *ray_tmax_out);
static void integrate(rstate_t *st, v3 from, v3 dir, float tmin, float tmax, int raylevel, float out_rgba[4], float *ray_tmax_out)
{
	integrate_d(st, from, dir, tmin, tmax, raylevel, 0, out_rgba, ray_tmax_out);
}
/* additional_depth: integrate()'s parameter of that name (integrator_path_tracer.cc:112,149): the largest Material::additional_depth_ met on the way down */
static void integrate_d(rstate_t *st, v3 from, v3 dir, float tmin, float tmax, int raylevel, int additional_depth, float out_rgba[4], float *ray_tmax_out)
{
	const yor_scene *s = st->s;
	const yor_render_desc *rd = st->rd;
	rgb col = C(0, 0, 0);
	float alpha;
	sp_t sp;
	float w = 0.f;
	if(rd->bg_transp) alpha = 0.0f;
	else alpha = 1.0f;
	if(scene_intersect(s, from, dir, tmin, &tmax, &sp, &st->cn))
	{
		if(raylevel == 0) st->include_lights = 1; /* :129-135 */
		unsigned bsdfs;
		bsdf_dat dat0;
		mat_t mat_here; const mat_t *material = mat_at(s, &sp, &mat_here);
		mat_init_bsdf(material, &dat0, &bsdfs);
		if(additional_depth < material->additional_depth) additional_depth = material->additional_depth;     /* :149 */
		v3 wo = vneg(dir);
		if(bsdfs & BSDF_EMIT) col = cadd(col, mat_emit(material, &sp, wo, st->include_lights));
		if(bsdfs & BSDF_DIFFUSE) col = cadd(col, estimate_all_direct_light(st, &sp, material, &dat0, wo));
		unsigned path_flags = rd->no_recursive ? BSDF_ALL : (BSDF_DIFFUSE);
		if(rd->integrator == YOR_INTEGRATOR_PATH && (bsdfs & path_flags))
		{
			rgb path_col = C(0, 0, 0);
			path_flags |= (BSDF_DIFFUSE | BSDF_REFLECT | BSDF_TRANSMIT);
			int n_samples = rd->path_samples / st->ray_division; if(n_samples < 1) n_samples = 1; /* max(1, n_paths_ / ray_division_) :182 */
			for(int i = 0; i < n_samples; ++i)
			{
				unsigned offs = (unsigned)rd->path_samples * st->pixel_sample + st->sampling_offs + (unsigned)i;
				rgb throughput, lcol, scol;
				sp_t sp_1 = sp, sp_2;
				sp_t *hit = &sp_1, *hit_2 = &sp_2;
				v3 pwo = wo;
				v3 p_dir = V(0, 0, 0); float p_tmin, p_tmax;
				bsdf_dat dat_n;
				float s_1 = yor_ri_vdc(offs, 0);
				float s_2 = (float)yor_scr_halton(2, offs);
				if(st->ray_division > 1) { s_1 = add_mod_1(s_1, st->dc_1); s_2 = add_mod_1(s_2, st->dc_2); }   /* :201-205 (the same lines in the bounce loop, :238-242, shift locals nothing reads) */
				sample_t sm; sm.s_1 = s_1; sm.s_2 = s_2; sm.pdf = 0.f; sm.flags = path_flags; sm.sampled_flags = BSDF_NONE;
				scol = mat_sample(material, &dat0, &sp, pwo, &p_dir, &sm, &w);
				scol = cscale(scol, w);
				throughput = scol;
				st->include_lights = 0;
				p_tmin = st->ray_min_dist;
				p_tmax = -1.0f;
				if(!scene_intersect(s, sp.p, p_dir, p_tmin, &p_tmax, hit, &st->cn)) continue;
				mat_t mat_hit; const mat_t *p_mat = mat_at(s, hit, &mat_hit);
				unsigned mat_bsdfs;
				mat_init_bsdf(p_mat, &dat_n, &mat_bsdfs);
				if(sm.sampled_flags != BSDF_NONE) pwo = vneg(p_dir);
				lcol = estimate_one_direct_light(st, hit, p_mat, &dat_n, pwo);
				if(mat_bsdfs & BSDF_EMIT) lcol = cadd(lcol, mat_emit(p_mat, hit, pwo, st->include_lights));
				path_col = cadd(path_col, cmul(lcol, throughput));
				for(int depth = 1; depth < rd->bounces; ++depth)
				{
					int d_4 = 4 * depth;
					sm.s_1 = (float)yor_scr_halton(d_4 + 3, offs);
					sm.s_2 = (float)yor_scr_halton(d_4 + 4, offs);
					sm.flags = BSDF_ALL;
					scol = mat_sample(p_mat, &dat_n, hit, pwo, &p_dir, &sm, &w);
					scol = cscale(scol, w);
					if(cblack(scol)) break;
					throughput = cmul(throughput, scol);
					/* :252-253 a bounce through a specular, glossy or filter lobe makes the next vertex show its lights */
					const int caustic = rd->trace_caustics && (sm.sampled_flags & (BSDF_SPECULAR | BSDF_GLOSSY | BSDF_FILTER)) != 0;
					st->include_lights = caustic;
					p_tmin = st->ray_min_dist;
					p_tmax = -1.0f;
					if(!scene_intersect(s, hit->p, p_dir, p_tmin, &p_tmax, hit_2, &st->cn)) break;
					{ sp_t *tmp = hit; hit = hit_2; hit_2 = tmp; }
					p_mat = mat_at(s, hit, &mat_hit);
					mat_init_bsdf(p_mat, &dat_n, &mat_bsdfs);
					pwo = vneg(p_dir);
					if(mat_bsdfs & BSDF_DIFFUSE) lcol = estimate_one_direct_light(st, hit, p_mat, &dat_n, pwo);
					else lcol = C(0, 0, 0);
					/* :276-279 the segment just travelled, if it ran inside an absorbing material (getVolumeHandler(inside)) */
					if((mat_bsdfs & BSDF_VOLUMETRIC) && vdot(hit->n, pwo) < 0 && p_mat->has_vol_i)
						throughput = cmul(throughput, beer_transmittance(p_mat->beer_sigma, p_tmax));
					if(depth > rd->rr_min_bounces)
					{	/* Russian roulette :282-288 */
						float random_value = (float)mwc_next(st->prng);
						float probability = fmaxf_(throughput.r, fmaxf_(throughput.g, throughput.b));
						if(probability <= 0.f || probability < random_value) break;
						throughput = cscale(throughput, 1.f / probability);
					}
					if((mat_bsdfs & BSDF_EMIT) && caustic) lcol = cadd(lcol, mat_emit(p_mat, hit, pwo, st->include_lights));      /* :290 */
					path_col = cadd(path_col, cmul(lcol, throughput));
				}
			}
			col = cadd(col, cdiv(path_col, (float)n_samples));
		}
		/* recursiveRaytrace, integrator_montecarlo.cc:782-1028: the perfect specular branch (:971-1025); no dispersive
		 * or glossy-recursive materials on this path, additional depth and transparent bias 0 */
		/* the glossy branch, :861-972: gsam trajectories through the glossy lobe, each a full integrate() one level down with the
		 * trajectory-splitting state set (materials here reflect only: the Reflect && !Transmit case, :897-918) */
		if(raylevel + 1 <= rd->raydepth + additional_depth && (bsdfs & BSDF_GLOSSY) && raylevel + 1 < 20)
		{
			st->include_lights = 1;
			int gsam = 8;
			const int old_division = st->ray_division, old_offset = st->ray_offset;
			const float old_dc_1 = st->dc_1, old_dc_2 = st->dc_2;
			if(st->ray_division > 1) { gsam = gsam / old_division; if(gsam < 1) gsam = 1; }
			st->ray_division *= gsam;
			int branch = st->ray_division * old_offset;
			unsigned offs = (unsigned)gsam * st->pixel_sample + st->sampling_offs;
			float d_1 = 1.f / (float)gsam;
			rgb gcol = C(0, 0, 0);
			halton_t hal_2, hal_3;
			halton_init(&hal_2, 2); halton_init(&hal_3, 3);
			halton_set_start(&hal_2, offs);
			halton_set_start(&hal_3, offs);
			for(int ns = 0; ns < gsam; ++ns)
			{
				st->dc_1 = (float)yor_scr_halton(2 * (raylevel + 1) + 1, (unsigned)branch + st->sampling_offs);
				st->dc_2 = (float)yor_scr_halton(2 * (raylevel + 1) + 2, (unsigned)branch + st->sampling_offs);
				st->ray_offset = branch;
				++offs; ++branch;
				float s_1 = halton_next(&hal_2);
				float s_2 = halton_next(&hal_3);
				if((material->flags & BSDF_GLOSSY) && (material->flags & BSDF_REFLECT) && !(material->flags & BSDF_TRANSMIT))
				{
					float gw = 0.f; v3 wi = V(0, 0, 0);
					sample_t sm; sm.s_1 = s_1; sm.s_2 = s_2; sm.pdf = 0.f; sm.flags = BSDF_GLOSSY | BSDF_REFLECT; sm.sampled_flags = BSDF_NONE;
					rgb mcol = mat_sample(material, &dat0, &sp, wo, &wi, &sm, &gw);
					float integ[4], ref_tmax;
					integrate_d(st, sp.p, wi, st->ray_min_dist, -1.0f, raylevel + 1, additional_depth, integ, &ref_tmax);
					rgb ic = C(integ[0], integ[1], integ[2]);
					if((bsdfs & BSDF_VOLUMETRIC) && vdot(sp.ng, wi) < 0 && material->has_vol_i) ic = cmul(ic, beer_transmittance(material->beer_sigma, ref_tmax));
					gcol = cadd(gcol, cscale(cmul(ic, mcol), gw));
				}
				else if((material->flags & BSDF_GLOSSY) && (material->flags & BSDF_REFLECT) && (material->flags & BSDF_TRANSMIT))
				{	/* :919-970: a lobe that reflects and transmits (rough glass): the two-direction sample, an integrate() per direction it reports.
					 * (dir[0] / mcol[0] / w[0] go with the `Reflect` test, dir[1] / mcol[1] / w[1] with `Transmit`, whatever the material put there.) */
					sample_t sm; sm.s_1 = s_1; sm.s_2 = s_2; sm.pdf = 0.f; sm.flags = BSDF_GLOSSY | BSDF_REFLECT | BSDF_TRANSMIT; sm.sampled_flags = BSDF_NONE;
					v3 gdir[2] = {V(0, 0, 0), V(0, 0, 0)}; rgb mcol[2] = {C(0, 0, 0), C(0, 0, 0)}; float gw[2] = {0.f, 0.f};
					mcol[0] = rough_glass_sample(material, &sp, wo, &sm, 1, gdir, &mcol[1], gw);
					if((sm.sampled_flags & BSDF_REFLECT) && !(sm.sampled_flags & BSDF_DISPERSIVE))
					{
						float integ[4], ref_tmax;
						integrate_d(st, sp.p, gdir[0], st->ray_min_dist, -1.0f, raylevel + 1, additional_depth, integ, &ref_tmax);
						rgb ic = C(integ[0], integ[1], integ[2]);
						if((bsdfs & BSDF_VOLUMETRIC) && vdot(sp.ng, gdir[0]) < 0 && material->has_vol_i) ic = cmul(ic, beer_transmittance(material->beer_sigma, ref_tmax));
						gcol = cadd(gcol, cmul(ic, cscale(mcol[0], gw[0])));
					}
					if(sm.sampled_flags & BSDF_TRANSMIT)
					{
						float integ[4], ref_tmax;
						integrate_d(st, sp.p, gdir[1], st->ray_min_dist, -1.0f, raylevel + 1, additional_depth, integ, &ref_tmax);
						rgb ic = C(integ[0], integ[1], integ[2]);
						if((bsdfs & BSDF_VOLUMETRIC) && vdot(sp.ng, gdir[1]) < 0 && material->has_vol_i) ic = cmul(ic, beer_transmittance(material->beer_sigma, ref_tmax));
						gcol = cadd(gcol, cmul(ic, cscale(mcol[1], gw[1])));
						alpha = integ[3];
					}
				}
			}
			col = cadd(col, cscale(gcol, d_1));
			st->ray_division = old_division; st->ray_offset = old_offset; st->dc_1 = old_dc_1; st->dc_2 = old_dc_2;
		}
		if(raylevel + 1 <= rd->raydepth + additional_depth && (bsdfs & (BSDF_SPECULAR | BSDF_FILTER)) && raylevel + 1 < 20)
		{
			st->include_lights = 1;
			int reflect = 0, refract = 0;
			v3 sdir[2]; rgb rcol[2];
			mat_get_specular(material, &dat0, &sp, wo, raylevel + 1, &reflect, &refract, sdir, rcol);
			if(reflect)
			{
				float integ[4], ref_tmax;
				integrate_d(st, sp.p, sdir[0], st->ray_min_dist, -1.0f, raylevel + 1, additional_depth, integ, &ref_tmax);
				rgb ic = C(integ[0], integ[1], integ[2]);
				/* :991-994 vol = material->getVolumeHandler(sp.ng_ * ref_ray.dir_ < 0); integ *= vcol */
				if((bsdfs & BSDF_VOLUMETRIC) && vdot(sp.ng, sdir[0]) < 0 && material->has_vol_i) ic = cmul(ic, beer_transmittance(material->beer_sigma, ref_tmax));
				col = cadd(col, cmul(ic, rcol[0]));
			}
			if(refract)
			{
				float integ[4], ref_tmax;
				v3 r_from = sp.p;
				float transp_bias_factor = material->transp_bias_factor;                     /* :1003-1011 */
				if(transp_bias_factor > 0.f)
				{
					if(material->transp_bias_mult) transp_bias_factor *= (float)(raylevel + 1);
					r_from = vadd(sp.p, vmul(sdir[1], transp_bias_factor));
				}
				integrate_d(st, r_from, sdir[1], st->ray_min_dist, -1.0f, raylevel + 1, additional_depth, integ, &ref_tmax);
				rgb ic = C(integ[0], integ[1], integ[2]);
				if((bsdfs & BSDF_VOLUMETRIC) && vdot(sp.ng, sdir[1]) < 0 && material->has_vol_i) ic = cmul(ic, beer_transmittance(material->beer_sigma, ref_tmax)); /* :1016-1019 */
				col = cadd(col, cmul(ic, rcol[1]));
				alpha = integ[3];
			}
		}
		if(rd->bg_transp_refract)
		{
			float m_alpha = mat_alpha(material, &dat0, &sp, wo);
			alpha = m_alpha + (1.f - m_alpha) * alpha;
		}
		else alpha = 1.0f;
	}
	else
	{
		if(rd->has_background && !rd->bg_transp_refract) col = cadd(col, C(rd->background[0], rd->background[1], rd->background[2]));
	}
	/* EmptyVolumeIntegrator: transmittance 1, integration 0 (integrator_empty_volume.cc:32-38) */
	if(rd->bg_transp) alpha = fmaxf_(alpha, 1.f - 1.f);
	out_rgba[0] = col.r; out_rgba[1] = col.g; out_rgba[2] = col.b; out_rgba[3] = alpha;
	if(ray_tmax_out) *ray_tmax_out = tmax;
}

/* ------------------------------------------------------------------ film
 * ImageFilm ctor imagefilm.cc:124-187, addSample :925-1015 */
#define FILTER_TABLE_SIZE 16
#define MAX_FILTER_SIZE 8
typedef struct
{
	int w, h, cx0, cx1, cy0, cy1;
	float filterw, table_scale;
	float table[FILTER_TABLE_SIZE * FILTER_TABLE_SIZE];
	float *pix; /* h*w*5 */
	float clamp_samples;              /* aa_clamp_samples_ */
	const unsigned char *flags;       /* adaptive passes: doMoreSamples (imagefilm.cc:917-920), NULL = every pixel */
} film_t;

/* Rgb::clampProportionalRgb, color.h:412-445 */
static void clamp_proportional_rgb(float c[3], float max_value)
{
	if(max_value > 0.f)
	{
		float max_rgb = fmaxf_(c[0], fmaxf_(c[1], c[2]));
		float proportional_adjustment = max_value / max_rgb;
		if(max_rgb > max_value)
		{
			if(c[0] >= max_rgb) { c[0] = max_value; c[1] *= proportional_adjustment; c[2] *= proportional_adjustment; }
			else if(c[1] >= max_rgb) { c[1] = max_value; c[0] *= proportional_adjustment; c[2] *= proportional_adjustment; }
			else { c[2] = max_value; c[0] *= proportional_adjustment; c[1] *= proportional_adjustment; }
		}
	}
}

static float filt_box(float dx, float dy) { (void)dx; (void)dy; return 1.f; }
static float filt_mitchell(float dx, float dy) /* :85-97 */
{
	float x = 2.f * yor_fsqrt(dx * dx + dy * dy);
	if(x >= 2.f) return (0.f);
	if(x >= 1.f) return (float)(x * (x * (x * -0.38888889f + 2.0f) - 3.33333333f) + 1.77777778f);
	return (float)(x * x * (1.16666666f * x - 2.0f) + 0.88888889f);
}
static float filt_gauss(float dx, float dy) /* :100-104 ; fExp__ = fExp2__((float)M_LOG2E * a) */
{
	float r_2 = dx * dx + dy * dy;
	float e = yor_fexp2((float)1.4426950408889634074 * (float)(-6 * r_2));
	return fmaxf_(0.f, (float)((double)e - 0.00247875));
}
static float filt_lanczos(float dx, float dy) /* :107-121 */
{
	float x = yor_fsqrt(dx * dx + dy * dy);
	if(x == 0.f) return 1.f;
	if(-2 < x && x < 2)
	{
		float a = (float)(Y_M_PI * (double)x);
		float b = (float)(Y_M_PI_2 * (double)x);
		return ((yor_fsin(a) * yor_fsin(b)) / (a * b));
	}
	return 0.f;
}
static inline int round2int(double val) { return (int)(val + (.5 - 1.4e-11)); } /* util_math.h:34-43 */

static void film_init(film_t *f, const yor_render_desc *rd, float *pix)
{
	f->w = rd->width; f->h = rd->height; f->cx0 = rd->xstart; f->cy0 = rd->ystart;
	f->cx1 = rd->xstart + rd->width; f->cy1 = rd->ystart + rd->height;
	f->filterw = (float)((double)rd->aa_pixelwidth * 0.5);
	float (*ffunc)(float, float) = filt_box;
	switch(rd->filter_type)
	{
		case YOR_FILTER_MITCHELL: ffunc = filt_mitchell; f->filterw *= 2.6f; break;
		case YOR_FILTER_LANCZOS: ffunc = filt_lanczos; break;
		case YOR_FILTER_GAUSS: ffunc = filt_gauss; f->filterw *= 2.f; break;
		default: ffunc = filt_box;
	}
	f->filterw = fminf_(fmaxf_(0.501f, f->filterw), 0.5f * MAX_FILTER_SIZE);
	float scale = 1.f / (float)FILTER_TABLE_SIZE;
	float *tp = f->table;
	for(int y = 0; y < FILTER_TABLE_SIZE; ++y)
		for(int x = 0; x < FILTER_TABLE_SIZE; ++x)
			*tp++ = ffunc((x + .5f) * scale, (y + .5f) * scale);
	f->table_scale = (float)(0.9999 * FILTER_TABLE_SIZE / (double)f->filterw);
	f->pix = pix;
	f->clamp_samples = rd->aa_clamp_samples;
	f->flags = NULL;
}

typedef struct { int x, y; float c[4]; float wt; } splat_t;
typedef struct { splat_t *v; size_t n, cap; } splat_list;

/* addSample :925-1015, combined pass only, aa_clamp_samples_ = 0, premult false.
 * own_only != NULL: contributions to pixels other than (x,y) are deferred into the list (used by
 * the multi-threaded mode so that tiles never write each other's pixels concurrently). */
static void film_add_sample(film_t *f, const float col_in[4], int x, int y, float dx, float dy, splat_list *deferred)
{
	int dx_0, dx_1, dy_0, dy_1, x_0, x_1, y_0, y_1;
	float col[4] = {col_in[0], col_in[1], col_in[2], col_in[3]};
	clamp_proportional_rgb(col, f->clamp_samples);            /* :975 (the same for every pixel of the footprint) */
	dx_0 = round2int((double)dx - (double)f->filterw); if(f->cx0 - x > dx_0) dx_0 = f->cx0 - x;
	dx_1 = round2int((double)dx + (double)f->filterw - 1.0); if(f->cx1 - x - 1 < dx_1) dx_1 = f->cx1 - x - 1;
	dy_0 = round2int((double)dy - (double)f->filterw); if(f->cy0 - y > dy_0) dy_0 = f->cy0 - y;
	dy_1 = round2int((double)dy + (double)f->filterw - 1.0); if(f->cy1 - y - 1 < dy_1) dy_1 = f->cy1 - y - 1;
	double x_offs = (double)dx - 0.5;
	int x_index[MAX_FILTER_SIZE + 1], y_index[MAX_FILTER_SIZE + 1];
	for(int i = dx_0, n = 0; i <= dx_1; ++i, ++n)
	{
		double d = fabs(((double)i - x_offs) * (double)f->table_scale);
		x_index[n] = (int)floor(d);
	}
	double y_offs = (double)dy - 0.5;
	for(int i = dy_0, n = 0; i <= dy_1; ++i, ++n)
	{
		double d = fabs(((double)i - y_offs) * (double)f->table_scale);
		y_index[n] = (int)floor(d);
	}
	x_0 = x + dx_0; x_1 = x + dx_1;
	y_0 = y + dy_0; y_1 = y + dy_1;
	for(int j = y_0; j <= y_1; ++j)
		for(int i = x_0; i <= x_1; ++i)
		{
			int offset = y_index[j - y_0] * FILTER_TABLE_SIZE + x_index[i - x_0];
			float filter_wt = f->table[offset];
			if(deferred && (i != x || j != y))
			{
				if(deferred->n == deferred->cap)
				{
					deferred->cap = deferred->cap ? deferred->cap * 2 : 256;
					deferred->v = (splat_t *)realloc(deferred->v, deferred->cap * sizeof(splat_t));
				}
				splat_t *sp = &deferred->v[deferred->n++];
				sp->x = i; sp->y = j; memcpy(sp->c, col, sizeof sp->c); sp->wt = filter_wt;
				continue;
			}
			float *p = f->pix + 5 * ((size_t)(j - f->cy0) * (size_t)f->w + (size_t)(i - f->cx0));
			p[0] += col[0] * filter_wt; p[1] += col[1] * filter_wt; p[2] += col[2] * filter_wt; p[3] += col[3] * filter_wt;
			p[4] += filter_wt;
		}
}

/* ------------------------------------------------------------------ renderTile
 * TiledIntegrator::renderTile, integrator_tiled.cc:309-521 (single pass, not adaptive) */
typedef struct
{
	const yor_scene *s; const yor_render_desc *rd; film_t *film;
	int n_tiles_x, n_tiles_y;
	int thread_id, n_threads;
	int *next_tile;
	counters_t cn; uint64_t camera_samples;
	splat_list deferred;
	/* the pass being rendered: renderPass(samples, offset, adaptive) (integrator_tiled.cc:261-307) */
	int pass_samples, pass_offset, pass_adaptive; float light_mult;
	/* correlative_sample_number_[thread] (integrator_tiled.h:91): zeroed once per render, before the first pass
	 * (integrator_tiled.cc:192-194), and carried over the passes */
	unsigned correlative_sample_number;
	/* per-tile rand() values for the tile seeds (integrator_tiled.cc:319), drawn in the order tiles are started; NULL: rd->tile_seed_rand for all */
	const int *tile_rand; int *tile_rand_next;
} worker_t;

static void render_tile(worker_t *wk, int tx, int ty, rstate_t *st)
{
	const yor_render_desc *rd = wk->rd;
	const camera_t *cam = &wk->s->cam;
	int ax = rd->xstart + tx * rd->tile_size, ay = rd->ystart + ty * rd->tile_size;
	int end_x = ax + rd->tile_size, end_y = ay + rd->tile_size;
	if(end_x > rd->xstart + rd->width) end_x = rd->xstart + rd->width;
	if(end_y > rd->ystart + rd->height) end_y = rd->ystart + rd->height;
	int n_samples = wk->pass_samples;
	int offset = wk->pass_offset + (int)rd->base_sampling_offset; /* renderPass(samples, offset) + base offset, :203,263 */
	int x = cam->resx;
	float dx = 0.5, dy = 0.5, d_1 = (float)(1.0 / (double)(float)n_samples);
	float wt;
	mwc_t prng;
	/* :319 — rand() of libc is replaced by rd->tile_seed_rand; only Russian roulette consumes the stream */
	int tile_rand = (int)rd->tile_seed_rand;
	if(wk->tile_rand) tile_rand = wk->tile_rand[__atomic_fetch_add(wk->tile_rand_next, 1, __ATOMIC_RELAXED)];
	mwc_init(&prng, (uint32_t)(tile_rand + offset * (x * ay + ax) + 123));
	if(getenv("YOR_VERBOSE") && ty == 0) fprintf(stderr, "[oracle] pass offset %d tile (%d,%d): seed %u, light counter %u\n", offset, tx, ty, (uint32_t)(tile_rand + offset * (x * ay + ax) + 123), st->correlative_sample_number);
	st->prng = &prng;
	int pass_offs = offset;
	for(int i = ay; i < end_y; ++i)
	{
		for(int j = ax; j < end_x; ++j)
		{
			if(wk->pass_adaptive && wk->film->flags && !wk->film->flags[(size_t)(i - rd->ystart) * (size_t)rd->width + (size_t)(j - rd->xstart)]) continue; /* :355 */
			st->sampling_offs = yor_fnv32a((uint32_t)i * yor_fnv32a((uint32_t)j)); /* :379 */
			halton_t hal_u, hal_v;                                                  /* :337-338, :382-383 */
			halton_init(&hal_u, 3); halton_init(&hal_v, 5);
			halton_set_start(&hal_u, (uint32_t)pass_offs + st->sampling_offs);
			halton_set_start(&hal_v, (uint32_t)pass_offs + st->sampling_offs);
			float lens_u = 0.5f, lens_v = 0.5f;
			for(int sample = 0; sample < n_samples; ++sample)
			{
				st->pixel_sample = (unsigned)(pass_offs + sample);
				if(rd->aa_passes > 1)
				{	/* :394-398: scrambled van der Corput / Sobol for multi-pass AA */
					dx = yor_ri_vdc(st->pixel_sample, st->sampling_offs);
					dy = yor_ri_s(st->pixel_sample, st->sampling_offs);
				}
				else if(n_samples > 1)
				{	/* :399-403 (aa_passes_ == 1) */
					dx = (float)((0.5 + (double)(float)sample) * (double)d_1);
					dy = yor_ri_lp((uint32_t)sample + st->sampling_offs, 0);
				}
				v3 from, dir; float tmin, tmax;
				if(cam->aperture != 0) { lens_u = halton_next(&hal_u); lens_v = halton_next(&hal_v); }   /* :405-409 */
				camera_shoot_lens(cam, j + dx, i + dy, lens_u, lens_v, &from, &dir, &tmin, &tmax, &wt);
				wk->camera_samples++;
				float c[4];
				g_trace_px = (float)j; g_trace_py = (float)i;
				integrate(st, from, dir, tmin, tmax, 0, c, NULL);
				if(c[3] > 1.f) c[3] = 1.f;                 /* :459 */
				c[0] *= wt; c[1] *= wt; c[2] *= wt; c[3] *= wt; /* :512 */
				if(g_trace_samples)
				{
					if(g_trace_n_samples < g_trace_samples_cap)
					{
						float *rec = g_trace_samples + 8 * g_trace_n_samples;
						rec[0] = (float)j; rec[1] = (float)i; rec[2] = dx; rec[3] = dy; rec[4] = c[0]; rec[5] = c[1]; rec[6] = c[2]; rec[7] = c[3];
					}
					++g_trace_n_samples;
				}
				film_add_sample(wk->film, c, j, i, dx, dy, wk->n_threads > 1 ? &wk->deferred : NULL);
			}
		}
	}
}

static void *worker_main(void *arg)
{
	worker_t *wk = (worker_t *)arg;
	const yor_render_desc *rd = wk->rd;
	rstate_t st;
	memset(&st, 0, sizeof st);
	st.s = wk->s; st.rd = rd;
	st.shadow_bias = rd->shadow_bias_auto ? (float)YAF_SHADOW_BIAS : rd->shadow_bias;   /* scene.cc:825 */
	st.ray_min_dist = rd->min_raydist_auto ? (float)MIN_RAYDIST : rd->min_raydist;      /* scene.cc:826 */
	st.light_mult = wk->light_mult;
	st.ray_division = 1;                                                                /* RenderState(), scene.h:76 */
	st.correlative_sample_number = wk->correlative_sample_number;
	int n_tiles = wk->n_tiles_x * wk->n_tiles_y;
	int shard_count = rd->shard_count > 0 ? rd->shard_count : 1;
	if(wk->n_threads == 1)
	{
		for(int t = 0; t < n_tiles; ++t)
		{
			if((t % shard_count) != rd->shard_index) continue;
			render_tile(wk, t % wk->n_tiles_x, t / wk->n_tiles_x, &st);
		}
	}
	else
	{	/* workers pull tiles from a shared counter, like renderWorker does from ImageFilm::nextArea (integrator_tiled.cc:48-66) */
		for(;;)
		{
			int t = __atomic_fetch_add(wk->next_tile, 1, __ATOMIC_RELAXED);
			if(t >= n_tiles) break;
			if((t % shard_count) != rd->shard_index) continue;
			render_tile(wk, t % wk->n_tiles_x, t / wk->n_tiles_x, &st);
		}
	}
	wk->cn = st.cn;
	wk->correlative_sample_number = st.correlative_sample_number;
	return NULL;
}

/* ImageFilm::darkThresholdCurveInterpolate, imagefilm.cc:1312-1328 */
static float dark_threshold_curve(float b)
{
	if(b <= 0.10f) return 0.0001f;
	else if(b > 0.10f && b <= 0.20f) return (0.0001f + (b - 0.10f) * (0.0010f - 0.0001f) / 0.10f);
	else if(b > 0.20f && b <= 0.30f) return (0.0010f + (b - 0.20f) * (0.0020f - 0.0010f) / 0.10f);
	else if(b > 0.30f && b <= 0.40f) return (0.0020f + (b - 0.30f) * (0.0035f - 0.0020f) / 0.10f);
	else if(b > 0.40f && b <= 0.50f) return (0.0035f + (b - 0.40f) * (0.0055f - 0.0035f) / 0.10f);
	else if(b > 0.50f && b <= 0.60f) return (0.0055f + (b - 0.50f) * (0.0075f - 0.0055f) / 0.10f);
	else if(b > 0.60f && b <= 0.70f) return (0.0075f + (b - 0.60f) * (0.0100f - 0.0075f) / 0.10f);
	else if(b > 0.70f && b <= 0.80f) return (0.0100f + (b - 0.70f) * (0.0150f - 0.0100f) / 0.10f);
	else if(b > 0.80f && b <= 0.90f) return (0.0150f + (b - 0.80f) * (0.0250f - 0.0150f) / 0.10f);
	else if(b > 0.90f && b <= 1.00f) return (0.0250f + (b - 0.90f) * (0.0400f - 0.0250f) / 0.10f);
	else if(b > 1.00f && b <= 1.20f) return (0.0400f + (b - 1.00f) * (0.0800f - 0.0400f) / 0.20f);
	else if(b > 1.20f && b <= 1.40f) return (0.0800f + (b - 1.20f) * (0.0950f - 0.0800f) / 0.20f);
	else if(b > 1.40f && b <= 1.80f) return (0.0950f + (b - 1.40f) * (0.1000f - 0.0950f) / 0.40f);
	else return 0.1000f;
}

/* Pixel::normalized (util_image_buffers.h:39-43) with Rgba / float (color.h:310-314: multiply by the rounded reciprocal) */
static void pixel_normalized(const float *p, float out[4])
{
	float w = p[4];
	if(w != 0.f) { float f = (float)(1.0 / (double)w); out[0] = p[0] * f; out[1] = p[1] * f; out[2] = p[2] * f; out[3] = p[3] * f; }
	else { out[0] = out[1] = out[2] = out[3] = 0.f; }
}
/* Rgba::colorDifference, color.h:447-464 */
static float color_difference(const float a[4], const float b[4], int use_rgb)
{
	float bri_a = 0.2126f * a[0] + 0.7152f * a[1] + 0.0722f * a[2];
	float bri_b = 0.2126f * b[0] + 0.7152f * b[1] + 0.0722f * b[2];
	float d = fabsf(bri_b - bri_a);
	if(use_rgb)
	{
		float rd = fabsf(b[0] - a[0]), gd = fabsf(b[1] - a[1]), bd = fabsf(b[2] - a[2]), ad = fabsf(b[3] - a[3]);
		if(d < rd) d = rd;
		if(d < gd) d = gd;
		if(d < bd) d = bd;
		if(d < ad) d = ad;
	}
	return d;
}

/* ImageFilm::nextPass, imagefilm.cc:270-480: which pixels get more samples.  Returns their number.
 * (no sampling-factor pass, not interactive) */
static int film_next_pass(const film_t *f, const yor_render_desc *rd, float aa_thesh, unsigned char *flags)
{
	const int w = f->w, h = f->h;
	memset(flags, 0, (size_t)w * (size_t)h);
	if(!(aa_thesh > 0.f)) { memset(flags, 1, (size_t)w * (size_t)h); return h * w; }   /* :319,460; doMoreSamples :919 */
	const int variance_half_edge = rd->aa_variance_edge_size / 2;
	float aa_thresh_scaled = aa_thesh;
#define PIX(x, y) (f->pix + 5 * ((size_t)(y) * (size_t)w + (size_t)(x)))
#define SET(x, y) flags[(size_t)(y) * (size_t)w + (size_t)(x)] = 1
	for(int y = 0; y < h - 1; ++y)
	{
		for(int x = 0; x < w - 1; ++x)
		{
			if(PIX(x, y)[4] <= 0.f) SET(x, y);                                              /* :335 */
			float pix_col[4], other[4];
			pixel_normalized(PIX(x, y), pix_col);
			float pix_col_bri = 0.2126f * fabsf(pix_col[0]) + 0.7152f * fabsf(pix_col[1]) + 0.0722f * fabsf(pix_col[2]);
			if(rd->aa_dark_detection_type == 1 && rd->aa_dark_threshold_factor > 0.f)
				aa_thresh_scaled = aa_thesh * ((1.f - rd->aa_dark_threshold_factor) + (pix_col_bri * rd->aa_dark_threshold_factor));
			else if(rd->aa_dark_detection_type == 2) aa_thresh_scaled = dark_threshold_curve(pix_col_bri);
			pixel_normalized(PIX(x + 1, y), other);
			if(color_difference(pix_col, other, rd->aa_detect_color_noise) >= aa_thresh_scaled) { SET(x, y); SET(x + 1, y); }
			pixel_normalized(PIX(x, y + 1), other);
			if(color_difference(pix_col, other, rd->aa_detect_color_noise) >= aa_thresh_scaled) { SET(x, y); SET(x, y + 1); }
			pixel_normalized(PIX(x + 1, y + 1), other);
			if(color_difference(pix_col, other, rd->aa_detect_color_noise) >= aa_thresh_scaled) { SET(x, y); SET(x + 1, y + 1); }
			if(x > 0)
			{
				pixel_normalized(PIX(x - 1, y + 1), other);
				if(color_difference(pix_col, other, rd->aa_detect_color_noise) >= aa_thresh_scaled) { SET(x, y); SET(x - 1, y + 1); }
			}
			if(rd->aa_variance_pixels > 0)
			{
				int variance_x = 0, variance_y = 0;
				for(int xd = -variance_half_edge; xd < variance_half_edge - 1; ++xd)
				{
					int xi = x + xd;
					if(xi < 0) xi = 0; else if(xi >= w - 1) xi = w - 2;
					float c0[4], c1[4];
					pixel_normalized(PIX(xi, y), c0); pixel_normalized(PIX(xi + 1, y), c1);
					if(color_difference(c0, c1, rd->aa_detect_color_noise) >= aa_thresh_scaled) ++variance_x;
				}
				for(int yd = -variance_half_edge; yd < variance_half_edge - 1; ++yd)
				{
					int yi = y + yd;
					if(yi < 0) yi = 0; else if(yi >= h - 1) yi = h - 2;
					float c0[4], c1[4];
					pixel_normalized(PIX(x, yi), c0); pixel_normalized(PIX(x, yi + 1), c1);
					if(color_difference(c0, c1, rd->aa_detect_color_noise) >= aa_thresh_scaled) ++variance_y;
				}
				if(variance_x + variance_y >= rd->aa_variance_pixels)
				{
					for(int xd = -variance_half_edge; xd < variance_half_edge; ++xd)
						for(int yd = -variance_half_edge; yd < variance_half_edge; ++yd)
						{
							int xi = x + xd; if(xi < 0) xi = 0; else if(xi >= w) xi = w - 1;
							int yi = y + yd; if(yi < 0) yi = 0; else if(yi >= h) yi = h - 1;
							SET(xi, yi);
						}
				}
			}
		}
	}
#undef PIX
#undef SET
	int n = 0;
	for(size_t i = 0; i < (size_t)w * (size_t)h; ++i) n += flags[i];
	return n;
}

/* one renderPass (integrator_tiled.cc:261-307): all tiles, then the deferred cross-pixel splats */
static void run_pass(yor_scene *s, const yor_render_desc *rd, film_t *film, worker_t *wk, pthread_t *th, int nthreads,
                     int samples, int offset, int adaptive, float light_mult)
{
	int next_tile = 0;
	for(int i = 0; i < nthreads; ++i)
	{
		wk[i].next_tile = &next_tile;
		wk[i].pass_samples = samples; wk[i].pass_offset = offset; wk[i].pass_adaptive = adaptive; wk[i].light_mult = light_mult;
		wk[i].deferred.v = NULL; wk[i].deferred.n = 0; wk[i].deferred.cap = 0;
		memset(&wk[i].cn, 0, sizeof wk[i].cn);
	}
	(void)s;
	if(nthreads == 1) worker_main(&wk[0]);
	else
	{
		for(int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker_main, &wk[i]);
		for(int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
		/* deferred cross-pixel splats, applied in thread order (sum order differs from the
		 * single-thread run by at most the position of these few additions) */
		for(int i = 0; i < nthreads; ++i)
		{
			for(size_t k = 0; k < wk[i].deferred.n; ++k)
			{
				splat_t *sp = &wk[i].deferred.v[k];
				float *p = film->pix + 5 * ((size_t)(sp->y - film->cy0) * (size_t)film->w + (size_t)(sp->x - film->cx0));
				p[0] += sp->c[0] * sp->wt; p[1] += sp->c[1] * sp->wt; p[2] += sp->c[2] * sp->wt; p[3] += sp->c[3] * sp->wt;
				p[4] += sp->wt;
			}
			free(wk[i].deferred.v);
			wk[i].deferred.v = NULL;
		}
	}
	(void)rd;
}

int yor_render(yor_scene *s, const yor_render_desc *rd, float *film_out, yor_stats *stats)
{
	{ const char *lp = getenv("YOR_LOG_RAYS"); if(g_ray_log) { fclose(g_ray_log); g_ray_log = NULL; } if(lp && *lp) g_ray_log = fopen(lp, "wb"); }
	if(rd->aa_passes < 1) return -1;
	if(rd->bounces > 12) return -2; /* scrHalton__ dims >= 50 are a racy LCG in the reference */
	for(int i = 0; i < s->n_mats; ++i)
		if(s->mats[i].flags & BSDF_DISPERSIVE) return -4; /* recursiveRaytrace: the dispersive branch is not restated */
	if(rd->tile_size <= 0 || rd->width <= 0 || rd->height <= 0 || rd->aa_minsamples <= 0) return -5;
	if(rd->aa_passes > 1 && rd->shard_count > 1) return -6; /* the noise detection needs the whole frame */
	struct timespec t0, t1;
	memset(film_out, 0, sizeof(float) * 5 * (size_t)rd->width * (size_t)rd->height);
	film_t film;
	film_init(&film, rd, film_out);
	int ntx = (rd->width + rd->tile_size - 1) / rd->tile_size, nty = (rd->height + rd->tile_size - 1) / rd->tile_size;
	int nthreads = rd->n_threads > 0 ? rd->n_threads : 1;
	worker_t *wk = (worker_t *)calloc((size_t)nthreads, sizeof(worker_t));
	pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
	counters_t total; memset(&total, 0, sizeof total);
	uint64_t camera_samples = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	int *tile_rand = NULL, tile_rand_next = 0;
	if(rd->rand_srand >= 0)
	{	/* one rand() per tile started, over all passes, after the values the constructors consumed */
		int total = (rd->rand_skip > 0 ? rd->rand_skip : 0) + ntx * nty * rd->aa_passes;
		int32_t *all = (int32_t *)malloc((size_t)total * sizeof(int32_t));
		yor_glibc_rand((uint32_t)rd->rand_srand, total, all);
		tile_rand = (int *)malloc((size_t)(ntx * nty * rd->aa_passes) * sizeof(int));
		for(int i = 0; i < ntx * nty * rd->aa_passes; ++i) tile_rand[i] = all[i + (rd->rand_skip > 0 ? rd->rand_skip : 0)];
		free(all);
	}
	for(int i = 0; i < nthreads; ++i)
	{
		wk[i].s = s; wk[i].rd = rd; wk[i].film = &film; wk[i].n_tiles_x = ntx; wk[i].n_tiles_y = nty;
		wk[i].thread_id = i; wk[i].n_threads = nthreads;
		wk[i].tile_rand = tile_rand; wk[i].tile_rand_next = &tile_rand_next;
	}
#define TALLY() do { for(int i = 0; i < nthreads; ++i) { total.rays_closest += wk[i].cn.rays_closest; total.rays_shadow += wk[i].cn.rays_shadow; \
	total.interior += wk[i].cn.interior; total.leaves += wk[i].cn.leaves; total.tests += wk[i].cn.tests; camera_samples += wk[i].camera_samples; wk[i].camera_samples = 0; } } while(0)
	/* TiledIntegrator::render, integrator_tiled.cc:116-258 */
	const int aa_samples = rd->aa_minsamples > 1 ? rd->aa_minsamples : 1;
	const int aa_inc_samples = rd->aa_inc_samples > 0 ? rd->aa_inc_samples : aa_samples;            /* scene.cc:765 */
	float aa_threshold = rd->aa_threshold;
	float aa_sample_multiplier = 1.f, aa_light_sample_multiplier = 1.f;
	const int aa_resampled_floor_pixels = (int)floorf(rd->aa_resampled_floor * (float)(rd->width * rd->height) / 100.f);
	run_pass(s, rd, &film, wk, th, nthreads, aa_samples, 0, 0, rd->aa_passes > 1 ? aa_light_sample_multiplier : rd->aa_light_sample_multiplier);
	TALLY();
	unsigned char *flags = NULL;
	if(rd->aa_passes > 1) flags = (unsigned char *)malloc((size_t)rd->width * (size_t)rd->height);
	int acum_aa_samples = aa_samples;
	int aa_threshold_changed = 1, resampled_pixels = 0;
	for(int i = 1; i < rd->aa_passes; ++i)
	{
		aa_sample_multiplier *= rd->aa_sample_multiplier_factor;
		aa_light_sample_multiplier *= rd->aa_light_sample_multiplier_factor;
		if(resampled_pixels <= 0 && !aa_threshold_changed) { /* :222-226: pass skipped */ }
		else
		{
			resampled_pixels = film_next_pass(&film, rd, aa_threshold, flags);
			aa_threshold_changed = 0;
		}
		int aa_samples_mult = (int)ceilf((float)aa_inc_samples * aa_sample_multiplier);
		if(resampled_pixels > 0)
		{
			film.flags = (aa_threshold > 0.f) ? flags : NULL;
			run_pass(s, rd, &film, wk, th, nthreads, aa_samples_mult, acum_aa_samples, 1, aa_light_sample_multiplier);
			TALLY();
		}
		acum_aa_samples += aa_samples_mult;
		if(resampled_pixels < aa_resampled_floor_pixels)
		{
			float aa_variation_ratio = fminf_(8.f, ((float)aa_resampled_floor_pixels / (float)resampled_pixels));
			aa_threshold *= (1.f - 0.1f * aa_variation_ratio);
			if(aa_threshold > 0.f) aa_threshold_changed = 1;
		}
	}
#undef TALLY
	free(flags);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(stats)
	{
		memset(stats, 0, sizeof *stats);
		stats->rays_closest = total.rays_closest; stats->rays_shadow = total.rays_shadow;
		stats->interior_steps = total.interior; stats->leaves = total.leaves; stats->tri_tests = total.tests;
		stats->camera_samples = camera_samples;
		stats->kd_nodes = s->n_nodes; stats->kd_leaf_refs = s->n_refs;
		stats->build_seconds = s->build_seconds;
		stats->render_seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	}
	free(wk); free(th); free(tile_rand);
	return 0;
}

/* test hook: record what renderTile hands to ImageFilm::addSample (8 floats per sample: x, y, dx, dy, r, g, b, a) and every closest-hit
 * query (12 floats: from, dir, tmin, tmax, t or -1, triangle index as int bits, the pixel being rendered), in call order; with_shadow:
 * the plain any-hit queries too (the caller's ray, the verdict in slot 8, -2 in the triangle slot).  Only meaningful for n_threads = 1.
 * NULL pointers switch it off.  yor_trace_counts reports how many of each the last renders produced (may exceed the capacities). */
void yor_set_trace(float *samples8, uint64_t cap_samples, float *rays12, uint64_t cap_rays, int with_shadow)
{
	g_trace_samples = samples8; g_trace_samples_cap = cap_samples; g_trace_n_samples = 0;
	g_trace_rays = rays12; g_trace_rays_cap = cap_rays; g_trace_n_rays = 0;
	g_trace_shadow = with_shadow;
}
void yor_trace_counts(uint64_t *n_samples, uint64_t *n_rays) { if(n_samples) *n_samples = g_trace_n_samples; if(n_rays) *n_rays = g_trace_n_rays; }

/* ------------------------------------------------------------------ ray-level entry points */
int yor_intersect(const yor_scene *s, int use_tree, const float from[3], const float dir[3], float tmin, float tmax,
                  int32_t *tri, float *t, float bary[3])
{
	v3 f = V(from[0], from[1], from[2]), d = V(dir[0], dir[1], dir[2]);
	float dis = tmax < 0 ? INFINITY : tmax, z = 0, bu = 0, bv = 0; int ti = -1;
	int hit = use_tree ? kd_intersect(s, f, d, tmin, dis, &ti, &z, &bu, &bv, NULL) : brute_intersect(s, f, d, tmin, dis, &ti, &z, &bu, &bv);
	*tri = hit ? ti : -1; *t = hit ? z : 0.f;
	bary[0] = hit ? 1 - bu - bv : 0.f; bary[1] = hit ? bu : 0.f; bary[2] = hit ? bv : 0.f;
	return hit;
}
int yor_is_shadowed(const yor_scene *s, int use_tree, const float from[3], const float dir[3], float tmin, float tmax)
{
	v3 f = V(from[0], from[1], from[2]), d = V(dir[0], dir[1], dir[2]);
	v3 sfrom = vadd(f, vmul(d, tmin));
	float dis = tmax < 0 ? INFINITY : tmax - 2 * tmin;
	return use_tree ? kd_intersect_s(s, sfrom, d, dis, NULL) : brute_intersect_s(s, sfrom, d, dis);
}

/* ------------------------------------------------------------------ component entry points */
void yor_create_cs(const float n[3], float u[3], float v[3])
{
	v3 uu, vv; create_cs(V(n[0], n[1], n[2]), &uu, &vv);
	u[0] = uu.x; u[1] = uu.y; u[2] = uu.z; v[0] = vv.x; v[1] = vv.y; v[2] = vv.z;
}
void yor_sample_cos_hemisphere(const float n[3], const float ru[3], const float rv[3], float s1, float s2, float out[3])
{
	v3 r = sample_cos_hemisphere(V(n[0], n[1], n[2]), V(ru[0], ru[1], ru[2]), V(rv[0], rv[1], rv[2]), s1, s2);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int yor_bound_cross(const float a[3], const float g[3], const float from[3], const float dir[3], float dist, float *enter, float *leave)
{
	return bound_cross(V(a[0], a[1], a[2]), V(g[0], g[1], g[2]), V(from[0], from[1], from[2]), V(dir[0], dir[1], dir[2]), enter, leave, dist);
}
void yor_camera_shoot(const yor_camera_desc *cam, float px, float py, float out9[9])
{
	camera_t c; camera_configure(&c, cam);
	v3 f, d; float tmin, tmax, wt;
	camera_shoot(&c, px, py, &f, &d, &tmin, &tmax, &wt);
	out9[0] = f.x; out9[1] = f.y; out9[2] = f.z; out9[3] = d.x; out9[4] = d.y; out9[5] = d.z; out9[6] = tmin; out9[7] = tmax; out9[8] = wt;
}
void yor_camera_shoot_lens(const yor_camera_desc *cam, float px, float py, float lu, float lv, float out9[9])
{
	camera_t c; camera_configure(&c, cam);
	v3 f, d; float tmin, tmax, wt;
	camera_shoot_lens(&c, px, py, lu, lv, &f, &d, &tmin, &tmax, &wt);
	out9[0] = f.x; out9[1] = f.y; out9[2] = f.z; out9[3] = d.x; out9[4] = d.y; out9[5] = d.z; out9[6] = tmin; out9[7] = tmax; out9[8] = wt;
}
int yor_arealight_illum_sample(const yor_light_desc *ld, const float p[3], float s1, float s2, float out8[8])
{
	light_t l; light_configure(&l, ld);
	v3 dir = V(0, 0, 0); float tmax = 0, pdf = 0; rgb col = C(0, 0, 0);
	int ok = arealight_illum_sample(&l, V(p[0], p[1], p[2]), s1, s2, &dir, &tmax, &pdf, &col);
	if(!ok) { dir = V(0, 0, 0); tmax = 0; pdf = 0; col = C(0, 0, 0); }
	out8[0] = dir.x; out8[1] = dir.y; out8[2] = dir.z; out8[3] = tmax; out8[4] = pdf; out8[5] = col.r; out8[6] = col.g; out8[7] = col.b;
	return ok;
}
int yor_arealight_intersect(const yor_light_desc *ld, const float from[3], const float dir[3], float out5[5])
{
	light_t l; light_configure(&l, ld);
	float t = 0, ipdf = 0; rgb col = C(0, 0, 0);
	int ok = arealight_intersect(&l, V(from[0], from[1], from[2]), V(dir[0], dir[1], dir[2]), &t, &col, &ipdf);
	if(!ok) { t = 0; ipdf = 0; col = C(0, 0, 0); }
	out5[0] = t; out5[1] = ipdf; out5[2] = col.r; out5[3] = col.g; out5[4] = col.b;
	return ok;
}
int yor_pointlight_illuminate(const yor_light_desc *ld, const float p[3], float out7[7])
{
	light_t l; light_configure(&l, ld);
	v3 dir = V(0, 0, 0); float tmax = 0; rgb col = C(0, 0, 0);
	int ok = pointlight_illuminate(&l, V(p[0], p[1], p[2]), &col, &dir, &tmax);
	out7[0] = dir.x; out7[1] = dir.y; out7[2] = dir.z; out7[3] = tmax; out7[4] = col.r; out7[5] = col.g; out7[6] = col.b;
	return ok;
}
void yor_material_probe(const yor_material_desc *md, const float in14[14], int32_t sample_flags,
                        int32_t *bsdf_flags, float eval3[3], float *pdf, int32_t *sampled_flags, float sample8[8])
{
	mat_t m; mat_configure(&m, md);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.n = V(in14[0], in14[1], in14[2]); sp.ng = V(in14[3], in14[4], in14[5]);
	create_cs(sp.n, &sp.nu, &sp.nv);
	v3 wo = V(in14[6], in14[7], in14[8]), wl = V(in14[9], in14[10], in14[11]);
	bsdf_dat dat; unsigned flags;
	mat_init_bsdf(&m, &dat, &flags);
	*bsdf_flags = (int32_t)flags;
	rgb e = mat_eval(&m, &dat, &sp, wo, wl, BSDF_ALL);
	eval3[0] = e.r; eval3[1] = e.g; eval3[2] = e.b;
	*pdf = mat_pdf(&m, &dat, &sp, wo, wl, BSDF_GLOSSY | BSDF_DIFFUSE | BSDF_DISPERSIVE | BSDF_REFLECT | BSDF_TRANSMIT);
	sample_t s; s.s_1 = in14[12]; s.s_2 = in14[13]; s.pdf = 0.f; s.flags = (unsigned)sample_flags; s.sampled_flags = BSDF_NONE;
	v3 wi = V(0, 0, 0); float w = 0.f;
	rgb sc = mat_sample(&m, &dat, &sp, wo, &wi, &s, &w);
	*sampled_flags = (int32_t)s.sampled_flags;
	sample8[0] = sc.r; sample8[1] = sc.g; sample8[2] = sc.b; sample8[3] = wi.x; sample8[4] = wi.y; sample8[5] = wi.z; sample8[6] = s.pdf; sample8[7] = w;
}
void yor_material_sample_two(const yor_material_desc *md, const float in14[14], int32_t sample_flags, int32_t *sampled_flags, float out15[15])
{
	mat_t m; mat_configure(&m, md);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.n = V(in14[0], in14[1], in14[2]); sp.ng = V(in14[3], in14[4], in14[5]);
	create_cs(sp.n, &sp.nu, &sp.nv);
	const v3 wo = V(in14[6], in14[7], in14[8]);
	sample_t s; s.s_1 = in14[12]; s.s_2 = in14[13]; s.pdf = 0.f; s.flags = (unsigned)sample_flags; s.sampled_flags = BSDF_NONE;
	v3 dir[2] = {V(0, 0, 0), V(0, 0, 0)}; rgb tcol = C(0, 0, 0), ret = C(0, 0, 0); float w[2] = {0.f, 0.f};
	if(m.type == YOR_MAT_ROUGH_GLASS) ret = rough_glass_sample(&m, &sp, wo, &s, 1, dir, &tcol, w);
	*sampled_flags = (int32_t)s.sampled_flags;
	const float o[15] = {dir[0].x, dir[0].y, dir[0].z, ret.r, ret.g, ret.b, w[0], dir[1].x, dir[1].y, dir[1].z, tcol.r, tcol.g, tcol.b, w[1], s.pdf};
	memcpy(out15, o, sizeof o);
}
void yor_material_transparency(const yor_material_desc *md, const float in14[14], float out3[3])
{
	mat_t m; mat_configure(&m, md);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.n = V(in14[0], in14[1], in14[2]); sp.ng = V(in14[3], in14[4], in14[5]);
	rgb t = mat_transparency(&m, &sp, V(in14[6], in14[7], in14[8]));
	out3[0] = t.r; out3[1] = t.g; out3[2] = t.b;
}
void yor_material_specular(const yor_material_desc *md, const float in14[14], int32_t raylevel, int32_t *flags, float out12[12], float *alpha)
{
	mat_t m; mat_configure(&m, md);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.n = V(in14[0], in14[1], in14[2]); sp.ng = V(in14[3], in14[4], in14[5]);
	create_cs(sp.n, &sp.nu, &sp.nv);
	v3 wo = V(in14[6], in14[7], in14[8]);
	bsdf_dat dat; unsigned bf;
	mat_init_bsdf(&m, &dat, &bf);
	int refl = 0, refr = 0; v3 d[2] = {V(0, 0, 0), V(0, 0, 0)}; rgb c[2] = {C(0, 0, 0), C(0, 0, 0)};
	mat_get_specular(&m, &dat, &sp, wo, raylevel, &refl, &refr, d, c);
	*flags = (refl ? 1 : 0) | (refr ? 2 : 0);
	if(!refl) { d[0] = V(0, 0, 0); c[0] = C(0, 0, 0); }
	if(!refr) { d[1] = V(0, 0, 0); c[1] = C(0, 0, 0); }
	out12[0] = d[0].x; out12[1] = d[0].y; out12[2] = d[0].z; out12[3] = c[0].r; out12[4] = c[0].g; out12[5] = c[0].b;
	out12[6] = d[1].x; out12[7] = d[1].y; out12[8] = d[1].z; out12[9] = c[1].r; out12[10] = c[1].g; out12[11] = c[1].b;
	*alpha = mat_alpha(&m, &dat, &sp, wo);
}
void yor_beer_transmittance(const float acol[3], double dist, float tmax, int32_t *ok, float out3[3])
{
	const rgb t = beer_transmittance(beer_sigma(acol, dist), tmax);
	*ok = 1;
	out3[0] = t.r; out3[1] = t.g; out3[2] = t.b;
}

void yor_lightmat_emit(const yor_material_desc *md, const float n[3], const float wo[3], int include_lights, float out3[3])
{
	mat_t m; mat_configure(&m, md);
	sp_t sp; memset(&sp, 0, sizeof sp);
	sp.n = V(n[0], n[1], n[2]);
	rgb e = mat_emit(&m, &sp, V(wo[0], wo[1], wo[2]), include_lights);
	out3[0] = e.r; out3[1] = e.g; out3[2] = e.b;
}
