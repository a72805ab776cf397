// Device-side scalar/vector helpers of the MI355X path-tracing core (HIP, gfx950 only).
//
// Every function states the reference expression it evaluates (file:line under /root/reference)
// because parity with the CPU integrator is decided by discrete events (which triangle, which
// lobe, shadowed or not): the arithmetic must round the way the reference's expressions round.
// The translation unit is compiled with -ffp-contract=off; HIP's default correctly-rounded
// f32 divide/sqrt is relied upon.  Where the reference mixes a double literal into a float
// expression the double intermediate is kept (f64 is cheap on CDNA4's vector ALU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yafgpu {

#define YG_DEV __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;
constexpr double kPi2 = 1.57079632679489661923;
constexpr double k1Pi = 0.31830988618379067154;
constexpr double k2Pi = 6.28318530717958647692;   // util_math_optimizations.h:84
constexpr double k12Pi = 0.15915494309189533577;  // :86
constexpr double k4Pi = 1.27323954473516268615;   // :87
constexpr double k4Pi2 = 0.40528473456935108578;  // :88

struct V3 { float x, y, z; };
struct Col { float r, g, b; };

YG_DEV V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
YG_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
YG_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
YG_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
YG_DEV V3 operator*(V3 b, float f) { return mk(f * b.x, f * b.y, f * b.z); }            // vector.h:159-167
YG_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }           // vector.h:154
YG_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // :194
YG_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

YG_DEV Col mkc(float r, float g, float b) { Col c; c.r = r; c.g = g; c.b = b; return c; }
YG_DEV Col operator+(Col a, Col b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
YG_DEV Col operator*(Col a, Col b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
YG_DEV Col operator*(Col b, float f) { return mkc(f * b.r, f * b.g, f * b.b); }         // color.h:261-269
YG_DEV Col operator/(Col b, float f) { return mkc(b.r / f, b.g / f, b.b / f); }         // color.h:271
YG_DEV bool is_black(Col c) { return c.r == 0.f && c.g == 0.f && c.b == 0.f; }

YG_DEV float smax(float a, float b) { return a < b ? b : a; }   // std::max
YG_DEV float smin(float a, float b) { return b < a ? b : a; }   // std::min

// ---- util_math_optimizations.h (FAST_MATH / FAST_TRIG are ON in the reference's default build)
YG_DEV float f_exp2(float x) // :116-129
{
	x = smin(x, 129.00000f);
	x = smax(x, -126.99999f);
	const int ipart = (int)(x - 0.5f);
	const float p = (x - (float)ipart);
	const float expi = __int_as_float((int)((unsigned)(ipart + 127) << 23));
	const float poly = (p * (p * (p * (p * (p * 1.8775767e-3f + 8.9893397e-3f) + 5.5826318e-2f) + 2.4015361e-1f) + 6.9315308e-1f) + 9.9999994e-1f);
	return expi * poly;
}
YG_DEV float f_log2(float x) // :131-142 ; POLYLOG :94 turns double at the unsuffixed 2.5988452
{
	const int i = __float_as_int(x);
	const float e = (float)(((i & 0x7F800000) >> 23) - 127);
	const float m = __int_as_float((i & 0x7FFFFF) | 0x3F800000);
	const float a = m * -3.4436006e-2f + 3.1821337e-1f;
	const float b = m * a + -1.2315303f;
	const double c = (double)(m * b) + 2.5988452;
	const double d = (double)m * c + (double)-3.3241990f;
	const double ee = (double)m * d + (double)3.1157899f;
	return ((float)ee * (m - 1.0f) + e);
}
YG_DEV float f_pow(float a, float b) { return f_exp2(f_log2(a) * b); } // :176-183
// NB: __fsqrt_rn lowers to the bare 1-ulp v_sqrt_f32 on gfx950; sqrtf gets the correctly rounded fix-up sequence
YG_DEV float f_sqrt(float a) { return __builtin_sqrtf(a); }          // :203-210 -> sqrt
YG_DEV float f_sin(float x) // :222-244
{
	if((double)x > k2Pi || (double)x < -k2Pi) x -= ((int)(x * (float)k12Pi)) * (float)k2Pi;
	if((double)x < -kPi) x += (float)k2Pi;
	else if((double)x > kPi) x -= (float)k2Pi;
	x = ((float)k4Pi * x) - ((float)k4Pi2 * x * fabsf(x));
	const float result = 0.225f * (x * fabsf(x) - x) + x;
	if(result <= -1.0f) return -1.0f;
	else if(result >= 1.0f) return 1.0f;
	else return result;
}
YG_DEV float f_cos(float x) { return f_sin(x + (float)kPi2); } // :246-253

YG_DEV V3 normalize(V3 v) // vector.h:227-238 ; (float)(1.0/sqrt) == 1.f/sqrt for a single correctly rounded divide
{
	float len = v.x * v.x + v.y * v.y + v.z * v.z;
	if(len != 0.f)
	{
		len = 1.0f / f_sqrt(len);
		v.x *= len; v.y *= len; v.z *= len;
	}
	return v;
}
YG_DEV float length(V3 v) { return f_sqrt(v.x * v.x + v.y * v.y + v.z * v.z); } // vector.h:222

YG_DEV void create_cs(V3 n, V3 &u, V3 &v) // vector.h:319-337
{
	if((n.x == 0.f) && (n.y == 0.f))
	{
		u = (n.z < 0.f) ? mk(-1.f, 0.f, 0.f) : mk(1.f, 0.f, 0.f);
		v = mk(0.f, 1.f, 0.f);
	}
	else
	{
		const float d = 1.0f / f_sqrt(n.y * n.y + n.x * n.x);
		u = mk(n.y * d, -n.x * d, 0.f);
		v = cross(n, u);
	}
}
YG_DEV V3 reflect_dir(V3 n, V3 v) // vector.h:273-278
{
	const float vn = dot(v, n);
	if(vn < 0.f) return -v;
	return n * (2.f * vn) - v;
}

// ---- util_mcqmc.h
constexpr double kMultRatio = 0.00000000023283064365386962890625; // :91
YG_DEV float clamp01(float v) { return smax(0.f, smin(1.f, v)); }
YG_DEV float ri_vdc(uint32_t bits, uint32_t r) // :93-101
{
	bits = __brev(bits);
	return clamp01((float)((double)(bits ^ r) * kMultRatio));
}
YG_DEV float ri_s(uint32_t i, uint32_t r) // :103-108
{
	for(uint32_t v = 1u << 31; i; i >>= 1, v ^= v >> 1)
		if(i & 1u) r ^= v;
	return clamp01((float)((double)r * kMultRatio));
}
YG_DEV float ri_lp(uint32_t i, uint32_t r) // :110-115
{
	for(uint32_t v = 1u << 31; i; i >>= 1, v |= v >> 1)
		if(i & 1u) r ^= v;
	return clamp01((float)((double)r * kMultRatio));
}
YG_DEV uint32_t fnv32a(uint32_t value) // :147-163
{
	uint32_t hash = 0x811c9dc5u;
#pragma unroll
	for(int i = 0; i < 4; ++i)
	{
		hash ^= (value >> (8 * i)) & 0xffu;
		hash *= 0x01000193u;
	}
	return hash;
}

struct Halton // :28-87, value kept in double as the reference demands
{
	uint32_t base; double inv_base, value;
	YG_DEV void init(uint32_t b) { base = b; inv_base = 1.0 / (double)b; value = 0.0; }
	YG_DEV void set_start(uint32_t i)
	{
		double factor = inv_base;
		value = 0.0;
		while(i > 0)
		{
			value += (double)(i % base) * factor;
			i /= base;
			factor *= inv_base;
		}
	}
	YG_DEV float next()
	{
		const double r = 0.9999999999 - value;
		if(inv_base < r) value += inv_base;
		else
		{
			double hh = 0.0, h = inv_base;
			while(h >= r) { hh = h; h *= inv_base; }
			value += hh + h - 1.0;
		}
		return smax(0.f, smin(1.f, (float)value));
	}
};

struct Mwc // Random, :173-192
{
	uint32_t x, c;
	YG_DEV void init(uint32_t seed) { x = 30903u; c = seed; }
	YG_DEV double next()
	{
		const uint32_t ya = 1791398085u, yah = ya >> 16, yal = ya & 65535u;
		const uint32_t xh = x >> 16, xl = x & 65535u;
		x = x * ya + c;
		c = xh * yah + ((xh * yal) >> 16) + ((xl * yah) >> 16);
		if(xl * yal >= ~c + 1u) c++;
		return (double)x * kMultRatio;
	}
};

YG_DEV float add_mod1(float a, float b) { const float s = a + b; return s > 1.f ? s - 1.f : s; } // util_sample.h:183-187

YG_DEV V3 sample_cos_hemisphere(V3 n, V3 ru, V3 rv, float s_1, float s_2) // util_sample.h:45-55
{
	if(s_1 >= 1.0f) return n;
	const float z_2 = (float)((double)s_2 * k2Pi);
	const V3 a = ru * f_cos(z_2) + rv * f_sin(z_2);
	return a * f_sqrt(1.0f - s_1) + n * f_sqrt(s_1);
}

} // namespace yafgpu
