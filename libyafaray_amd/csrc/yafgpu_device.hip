// MI355X (gfx950) path-tracing core: flattened kd-tree traversal, surface shading with MIS direct
// lighting, and deterministic film accumulation — hand-written HIP, one persistent launch per pass.
//
// Reference path: TiledIntegrator::renderTile (integrator_tiled.cc:309-521) ->
// PathIntegrator::integrate (integrator_path_tracer.cc:112-347) -> Scene::intersect / isShadowed
// (scene.cc:896-994) -> TriKdTree::intersect / intersectS (kdtree_triangle.cc:684-977) ->
// Triangle::intersect (triangle.h:223-259), MonteCarloIntegrator::doLightEstimation
// (integrator_montecarlo.cc:78-345), ImageFilm::addSample (imagefilm.cc:925-1015).
//
// Execution model (DESIGN.md has the full picture):
//   * one lane = one camera sample; a wave owns P pixels x L lanes (L = min(spp,64)) and walks their
//     samples in index order, so each pixel's film sum is a sequential sum in sample order with no
//     atomics (ImageFilm::addSample's order for a single-threaded reference render);
//   * persistent waves pull "units" (pixel groups of one tile) from 8 queues, one per XCD, and
//     steal from the others when their own runs dry, so the waves of one XCD share an image region
//     and hence a kd-tree working set in that XCD's L2;
//   * integrate() is an explicit state machine with exactly one closest-hit trace site and one
//     any-hit trace site, so the traversal loops exist once in the instruction stream;
//   * traversal keeps a short per-lane stack in LDS ([slot][lane], conflict-free), with the
//     classic kd-restart fallback when more than kStack far-children are pending.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>

#include "../../include/yafgpu.h"
#include "kdtree_build.h"
#include "yafgpu_math.h"
#include "yafgpu_shading.h"
#include "yafgpu_texture.h"

namespace yafgpu {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
#ifndef YAFGPU_STACK
#define YAFGPU_STACK 8                // 8 slots: 8 waves/SIMD for the traversal kernels, 0.8 % of rays restart (C2)
#endif
#ifndef YAFGPU_WAVES
#define YAFGPU_WAVES 4                // __launch_bounds__ min waves/SIMD of the render kernel (C2 measured: 1:368 2:667 3:781 4:793 Mrays/s)
#endif
constexpr int kStack = YAFGPU_STACK;  // per-lane LDS stack slots (power of two)
// Top of the tree apart (north_star: "LDS-staged node tiles"; measured in profiles/r03_ab_toptree.txt).  0: off.  1: the first kTopDepth levels
// are walked in a heap-ordered copy (DevScene::top) that stays resident in the vector L1.  2: that copy is staged into LDS by every workgroup.
#ifndef YAFGPU_TRACE_TOP
#define YAFGPU_TRACE_TOP 0
#endif
#ifndef YAFGPU_TOP_DEPTH
#define YAFGPU_TOP_DEPTH 10
#endif
constexpr int kTopDepth = YAFGPU_TOP_DEPTH, kTopN = (1 << kTopDepth) - 1, kTopBottom = (1 << (kTopDepth - 1)) - 1;      // entries; first entry of the last level
constexpr uint32_t kTopTag = 0x80000000u;      // a node id with this bit is a heap index into the top copy
constexpr int kDepthCap = 48;         // host tree depth cap; deeper pending lists restart
constexpr int kQueues = 8;            // one per XCD
constexpr float kMinRayDist = (float)0.00005;   // MIN_RAYDIST, CMakeLists.txt:46-48
constexpr float kShadowBias = (float)0.0005;    // YAF_SHADOW_BIAS, CMakeLists.txt:50-52

// ------------------------------------------------------------------------------------------------
// device scene
struct DevScene
{
	const uint2 *nodes;          // 8-byte kd nodes (kdtree_build.h)
	const uint4 *nodes2;         // (node i, a copy of its right child): the pair layout of the traversal kernels (YAFGPU_TRACE_PAIR), or nullptr
	const uint2 *nodes_blk;      // the same tree in 64-B blocks of three levels (YAFGPU_TRACE_BLOCKS): node id = block * 8 + slot, slot s < 3 has its
	                             // children in slots 2s + 1, 2s + 2; a node in slots 3..6 keeps (in the child field) the block of its left child, the right
	                             // child's block is the next one; both children sit in their blocks' slot 0.  Or nullptr.
	const uint4 *top;            // YAFGPU_TRACE_TOP: the tree's first kTopDepth levels in heap order (children of entry h at 2h + 1, 2h + 2):
	                             // (split, flags as in `nodes`, the node's own index in `nodes`, 0); entries no node maps to are empty leaves.  Or nullptr.
	const uint32_t *refs;        // leaf references
	const float4 *tri;           // 3 x float4 per triangle: (a, eps) (e1, mat|vis<<30) (e2, 0)
	const float4 *tri_ng;        // geometric normal + smooth flag
	const float4 *tri_vn;        // 3 x float4 per triangle (vertex normals) or nullptr
	const yafgpu_material *mats;
	const yafgpu_light *lights;
	const int *faure;            // concatenated Faure permutations
	const int *faure_off;        // [50] offsets into faure
	const double *inv_prims;     // [50]
	int n_lights, n_tris, n_mats, n_faure;      // n_faure: ints in the concatenated Faure permutations
	uint32_t n_nodes;
	float blo[3], bhi[3];
	yafgpu_camera cam;
	TexScene tex;                // textures, texels, shader nodes, per-triangle texture coordinates (nodes == nullptr: none)
};

struct RenderArgs
{
	DevScene sc;
	yafgpu_render_params rp;
	float shadow_bias, ray_min_dist, filterw;
	float table_scale; int wide_filter;   // wide_filter: footprint beyond the 2x2 box case -> table weights + atomics (wavefront accumulate)
	const float *filter_table;            // 16x16 reconstruction-filter table (ImageFilm ctor, imagefilm.cc:152-176)
	int lanes_per_pixel, pixels_per_wave, iters;
	int n_tiles; uint32_t n_units;
	const int4 *tile_rect;         // x0,y0,w,h per tile of this shard
	const uint32_t *unit_prefix;   // n_tiles+1
	uint32_t *queue_next;          // kQueues counters, 32 words apart
	uint32_t queue_begin[kQueues + 1];
	float *planes;
	yafgpu_counters *counters;
};

struct LaneCounters { uint32_t closest, shadow, interior, leaves, tests, samples, restarts; };

__constant__ int c_prims[50] = {1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67,
                                71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167,
                                173, 179, 181, 191, 193, 197, 199, 211, 223, 227};

// scrHalton__, include/common/scr_halton.h:52-75
YG_DEV double scr_halton(const DevScene &sc, int dim, uint32_t n)
{
	double value = 0.0;
	const int *sigma = sc.faure + sc.faure_off[dim];
	const uint32_t base = (uint32_t)c_prims[dim];
	double f, factor, dn = (double)n;
	f = factor = sc.inv_prims[dim];
	while(n > 0)
	{
		value += (double)(sigma[n % base]) * factor;
		dn *= f;
		n = (uint32_t)dn;
		factor *= f;
	}
	return fmax(1.0e-36, fmin(1.0, value));
}

// Bound::cross, include/common/bound.h:144-212
// (inv_dir = 1/dir per component, computed once by the caller: the traversal needs the same three quotients)
YG_DEV bool bound_cross(const DevScene &sc, V3 from, V3 dir, V3 inv_dir, float dist, float &enter, float &leave)
{
	const V3 a0 = mk(sc.blo[0], sc.blo[1], sc.blo[2]), a1 = mk(sc.bhi[0], sc.bhi[1], sc.bhi[2]);
	const V3 p = from - a0;
	float lmin = -1e38f, lmax = 1e38f, ltmin, ltmax;
	if(dir.x != 0.f)
	{
		const float inv = inv_dir.x;
		if(inv > 0.f) { lmin = -p.x * inv; lmax = ((a1.x - a0.x) - p.x) * inv; }
		else { lmin = ((a1.x - a0.x) - p.x) * inv; lmax = -p.x * inv; }
		if((lmax < 0.f) || (lmin > dist)) return false;
	}
	if(dir.y != 0.f)
	{
		const float inv = inv_dir.y;
		if(inv > 0.f) { ltmin = -p.y * inv; ltmax = ((a1.y - a0.y) - p.y) * inv; }
		else { ltmin = ((a1.y - a0.y) - p.y) * inv; ltmax = -p.y * inv; }
		lmin = smax(ltmin, lmin);
		lmax = smin(ltmax, lmax);
		if((lmax < 0.f) || (lmin > dist)) return false;
	}
	if(dir.z != 0.f)
	{
		const float inv = inv_dir.z;
		if(inv > 0.f) { ltmin = -p.z * inv; ltmax = ((a1.z - a0.z) - p.z) * inv; }
		else { ltmin = ((a1.z - a0.z) - p.z) * inv; ltmax = -p.z * inv; }
		lmin = smax(ltmin, lmin);
		lmax = smin(ltmax, lmax);
		if((lmax < 0.f) || (lmin > dist)) return false;
	}
	if((lmin <= lmax) && (lmax >= 0.f) && (lmin <= dist)) { enter = lmin; leave = lmax; return true; }
	return false;
}

// Triangle::intersect, include/common/triangle.h:223-259, on the 48-byte record
YG_DEV bool tri_test(const float4 r0, const float4 r1, const float4 r2, V3 from, V3 dir, float &t, float &u, float &v)
{
	const V3 a = mk(r0.x, r0.y, r0.z), e1 = mk(r1.x, r1.y, r1.z), e2 = mk(r2.x, r2.y, r2.z);
	const float eps = r0.w;
	const V3 pvec = cross(dir, e2);
	const float det = dot(e1, pvec);
	if(det > -eps && det < eps) return false;
	const float inv_det = 1.f / det;
	const V3 tvec = from - a;
	u = dot(tvec, pvec) * inv_det;
	if(u < 0.f || u > 1.f) return false;
	const V3 qvec = cross(tvec, e1);
	v = dot(dir, qvec) * inv_det;
	if((v < 0.f) || ((u + v) > 1.f)) return false;
	t = dot(e2, qvec) * inv_det;
	if(t < eps) return false;
	return true;
}

// the same test, straight-line: identical operations and roundings, the four rejections combined at the end
// (a rejected lane may compute with inf / NaN on the way; its result is discarded)
YG_DEV bool tri_test_flat(const float4 r0, const float4 r1, const float4 r2, V3 from, V3 dir, float &t, float &u, float &v)
{
	const V3 a = mk(r0.x, r0.y, r0.z), e1 = mk(r1.x, r1.y, r1.z), e2 = mk(r2.x, r2.y, r2.z);
	const float eps = r0.w;
	const V3 pvec = cross(dir, e2);
	const float det = dot(e1, pvec);
	const bool det_ok = !(det > -eps && det < eps);
	const float inv_det = 1.f / det;
	const V3 tvec = from - a;
	u = dot(tvec, pvec) * inv_det;
	const bool u_ok = !(u < 0.f || u > 1.f);
	const V3 qvec = cross(tvec, e1);
	v = dot(dir, qvec) * inv_det;
	const bool v_ok = !((v < 0.f) || ((u + v) > 1.f));
	t = dot(e2, qvec) * inv_det;
	return det_ok && u_ok && v_ok && !(t < eps);
}

// per-lane stack in LDS: column `lane` of a [kStack][64] array of (node, tmax)
struct LaneStack
{
	uint2 *col;     // &stack[0][lane]; slot s lives at col[s * kWave]
	int sp, lo;     // entries [lo, sp) are live (ring of kStack); lo > 0: older pending far-children were overwritten
	YG_DEV void reset() { sp = 0; lo = 0; }
	YG_DEV bool empty() const { return sp == lo; }
	YG_DEV bool lost() const { return lo > 0; }      // a restart will recover what was overwritten
	YG_DEV void push(uint32_t node, float tmax)
	{
		col[(sp & (kStack - 1)) * kWave] = make_uint2(node, __float_as_uint(tmax));
		++sp;
		lo = max(lo, sp - kStack);
	}
	YG_DEV void pop(uint32_t &node, float &tmax)
	{
		--sp;
		const uint2 e = col[(sp & (kStack - 1)) * kWave];
		node = e.x; tmax = __uint_as_float(e.y);
	}
};

// Where a kd-restart resumes.  Normally at the exit of the cell just left.  If that cell had zero length (tmax == tmin)
// the walk may not have advanced at all since the previous restart: a tree with more split planes crossed at one
// distance than the short stack has slots (a degenerate chain of identical planes) would then restart at the same
// distance for ever.  Stepping to the next representable distance guarantees progress; what it can skip is a hit at
// exactly that distance in a zero-length cell of such a tree.
YG_DEV float restart_from(float tmin, float tmax)
{
	return (tmax > tmin) ? tmax : tmax + fmaxf(fabsf(tmax) * 1.2e-7f, 1e-30f);
}

// Closest hit (kAny == false): TriKdTree::intersect, kdtree_triangle.cc:684-837 — the nearest
//   triangle with ray_tmin <= t < dist whose material is visible to camera rays (:786).
// Any hit (kAny == true): TriKdTree::intersectS, :840-977 — any triangle with 0 <= t < dist whose
//   material casts shadows (:938).
// Both walk [t_enter, t_exit] of the ray against the tree bound front to back; a triangle is
// referenced by every leaf its bounds overlap, so the result does not depend on tree topology.
template<bool kAny, bool kStats>
YG_DEV bool kd_trace(const DevScene &sc, LaneStack &stk, V3 from, V3 dir, float ray_tmin, float dist,
                     int &tri_out, float &t_out, float &bu, float &bv, LaneCounters &cn)
{
	float a, b;
	if(sc.n_nodes == 0u) return false;
	const V3 inv_dir = mk(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
	if(!bound_cross(sc, from, dir, inv_dir, dist, a, b)) return false;
	const float t_exit = b;
	float tmin = smax(a, 0.f), tmax = t_exit;
	float z = dist;
	bool hit = false;
	uint32_t node = 0u;
	stk.reset();
	for(;;)
	{
		if(z < tmin) break;       // kdtree_triangle.cc:717 (dist < entry distance)
		uint2 nd = sc.nodes[node];
		while((nd.y & 3u) != 3u)
		{
			const int axis = (int)(nd.y & 3u);
			const float split = __uint_as_float(nd.x);
			const float o = comp(from, axis), d = comp(dir, axis);
			const float tplane = (split - o) * comp(inv_dir, axis);
			const bool below = (o < split) || (o == split && d <= 0.f);
			const uint32_t left = node + 1u, right = nd.y >> 2;
			const uint32_t near_c = below ? left : right, far_c = below ? right : left;
			if(kStats) ++cn.interior;
			if(!(tplane <= tmax) || tplane <= 0.f) node = near_c;       // plane beyond the cell or behind the origin (also NaN)
			else if(tplane < tmin) node = far_c;
			else { stk.push(far_c, tmax); node = near_c; tmax = tplane; }
			nd = sc.nodes[node];
		}
		{
			const uint32_t np = nd.y >> 2, first = nd.x;
			if(kStats) ++cn.leaves;
			for(uint32_t i = 0; i < np; ++i)
			{
				const uint32_t ti = sc.refs[first + i];
				const float4 r0 = sc.tri[3u * ti], r1 = sc.tri[3u * ti + 1u], r2 = sc.tri[3u * ti + 2u];
				float t, u, v;
				if(kStats) ++cn.tests;
				if(tri_test(r0, r1, r2, from, dir, t, u, v))
				{
					const uint32_t vis = __float_as_uint(r1.w) >> 30;
					if(kAny)
					{
						if(t < dist && t >= 0.f && (vis == 0u || vis == 2u)) return true;
					}
					else if(t < z && t >= ray_tmin && (vis == 0u || vis == 1u))
					{
						z = t; tri_out = (int)ti; bu = u; bv = v; hit = true;
					}
				}
			}
		}
		if(!kAny && hit && z <= tmax) break;   // :822
		if(stk.empty())
		{
			if(!stk.lost() || tmax >= t_exit) break;
			// kd-restart: pending far-children were lost to the short stack; resume at the cell exit
			tmin = restart_from(tmin, tmax); tmax = t_exit; node = 0u; stk.reset();
			if(kStats) ++cn.restarts;
			continue;
		}
		tmin = tmax;
		stk.pop(node, tmax);
	}
	t_out = z;
	return hit;
}

// Triangle::getSurface, src/common/triangle.cc:30-133 (no UV / orco)
YG_DEV void get_surface(const DevScene &sc, int ti, V3 hitp, float bu, float bv, SurfPt &sp)
{
	const float4 g = sc.tri_ng[ti];
	sp.ng = mk(g.x, g.y, g.z);
	if(sc.tri_vn != nullptr && __float_as_uint(g.w) != 0u)
	{
		const float u = 1.f - bu - bv, v = bu, w = bv;   // b_0, b_1, b_2 (triangle.h:252-254, triangle.cc:34)
		const float4 na = sc.tri_vn[3 * ti], nb = sc.tri_vn[3 * ti + 1], nc = sc.tri_vn[3 * ti + 2];
		sp.n = normalize(mk(na.x, na.y, na.z) * u + mk(nb.x, nb.y, nb.z) * v + mk(nc.x, nc.y, nc.z) * w);
	}
	else sp.n = sp.ng;
	sp.mat = (int)(__float_as_uint(sc.tri[3 * ti + 1].w) & 0x3FFFFFFFu);
	sp.p = hitp;
	create_cs(sp.n, sp.nu, sp.nv);
}

// Transparent shadows: TriKdTree::intersectTs, kdtree_triangle.cc:983-1162.  Every triangle with ray_tmin <= t < dist
// whose material casts shadows either blocks the ray (opaque), or — once per triangle, the reference's std::set
// `filtered` — multiplies its transparency into filt; more than max_depth transparent triangles block it too.  The
// product is taken in visiting order, as there (so its last bits depend on tree topology, there as here).
constexpr int kTsMaxDepth = 8;
YG_DEV bool kd_trace_ts(const DevScene &sc, LaneStack &stk, uint32_t *seen /* [kTsMaxDepth + 1] */, V3 from, V3 dir, float ray_tmin, float dist,
                        int max_depth, Col &filt)
{
	filt = mkc(1.f, 1.f, 1.f);
	float a, b;
	if(sc.n_nodes == 0u) return false;
	const V3 inv_dir = mk(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
	if(!bound_cross(sc, from, dir, inv_dir, dist, a, b)) return false;
	const float t_exit = b;
	float tmin = smax(a, 0.f), tmax = t_exit;
	uint32_t node = 0u;
	int depth = 0, n_seen = 0;
	stk.reset();
	for(;;)
	{
		if(dist < tmin) break;
		uint2 nd = sc.nodes[node];
		while((nd.y & 3u) != 3u)
		{
			const int axis = (int)(nd.y & 3u);
			const float split = __uint_as_float(nd.x);
			const float o = comp(from, axis), d = comp(dir, axis);
			const float tplane = (split - o) * comp(inv_dir, axis);
			const bool below = (o < split) || (o == split && d <= 0.f);
			const uint32_t left = node + 1u, right = nd.y >> 2;
			const uint32_t near_c = below ? left : right, far_c = below ? right : left;
			if(!(tplane <= tmax) || tplane <= 0.f) node = near_c;
			else if(tplane < tmin) node = far_c;
			else { stk.push(far_c, tmax); node = near_c; tmax = tplane; }
			nd = sc.nodes[node];
		}
		const uint32_t np = nd.y >> 2, first = nd.x;
		for(uint32_t i = 0; i < np; ++i)
		{
			const uint32_t ti = sc.refs[first + i];
			const float4 r0 = sc.tri[3u * ti], r1 = sc.tri[3u * ti + 1u], r2 = sc.tri[3u * ti + 2u];
			float t, u, v;
			if(!tri_test(r0, r1, r2, from, dir, t, u, v)) continue;
			const uint32_t vis = __float_as_uint(r1.w) >> 30;
			if(!(t < dist && t >= ray_tmin && (vis == 0u || vis == 2u))) continue;
			const yafgpu_material &m = sc.mats[__float_as_uint(r1.w) & 0x3FFFFFFFu];
			if(!mat_is_transparent(m)) return true;
			bool known = false;
			for(int k = 0; k < n_seen; ++k) known = known || (seen[k * kWave] == ti);
			if(known) continue;
			if(depth >= max_depth) return true;
			if(n_seen <= kTsMaxDepth) seen[(n_seen++) * kWave] = ti;
			SurfPt sp;
			get_surface(sc, (int)ti, from + dir * t, u, v, sp);
			if(m.n_nodes > 0 && sc.tex.nodes != nullptr)
			{	// getTransparency reads the diffuse shader and the component nodes (material_shiny_diffuse.cc:541-563)
				TexPoint tp; tex_point(sc.tex, (int)ti, u, v, sp.p, sp.n, sp.ng, tp);
				yafgpu_material tmp; mat_resolve(sc.tex, sc.cam, m, tp, tmp);
				filt = filt * mat_transparency(tmp, sp, dir);
			}
			else filt = filt * mat_transparency(m, sp, dir);
			++depth;
		}
		if(stk.empty())
		{
			if(!stk.lost() || tmax >= t_exit) break;
			tmin = restart_from(tmin, tmax); tmax = t_exit; node = 0u; stk.reset();
			continue;
		}
		tmin = tmax;
		stk.pop(node, tmax);
	}
	return false;
}

// ------------------------------------------------------------------------------------------------
// direct lighting: MonteCarloIntegrator::doLightEstimation, integrator_montecarlo.cc:78-345,
// over lights [l_begin, l_end).  One any-hit trace site serves the Dirac branch (:94-148), the
// light-sampling half (:161-262) and the BSDF-sampling half (:285-333) of the area-light MIS.
template<bool kStats>
YG_DEV Col direct_light(const RenderArgs &ra, LaneStack &stk, const SurfPt &sp, const yafgpu_material &mat, const BsdfDat &dat, V3 wo,
                        int l_begin, int l_end, uint32_t pixel_sample, uint32_t sampling_offs, LaneCounters &cn)
{
	const DevScene &sc = ra.sc;
	Col total = mkc(0.f, 0.f, 0.f);
	const uint32_t kMisFlags = kGlossy | kDiffuse | kDispersive | kReflect | kTransmit;
	for(int li = l_begin; li < l_end; ++li)
	{
		const yafgpu_light &light = sc.lights[li];
		const bool cast_shadows = light.cast_shadows && mat.receive_shadows;
		const bool dirac = light.type == YAFGPU_LIGHT_POINT;
		const int n = dirac ? 1 : (int)ceilf((float)light.samples * ra.rp.aa_light_sample_multiplier);
		const float inv_ns = 1.f / (float)n;
		const uint32_t offs = (uint32_t)n * pixel_sample + sampling_offs + (uint32_t)li * 4567u; // LOFFS_DELTA :45
		Col ccol = mkc(0.f, 0.f, 0.f), ccol_2 = mkc(0.f, 0.f, 0.f), col_dirac = mkc(0.f, 0.f, 0.f);
		Halton hal_2, hal_3;
		hal_2.init(2u); hal_3.init(3u);
		const int n_phase = dirac ? 1 : 2;
		for(int phase = 0; phase < n_phase; ++phase)
		{
			hal_2.set_start(offs - 1u);
			hal_3.set_start(offs - 1u);
			for(int i = 0; i < n; ++i)
			{
				V3 r_dir = mk(0.f, 0.f, 0.f);
				float r_tmin = 0.f, r_tmax = -1.f;
				Col contrib = mkc(0.f, 0.f, 0.f);
				bool want = false;
				if(dirac)
				{
					Col lcol;
					if(pointlight_illuminate(light, sp.p, lcol, r_dir, r_tmax))
					{
						r_tmin = ra.rp.shadow_bias_auto ? ra.shadow_bias * smax(1.f, length(sp.p)) : ra.shadow_bias;
						const float angle = mat.flat ? 1.f : fabsf(dot(sp.n, r_dir));
						contrib = (mat_eval(mat, dat, sp, wo, r_dir, kAll) * lcol) * angle;
						want = true;
					}
				}
				else
				{
					const float s_1 = hal_2.next(), s_2 = hal_3.next();
					if(phase == 0)
					{
						float ls_pdf;
						if(arealight_illum_sample(light, sp.p, s_1, s_2, r_dir, r_tmax, ls_pdf))
						{
							r_tmin = ra.rp.shadow_bias_auto ? ra.shadow_bias * smax(1.f, length(sp.p)) : ra.shadow_bias;
							if(ls_pdf > 1e-6f)
							{
								const Col surf_col = mat_eval(mat, dat, sp, wo, r_dir, kAll);
								const float angle = mat.flat ? 1.f : fabsf(dot(sp.n, r_dir));
								const float m_pdf = mat_pdf(mat, dat, sp, wo, r_dir, kMisFlags);
								const Col ls_col = col3(light.color);
								if(m_pdf > 1e-6f)
								{
									const float l_2 = ls_pdf * ls_pdf, m_2 = m_pdf * m_pdf;
									const float w = l_2 / (l_2 + m_2);
									contrib = (((surf_col * ls_col) * angle) * w) / ls_pdf;
								}
								else contrib = ((surf_col * ls_col) * angle) / ls_pdf;
							}
							// the reference traces the shadow ray before it looks at the pdf (:170-179)
							want = true;
						}
					}
					else
					{
						r_tmin = ra.rp.min_raydist_auto ? ra.ray_min_dist * smax(1.f, length(sp.p)) : ra.ray_min_dist;
						float W = 0.f;
						BsdfSample bs; bs.s_1 = s_1; bs.s_2 = s_2; bs.pdf = 0.f; bs.flags = kMisFlags; bs.sampled = kNone;
						const Col surf_col = mat_sample(mat, dat, sp, wo, r_dir, bs, W);
						float light_ipdf;
						if(bs.pdf > 1e-6f && arealight_intersect(light, sp.p, r_dir, r_tmax, light_ipdf))
						{
							if(light_ipdf > 1e-6f)
							{
								const float l_pdf = 1.f / light_ipdf;
								const float l_2 = l_pdf * l_pdf, m_2 = bs.pdf * bs.pdf;
								const float w = m_2 / (l_2 + m_2);
								contrib = ((surf_col * col3(light.color)) * w) * W;
							}
							want = true;
						}
					}
				}
				if(want)
				{
					bool shadowed = false;
					if(cast_shadows)
					{	// Scene::isShadowed, scene.cc:962-994
						const V3 sfrom = sp.p + r_dir * r_tmin;
						const float dis = (r_tmax < 0.f) ? INFINITY : r_tmax - 2.f * r_tmin;
						int ti; float tt, uu, vv;
						++cn.shadow;
						shadowed = kd_trace<true, kStats>(sc, stk, sfrom, r_dir, 0.f, dis, ti, tt, uu, vv, cn);
					}
					if(!shadowed)
					{
						if(dirac) col_dirac = col_dirac + contrib;
						else if(phase == 0) ccol = ccol + contrib;
						else ccol_2 = ccol_2 + contrib;
					}
				}
			}
		}
		Col col = mkc(0.f, 0.f, 0.f);
		if(dirac) col = col + col_dirac;
		else { col = col + ccol * inv_ns; col = col + ccol_2 * inv_ns; }
		total = total + col;
	}
	return total;
}

// ------------------------------------------------------------------------------------------------
// PathIntegrator::integrate, integrator_path_tracer.cc:112-347, as a per-lane state machine.
// Supported lobes: diffuse/translucent shinydiffuse, glossy(as_diffuse), light_mat — the host
// rejects materials that would need recursiveRaytrace (integrator_montecarlo.cc:782-1028).
enum : int { kStPrimary = 0, kStFirst = 1, kStDepth = 2 };

template<bool kStats>
YG_DEV void integrate(const RenderArgs &ra, LaneStack &stk, V3 from, V3 dir, float tmin, float tmax,
                      uint32_t pixel_sample, uint32_t sampling_offs, uint32_t sample_ordinal, float out[4], LaneCounters &cn)
{
	const DevScene &sc = ra.sc;
	const yafgpu_render_params &rp = ra.rp;
	Col col = mkc(0.f, 0.f, 0.f);
	float alpha = rp.bg_transp ? 0.f : 1.f;

	int stage = kStPrimary;
	SurfPt sp0, hit;                   // camera hit, current path vertex
	BsdfDat dat0, dat_n;
	V3 wo0 = mk(0.f, 0.f, 0.f), pwo = mk(0.f, 0.f, 0.f);
	int mat0 = 0;
	uint32_t bsdfs0 = 0u;
	Col path_col = mkc(0.f, 0.f, 0.f), throughput = mkc(1.f, 1.f, 1.f);
	int path_i = 0, depth = 0;
	float last_w = 0.f;          // integrate()'s `w`: Material::sample may leave it untouched (see st_extend)
	const int n_paths = rp.path_samples > 1 ? rp.path_samples : 1;
	uint32_t offs = 0u;
	uint32_t sampled_flags = kNone;
	uint32_t one_light_calls = 0u;
	Mwc rr; rr.init(fnv32a(sample_ordinal) + 123u);   // see DESIGN.md: Russian-roulette stream (row N4)

	V3 r_from = from, r_dir = dir;
	float r_tmin = tmin, r_tmax = tmax;
	bool running = true;
	while(running)
	{
		// ---- the one closest-hit trace site: Scene::intersect, scene.cc:896-927
		int ti = -1; float z = 0.f, bu = 0.f, bv = 0.f;
		const float dis = (r_tmax < 0.f) ? INFINITY : r_tmax;
		++cn.closest;
		const bool got = kd_trace<false, kStats>(sc, stk, r_from, r_dir, r_tmin, dis, ti, z, bu, bv, cn);
		bool start_path = false;     // begin path sample `path_i` from the camera hit
		bool extend = false;         // sample the current vertex and continue the path
		if(stage == kStPrimary)
		{
			if(!got)
			{
				if(rp.has_background && !rp.bg_transp_refract) col = col + mkc(rp.background[0], rp.background[1], rp.background[2]);
				break;
			}
			get_surface(sc, ti, r_from + r_dir * z, bu, bv, sp0);
			const yafgpu_material &m = sc.mats[sp0.mat];
			mat0 = sp0.mat;
			bsdfs0 = mat_init_bsdf(m, dat0);
			wo0 = -r_dir;
			if(bsdfs0 & kEmit) col = col + mat_emit(m, sp0, wo0, true);                     // :152, include_lights_ = true (:133)
			if(bsdfs0 & kDiffuse) col = col + direct_light<kStats>(ra, stk, sp0, m, dat0, wo0, 0, sc.n_lights, pixel_sample, sampling_offs, cn); // :156
			alpha = 1.f;
			if(rp.bg_transp_refract)
			{
				const float m_alpha = mat_alpha(m, dat0, sp0, wo0);
				alpha = m_alpha + (1.f - m_alpha) * (rp.bg_transp ? 0.f : 1.f);
			}
			const uint32_t path_flags = rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse;
			if(rp.integrator != YAFGPU_INTEGRATOR_PATH || !(bsdfs0 & path_flags)) break;
			path_i = 0;
			start_path = true;
		}
		else if(stage == kStFirst)
		{
			if(!got) { ++path_i; start_path = true; }                                        // :218 `continue`
			else
			{
				get_surface(sc, ti, r_from + r_dir * z, bu, bv, hit);
				const yafgpu_material &pm = sc.mats[hit.mat];
				const uint32_t mb = mat_init_bsdf(pm, dat_n);
				if(sampled_flags != kNone) pwo = -r_dir;                                      // :224
				Col lcol = mkc(0.f, 0.f, 0.f);
				if(sc.n_lights > 0)
				{	// estimateOneDirectLight, integrator_montecarlo.cc:62-76
					int lnum = 0;
					if(sc.n_lights > 1)
					{
						Halton h2; h2.init(2u);
						h2.set_start(rp.base_sampling_offset + (sample_ordinal * 16u + one_light_calls) - 1u);
						lnum = min((int)(h2.next() * (float)sc.n_lights), sc.n_lights - 1);
					}
					++one_light_calls;
					lcol = direct_light<kStats>(ra, stk, hit, pm, dat_n, pwo, lnum, lnum + 1, pixel_sample, sampling_offs, cn) * (float)sc.n_lights;
				}
				if(mb & kEmit) lcol = lcol + mat_emit(pm, hit, pwo, false);                   // :226
				path_col = path_col + lcol * throughput;                                       // :228
				depth = 1;
				if(depth < rp.bounces) extend = true;
				else { ++path_i; start_path = true; }
			}
		}
		else
		{
			if(!got) { ++path_i; start_path = true; }                                        // :259-266 `break`
			else
			{
				get_surface(sc, ti, r_from + r_dir * z, bu, bv, hit);
				const yafgpu_material &pm = sc.mats[hit.mat];
				const uint32_t mb = mat_init_bsdf(pm, dat_n);
				pwo = -r_dir;                                                                  // :271
				Col lcol = mkc(0.f, 0.f, 0.f);
				if((mb & kDiffuse) && sc.n_lights > 0)
				{
					int lnum = 0;
					if(sc.n_lights > 1)
					{
						Halton h2; h2.init(2u);
						h2.set_start(rp.base_sampling_offset + (sample_ordinal * 16u + one_light_calls) - 1u);
						lnum = min((int)(h2.next() * (float)sc.n_lights), sc.n_lights - 1);
					}
					++one_light_calls;
					lcol = direct_light<kStats>(ra, stk, hit, pm, dat_n, pwo, lnum, lnum + 1, pixel_sample, sampling_offs, cn) * (float)sc.n_lights;
				}
				bool alive = true;
				if(depth > rp.rr_min_bounces)
				{	// Russian roulette :282-288
					const float random_value = (float)rr.next();
					const float probability = smax(throughput.r, smax(throughput.g, throughput.b));
					if(probability <= 0.f || probability < random_value) alive = false;
					else throughput = throughput * (1.f / probability);
				}
				if(alive)
				{
					path_col = path_col + lcol * throughput;                                   // :292
					++depth;
					if(depth < rp.bounces) extend = true;
					else { ++path_i; start_path = true; }
				}
				else { ++path_i; start_path = true; }
			}
		}

		if(extend)
		{
			// next segment from the current vertex, :232-257
			const yafgpu_material &pm = sc.mats[hit.mat];
			const int d_4 = 4 * depth;
			BsdfSample bs;
			bs.s_1 = (float)scr_halton(sc, d_4 + 3, offs);
			bs.s_2 = (float)scr_halton(sc, d_4 + 4, offs);
			bs.pdf = 0.f; bs.sampled = kNone; bs.flags = kAll;
			float w = last_w;
			V3 p_dir = r_dir;
			const Col scol = mat_sample(pm, dat_n, hit, pwo, p_dir, bs, w) * w;
			last_w = w;
			if(is_black(scol)) { ++path_i; start_path = true; }                               // :249 `break`
			else
			{
				throughput = throughput * scol;
				r_from = hit.p; r_dir = p_dir; r_tmin = ra.ray_min_dist; r_tmax = -1.f;
				stage = kStDepth;
			}
		}
		if(start_path)
		{
			if(path_i >= n_paths) { col = col + path_col / (float)n_paths; break; }           // :297
			// first segment from the camera hit, :186-216
			const yafgpu_material &m = sc.mats[mat0];
			offs = (uint32_t)rp.path_samples * pixel_sample + sampling_offs + (uint32_t)path_i;
			BsdfSample bs;
			bs.s_1 = ri_vdc(offs, 0u);
			bs.s_2 = (float)scr_halton(sc, 2, offs);
			bs.pdf = 0.f; bs.sampled = kNone;
			bs.flags = (rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse) | kDiffuse | kReflect | kTransmit;
			float w = last_w;
			V3 p_dir = mk(0.f, 0.f, 0.f);
			pwo = wo0;
			const Col scol = mat_sample(m, dat0, sp0, pwo, p_dir, bs, w) * w;
			last_w = w;
			throughput = scol;
			sampled_flags = bs.sampled;
			r_from = sp0.p; r_dir = p_dir; r_tmin = ra.ray_min_dist; r_tmax = -1.f;
			stage = kStFirst;
		}
	}
	// EmptyVolumeIntegrator (integrator_empty_volume.cc:32-38): transmittance 1, in-scatter 0
	if(rp.bg_transp) alpha = smax(alpha, 0.f);
	out[0] = col.r; out[1] = col.g; out[2] = col.b; out[3] = alpha;
}

YG_DEV int round2int(double v) { return (int)(v + (.5 - 1.4e-11)); } // util_math.h:34-43

YG_DEV uint32_t read_xcc_id()
{
	uint32_t v;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
	return v & 7u;
}

template<typename T> YG_DEV T wave_bcast(T v, int src) { return __shfl(v, src, kWave); }

YG_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for(int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
	return v;
}

#ifndef YAFGPU_VARIANT_TU      // the one-kernel pipeline belongs to the main unit
// ------------------------------------------------------------------------------------------------
// The render pass: TiledIntegrator::renderTile (integrator_tiled.cc:309-521) for every tile of the
// shard + ImageFilm::addSample (imagefilm.cc:925-1015) for the box filter of width <= 1.002 px
// (filterw = 0.501 after the clamp at :165): a sample lands on its own pixel and, when dx (dy)
// >= 0.999, also on the right (lower) neighbour, weight 1 each.
template<bool kStats>
__global__ __launch_bounds__(kBlock, YAFGPU_WAVES) void render_kernel(const RenderArgs ra)
{
	__shared__ uint2 s_stack[kWavesPerBlock][kStack][kWave];
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	LaneStack stk;
	stk.col = &s_stack[wave][0][lane];
	LaneCounters cn = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
	const yafgpu_render_params &rp = ra.rp;
	const int L = ra.lanes_per_pixel, P = ra.pixels_per_wave;
	const int group = lane / L, lane_in_pixel = lane - group * L;
	const int n_samples = rp.aa_minsamples;
	const float d_1 = (float)(1.0 / (double)(float)n_samples);    // integrator_tiled.cc:316
	const int cx0 = rp.xstart, cy0 = rp.ystart, cx1 = rp.xstart + rp.width, cy1 = rp.ystart + rp.height;
	const size_t plane_stride = (size_t)rp.width * (size_t)rp.height * YAFGPU_FILM_CHANNELS;
	const uint32_t my_queue = read_xcc_id();

	for(int qi = 0; qi < kQueues; ++qi)
	{
		const uint32_t q = (my_queue + (uint32_t)qi) & (kQueues - 1);
		const uint32_t q_begin = ra.queue_begin[q], q_end = ra.queue_begin[q + 1];
		for(;;)
		{
			uint32_t u = 0u;
			if(lane == 0) u = atomicAdd(&ra.queue_next[q * 32u], 1u);
			u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u) + q_begin;
			if(u >= q_end) break;
			// tile of this unit (wave-uniform binary search over the prefix sums)
			int lo = 0, hi = ra.n_tiles;
			while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(ra.unit_prefix[mid] <= u) lo = mid; else hi = mid; }
			const int4 rect = ra.tile_rect[lo];
			const int k = (int)(u - ra.unit_prefix[lo]);
			const int qpix = k * P + group;
			const bool pix_ok = (group < P) && (qpix < rect.z * rect.w);
			const int px = rect.x + (pix_ok ? qpix % rect.z : 0), py = rect.y + (pix_ok ? qpix / rect.z : 0);
			const uint32_t sampling_offs = fnv32a((uint32_t)py * fnv32a((uint32_t)px));       // :379
			float acc[YAFGPU_FILM_PLANES][YAFGPU_FILM_CHANNELS];
#pragma unroll
			for(int a = 0; a < YAFGPU_FILM_PLANES; ++a)
#pragma unroll
				for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) acc[a][c] = 0.f;

			for(int it = 0; it < ra.iters; ++it)
			{
				const int sample = it * L + lane_in_pixel;
				const bool active = pix_ok && sample < n_samples;
				float c4[4] = {0.f, 0.f, 0.f, 0.f};
				uint32_t flag = 0u;
				if(active)
				{
					float dx = 0.5f, dy = 0.5f;
					if(n_samples > 1)
					{	// :399-403 (single pass)
						dx = (float)((0.5 + (double)(float)sample) * (double)d_1);
						dy = ri_lp((uint32_t)sample + sampling_offs, 0u);
					}
					V3 from, dir; float tmin, tmax;
					camera_shoot(ra.sc.cam, (float)px + dx, (float)py + dy, from, dir, tmin, tmax);   // :410
					++cn.samples;
					const uint32_t pixel_sample = rp.base_sampling_offset + (uint32_t)sample;          // :389
					const uint32_t ordinal = ((uint32_t)(py - cy0) * (uint32_t)rp.width + (uint32_t)(px - cx0)) * (uint32_t)n_samples + (uint32_t)sample;
					integrate<kStats>(ra, stk, from, dir, tmin, tmax, pixel_sample, sampling_offs, ordinal, c4, cn);
					if(c4[3] > 1.f) c4[3] = 1.f;                                                     // :459
					// footprint, imagefilm.cc:933-936 with filterw = 0.501
					const int dx_1 = min(cx1 - px - 1, round2int((double)dx + (double)ra.filterw - 1.0));
					const int dy_1 = min(cy1 - py - 1, round2int((double)dy + (double)ra.filterw - 1.0));
					flag = 1u | (dx_1 >= 1 ? 2u : 0u) | (dy_1 >= 1 ? 4u : 0u);
				}
				// sequential per-pixel sums in sample order; every lane of a group replays its group
				const int base = group * L;
				for(int s = 0; s < L; ++s)
				{
					const int src = (base + s) & (kWave - 1);
					const uint32_t f = wave_bcast(flag, src);
					const float r = wave_bcast(c4[0], src), g = wave_bcast(c4[1], src), b = wave_bcast(c4[2], src), al = wave_bcast(c4[3], src);
					if(f & 1u)
					{
						acc[0][0] += r; acc[0][1] += g; acc[0][2] += b; acc[0][3] += al; acc[0][4] += 1.f;
						if(f & 2u) { acc[1][0] += r; acc[1][1] += g; acc[1][2] += b; acc[1][3] += al; acc[1][4] += 1.f; }
						if(f & 4u) { acc[2][0] += r; acc[2][1] += g; acc[2][2] += b; acc[2][3] += al; acc[2][4] += 1.f; }
						if((f & 6u) == 6u) { acc[3][0] += r; acc[3][1] += g; acc[3][2] += b; acc[3][3] += al; acc[3][4] += 1.f; }
					}
				}
			}
			if(pix_ok && lane_in_pixel == 0)
			{
				const size_t pix = ((size_t)(py - cy0) * (size_t)rp.width + (size_t)(px - cx0)) * YAFGPU_FILM_CHANNELS;
#pragma unroll
				for(int a = 0; a < YAFGPU_FILM_PLANES; ++a)
				{
					if(a == 0 || acc[a][4] != 0.f)
					{
						float *dst = ra.planes + (size_t)a * plane_stride + pix;
#pragma unroll
						for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) dst[c] = acc[a][c];
					}
				}
			}
		}
	}
	if(ra.counters != nullptr)
	{
		const uint32_t v0 = wave_sum(cn.closest), v1 = wave_sum(cn.shadow), v2 = wave_sum(cn.interior), v3 = wave_sum(cn.leaves),
		               v4 = wave_sum(cn.tests), v5 = wave_sum(cn.samples), v6 = wave_sum(cn.restarts);
		if(lane == 0)
		{
			atomicAdd((unsigned long long *)&ra.counters->rays_closest, (unsigned long long)v0);
			atomicAdd((unsigned long long *)&ra.counters->rays_shadow, (unsigned long long)v1);
			atomicAdd((unsigned long long *)&ra.counters->camera_samples, (unsigned long long)v5);
			if(kStats)
			{
				atomicAdd((unsigned long long *)&ra.counters->interior_steps, (unsigned long long)v2);
				atomicAdd((unsigned long long *)&ra.counters->leaves, (unsigned long long)v3);
				atomicAdd((unsigned long long *)&ra.counters->tri_tests, (unsigned long long)v4);
				atomicAdd((unsigned long long *)&ra.counters->restarts, (unsigned long long)v6);
			}
		}
	}
}

#endif // YAFGPU_VARIANT_TU

} // namespace yafgpu
#include "yafgpu_wavefront.h"
#ifndef YAFGPU_VARIANT_TU      // everything below: kernels and host code of the main unit
namespace yafgpu {

// film[y][x] = own + right(x-1,y) + down(x,y-1) + diag(x-1,y-1): the neighbours' splats onto this pixel
__global__ __launch_bounds__(kBlock) void combine_kernel(const float *planes, float *film, int w, int h)
{
	const size_t n = (size_t)w * (size_t)h;
	const size_t stride = n * YAFGPU_FILM_CHANNELS;
	for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const int x = (int)(i % (size_t)w), y = (int)(i / (size_t)w);
#pragma unroll
		for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c)
		{
			float v = planes[i * YAFGPU_FILM_CHANNELS + c];
			if(x > 0) v += planes[stride + (i - 1) * YAFGPU_FILM_CHANNELS + c];
			if(y > 0) v += planes[2 * stride + (i - (size_t)w) * YAFGPU_FILM_CHANNELS + c];
			if(x > 0 && y > 0) v += planes[3 * stride + (i - (size_t)w - 1) * YAFGPU_FILM_CHANNELS + c];
			film[i * YAFGPU_FILM_CHANNELS + c] = v;
		}
	}
}

// ray batches: Scene::intersect / Scene::isShadowed on arrays (tests and kernel-level measurements)
template<bool kAny>
__global__ __launch_bounds__(kBlock) void trace_kernel(const DevScene sc, int n, const float *rays, int *tri, float *t, float *bary, int *shadowed)
{
	__shared__ uint2 s_stack[kWavesPerBlock][kStack][kWave];
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	LaneStack stk;
	stk.col = &s_stack[wave][0][lane];
	LaneCounters cn = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
	for(int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x))
	{
		const float *r = rays + 8 * (size_t)i;
		const V3 from = mk(r[0], r[1], r[2]), dir = mk(r[3], r[4], r[5]);
		const float tmin = r[6], tmax = r[7];
		int ti = -1; float z = 0.f, bu = 0.f, bv = 0.f;
		if(kAny)
		{
			const V3 sfrom = from + dir * tmin;
			const float dis = (tmax < 0.f) ? INFINITY : tmax - 2.f * tmin;
			shadowed[i] = kd_trace<true, false>(sc, stk, sfrom, dir, 0.f, dis, ti, z, bu, bv, cn) ? 1 : 0;
		}
		else
		{
			const float dis = (tmax < 0.f) ? INFINITY : tmax;
			const bool h = kd_trace<false, false>(sc, stk, from, dir, tmin, dis, ti, z, bu, bv, cn);
			tri[i] = h ? ti : -1; t[i] = h ? z : 0.f;
			bary[3 * i] = h ? 1.f - bu - bv : 0.f; bary[3 * i + 1] = h ? bu : 0.f; bary[3 * i + 2] = h ? bv : 0.f;
		}
	}
}

// Component probe: evaluates the device-side restatements of the reference's leaf functions on
// arrays, so that tests can pin them against the reference's own golden vectors (tests/golden).
// Integers travel as float bit patterns.  One thread per item; no LDS, no traversal.
__global__ __launch_bounds__(kBlock) void probe_kernel(const DevScene sc, int op, int n, const float *in, int n_in, float *out, int n_out)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if(i >= n) return;
	const float *x = in + (size_t)i * (size_t)n_in;
	float *o = out + (size_t)i * (size_t)n_out;
	switch(op)
	{
		case 1:
		{
			const float ax = fabsf(x[0]) + 1e-3f;
			o[0] = f_sin(x[0]); o[1] = f_cos(x[0]); o[2] = f_exp2(x[0]); o[3] = f_log2(ax); o[4] = f_sqrt(ax);
			break;
		}
		case 2: o[0] = f_pow(x[0], x[1]); break;
		case 3:
		{
			const uint32_t bits = __float_as_uint(x[0]), r = __float_as_uint(x[1]);
			o[0] = ri_vdc(bits, r); o[1] = ri_lp(bits, r); o[2] = __uint_as_float(fnv32a(bits)); o[3] = ri_s(bits, r);
			break;
		}
		case 4:
		{
			const double v = scr_halton(sc, (int)__float_as_uint(x[0]), __float_as_uint(x[1]));
			o[0] = (float)v;
			o[1] = __uint_as_float((uint32_t)(__double_as_longlong(v) & 0xffffffffll));
			o[2] = __uint_as_float((uint32_t)((unsigned long long)__double_as_longlong(v) >> 32));
			break;
		}
		case 5:
		{
			Halton h; h.init(__float_as_uint(x[0])); h.set_start(__float_as_uint(x[1]));
			for(int k = 0; k < 6; ++k) o[k] = h.next();
			break;
		}
		case 6:
		{
			const V3 nn = mk(x[0], x[1], x[2]);
			V3 u, v; create_cs(nn, u, v);
			const V3 w = sample_cos_hemisphere(nn, u, v, x[3], x[4]);
			o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = v.x; o[4] = v.y; o[5] = v.z; o[6] = w.x; o[7] = w.y; o[8] = w.z;
			break;
		}
		case 7:
		{
			V3 f, d; float t0, t1;
			if(n_in >= 4) camera_shoot(sc.cam, x[0], x[1], x[2], x[3], f, d, t0, t1);
			else camera_shoot(sc.cam, x[0], x[1], f, d, t0, t1);
			o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = d.x; o[4] = d.y; o[5] = d.z; o[6] = t0; o[7] = t1; o[8] = 1.f;
			break;
		}
		case 8:
		{
			const yafgpu_light &l = sc.lights[0];
			V3 d = mk(0.f, 0.f, 0.f); float tmax = 0.f, pdf = 0.f;
			const bool ok = arealight_illum_sample(l, mk(x[0], x[1], x[2]), x[3], x[4], d, tmax, pdf);
			o[0] = ok ? 1.f : 0.f;
			o[1] = ok ? d.x : 0.f; o[2] = ok ? d.y : 0.f; o[3] = ok ? d.z : 0.f; o[4] = ok ? tmax : 0.f; o[5] = ok ? pdf : 0.f;
			o[6] = ok ? l.color[0] : 0.f; o[7] = ok ? l.color[1] : 0.f; o[8] = ok ? l.color[2] : 0.f;
			break;
		}
		case 9:
		{
			const yafgpu_light &l = sc.lights[0];
			float t = 0.f, ipdf = 0.f;
			const bool ok = arealight_intersect(l, mk(x[0], x[1], x[2]), mk(x[3], x[4], x[5]), t, ipdf);
			o[0] = ok ? 1.f : 0.f; o[1] = ok ? t : 0.f; o[2] = ok ? ipdf : 0.f;
			o[3] = ok ? l.color[0] : 0.f; o[4] = ok ? l.color[1] : 0.f; o[5] = ok ? l.color[2] : 0.f;
			break;
		}
		case 10:
		{
			const yafgpu_light &l = sc.lights[1];
			Col c = mkc(0.f, 0.f, 0.f); V3 d = mk(0.f, 0.f, 0.f); float tmax = 0.f;
			pointlight_illuminate(l, mk(x[0], x[1], x[2]), c, d, tmax);
			o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = tmax; o[4] = c.r; o[5] = c.g; o[6] = c.b;
			break;
		}
		case 11:
		{	// x = material index, n, ng, wo, wl, s1, s2, sample flags
			const yafgpu_material &m = sc.mats[__float_as_uint(x[0])];
			SurfPt sp; sp.n = mk(x[1], x[2], x[3]); sp.ng = mk(x[4], x[5], x[6]); sp.p = mk(0.f, 0.f, 0.f); sp.mat = 0;
			create_cs(sp.n, sp.nu, sp.nv);
			const V3 wo = mk(x[7], x[8], x[9]), wl = mk(x[10], x[11], x[12]);
			BsdfDat d;
			const uint32_t fl = mat_init_bsdf(m, d);
			const Col e = mat_eval(m, d, sp, wo, wl, kAll);
			const float pdf = mat_pdf(m, d, sp, wo, wl, kGlossy | kDiffuse | kDispersive | kReflect | kTransmit);
			BsdfSample bs; bs.s_1 = x[13]; bs.s_2 = x[14]; bs.pdf = 0.f; bs.flags = __float_as_uint(x[15]); bs.sampled = kNone;
			V3 wi = mk(0.f, 0.f, 0.f); float w = 0.f;
			const Col sc_ = mat_sample(m, d, sp, wo, wi, bs, w);
			o[0] = __uint_as_float(fl); o[1] = e.r; o[2] = e.g; o[3] = e.b; o[4] = pdf; o[5] = __uint_as_float(bs.sampled);
			o[6] = sc_.r; o[7] = sc_.g; o[8] = sc_.b; o[9] = wi.x; o[10] = wi.y; o[11] = wi.z; o[12] = bs.pdf; o[13] = w;
			const Col em = mat_emit(m, sp, wo, x[15] != 0.f);
			o[14] = em.r; o[15] = em.g; o[16] = em.b;
			break;
		}
		case 12:
		{	// Material::getSpecular + getAlpha: x = material index, n, ng, wo
			const yafgpu_material &m = sc.mats[__float_as_uint(x[0])];
			SurfPt sp; sp.n = mk(x[1], x[2], x[3]); sp.ng = mk(x[4], x[5], x[6]); sp.p = mk(0.f, 0.f, 0.f); sp.mat = 0;
			create_cs(sp.n, sp.nu, sp.nv);
			const V3 wo = mk(x[7], x[8], x[9]);
			BsdfDat d;
			mat_init_bsdf(m, d);
			bool refl, refr; V3 d0, d1; Col c0, c1;
			mat_get_specular(m, d, sp, wo, (n_in >= 11) ? (int)x[10] : 1, refl, refr, d0, c0, d1, c1);
			o[0] = __uint_as_float((refl ? 1u : 0u) | (refr ? 2u : 0u));
			o[1] = d0.x; o[2] = d0.y; o[3] = d0.z; o[4] = c0.r; o[5] = c0.g; o[6] = c0.b;
			o[7] = d1.x; o[8] = d1.y; o[9] = d1.z; o[10] = c1.r; o[11] = c1.g; o[12] = c1.b;
			o[13] = mat_alpha(m, d, sp, wo);
			const Col tr = mat_transparency(m, sp, wo);
			o[14] = tr.r; o[15] = tr.g; o[16] = tr.b;
			break;
		}
		case 13:
		{	// image texture lookup: in (p.xyz, texture index) -> getColor rgba, getFloat
			const int ti = (int)__float_as_uint(x[3]);
			if(sc.tex.nodes == nullptr && sc.tex.textures == nullptr) break;
			if(ti < 0 || ti >= sc.tex.n_textures) break;
			const Rgba4 c = tex_get_color(sc.tex, sc.tex.textures[ti], mk(x[0], x[1], x[2]));
			o[0] = c.r; o[1] = c.g; o[2] = c.b; o[3] = c.a; o[4] = tex_get_float(sc.tex, sc.tex.textures[ti], mk(x[0], x[1], x[2]));
			break;
		}
		case 14:
		{	// node stack of material x[18] at a surface point (p, n, ng, orco_p, orco_ng, u, v): n_nodes x (rgba, scalar); one node
			// range at a time (x[19] = first node of the range, x[20] = count <= kMaxNodes) so that graphs larger than a material's
			// limit can be pinned piecewise by tests that arrange the ranges to be closed under dependencies
			if(sc.tex.nodes == nullptr) break;
			TexPoint tp;
			tp.p = mk(x[0], x[1], x[2]); tp.n = mk(x[3], x[4], x[5]); tp.ng = mk(x[6], x[7], x[8]);
			tp.orco_p = mk(x[9], x[10], x[11]); tp.orco_ng = mk(x[12], x[13], x[14]); tp.u = x[15]; tp.v = x[16];
			const int first = (int)__float_as_uint(x[18]), cnt = min((int)__float_as_uint(x[19]), kMaxNodes);
			NodeResult stack[kMaxNodes];
			nodes_eval(sc.tex, sc.tex.nodes + first, cnt, sc.cam, tp, stack);
			for(int k = 0; k < cnt && 5 * k + 4 < n_out; ++k) { o[5 * k] = stack[k].col.r; o[5 * k + 1] = stack[k].col.g; o[5 * k + 2] = stack[k].col.b; o[5 * k + 3] = stack[k].col.a; o[5 * k + 4] = stack[k].f; }
			break;
		}
		case 15:
		{	// bump mapping: evalDerivative of a node range at a surface point — op 14's 20 words with x[17] = the scale applied to the last
			// node's derivative, then ds_du, ds_dv, nu, nv, has_uv: n_nodes x (du, dv, 0, alpha, f), then (out + 5 * cnt) n, nu, nv after applyBump
			if(sc.tex.nodes == nullptr || n_in < 33) break;
			TexPoint tp;
			tp.p = mk(x[0], x[1], x[2]); tp.n = mk(x[3], x[4], x[5]); tp.ng = mk(x[6], x[7], x[8]);
			tp.orco_p = mk(x[9], x[10], x[11]); tp.orco_ng = mk(x[12], x[13], x[14]); tp.u = x[15]; tp.v = x[16];
			tp.ds_du = mk(x[20], x[21], x[22]); tp.ds_dv = mk(x[23], x[24], x[25]); tp.nu = mk(x[26], x[27], x[28]); tp.nv = mk(x[29], x[30], x[31]);
			tp.has_uv = x[32] != 0.f;
			const int first = (int)__float_as_uint(x[18]), cnt = min((int)__float_as_uint(x[19]), kMaxNodes);
			NodeResult stack[kMaxNodes];
			nodes_eval_derivative(sc.tex, sc.tex.nodes + first, cnt, sc.cam, tp, stack);
			for(int k = 0; k < cnt && 5 * k + 4 < n_out; ++k) { o[5 * k] = stack[k].col.r; o[5 * k + 1] = stack[k].col.g; o[5 * k + 2] = stack[k].col.b; o[5 * k + 3] = stack[k].col.a; o[5 * k + 4] = stack[k].f; }
			if(cnt > 0 && 5 * cnt + 9 <= n_out)
			{
				V3 n = tp.n, nu = tp.nu, nv = tp.nv;
				apply_bump(n, nu, nv, stack[cnt - 1].col.r * x[17], stack[cnt - 1].col.g * x[17]);
				float *q = o + 5 * cnt;
				q[0] = n.x; q[1] = n.y; q[2] = n.z; q[3] = nu.x; q[4] = nu.y; q[5] = nu.z; q[6] = nv.x; q[7] = nv.y; q[8] = nv.z;
			}
			break;
		}
		case 16:
		{	// the two-direction Material::sample of recursiveRaytrace's glossy branch (rough glass): op 11's 16 words in,
			// dir[0], ret, w[0], dir[1], tcol, w[1], pdf, sampled flags out
			const yafgpu_material &m = sc.mats[__float_as_uint(x[0])];
			if(m.type != YAFGPU_MAT_ROUGH_GLASS || n_out < 16) break;
			SurfPt sp; sp.n = mk(x[1], x[2], x[3]); sp.ng = mk(x[4], x[5], x[6]); sp.p = mk(0.f, 0.f, 0.f); sp.mat = 0;
			create_cs(sp.n, sp.nu, sp.nv);
			const V3 wo = mk(x[7], x[8], x[9]);
			BsdfSample bs; bs.s_1 = x[13]; bs.s_2 = x[14]; bs.pdf = 0.f; bs.flags = __float_as_uint(x[15]); bs.sampled = kNone;
			V3 d0 = mk(0.f, 0.f, 0.f), d1 = d0; Col tcol = mkc(0.f, 0.f, 0.f); float w0 = 0.f, w1 = 0.f;
			const Col ret = rough_glass_sample(m, sp, wo, bs, true, d0, w0, d1, tcol, w1);
			o[0] = d0.x; o[1] = d0.y; o[2] = d0.z; o[3] = ret.r; o[4] = ret.g; o[5] = ret.b; o[6] = w0;
			o[7] = d1.x; o[8] = d1.y; o[9] = d1.z; o[10] = tcol.r; o[11] = tcol.g; o[12] = tcol.b; o[13] = w1; o[14] = bs.pdf; o[15] = __uint_as_float(bs.sampled);
			break;
		}
		default: break;
	}
}

} // namespace yafgpu

// ================================================================================================
// host side of the narrow ABI
using namespace yafgpu;

static float host_fsin(float x);
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_OK(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) return fail(-100, std::string(#expr) + ": " + hipGetErrorString(e_)); } while(0)

// device allocation released on every exit path of the host entry points
template<typename T> struct DevMem
{
	T *p = nullptr;
	DevMem() = default;
	DevMem(const DevMem &) = delete; DevMem &operator=(const DevMem &) = delete;
	~DevMem() { if(p) (void)hipFree(p); }
	hipError_t alloc(size_t count) { return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)); }
	operator T *() const { return p; }
};
struct EventPair
{
	hipEvent_t e[2] = {nullptr, nullptr};
	~EventPair() { for(hipEvent_t x : e) if(x) (void)hipEventDestroy(x); }
};

struct yafgpu_scene
{
	DevScene dev{};
	std::vector<void *> allocs;
	KdTree tree;
	yafgpu_tree_info info{};
	std::vector<yafgpu_material> mats;
	std::vector<yafgpu_light> h_lights;
	int n_lights = 0;
	// per-render scratch, grown on demand
	int4 *d_tiles = nullptr; uint32_t *d_prefix = nullptr; uint32_t *d_queue = nullptr; size_t tiles_cap = 0;
	// the tile list of the last launch stays resident; it is re-uploaded only when its key changes
	std::vector<int4> h_tiles; std::vector<uint32_t> h_prefix;
	int tile_key[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
	// wavefront workspace (allocated on first use, sized for kWfMaxPaths paths or the whole frame)
	std::vector<uint32_t> h_pix_prefix; uint32_t *d_pix_prefix = nullptr; size_t pix_prefix_cap = 0;
	float4 *wf_state = nullptr, *wf_results = nullptr; uint32_t *wf_queues = nullptr, *wf_counts = nullptr, *wf_verdict = nullptr, *wf_pix_xy = nullptr; uint32_t wf_cap = 0;
	hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // the any-hit launch of an iteration runs beside the closest-hit one
	// Pass pipelining (render_wavefront): consecutive passes that do not depend on each other's film take turns on two internal streams, each
	// with its own set of the wavefront buffers, so that one pass's launch tails are filled by the other's launches; the film is added to
	// on the caller's stream, in call order.  `alt[]` are the sets that are not in the members above at the moment.
	struct WfSet
	{
		uint32_t *d_pix_prefix = nullptr; size_t pix_prefix_cap = 0;
		float4 *wf_state = nullptr, *wf_results = nullptr, *wf_filt = nullptr;
		uint32_t *wf_queues = nullptr, *wf_counts = nullptr, *wf_verdict = nullptr, *wf_pix_xy = nullptr;
		uint32_t wf_cap = 0, wf_filt_cap = 0; int wf_frames = 0;
		hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	} alt[2];
	static constexpr int kPipeMax = 3;      // passes in flight at most (buffer sets: the members above + alt[])
	hipStream_t pipe_stream[kPipeMax] = {};
	hipEvent_t pipe_done[kPipeMax] = {}, pipe_acc[kPipeMax] = {}, pipe_sync = nullptr;
	bool pipe_acc_set[kPipeMax] = {}, pipe_prev = false;
	int pipe_next = 0, pass_pipelining = -1;      // -1: by size (render_wavefront), 0 / 1: forced (yafgpu_scene_set_pass_pipelining)
	yafgpu_counters *pipe_counters[kPipeMax] = {};      // a pipelined pass counts here; the sums reach the caller's block on the caller's stream
	uint32_t mat_mask = 0u;              // bit per YAFGPU_MAT_* present; picks the shading kernel variant
	bool has_volumetric = false;
	int max_add_depth = 0;               // the largest Material::additional_depth_ of the scene: recursion frames beyond raydepth
	bool has_glossy_two = false;         // rough glass: a glossy trajectory sends two rays (the replay's call count)
	bool has_glossy = false;             // some material has a glossy lobe that recursiveRaytrace samples (glossy / coated_glossy with as_diffuse off): 12-record frames
	bool has_aniso = false;              // some material has the anisotropic glossy lobe: the general shading kernel
	bool has_bump = false;               // some material has a bump shader: the shading frame is parked per vertex (records 24 / 25, frame record 12)
	bool has_textures = false;           // some material in use has shader nodes: the general shading kernel, texture coordinates parked per path
	bool has_specular = false, has_transparent = false; int wf_frames = 0; float4 *wf_filt = nullptr; uint32_t wf_filt_cap = 0;      // recursiveRaytrace frames allocated behind the working records
	float *d_filter_table = nullptr;
	// serial-state replay tables (WfArgs::replay)
	uint32_t *rp_flags = nullptr; float *rp_p = nullptr; uint8_t *rp_kill = nullptr, *rp_calls = nullptr; uint32_t *rp_base = nullptr; size_t rp_ents = 0; uint32_t rp_prob = 0;
	uint32_t *rp_seg_begin = nullptr, *rp_seg_seed = nullptr, *rp_seg_total = nullptr, *rp_seg_base = nullptr, *rp_counter = nullptr; size_t rp_segs = 0;
	// render targets of the host-film entry points, kept between calls (allocating and freeing 100 MB per render cost up to half a
	// second a call on this runtime — ten times the pass itself at 1024x1024)
	float *rt_planes = nullptr, *rt_film = nullptr; yafgpu_counters *rt_cnt = nullptr; size_t rt_planes_n = 0, rt_film_n = 0;
	int n_cus = 0;                                            // compute units of the device the scene lives on
	uint8_t *rt_flags = nullptr; size_t rt_flags_n = 0;       // resample flags of the detection step between adaptive passes
	float4 *rp_hits = nullptr; size_t rp_hits_cap = 0;       // closest-hit answers of the record pass (WfArgs::hit_cache)
	uint32_t lc_host_counter = 0;                        // correlative_sample_number_ of a sharded render: the same value on every rank (lc_exchange_counts)
	std::vector<uint32_t> h_seg_base;
	std::vector<uint32_t> h_listed;                      // pixels of a masked (adaptive) pass, in tile order
	std::vector<uint32_t> h_seg_begin, h_seg_seed;       // every chunk's segments of the pass, uploaded once (scene-owned: an async copy may read them late)
	yafgpu_exchange_fn exchange = nullptr; void *exchange_user = nullptr;     // yafgpu_scene_set_exchange
	const volatile int32_t *abort_flag = nullptr;      // polled between chunks and passes (yafgpu_scene_set_abort_flag)
	bool aborted() const { return abort_flag && *abort_flag != 0; }
	bool profiling = false;
	double prof_ms[4] = {0, 0, 0, 0}; uint64_t prof_launches[4] = {0, 0, 0, 0};   // trace closest, trace shadow, shade, other
};

template<typename T> static int upload(yafgpu_scene *s, const T *src, size_t n, const T **dst)
{
	void *p = nullptr;
	const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
	HIP_OK(hipMalloc(&p, bytes));
	s->allocs.push_back(p);
	if(n) HIP_OK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
	s->info.device_bytes += bytes;
	*dst = (const T *)p;
	return 0;
}

// Faure permutations (the reference ships them as a table, src/common/faure_tables.cc; they are
// the standard construction of Faure 1992 and are regenerated here rather than copied)
static void faure_perm(int b, std::vector<int> &out)
{
	if(b == 2) { out = {0, 1}; return; }
	if(b % 2 == 0)
	{
		std::vector<int> half; faure_perm(b / 2, half);
		out.resize((size_t)b);
		for(int i = 0; i < b / 2; ++i) { out[(size_t)i] = 2 * half[(size_t)i]; out[(size_t)(b / 2 + i)] = 2 * half[(size_t)i] + 1; }
	}
	else
	{
		std::vector<int> prev; faure_perm(b - 1, prev);
		const int m = (b - 1) / 2;
		for(int &v : prev) if(v >= m) ++v;
		out.assign(prev.begin(), prev.begin() + m);
		out.push_back(m);
		out.insert(out.end(), prev.begin() + m, prev.end());
	}
}

static const int kPrimsHost[50] = {1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67,
                                   71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167,
                                   173, 179, 181, 191, 193, 197, 199, 211, 223, 227};

extern "C" {

const char *yafgpu_last_error(void) { return g_err.c_str(); }
extern "C" void yafgpu_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

int yafgpu_device_count(void)
{
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}
int yafgpu_set_device(int device) { HIP_OK(hipSetDevice(device)); return 0; }

uint64_t yafgpu_planes_bytes(int32_t width, int32_t height)
{
	return (uint64_t)YAFGPU_FILM_PLANES * (uint64_t)width * (uint64_t)height * YAFGPU_FILM_CHANNELS * sizeof(float);
}

int yafgpu_scene_create(const yafgpu_scene_desc *d, yafgpu_scene_t **out)
{
	if(!d || !out) return fail(-1, "null argument");
	if(d->n_tris < 0 || d->n_materials <= 0) return fail(-2, "scene needs at least one material");
	// the light-estimate bookkeeping of a parked path packs the light index and the end of its light range in 8 bits each
	// (pack_dlc, yafgpu_wavefront.h)
	if(d->n_lights < 0 || d->n_lights > 255) return fail(-2, "more than 255 lights: the device path indexes lights with 8 bits");
	{	// a NaN or infinite coordinate poisons the scene bound and with it every ray's clip against it: refuse it here
		const size_t nf = (size_t)d->n_tris * 9;
		for(size_t k = 0; k < nf; ++k)
			if(!std::isfinite(d->verts[k])) return fail(-3, "non-finite vertex coordinate in triangle " + std::to_string(k / 9));
	}
	for(int i = 0; i < d->n_tris; ++i)
		if(d->tri_mat[i] < 0 || d->tri_mat[i] >= d->n_materials) return fail(-3, "triangle material index out of range");
	for(int i = 0; i < d->n_materials; ++i)
	{
		// recursiveRaytrace (integrator_montecarlo.cc:782-1028): the perfect specular branch and both cases of the glossy branch (reflect
		// only; reflect + transmit, which is rough glass and nothing else) are on the device path; the dispersive branch is not
		if(d->materials[i].type < 0 || d->materials[i].type > YAFGPU_MAT_ROUGH_GLASS) return fail(-3, "material " + std::to_string(i) + ": unknown type");
		if(d->materials[i].bsdf_flags & kDispersive)
			return fail(-4, "material with a dispersive lobe needs recursiveRaytrace's dispersive branch, which the GPU path does not implement");
		if((d->materials[i].bsdf_flags & kGlossy) && (d->materials[i].bsdf_flags & kTransmit) && d->materials[i].type != YAFGPU_MAT_ROUGH_GLASS)
			return fail(-4, "a glossy transmission lobe on a material that is not rough glass: the reflect + transmit case of recursiveRaytrace's glossy branch takes RoughGlassMaterial's two-direction sample");
		if(d->materials[i].type == YAFGPU_MAT_ROUGH_GLASS && !(d->materials[i].rg_a2 > 0.f))
			return fail(-3, "rough glass material " + std::to_string(i) + ": rg_a2 (alpha squared) must be positive");
	}
	auto *s = new yafgpu_scene();
	{	// What the scene's materials ask of the pipeline — recursion frames, the glossy loop's wider frames, extra depth, the transparent-shadow
		// kernel, the kernel variant — is taken from the materials some triangle actually USES: a definition nothing refers to can never be
		// hit, and must not cost frames, a kernel variant or (through the size of the replay's event tables) the exactness of the serial state
		std::vector<char> used((size_t)d->n_materials, 0);
		for(int i = 0; i < d->n_tris; ++i) used[(size_t)d->tri_mat[i]] = 1;
		for(int i = 0; i < d->n_materials; ++i)
		{
			if(!used[(size_t)i]) continue;
			const yafgpu_material &m = d->materials[i];
			s->mat_mask |= 1u << (uint32_t)m.type;
			if(m.bsdf_flags & kVolumetric) s->has_volumetric = true;
			if(m.bsdf_flags & (kSpecular | kFilter)) s->has_specular = true;
			if(m.anisotropic) s->has_aniso = true;
			s->max_add_depth = std::max(s->max_add_depth, std::min(std::max(m.additional_depth, 0), 15));
			if(m.bsdf_flags & kGlossy) s->has_glossy = true;
			if(m.type == YAFGPU_MAT_ROUGH_GLASS) s->has_glossy_two = true;
			if((m.type == YAFGPU_MAT_SHINYDIFFUSE && m.is_transparent) || ((m.type == YAFGPU_MAT_GLASS || m.type == YAFGPU_MAT_ROUGH_GLASS) && m.fake_shadow)) s->has_transparent = true;
		}
	}
	const auto t0 = std::chrono::steady_clock::now();
	{
		// 0: by size -- the device builder wins from a few ten thousand triangles on (1 M: 0.07 s against 0.33 s)
		bool on_device = d->build_on_device > 0 || (d->build_on_device == 0 && d->n_tris >= 65536);
		if(const char *e = std::getenv("YAFGPU_BUILD")) on_device = std::strcmp(e, "device") == 0;
		if(on_device)
		{
			std::string err;
			// heavily overlapping geometry can outgrow the builder's arrays: more room, and past that the host builder
			// (same format, same cost model) rather than no tree
			const int brc = build_kdtree_device_retry(d->verts, d->n_tris, kDepthCap, s->tree, &err);
			if(brc == -2 || brc == -3) build_kdtree(d->verts, d->n_tris, kDepthCap, d->build_threads, s->tree);     // arrays outgrown / no device memory for them
			else if(brc) { delete s; return fail(-20, err); }
		}
		else build_kdtree(d->verts, d->n_tris, kDepthCap, d->build_threads, s->tree);
	}
	s->info.build_seconds = s->tree.build_seconds;
	s->info.n_nodes = (uint32_t)s->tree.nodes.size();
	s->info.n_leaf_refs = (uint32_t)s->tree.refs.size();
	s->info.max_depth = (uint32_t)s->tree.max_depth;
	s->info.n_tris = (uint32_t)d->n_tris;
	const auto t1 = std::chrono::steady_clock::now();

	// triangle records: Triangle::updateIntersectionCachedValues (triangle.h:197-207), recNormal (:295-302)
	const size_t nt = (size_t)d->n_tris;
	std::vector<float4> rec(3 * nt), ng(nt), vn;
	bool any_smooth = false;
	for(size_t i = 0; i < nt; ++i)
	{
		const float *v = d->verts + 9 * i;
		const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
		const float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
		const float l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
		const float eps = (float)((double)0.1f * 0.00005 * (double)std::max(l1, l2));
		const uint32_t mat = (uint32_t)d->tri_mat[i];
		const uint32_t vis = (uint32_t)d->materials[mat].visibility & 3u;
		float mw; const uint32_t packed = mat | (vis << 30); std::memcpy(&mw, &packed, 4);
		rec[3 * i] = make_float4(v[0], v[1], v[2], eps);
		rec[3 * i + 1] = make_float4(e1[0], e1[1], e1[2], mw);
		rec[3 * i + 2] = make_float4(e2[0], e2[1], e2[2], 0.f);
		float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
		float len = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
		if(len != 0.f) { len = 1.0f / std::sqrt(len); n[0] *= len; n[1] *= len; n[2] *= len; }
		bool smooth = false;
		if(d->vnormals)
		{
			const float *q = d->vnormals + 9 * i;
			for(int k = 0; k < 9; ++k) if(q[k] != 0.f) smooth = true;
		}
		float sw; const uint32_t sbits = smooth ? 1u : 0u; std::memcpy(&sw, &sbits, 4);
		ng[i] = make_float4(n[0], n[1], n[2], sw);
		any_smooth |= smooth;
	}
	if(any_smooth)
	{
		vn.resize(3 * nt);
		for(size_t i = 0; i < nt; ++i)
		{
			const float *q = d->vnormals + 9 * i;
			for(int c = 0; c < 3; ++c)
			{
				float x = q[3 * c], y = q[3 * c + 1], z = q[3 * c + 2];
				if(x == 0.f && y == 0.f && z == 0.f) { x = ng[i].x; y = ng[i].y; z = ng[i].z; }
				vn[3 * i + (size_t)c] = make_float4(x, y, z, 0.f);
			}
		}
	}
	// QMC tables
	std::vector<int> faure; std::vector<int> foff(50); std::vector<double> invp(50);
	for(int dim = 0; dim < 50; ++dim)
	{
		foff[(size_t)dim] = (int)faure.size();
		std::vector<int> p;
		if(dim < 3) p = {0, 1, 2};      // faure_tables.cc:437: dims 0-2 share the length-3 identity
		else faure_perm(kPrimsHost[dim], p);
		faure.insert(faure.end(), p.begin(), p.end());
		char buf[32];
		std::snprintf(buf, sizeof buf, "%.9f", 1.0 / (double)kPrimsHost[dim]); // scr_halton.h:37-47 prints 9 decimals
		invp[(size_t)dim] = std::strtod(buf, nullptr);
	}
	int rc = 0;
	DevScene &dv = s->dev;
	const uint2 *nodes = nullptr;
	{	// the traversal kernels fetch a window of 8 nodes starting at the current one: pad the array so the last window stays in bounds
		std::vector<KdNode> padded(s->tree.nodes);
		padded.resize(padded.size() + 8, KdNode{0u, 3u});
		if((rc = upload(s, (const uint2 *)padded.data(), padded.size(), &nodes))) { yafgpu_scene_destroy(s); return rc; }
	}
	dv.nodes = nodes;
	dv.nodes2 = nullptr;
#if YAFGPU_TRACE_PAIR
	{
		const std::vector<KdNode> &tn = s->tree.nodes;
		std::vector<uint4> pairs(tn.size() + 8, make_uint4(0u, 3u, 0u, 3u));
		for(size_t i = 0; i < tn.size(); ++i)
		{
			const uint2 nd = *(const uint2 *)&tn[i];
			uint4 pr = make_uint4(nd.x, nd.y, 0u, 3u);
			if((nd.y & 3u) != 3u) { const uint2 r = *(const uint2 *)&tn[nd.y >> 2]; pr.z = r.x; pr.w = r.y; }
			pairs[i] = pr;
		}
		const uint4 *d_pairs = nullptr;
		if((rc = upload(s, pairs.data(), pairs.size(), &d_pairs))) { yafgpu_scene_destroy(s); return rc; }
		dv.nodes2 = d_pairs;
	}
#endif
	dv.top = nullptr;
#if YAFGPU_TRACE_TOP
	if(!s->tree.nodes.empty())
	{	// the top of the tree once more, in heap order: every ray's first steps touch these few nodes (a breadth-first prefix is 128
		// cache lines at depth 10; in the depth-first array the same nodes are spread over as many lines as there are left spines)
		const std::vector<KdNode> &tn = s->tree.nodes;
		std::vector<uint4> top((size_t)kTopN, make_uint4(0u, 3u, 0u, 0u));
		std::vector<std::pair<uint32_t, uint32_t>> todo;      // (index in the depth-first array, heap index)
		todo.emplace_back(0u, 0u);
		while(!todo.empty())
		{
			const auto [g, h] = todo.back(); todo.pop_back();
			const uint2 nd = *(const uint2 *)&tn[g];
			top[h] = make_uint4(nd.x, nd.y, g, 0u);
			if((nd.y & 3u) != 3u && 2u * h + 2u < (uint32_t)kTopN) { todo.emplace_back(nd.y >> 2, 2u * h + 2u); todo.emplace_back(g + 1u, 2u * h + 1u); }
		}
		if(tn.size() >= kTopTag) { yafgpu_scene_destroy(s); return fail(-2, "kd-tree too large for tagged node indices"); }
		if((rc = upload(s, top.data(), top.size(), &dv.top))) { yafgpu_scene_destroy(s); return rc; }
	}
#endif
	dv.nodes_blk = nullptr;
#if YAFGPU_TRACE_BLOCKS
	if(!s->tree.nodes.empty())
	{	// the depth-first array (left child = next node, right child in the node) laid out again in blocks of three levels
		const std::vector<KdNode> &tn = s->tree.nodes;
		std::vector<uint2> blk(8, make_uint2(0u, 3u));
		std::vector<std::pair<uint32_t, uint32_t>> todo;      // (depth-first index of a subtree's root, its block)
		todo.emplace_back(0u, 0u);
		while(!todo.empty())
		{
			const auto [root, b] = todo.back(); todo.pop_back();
			uint32_t at[7]; bool have[7] = {true, false, false, false, false, false, false};
			at[0] = root;
			for(int sl = 0; sl < 7; ++sl)
			{
				if(!have[sl]) continue;
				const uint2 nd = *(const uint2 *)&tn[at[sl]];
				uint2 out = nd;
				if((nd.y & 3u) != 3u)
				{
					const uint32_t left = at[sl] + 1u, right = nd.y >> 2;
					if(sl < 3) { at[2 * sl + 1] = left; at[2 * sl + 2] = right; have[2 * sl + 1] = have[2 * sl + 2] = true; out.y = nd.y & 3u; }
					else
					{
						const uint32_t bl = (uint32_t)(blk.size() / 8);
						blk.resize(blk.size() + 16, make_uint2(0u, 3u));
						out.y = (nd.y & 3u) | (bl << 2);
						todo.emplace_back(right, bl + 1u);
						todo.emplace_back(left, bl);
					}
				}
				blk[(size_t)b * 8 + (size_t)sl] = out;
			}
		}
		if(blk.size() / 8 >= (1u << 27)) { yafgpu_scene_destroy(s); return fail(-2, "kd-tree too large for the block layout"); }
		if((rc = upload(s, blk.data(), blk.size(), &dv.nodes_blk))) { yafgpu_scene_destroy(s); return rc; }
	}
#endif
	if((rc = upload(s, s->tree.refs.data(), s->tree.refs.size(), &dv.refs))) { yafgpu_scene_destroy(s); return rc; }
	if((rc = upload(s, rec.data(), rec.size(), &dv.tri))) { yafgpu_scene_destroy(s); return rc; }
	if((rc = upload(s, ng.data(), ng.size(), &dv.tri_ng))) { yafgpu_scene_destroy(s); return rc; }
	dv.tri_vn = nullptr;
	if(any_smooth && (rc = upload(s, vn.data(), vn.size(), &dv.tri_vn))) { yafgpu_scene_destroy(s); return rc; }
	if((rc = upload(s, d->materials, (size_t)d->n_materials, &dv.mats))) { yafgpu_scene_destroy(s); return rc; }
	if((rc = upload(s, d->lights, (size_t)d->n_lights, &dv.lights))) { yafgpu_scene_destroy(s); return rc; }
	std::memset(&dv.tex, 0, sizeof dv.tex);
	if(d->n_nodes > 0 && d->nodes)
	{	// shader nodes, image textures and the per-triangle texture coordinates they read
		for(int i = 0; i < d->n_materials; ++i)
		{
			const yafgpu_material &m = d->materials[i];
			if(m.n_nodes < 0 || m.n_nodes > kMaxNodes || m.node_first < 0 || m.node_first + m.n_nodes > d->n_nodes)
			{ yafgpu_scene_destroy(s); return fail(-24, "a material's shader nodes: more than " + std::to_string(kMaxNodes) + " nodes, or a range outside the node array"); }
			if(m.n_nodes > 0) s->has_textures = true;
			if(m.n_bump < 0 || m.n_bump > kMaxNodes || (m.n_bump > 0 && (m.bump_first < 0 || m.bump_first + m.n_bump > d->n_nodes || m.sh_bump < 0 || m.sh_bump >= m.n_bump)))
			{ yafgpu_scene_destroy(s); return fail(-24, "a material's bump shader: more than " + std::to_string(kMaxNodes) + " nodes, or a range outside the node array"); }
			if(m.n_bump > 0) { s->has_textures = true; s->has_bump = true; }
			// every reference inside the material — a shader slot, a node's inputs — is -1 or names a node BEFORE the one that reads it
			// (evaluation order): the device indexes a per-lane result stack with them (nodes_eval, mat_resolve, nodes_eval_derivative)
			const int slots[] = {m.sh_diffuse, m.sh_mirror_color, m.sh_mirror, m.sh_transparency, m.sh_translucency, m.sh_sigma_oren, m.sh_diffuse_refl, m.sh_ior,
			                     m.sh_glossy, m.sh_glossy_reflect, m.sh_exponent, m.sh_filter_color};
			for(int sl : slots)
				if(m.n_nodes > 0 && (sl < -1 || sl >= m.n_nodes)) { yafgpu_scene_destroy(s); return fail(-24, "a material's shader slot names a node outside its node range"); }
			auto range_ok = [&](int first, int count) {
				for(int k = 0; k < count; ++k)
				{
					const yafgpu_node &n = d->nodes[first + k];
					auto ref_ok = [&](int r) { return r >= -1 && r < k; };
					if(n.type == YAFGPU_NODE_MIX && !(ref_ok(n.input1) && ref_ok(n.input2) && ref_ok(n.factor))) return false;
					if(n.type == YAFGPU_NODE_LAYER && !(n.input >= 0 && n.input < k && ref_ok(n.upper))) return false;
					if(n.type < YAFGPU_NODE_TEXTURE_MAPPER || n.type > YAFGPU_NODE_LAYER) return false;
				}
				return true;
			};
			if(!range_ok(m.node_first, m.n_nodes) || (m.n_bump > 0 && !range_ok(m.bump_first, m.n_bump)))
			{ yafgpu_scene_destroy(s); return fail(-24, "a shader node refers to a node that is not evaluated before it (or has an unknown type; a layer needs an input)"); }
		}
		for(int i = 0; i < d->n_nodes; ++i)
		{
			const yafgpu_node &n = d->nodes[i];
			if(n.type == YAFGPU_NODE_TEXTURE_MAPPER && (n.texture < 0 || n.texture >= d->n_textures)) { yafgpu_scene_destroy(s); return fail(-24, "a texture_mapper node refers to a texture that does not exist"); }
		}
		for(int i = 0; i < d->n_textures; ++i)
		{
			const yafgpu_texture &t = d->textures[i];
			if(t.width <= 0 || t.height <= 0 || (uint64_t)t.texel_first + (uint64_t)t.width * (uint64_t)t.height > d->n_texels) { yafgpu_scene_destroy(s); return fail(-24, "a texture's texel range lies outside the texel array"); }
		}
		const float4 *texels = nullptr;
		if((rc = upload(s, d->nodes, (size_t)d->n_nodes, &dv.tex.nodes))) { yafgpu_scene_destroy(s); return rc; }
		if((rc = upload(s, d->textures, (size_t)std::max(d->n_textures, 0), &dv.tex.textures))) { yafgpu_scene_destroy(s); return rc; }
		if((rc = upload(s, (const float4 *)d->texels, (size_t)d->n_texels, &texels))) { yafgpu_scene_destroy(s); return rc; }
		dv.tex.texels = texels; dv.tex.n_textures = d->n_textures;
		if(d->tri_uv && (rc = upload(s, d->tri_uv, nt * 6, &dv.tex.tri_uv))) { yafgpu_scene_destroy(s); return rc; }
		if(d->tri_orco && (rc = upload(s, d->tri_orco, nt * 9, &dv.tex.tri_orco))) { yafgpu_scene_destroy(s); return rc; }
		if(s->has_bump)
		{	// the third edge of Triangle::getSurface's dPdU / dPdV (triangle.cc:80-111): c - b, rounded once like e1 and e2 of the record
			std::vector<float> e3(nt * 3);
			for(size_t i = 0; i < nt; ++i) for(int k = 0; k < 3; ++k) e3[3 * i + k] = d->verts[9 * i + 6 + k] - d->verts[9 * i + 3 + k];
			if((rc = upload(s, e3.data(), e3.size(), &dv.tex.tri_e3))) { yafgpu_scene_destroy(s); return rc; }
			dv.tex.has_bump = 1;
		}
	}
	if((rc = upload(s, faure.data(), faure.size(), &dv.faure))) { yafgpu_scene_destroy(s); return rc; }
	dv.n_faure = (int)faure.size();
	if((rc = upload(s, foff.data(), foff.size(), &dv.faure_off))) { yafgpu_scene_destroy(s); return rc; }
	if((rc = upload(s, invp.data(), invp.size(), &dv.inv_prims))) { yafgpu_scene_destroy(s); return rc; }
	dv.n_lights = d->n_lights; dv.n_tris = d->n_tris; dv.n_mats = d->n_materials; dv.n_nodes = (uint32_t)s->tree.nodes.size();
	for(int k = 0; k < 3; ++k) { dv.blo[k] = s->tree.bound_lo[k]; dv.bhi[k] = s->tree.bound_hi[k]; }
	dv.cam = d->camera;
	{	// PerspectiveCamera ctor, camera_perspective.cc:42-54: corner table of the polygonal bokeh shapes
		for(float &v : dv.cam.ls) v = 0.f;
		int ns = dv.cam.bokeh_type;
		if(ns >= 3 && ns <= 6)
		{
			float w = (float)((double)dv.cam.bokeh_rotation * 0.01745329251994329576922);
			const float wi = (float)(6.28318530717958647692 / (double)(float)ns);
			ns = (ns + 2) * 2;
			for(int i = 0; i < ns; i += 2)
			{
				dv.cam.ls[i] = host_fsin(w + (float)1.57079632679489661923);     // fCos__
				dv.cam.ls[i + 1] = host_fsin(w);
				w += wi;
			}
		}
	}
	s->mats.assign(d->materials, d->materials + d->n_materials);
	s->n_lights = d->n_lights;
	s->h_lights.assign(d->lights, d->lights + d->n_lights);
	s->info.upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
	(void)t0;
	*out = s;
	return 0;
}

void yafgpu_scene_destroy(yafgpu_scene_t *s)
{
	if(!s) return;
	for(void *p : s->allocs) (void)hipFree(p);
	if(s->d_tiles) (void)hipFree(s->d_tiles);
	if(s->d_prefix) (void)hipFree(s->d_prefix);
	if(s->d_queue) (void)hipFree(s->d_queue);
	if(s->d_pix_prefix) (void)hipFree(s->d_pix_prefix);
	if(s->wf_state) (void)hipFree(s->wf_state);
	if(s->wf_results) (void)hipFree(s->wf_results);
	if(s->wf_queues) (void)hipFree(s->wf_queues);
	if(s->wf_counts) (void)hipFree(s->wf_counts);
	if(s->wf_verdict) (void)hipFree(s->wf_verdict);
	if(s->wf_pix_xy) (void)hipFree(s->wf_pix_xy);
	if(s->wf_filt) (void)hipFree(s->wf_filt);
	for(yafgpu_scene::WfSet &o : s->alt)
	{
		for(void *q : {(void *)o.d_pix_prefix, (void *)o.wf_state, (void *)o.wf_results, (void *)o.wf_filt, (void *)o.wf_queues, (void *)o.wf_counts, (void *)o.wf_verdict, (void *)o.wf_pix_xy})
			if(q) (void)hipFree(q);
		if(o.side_stream) (void)hipStreamDestroy(o.side_stream);
		if(o.ev_fork) (void)hipEventDestroy(o.ev_fork);
		if(o.ev_join) (void)hipEventDestroy(o.ev_join);
	}
	for(int k = 0; k < yafgpu_scene::kPipeMax; ++k)
	{
		if(s->pipe_stream[k]) (void)hipStreamDestroy(s->pipe_stream[k]);
		if(s->pipe_done[k]) (void)hipEventDestroy(s->pipe_done[k]);
		if(s->pipe_acc[k]) (void)hipEventDestroy(s->pipe_acc[k]);
	}
	if(s->pipe_sync) (void)hipEventDestroy(s->pipe_sync);
	for(int k = 0; k < yafgpu_scene::kPipeMax; ++k) if(s->pipe_counters[k]) (void)hipFree(s->pipe_counters[k]);
	if(s->side_stream) (void)hipStreamDestroy(s->side_stream);
	if(s->ev_fork) (void)hipEventDestroy(s->ev_fork);
	if(s->ev_join) (void)hipEventDestroy(s->ev_join);
	if(s->d_filter_table) (void)hipFree(s->d_filter_table);
	for(void *q : {(void *)s->rp_flags, (void *)s->rp_p, (void *)s->rp_kill, (void *)s->rp_calls, (void *)s->rp_base, (void *)s->rp_seg_begin,
	               (void *)s->rp_seg_seed, (void *)s->rp_seg_total, (void *)s->rp_seg_base, (void *)s->rp_counter, (void *)s->rp_hits,
	               (void *)s->rt_planes, (void *)s->rt_film, (void *)s->rt_cnt, (void *)s->rt_flags}) if(q) (void)hipFree(q);
	delete s;
}

int yafgpu_scene_info(const yafgpu_scene_t *s, yafgpu_tree_info *info)
{
	if(!s || !info) return fail(-1, "null argument");
	*info = s->info;
	return 0;
}

int yafgpu_scene_get_tree(const yafgpu_scene_t *s, uint32_t *nodes, uint32_t *refs, float bound6[6])
{
	if(!s) return fail(-1, "null argument");
	if(nodes) std::memcpy(nodes, s->tree.nodes.data(), s->tree.nodes.size() * sizeof(KdNode));
	if(refs) std::memcpy(refs, s->tree.refs.data(), s->tree.refs.size() * sizeof(uint32_t));
	if(bound6) for(int k = 0; k < 3; ++k) { bound6[k] = s->tree.bound_lo[k]; bound6[3 + k] = s->tree.bound_hi[k]; }
	return 0;
}

// ---- reconstruction-filter table on the host: ImageFilm ctor + filter functions, src/common/imagefilm.cc:60-121,152-176.
// fExp__/fSin__ are the reference's polynomial approximations (util_math_optimizations.h:116-129,222-244), restated
// for the host so that the table holds the values the reference's film would hold.
static float host_fexp2(float x)
{
	x = std::min(x, 129.00000f);
	x = std::max(x, -126.99999f);
	const int ipart = (int)(x - 0.5f);
	const float p = (x - (float)ipart);
	int bits = (int)((unsigned)(ipart + 127) << 23);
	float expi; std::memcpy(&expi, &bits, 4);
	const float poly = (p * (p * (p * (p * (p * 1.8775767e-3f + 8.9893397e-3f) + 5.5826318e-2f) + 2.4015361e-1f) + 6.9315308e-1f) + 9.9999994e-1f);
	return expi * poly;
}
static float host_fsin(float x)
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
static float host_filter(int type, float dx, float dy)
{
	switch(type)
	{
		case YAFGPU_FILTER_MITCHELL:
		{
			const float x = 2.f * std::sqrt(dx * dx + dy * dy);
			if(x >= 2.f) return 0.f;
			if(x >= 1.f) return (float)(x * (x * (x * -0.38888889f + 2.0f) - 3.33333333f) + 1.77777778f);
			return (float)(x * x * (1.16666666f * x - 2.0f) + 0.88888889f);
		}
		case YAFGPU_FILTER_GAUSS:
		{
			const float r_2 = dx * dx + dy * dy;
			const float e = host_fexp2((float)1.4426950408889634074 * (float)(-6 * r_2));
			return std::max(0.f, (float)((double)e - 0.00247875));
		}
		case YAFGPU_FILTER_LANCZOS:
		{
			const float x = std::sqrt(dx * dx + dy * dy);
			if(x == 0.f) return 1.f;
			if(-2 < x && x < 2)
			{
				const float a = (float)(3.14159265358979323846 * (double)x), b = (float)(1.57079632679489661923 * (double)x);
				return ((host_fsin(a) * host_fsin(b)) / (a * b));
			}
			return 0.f;
		}
		default: return 1.f;
	}
}
static void host_filter_table(int type, float table[256])
{
	const float scale = 1.f / 16.f;
	for(int y = 0; y < 16; ++y)
		for(int x = 0; x < 16; ++x) table[y * 16 + x] = host_filter(type, (x + .5f) * scale, (y + .5f) * scale);
}

static int validate(const yafgpu_scene *s, const yafgpu_render_params *rp)
{
	if(rp->width <= 0 || rp->height <= 0 || rp->aa_minsamples <= 0 || rp->tile_size <= 0) return fail(-10, "empty image, sample count or tile size");
	if(rp->xstart < 0 || rp->ystart < 0 || rp->xstart + rp->width > 65535 || rp->ystart + rp->height > 65535) return fail(-10, "render window outside [0, 65535) pixels");
	if(rp->bounces > 12) return fail(-11, "bounces > 12 would use scrHalton dimensions >= 50, which are a global racy LCG in the reference (scr_halton.h:70-73)");
	if(rp->filter_type < YAFGPU_FILTER_BOX || rp->filter_type > YAFGPU_FILTER_LANCZOS) return fail(-12, "unknown filter type");
	if(rp->shard_count < 1 || rp->shard_index < 0 || rp->shard_index >= rp->shard_count) return fail(-13, "bad shard index/count");
	if(rp->integrator != YAFGPU_INTEGRATOR_PATH && rp->integrator != YAFGPU_INTEGRATOR_DIRECT) return fail(-14, "unknown integrator");
	if(rp->path_samples > 8191) return fail(-11, "path_samples > 8191: the device path counts a level's path samples in 13 bits of the control word");
	// the sample index of a light estimate in flight is packed in 12 bits (pack_dlc); validate() runs for every pass, so a
	// light-sample multiplier that grows over the passes of an adaptive render is caught when it gets there
	for(const yafgpu_light &l : s->h_lights)
		if(l.type != YAFGPU_LIGHT_POINT && std::ceil((float)l.samples * rp->aa_light_sample_multiplier) > 4095.f)
			return fail(-19, "an area light with more than 4095 samples per estimate (samples x AA light-sample multiplier): the device path counts them in 12 bits");
	return 0;
}

// ---- wavefront pass -------------------------------------------------------------------------
static constexpr uint32_t kWfMaxPaths = 32u << 20;   // paths in flight per chunk: 32 Mi x 304 B = 9.5 GiB of parked state

static int wf_grid(const void *kernel, int cus)
{
	// (the occupancy of a kernel does not change between passes: asked once per kernel)
	static std::mutex mu; static std::map<const void *, int> known;
	int per_cu = 0;
	{
		std::lock_guard<std::mutex> lock(mu);
		auto it = known.find(kernel);
		if(it != known.end()) per_cu = it->second;
		else
		{
			if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
			known[kernel] = per_cu;
		}
	}
	int cap = 8;
	if(const char *e = std::getenv("YAFGPU_BLOCKS_PER_CU")) cap = std::max(1, std::atoi(e));   // occupancy experiments
	return cus * std::min(per_cu, cap);
}

// Scene-specialised builds of wf_shade (yafgpu_shade_variant.hip), most specialised first.  A variant serves a scene
// whose material types are a subset of its mask and which needs recursiveRaytrace only if the variant has it; every
// other scene takes the general kernel of this unit.  YAFGPU_SHADE_VARIANT=general forces the general kernel.
extern "C" {
#define YG_DECLARE_SHADE_VARIANT(name) \
	void yafgpu_shade_##name##_describe(uint32_t *, int *, int *, int *); const void *yafgpu_shade_##name##_kernel(); \
	int yafgpu_shade_##name##_launch(const void *, size_t, int, hipStream_t);
YG_DECLARE_SHADE_VARIANT(diffuse)
YG_DECLARE_SHADE_VARIANT(glossy)
YG_DECLARE_SHADE_VARIANT(diffuse_mp)
YG_DECLARE_SHADE_VARIANT(glossy_mp)
YG_DECLARE_SHADE_VARIANT(diffuse_rec)
YG_DECLARE_SHADE_VARIANT(glossy_rec)
YG_DECLARE_SHADE_VARIANT(full)
#undef YG_DECLARE_SHADE_VARIANT
}
struct ShadeVariant
{
	const char *name;
	void (*describe)(uint32_t *, int *, int *, int *);
	const void *(*kernel)();
	int (*launch)(const void *, size_t, int, hipStream_t);
};
static const ShadeVariant kShadeVariants[] = {
	{"diffuse", yafgpu_shade_diffuse_describe, yafgpu_shade_diffuse_kernel, yafgpu_shade_diffuse_launch},
	{"glossy", yafgpu_shade_glossy_describe, yafgpu_shade_glossy_kernel, yafgpu_shade_glossy_launch},
	// (with a second MIS pair per park, YAFGPU_FEAT_MULTI: for scenes whose light estimates have one to offer)
	{"diffuse_mp", yafgpu_shade_diffuse_mp_describe, yafgpu_shade_diffuse_mp_kernel, yafgpu_shade_diffuse_mp_launch},
	{"glossy_mp", yafgpu_shade_glossy_mp_describe, yafgpu_shade_glossy_mp_kernel, yafgpu_shade_glossy_mp_launch},
	// (the programs of a serial-state replay's record pass: no light estimate, YAFGPU_FEAT_LIGHTS=0)
	{"diffuse_rec", yafgpu_shade_diffuse_rec_describe, yafgpu_shade_diffuse_rec_kernel, yafgpu_shade_diffuse_rec_launch},
	{"glossy_rec", yafgpu_shade_glossy_rec_describe, yafgpu_shade_glossy_rec_kernel, yafgpu_shade_glossy_rec_launch},
	{"full", yafgpu_shade_full_describe, yafgpu_shade_full_kernel, yafgpu_shade_full_launch},      // everything but shader nodes
};
static const ShadeVariant *pick_shade_variant(const yafgpu_scene *s, int frames, bool record_pass = false, bool want_multi = false)
{
	if(const char *e = std::getenv("YAFGPU_SHADE_VARIANT")) if(std::strcmp(e, "general") == 0) return nullptr;
	if(record_pass) if(const char *e = std::getenv("YAFGPU_RECORD_VARIANT")) if(std::atoi(e) == 0) return nullptr;      // (A/B: the record pass on the pass's own kernel)
	const bool needs_recurse = frames > 0 || s->has_volumetric;
	if(s->has_textures || s->has_aniso) return nullptr;        // the variants are built without shader nodes and without the anisotropic lobe
	for(const ShadeVariant &v : kShadeVariants)
	{
		uint32_t mask = 0u; int recurse = 0, lights = 1, multi = 0;
		v.describe(&mask, &recurse, &lights, &multi);
		if((lights == 0) != record_pass) continue;
		if(!record_pass && (multi != 0) != want_multi) continue;
		if((s->mat_mask & ~mask) == 0u && (recurse || !needs_recurse)) return &v;
	}
	return nullptr;
}

// Serial-state replay (WfArgs::replay): wanted when the reference's serial state is consumed at all — a roulette test
// can happen (some depth in [1, bounces) lies above russian_roulette_min_bounces) or estimateOneDirectLight has a choice
// (more than one light).  With recursiveRaytrace a sample's events are a tree of integrate() calls, kept per call in the depth-first
// order the reference walks it (up to 255 calls per sample; beyond that the per-sample streams stand in).  The light counter of a SHARDED frame needs every rank's calls per tile (a tile
// starts with the sum over all tiles before it): with an exchange function attached the ranks share them (lc_sharded),
// without one the per-sample ordinals stand in for the counter.
struct ReplayPlan { int frames; bool need_rr, need_lc, replay, replay_lights, lc_sharded; int ev_m; };
static ReplayPlan replay_plan(const yafgpu_scene *s, const yafgpu_render_params &rp)
{
	ReplayPlan p{};
	p.frames = ((s->has_specular || s->has_glossy) && rp.raydepth + s->max_add_depth > 0) ? rp.raydepth + s->max_add_depth : 0;
	const bool path = rp.integrator == YAFGPU_INTEGRATOR_PATH;
	p.need_rr = path && rp.bounces - 1 > rp.rr_min_bounces;
	p.need_lc = path && s->n_lights > 1;
	// integrate() calls one camera sample can make (WfArgs::ev_m): without glossy-recursive materials every call sends at most a
	// reflected and a transmitted ray, frames levels deep; with them a call whose trajectory splitting is still 1 also sends 8 glossy
	// trajectories (two rays each through rough glass), whose calls (division >= 8: one trajectory each) send at most 3.  The ordinal has 8 bits.
	{
		long long t = 1, g = 1;
		const long long per_traj = s->has_glossy_two ? 2 : 1;
		for(int k = 0; k < p.frames; ++k) { const long long t2 = s->has_glossy ? 1 + 8 * per_traj * g + 2 * t : 1 + 2 * t; g = 1 + 3 * g; t = std::min<long long>(t2, 1 << 20); }
		p.ev_m = (int)t;
	}
	p.replay = rp.serial_replay != 0 && p.ev_m <= 255 && (p.need_rr || p.need_lc);
	if(const char *e = std::getenv("YAFGPU_SERIAL_REPLAY")) if(std::atoi(e) == 0) p.replay = false;
	p.lc_sharded = p.replay && p.need_lc && rp.shard_count > 1 && s->exchange != nullptr;
	p.replay_lights = p.replay && p.need_lc && (rp.shard_count == 1 || p.lc_sharded);
	if(p.replay && !p.need_rr && !p.replay_lights) p.replay = false;
	if(!p.replay) { p.replay_lights = false; p.lc_sharded = false; }
	return p;
}
// The light counter across ranks: every rank contributes the estimateOneDirectLight calls of its own tiles (global tile index,
// count), the exchange function sums the table over the ranks (each entry has one writer; two 16-bit halves as floats, so the sums
// are exact), and every rank takes the same exclusive scan in the reference's tile order on top of the counter so far.  Every rank
// of the render calls this once per pass, with or without tiles of its own.
static int lc_exchange_counts(yafgpu_scene *s, const yafgpu_render_params &rp, const std::vector<std::pair<int, uint32_t>> &own, std::vector<uint32_t> *base_of_tile)
{
	const int ntx = (rp.width + rp.tile_size - 1) / rp.tile_size, nty = (rp.height + rp.tile_size - 1) / rp.tile_size;
	const size_t n = (size_t)ntx * (size_t)nty;
	std::vector<float> h(2 * n, 0.f);
	for(const auto &e : own) { h[2 * (size_t)e.first] = (float)(e.second & 0xffffu); h[2 * (size_t)e.first + 1] = (float)(e.second >> 16); }
	DevMem<float> d;
	HIP_OK(d.alloc(2 * n));
	HIP_OK(hipMemcpy(d, h.data(), 2 * n * sizeof(float), hipMemcpyHostToDevice));
	HIP_OK(hipDeviceSynchronize());
	if(s->exchange(s->exchange_user, d, (uint64_t)(2 * n))) return fail(-31, "the exchange function reported a failure (light counter)");
	HIP_OK(hipMemcpy(h.data(), d, 2 * n * sizeof(float), hipMemcpyDeviceToHost));
	uint32_t run = rp.accumulate ? s->lc_host_counter : 0u;      // zeroed once per render, before its first pass (integrator_tiled.cc:192-194)
	if(base_of_tile) base_of_tile->resize(n);
	for(size_t t = 0; t < n; ++t)
	{
		if(base_of_tile) (*base_of_tile)[t] = run;
		run += (uint32_t)h[2 * t] + ((uint32_t)h[2 * t + 1] << 16);
	}
	s->lc_host_counter = run;
	return 0;
}

__global__ void add_counters(yafgpu_counters *dst, const yafgpu_counters *src)
{
	const unsigned i = threadIdx.x;
	if(i < sizeof(yafgpu_counters) / sizeof(uint64_t)) ((uint64_t *)dst)[i] += ((const uint64_t *)src)[i];
}
static void swap_wf_sets(yafgpu_scene *s, int which)
{
	yafgpu_scene::WfSet &o = s->alt[which];
	std::swap(s->d_pix_prefix, o.d_pix_prefix); std::swap(s->pix_prefix_cap, o.pix_prefix_cap);
	std::swap(s->wf_state, o.wf_state); std::swap(s->wf_results, o.wf_results); std::swap(s->wf_filt, o.wf_filt);
	std::swap(s->wf_queues, o.wf_queues); std::swap(s->wf_counts, o.wf_counts); std::swap(s->wf_verdict, o.wf_verdict); std::swap(s->wf_pix_xy, o.wf_pix_xy);
	std::swap(s->wf_cap, o.wf_cap); std::swap(s->wf_filt_cap, o.wf_filt_cap); std::swap(s->wf_frames, o.wf_frames);
	std::swap(s->side_stream, o.side_stream); std::swap(s->ev_fork, o.ev_fork); std::swap(s->ev_join, o.ev_join);
}

static int render_wavefront(yafgpu_scene *s, RenderArgs &ra, hipStream_t caller, bool stats)
{
	hipStream_t stream = caller;
	const yafgpu_render_params &rp = ra.rp;
	const uint32_t spp = (uint32_t)rp.aa_minsamples;
	// per-tile pixel prefix (same tile list as the unit prefix)
	std::vector<uint32_t> &pp = s->h_pix_prefix;
	pp.assign(1, 0u);
	for(const int4 &r : s->h_tiles) pp.push_back(pp.back() + (uint32_t)(r.z * r.w));
	// a resample mask (adaptive pass): the pixels of this shard's tiles that are flagged, in tile order
	std::vector<uint32_t> &listed = s->h_listed;        // scene-owned: uploaded per chunk with async copies
	std::vector<uint32_t> listed_prefix(1, 0u);
	const bool masked = rp.resample_mask != nullptr;
	if(masked)
	{
		HIP_OK(hipStreamSynchronize(stream));             // the previous pass's uploads of the list have been read
		listed.clear();
		for(const int4 &r : s->h_tiles)
		{
			for(int y = r.y; y < r.y + r.w; ++y)
				for(int x = r.x; x < r.x + r.z; ++x)
					if(rp.resample_mask[(size_t)(y - rp.ystart) * (size_t)rp.width + (size_t)(x - rp.xstart)]) listed.push_back((uint32_t)x | ((uint32_t)y << 16));
			listed_prefix.push_back((uint32_t)listed.size());
		}
		if(listed.empty()) return replay_plan(s, rp).lc_sharded ? lc_exchange_counts(s, rp, {}, nullptr) : 0;      // (the other ranks wait for this one's counts)
	}
	const std::vector<uint32_t> &tile_px = masked ? listed_prefix : pp;      // pixels of the pass before every tile of the shard
	const uint32_t n_pixels_total = tile_px.back();
	// Pass pipelining: this pass's path work goes to one of two internal streams with its own buffer set and does NOT wait for what the
	// caller's stream holds (the previous pass, its film combine, a reduce); only the film accumulation is put on the caller's stream, after
	// the path work.  Eligible: one chunk, no serial-state replay (its tables are per scene), no recursion polling, no resample mask (the
	// host reads the film between such passes anyway), no per-kernel profiling.
	bool pass_piped = false; int pipe_k = 0;
	struct SwapBack { yafgpu_scene *s; int which; ~SwapBack() { if(which >= 0) swap_wf_sets(s, which); } } swap_back{s, -1};
	{
		// measured (profiles/r03_ab_pipeline.txt): +18 % at an eighth of the metric frame, +10 % at a quarter, +4 % at half, -3 % at the whole
		// frame (two full-size passes only compete) -- on by size, like the two traversal launches side by side
		bool want = s->pass_pipelining < 0 ? (uint64_t)n_pixels_total * spp <= (12ull << 20) : s->pass_pipelining != 0;
		if(const char *e = std::getenv("YAFGPU_PASS_PIPELINE")) want = std::atoi(e) != 0;
		uint32_t mp = kWfMaxPaths;
		if(const char *e = std::getenv("YAFGPU_WF_CHUNK")) mp = std::max(256u, (uint32_t)std::strtoul(e, nullptr, 10));
		const ReplayPlan pl = replay_plan(s, rp);
		pass_piped = want && !masked && !stats && !s->profiling && !pl.replay && pl.frames == 0 && (uint64_t)n_pixels_total * spp <= mp && !std::getenv("YAFGPU_CHUNK_PIPELINE");
		if(pass_piped)
		{
			int depth = 2;      // (a third pass in flight adds nothing: two chains already keep a launch beside every tail, profiles/r03_ab_pipeline.txt)
			if(const char *e = std::getenv("YAFGPU_PASS_PIPELINE_DEPTH")) depth = std::min(std::max(std::atoi(e), 2), (int)yafgpu_scene::kPipeMax);
			for(int k = 0; k < depth; ++k) if(!s->pipe_stream[k])
			{
				HIP_OK(hipStreamCreateWithFlags(&s->pipe_stream[k], hipStreamNonBlocking));
				HIP_OK(hipEventCreateWithFlags(&s->pipe_done[k], hipEventDisableTiming));
				HIP_OK(hipEventCreateWithFlags(&s->pipe_acc[k], hipEventDisableTiming));
			}
			if(!s->pipe_sync) HIP_OK(hipEventCreateWithFlags(&s->pipe_sync, hipEventDisableTiming));
			if(!s->pipe_prev)
			{	// the first of a run (or new tile arrays): what the caller's stream holds so far precedes both internal streams, once
				HIP_OK(hipEventRecord(s->pipe_sync, caller));
				for(int k = 0; k < depth; ++k) HIP_OK(hipStreamWaitEvent(s->pipe_stream[k], s->pipe_sync, 0));
			}
			if(s->pipe_next >= depth) s->pipe_next = 0;
			pipe_k = s->pipe_next; s->pipe_next = (s->pipe_next + 1) % depth;
			if(pipe_k > 0) { swap_wf_sets(s, pipe_k - 1); swap_back.which = pipe_k - 1; }
			stream = s->pipe_stream[pipe_k];
			// the set's previous pass has been added to the film (its results and pixel list are free again)
			if(s->pipe_acc_set[pipe_k]) HIP_OK(hipStreamWaitEvent(stream, s->pipe_acc[pipe_k], 0));
			if(ra.counters)
			{
				if(!s->pipe_counters[pipe_k]) HIP_OK(hipMalloc((void **)&s->pipe_counters[pipe_k], sizeof(yafgpu_counters)));
				HIP_OK(hipMemsetAsync(s->pipe_counters[pipe_k], 0, sizeof(yafgpu_counters), stream));
			}
		}
		s->pipe_prev = pass_piped;
	}
	yafgpu_counters *const caller_counters = ra.counters;
	if(pass_piped && ra.counters) ra.counters = s->pipe_counters[pipe_k];
	if(pp.size() > s->pix_prefix_cap)
	{
		HIP_OK(hipStreamSynchronize(stream));
		if(s->d_pix_prefix) (void)hipFree(s->d_pix_prefix);
		s->pix_prefix_cap = pp.size();
		HIP_OK(hipMalloc((void **)&s->d_pix_prefix, s->pix_prefix_cap * sizeof(uint32_t)));
	}
	// on `stream`: kernels of the previous pass may still be reading the table (a null-stream copy does not order against a
	// non-blocking stream); the source is pageable, so the call returns once it has been staged
	HIP_OK(hipMemcpyAsync(s->d_pix_prefix, pp.data(), pp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
	uint32_t max_paths = kWfMaxPaths;
	if(const char *e = std::getenv("YAFGPU_WF_CHUNK")) max_paths = std::max(256u, (uint32_t)std::strtoul(e, nullptr, 10));     // tests chunk tiny frames
	// recursiveRaytrace: a frame of 5 records per level a camera hit may recurse to
	const ReplayPlan plan = replay_plan(s, rp);
	const int frames = plan.frames;
	const int frame_recs = s->has_glossy ? (s->has_bump ? 13 : 12) : 5;
	if(frames > 7) return fail(-17, "raydepth + additionaldepth > 7 with mirror / transparent / glossy-recursive materials: the device path keeps at most 7 recursion frames per sample");
	const bool path = rp.integrator == YAFGPU_INTEGRATOR_PATH;
	const bool need_rr = plan.need_rr, replay = plan.replay, replay_lights = plan.replay_lights, lc_sharded = plan.lc_sharded;
	const uint32_t n_ps = (uint32_t)std::max(rp.path_samples, 1), n_prob = (uint32_t)std::max(rp.bounces - 1, 1);
	const uint32_t ev_m = replay ? (uint32_t)plan.ev_m : 1u;
	if(replay && ev_m > 1)
	{	// the event tables grow with the calls a sample can make: keep a chunk's tables within 12 GB
		const uint64_t per_slot = (uint64_t)ev_m * n_ps * (4 + 4 * n_prob + 2) + 4;
		max_paths = (uint32_t)std::min<uint64_t>(max_paths, std::max<uint64_t>((12ull << 30) / per_slot, 4096));
	}
	// chunks: runs of pixels whose paths are in flight together.  With the replay a chunk is a run of whole tiles (a tile's
	// stream is walked in one go); without it any run of at most max_paths / spp pixels.
	struct Chunk { uint32_t pixel_begin, n_pixels, tile_begin, tile_end; };
	std::vector<Chunk> chunks;
	if(replay)
	{
		const uint32_t n_t = (uint32_t)s->h_tiles.size();
		for(uint32_t t0 = 0; t0 < n_t;)
		{
			uint32_t t1 = t0 + 1;
			while(t1 < n_t && (uint64_t)(tile_px[t1 + 1] - tile_px[t0]) * spp <= max_paths) ++t1;
			chunks.push_back({tile_px[t0], tile_px[t1] - tile_px[t0], t0, t1});
			t0 = t1;
		}
	}
	else
	{
		const uint32_t chunk_pixels = std::max(1u, std::min(n_pixels_total, std::max(1u, max_paths / spp)));
		for(uint32_t pb = 0; pb < n_pixels_total; pb += chunk_pixels) chunks.push_back({pb, std::min(chunk_pixels, n_pixels_total - pb), 0u, 0u});
	}
	uint32_t cap_pixels = 1u;
	for(const Chunk &ch : chunks) cap_pixels = std::max(cap_pixels, ch.n_pixels);
	if((uint64_t)cap_pixels * spp > (1ull << 27)) return fail(-23, "one tile's samples exceed the 2^27 paths a wavefront chunk can hold: reduce tile_size or the samples per pass");
	const uint32_t cap = cap_pixels * spp;
	if(cap > s->wf_cap || frames * frame_recs > s->wf_frames)
	{
		HIP_OK(hipStreamSynchronize(stream));
		if(s->wf_state) (void)hipFree(s->wf_state);
		if(s->wf_results) (void)hipFree(s->wf_results);
		if(s->wf_queues) (void)hipFree(s->wf_queues);
		if(s->wf_verdict) (void)hipFree(s->wf_verdict);
		if(s->wf_pix_xy) (void)hipFree(s->wf_pix_xy);
		s->wf_state = nullptr; s->wf_results = nullptr; s->wf_queues = nullptr; s->wf_verdict = nullptr; s->wf_pix_xy = nullptr; s->wf_cap = 0;
		HIP_OK(hipMalloc((void **)&s->wf_state, (size_t)(kWfRecs + frame_recs * frames) * cap * sizeof(float4)));
		s->wf_frames = frames * frame_recs;      // frame records allocated per path
		HIP_OK(hipMalloc((void **)&s->wf_results, (size_t)cap * sizeof(float4)));
		// per buffer set: closest (cap), shadow rays (4*cap: up to two MIS pairs per park), resume (cap)
		HIP_OK(hipMalloc((void **)&s->wf_queues, (size_t)12 * cap * sizeof(uint32_t)));
		HIP_OK(hipMalloc((void **)&s->wf_verdict, ((size_t)4 * cap + 31) / 32 * sizeof(uint32_t) + 64));      // one bit per shadow ray
		HIP_OK(hipMalloc((void **)&s->wf_pix_xy, (size_t)cap * sizeof(uint32_t)));     // pixels of a chunk <= paths of a chunk
		s->wf_cap = cap;
	}
	uint32_t hit_k = 0; bool use_hits = false;
	if(replay)
	{	// event tables of the record pass, per path sample; segment tables per tile
		const size_t ents = (size_t)s->wf_cap * ev_m * n_ps;
		if(ents > s->rp_ents || n_prob > s->rp_prob)
		{
			HIP_OK(hipStreamSynchronize(stream));
			for(void *q : {(void *)s->rp_flags, (void *)s->rp_p, (void *)s->rp_kill, (void *)s->rp_calls, (void *)s->rp_base}) if(q) (void)hipFree(q);
			s->rp_flags = nullptr; s->rp_p = nullptr; s->rp_kill = nullptr; s->rp_calls = nullptr; s->rp_base = nullptr; s->rp_ents = 0;
			HIP_OK(hipMalloc((void **)&s->rp_flags, ents * sizeof(uint32_t)));
			HIP_OK(hipMalloc((void **)&s->rp_p, ents * n_prob * sizeof(float)));
			HIP_OK(hipMalloc((void **)&s->rp_kill, ents));
			HIP_OK(hipMalloc((void **)&s->rp_calls, ents));
			HIP_OK(hipMalloc((void **)&s->rp_base, (size_t)s->wf_cap * sizeof(uint32_t)));
			s->rp_ents = ents; s->rp_prob = n_prob;
		}
		// the record pass's closest-hit answers, one per (call, path sample, segment): kept when a camera sample's fit in 1 KB (the
		// final pass then looks its closest hits up instead of tracing them again); never in a stats pass, whose per-ray traversal
		// counts are the point
		hit_k = ev_m * n_ps * ((uint32_t)std::max(rp.bounces, 1) + 1u);
		use_hits = !stats && (size_t)hit_k * sizeof(float4) <= 1024 && (uint64_t)s->wf_cap * hit_k < (1ull << 32);      // (wf_hit_key is a 32-bit index)
		if(const char *e = std::getenv("YAFGPU_HIT_CACHE")) use_hits = use_hits && std::atoi(e) != 0;
		if(use_hits && (size_t)s->wf_cap * hit_k > s->rp_hits_cap)
		{
			HIP_OK(hipStreamSynchronize(stream));
			if(s->rp_hits) (void)hipFree(s->rp_hits);
			s->rp_hits = nullptr; s->rp_hits_cap = 0;
			HIP_OK(hipMalloc((void **)&s->rp_hits, (size_t)s->wf_cap * hit_k * sizeof(float4)));
			s->rp_hits_cap = (size_t)s->wf_cap * hit_k;
		}
		// every chunk's tiles as segments — first pixel of each (chunk-local) and the seed of its Random:
		// rand() + offset * (resx * tile.y + tile.x) + 123 (integrator_tiled.cc:319), offset = pass offset + base sampling
		// offset (:203,263).  One table for the whole pass, uploaded once: chunk k reads its slice [seg_off[k], ...).
		HIP_OK(hipStreamSynchronize(stream));        // the previous pass's upload of the table has been read
		s->h_seg_begin.clear(); s->h_seg_seed.clear();
		{
			const int ntx = (rp.width + rp.tile_size - 1) / rp.tile_size;
			const uint32_t offset = rp.pass_offset + rp.base_sampling_offset;
			for(const Chunk &ch : chunks)
			{
				for(uint32_t t = ch.tile_begin; t <= ch.tile_end; ++t) s->h_seg_begin.push_back(tile_px[t] - ch.pixel_begin);
				for(uint32_t t = ch.tile_begin; t < ch.tile_end; ++t)
				{
					const int4 &r = s->h_tiles[t];
					const int t_global = ((r.y - rp.ystart) / rp.tile_size) * ntx + (r.x - rp.xstart) / rp.tile_size;
					const uint32_t rnd = rp.tile_rand ? (uint32_t)rp.tile_rand[t_global] : 0u;
					s->h_seg_seed.push_back(rnd + offset * ((uint32_t)s->dev.cam.resx * (uint32_t)r.y + (uint32_t)r.x) + 123u);
				}
				s->h_seg_seed.push_back(0u);           // keeps the two tables aligned (n + 1 entries per chunk)
			}
		}
		if(std::getenv("YAFGPU_VERBOSE"))
		{
			std::fprintf(stderr, "[yafgpu] replay pass_offset %u accumulate %d spp %u: seeds", rp.pass_offset, rp.accumulate, spp);
			for(size_t k = 0; k < std::min<size_t>(s->h_seg_seed.size(), 6); ++k) std::fprintf(stderr, " %u", s->h_seg_seed[k]);
			std::fprintf(stderr, " ... entries/tile");
			for(size_t k = 0; k + 1 < std::min<size_t>(s->h_seg_begin.size(), 7); ++k) std::fprintf(stderr, " %u", s->h_seg_begin[k + 1] - s->h_seg_begin[k]);
			std::fprintf(stderr, "\n");
		}
		const size_t segs = s->h_seg_begin.size();
		if(segs > s->rp_segs)
		{
			HIP_OK(hipStreamSynchronize(stream));
			for(void *q : {(void *)s->rp_seg_begin, (void *)s->rp_seg_seed, (void *)s->rp_seg_total, (void *)s->rp_seg_base}) if(q) (void)hipFree(q);
			s->rp_seg_begin = nullptr; s->rp_seg_seed = nullptr; s->rp_seg_total = nullptr; s->rp_seg_base = nullptr; s->rp_segs = 0;
			HIP_OK(hipMalloc((void **)&s->rp_seg_begin, segs * sizeof(uint32_t)));
			HIP_OK(hipMalloc((void **)&s->rp_seg_seed, segs * sizeof(uint32_t)));
			HIP_OK(hipMalloc((void **)&s->rp_seg_total, segs * sizeof(uint32_t)));
			HIP_OK(hipMalloc((void **)&s->rp_seg_base, segs * sizeof(uint32_t)));
			s->rp_segs = segs;
		}
		HIP_OK(hipMemcpyAsync(s->rp_seg_begin, s->h_seg_begin.data(), segs * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		HIP_OK(hipMemcpyAsync(s->rp_seg_seed, s->h_seg_seed.data(), segs * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		if(!s->rp_counter) HIP_OK(hipMalloc((void **)&s->rp_counter, sizeof(uint32_t)));
		// correlative_sample_number_ is zeroed once per render, before its first pass (integrator_tiled.cc:192-194)
		if(!rp.accumulate) HIP_OK(hipMemsetAsync(s->rp_counter, 0, sizeof(uint32_t), stream));
	}
	// transparent shadows only cost anything when a material can be transparent to a shadow ray
	const bool transp = rp.transp_shad != 0 && s->has_transparent;
	if(transp && rp.shadow_depth > kTsMaxDepth) return fail(-18, "shadowDepth > 8 with transparent shadows: the device path remembers at most 9 filtered triangles per shadow ray");
	if(transp && s->wf_cap > s->wf_filt_cap)
	{
		if(s->wf_filt) (void)hipFree(s->wf_filt);
		s->wf_filt = nullptr; s->wf_filt_cap = 0;
		HIP_OK(hipMalloc((void **)&s->wf_filt, (size_t)2 * s->wf_cap * sizeof(float4)));
		s->wf_filt_cap = s->wf_cap;
	}
	if(!s->wf_counts) HIP_OK(hipMalloc((void **)&s->wf_counts, 128 * sizeof(uint32_t)));      // two sets of 2 x 32 (chunk pipelining)
	if(s->n_cus <= 0)
	{
		int dev = 0; hipDeviceProp_t prop;
		HIP_OK(hipGetDevice(&dev));
		HIP_OK(hipGetDeviceProperties(&prop, dev));
		s->n_cus = prop.multiProcessorCount;
	}
	const int cus = s->n_cus;
	const int g_trace_c = stats ? wf_grid((const void *)wf_trace<false, true>, cus) : wf_grid((const void *)wf_trace<false, false>, cus);
	const int g_trace_s = stats ? wf_grid((const void *)wf_trace<true, true>, cus) : wf_grid((const void *)wf_trace<true, false>, cus);
	// two MIS pairs per park (WfArgs::multi): not with transparent shadows (their filter products are kept per pair), not with recursion frames —
	// and only where a light estimate can have a second pair at all: the kernels that carry it are a little slower on the first
	bool want_multi = !(rp.transp_shad != 0 && s->has_transparent) && frames == 0;
	if(const char *e = std::getenv("YAFGPU_MULTI_PAIR")) if(std::atoi(e) == 0) want_multi = false;
	if(want_multi)
	{
		int pairs = 0;
		for(int i = 0; i < s->n_lights; ++i)
		{
			const yafgpu_light &l = s->h_lights[(size_t)i];
			pairs += l.type == YAFGPU_LIGHT_POINT ? 1 : (int)std::ceil((float)l.samples * rp.aa_light_sample_multiplier);
		}
		want_multi = pairs > 1;
	}
	const ShadeVariant *shade_variant = pick_shade_variant(s, frames, false, want_multi);
	if(!shade_variant && want_multi)
	{	// no kernel with the second pair for these materials: the one without it rather than the general kernel
		shade_variant = pick_shade_variant(s, frames, false, false);
		if(shade_variant) want_multi = false;
	}
	if(std::getenv("YAFGPU_VERBOSE")) std::fprintf(stderr, "[yafgpu] shading kernel: %s (materials 0x%x, frames %d), serial replay: %s\n", shade_variant ? shade_variant->name : "general", s->mat_mask, frames,
	                                               replay ? (replay_lights ? (need_rr ? "roulette + light counter" : "light counter") : "roulette") : "off");
	const int g_shade = wf_grid(shade_variant ? shade_variant->kernel() : (const void *)wf_shade, cus);
	// a record pass runs its own, smaller program where one was built for the scene's materials (else the pass's kernel, which branches on WfArgs::replay)
	const ShadeVariant *record_variant = replay ? pick_shade_variant(s, frames, true) : nullptr;
	const int g_shade_rec = record_variant ? wf_grid(record_variant->kernel(), cus) : g_shade;
	// upper bound of kd-tree queries per path = iterations needed (every path advances one query per iteration)
	int r_all = 0, r_one = 0;
	for(int i = 0; i < s->n_lights; ++i)
	{
		const yafgpu_light &l = s->h_lights[(size_t)i];
		const int r = l.type == YAFGPU_LIGHT_POINT ? 1 : (int)std::ceil((float)l.samples * rp.aa_light_sample_multiplier);   // shadow parks (MIS pairs)
		r_all += r; r_one = std::max(r_one, r);
	}
	int iters = 1 + r_all;
	if(rp.integrator == YAFGPU_INTEGRATOR_PATH)
		iters += std::max(1, rp.path_samples) * ((1 + r_one) + std::max(0, rp.bounces - 1) * (1 + r_one));
	// a record pass asks for closest hits only: the camera ray + per path sample one query per segment
	const int iters_record = 1 + (path ? std::max(1, rp.path_samples) * std::max(1, rp.bounces) : 0);
	// opt-in (YAFGPU_OVERLAP=1): +2-4 % on the bench scenes, but per-kernel durations then overlap in a profiler trace, so
	// the default keeps one kernel on the GPU at a time and the roofline numbers comparable with rocprofv3's
	// ... except for small chunks (a tile shard of a multi-GPU render, a small frame): with a few rays per lane a persistent
	// launch is mostly tail, and the two traversal launches of an iteration side by side are worth +5 % at half the metric
	// frame, +8 % at a quarter, +12 % at an eighth (bench.py --emulate-shard).  YAFGPU_OVERLAP=0 / 1 force either.
	// ... and always since a path parks its next segment beside a vertex's last shadow pair (WfArgs::speculate): the middle phases of a
	// pass then have BOTH queues full, and the two launches side by side are worth +2 % on the whole metric frame too (profiles/r03_ab_speculate.txt).
	// The per-kernel durations of the roofline come from the profiled pass, which keeps one kernel on the GPU at a time either way.
	bool overlap = (uint64_t)cap_pixels * spp < (12ull << 20);
	{
		const char *sp = std::getenv("YAFGPU_SPECULATE");
		if(!(sp && std::atoi(sp) == 0)) overlap = true;
	}
	if(const char *e = std::getenv("YAFGPU_OVERLAP")) overlap = std::atoi(e) != 0;
	if(overlap && !s->side_stream)
	{
		HIP_OK(hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking));
		HIP_OK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
		HIP_OK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
	}
	EventPair evp;          // destroyed on every return
	hipEvent_t (&ev)[2] = evp.e;
	if(s->profiling) { HIP_OK(hipEventCreate(&ev[0])); HIP_OK(hipEventCreate(&ev[1])); for(int k = 0; k < 4; ++k) { s->prof_ms[k] = 0; s->prof_launches[k] = 0; } }
	auto timed = [&](int slot, auto &&launch) -> int {
		if(s->profiling) HIP_OK(hipEventRecord(ev[0], stream));
		launch();
		HIP_OK(hipGetLastError());
		if(s->profiling)
		{
			HIP_OK(hipEventRecord(ev[1], stream));
			HIP_OK(hipEventSynchronize(ev[1]));
			float ms = 0.f; HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[1]));
			s->prof_ms[slot] += ms; s->prof_launches[slot] += 1;
		}
		return 0;
	};
	// phase 0: the whole program for a chunk.  A sharded light counter (lc_sharded) splits it: phase 1 = record pass + the tiles'
	// roulette walk and call counts, for every chunk; then the ranks exchange the counts; phase 2 = the rest, from the bases the
	// exchange gave (a pass of one chunk keeps its events from phase 1, one of several records them again).
	std::vector<std::pair<int, uint32_t>> own_calls;
	auto process_chunk = [&](const Chunk &ch, size_t &seg_off, int phase) -> int
	{
		if(s->aborted()) return fail(-30, "aborted");
		WfArgs a{};
		a.ra = ra;
		a.state = s->wf_state; a.cap = s->wf_cap; a.results = s->wf_results; a.frames = frames; a.frame_recs = frame_recs; a.has_glossy = s->has_glossy ? 1 : 0;
		a.pixel_begin = ch.pixel_begin; a.n_pixels = ch.n_pixels; a.n_paths = a.n_pixels * spp;
		a.pix_prefix = s->d_pix_prefix; a.pix_xy = s->wf_pix_xy; a.pix_listed = masked ? 1 : 0;
		if(masked) HIP_OK(hipMemcpyAsync(s->wf_pix_xy, listed.data() + ch.pixel_begin, (size_t)a.n_pixels * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		a.ev_flags = s->rp_flags; a.ev_p = s->rp_p; a.ev_kill = s->rp_kill; a.ev_calls = s->rp_calls; a.lc_base = s->rp_base;
		a.replay_lights = replay_lights ? 1 : 0; a.ev_m = (int)ev_m;
		a.multi = want_multi ? 1 : 0;
		a.speculate = 1;      // the next segment beside a vertex's last shadow pair (WfArgs::speculate); YAFGPU_SPECULATE=0: the sequential phases
		if(const char *e = std::getenv("YAFGPU_SPECULATE")) a.speculate = std::atoi(e) != 0 ? 1 : 0;
		a.hit_cache = use_hits ? s->rp_hits : nullptr; a.hit_k = (int)hit_k;
		const size_t cp = s->wf_cap;
		uint32_t *qset[2][3] = {{s->wf_queues, s->wf_queues + cp, s->wf_queues + 5 * cp},
		                        {s->wf_queues + 6 * cp, s->wf_queues + 7 * cp, s->wf_queues + 11 * cp}};   // closest, shadow rays (4*cap), resume
		uint32_t *cnt[2] = {s->wf_counts, s->wf_counts + 32};
		a.verdict = s->wf_verdict; a.shadow_filt = transp ? s->wf_filt : nullptr;
		const uint32_t g_gen = std::min<uint32_t>((a.n_paths + kBlock - 1) / kBlock, (uint32_t)cus * 8u);
		int rc;
		// one run of the path program over the chunk: generate, then iterations of {closest-hit, any-hit, shade} until
		// every path has ended.  Without recursion the number of queries per path is bounded a priori (n_iters) and the
		// loop never asks the device anything.  With recursiveRaytrace a sample may visit up to 2^raydepth levels, so past
		// that bound the loop runs on while any queue is non-empty (one 20-byte read-back per iteration).
		auto run = [&](int n_iters, bool record) -> int {
			a.cnt_in = cnt[0]; a.cnt_out = cnt[1];
			a.q_closest_in = nullptr; a.q_shadow_in = qset[0][1]; a.q_resume_in = qset[0][2];
			a.q_closest_out = qset[1][0]; a.q_shadow_out = qset[1][1]; a.q_resume_out = qset[1][2];
			if((rc = timed(3, [&] { hipLaunchKernelGGL(wf_generate, dim3(g_gen), dim3(kBlock), 0, stream, a); }))) return rc;
			int cur = 0;
			const int iter_cap = n_iters * (frames > 0 ? (1 << (frames + 1)) : 1) * (s->has_glossy ? (s->has_glossy_two ? 32 : 16) : 1);      // (a safety net: the loop ends when the queues are empty)
			for(int it = 0; it < iter_cap; ++it)
			{
				if(frames > 0 && it >= n_iters)
				{
					uint32_t pending[5];
					HIP_OK(hipMemcpyAsync(pending, a.cnt_in, sizeof pending, hipMemcpyDeviceToHost, stream));
					HIP_OK(hipStreamSynchronize(stream));
					if(pending[0] == 0u && pending[1] == 0u && pending[4] == 0u) break;
				}
				else if(frames == 0 && it >= n_iters) break;
				HIP_OK(hipMemsetAsync(a.cnt_out, 0, 8 * sizeof(uint32_t), stream));
				const bool overlap_now = overlap && it > 0 && !s->profiling && !record;
				if(overlap_now)
				{	// fork point: everything enqueued so far (the previous shade, the counter reset) precedes both launches
					HIP_OK(hipEventRecord(s->ev_fork, stream));
					HIP_OK(hipStreamWaitEvent(s->side_stream, s->ev_fork, 0));
				}
				if(!record && a.replay == 2 && a.hit_cache != nullptr)
				{	// the record pass answered these queries already: wf_shade reads them from its cache where it would read the traversal's answers
				}
				else if((rc = timed(0, [&] {
					if(stats) hipLaunchKernelGGL((wf_trace<false, true>), dim3(g_trace_c), dim3(kBlock), 0, stream, a);
					else hipLaunchKernelGGL((wf_trace<false, false>), dim3(g_trace_c), dim3(kBlock), 0, stream, a); }))) return rc;
				// The two traversal launches of an iteration are independent (each drains its own queue, writes its own
				// answers), and a persistent kernel's tail leaves CUs idle: outside profiling the any-hit launch goes to a
				// side stream so that its waves fill the closest-hit launch's tail (and vice versa).
				const bool fork = overlap_now;
				hipStream_t any_stream = fork ? s->side_stream : stream;
				if(it > 0 && !record)      // (a record pass has no shadow rays)
				{
					HIP_OK(hipMemsetAsync(s->wf_verdict, 0, ((size_t)4 * a.n_paths + 31) / 32 * sizeof(uint32_t), any_stream));     // occluded rays set their bit
					if((rc = timed(1, [&] {
						if(transp) hipLaunchKernelGGL(wf_trace_ts, dim3(cus * 8), dim3(kBlock), 0, any_stream, a);
						else if(stats) hipLaunchKernelGGL((wf_trace<true, true>), dim3(g_trace_s), dim3(kBlock), 0, any_stream, a);
						else hipLaunchKernelGGL((wf_trace<true, false>), dim3(g_trace_s), dim3(kBlock), 0, any_stream, a); }))) return rc;
				}
				if(fork)
				{
					HIP_OK(hipEventRecord(s->ev_join, s->side_stream));
					HIP_OK(hipStreamWaitEvent(stream, s->ev_join, 0));
				}
				int variant_rc = 0;
				if((rc = timed(2, [&] {
					if(record && record_variant) variant_rc = record_variant->launch(&a, sizeof a, g_shade_rec, stream);
					else if(shade_variant) variant_rc = shade_variant->launch(&a, sizeof a, g_shade, stream);
					else hipLaunchKernelGGL(wf_shade, dim3(g_shade), dim3(kBlock), 0, stream, a); }))) return rc;
				if(variant_rc) return fail(-21, "shading kernel variant and main unit disagree on the argument layout");
				// swap queues: what shade produced is the next iteration's input
				cur ^= 1;
				a.cnt_in = cnt[cur]; a.cnt_out = cnt[cur ^ 1];
				a.q_closest_in = qset[cur][0]; a.q_shadow_in = qset[cur][1]; a.q_resume_in = qset[cur][2];
				a.q_closest_out = qset[cur ^ 1][0]; a.q_shadow_out = qset[cur ^ 1][1]; a.q_resume_out = qset[cur ^ 1][2];
			}
			return 0;
		};
		if(replay)
		{
			const bool have_events = phase == 2 && chunks.size() == 1;
			if(!have_events)
			{	// record pass: the paths alone (no light estimates, no roulette kills), rays not counted
				a.replay = 1;
				yafgpu_counters *const keep = a.ra.counters;
				a.ra.counters = nullptr;
				HIP_OK(hipMemsetAsync(s->rp_flags, 0, (size_t)a.n_paths * ev_m * n_ps * sizeof(uint32_t), stream));
				if((rc = run(iters_record, true))) return rc;
				a.ra.counters = keep;
			}
			const uint32_t n_seg = ch.tile_end - ch.tile_begin;
			ReplayArgs r{};
			r.seg_begin = s->rp_seg_begin + seg_off; r.seg_seed = s->rp_seg_seed + seg_off; r.n_seg = n_seg; r.spp = spp; r.n_paths = ev_m * n_ps; r.n_prob = n_prob;
			r.bounces = (uint32_t)std::max(rp.bounces, 1);
			r.ev_flags = s->rp_flags; r.ev_p = s->rp_p; r.ev_kill = s->rp_kill; r.ev_calls = s->rp_calls; r.lc_base = s->rp_base;
			r.seg_total = s->rp_seg_total + seg_off; r.lc_counter = s->rp_counter;
			r.seg_base_in = phase == 2 ? s->rp_seg_base + seg_off : nullptr;
			if((rc = timed(3, [&] {
				if(!have_events) hipLaunchKernelGGL(wf_replay_tiles, dim3(n_seg), dim3(kWave), 0, stream, r);
				if(phase == 0) hipLaunchKernelGGL(wf_replay_bases, dim3(1), dim3(1), 0, stream, r);
				if(phase != 1) hipLaunchKernelGGL(wf_replay_samples, dim3(n_seg), dim3(kWave), 0, stream, r); }))) return rc;
			if(phase == 1)
			{	// this chunk's calls per tile, by global tile index
				std::vector<uint32_t> totals(n_seg);
				HIP_OK(hipMemcpyAsync(totals.data(), s->rp_seg_total + seg_off, n_seg * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
				HIP_OK(hipStreamSynchronize(stream));
				const int ntx = (rp.width + rp.tile_size - 1) / rp.tile_size;
				for(uint32_t k = 0; k < n_seg; ++k)
				{
					const int4 &t = s->h_tiles[ch.tile_begin + k];
					own_calls.emplace_back(((t.y - rp.ystart) / rp.tile_size) * ntx + (t.x - rp.xstart) / rp.tile_size, totals[k]);
				}
				seg_off += n_seg + 1;
				return 0;
			}
			if(std::getenv("YAFGPU_VERBOSE"))
			{
				uint32_t cnt_now = 0;
				HIP_OK(hipMemcpyAsync(&cnt_now, s->rp_counter, sizeof cnt_now, hipMemcpyDeviceToHost, stream));
				HIP_OK(hipStreamSynchronize(stream));
				std::fprintf(stderr, "[yafgpu] replay chunk of %u tiles: light counter now %u\n", n_seg, cnt_now);
			}
			a.replay = 2;
			seg_off += n_seg + 1;
		}
		if((rc = run(iters, false))) return rc;
		const uint32_t g_acc = std::min<uint32_t>((a.n_pixels + kBlock - 1) / kBlock, (uint32_t)cus * 8u);
		if(pass_piped)
		{	// the film on the caller's stream, after this pass's path work and (by that stream's order) after the previous pass's film
			HIP_OK(hipEventRecord(s->pipe_done[pipe_k], stream));
			HIP_OK(hipStreamWaitEvent(caller, s->pipe_done[pipe_k], 0));
			WfArgs acc = a;
			acc.ra.counters = caller_counters;
			hipLaunchKernelGGL(wf_accumulate, dim3(g_acc), dim3(kBlock), 0, caller, acc);
			if(caller_counters) hipLaunchKernelGGL(add_counters, dim3(1), dim3(64), 0, caller, caller_counters, (const yafgpu_counters *)s->pipe_counters[pipe_k]);
			HIP_OK(hipGetLastError());
			HIP_OK(hipEventRecord(s->pipe_acc[pipe_k], caller));
			s->pipe_acc_set[pipe_k] = true;
			return 0;
		}
		if((rc = timed(3, [&] { hipLaunchKernelGGL(wf_accumulate, dim3(g_acc), dim3(kBlock), 0, stream, a); }))) return rc;
		return 0;
	};
	int rc = 0;
	// Chunk pipelining.  The launches of a pass form a chain (trace -> shade -> trace ...), and a persistent traversal launch ends in a
	// tail as long as one ray's walk (~0.3 ms at 1 M triangles) during which the GPU drains: four exposed tails per pass, 5 % of the
	// metric pass and a third of an eighth-of-the-frame shard's.  Two halves of the pass's pixels, each with its own state, queues
	// and stream, run the same chain side by side: one's tail is filled by the other's launches.  The film planes are added to by
	// one accumulate launch at a time, after both.  (Not with the serial-state replay, recursion polling, a resample mask or per-kernel
	// profiling, which keep the sequential path below.)
	// MEASURED, OFF (opt in with YAFGPU_CHUNK_PIPELINE=1): the two streams' persistent launches mostly take turns instead of filling
	// each other's tails, and every half pays the full tail — 25.5 / 13.8 / 8.0 / 5.5 ms per pass at 1, 1/2, 1/4, 1/8 of the metric frame
	// against 25.5 / 13.9 / 7.9 / 4.75 ms with the closest-hit / any-hit overlap alone.
	bool pipelined = false;
	if(const char *e = std::getenv("YAFGPU_CHUNK_PIPELINE"))
		pipelined = std::atoi(e) != 0 && !replay && frames == 0 && !masked && !s->profiling && !stats && (uint64_t)n_pixels_total * spp >= (1u << 16);
	if(pipelined)
	{
		if(!s->side_stream)
		{
			HIP_OK(hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking));
			HIP_OK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
			HIP_OK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
		}
		struct Ctx { WfArgs a; uint32_t *qset[2][3]; uint32_t *cnt[2]; hipStream_t st; uint32_t g_gen; int cur; };
		// halves of every chunk, two at a time (a chunk already fits the allocated capacity: its halves fit side by side)
		for(const Chunk &ch : chunks)
		{
			if(s->aborted()) return fail(-30, "aborted");
			const uint32_t px_a = (ch.n_pixels + 1u) / 2u, px_b = ch.n_pixels - px_a;
			Ctx cx[2];
			const int n_ctx = px_b > 0 ? 2 : 1;
			size_t slot_off = 0;
			for(int k = 0; k < n_ctx; ++k)
			{
				Ctx &c = cx[k];
				const uint32_t px = k == 0 ? px_a : px_b, px_begin = ch.pixel_begin + (k == 0 ? 0u : px_a);
				const size_t np = (size_t)px * spp;
				c.a = WfArgs{};
				c.a.ra = ra;
				c.a.state = s->wf_state + slot_off; c.a.cap = s->wf_cap; c.a.results = s->wf_results + slot_off;
				c.a.frames = 0; c.a.frame_recs = frame_recs; c.a.has_glossy = s->has_glossy ? 1 : 0;
				c.a.pixel_begin = px_begin; c.a.n_pixels = px; c.a.n_paths = (uint32_t)np;
				c.a.pix_prefix = s->d_pix_prefix; c.a.pix_xy = s->wf_pix_xy + (k == 0 ? 0u : px_a); c.a.pix_listed = 0;
				c.a.ev_m = 1;
				uint32_t *qb = s->wf_queues + 12 * slot_off;
				c.qset[0][0] = qb; c.qset[0][1] = qb + np; c.qset[0][2] = qb + 5 * np;
				c.qset[1][0] = qb + 6 * np; c.qset[1][1] = qb + 7 * np; c.qset[1][2] = qb + 11 * np;
				c.cnt[0] = s->wf_counts + 64 * k; c.cnt[1] = s->wf_counts + 64 * k + 32;
				c.a.verdict = s->wf_verdict + (4 * slot_off + 31) / 32 + (k ? 1 : 0);
				c.a.shadow_filt = transp ? s->wf_filt + 2 * slot_off : nullptr;
				c.st = k == 0 ? stream : s->side_stream;
				c.g_gen = std::min<uint32_t>((uint32_t)((np + kBlock - 1) / kBlock), (uint32_t)cus * 8u);
				c.cur = 0;
				slot_off += np;
			}
			// fork: what the caller's stream holds so far (the previous pass, the prefix upload) precedes both halves
			HIP_OK(hipEventRecord(s->ev_fork, stream));
			HIP_OK(hipStreamWaitEvent(s->side_stream, s->ev_fork, 0));
			for(int k = 0; k < n_ctx; ++k)
			{
				Ctx &c = cx[k];
				c.a.cnt_in = c.cnt[0]; c.a.cnt_out = c.cnt[1];
				c.a.q_closest_in = nullptr; c.a.q_shadow_in = c.qset[0][1]; c.a.q_resume_in = c.qset[0][2];
				c.a.q_closest_out = c.qset[1][0]; c.a.q_shadow_out = c.qset[1][1]; c.a.q_resume_out = c.qset[1][2];
				hipLaunchKernelGGL(wf_generate, dim3(c.g_gen), dim3(kBlock), 0, c.st, c.a);
			}
			HIP_OK(hipGetLastError());
			for(int it = 0; it < iters; ++it)
				for(int k = 0; k < n_ctx; ++k)
				{
					Ctx &c = cx[k];
					HIP_OK(hipMemsetAsync(c.a.cnt_out, 0, 8 * sizeof(uint32_t), c.st));
					hipLaunchKernelGGL((wf_trace<false, false>), dim3(g_trace_c), dim3(kBlock), 0, c.st, c.a);
					if(it > 0)
					{
						HIP_OK(hipMemsetAsync(c.a.verdict, 0, ((size_t)4 * c.a.n_paths + 31) / 32 * sizeof(uint32_t), c.st));
						if(transp) hipLaunchKernelGGL(wf_trace_ts, dim3(cus * 8), dim3(kBlock), 0, c.st, c.a);
						else hipLaunchKernelGGL((wf_trace<true, false>), dim3(g_trace_s), dim3(kBlock), 0, c.st, c.a);
					}
					if(shade_variant) { if(shade_variant->launch(&c.a, sizeof c.a, g_shade, c.st)) return fail(-21, "shading kernel variant and main unit disagree on the argument layout"); }
					else hipLaunchKernelGGL(wf_shade, dim3(g_shade), dim3(kBlock), 0, c.st, c.a);
					HIP_OK(hipGetLastError());
					c.cur ^= 1;
					c.a.cnt_in = c.cnt[c.cur]; c.a.cnt_out = c.cnt[c.cur ^ 1];
					c.a.q_closest_in = c.qset[c.cur][0]; c.a.q_shadow_in = c.qset[c.cur][1]; c.a.q_resume_in = c.qset[c.cur][2];
					c.a.q_closest_out = c.qset[c.cur ^ 1][0]; c.a.q_shadow_out = c.qset[c.cur ^ 1][1]; c.a.q_resume_out = c.qset[c.cur ^ 1][2];
				}
			// join, then the film: one accumulate launch at a time
			HIP_OK(hipEventRecord(s->ev_join, s->side_stream));
			HIP_OK(hipStreamWaitEvent(stream, s->ev_join, 0));
			for(int k = 0; k < n_ctx; ++k)
			{
				const uint32_t g_acc = std::min<uint32_t>((cx[k].a.n_pixels + kBlock - 1) / kBlock, (uint32_t)cus * 8u);
				hipLaunchKernelGGL(wf_accumulate, dim3(g_acc), dim3(kBlock), 0, stream, cx[k].a);
			}
			HIP_OK(hipGetLastError());
		}
		return 0;
	}
	if(lc_sharded)
	{
		size_t seg_off = 0;
		for(const Chunk &ch : chunks) if((rc = process_chunk(ch, seg_off, 1))) return rc;
		std::vector<uint32_t> base_of_tile;
		if((rc = lc_exchange_counts(s, rp, own_calls, &base_of_tile))) return rc;
		// the bases in the layout of the segment tables (n + 1 entries per chunk)
		const int ntx = (rp.width + rp.tile_size - 1) / rp.tile_size;
		s->h_seg_base.clear();
		for(const Chunk &ch : chunks)
		{
			for(uint32_t t = ch.tile_begin; t < ch.tile_end; ++t)
			{
				const int4 &r = s->h_tiles[t];
				s->h_seg_base.push_back(base_of_tile[(size_t)(((r.y - rp.ystart) / rp.tile_size) * ntx + (r.x - rp.xstart) / rp.tile_size)]);
			}
			s->h_seg_base.push_back(0u);
		}
		HIP_OK(hipMemcpyAsync(s->rp_seg_base, s->h_seg_base.data(), s->h_seg_base.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
	}
	size_t seg_off = 0;
	for(const Chunk &ch : chunks) if((rc = process_chunk(ch, seg_off, lc_sharded ? 2 : 0))) return rc;
	return 0;
}

int yafgpu_render_tiles(yafgpu_scene_t *s, const yafgpu_render_params *rp, float *d_planes, yafgpu_counters *d_counters, void *stream_)
{
	if(!s || !rp || !d_planes) return fail(-1, "null argument");
	int rc = validate(s, rp);
	if(rc) return rc;
	hipStream_t stream = (hipStream_t)stream_;
	RenderArgs ra{};
	ra.sc = s->dev;
	ra.rp = *rp;
	ra.shadow_bias = rp->shadow_bias_auto ? kShadowBias : rp->shadow_bias;     // scene.cc:825
	ra.ray_min_dist = rp->min_raydist_auto ? kMinRayDist : rp->min_raydist;    // scene.cc:826
	{	// ImageFilm ctor, imagefilm.cc:127,152-176: half-width, per-type widening, clamp to [0.501, 4], 16x16 table
		float fw = (float)((double)rp->aa_pixelwidth * 0.5);
		if(rp->filter_type == YAFGPU_FILTER_MITCHELL) fw *= 2.6f;
		else if(rp->filter_type == YAFGPU_FILTER_GAUSS) fw *= 2.f;
		ra.filterw = std::min(std::max(0.501f, fw), 0.5f * 8.f);
		ra.table_scale = (float)(0.9999 * 16 / (double)ra.filterw);
		ra.wide_filter = (rp->filter_type != YAFGPU_FILTER_BOX || ra.filterw > 0.501f) ? 1 : 0;
		ra.filter_table = nullptr;
		if(ra.wide_filter)
		{
			float table[256];
			host_filter_table(rp->filter_type, table);
			if(!s->d_filter_table) HIP_OK(hipMalloc((void **)&s->d_filter_table, sizeof table));
			HIP_OK(hipMemcpyAsync(s->d_filter_table, table, sizeof table, hipMemcpyHostToDevice, stream));
			ra.filter_table = s->d_filter_table;
		}
	}
	const int spp = rp->aa_minsamples;
	ra.lanes_per_pixel = std::min(spp, kWave);
	ra.pixels_per_wave = kWave / ra.lanes_per_pixel;
	ra.iters = (spp + ra.lanes_per_pixel - 1) / ra.lanes_per_pixel;
	// tiles of this shard, row-major (ImageSplitter linear order, imagesplitter.cc:30-60)
	const int key[8] = {rp->width, rp->height, rp->xstart, rp->ystart, rp->tile_size, rp->shard_index, rp->shard_count, ra.pixels_per_wave};
	const bool same = std::memcmp(key, s->tile_key, sizeof key) == 0;
	if(!same)
	{
		const int ntx = (rp->width + rp->tile_size - 1) / rp->tile_size, nty = (rp->height + rp->tile_size - 1) / rp->tile_size;
		std::vector<int4> &tiles = s->h_tiles; std::vector<uint32_t> &prefix = s->h_prefix;
		tiles.clear(); prefix.clear();
		prefix.push_back(0u);
		for(int t = 0; t < ntx * nty; ++t)
		{
			if(t % rp->shard_count != rp->shard_index) continue;
			const int tx = t % ntx, ty = t / ntx;
			int4 r;
			r.x = rp->xstart + tx * rp->tile_size; r.y = rp->ystart + ty * rp->tile_size;
			r.z = std::min(rp->tile_size, rp->xstart + rp->width - r.x); r.w = std::min(rp->tile_size, rp->ystart + rp->height - r.y);
			tiles.push_back(r);
			const uint32_t units = (uint32_t)((r.z * r.w + ra.pixels_per_wave - 1) / ra.pixels_per_wave);
			prefix.push_back(prefix.back() + units);
		}
	}
	ra.n_tiles = (int)s->h_tiles.size();
	ra.n_units = s->h_prefix.back();
	if(!rp->accumulate) HIP_OK(hipMemsetAsync(d_planes, 0, yafgpu_planes_bytes(rp->width, rp->height), stream));
	if(ra.n_tiles == 0)
	{	// a rank without tiles still owes the others its (empty) share of the light-counter exchange
		const char *pl = std::getenv("YAFGPU_PIPELINE");
		const bool mega = pl && std::strcmp(pl, "megakernel") == 0;
		return (!mega && replay_plan(s, *rp).lc_sharded) ? lc_exchange_counts(s, *rp, {}, nullptr) : 0;
	}
	if(!s->d_queue) HIP_OK(hipMalloc((void **)&s->d_queue, kQueues * 32 * sizeof(uint32_t)));
	if(!same)
	{
		if(s->h_tiles.size() > s->tiles_cap)
		{
			HIP_OK(hipStreamSynchronize(stream));      // the previous pass may still read the old arrays
			if(s->d_tiles) (void)hipFree(s->d_tiles);
			if(s->d_prefix) (void)hipFree(s->d_prefix);
			s->tiles_cap = s->h_tiles.size();
			HIP_OK(hipMalloc((void **)&s->d_tiles, s->tiles_cap * sizeof(int4)));
			HIP_OK(hipMalloc((void **)&s->d_prefix, (s->tiles_cap + 1) * sizeof(uint32_t)));
		}
		HIP_OK(hipMemcpyAsync(s->d_tiles, s->h_tiles.data(), s->h_tiles.size() * sizeof(int4), hipMemcpyHostToDevice, stream));
		HIP_OK(hipMemcpyAsync(s->d_prefix, s->h_prefix.data(), s->h_prefix.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		std::memcpy(s->tile_key, key, sizeof key);
		s->pipe_prev = false;      // (pipelined passes: the new tile arrays precede the internal streams' next launches)
	}
	HIP_OK(hipMemsetAsync(s->d_queue, 0, kQueues * 32 * sizeof(uint32_t), stream));
	ra.tile_rect = s->d_tiles; ra.unit_prefix = s->d_prefix; ra.queue_next = s->d_queue;
	for(int q = 0; q <= kQueues; ++q) ra.queue_begin[q] = (uint32_t)(((uint64_t)ra.n_units * (uint64_t)q) / kQueues);
	ra.planes = d_planes;
	ra.counters = d_counters;
	const bool stats = d_counters != nullptr && std::getenv("YAFGPU_STATS") != nullptr;
	{
		const char *pl = std::getenv("YAFGPU_PIPELINE");
		const bool mega = pl && std::strcmp(pl, "megakernel") == 0;
		if(!mega) return render_wavefront(s, ra, stream, stats);
		if(ra.wide_filter) return fail(-15, "the one-kernel pipeline implements the box filter of width <= 1.002 only; use the wavefront pipeline");
		if(s->dev.cam.aperture != 0.f) return fail(-15, "the one-kernel pipeline has the pinhole camera only; use the wavefront pipeline");
		// transpShad changes which hits occlude even without a transparent material (intersectTs skips hits before tmin_)
		if(rp->transp_shad) return fail(-15, "the one-kernel pipeline has no transparent shadows (transpShad); use the wavefront pipeline");
		if((s->has_specular || s->has_glossy) && rp->raydepth + s->max_add_depth > 0) return fail(-15, "the one-kernel pipeline has no recursiveRaytrace; use the wavefront pipeline for mirror / transparent / glossy-recursive materials");
		if(s->has_textures) return fail(-15, "the one-kernel pipeline has no shader nodes / textures; use the wavefront pipeline");
		if(rp->trace_caustics && (s->has_specular || s->has_glossy)) return fail(-15, "the one-kernel pipeline has no path caustics (caustic_type path with specular / glossy lobes); use the wavefront pipeline");
		if(rp->serial_replay && rp->integrator == YAFGPU_INTEGRATOR_PATH && (rp->bounces - 1 > rp->rr_min_bounces || s->n_lights > 1))
			return fail(-15, "the one-kernel pipeline cannot replay the reference's serial state (Russian roulette stream, light counter); use the wavefront pipeline or switch the replay off");
		if(rp->multi_pass || rp->accumulate || rp->resample_mask || rp->aa_clamp_samples != 0.f || rp->pass_offset != 0u)
			return fail(-15, "the one-kernel pipeline renders single-pass films only; use the wavefront pipeline");
	}
	int dev = 0; hipDeviceProp_t prop;
	HIP_OK(hipGetDevice(&dev));
	HIP_OK(hipGetDeviceProperties(&prop, dev));
	int blocks_per_cu = 0;
	HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, render_kernel<false>, kBlock, 0));
	blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
	const uint32_t want = (ra.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
	const uint32_t grid = std::max(1u, std::min(want, (uint32_t)(prop.multiProcessorCount * blocks_per_cu)));
	if(stats) hipLaunchKernelGGL(render_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, ra);
	else hipLaunchKernelGGL(render_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, ra);
	HIP_OK(hipGetLastError());
	return 0;
}

int yafgpu_film_combine(const float *d_planes, float *d_film, int32_t width, int32_t height, void *stream_)
{
	if(!d_planes || !d_film || width <= 0 || height <= 0) return fail(-1, "bad argument");
	const size_t n = (size_t)width * (size_t)height;
	const uint32_t grid = (uint32_t)std::min<size_t>((n + kBlock - 1) / kBlock, 2048);
	hipLaunchKernelGGL(combine_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream_, d_planes, d_film, width, height);
	HIP_OK(hipGetLastError());
	return 0;
}

// planes, combined film and counters of a w x h frame, owned by the scene and reused while the size stays
static int render_targets(yafgpu_scene *s, int w, int h, float **planes, float **film, yafgpu_counters **cnt)
{
	const size_t planes_n = yafgpu_planes_bytes(w, h) / sizeof(float), film_n = (size_t)w * (size_t)h * YAFGPU_FILM_CHANNELS;
	if(planes_n > s->rt_planes_n || film_n > s->rt_film_n)
	{
		HIP_OK(hipDeviceSynchronize());
		if(s->rt_planes) (void)hipFree(s->rt_planes);
		if(s->rt_film) (void)hipFree(s->rt_film);
		s->rt_planes = nullptr; s->rt_film = nullptr; s->rt_planes_n = 0; s->rt_film_n = 0;
		HIP_OK(hipMalloc((void **)&s->rt_planes, planes_n * sizeof(float)));
		HIP_OK(hipMalloc((void **)&s->rt_film, film_n * sizeof(float)));
		s->rt_planes_n = planes_n; s->rt_film_n = film_n;
	}
	if(!s->rt_cnt) HIP_OK(hipMalloc((void **)&s->rt_cnt, sizeof(yafgpu_counters)));
	*planes = s->rt_planes; *film = s->rt_film; *cnt = s->rt_cnt;
	return 0;
}

int yafgpu_render_to_host(yafgpu_scene_t *s, const yafgpu_render_params *rp, float *h_film, yafgpu_counters *h_counters)
{
	if(!s || !rp || !h_film) return fail(-1, "null argument");
	if(rp->width <= 0 || rp->height <= 0) return fail(-10, "empty image");
	float *d_planes = nullptr, *d_film = nullptr; yafgpu_counters *d_cnt = nullptr;
	const size_t film_bytes = (size_t)rp->width * (size_t)rp->height * YAFGPU_FILM_CHANNELS * sizeof(float);
	int rc = render_targets(s, rp->width, rp->height, &d_planes, &d_film, &d_cnt);
	if(rc) return rc;
	HIP_OK(hipMemset(d_cnt, 0, sizeof(yafgpu_counters)));
	rc = yafgpu_render_tiles(s, rp, d_planes, d_cnt, nullptr);
	if(!rc) rc = yafgpu_film_combine(d_planes, d_film, rp->width, rp->height, nullptr);
	if(!rc)
	{
		hipError_t e = hipDeviceSynchronize();
		if(e != hipSuccess) rc = fail(-100, std::string("render: ") + hipGetErrorString(e));
	}
	if(!rc)
	{
		if(hipMemcpy(h_film, d_film, film_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(-100, "film download failed");
		if(h_counters && hipMemcpy(h_counters, d_cnt, sizeof(yafgpu_counters), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(-100, "counter download failed");
	}
	return rc;
}

// ---- multi-pass anti-aliasing: TiledIntegrator::render (integrator_tiled.cc:116-258) -----------------------------
// The noise detection between passes (ImageFilm::nextPass, imagefilm.cc:270-480) is an image-space pass over the
// film so far: aa_detect_kernel below; next_pass_mask is the same on the host (YAFGPU_AA_DETECT=host), kept as its checker.
namespace {
__host__ __device__ float dark_threshold_curve(float b)   // ImageFilm::darkThresholdCurveInterpolate, imagefilm.cc:1312-1328
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
struct Px { float c[4]; };
__host__ __device__ Px px_normalized(const float *p)   // Pixel::normalized, util_image_buffers.h:39-43; Rgba / float, color.h:310-314
{
	Px o;
	if(p[4] != 0.f) { const float f = (float)(1.0 / (double)p[4]); for(int k = 0; k < 4; ++k) o.c[k] = p[k] * f; }
	else for(int k = 0; k < 4; ++k) o.c[k] = 0.f;
	return o;
}
__host__ __device__ float px_difference(const Px &a, const Px &b, bool use_rgb)   // Rgba::colorDifference, color.h:447-464
{
	const float bri_a = 0.2126f * a.c[0] + 0.7152f * a.c[1] + 0.0722f * a.c[2];
	const float bri_b = 0.2126f * b.c[0] + 0.7152f * b.c[1] + 0.0722f * b.c[2];
	float d = fabsf(bri_b - bri_a);
	if(use_rgb) for(int k = 0; k < 4; ++k) { const float dk = fabsf(b.c[k] - a.c[k]); if(d < dk) d = dk; }
	return d;
}
// which pixels get more samples; returns their number
int next_pass_mask(const float *film, int w, int h, const yafgpu_aa_schedule &aa, float aa_thesh, std::vector<uint8_t> &flags)
{
	flags.assign((size_t)w * (size_t)h, 0);
	if(!(aa_thesh > 0.f)) { std::fill(flags.begin(), flags.end(), (uint8_t)1); return w * h; }   // :319,460; doMoreSamples :919
	const int half = aa.variance_edge_size / 2;
	float scaled = aa_thesh;
	auto P = [&](int x, int y) { return film + 5 * ((size_t)y * (size_t)w + (size_t)x); };
	auto set = [&](int x, int y) { flags[(size_t)y * (size_t)w + (size_t)x] = 1; };
	const bool rgb = aa.detect_color_noise != 0;
	for(int y = 0; y < h - 1; ++y)
		for(int x = 0; x < w - 1; ++x)
		{
			if(P(x, y)[4] <= 0.f) set(x, y);
			const Px c = px_normalized(P(x, y));
			const float bri = 0.2126f * std::fabs(c.c[0]) + 0.7152f * std::fabs(c.c[1]) + 0.0722f * std::fabs(c.c[2]);
			if(aa.dark_detection_type == 1 && aa.dark_threshold_factor > 0.f) scaled = aa_thesh * ((1.f - aa.dark_threshold_factor) + (bri * aa.dark_threshold_factor));
			else if(aa.dark_detection_type == 2) scaled = dark_threshold_curve(bri);
			if(px_difference(c, px_normalized(P(x + 1, y)), rgb) >= scaled) { set(x, y); set(x + 1, y); }
			if(px_difference(c, px_normalized(P(x, y + 1)), rgb) >= scaled) { set(x, y); set(x, y + 1); }
			if(px_difference(c, px_normalized(P(x + 1, y + 1)), rgb) >= scaled) { set(x, y); set(x + 1, y + 1); }
			if(x > 0 && px_difference(c, px_normalized(P(x - 1, y + 1)), rgb) >= scaled) { set(x, y); set(x - 1, y + 1); }
			if(aa.variance_pixels > 0)
			{
				int vx = 0, vy = 0;
				for(int xd = -half; xd < half - 1; ++xd)
				{
					int xi = x + xd; if(xi < 0) xi = 0; else if(xi >= w - 1) xi = w - 2;
					if(px_difference(px_normalized(P(xi, y)), px_normalized(P(xi + 1, y)), rgb) >= scaled) ++vx;
				}
				for(int yd = -half; yd < half - 1; ++yd)
				{
					int yi = y + yd; if(yi < 0) yi = 0; else if(yi >= h - 1) yi = h - 2;
					if(px_difference(px_normalized(P(x, yi)), px_normalized(P(x, yi + 1)), rgb) >= scaled) ++vy;
				}
				if(vx + vy >= aa.variance_pixels)
					for(int xd = -half; xd < half; ++xd)
						for(int yd = -half; yd < half; ++yd)
						{
							int xi = x + xd; if(xi < 0) xi = 0; else if(xi >= w) xi = w - 1;
							int yi = y + yd; if(yi < 0) yi = 0; else if(yi >= h) yi = h - 1;
							set(xi, yi);
						}
			}
		}
	int n = 0;
	for(uint8_t f : flags) n += f;
	return n;
}
// The same on the device, one thread per pixel of the loop above (every store is the same 1: the order does not matter), on the
// combined film that is on the device anyway — the host version costs 10 ms per pass at 1024 x 1024 plus the film's way down.
__global__ __launch_bounds__(kBlock) void aa_detect_kernel(const float *film, int w, int h, yafgpu_aa_schedule aa, float aa_thesh, uint8_t *flags)
{
	const int n = (w - 1) * (h - 1);
	const int half = aa.variance_edge_size / 2;
	const bool rgb = aa.detect_color_noise != 0;
	for(int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x))
	{
		const int x = i % (w - 1), y = i / (w - 1);
		auto P = [&](int px, int py) { return film + 5 * ((size_t)py * (size_t)w + (size_t)px); };
		auto set = [&](int px, int py) { flags[(size_t)py * (size_t)w + (size_t)px] = 1; };
		float scaled = aa_thesh;
		if(P(x, y)[4] <= 0.f) set(x, y);
		const Px c = px_normalized(P(x, y));
		const float bri = 0.2126f * fabsf(c.c[0]) + 0.7152f * fabsf(c.c[1]) + 0.0722f * fabsf(c.c[2]);
		if(aa.dark_detection_type == 1 && aa.dark_threshold_factor > 0.f) scaled = aa_thesh * ((1.f - aa.dark_threshold_factor) + (bri * aa.dark_threshold_factor));
		else if(aa.dark_detection_type == 2) scaled = dark_threshold_curve(bri);
		if(px_difference(c, px_normalized(P(x + 1, y)), rgb) >= scaled) { set(x, y); set(x + 1, y); }
		if(px_difference(c, px_normalized(P(x, y + 1)), rgb) >= scaled) { set(x, y); set(x, y + 1); }
		if(px_difference(c, px_normalized(P(x + 1, y + 1)), rgb) >= scaled) { set(x, y); set(x + 1, y + 1); }
		if(x > 0 && px_difference(c, px_normalized(P(x - 1, y + 1)), rgb) >= scaled) { set(x, y); set(x - 1, y + 1); }
		if(aa.variance_pixels > 0)
		{
			int vx = 0, vy = 0;
			for(int xd = -half; xd < half - 1; ++xd)
			{
				int xi = x + xd; if(xi < 0) xi = 0; else if(xi >= w - 1) xi = w - 2;
				if(px_difference(px_normalized(P(xi, y)), px_normalized(P(xi + 1, y)), rgb) >= scaled) ++vx;
			}
			for(int yd = -half; yd < half - 1; ++yd)
			{
				int yi = y + yd; if(yi < 0) yi = 0; else if(yi >= h - 1) yi = h - 2;
				if(px_difference(px_normalized(P(x, yi)), px_normalized(P(x, yi + 1)), rgb) >= scaled) ++vy;
			}
			if(vx + vy >= aa.variance_pixels)
				for(int xd = -half; xd < half; ++xd)
					for(int yd = -half; yd < half; ++yd)
					{
						int xi = x + xd; if(xi < 0) xi = 0; else if(xi >= w) xi = w - 1;
						int yi = y + yd; if(yi < 0) yi = 0; else if(yi >= h) yi = h - 1;
						set(xi, yi);
					}
		}
	}
}
} // namespace

int yafgpu_render_passes_to_host(yafgpu_scene_t *s, const yafgpu_render_params *rp_in, const yafgpu_aa_schedule *aa_in,
                                 float *h_film, yafgpu_counters *h_counters, int32_t *resampled_out)
{
	if(!s || !rp_in || !h_film) return fail(-1, "null argument");
	yafgpu_aa_schedule aa{};
	if(aa_in) aa = *aa_in;
	if(aa.passes < 1) aa.passes = 1;
	const bool exchange = aa.passes > 1 && rp_in->shard_count > 1;
	if(exchange && !s->exchange) return fail(-16, "multi-pass anti-aliasing on a sharded frame needs an exchange function (yafgpu_scene_set_exchange): the noise detection between passes reads every pixel");
	yafgpu_render_params rp = *rp_in;
	const int w = rp.width, h = rp.height;
	if(w <= 0 || h <= 0) return fail(-10, "empty image");
	float *d_planes = nullptr, *d_film = nullptr; yafgpu_counters *d_cnt = nullptr;
	const size_t film_bytes = (size_t)w * (size_t)h * YAFGPU_FILM_CHANNELS * sizeof(float);
	{ const int rc_t = render_targets(s, w, h, &d_planes, &d_film, &d_cnt); if(rc_t) return rc_t; }
	HIP_OK(hipMemset(d_cnt, 0, sizeof(yafgpu_counters)));
	auto film_now = [&](bool download = true) -> int {
		int rc = yafgpu_film_combine(d_planes, d_film, w, h, nullptr);
		if(!rc && download && hipMemcpy(h_film, d_film, film_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(-100, "film download failed");
		return rc;
	};
	// the whole frame's film for the detection step of a sharded render: all ranks' planes summed (exactly: one writer per
	// element), then combined in the single-GPU order
	DevMem<float> d_all;
	const size_t plane_floats = yafgpu_planes_bytes(w, h) / sizeof(float);
	auto film_of_all_ranks = [&](bool download = true) -> int {
		if(!d_all.p && d_all.alloc(plane_floats) != hipSuccess) return fail(-3, "out of device memory (plane exchange)");
		if(hipMemcpy(d_all, d_planes, plane_floats * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return fail(-100, "plane copy failed");
		if(hipDeviceSynchronize() != hipSuccess) return fail(-100, "render failed before the plane exchange");
		if(s->exchange(s->exchange_user, d_all, (uint64_t)plane_floats)) return fail(-31, "the plane exchange function reported a failure");
		int rc = yafgpu_film_combine(d_all, d_film, w, h, nullptr);
		if(!rc && download && hipMemcpy(h_film, d_film, film_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(-100, "film download failed");
		return rc;
	};
	// the detection step on the film that d_film holds: flags down (1 byte per pixel), counted here; < 0: error
	const bool detect_on_host = [] { const char *e = std::getenv("YAFGPU_AA_DETECT"); return e && std::strcmp(e, "host") == 0; }();
	auto detect = [&](float aa_thesh, std::vector<uint8_t> &flags) -> int {
		const size_t n_px = (size_t)w * (size_t)h;
		flags.assign(n_px, 0);
		if(!(aa_thesh > 0.f)) { std::fill(flags.begin(), flags.end(), (uint8_t)1); return w * h; }   // imagefilm.cc:319,460; doMoreSamples :919
		if(n_px > s->rt_flags_n)
		{
			if(s->rt_flags) (void)hipFree(s->rt_flags);
			s->rt_flags = nullptr; s->rt_flags_n = 0;
			if(hipMalloc((void **)&s->rt_flags, n_px) != hipSuccess) return fail(-3, "out of device memory (resample flags)");
			s->rt_flags_n = n_px;
		}
		if(hipMemsetAsync(s->rt_flags, 0, n_px, nullptr) != hipSuccess) return fail(-100, "flag reset failed");
		if(w > 1 && h > 1)
		{
			const uint32_t grid = (uint32_t)std::min<size_t>(((size_t)(w - 1) * (size_t)(h - 1) + kBlock - 1) / kBlock, 4096);
			hipLaunchKernelGGL(aa_detect_kernel, dim3(grid), dim3(kBlock), 0, nullptr, (const float *)d_film, w, h, aa, aa_thesh, s->rt_flags);
			if(hipGetLastError() != hipSuccess) return fail(-100, "detection kernel launch failed");
		}
		if(hipMemcpy(flags.data(), s->rt_flags, n_px, hipMemcpyDeviceToHost) != hipSuccess) return fail(-100, "flag download failed");
		int n = 0;
		for(uint8_t f : flags) n += f;
		return n;
	};
	// integrator_tiled.cc:136-258
	const int aa_samples = std::max(1, rp.aa_minsamples);
	const int aa_inc = aa.inc_samples > 0 ? aa.inc_samples : aa_samples;       // scene.cc:765
	float threshold = aa.threshold, sample_mult = 1.f, light_mult = 1.f;
	const int floor_pixels = (int)std::floor(aa.resampled_floor * (float)(w * h) / 100.f);
	rp.aa_minsamples = aa_samples; rp.multi_pass = aa.passes > 1 ? 1 : 0; rp.pass_offset = 0u; rp.accumulate = 0; rp.resample_mask = nullptr;
	if(aa.passes > 1) rp.aa_light_sample_multiplier = light_mult;
	// the libc rand() stream the tile seeds come from (integrator_tiled.cc:319): one value per tile per pass that runs, after
	// the values the last Material / ObjectGeometric constructor consumed (aa.rand_skip)
	std::vector<int32_t> rand_stream;
	const int n_tiles_frame = ((w + rp.tile_size - 1) / std::max(rp.tile_size, 1)) * ((h + rp.tile_size - 1) / std::max(rp.tile_size, 1));
	size_t rand_pos = (size_t)std::max(aa.rand_skip, 0);
	if(aa.rand_srand >= 0 && rp.tile_size > 0)
	{
		rand_stream.resize(rand_pos + (size_t)n_tiles_frame * (size_t)aa.passes);
		yafgpu_glibc_rand((uint32_t)aa.rand_srand, (int32_t)rand_stream.size(), rand_stream.data());
		rp.tile_rand = rand_stream.data() + rand_pos; rand_pos += (size_t)n_tiles_frame;
	}
	int rc = yafgpu_render_tiles(s, &rp, d_planes, d_cnt, nullptr);
	if(resampled_out) resampled_out[0] = w * h;
	std::vector<uint8_t> mask;
	int acum = aa_samples, resampled = 0; bool threshold_changed = true;
	for(int i = 1; i < aa.passes && !rc; ++i)
	{
		if(s->aborted()) { rc = fail(-30, "aborted"); break; }
		sample_mult *= aa.sample_multiplier_factor;
		light_mult *= aa.light_sample_multiplier_factor;
		if(!(resampled <= 0 && !threshold_changed))
		{
			if((rc = exchange ? film_of_all_ranks(detect_on_host) : film_now(detect_on_host))) break;
			if(detect_on_host) resampled = next_pass_mask(h_film, w, h, aa, threshold, mask);
			else if((resampled = detect(threshold, mask)) < 0) { rc = resampled; break; }
			threshold_changed = false;
		}
		const int n = (int)std::ceil((float)aa_inc * sample_mult);
		if(resampled_out) resampled_out[i] = resampled > 0 ? resampled : 0;
		if(resampled > 0)
		{
			rp.aa_minsamples = n; rp.pass_offset = (uint32_t)acum; rp.accumulate = 1;
			rp.aa_light_sample_multiplier = light_mult;
			rp.resample_mask = threshold > 0.f ? mask.data() : nullptr;
			if(!rand_stream.empty()) { rp.tile_rand = rand_stream.data() + rand_pos; rand_pos += (size_t)n_tiles_frame; }
			rc = yafgpu_render_tiles(s, &rp, d_planes, d_cnt, nullptr);
		}
		acum += n;
		if(resampled < floor_pixels)
		{
			const float ratio = std::min(8.f, ((float)floor_pixels / (float)resampled));
			threshold *= (1.f - 0.1f * ratio);
			if(threshold > 0.f) threshold_changed = true;
		}
	}
	if(!rc) rc = film_now();
	if(!rc)
	{
		hipError_t e = hipDeviceSynchronize();
		if(e != hipSuccess) rc = fail(-100, std::string("render: ") + hipGetErrorString(e));
	}
	if(!rc && h_counters && hipMemcpy(h_counters, d_cnt, sizeof(yafgpu_counters), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(-100, "counter download failed");
	return rc;
}

static int trace_batch(yafgpu_scene_t *s, int32_t n, const float *rays, int32_t *tri, float *t, float *bary, int32_t *shadowed, bool any)
{
	if(!s || !rays || n < 0) return fail(-1, "bad argument");
	if(n == 0) return 0;
	DevMem<float> d_rays, d_t, d_b; DevMem<int> d_tri, d_sh;
	HIP_OK(d_rays.alloc((size_t)n * 8));
	HIP_OK(hipMemcpy(d_rays, rays, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice));
	HIP_OK(d_tri.alloc((size_t)n));
	HIP_OK(d_t.alloc((size_t)n));
	HIP_OK(d_b.alloc((size_t)n * 3));
	HIP_OK(d_sh.alloc((size_t)n));
	const uint32_t grid = (uint32_t)std::min((n + kBlock - 1) / kBlock, 4096);
	if(any) hipLaunchKernelGGL(trace_kernel<true>, dim3(grid), dim3(kBlock), 0, nullptr, s->dev, n, d_rays.p, d_tri.p, d_t.p, d_b.p, d_sh.p);
	else hipLaunchKernelGGL(trace_kernel<false>, dim3(grid), dim3(kBlock), 0, nullptr, s->dev, n, d_rays.p, d_tri.p, d_t.p, d_b.p, d_sh.p);
	HIP_OK(hipGetLastError());
	HIP_OK(hipDeviceSynchronize());
	if(any) HIP_OK(hipMemcpy(shadowed, d_sh, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
	else
	{
		HIP_OK(hipMemcpy(tri, d_tri, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy(t, d_t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy(bary, d_b, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
	}
	return 0;
}

int yafgpu_trace_closest(yafgpu_scene_t *s, int32_t n, const float *rays, int32_t *tri, float *t, float *bary)
{
	if(!tri || !t || !bary) return fail(-1, "null output");
	return trace_batch(s, n, rays, tri, t, bary, nullptr, false);
}
int yafgpu_trace_shadow(yafgpu_scene_t *s, int32_t n, const float *rays, int32_t *shadowed)
{
	if(!shadowed) return fail(-1, "null output");
	return trace_batch(s, n, rays, nullptr, nullptr, nullptr, shadowed, true);
}

void yafgpu_glibc_rand(uint32_t seed, int32_t count, int32_t *out)
{	// glibc stdlib/random_r.c, TYPE_3: 31 words seeded by the Lehmer generator 16807 x mod 2^31 - 1 (Schrage's method), then
	// r[i] = r[i-3] + r[i-31]; the first 310 values are discarded; rand() returns r >> 1
	if(count <= 0 || !out) return;
	std::vector<uint32_t> r((size_t)count + 344);
	if(seed == 0u) seed = 1u;
	r[0] = seed;
	for(int i = 1; i < 31; ++i)
	{
		int32_t word = (int32_t)r[(size_t)i - 1];
		const long hi = word / 127773, lo = word % 127773;
		word = (int32_t)(16807 * lo - 2836 * hi);
		if(word < 0) word += 2147483647;
		r[(size_t)i] = (uint32_t)word;
	}
	for(size_t i = 31; i < 34; ++i) r[i] = r[i - 31];
	for(size_t i = 34; i < r.size(); ++i) r[i] = r[i - 31] + r[i - 3];
	for(int32_t k = 0; k < count; ++k) out[k] = (int32_t)(r[(size_t)k + 344] >> 1);
}

int yafgpu_scene_set_exchange(yafgpu_scene_t *s, yafgpu_exchange_fn fn, void *user)
{
	if(!s) return fail(-1, "null argument");
	s->exchange = fn; s->exchange_user = user;
	return 0;
}

int yafgpu_scene_set_abort_flag(yafgpu_scene_t *s, const volatile int32_t *flag)
{
	if(!s) return fail(-1, "null argument");
	s->abort_flag = flag;
	return 0;
}

int yafgpu_scene_set_pass_pipelining(yafgpu_scene_t *s, int32_t mode)
{
	if(!s) return fail(-1, "null scene");
	s->pass_pipelining = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
	return 0;
}
int yafgpu_set_profiling(yafgpu_scene_t *s, int32_t enable)
{
	if(!s) return fail(-1, "null argument");
	s->profiling = enable != 0;
	return 0;
}
int yafgpu_get_profile(const yafgpu_scene_t *s, double ms[4], uint64_t launches[4])
{
	if(!s) return fail(-1, "null argument");
	for(int k = 0; k < 4; ++k) { ms[k] = s->prof_ms[k]; launches[k] = s->prof_launches[k]; }
	return 0;
}

int yafgpu_probe(yafgpu_scene_t *s, int32_t op, int32_t n, const float *in, int32_t n_in, float *out, int32_t n_out)
{
	if(!s || !in || !out || n < 0 || n_in <= 0 || n_out <= 0) return fail(-1, "bad argument");
	if(n == 0) return 0;
	DevMem<float> d_in, d_out;
	HIP_OK(d_in.alloc((size_t)n * (size_t)n_in));
	HIP_OK(d_out.alloc((size_t)n * (size_t)n_out));
	HIP_OK(hipMemcpy(d_in, in, (size_t)n * (size_t)n_in * sizeof(float), hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_out, 0, (size_t)n * (size_t)n_out * sizeof(float)));
	hipLaunchKernelGGL(probe_kernel, dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, s->dev, op, n, d_in.p, n_in, d_out.p, n_out);
	HIP_OK(hipGetLastError());
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipMemcpy(out, d_out, (size_t)n * (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost));
	return 0;
}

} // extern "C"
#endif // YAFGPU_VARIANT_TU
