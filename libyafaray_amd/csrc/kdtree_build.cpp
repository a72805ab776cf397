// Flattened SAH kd-tree build (host).  See kdtree_build.h.
//
// Strategy: recursive top-down build; nodes with many primitives pick their plane by binning the
// primitive bounds (64 bins per axis), small nodes (<= 48 prims) by an exact sweep over the sorted
// bound edges.  Subtrees below a fan-out level are built by a small thread pool into private arrays
// and spliced into depth-first order afterwards, so a 1 M-triangle tree does not serialise on one
// host core the way the reference's does (6.1 s measured).
#include "kdtree_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>

namespace yafgpu {
namespace {

struct Box { float lo[3], hi[3]; };

struct Params
{
	const Box *prim_box;
	const float *verts;
	int depth_cap;
	float cost_ratio, empty_bonus;
};

struct Sub   // a subtree in private storage, indices relative to its own arrays
{
	std::vector<KdNode> nodes;
	std::vector<uint32_t> refs;
	int depth = 0;
};

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

#ifndef KD_BINS
#define KD_BINS 64
#endif
#ifndef KD_SWEEP
#define KD_SWEEP 48
#endif
constexpr int kBins = KD_BINS;
constexpr int kSweepMax = KD_SWEEP;

struct Split { int axis = -1; float pos = 0.f; float cost = INFINITY; };

// SAH cost of a plane, with the reference's empty-space bonus model: the bonus grows with the
// fraction of the node the empty side takes (kdtree_triangle.cc:431-433) and decays with depth
// (:498: e_bonus *= 1.1 - depth/max_depth), so thin empty slivers deep in the tree are not cut off.
inline float sah_cost(const Params &p, const float d[3], int axis, float l1, uint32_t nl, uint32_t nr, float inv_total_sa, float e_bonus)
{
	const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
	const float cap = d[a1] * d[a2], rim = d[a1] + d[a2];
	const float l2 = d[axis] - l1;
	const float below = cap + l1 * rim, above = cap + l2 * rim;
	const float raw = below * (float)nl + above * (float)nr;
	float eb = 0.f;
	if(nr == 0) eb = (0.1f + l2 / d[axis]) * e_bonus * raw;
	else if(nl == 0) eb = (0.1f + l1 / d[axis]) * e_bonus * raw;
	return p.cost_ratio + inv_total_sa * (raw - eb);
}

// Bounds of (triangle ∩ box), by Sutherland-Hodgman clipping in double precision ("perfect splits";
// the reference clips too, for nodes of <= 32 prims: kdtree_triangle.cc:483-515).  The box is grown
// by a small margin before clipping so that a triangle is only ever dropped from a subtree when it is
// clearly outside it; the result is intersected with the node box.
bool clip_tri_to_box(const float *v, const Box &box, Box &out)
{
	double poly[16][3], tmp[16][3];
	int n = 3;
	for(int i = 0; i < 3; ++i) for(int k = 0; k < 3; ++k) poly[i][k] = v[3 * i + k];
	for(int axis = 0; axis < 3 && n > 0; ++axis)
	{
		const double ext = (double)box.hi[axis] - (double)box.lo[axis];
		const double margin = 1e-5 * ext + 1e-7 * (std::fabs((double)box.lo[axis]) + std::fabs((double)box.hi[axis])) + 1e-30;
		for(int side = 0; side < 2 && n > 0; ++side)
		{
			const double plane = side == 0 ? (double)box.lo[axis] - margin : (double)box.hi[axis] + margin;
			int m = 0;
			for(int i = 0; i < n; ++i)
			{
				const double *p = poly[i], *q = poly[(i + 1) % n];
				const bool pin = side == 0 ? p[axis] >= plane : p[axis] <= plane;
				const bool qin = side == 0 ? q[axis] >= plane : q[axis] <= plane;
				if(pin) { for(int k = 0; k < 3; ++k) tmp[m][k] = p[k]; ++m; }
				if(pin != qin)
				{
					const double t = (plane - p[axis]) / (q[axis] - p[axis]);
					for(int k = 0; k < 3; ++k) tmp[m][k] = p[k] + t * (q[k] - p[k]);
					tmp[m][axis] = plane;
					++m;
				}
			}
			n = m;
			for(int i = 0; i < n; ++i) for(int k = 0; k < 3; ++k) poly[i][k] = tmp[i][k];
		}
	}
	if(n == 0) return false;
	for(int k = 0; k < 3; ++k)
	{
		double lo = poly[0][k], hi = poly[0][k];
		for(int i = 1; i < n; ++i) { lo = std::min(lo, poly[i][k]); hi = std::max(hi, poly[i][k]); }
		// round outward, then clamp to the node box
		float flo = (float)lo, fhi = (float)hi;
		if((double)flo > lo) flo = std::nextafter(flo, -INFINITY);
		if((double)fhi < hi) fhi = std::nextafter(fhi, INFINITY);
		out.lo[k] = std::max(flo, box.lo[k]);
		out.hi[k] = std::min(fhi, box.hi[k]);
		if(out.lo[k] > out.hi[k]) { const float mid = std::min(std::max(flo, box.lo[k]), box.hi[k]); out.lo[k] = out.hi[k] = mid; }
	}
	return true;
}

Split find_split_binned(const Params &p, const Box &box, const uint32_t *prims, uint32_t np, float e_bonus)
{
	Split best;
	float d[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
	const float total_sa = d[0] * d[1] + d[0] * d[2] + d[1] * d[2];
	if(!(total_sa > 0.f)) return best;
	const float inv_total_sa = 1.f / total_sa;
	for(int axis = 0; axis < 3; ++axis)
	{
		if(!(d[axis] > 0.f)) continue;
		uint32_t starts[kBins + 1] = {0}, ends[kBins + 1] = {0};
		const float scale = (float)kBins / d[axis], lo = box.lo[axis];
		for(uint32_t i = 0; i < np; ++i)
		{
			const Box &b = p.prim_box[prims[i]];
			int s = (int)std::floor((b.lo[axis] - lo) * scale);
			int e = (int)std::ceil((b.hi[axis] - lo) * scale);
			s = std::min(std::max(s, 0), kBins);
			e = std::min(std::max(e, 0), kBins);
			++starts[s]; ++ends[e];
		}
		uint32_t nl = 0, nr = np;
		for(int k = 1; k < kBins; ++k)
		{
			nl += starts[k - 1];   // prims beginning before plane k
			nr -= ends[k];         // prims ending at or before plane k no longer reach the right side
			const float l1 = (float)k / scale;
			const float pos = lo + l1;
			// a plane that rounds onto a face of the node splits nothing off (same box, same prims, one level deeper —
			// and the empty-side bonus would pick it again and again in a node squeezed onto coplanar prims)
			if(!(pos > box.lo[axis] && pos < box.hi[axis])) continue;
			const float c = sah_cost(p, d, axis, l1, nl, nr, inv_total_sa, e_bonus);
			if(c < best.cost) { best.cost = c; best.axis = axis; best.pos = pos; }
		}
	}
	return best;
}

Split find_split_sweep(const Params &p, const Box &box, const Box *pbox, uint32_t np, float e_bonus)
{
	Split best;
	float d[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
	const float total_sa = d[0] * d[1] + d[0] * d[2] + d[1] * d[2];
	if(!(total_sa > 0.f)) return best;
	const float inv_total_sa = 1.f / total_sa;
	struct Edge { float pos; int is_end; };
	Edge edges[2 * kSweepMax];
	for(int axis = 0; axis < 3; ++axis)
	{
		if(!(d[axis] > 0.f)) continue;
		for(uint32_t i = 0; i < np; ++i)
		{
			const Box &b = pbox[i];
			edges[2 * i] = {b.lo[axis], 0};
			edges[2 * i + 1] = {b.hi[axis], 1};
		}
		// ends sort before starts at equal position so a plane through touching prims separates them
		std::sort(edges, edges + 2 * np, [](const Edge &x, const Edge &y) { return x.pos == y.pos ? x.is_end > y.is_end : x.pos < y.pos; });
		uint32_t nl = 0, nr = np;
		for(uint32_t i = 0; i < 2 * np; ++i)
		{
			if(edges[i].is_end) --nr;
			const float pos = edges[i].pos;
			if(pos > box.lo[axis] && pos < box.hi[axis])
			{
				const float c = sah_cost(p, d, axis, pos - box.lo[axis], nl, nr, inv_total_sa, e_bonus);
				if(c < best.cost) { best.cost = c; best.axis = axis; best.pos = pos; }
			}
			if(!edges[i].is_end) ++nl;
		}
	}
	return best;
}

void emit_leaf(Sub &s, const uint32_t *prims, uint32_t np, int depth)
{
	KdNode n;
	n.a = (uint32_t)s.refs.size();
	n.b = 3u | (np << 2);
	s.nodes.push_back(n);
	// ascending triangle index inside a leaf: the first-visited-wins tie rule of the traversal
	// (kdtree_triangle.cc:782) then prefers the lower index, independent of build order
	size_t at = s.refs.size();
	s.refs.insert(s.refs.end(), prims, prims + np);
	std::sort(s.refs.begin() + (long)at, s.refs.end());
	s.depth = std::max(s.depth, depth);
}

// returns false when the node must become a leaf (prims may have shrunk: triangles that the exact
// clip shows to lie outside this node are removed)
bool choose_and_partition(const Params &p, const Box &box, std::vector<uint32_t> &prims, int depth, int &bad_refines,
                          Split &sp, std::vector<uint32_t> &left, std::vector<uint32_t> &right)
{
	uint32_t np = (uint32_t)prims.size();
	if(np <= 1 || depth >= p.depth_cap) return false;
	const float e_bonus = p.empty_bonus * (1.1f - (float)depth / (float)p.depth_cap);
	Box local[kSweepMax];
	const bool small = np <= (uint32_t)kSweepMax;
	if(small)
	{
		uint32_t m = 0;
		for(uint32_t i = 0; i < np; ++i)
		{
			Box b;
			if(clip_tri_to_box(p.verts + 9 * (size_t)prims[i], box, b)) { local[m] = b; prims[m] = prims[i]; ++m; }
		}
		prims.resize(m);
		np = m;
		if(np <= 1) return false;
		sp = find_split_sweep(p, box, local, np, e_bonus);
	}
	else sp = find_split_binned(p, box, prims.data(), np, e_bonus);
	if(sp.axis < 0) return false;
	const float leaf_cost = (float)np;
	if(sp.cost > leaf_cost) ++bad_refines;
	if((sp.cost > 1.6f * leaf_cost && np < 16) || bad_refines >= 2) return false;
	left.clear(); right.clear();
	for(uint32_t i = 0; i < np; ++i)
	{
		const Box &b = small ? local[i] : p.prim_box[prims[i]];
		// same membership as the cost model: a prim that merely touches the plane belongs to one side
		// (its hit at t == t_plane is still found: the traversal accepts any t below the current best,
		// not only hits inside the cell); a prim lying in the plane goes left
		const float lo = b.lo[sp.axis], hi = b.hi[sp.axis];
		if(lo < sp.pos || (lo == sp.pos && hi == sp.pos)) left.push_back(prims[i]);
		if(hi > sp.pos) right.push_back(prims[i]);
	}
	if(left.size() == np && right.size() == np) return false;
	return true;
}

void build_seq(const Params &p, Sub &s, const Box &box, std::vector<uint32_t> &prims, int depth, int bad_refines)
{
	Split sp;
	std::vector<uint32_t> left, right;
	if(!choose_and_partition(p, box, prims, depth, bad_refines, sp, left, right)) { emit_leaf(s, prims.data(), (uint32_t)prims.size(), depth); return; }
	std::vector<uint32_t>().swap(prims); // release before recursing
	const size_t me = s.nodes.size();
	s.nodes.push_back({f2u(sp.pos), (uint32_t)sp.axis});
	Box lb = box, rb = box;
	lb.hi[sp.axis] = sp.pos; rb.lo[sp.axis] = sp.pos;
	build_seq(p, s, lb, left, depth + 1, bad_refines);
	s.nodes[me].b = (uint32_t)sp.axis | ((uint32_t)s.nodes.size() << 2);
	build_seq(p, s, rb, right, depth + 1, bad_refines);
}

void splice(Sub &dst, const Sub &src)
{
	const uint32_t node_off = (uint32_t)dst.nodes.size(), ref_off = (uint32_t)dst.refs.size();
	for(KdNode n : src.nodes)
	{
		if((n.b & 3u) == 3u) n.a += ref_off;
		else n.b = (n.b & 3u) | (((n.b >> 2) + node_off) << 2);
		dst.nodes.push_back(n);
	}
	dst.refs.insert(dst.refs.end(), src.refs.begin(), src.refs.end());
	dst.depth = std::max(dst.depth, src.depth);
}

// top levels: fan out into futures until fan_depth, then sequential subtrees
Sub build_par(const Params &p, const Box &box, std::vector<uint32_t> prims, int depth, int bad_refines, int fan_depth)
{
	Sub s;
	if(depth >= fan_depth || prims.size() < 4096)
	{
		build_seq(p, s, box, prims, depth, bad_refines);
		return s;
	}
	Split sp;
	std::vector<uint32_t> left, right;
	if(!choose_and_partition(p, box, prims, depth, bad_refines, sp, left, right)) { emit_leaf(s, prims.data(), (uint32_t)prims.size(), depth); return s; }
	std::vector<uint32_t>().swap(prims);
	Box lb = box, rb = box;
	lb.hi[sp.axis] = sp.pos; rb.lo[sp.axis] = sp.pos;
	auto fr = std::async(std::launch::async, [&p, rb, depth, bad_refines, fan_depth](std::vector<uint32_t> r) {
		return build_par(p, rb, std::move(r), depth + 1, bad_refines, fan_depth);
	}, std::move(right));
	Sub ls = build_par(p, lb, std::move(left), depth + 1, bad_refines, fan_depth);
	Sub rs = fr.get();
	s.nodes.reserve(1 + ls.nodes.size() + rs.nodes.size());
	s.refs.reserve(ls.refs.size() + rs.refs.size());
	s.nodes.push_back({f2u(sp.pos), (uint32_t)sp.axis});
	splice(s, ls);
	s.nodes[0].b = (uint32_t)sp.axis | ((uint32_t)s.nodes.size() << 2);
	splice(s, rs);
	return s;
}

} // namespace

void build_kdtree(const float *verts, int n_tris, int depth_cap, int threads, KdTree &out)
{
	const auto t0 = std::chrono::steady_clock::now();
	out.nodes.clear(); out.refs.clear(); out.max_depth = 0;
	for(int k = 0; k < 3; ++k) { out.bound_lo[k] = 0.f; out.bound_hi[k] = 0.f; }
	if(n_tris <= 0) { out.build_seconds = 0; return; }
	std::vector<Box> boxes((size_t)n_tris);
	Box all;
	for(int i = 0; i < n_tris; ++i)
	{
		const float *v = verts + 9 * (size_t)i;
		Box &b = boxes[(size_t)i];
		for(int k = 0; k < 3; ++k)
		{
			b.lo[k] = std::min(v[k], std::min(v[3 + k], v[6 + k]));
			b.hi[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
		}
		if(i == 0) all = b;
		else for(int k = 0; k < 3; ++k) { all.lo[k] = std::min(all.lo[k], b.lo[k]); all.hi[k] = std::max(all.hi[k], b.hi[k]); }
	}
	// the reference grows its tree bound by 0.1 % per side (kdtree_triangle.cc:110-116); same here so
	// that rays clipped against it enter and leave at the same distances
	for(int k = 0; k < 3; ++k)
	{
		const double grow = (double)(all.hi[k] - all.lo[k]) * 0.001;
		all.lo[k] = (float)((double)all.lo[k] - grow);
		all.hi[k] = (float)((double)all.hi[k] + grow);
	}
	Params p;
	p.prim_box = boxes.data();
	p.verts = verts;
	int md = (int)(7.0f + 1.66f * std::log((float)n_tris)); // kdtree_triangle.cc:89
	p.depth_cap = std::min(md, depth_cap);
	p.cost_ratio = 0.8f; p.empty_bonus = 0.33f;             // scene.cc:818
	const double log_leaves = 1.442695f * std::log((double)n_tris); // kdtree_triangle.cc:90,100
	if(log_leaves > 16.0) p.cost_ratio += (float)(0.25 * (log_leaves - 16.0));
	if(const char *e = std::getenv("YAFGPU_COST_RATIO")) p.cost_ratio = (float)std::atof(e);      // experiments: node-step cost / triangle-test cost
	if(threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
	int fan_depth = 0;
	while((1 << fan_depth) < 2 * threads && fan_depth < 8) ++fan_depth;
	if(threads == 1) fan_depth = 0;
	std::vector<uint32_t> prims((size_t)n_tris);
	for(int i = 0; i < n_tris; ++i) prims[(size_t)i] = (uint32_t)i;
	Sub root = build_par(p, all, std::move(prims), 0, 0, fan_depth);
	out.nodes = std::move(root.nodes);
	out.refs = std::move(root.refs);
	out.max_depth = root.depth;
	for(int k = 0; k < 3; ++k) { out.bound_lo[k] = all.lo[k]; out.bound_hi[k] = all.hi[k]; }
	out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace yafgpu

// ---- host-only C entry points (include/yafgpu.h) ----
#include "../../include/yafgpu.h"
struct yafgpu_kdtree { yafgpu::KdTree t; int n_tris; };
extern "C" {
yafgpu_kdtree_t *yafgpu_kdtree_build(const float *verts, int32_t n_tris, int32_t threads)
{
	auto *k = new yafgpu_kdtree();
	k->n_tris = n_tris;
	yafgpu::build_kdtree(verts, n_tris, 48, threads, k->t);
	return k;
}
void yafgpu_internal_set_error(const char *msg);    // yafgpu_device.hip: the library's last-error string
yafgpu_kdtree_t *yafgpu_kdtree_build_device(const float *verts, int32_t n_tris)
{
	auto *k = new yafgpu_kdtree();
	k->n_tris = n_tris;
	std::string err;
	if(yafgpu::build_kdtree_device_retry(verts, n_tris, 48, k->t, &err)) { yafgpu_internal_set_error(err.c_str()); delete k; return nullptr; }
	return k;
}
void yafgpu_kdtree_info(const yafgpu_kdtree_t *k, yafgpu_tree_info *info)
{
	info->n_nodes = (uint32_t)k->t.nodes.size(); info->n_leaf_refs = (uint32_t)k->t.refs.size();
	info->max_depth = (uint32_t)k->t.max_depth; info->n_tris = (uint32_t)k->n_tris;
	info->build_seconds = k->t.build_seconds; info->upload_seconds = 0; info->device_bytes = 0;
}
void yafgpu_kdtree_get(const yafgpu_kdtree_t *k, uint32_t *nodes, uint32_t *refs, float bound6[6])
{
	if(nodes && !k->t.nodes.empty()) std::memcpy(nodes, k->t.nodes.data(), k->t.nodes.size() * sizeof(yafgpu::KdNode));
	if(refs && !k->t.refs.empty()) std::memcpy(refs, k->t.refs.data(), k->t.refs.size() * sizeof(uint32_t));
	if(bound6) for(int i = 0; i < 3; ++i) { bound6[i] = k->t.bound_lo[i]; bound6[3 + i] = k->t.bound_hi[i]; }
}
void yafgpu_kdtree_destroy(yafgpu_kdtree_t *k) { delete k; }
}
