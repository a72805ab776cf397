// Device-side SAH kd-tree build.  Same tree format, build parameters and cost model as the host
// builder (kdtree_build.cpp: depth cap 7 + 1.66 ln N, cost ratio 0.8 (+ penalty above 65 536 prims), empty bonus 0.33
// scaled by the empty side's share and decaying with depth — kdtree_triangle.cc:89-100,431-433,498, scene.cc:818);
// what differs is how the work is laid out:
//
//   * breadth first: one launch per tree level, one workgroup per node of the level.  A node's references (triangle
//     index + its bounds clipped to the node box, 32 bytes) are a contiguous segment of the level's reference array;
//   * plane search: above 64 references the workgroup bins the references' extents into 32 bins per axis in LDS and
//     evaluates the 31 planes per axis; at or below it every reference edge is a candidate plane and each thread
//     counts the two sides of one candidate (the references' bounds sit in LDS);
//   * partition: the workgroup counts both sides, reserves its output segment in the next level's array with one
//     atomic, and scatters (a reference that straddles the plane goes to both sides, its bounds clipped to each);
//   * references are clipped box-against-box only (the host builder clips the triangle itself below 48 prims), so
//     leaves hold a few more references; results of queries do not depend on that.
//
// The level loop reads back the counters once per level.  The breadth-first node array is then flattened on the device
// into the depth-first order the traversal wants (near child = next node); leaf references are sorted by triangle
// index on the way, so the tree is the same whatever order the atomics resolved in.
#include "kdtree_build.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace yafgpu {
namespace {

constexpr int kBins = 32;
constexpr int kSmall = 64;         // at or below: exact candidates instead of bins

struct Ref { float lo[3]; uint32_t tri; float hi[3]; uint32_t pad; };            // 32 B
struct Work { uint32_t begin, count, node, depth_bad; float lo[3], hi[3]; };       // depth | bad_refines << 16
struct BfsNode { uint32_t a, b, c, d; };   // interior: split bits, axis, left, right; leaf: first, 3 | count << 2, 0, 0

struct BuildArgs
{
	const float *verts;            // 9 floats per triangle
	const Ref *refs_in; Ref *refs_out;
	const Work *work_in; Work *work_out;
	BfsNode *nodes; uint32_t *leaf_refs;
	uint32_t *counters;            // [0] refs out and [1] work items out (bumped together as one 64-bit word), [3] leaf refs, [4] overflow flag
	uint32_t n_work, node_base;    // children take node indices node_base + their work item's index

	uint32_t cap_refs, cap_work, cap_nodes, cap_leaf_refs;
	int depth_cap;
	float cost_ratio, empty_bonus;
};

__device__ float sah_cost(const float d[3], int axis, float l1, uint32_t nl, uint32_t nr, float inv_total_sa, float e_bonus, float cost_ratio)
{
	const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
	const float cap = d[a1] * d[a2], rim = d[a1] + d[a2];
	const float l2 = d[axis] - l1;
	const float below = cap + l1 * rim, above = cap + l2 * rim;
	const float raw = below * (float)nl + above * (float)nr;
	float eb = 0.f;
	if(nr == 0u) eb = (0.1f + l2 / d[axis]) * e_bonus * raw;
	else if(nl == 0u) eb = (0.1f + l1 / d[axis]) * e_bonus * raw;
	return cost_ratio + inv_total_sa * (raw - eb);
}

__global__ void init_refs(const float *verts, int n, Ref *refs)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const float *v = verts + 9 * (size_t)i;
	Ref r;
	for(int k = 0; k < 3; ++k)
	{
		r.lo[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
		r.hi[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
	}
	r.tri = (uint32_t)i; r.pad = 0u;
	refs[i] = r;
}


// Bounds of (triangle ∩ box) by Sutherland-Hodgman clipping in double precision, the same rule as the host builder's
// clip_tri_to_box (kdtree_build.cpp; the reference clips below 32 prims, kdtree_triangle.cc:483-515): the box is grown
// by a small margin first so that a triangle only leaves a subtree when it is clearly outside it, and the result is
// rounded outward and clamped to the box.  Returns false when nothing of the triangle is inside.
__device__ bool clip_tri_to_box(const float *v, const float blo[3], const float bhi[3], float out[6])
{
	double poly[10][3], tmp[10][3];
	int n = 3;
	for(int i = 0; i < 3; ++i) for(int k = 0; k < 3; ++k) poly[i][k] = (double)v[3 * i + k];
	for(int axis = 0; axis < 3 && n > 0; ++axis)
	{
		const double ext = (double)bhi[axis] - (double)blo[axis];
		const double margin = 1e-5 * ext + 1e-7 * (fabs((double)blo[axis]) + fabs((double)bhi[axis])) + 1e-30;
		for(int side = 0; side < 2 && n > 0; ++side)
		{
			const double plane = side == 0 ? (double)blo[axis] - margin : (double)bhi[axis] + margin;
			int m = 0;
			for(int i = 0; i < n; ++i)
			{
				const double *p = poly[i], *q = poly[(i + 1 == n) ? 0 : i + 1];
				const bool pin = side == 0 ? p[axis] >= plane : p[axis] <= plane;
				const bool qin = side == 0 ? q[axis] >= plane : q[axis] <= plane;
				if(pin && m < 10) { for(int k = 0; k < 3; ++k) tmp[m][k] = p[k]; ++m; }
				if(pin != qin && m < 10)
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
		for(int i = 1; i < n; ++i) { lo = fmin(lo, poly[i][k]); hi = fmax(hi, poly[i][k]); }
		float flo = (float)lo, fhi = (float)hi;
		if((double)flo > lo) flo = nextafterf(flo, -INFINITY);
		if((double)fhi < hi) fhi = nextafterf(fhi, INFINITY);
		float olo = fmaxf(flo, blo[k]), ohi = fminf(fhi, bhi[k]);
		if(olo > ohi) { const float mid = fminf(fmaxf(flo, blo[k]), bhi[k]); olo = ohi = mid; }
		out[k] = olo; out[3 + k] = ohi;
	}
	return true;
}

// one workgroup per node of the level; the host picks the workgroup size from the level's average node size
template<int kBuildBlock>
__global__ __launch_bounds__(kBuildBlock) void build_level(const BuildArgs a)
{
	__shared__ uint32_t s_starts[3][kBins + 1], s_ends[3][kBins + 1];
	__shared__ float s_box[kSmall][6];
	__shared__ uint32_t s_tri[kSmall], s_np;
	__shared__ float s_best_cost[kBuildBlock]; __shared__ uint32_t s_best_key[kBuildBlock]; __shared__ float s_best_pos[kBuildBlock];
	__shared__ uint32_t s_count[2], s_cursor[2], s_base[4];
	__shared__ float s_split; __shared__ int s_axis;
	const int tid = (int)threadIdx.x;
	for(uint32_t wi = blockIdx.x; wi < a.n_work; wi += gridDim.x)
	{
		const Work w = a.work_in[wi];
		uint32_t np = w.count;
		const bool small = np <= (uint32_t)kSmall;
		const int depth = (int)(w.depth_bad & 0xffffu);
		int bad = (int)(w.depth_bad >> 16);
		const float d[3] = {w.hi[0] - w.lo[0], w.hi[1] - w.lo[1], w.hi[2] - w.lo[2]};
		const float total_sa = d[0] * d[1] + d[0] * d[2] + d[1] * d[2];
		bool leaf = np <= 1u || depth >= a.depth_cap || !(total_sa > 0.f);
		float best_cost = INFINITY, best_pos = 0.f; int best_axis = -1;
		if(small)
		{	// small node: from here on its references live in LDS.  Unless it is a leaf already, each is clipped against
			// the node box first (its bounds tighten; one that misses the box altogether is dropped)
			if(tid < 64)
			{
				bool keep = false; float bx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; uint32_t tri = 0u;
				if((uint32_t)tid < np)
				{
					const Ref r = a.refs_in[w.begin + (uint32_t)tid];
					tri = r.tri; keep = true;
					if(!leaf) keep = clip_tri_to_box(a.verts + 9 * (size_t)tri, w.lo, w.hi, bx);
					else for(int k = 0; k < 3; ++k) { bx[k] = r.lo[k]; bx[3 + k] = r.hi[k]; }
				}
				const unsigned long long m = __ballot(keep);
				const uint32_t at = (uint32_t)__popcll(m & ((1ull << tid) - 1ull));
				if(keep) { for(int k = 0; k < 6; ++k) s_box[at][k] = bx[k]; s_tri[at] = tri; }
				if(tid == 0) s_np = (uint32_t)__popcll(m);
			}
			__syncthreads();
			np = s_np;
			if(np <= 1u) leaf = true;
		}
		if(!leaf)
		{
			const float inv_total_sa = 1.f / total_sa;
			const float e_bonus = a.empty_bonus * (1.1f - (float)depth / (float)a.depth_cap);
			float my_cost = INFINITY, my_pos = 0.f; uint32_t my_key = 0xffffffffu;      // key = axis << 28 | bin plane: ties go to the lowest
			if(np > (uint32_t)kSmall)
			{
				for(int i = tid; i < 3 * (kBins + 1); i += kBuildBlock) { (&s_starts[0][0])[i] = 0u; (&s_ends[0][0])[i] = 0u; }
				__syncthreads();
				// four references in flight per thread: a lone workgroup streaming a huge node is bound by load latency
				for(uint32_t i0 = (uint32_t)tid; i0 < np; i0 += 4u * kBuildBlock)
				{
					Ref r[4];
#pragma unroll
					for(int u = 0; u < 4; ++u) { const uint32_t i = i0 + (uint32_t)u * kBuildBlock; if(i < np) r[u] = a.refs_in[w.begin + i]; }
#pragma unroll
					for(int u = 0; u < 4; ++u)
					{
						if(i0 + (uint32_t)u * kBuildBlock >= np) continue;
						for(int axis = 0; axis < 3; ++axis)
						{
							if(!(d[axis] > 0.f)) continue;
							const float scale = (float)kBins / d[axis];
							int s = (int)floorf((r[u].lo[axis] - w.lo[axis]) * scale), e = (int)ceilf((r[u].hi[axis] - w.lo[axis]) * scale);
							s = min(max(s, 0), kBins); e = min(max(e, 0), kBins);
							atomicAdd(&s_starts[axis][s], 1u); atomicAdd(&s_ends[axis][e], 1u);
						}
					}
				}
				__syncthreads();
				// thread (axis, k): plane k of the axis; the prefix sums are short enough to redo per thread
				for(int t = tid; t < 3 * (kBins - 1); t += kBuildBlock)
				{
					const int axis = t / (kBins - 1), k = 1 + t % (kBins - 1);
					if(!(d[axis] > 0.f)) continue;
					uint32_t nl = 0u, nr = np;
					for(int j = 1; j <= k; ++j) { nl += s_starts[axis][j - 1]; nr -= s_ends[axis][j]; }
					const float l1 = (float)k / ((float)kBins / d[axis]);
					const float pos = w.lo[axis] + l1;
					if(!(pos > w.lo[axis] && pos < w.hi[axis])) continue;      // rounded onto a face of the node: splits nothing off
					const float c = sah_cost(d, axis, l1, nl, nr, inv_total_sa, e_bonus, a.cost_ratio);
					const uint32_t key = ((uint32_t)axis << 28) | (uint32_t)k;
					if(c < my_cost || (c == my_cost && key < my_key)) { my_cost = c; my_pos = pos; my_key = key; }
				}
			}
			else
			{
				// candidate t: axis, reference, which edge
				for(int t = tid; t < (int)np * 6; t += kBuildBlock)
				{
					const int axis = t / (2 * (int)np), rest = t % (2 * (int)np), j = rest >> 1, hi_edge = rest & 1;
					if(!(d[axis] > 0.f)) continue;
					const float pos = s_box[j][axis + 3 * hi_edge];
					if(!(pos > w.lo[axis] && pos < w.hi[axis])) continue;
					uint32_t nl = 0u, nr = 0u;
					for(int q = 0; q < (int)np; ++q)
					{
						const float lo = s_box[q][axis], hi = s_box[q][axis + 3];
						nl += (lo < pos || (lo == pos && hi == pos)) ? 1u : 0u;
						nr += (hi > pos) ? 1u : 0u;
					}
					const float c = sah_cost(d, axis, pos - w.lo[axis], nl, nr, inv_total_sa, e_bonus, a.cost_ratio);
					// ties: lowest axis, then lowest plane -- never the reference's place in the segment, which the atomics decide
					const uint32_t key = (uint32_t)axis << 28;
					if(c < my_cost || (c == my_cost && (key < my_key || (key == my_key && pos < my_pos)))) { my_cost = c; my_pos = pos; my_key = key; }
				}
			}
			s_best_cost[tid] = my_cost; s_best_key[tid] = my_key; s_best_pos[tid] = my_pos;
			__syncthreads();
			for(int step = kBuildBlock / 2; step > 0; step >>= 1)
			{
				if(tid < step)
				{
					const float c2 = s_best_cost[tid + step]; const uint32_t k2 = s_best_key[tid + step];
					if(c2 < s_best_cost[tid] || (c2 == s_best_cost[tid] && (k2 < s_best_key[tid] || (k2 == s_best_key[tid] && s_best_pos[tid + step] < s_best_pos[tid]))))
					{ s_best_cost[tid] = c2; s_best_key[tid] = k2; s_best_pos[tid] = s_best_pos[tid + step]; }
				}
				__syncthreads();
			}
			best_cost = s_best_cost[0]; best_pos = s_best_pos[0];
			best_axis = (s_best_key[0] == 0xffffffffu) ? -1 : (int)(s_best_key[0] >> 28);
			__syncthreads();
			if(best_axis < 0) leaf = true;
			else
			{
				const float leaf_cost = (float)np;
				if(best_cost > leaf_cost) ++bad;
				if((best_cost > 1.6f * leaf_cost && np < 16u) || bad >= 2) leaf = true;
			}
		}
		if(!leaf)
		{	// count both sides
			if(tid < 2) s_count[tid] = 0u;
			__syncthreads();
			uint32_t nl = 0u, nr = 0u;
			if(small)
			{
				for(uint32_t i = (uint32_t)tid; i < np; i += kBuildBlock)
				{
					const float lo = s_box[i][best_axis], hi = s_box[i][best_axis + 3];
					nl += (lo < best_pos || (lo == best_pos && hi == best_pos)) ? 1u : 0u;
					nr += (hi > best_pos) ? 1u : 0u;
				}
			}
			else
			{
				for(uint32_t i0 = (uint32_t)tid; i0 < np; i0 += 4u * kBuildBlock)
				{
					float lo[4], hi[4];
#pragma unroll
					for(int u = 0; u < 4; ++u)
					{
						const uint32_t i = i0 + (uint32_t)u * kBuildBlock;
						lo[u] = INFINITY; hi[u] = -INFINITY;       // counts on neither side
						if(i < np) { const Ref &r = a.refs_in[w.begin + i]; lo[u] = r.lo[best_axis]; hi[u] = r.hi[best_axis]; }
					}
#pragma unroll
					for(int u = 0; u < 4; ++u)
					{
						nl += (lo[u] < best_pos || (lo[u] == best_pos && hi[u] == best_pos)) ? 1u : 0u;
						nr += (hi[u] > best_pos) ? 1u : 0u;
					}
				}
			}
			if(nl) atomicAdd(&s_count[0], nl);
			if(nr) atomicAdd(&s_count[1], nr);
			__syncthreads();
			if(s_count[0] == np && s_count[1] == np) leaf = true;
			__syncthreads();
		}
		if(leaf)
		{
			if(tid == 0)
			{
				uint32_t first = atomicAdd(&a.counters[3], np);
				if(first + np > a.cap_leaf_refs) { a.counters[4] = 1u; first = 0u; s_base[0] = 0xffffffffu; }
				else s_base[0] = first;
				BfsNode n; n.a = first; n.b = 3u | (np << 2); n.c = 0u; n.d = 0u;
				a.nodes[w.node] = n;
			}
			__syncthreads();
			const uint32_t first = s_base[0];
			if(first != 0xffffffffu)
				for(uint32_t i = (uint32_t)tid; i < np; i += kBuildBlock) a.leaf_refs[first + i] = small ? s_tri[i] : a.refs_in[w.begin + i].tri;
			__syncthreads();
			continue;
		}
		// reserve the children: output segment, two nodes, two work items
		if(tid == 0)
		{
			const uint32_t nl = s_count[0], nr = s_count[1];
			// every node of the level bumps these: one 64-bit atomic (low word references, high word work items) keeps the
			// same-address traffic at one operation per node
			const unsigned long long got = atomicAdd((unsigned long long *)a.counters, (unsigned long long)(nl + nr) | (2ull << 32));
			const uint32_t out = (uint32_t)got, wk = (uint32_t)(got >> 32), nd = a.node_base + wk;
			const bool over = out + nl + nr > a.cap_refs || wk + 2u > a.cap_work || nd + 2u > a.cap_nodes;
			if(over) a.counters[4] = 1u;
			s_base[0] = over ? 0xffffffffu : out; s_base[1] = wk; s_base[2] = nd;
			s_cursor[0] = 0u; s_cursor[1] = 0u;
			s_split = best_pos; s_axis = best_axis;
			if(!over)
			{
				BfsNode n; n.a = __float_as_uint(best_pos); n.b = (uint32_t)best_axis; n.c = nd; n.d = nd + 1u;
				a.nodes[w.node] = n;
				Work wl = w, wr = w;
				wl.begin = out; wl.count = nl; wl.node = nd; wl.depth_bad = (uint32_t)(depth + 1) | ((uint32_t)bad << 16); wl.hi[best_axis] = best_pos;
				wr.begin = out + nl; wr.count = nr; wr.node = nd + 1u; wr.depth_bad = wl.depth_bad; wr.lo[best_axis] = best_pos;
				a.work_out[wk] = wl; a.work_out[wk + 1u] = wr;
			}
			else
			{	// out of room: close the node as a leaf so that the arrays stay consistent; the host reports the overflow
				BfsNode n; n.a = 0u; n.b = 3u; n.c = 0u; n.d = 0u;
				a.nodes[w.node] = n;
			}
		}
		__syncthreads();
		const uint32_t out = s_base[0];
		if(out != 0xffffffffu)
		{
			const uint32_t nl = s_count[0];
			const int axis = s_axis; const float pos = s_split;
			// a wave reserves its slots with one LDS atomic per side
			const int lane = tid & 63;
			auto fetch = [&](uint32_t i, Ref &r) -> bool
			{
				if(i >= np) return false;
				if(small) { for(int k = 0; k < 3; ++k) { r.lo[k] = s_box[i][k]; r.hi[k] = s_box[i][3 + k]; } r.tri = s_tri[i]; }
				else r = a.refs_in[w.begin + i];
				return true;
			};
			Ref ahead{};
			bool ahead_valid = fetch((uint32_t)tid, ahead);
			for(uint32_t base = 0u; base < np; base += kBuildBlock)
			{
				const Ref r = ahead;
				const bool valid = ahead_valid;
				ahead_valid = fetch(base + kBuildBlock + (uint32_t)tid, ahead);      // the next round's load overlaps this round's stores
				const float lo = r.lo[axis], hi = r.hi[axis];
				const bool go_l = valid && (lo < pos || (lo == pos && hi == pos)), go_r = valid && hi > pos;
				const unsigned long long ml = __ballot(go_l), mr = __ballot(go_r);
				uint32_t bl = 0u, br = 0u;
				if(lane == 0)
				{
					if(ml) bl = atomicAdd(&s_cursor[0], (uint32_t)__popcll(ml));
					if(mr) br = atomicAdd(&s_cursor[1], (uint32_t)__popcll(mr));
				}
				bl = (uint32_t)__shfl((int)bl, 0); br = (uint32_t)__shfl((int)br, 0);
				const unsigned long long below = (1ull << lane) - 1ull;
				if(go_l)
				{
					Ref l = r; l.hi[axis] = fminf(hi, pos);
					a.refs_out[out + bl + (uint32_t)__popcll(ml & below)] = l;
				}
				if(go_r)
				{
					Ref rr = r; rr.lo[axis] = fmaxf(lo, pos);
					a.refs_out[out + nl + br + (uint32_t)__popcll(mr & below)] = rr;
				}
			}
		}
		__syncthreads();
	}
}


// ---- breadth-first -> depth-first on the device ----
// Children are always allocated after their parent, so node indices grow with the level: one launch per level, bottom
// up, gives every node its subtree's node and leaf-reference counts; one per level, top down, gives every node its
// depth-first index (near child = next node) and the start of its leaf references; a last pass writes the final arrays.
struct FlatArgs
{
	const BfsNode *nodes; const uint32_t *leaf_refs;
	uint32_t *sub_nodes, *sub_refs, *dfs, *ref_start;
	KdNode *out_nodes; uint32_t *out_refs;
	uint32_t *big_leaves, *n_big;        // leaves too long for the in-thread sort: sorted by the host afterwards
	uint32_t cap_big;
};

__global__ void flat_sizes(const FlatArgs f, uint32_t begin, uint32_t end)
{
	const uint32_t n = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if(n >= end) return;
	const BfsNode b = f.nodes[n];
	if((b.b & 3u) == 3u) { f.sub_nodes[n] = 1u; f.sub_refs[n] = b.b >> 2; }
	else { f.sub_nodes[n] = 1u + f.sub_nodes[b.c] + f.sub_nodes[b.d]; f.sub_refs[n] = f.sub_refs[b.c] + f.sub_refs[b.d]; }
}

__global__ void flat_place(const FlatArgs f, uint32_t begin, uint32_t end)
{
	const uint32_t n = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if(n >= end) return;
	const BfsNode b = f.nodes[n];
	if((b.b & 3u) == 3u) return;
	const uint32_t me = f.dfs[n], rs = f.ref_start[n];
	f.dfs[b.c] = me + 1u; f.ref_start[b.c] = rs;
	f.dfs[b.d] = me + 1u + f.sub_nodes[b.c]; f.ref_start[b.d] = rs + f.sub_refs[b.c];
}

constexpr uint32_t kSortInThread = 48u;

__global__ void flat_emit(const FlatArgs f, uint32_t n_nodes)
{
	const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
	if(n >= n_nodes) return;
	const BfsNode b = f.nodes[n];
	const uint32_t me = f.dfs[n];
	if((b.b & 3u) != 3u) { f.out_nodes[me] = KdNode{b.a, (b.b & 3u) | (f.dfs[b.d] << 2)}; return; }
	const uint32_t cnt = b.b >> 2, first = f.ref_start[n];
	f.out_nodes[me] = KdNode{first, 3u | (cnt << 2)};
	uint32_t *dst = f.out_refs + first;
	const uint32_t *src = f.leaf_refs + b.a;
	if(cnt > kSortInThread)
	{
		for(uint32_t i = 0u; i < cnt; ++i) dst[i] = src[i];
		const uint32_t slot = atomicAdd(f.n_big, 1u);
		if(slot < f.cap_big) f.big_leaves[slot] = me;
		return;
	}
	for(uint32_t i = 0u; i < cnt; ++i)
	{	// insertion sort by triangle index: the leaf's order no longer depends on how the atomics resolved
		const uint32_t v = src[i];
		uint32_t j = i;
		while(j > 0u && dst[j - 1u] > v) { dst[j] = dst[j - 1u]; --j; }
		dst[j] = v;
	}
}

struct DevBuf
{
	void *p = nullptr;
	~DevBuf() { if(p) (void)hipFree(p); }
	bool alloc(size_t bytes) { return hipMalloc(&p, bytes) == hipSuccess; }
};

} // namespace

// returns 0 on success; a negative code and *err on failure (no fallback: the caller decides)
int build_kdtree_device(const float *verts, int n_tris, int depth_cap, KdTree &out, std::string *err, int room)
{
	// the library's code object is loaded by its first launch (~0.1 s once per process): not build time
	hipLaunchKernelGGL(init_refs, dim3(1), dim3(64), 0, nullptr, (const float *)nullptr, 0, (Ref *)nullptr);
	(void)hipDeviceSynchronize();
	const auto t0 = std::chrono::steady_clock::now();
	out.nodes.clear(); out.refs.clear(); out.max_depth = 0;
	for(int k = 0; k < 3; ++k) { out.bound_lo[k] = 0.f; out.bound_hi[k] = 0.f; }
	if(n_tris <= 0) { out.build_seconds = 0; return 0; }
	auto fail = [&](const char *m) { if(err) *err = m; return -1; };
	// tree bound, grown by 0.1 % per side like the reference's (kdtree_triangle.cc:110-116)
	float lo[3] = {verts[0], verts[1], verts[2]}, hi[3] = {verts[0], verts[1], verts[2]};
	for(size_t i = 0; i < (size_t)n_tris * 3; ++i)
		for(int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], verts[3 * i + k]); hi[k] = std::max(hi[k], verts[3 * i + k]); }
	for(int k = 0; k < 3; ++k)
	{
		const double grow = (double)(hi[k] - lo[k]) * 0.001;
		lo[k] = (float)((double)lo[k] - grow); hi[k] = (float)((double)hi[k] + grow);
	}
	const int md = std::min((int)(7.0f + 1.66f * std::log((float)n_tris)), depth_cap);
	float cost_ratio = 0.8f;
	const double log_leaves = 1.442695f * std::log((double)n_tris);
	if(log_leaves > 16.0) cost_ratio += (float)(0.25 * (log_leaves - 16.0));
	if(const char *e = std::getenv("YAFGPU_COST_RATIO")) cost_ratio = (float)std::atof(e);      // experiments: node-step cost / triangle-test cost
	float empty_bonus = 0.33f;
	if(const char *e = std::getenv("YAFGPU_EMPTY_BONUS")) empty_bonus = (float)std::atof(e);

	const size_t n = (size_t)n_tris;
	// room: references / nodes the arrays hold per triangle (x8).  Overlapping geometry (long needles, stacked sheets)
	// multiplies references; the caller retries with more room when a build reports -2
	const size_t r8 = (size_t)std::max(room, 1) * 8;
	const uint32_t cap_refs = (uint32_t)std::min<size_t>(r8 * n + 16384, 0x7fffffffu), cap_work = (uint32_t)std::min<size_t>(r8 / 2 * n + 1024, 0x7fffffffu),
	               cap_nodes = (uint32_t)std::min<size_t>(r8 * n + 1024, 0x7fffffffu), cap_leaf = (uint32_t)std::min<size_t>(r8 * n + 4096, 0x7fffffffu);
	DevBuf d_verts, d_refs[2], d_work[2], d_nodes, d_leaf, d_cnt;
	if(!d_verts.alloc(n * 9 * sizeof(float)) || !d_refs[0].alloc((size_t)cap_refs * sizeof(Ref)) || !d_refs[1].alloc((size_t)cap_refs * sizeof(Ref)) ||
	   !d_work[0].alloc((size_t)cap_work * sizeof(Work)) || !d_work[1].alloc((size_t)cap_work * sizeof(Work)) ||
	   !d_nodes.alloc((size_t)cap_nodes * sizeof(BfsNode)) || !d_leaf.alloc((size_t)cap_leaf * sizeof(uint32_t)) || !d_cnt.alloc(8 * sizeof(uint32_t)))
	{	// -3: no room on the device for the builder's arrays (a bigger retry, or a card shared with a large scene): the caller
		// takes the host builder, which produces the same format
		if(err) *err = "device kd build: out of device memory";
		(void)hipGetLastError();
		return -3;
	}
	if(hipMemcpy(d_verts.p, verts, n * 9 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail("device kd build: vertex upload failed");
	hipLaunchKernelGGL(init_refs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const float *)d_verts.p, n_tris, (Ref *)d_refs[0].p);
	Work root{};
	root.begin = 0u; root.count = (uint32_t)n_tris; root.node = 0u; root.depth_bad = 0u;
	for(int k = 0; k < 3; ++k) { root.lo[k] = lo[k]; root.hi[k] = hi[k]; }
	if(hipMemcpy(d_work[0].p, &root, sizeof root, hipMemcpyHostToDevice) != hipSuccess) return fail("device kd build: upload failed");
	uint32_t counters[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
	if(hipMemcpy(d_cnt.p, counters, sizeof counters, hipMemcpyHostToDevice) != hipSuccess) return fail("device kd build: upload failed");
	uint32_t n_work = 1u, refs_in_level = (uint32_t)n_tris;
	int deepest = 0;
	uint32_t node_count = 1u;            // node 0 is the root
	int cur = 0;
	const bool verbose = std::getenv("YAFGPU_BUILD_VERBOSE") != nullptr;
	auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
	if(verbose) { (void)hipDeviceSynchronize(); std::fprintf(stderr, "[kd device] setup %.4f s\n", since(t0)); }
	auto t_levels = std::chrono::steady_clock::now();
	std::vector<uint32_t> level_begin;       // first node index of every level (+ the end)
	level_begin.push_back(0u);
	for(int level = 0; level <= md + 1 && n_work > 0u; ++level)
	{
		BuildArgs a{};
		a.refs_in = (const Ref *)d_refs[cur].p; a.refs_out = (Ref *)d_refs[cur ^ 1].p;
		a.work_in = (const Work *)d_work[cur].p; a.work_out = (Work *)d_work[cur ^ 1].p;
		a.verts = (const float *)d_verts.p;
		a.nodes = (BfsNode *)d_nodes.p; a.leaf_refs = (uint32_t *)d_leaf.p; a.counters = (uint32_t *)d_cnt.p;
		a.n_work = n_work; a.node_base = node_count; a.cap_refs = cap_refs; a.cap_work = cap_work; a.cap_nodes = cap_nodes; a.cap_leaf_refs = cap_leaf;
		a.depth_cap = md; a.cost_ratio = cost_ratio; a.empty_bonus = empty_bonus;
		const uint32_t zero2[2] = {0u, 0u};
		if(hipMemcpy(d_cnt.p, zero2, sizeof zero2, hipMemcpyHostToDevice) != hipSuccess) return fail("device kd build: counter reset failed");
		level_begin.push_back(node_count);
		// workgroup size by the level's shape: few huge nodes -> 1024 threads each; many small ones -> one wave each
		const dim3 grid(std::min<uint32_t>(n_work, 65535u * 4u));
		const uint32_t avg = refs_in_level / n_work;
		if(n_work <= 256u && avg > 2048u) hipLaunchKernelGGL(build_level<1024>, grid, dim3(1024), 0, nullptr, a);
		else if(avg > 96u) hipLaunchKernelGGL(build_level<256>, grid, dim3(256), 0, nullptr, a);
		else hipLaunchKernelGGL(build_level<64>, grid, dim3(64), 0, nullptr, a);
		if(hipGetLastError() != hipSuccess) return fail("device kd build: launch failed");
		if(hipMemcpy(counters, d_cnt.p, sizeof counters, hipMemcpyDeviceToHost) != hipSuccess) return fail("device kd build: kernel failed");
		if(counters[4]) { if(err) *err = "device kd build: reference / node arrays overflowed"; return -2; }
		if(verbose) std::fprintf(stderr, "[kd device] level %d: %u nodes, %u refs out, %.4f s\n", level, n_work, counters[0], since(t_levels));
		n_work = counters[1]; refs_in_level = counters[0];
		node_count += n_work;
		deepest = level;
		cur ^= 1;
	}
	auto t_flat = std::chrono::steady_clock::now();
	if(n_work > 0u) return fail("device kd build: depth cap exceeded");
	// flatten breadth-first -> depth-first; the reference arrays of the level loop are free now and hold the scratch
	const uint32_t n_nodes = node_count, n_leaf_refs = counters[3];
	level_begin.push_back(n_nodes);
	constexpr uint32_t kCapBig = 4096u;
	FlatArgs f{};
	f.nodes = (const BfsNode *)d_nodes.p; f.leaf_refs = (const uint32_t *)d_leaf.p;
	uint32_t *scratch = (uint32_t *)d_refs[0].p;          // 4 * cap_nodes + 1 + kCapBig words <= 8 * cap_refs
	f.sub_nodes = scratch; f.sub_refs = scratch + cap_nodes; f.dfs = scratch + 2 * (size_t)cap_nodes; f.ref_start = scratch + 3 * (size_t)cap_nodes;
	f.n_big = scratch + 4 * (size_t)cap_nodes; f.big_leaves = f.n_big + 1; f.cap_big = kCapBig;
	f.out_nodes = (KdNode *)d_refs[1].p; f.out_refs = (uint32_t *)((char *)d_refs[1].p + (size_t)cap_nodes * sizeof(KdNode));
	static_assert(sizeof(KdNode) == 8, "KdNode layout");
	if((4 * (size_t)cap_nodes + 1 + kCapBig) * 4 > (size_t)cap_refs * sizeof(Ref) || (size_t)cap_nodes * 8 + (size_t)cap_leaf * 4 > (size_t)cap_refs * sizeof(Ref))
		return fail("device kd build: scratch sizing");
	const uint32_t zero = 0u;
	if(hipMemcpy(f.dfs, &zero, 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(f.ref_start, &zero, 4, hipMemcpyHostToDevice) != hipSuccess ||
	   hipMemcpy(f.n_big, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
		return fail("device kd build: upload failed");
	const size_t n_levels = level_begin.size() - 1;
	for(size_t l = n_levels; l-- > 0;)
	{
		const uint32_t b = level_begin[l], e = level_begin[l + 1];
		if(e > b) hipLaunchKernelGGL(flat_sizes, dim3((e - b + 255u) / 256u), dim3(256), 0, nullptr, f, b, e);
	}
	for(size_t l = 0; l < n_levels; ++l)
	{
		const uint32_t b = level_begin[l], e = level_begin[l + 1];
		if(e > b) hipLaunchKernelGGL(flat_place, dim3((e - b + 255u) / 256u), dim3(256), 0, nullptr, f, b, e);
	}
	hipLaunchKernelGGL(flat_emit, dim3((n_nodes + 255u) / 256u), dim3(256), 0, nullptr, f, n_nodes);
	if(hipGetLastError() != hipSuccess) return fail("device kd build: launch failed");
	out.nodes.resize(n_nodes); out.refs.resize(n_leaf_refs);
	uint32_t n_big = 0u;
	if(hipMemcpy(out.nodes.data(), f.out_nodes, (size_t)n_nodes * sizeof(KdNode), hipMemcpyDeviceToHost) != hipSuccess ||
	   (n_leaf_refs && hipMemcpy(out.refs.data(), f.out_refs, (size_t)n_leaf_refs * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) ||
	   hipMemcpy(&n_big, f.n_big, 4, hipMemcpyDeviceToHost) != hipSuccess)
		return fail("device kd build: download failed");
	if(n_big > kCapBig)
	{	// more long leaves than the list holds: find them all on the host instead
		for(const KdNode &nd : out.nodes)
			if((nd.b & 3u) == 3u && (nd.b >> 2) > kSortInThread) std::sort(out.refs.begin() + nd.a, out.refs.begin() + nd.a + (nd.b >> 2));
	}
	else if(n_big)
	{
		std::vector<uint32_t> big(n_big);
		if(hipMemcpy(big.data(), f.big_leaves, (size_t)n_big * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("device kd build: download failed");
		for(uint32_t me : big) { const KdNode &nd = out.nodes[me]; std::sort(out.refs.begin() + nd.a, out.refs.begin() + nd.a + (nd.b >> 2)); }
	}
	if(verbose) std::fprintf(stderr, "[kd device] flatten + download %.4f s (%u long leaves)\n", since(t_flat), n_big);
	out.max_depth = deepest;
	for(int k = 0; k < 3; ++k) { out.bound_lo[k] = lo[k]; out.bound_hi[k] = hi[k]; }
	out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return 0;
}

int build_kdtree_device_retry(const float *verts, int n_tris, int depth_cap, KdTree &out, std::string *err)
{
	int rc = -2;
	for(int room = 1; room <= 16 && rc == -2; room *= 4) rc = build_kdtree_device(verts, n_tris, depth_cap, out, err, room);
	return rc;
}

} // namespace yafgpu
