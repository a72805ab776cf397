// Host-side SAH kd-tree builder producing the flattened, index-based tree the HIP traversal
// kernels walk.  This is NOT a mirror of the reference's builder (TriKdTree::buildTree,
// src/common/kdtree_triangle.cc:468-675, which depends on double-precision triangle clipping);
// only the results of closest-hit / any-hit queries are contractual.  It keeps
// the reference's build *parameters* (scene.cc:818, kdtree_triangle.cc:89-100).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace yafgpu {

// 8-byte node, interior and leaf alike.
//   interior: a = float bits of the split position, b = axis | (right_child_index << 2); the near
//             (left/below) child is always the next node in memory (depth-first layout)
//   leaf    : a = index of the first entry in the leaf-reference array, b = 3 | (prim_count << 2)
struct KdNode { uint32_t a, b; };

struct KdTree
{
	std::vector<KdNode> nodes;
	std::vector<uint32_t> refs;
	float bound_lo[3], bound_hi[3];
	int max_depth = 0;      // deepest leaf actually produced
	double build_seconds = 0;
};

// verts: n_tris*9 floats (a,b,c).  depth_cap bounds the tree depth (the traversal stack never
// needs more entries than the depth).  threads<=0: hardware concurrency.
void build_kdtree(const float *verts, int n_tris, int depth_cap, int threads, KdTree &out);

// The same tree built on the GPU (kdtree_build_device.hip): breadth-first binned / exact-candidate SAH with the same
// parameters and cost model.  Returns 0, or a negative code with *err set (never falls back to the host builder).
// returns 0, -1 (device error) or -2 (the arrays sized for `room` x 8 references per triangle overflowed)
int build_kdtree_device(const float *verts, int n_tris, int depth_cap, KdTree &out, std::string *err, int room = 1);
// the same with more room on -2 (1, 4, 16), for callers that want a device-built tree or an error
int build_kdtree_device_retry(const float *verts, int n_tris, int depth_cap, KdTree &out, std::string *err);

} // namespace yafgpu
