// A scene-specialised build of the shading kernel (wf_shade).
//
// The shading kernel is one straight-line program over every material type and feature of the path; its register
// and scratch budget is set by the union of them all (168 VGPRs + 116 B of scratch at 3 waves per SIMD).  Scenes
// that use a subset run a kernel compiled for that subset: this file is compiled once per variant with
//
//   -DYAFGPU_VARIANT_NAME=<name>  -DYAFGPU_MAT_MASK=<bit per YAFGPU_MAT_* handled>  -DYAFGPU_FEAT_RECURSE=<0|1>  [-DYAFGPU_FEAT_LIGHTS=0]  [-DYAFGPU_FEAT_MULTI=0]
//
// (FEAT_LIGHTS=0: the program of a serial-state replay's RECORD pass — no light estimate, the vertex of a resume in registers)
//
// and includes the main unit with everything but wf_shade and what it calls compiled out.  All of its symbols live in
// their own namespace (the macro below renames `yafgpu`), so the variants and the main unit link into one library; the
// main unit reaches a variant through the three C functions at the bottom (yafgpu_device.hip: shade_variants).
#ifndef YAFGPU_VARIANT_NAME
#error "compile with -DYAFGPU_VARIANT_NAME=..., -DYAFGPU_MAT_MASK=..., -DYAFGPU_FEAT_RECURSE=..."
#endif
#define YAFGPU_VARIANT_TU 1
#define YG_CAT2(a, b) a##b
#define YG_CAT(a, b) YG_CAT2(a, b)
#define yafgpu YG_CAT(yafgpu_shade_, YAFGPU_VARIANT_NAME)
#include "yafgpu_device.hip"

namespace vns = yafgpu;
#undef yafgpu

extern "C" {

// material types / features this variant was compiled for
void YG_CAT(YG_CAT(yafgpu_shade_, YAFGPU_VARIANT_NAME), _describe)(uint32_t *mat_mask, int *recurse, int *lights, int *multi)
{
	*mat_mask = (uint32_t)(YAFGPU_MAT_MASK); *recurse = YAFGPU_FEAT_RECURSE; *lights = YAFGPU_FEAT_LIGHTS; *multi = YAFGPU_FEAT_LIGHTS && YAFGPU_FEAT_MULTI;
}
const void *YG_CAT(YG_CAT(yafgpu_shade_, YAFGPU_VARIANT_NAME), _kernel)() { return (const void *)vns::wf_shade; }
// args: the main unit's WfArgs (same definition, so the same layout)
int YG_CAT(YG_CAT(yafgpu_shade_, YAFGPU_VARIANT_NAME), _launch)(const void *args, size_t bytes, int grid, hipStream_t stream)
{
	vns::WfArgs a;
	if(bytes != sizeof a) return -1;
	std::memcpy(&a, args, sizeof a);
	hipLaunchKernelGGL(vns::wf_shade, dim3((unsigned)grid), dim3(vns::kBlock), 0, stream, a);
	return 0;       // the caller checks hipGetLastError like after its own launches
}

} // extern "C"
