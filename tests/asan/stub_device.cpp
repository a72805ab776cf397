// TEST SUPPORT: stands in for the device unit (yafgpu_device.hip, kdtree_build_device.hip) so that the host side of the
// C ABI — parameter maps, the Interface state machine, geometry assembly, smoothMesh, the XML loader, the host kd
// builder — can run under AddressSanitizer / UBSan on a machine without a GPU.  Every device entry point reports
// "no device"; nothing here is part of the product.
#include "../../include/yafgpu.h"
#include "../../libyafaray_amd/csrc/kdtree_build.h"

#include <string>

static thread_local std::string g_err = "sanitizer build: no device";

namespace yafgpu {
int build_kdtree_device_retry(const float *, int, int, KdTree &, std::string *err) { if(err) *err = "sanitizer build: no device"; return -1; }
}

extern "C" {
const char *yafgpu_last_error(void) { return g_err.c_str(); }
void yafgpu_internal_set_error(const char *m) { g_err = m ? m : ""; }
int yafgpu_scene_create(const yafgpu_scene_desc *, yafgpu_scene_t **out) { if(out) *out = nullptr; return -100; }
void yafgpu_scene_destroy(yafgpu_scene_t *) {}
int yafgpu_scene_info(const yafgpu_scene_t *, yafgpu_tree_info *) { return -100; }
int yafgpu_render_tiles(yafgpu_scene_t *, const yafgpu_render_params *, float *, yafgpu_counters *, void *) { return -100; }
int yafgpu_render_passes_to_host(yafgpu_scene_t *, const yafgpu_render_params *, const yafgpu_aa_schedule *, float *, yafgpu_counters *, int32_t *) { return -100; }
int yafgpu_trace_closest(yafgpu_scene_t *, int32_t, const float *, int32_t *, float *, float *) { return -100; }
int yafgpu_trace_shadow(yafgpu_scene_t *, int32_t, const float *, int32_t *) { return -100; }
int yafgpu_set_profiling(yafgpu_scene_t *, int32_t) { return -100; }
int yafgpu_scene_set_pass_pipelining(yafgpu_scene_t *, int32_t) { return -100; }
int yafgpu_scene_set_abort_flag(yafgpu_scene_t *, const volatile int32_t *) { return -100; }
int yafgpu_scene_set_exchange(yafgpu_scene_t *, yafgpu_exchange_fn, void *) { return -100; }
void yafgpu_glibc_rand(uint32_t, int32_t count, int32_t *out) { for(int32_t k = 0; k < count; ++k) out[k] = 0; }
int yafgpu_get_profile(const yafgpu_scene_t *, double *, uint64_t *) { return -100; }
int yafgpu_device_count(void) { return 0; }
int yafgpu_set_device(int) { return -100; }
uint64_t yafgpu_planes_bytes(int32_t w, int32_t h) { return (uint64_t)4 * (uint64_t)w * (uint64_t)h * 5u * sizeof(float); }
int yafgpu_film_combine(const float *, float *, int32_t, int32_t, void *) { return -100; }
int yafgpu_render_to_host(yafgpu_scene_t *, const yafgpu_render_params *, float *, yafgpu_counters *) { return -100; }
int yafgpu_scene_get_tree(const yafgpu_scene_t *, uint32_t *, uint32_t *, float *) { return -100; }
int yafgpu_probe(yafgpu_scene_t *, int32_t, int32_t, const float *, int32_t, float *, int32_t) { return -100; }
}
