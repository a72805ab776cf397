// Wavefront form of the path-tracing pass (included by yafgpu_device.hip after the scene/traversal code).
//
// The per-sample program of PathIntegrator::integrate (integrator_path_tracer.cc:112-347) is run as a
// coroutine: a path executes until it needs a kd-tree query, parks its state in HBM, and is resumed
// after a *trace* kernel has answered the query.  Three kinds of kernels alternate:
//
//   wf_trace<closest|any>  — nothing but traversal: reads 32 B of ray per path from a compacted
//                            queue, walks the tree (kd_trace), writes 16 B of hit / 4 B of verdict.
//                            ~60 VGPRs, so it runs at full occupancy, and every lane has a live ray.
//   wf_shade               — resumes every path that was just answered, runs material / light /
//                            sampling code until the next query, and appends the path to the
//                            closest-hit or the any-hit queue (wave-aggregated atomics).
//   wf_accumulate          — when all paths of a chunk have ended: one thread per pixel adds the
//                            per-sample results in sample order (ImageFilm::addSample's order).
//
// The arithmetic is the same, operation for operation, as in the one-kernel path (integrate /
// direct_light in yafgpu_device.hip); results are bit-identical between the two (tests/test_gpu_parity.py).
#pragma once

namespace yafgpu {

constexpr int kWfRecs = 32;   // float4 records of parked state per path (512 B; 28..31 hold a SECOND MIS pair parked with the first (WfArgs::multi); 22 and 23 are only touched in textured scenes, 24 and 25 only with bump mapping, 26 and 27 by a path that parks its next segment beside a vertex's last shadow pair)

struct WfArgs
{
	RenderArgs ra;
	float4 *state; uint32_t cap;      // record k of path s lives at state[k * cap + s]
	float4 *results;                  // final rgba per path
	uint32_t n_paths, pixel_begin, n_pixels;
	const uint32_t *pix_prefix;       // n_tiles+1 prefix sums of pixels per tile of this shard
	int frames;                       // levels of recursiveRaytrace frames behind the kWfRecs records (0: none allocated)
	int frame_recs, has_glossy;       // records per frame: 5, or 12 when a material has a glossy lobe recursiveRaytrace samples (its loop state)
	int pix_listed;                   // the chunk's pixels are given by pix_xy (a resample mask picked them), not by the tile list
	uint32_t *pix_xy;                 // px | py << 16 per pixel of the chunk: written by wf_generate, so that resuming a path
	                                  // costs one load instead of a binary search over the tile prefix (9 dependent loads)
	// closest queue: one entry per path (the path's ray).  shadow queue: one entry per shadow RAY,
	// slot | which<<31 — a path parks with up to two (the light-sampling and the BSDF-sampling ray of one
	// MIS pair).  resume queue: the paths (once each) that wait for shadow answers.
	const uint32_t *q_closest_in, *q_shadow_in, *q_resume_in;   // nullptr closest queue = identity (first iteration)
	uint32_t *q_closest_out, *q_shadow_out, *q_resume_out;
	uint32_t *verdict;                // any-hit answers: BIT 4*slot + 2*pair + which, set for an occluded ray (zeroed before every any-hit launch)
	float4 *shadow_filt;              // [2*slot + which] product of the transparencies a shadow ray passed (transparent shadows), or nullptr
	// Serial-state replay (DESIGN.md "serial state").  The reference keeps two pieces of state that run
	// through the samples of a render in order: the per-tile MWC stream Russian roulette draws from (integrator_tiled.cc:319,
	// integrator_path_tracer.cc:282-288) and the per-thread counter estimateOneDirectLight picks its light with
	// (integrator_montecarlo.cc:62-76).  replay 1 = RECORD pass: paths run without light estimates and without roulette
	// kills and note, per path sample, at which depths estimateOneDirectLight is called and the survival probability of
	// every roulette test; wf_replay_scan then walks each tile's samples in the reference's order with the tile's stream
	// and leaves the depth each path is killed at and the counter value each sample starts with; replay 2 = FINAL pass:
	// the program proper, taking both from those tables.  replay 0: per-sample streams (no serial state).
	int replay, replay_lights;        // replay_lights: the counter is replayed too (one GPU; else the per-sample ordinal)
	// With recursiveRaytrace a camera sample is a TREE of integrate() calls, which the reference (and the frames of this path program)
	// walks depth first: a sample's events are kept per call, in that order.  ev_m = entries of calls per camera sample (1 without
	// recursion; else the most a sample can make); the call's ordinal rides in the top byte of record 19's z (the sample's light
	// calls so far in the low 24 bits), counted up by st_after_closest whenever a level below the camera's starts.
	int ev_m;
	// The record pass has already answered every closest-hit query the final pass will ask (the same paths, cut short by the roulette
	// kills): it keeps the answers — one per (call, path sample, segment), hit_k per camera sample — and the final pass's closest-hit
	// launches are dropped: wf_shade looks the answers up (wf_hit_key).  nullptr: not kept (too many per sample, or a stats pass), the rays are traced again.
	float4 *hit_cache; int hit_k;
	uint32_t *ev_flags;               // [(path * ev_m + call) * P + path_sample]: bit d = light call at depth d, bit 16 + d = roulette test at depth d
	float *ev_p;                      // [(path * P + path_sample) * (bounces - 1) + d - 1]: probability of the test at depth d
	uint8_t *ev_kill;                 // [path * P + path_sample]: depth of the test that kills it (255: none)
	uint8_t *ev_calls;                // [path * P + path_sample]: light calls it makes (up to the kill)
	uint32_t *lc_base;                // [path]: correlative_sample_number_ when the sample starts
	// Next segment beside the last shadow pair (DESIGN.md "fewer phases per vertex"): a vertex's continuation ray does not depend on its
	// shadow answers (the roulette test reads the throughput, the sampler the vertex), so a path that is about to park for the LAST shadow
	// pair of a vertex's light estimate samples the next segment right away and parks for both: one traversal phase, one state round trip
	// and one launch tail less per vertex.  Only where nothing in between can change course: no recursion frames, and a roulette test at
	// this vertex only when the serial-state replay has its outcome on the table.  0: off (comparison runs, YAFGPU_SPECULATE=0).
	int speculate;
	// Two MIS pairs per park (DESIGN.md "pairs per park"): a vertex whose light estimate has another pair to go after the one it is about to
	// park for — another sample of the light, or the next light — evaluates that one too and parks for both (records 28..31, two more verdict
	// bits, two more bits of the shadow-queue entry): the shadow answers steer nothing, so the second pair is what the resumed path would have
	// evaluated next, and a light with n samples costs n / 2 state round trips per vertex.  Not with transparent shadows or recursion frames.
	int multi;
	uint32_t *cnt_in;                 // [0] closest count, [1] shadow-ray count, [2] closest fetch cursor, [3] shadow fetch cursor, [4] resume count
	uint32_t *cnt_out;                // same layout, filled by wf_shade for the next iteration
};

enum : int { kPcAfterClosest = 1, kPcAfterShadow = 2, kPcAfterBoth = 3 };      // Both: the shadow pair of a vertex AND the next segment's closest hit are out
enum : int { kReqDone = 0, kReqClosest = 1, kReqShadow = 2, kReqBoth = 3 };

YG_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
YG_DEV float4 f4(Col c, float w) { return make_float4(c.r, c.g, c.b, w); }
YG_DEV V3 v3(float4 f) { return mk(f.x, f.y, f.z); }
YG_DEV Col c3(float4 f) { return mkc(f.x, f.y, f.z); }
YG_DEV float fbits(uint32_t u) { return __uint_as_float(u); }
YG_DEV uint32_t ubits(float f) { return __float_as_uint(f); }

// pixel of path slot s: chunk-local pixel -> tile (binary search over the per-tile pixel prefix) -> (px, py)
YG_DEV void wf_pixel_of(const WfArgs &a, uint32_t pixel_local, int &px, int &py)
{
	const uint32_t g = a.pixel_begin + pixel_local;
	int lo = 0, hi = a.ra.n_tiles;
	while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(a.pix_prefix[mid] <= g) lo = mid; else hi = mid; }
	const int4 rect = a.ra.tile_rect[lo];
	const int q = (int)(g - a.pix_prefix[lo]);
	px = rect.x + q % rect.z; py = rect.y + q / rect.z;
}

YG_DEV void make_sp(V3 p, V3 n, V3 ng, int mat, SurfPt &sp) { sp.p = p; sp.n = n; sp.ng = ng; sp.mat = mat; create_cs(n, sp.nu, sp.nv); }

// The (s_1, s_2) of light sample `is`: doLightEstimation restarts Halton(2)/Halton(3) at offs-1 for both
// halves of the MIS pair (integrator_montecarlo.cc:164-165,285-286), so the pair shares them.
YG_DEV int dl_area_samples(const RenderArgs &ra, const yafgpu_light &light, int division)      // integrator_montecarlo.cc:153-154
{
	int n = (int)ceilf((float)light.samples * ra.rp.aa_light_sample_multiplier);
	if(division > 1) n = max(1, n / division);
	return n;
}
YG_DEV void dl_samples(const RenderArgs &ra, const yafgpu_light &light, int li, int is, int division, uint32_t pixel_sample, uint32_t sampling_offs, float &s_1, float &s_2)
{
	const int n = dl_area_samples(ra, light, division);
	const uint32_t offs = (uint32_t)n * pixel_sample + sampling_offs + (uint32_t)li * 4567u;
	Halton hal_2, hal_3;
	hal_2.init(2u); hal_3.init(3u);
	hal_2.set_start(offs - 1u); hal_3.set_start(offs - 1u);
	s_1 = 0.f; s_2 = 0.f;
	for(int k = 0; k <= is; ++k) { s_1 = hal_2.next(); s_2 = hal_3.next(); }   // the incremental sequence, replayed
}

// One candidate of MonteCarloIntegrator::doLightEstimation (integrator_montecarlo.cc:78-345): light
// `li`, half `phase` of the MIS pair (0 light sampling :161-262, 1 BSDF sampling :285-333; Dirac lights
// have a single half :94-148).  Returns whether a shadow ray is wanted and, if so, the ray and the
// radiance it would carry if unoccluded — identical arithmetic to direct_light().
YG_DEV bool dl_candidate(const RenderArgs &ra, const yafgpu_light &light, int phase, float s_1, float s_2, const SurfPt &sp, const yafgpu_material &mat,
                         const BsdfDat &dat, V3 wo, V3 &r_dir, float &r_tmin, float &r_tmax, Col &contrib)
{
	const uint32_t kMisFlags = kGlossy | kDiffuse | kDispersive | kReflect | kTransmit;
	r_dir = mk(0.f, 0.f, 0.f); r_tmin = 0.f; r_tmax = -1.f;
	contrib = mkc(0.f, 0.f, 0.f);
	if(light.type == YAFGPU_LIGHT_POINT)
	{
		Col lcol;
		if(!pointlight_illuminate(light, sp.p, lcol, r_dir, r_tmax)) return false;
		r_tmin = ra.rp.shadow_bias_auto ? ra.shadow_bias * smax(1.f, length(sp.p)) : ra.shadow_bias;
		const float angle = mat.flat ? 1.f : fabsf(dot(sp.n, r_dir));
		contrib = (mat_eval(mat, dat, sp, wo, r_dir, kAll) * lcol) * angle;
		return true;
	}
	if(phase == 0)
	{
		float ls_pdf;
		if(!arealight_illum_sample(light, sp.p, s_1, s_2, r_dir, r_tmax, ls_pdf)) return false;
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
		return true;
	}
	r_tmin = ra.rp.min_raydist_auto ? ra.ray_min_dist * smax(1.f, length(sp.p)) : ra.ray_min_dist;
	float W = 0.f;
	BsdfSample bs; bs.s_1 = s_1; bs.s_2 = s_2; bs.pdf = 0.f; bs.flags = kMisFlags; bs.sampled = kNone;
	const Col surf_col = mat_sample(mat, dat, sp, wo, r_dir, bs, W);
	float light_ipdf;
	if(!(bs.pdf > 1e-6f && arealight_intersect(light, sp.p, r_dir, r_tmax, light_ipdf))) return false;
	if(light_ipdf > 1e-6f)
	{
		const float l_pdf = 1.f / light_ipdf;
		const float l_2 = l_pdf * l_pdf, m_2 = bs.pdf * bs.pdf;
		const float w = m_2 / (l_2 + m_2);
		contrib = ((surf_col * col3(light.color)) * w) * W;
	}
	return true;
}

// ---- the per-path program, cut into steps that talk to each other through the parked records ----
//
// Record map (float4 each, record k of path s at state[k*cap + s]):
//   r0  ray origin | tmin            r1  ray direction | tmax          r2  closest-hit answer (tri, t, u, v) — indexed by QUEUE POSITION, not by path
//       (the traversal kernel writes entry i of its queue, wf_shade walks the same queue: both sides stream)
//   r3  sp0.p | mat0     r4 sp0.n    r5  sp0.ng | bsdfs0     r6  wo0                      (camera hit)
//   r7  hit.p | mat      r8 hit.n    r9  hit.ng              r10 pwo                      (current path vertex)
//   r11 throughput | rr.x            r12 path_col | rr.c     r13 col | pc,stage,dl_on_sp0,depth,path_i
//   r14 pending A | li,l_end,mask,is r15 ccol   r16 ccol_2   r17 col_dirac   r18 total   (light estimate in flight)
//   r19 offs, sampled_flags, one_light_calls, alpha
//   r20 second shadow ray: direction | tmin      r21 pending B | tmax of the second ray
//   r22 camera hit: triangle, barycentrics       r23 current vertex: triangle, barycentrics   (textured scenes: the nodes read
//       texture coordinates, Triangle::getSurface triangle.cc:46-79, which are interpolated again when a step needs them)
// Each step loads only what it uses and stores what it produced, so that no step keeps the whole path
// in registers: the live set of wf_shade is that of its widest step, not of the whole integrator.
// Addressing: the record base state + k*cap is wave-uniform (scalar registers) and the path's byte offset slot*16
// fits 32 bits (cap <= 2^27), so an access is one `global_load/store ... v_off, s[base]` with no per-lane 64-bit
// pointer arithmetic — and no per-record pointers for the optimizer to keep alive across the whole step loop.
YG_DEV float4 &wf_rec(const WfArgs &a, int k, uint32_t slot)
{
	return *(float4 *)((char *)(a.state + (size_t)k * a.cap) + (slot << 4));
}
#define REC(k) wf_rec(a, (k), slot)
// index of (camera sample, integrate() call, path sample) in the event tables of the serial-state replay; z19 = record 19's z
YG_DEV uint32_t wf_event(const WfArgs &a, uint32_t slot, uint32_t z19, int path_i)
{
	const uint32_t call = a.ev_m > 1 ? min(z19 >> 24, (uint32_t)a.ev_m - 1u) : 0u;
	return (slot * (uint32_t)a.ev_m + call) * (uint32_t)max(a.ra.rp.path_samples, 1) + (uint32_t)path_i;
}

// The material at a path vertex: the record itself or, for a material with shader nodes, its resolved copy in `tmp`
// (yafgpu_texture.h mat_resolve).  wf_mat_hit: triangle and barycentrics at hand; wf_mat_parked: from record 22 / 23.
YG_DEV const yafgpu_material &wf_mat_hit(const DevScene &sc, const SurfPt &sp, int tri, float bu, float bv, yafgpu_material &tmp)
{
	const yafgpu_material &m = sc.mats[sp.mat];
#if YAFGPU_FEAT_TEXTURE
	if(m.n_nodes > 0 && sc.tex.nodes != nullptr)
	{
		TexPoint tp; tex_point(sc.tex, tri, bu, bv, sp.p, sp.n, sp.ng, tp);
		mat_resolve(sc.tex, sc.cam, m, tp, tmp);
		return tmp;
	}
#endif
	return m;
}
// Bump mapping (NodeMaterial::evalBump + Material::applyBump at the head of every initBsdf, material_shiny_diffuse.cc:171-175,
// material_glossy.cc:56, material_coated_glossy.cc:73, material_glass.cc:62): the shading frame the rest of the vertex sees.
// Record 24 / 25 parks nu of the path's two vertices with .w = 1 for a bumped frame (nv = n x nu then), 0 for create_cs's.
YG_DEV bool wf_bump_hit(const DevScene &sc, SurfPt &sp, int tri, float bu, float bv)
{
#if YAFGPU_FEAT_TEXTURE
	const yafgpu_material &m = sc.mats[sp.mat];
	if(m.n_bump <= 0 || !sc.tex.has_bump) return false;
	TexPoint tp; tex_point(sc.tex, tri, bu, bv, sp.p, sp.n, sp.ng, tp);
	const float4 r1 = sc.tri[3 * tri + 1], r2 = sc.tri[3 * tri + 2];
	tex_point_derivatives(sc.tex, tri, mk(r1.x, r1.y, r1.z), mk(r2.x, r2.y, r2.z), sp.n, sp.nu, sp.nv, tp);
	NodeResult stack[kMaxNodes];
	nodes_eval_derivative(sc.tex, sc.tex.nodes + m.bump_first, min(m.n_bump, kMaxNodes), sc.cam, tp, stack);
	apply_bump(sp.n, sp.nu, sp.nv, stack[m.sh_bump].col.r, stack[m.sh_bump].col.g);
	return true;
#else
	(void)sc; (void)sp; (void)tri; (void)bu; (void)bv;
	return false;
#endif
}
// the parked frame of vertex 0 / 1 back onto a surface point rebuilt from its records
YG_DEV void wf_frame_set(SurfPt &sp, const float4 nu)
{
	if(nu.w != 0.f) { sp.nu = v3(nu); sp.nv = normalize(cross(sp.n, sp.nu)); }
}
YG_DEV void wf_frame_parked(const WfArgs &a, uint32_t slot, int vertex, SurfPt &sp)
{
#if YAFGPU_FEAT_TEXTURE
	if(a.ra.sc.tex.has_bump) wf_frame_set(sp, REC(24 + vertex));
#else
	(void)a; (void)slot; (void)vertex; (void)sp;
#endif
}
YG_DEV const yafgpu_material &wf_mat_parked(const WfArgs &a, uint32_t slot, int vertex, const SurfPt &sp, yafgpu_material &tmp)
{
	const yafgpu_material &m = a.ra.sc.mats[sp.mat];
#if YAFGPU_FEAT_TEXTURE
	if(m.n_nodes > 0 && a.ra.sc.tex.nodes != nullptr)
	{
		const float4 r = REC(22 + vertex);
		return wf_mat_hit(a.ra.sc, sp, (int)ubits(r.x), r.y, r.z, tmp);
	}
#endif
	return m;
}

struct Ctl { Col col; int pc, stage, dl_on_sp0, depth, path_i, level, incl, add, mask2; };   // mask2: the second pair parked with the first (WfArgs::multi): which of its rays are out   // level: raylevel of recursiveRaytrace; incl: RenderState::include_lights_; add: integrate()'s additional_depth
YG_DEV uint32_t pack_ctl(const Ctl &c) { return (uint32_t)c.pc | ((uint32_t)c.stage << 2) | ((uint32_t)c.dl_on_sp0 << 4) | ((uint32_t)c.level << 5) | ((uint32_t)c.depth << 8) | ((uint32_t)c.add << 12) | ((uint32_t)c.incl << 16) | (((uint32_t)c.path_i & 0x1fffu) << 17) | ((uint32_t)c.mask2 << 30); }
YG_DEV Ctl load_ctl(const WfArgs &a, uint32_t slot)
{
	const float4 r = REC(13);
	const uint32_t w = ubits(r.w);
	Ctl c; c.col = c3(r);
	c.pc = (int)(w & 3u); c.stage = (int)((w >> 2) & 3u); c.dl_on_sp0 = (int)((w >> 4) & 1u); c.level = (int)((w >> 5) & 7u);
	c.depth = (int)((w >> 8) & 0xfu); c.add = (int)((w >> 12) & 0xfu); c.incl = (int)((w >> 16) & 1u); c.path_i = (int)((w >> 17) & 0x1fffu); c.mask2 = (int)(w >> 30);      // depth < bounces <= 12; path samples < 8192
	return c;
}
YG_DEV uint32_t pack_dlc(int li, int l_end, int mask, int is) { return (uint32_t)li | ((uint32_t)l_end << 8) | ((uint32_t)mask << 16) | ((uint32_t)is << 20); }

enum { W_AFTER_CLOSEST, W_AFTER_SHADOW, W_DL_NEXT, W_DL_EVAL, W_DL_DONE, W_RECURSE, W_RETURN, W_EXTEND, W_START_PATH, W_FINISH, W_PARK_CLOSEST, W_PARK_SHADOW,
       W_GLOSSY_NEXT, W_RECURSE_SPEC, W_PARK_BOTH, W_NEXT_VERTEX };

// register-resident copies of records 11, 12, 14, 18 (and, with YAFGPU_HOT_ACC, the accumulators 15..17) during
// one advance (see wf_advance); the other records of the 11..18 range go straight to memory
#ifndef YAFGPU_HOT_ACC
#define YAFGPU_HOT_ACC 1      // C2 shade: 0 -> 7.8 ms, 1 -> 6.9 ms per pass (3 waves/SIMD)
#endif
// acc_zero: the light-estimate accumulators 15..17 are all zero and their records in memory are NOT kept up to date — the state
// between two light estimates, and, with one-sample lights, at every park: the flag travels in bit 18 of the word record 14 keeps
// (kDlcAccZero), so a resumed shadow answer neither loads three records of zeros nor, having closed its light, writes them back
// (6 of the ~21 record writes and 3 of the ~22 record reads of a vertex of the benchmark scenes).
#ifndef YAFGPU_ACC_ZERO_FLAG
#define YAFGPU_ACC_ZERO_FLAG 1      // 0: the accumulators are always kept in memory (A/B: profiles/r02_ab_acczero.txt)
#endif
// tot_zero: the same for record 18, the estimate's sum over the lights closed so far (zero from the start of a vertex's estimate to
// its first closed light, dead once st_dl_done has taken it): bit 19 (kDlcTotZero).
#ifndef YAFGPU_FEAT_LIGHTS
#define YAFGPU_FEAT_LIGHTS 1      // (see below, at A_REPLAY)
#endif
struct Hot
{
	float4 r11, r12, r14, r15, r16, r17, r18; uint32_t valid, dirty, acc_zero, tot_zero;
#if !YAFGPU_FEAT_LIGHTS
	float4 vt[8]; uint32_t vvalid;      // records 3..10 of this resume's vertex, in registers (a record pass's kernel)
#endif
};
constexpr uint32_t kHotAccBits = 0x70u, kHot14 = 0x08u, kHot18 = 0x80u;
constexpr uint32_t kDlcAccZero = 1u << 18, kDlcTotZero = 1u << 19;
template<int K> constexpr bool hot_cached() { return K == 11 || K == 12 || K == 14 || K == 18 || (YAFGPU_HOT_ACC && K >= 15 && K <= 17); }
template<int K> YG_DEV float4 &hot_ref(Hot &h)
{
	static_assert(K == 11 || K == 12 || (K >= 14 && K <= 18), "not a hot record");
	if constexpr(K == 11) return h.r11; else if constexpr(K == 12) return h.r12; else if constexpr(K == 14) return h.r14;
	else if constexpr(K == 15) return h.r15; else if constexpr(K == 16) return h.r16; else if constexpr(K == 17) return h.r17;
	else return h.r18;
}
template<int K> YG_DEV float4 hot_get(const WfArgs &a, uint32_t slot, Hot &h)
{
	if constexpr(!hot_cached<K>()) return REC(K);
	else
	{
		constexpr uint32_t bit = 1u << (K - 11);
		if(!(h.valid & bit)) { hot_ref<K>(h) = REC(K); h.valid |= bit; }
		return hot_ref<K>(h);
	}
}
template<int K> YG_DEV void hot_set(const WfArgs &a, uint32_t slot, Hot &h, float4 v)
{
	if constexpr(!hot_cached<K>()) REC(K) = v;
	else
	{
		constexpr uint32_t bit = 1u << (K - 11);
		if constexpr(YAFGPU_HOT_ACC && K >= 15 && K <= 17)
		{
			if(h.acc_zero)
			{	// no longer all zero: from here on the three records are kept in memory again (the other two as the zeros they hold)
				h.acc_zero = 0u;
				if(!(h.valid & kHot14)) { h.r14 = REC(14); h.valid |= kHot14; }
				h.dirty |= kHotAccBits | kHot14;
			}
		}
		if constexpr(YAFGPU_ACC_ZERO_FLAG && K == 18)
		{
			if(h.tot_zero)
			{
				h.tot_zero = 0u;
				if(!(h.valid & kHot14)) { h.r14 = REC(14); h.valid |= kHot14; }
				h.dirty |= kHot14;
			}
		}
		hot_ref<K>(h) = v; h.valid |= bit; h.dirty |= bit;
	}
}
// record 18 := 0 (the start of a vertex's light estimate; its sum taken by st_dl_done)
YG_DEV void hot_zero_tot(const WfArgs &a, uint32_t slot, Hot &h)
{
	const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
	if(!YAFGPU_ACC_ZERO_FLAG) { hot_set<18>(a, slot, h, z4); return; }
	h.r18 = z4; h.valid |= kHot18; h.dirty &= ~kHot18;
	if(!h.tot_zero)
	{
		h.tot_zero = 1u;
		if(!(h.valid & kHot14)) { h.r14 = REC(14); h.valid |= kHot14; }
		h.dirty |= kHot14;
	}
}
// accumulators := 0 (the start of a light estimate, a light closed)
YG_DEV void hot_zero_acc(const WfArgs &a, uint32_t slot, Hot &h)
{
	const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
	if(!YAFGPU_HOT_ACC) { REC(15) = z4; REC(16) = z4; REC(17) = z4; return; }
	if(!YAFGPU_ACC_ZERO_FLAG) { hot_set<15>(a, slot, h, z4); hot_set<16>(a, slot, h, z4); hot_set<17>(a, slot, h, z4); return; }
	h.r15 = z4; h.r16 = z4; h.r17 = z4;
	h.valid |= kHotAccBits; h.dirty &= ~kHotAccBits;
	if(!h.acc_zero)
	{
		h.acc_zero = 1u;
		if(!(h.valid & kHot14)) { h.r14 = REC(14); h.valid |= kHot14; }
		h.dirty |= kHot14;
	}
}
// with_path: throughput and path colour (11, 12) exist — not during the camera vertex's own estimate (st_start_path sets them up)
YG_DEV void hot_preload(const WfArgs &a, uint32_t slot, Hot &h, bool with_path)
{
	h.r14 = REC(14);
	h.valid = 0x88u; h.dirty = 0u; h.acc_zero = 0u;
	if(with_path || !YAFGPU_ACC_ZERO_FLAG) { h.r11 = REC(11); h.r12 = REC(12); h.valid |= 0x03u; }
	h.tot_zero = (YAFGPU_ACC_ZERO_FLAG && (ubits(h.r14.w) & kDlcTotZero)) ? 1u : 0u;
	if(h.tot_zero) h.r18 = make_float4(0.f, 0.f, 0.f, 0.f); else h.r18 = REC(18);
	if(YAFGPU_HOT_ACC)
	{
		h.acc_zero = (YAFGPU_ACC_ZERO_FLAG && (ubits(h.r14.w) & kDlcAccZero)) ? 1u : 0u;
		if(h.acc_zero) { const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f); h.r15 = z4; h.r16 = z4; h.r17 = z4; }
		else { h.r15 = REC(15); h.r16 = REC(16); h.r17 = REC(17); }
		h.valid |= kHotAccBits;
	}
}
// to_closest: the path parks for a closest-hit query — the light estimate's records (14..18) are dead then, st_after_closest starts them afresh
YG_DEV void hot_flush(const WfArgs &a, uint32_t slot, const Hot &h, bool to_closest)
{
	if(h.dirty & 0x01u) REC(11) = h.r11;
	if(h.dirty & 0x02u) REC(12) = h.r12;
	if(to_closest && YAFGPU_ACC_ZERO_FLAG) return;
	if(h.dirty & 0x08u)
	{
		float4 r14 = h.r14;
		if(YAFGPU_ACC_ZERO_FLAG) r14.w = fbits((ubits(r14.w) & ~(kDlcAccZero | kDlcTotZero)) | (h.acc_zero ? kDlcAccZero : 0u) | (h.tot_zero ? kDlcTotZero : 0u));
		REC(14) = r14;
	}
	if(YAFGPU_HOT_ACC && !h.acc_zero)
	{
		if(h.dirty & 0x10u) REC(15) = h.r15;
		if(h.dirty & 0x20u) REC(16) = h.r16;
		if(h.dirty & 0x40u) REC(17) = h.r17;
	}
	if((h.dirty & 0x80u) && !h.tot_zero) REC(18) = h.r18;
}
#define HGET(k) hot_get<k>(a, slot, h)
#define HSET(k, v) hot_set<k>(a, slot, h, (v))
#if !YAFGPU_FEAT_LIGHTS
// the vertex records of a record pass's kernel: kept in registers for the resume; record 6 also goes to memory (its .w is integrate()'s `w`,
// which lives across resumes), records 3..5 only when a later path sample will start from the camera hit again, 7..10 never
template<int K> YG_DEV void vtx_set(const WfArgs &a, uint32_t slot, Hot &h, float4 v)
{
	static_assert(K >= 3 && K <= 10, "not a vertex record");
	h.vt[K - 3] = v; h.vvalid |= 1u << (K - 3);
	if(K == 6 || (K <= 5 && a.ra.rp.path_samples > 1)) REC(K) = v;
}
template<int K> YG_DEV float4 vtx_get(const WfArgs &a, uint32_t slot, Hot &h)
{
	static_assert(K >= 3 && K <= 10, "not a vertex record");
	return (h.vvalid & (1u << (K - 3))) ? h.vt[K - 3] : REC(K);
}
#endif

#define FREC(L, j) wf_rec(a, kWfRecs + a.frame_recs * (L) + (j), slot)      // recursion frames, see st_recurse
// path caustics (yafgpu_render_params::trace_caustics): a kernel built with 0 serves renders with caustic_type "none"
#ifndef YAFGPU_FEAT_CAUSTIC
#define YAFGPU_FEAT_CAUSTIC 1
#endif
// The light estimate.  A kernel built with 0 is the program of a RECORD pass and nothing else (WfArgs::replay == 1 is a given): it follows the
// paths, notes the light calls and the roulette probabilities, keeps the hits — no shadow parks, no estimate steps, and the vertex a hit makes
// goes to the sampler of the same resume in registers instead of through records 7..10 (and 3..5 when no later path sample needs them).
#ifndef YAFGPU_FEAT_LIGHTS
#define YAFGPU_FEAT_LIGHTS 1
#endif
// Two MIS pairs per park (WfArgs::multi).  A kernel built with 0 parks one pair at a time: the second pair's bookkeeping costs the path
// program 14 more spilled registers (the diffuse variant's launches +10 %), so scenes whose light estimates never have a second pair to
// offer — one light with one sample, the benchmark scenes — run kernels built without it (yafgpu_device.hip: pick_shade_variant).
#ifndef YAFGPU_FEAT_MULTI
#define YAFGPU_FEAT_MULTI 1
#endif
#if YAFGPU_FEAT_LIGHTS
#define A_REPLAY (a.replay)
#define VGET(k) REC(k)
#define VSET(k, v) (REC(k) = (v))
#else
#define A_REPLAY 1
#define VGET(k) vtx_get<k>(a, slot, h)
#define VSET(k, v) vtx_set<k>(a, slot, h, (v))
#endif
// recursiveRaytrace (frames, absorption): a kernel built with 0 serves scenes without specular / filter materials
#ifndef YAFGPU_FEAT_RECURSE
#define YAFGPU_FEAT_RECURSE 1
#endif
// Trajectory splitting (RenderState::ray_division_ / ray_offset_ / dc_1_ / dc_2_, scene.h:88-91): set by recursiveRaytrace's
// glossy branch for the integrate() calls below it, read by the light, path and glossy samplers of that level.  The level
// above keeps it in its frame (record 5); level 0 and scenes without such materials have (1, 0, 0, 0).
struct DivState { int division, offset; float dc_1, dc_2; };
YG_DEV DivState wf_div(const WfArgs &a, uint32_t slot, int level)
{
	DivState d; d.division = 1; d.offset = 0; d.dc_1 = 0.f; d.dc_2 = 0.f;
	if(YAFGPU_FEAT_RECURSE && a.has_glossy && level > 0)
	{
		const float4 f = FREC(level - 1, 5);
		d.division = (int)ubits(f.x); d.offset = (int)ubits(f.y); d.dc_1 = f.z; d.dc_2 = f.w;
	}
	return d;
}
YG_DEV float add_mod_1(float x, float y) { const float t = x + y; return t > 1 ? t - 1.f : t; }      // util_sample.h:183-187

// where the answer of the closest-hit query a path is parked on lives in the record pass's cache: from what its control word and
// record 19 hold at the park — (call, path sample, segment); a level's own ray (segment 0) belongs to the call about to start
YG_DEV uint32_t wf_hit_key(const WfArgs &a, uint32_t slot, uint32_t ctl, uint32_t z19)
{
	const int stage = (int)((ctl >> 2) & 3u), level = (int)((ctl >> 5) & 7u), depth = (int)((ctl >> 8) & 0xfu), path_i = (int)((ctl >> 17) & 0x1fffu);
	const uint32_t n_ps = (uint32_t)max(a.ra.rp.path_samples, 1), per_path = (uint32_t)max(a.ra.rp.bounces, 1) + 1u;
	uint32_t call = a.ev_m > 1 ? (z19 >> 24) : 0u, seg = 0u, ps = 0u;
	if(stage == kStPrimary) call = level > 0 ? call + 1u : 0u;      // (the camera ray: record 19 is not set up yet)
	else { ps = (uint32_t)path_i; seg = stage == kStFirst ? 1u : 1u + (uint32_t)depth; }
	call = min(call, (uint32_t)a.ev_m - 1u); seg = min(seg, per_path - 1u); ps = min(ps, n_ps - 1u);
	return slot * (uint32_t)a.hit_k + (call * n_ps + ps) * per_path + seg;
}
// the control word a segment parked BESIDE a shadow pair (st_beside) would have been parked with by st_start_path / st_extend after the
// pair: camera hit -> first segment; first hit -> depth 1; depth d -> depth d + 1 (the word in record 13 is still the vertex's)
YG_DEV uint32_t wf_ctl_of_beside(uint32_t ctl)
{
	const uint32_t stage = (ctl >> 2) & 3u, depth = (ctl >> 8) & 0xfu;
	const uint32_t stage_n = stage == (uint32_t)kStPrimary ? (uint32_t)kStFirst : (uint32_t)kStDepth;
	const uint32_t depth_n = stage == (uint32_t)kStPrimary ? 0u : (stage == (uint32_t)kStFirst ? 1u : depth + 1u);
	return (ctl & ~((3u << 2) | (0xfu << 8))) | (stage_n << 2) | (depth_n << 8);
}

// the closest-hit query of this path was answered: shade the new vertex up to its light estimate
// beside: the segment was parked beside the previous vertex's shadow pair — its direction is in record 26, its origin the vertex in record 0
YG_DEV int st_after_closest(const WfArgs &a, uint32_t slot, Hot &h, Ctl &c, uint32_t ordinal, const float4 ans, bool beside)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc; const yafgpu_render_params &rp = ra.rp;
	const int tri = (int)ubits(ans.x);
	const bool got = tri >= 0;
	if(A_REPLAY == 1 && a.hit_cache != nullptr)      // record pass: keep the answer for the final pass (the control word is still the park's)
		a.hit_cache[wf_hit_key(a, slot, pack_ctl(c), a.ev_m > 1 ? ubits(REC(19).z) : 0u)] = ans;
	if(c.stage == kStPrimary)
	{
		c.col = mkc(0.f, 0.f, 0.f);
		float alpha = rp.bg_transp ? 0.f : 1.f;
		// a recursion level's own ray: the level above wants its tmax_ (the hit distance; -1 on a miss) for absorption
		if(YAFGPU_FEAT_RECURSE && c.level > 0) FREC(c.level - 1, 3).w = got ? ans.y : -1.f;
		if(!got)
		{
			if(rp.has_background && !rp.bg_transp_refract) c.col = c.col + mkc(rp.background[0], rp.background[1], rp.background[2]);
			// (z: the sample's call ordinal and light calls so far belong to the whole sample, see WfArgs::ev_m)
			REC(19) = make_float4(0.f, 0.f, (a.ev_m > 1 && c.level > 0) ? fbits(ubits(REC(19).z) + (1u << 24)) : 0.f, alpha);      // (a call that ends here is a call too: its ordinal is its own)
			return W_RETURN;
		}
		if(c.level == 0) c.incl = 1;                                                         // integrator_path_tracer.cc:129-135
		const float4 r0 = REC(0), r1 = REC(1);
		const V3 dir = v3(r1);
		SurfPt sp0;
		get_surface(sc, tri, v3(r0) + dir * ans.y, ans.z, ans.w, sp0);
		if(YAFGPU_FEAT_TEXTURE && sc.tex.has_bump) { const bool bumped = wf_bump_hit(sc, sp0, tri, ans.z, ans.w); REC(24) = f4(sp0.nu, bumped ? 1.f : 0.f); }
		yafgpu_material m_tmp;
		const yafgpu_material &m = wf_mat_hit(sc, sp0, tri, ans.z, ans.w, m_tmp);
		if(YAFGPU_FEAT_TEXTURE && sc.tex.nodes != nullptr) REC(22) = make_float4(fbits((uint32_t)tri), ans.z, ans.w, 0.f);
		BsdfDat dat0;
		const uint32_t bsdfs0 = mat_init_bsdf(m, dat0);
		if(YAFGPU_FEAT_RECURSE) c.add = max(c.add, min(m.additional_depth, 15));                 // integrator_path_tracer.cc:149
		const V3 wo0 = -dir;
		if(bsdfs0 & kEmit) c.col = c.col + mat_emit(m, sp0, wo0, c.incl != 0);                // :152 (include_lights_ :133)
		alpha = 1.f;
		if(rp.bg_transp_refract)
		{
			const float m_alpha = mat_alpha(m, dat0, sp0, wo0);
			alpha = m_alpha + (1.f - m_alpha) * (rp.bg_transp ? 0.f : 1.f);
		}
		VSET(3, f4(sp0.p, fbits((uint32_t)sp0.mat))); VSET(4, f4(sp0.n, 0.f)); VSET(5, f4(sp0.ng, fbits(bsdfs0))); VSET(6, f4(wo0, 0.f));
		// (throughput, path colour and the roulette stream of this level's path samples: st_start_path, at the first of them)
		uint32_t z19 = 0u;
		if(a.ev_m > 1 && c.level > 0) z19 = ubits(REC(19).z) + (1u << 24);      // the next integrate() call of this camera sample
		REC(19) = make_float4(fbits(0u), fbits((uint32_t)kNone), fbits(z19), alpha);
		// (a record pass follows the paths and nothing else: the light estimate's records — 12, 14..18, and 11 without a roulette test to record —
		// are neither written nor read by it; the final pass sets every one of them up again)
		const bool rec = A_REPLAY == 1;
		if(!rec) hot_zero_tot(a, slot, h);
		c.path_i = 0; c.depth = 0;
		if((bsdfs0 & kDiffuse) && sc.n_lights > 0 && A_REPLAY != 1)      // (a record pass only follows the paths)
		{
			HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(0, sc.n_lights, 0, 0))));
			hot_zero_acc(a, slot, h);
			c.dl_on_sp0 = 1;
			return W_DL_NEXT;
		}
		if(!rec) HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(0, 0, 0, 0))));
		return W_DL_DONE;
	}
	if(!got) { ++c.path_i; return W_START_PATH; }                                              // :218 / :259-266
	const float4 r0 = REC(0), r1 = beside ? REC(26) : REC(1);
	const V3 dir = v3(r1);
	SurfPt hit;
	get_surface(sc, tri, v3(r0) + dir * ans.y, ans.z, ans.w, hit);
	if(YAFGPU_FEAT_TEXTURE && sc.tex.has_bump) { const bool bumped = wf_bump_hit(sc, hit, tri, ans.z, ans.w); REC(25) = f4(hit.nu, bumped ? 1.f : 0.f); }
	yafgpu_material pm_tmp;
	const yafgpu_material &pm = wf_mat_hit(sc, hit, tri, ans.z, ans.w, pm_tmp);
	if(YAFGPU_FEAT_TEXTURE && sc.tex.nodes != nullptr) REC(23) = make_float4(fbits((uint32_t)tri), ans.z, ans.w, 0.f);
	BsdfDat dat_n;
	const uint32_t mb = mat_init_bsdf(pm, dat_n);
	float4 misc = REC(19);
	V3 pwo = -dir;                                                                              // :271
	if(c.stage == kStFirst && ubits(misc.y) == kNone) pwo = v3(REC(10));                       // :224: keeps the first segment's pwo
	// .w of 8..10: p_ray.dir_ of the segment that ended here — what Material::sample leaves in `wi` when it samples nothing
	VSET(7, f4(hit.p, fbits((uint32_t)hit.mat))); VSET(8, f4(hit.n, dir.x)); VSET(9, f4(hit.ng, dir.y)); VSET(10, f4(pwo, dir.z));
	if(A_REPLAY != 1) hot_zero_tot(a, slot, h);
	if(YAFGPU_FEAT_RECURSE && (mb & kVolumetric) && c.stage == kStDepth && pm.has_vol_i && dot(hit.n, pwo) < 0.f)
	{	// integrator_path_tracer.cc:276-279: the segment ran inside an absorbing material (lcol does not depend on it)
		const float4 r11 = HGET(11);
		HSET(11, f4(c3(r11) * beer_transmittance(pm.beer_sigma, ans.y), r11.w));
	}
	const bool want_dl = sc.n_lights > 0 && (c.stage == kStFirst || (mb & kDiffuse));
	if(want_dl && A_REPLAY == 1)
	{	// record pass: note the call (its depth: 0 at the first hit), skip the estimate — it does not steer the path
		const uint32_t e = wf_event(a, slot, ubits(misc.z), c.path_i);
		a.ev_flags[e] |= 1u << (c.stage == kStFirst ? 0 : c.depth);
		return W_DL_DONE;
	}
	if(want_dl)
	{	// estimateOneDirectLight, integrator_montecarlo.cc:62-76
		const uint32_t calls = ubits(misc.z) & 0xffffffu;
		int lnum = 0;
		if(sc.n_lights > 1)
		{
			// correlative_sample_number_[thread]: the number of calls before this one in the reference's single-thread
			// order, replayed (lc_base: where this sample starts); without the replay a per-sample ordinal stands in
			const uint32_t counter = (A_REPLAY == 2 && a.replay_lights) ? a.lc_base[slot] + calls : ordinal * 16u + calls;
			Halton h2; h2.init(2u);
			h2.set_start(rp.base_sampling_offset + counter - 1u);
			lnum = min((int)(h2.next() * (float)sc.n_lights), sc.n_lights - 1);
		}
		if(sc.n_lights > 1) { misc.z = fbits(ubits(misc.z) + 1u); REC(19) = misc; }      // (only the choice among several lights reads it)
		HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(lnum, lnum + 1, 0, 0))));
		hot_zero_acc(a, slot, h);
		c.dl_on_sp0 = 0;
		return W_DL_NEXT;
	}
	if(A_REPLAY != 1) HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(0, 0, 0, 0))));   // l_end == 0: no light estimate ran
	return W_DL_DONE;
}

// the shadow rays of one MIS pair were answered: add what was unoccluded
// second: the SECOND pair of the park (WfArgs::multi) — the path stands at its W_DL_EVAL (record 14 names the pair, st_dl_next put it there); what
// st_dl_eval would have left is in records 29 (first ray's contribution) and 31 (second ray's), which rays went out in mask2
YG_DEV int st_after_shadow(const WfArgs &a, uint32_t slot, Hot &h, uint2 verdict, bool second = false, int mask2 = 0)
{
	const DevScene &sc = a.ra.sc;
	float4 r14 = HGET(14);
	const uint32_t w = ubits(r14.w);
	const int li = (int)(w & 0xffu), l_end = (int)((w >> 8) & 0xffu), mask = second ? mask2 : (int)((w >> 16) & 0x3u), is = (int)(w >> 20);
	if(second) { const float4 r29 = REC(29); r14.x = r29.x; r14.y = r29.y; r14.z = r29.z; }
	const bool dirac = sc.lights[li].type == YAFGPU_LIGHT_POINT;
	// transparent shadows (integrator_montecarlo.cc:114,182,309): what an unblocked ray picked up on its way scales the
	// light.  (The reference scales the light colour before forming the contribution, here the parked contribution is
	// scaled: same product, other rounding order.)
	Col fa = mkc(1.f, 1.f, 1.f), fb = fa;
	if(a.shadow_filt != nullptr) { fa = c3(a.shadow_filt[2u * slot]); fb = c3(a.shadow_filt[2u * slot + 1u]); }
	if((mask & 1) && verdict.x == 0u)
	{
		const Col add = (a.shadow_filt != nullptr) ? c3(r14) * fa : c3(r14);
		if(dirac) HSET(17, f4(c3(HGET(17)) + add, 0.f));
		else HSET(15, f4(c3(HGET(15)) + add, 0.f));
	}
	if((mask & 2) && verdict.y == 0u)
	{
		const float4 rb = second ? REC(31) : REC(21);
		const Col add = (a.shadow_filt != nullptr) ? c3(rb) * fb : c3(rb);
		HSET(16, f4(c3(HGET(16)) + add, 0.f));
	}
	r14.w = fbits(pack_dlc(li, l_end, 0, is + 1));
	HSET(14, r14);
	return W_DL_NEXT;
}

// direct_light()'s loops, with the two halves of a MIS pair (same light, same sample index) taken together:
// each half adds into its own accumulator (ccol / ccol_2) in sample order, exactly as `for phase { for is }` does.
// The loops are cut in two steps.  st_dl_next is the bookkeeping: it closes every light whose samples are all in
// and says whether a candidate pair has to be evaluated next (W_DL_EVAL).  st_dl_eval evaluates that pair — the
// widest step of the path program (81 VGPRs).
YG_DEV int st_dl_next(const WfArgs &a, uint32_t slot, Hot &h, int level)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc;
	const int division = wf_div(a, slot, level).division;
	const uint32_t w = ubits(HGET(14).w);
	int li = (int)(w & 0xffu), is = (int)(w >> 20);
	const int l_end = (int)((w >> 8) & 0xffu);
	while(li < l_end)
	{
		const yafgpu_light &light = sc.lights[li];
		const bool dirac = light.type == YAFGPU_LIGHT_POINT;
		const int n = dirac ? 1 : dl_area_samples(ra, light, division);
		if(is < n)
		{
			HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(li, l_end, 0, is))));
			return W_DL_EVAL;
		}
		const float inv_ns = 1.f / (float)n;
		Col col = mkc(0.f, 0.f, 0.f);
		if(dirac) col = col + c3(HGET(17));
		else { col = col + c3(HGET(15)) * inv_ns; col = col + c3(HGET(16)) * inv_ns; }
		HSET(18, f4(c3(HGET(18)) + col, 0.f));
		hot_zero_acc(a, slot, h);
		is = 0; ++li;
	}
	HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(li, l_end, 0, is))));
	return W_DL_DONE;
}

// second: evaluate the pair named by `w2` as the SECOND pair of the park about to happen (WfArgs::multi): its rays and contributions go to records
// 28..31, record 14 and the accumulators are left alone; W_PARK_SHADOW with the rays wanted in out_mask, anything else: no second pair
YG_DEV int st_dl_eval(const WfArgs &a, uint32_t slot, Hot &h, const Ctl &c, uint32_t pixel_sample, uint32_t sampling_offs, int &out_mask, bool second = false, uint32_t w2 = 0u)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc;
	const uint32_t w = second ? w2 : ubits(HGET(14).w);
	const int li = (int)(w & 0xffu), is = (int)(w >> 20), l_end = (int)((w >> 8) & 0xffu);
	SurfPt sp; V3 wo;
	if(c.dl_on_sp0) { const float4 p = REC(3); make_sp(v3(p), v3(REC(4)), v3(REC(5)), (int)ubits(p.w), sp); wo = v3(REC(6)); }
	else { const float4 p = REC(7); make_sp(v3(p), v3(REC(8)), v3(REC(9)), (int)ubits(p.w), sp); wo = v3(REC(10)); }
	wf_frame_parked(a, slot, c.dl_on_sp0 ? 0 : 1, sp);
	yafgpu_material mat_tmp;
	const yafgpu_material &mat = wf_mat_parked(a, slot, c.dl_on_sp0 ? 0 : 1, sp, mat_tmp);
	BsdfDat dat; mat_init_bsdf(mat, dat);
	const yafgpu_light &light = sc.lights[li];
	const bool dirac = light.type == YAFGPU_LIGHT_POINT;
	const bool cast_shadows = light.cast_shadows && mat.receive_shadows;
	if(second && !cast_shadows) return W_DL_NEXT;      // (its contributions would go straight into the accumulators: out of order before the first pair's)
	float s_1 = 0.f, s_2 = 0.f;
	if(!dirac) dl_samples(ra, light, li, is, wf_div(a, slot, c.level).division, pixel_sample, sampling_offs, s_1, s_2);
	V3 d; float tmin, tmax; Col contrib;
	int mask = 0;
	Col pending_a = mkc(0.f, 0.f, 0.f);
	float tmin_a = 0.f;
	if(dl_candidate(ra, light, 0, s_1, s_2, sp, mat, dat, wo, d, tmin, tmax, contrib))
	{
		if(cast_shadows) { pending_a = contrib; tmin_a = tmin; wf_rec(a, second ? 28 : 1, slot) = f4(d, tmax); mask |= 1; }
		else if(dirac) HSET(17, f4(c3(HGET(17)) + contrib, 0.f));
		else HSET(15, f4(c3(HGET(15)) + contrib, 0.f));
	}
	if(!dirac && dl_candidate(ra, light, 1, s_1, s_2, sp, mat, dat, wo, d, tmin, tmax, contrib))
	{
		if(cast_shadows) { wf_rec(a, second ? 30 : 20, slot) = f4(d, tmin); wf_rec(a, second ? 31 : 21, slot) = f4(contrib, tmax); mask |= 2; }
		else HSET(16, f4(c3(HGET(16)) + contrib, 0.f));
	}
	if(second)
	{
		if(!mask) return W_DL_NEXT;
		REC(29) = f4(pending_a, tmin_a);      // (the origin is the first pair's, record 0)
		out_mask = mask;
		return W_PARK_SHADOW;
	}
	if(mask)
	{
		REC(0) = f4(sp.p, tmin_a);
		HSET(14, f4(pending_a, fbits(pack_dlc(li, l_end, mask, is))));
		out_mask = mask;
		return W_PARK_SHADOW;
	}
	HSET(14, make_float4(0.f, 0.f, 0.f, fbits(pack_dlc(li, l_end, 0, is + 1))));
	return W_DL_NEXT;
}

// the light estimate of the current vertex is complete: book it and decide how the path goes on
// beside: the next segment was sampled and parked together with this vertex's last shadow pair (st_beside): its answer is in too
YG_DEV int st_dl_done(const WfArgs &a, uint32_t slot, Hot &h, Ctl &c, bool beside)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc; const yafgpu_render_params &rp = ra.rp;
	// a record pass: no estimate ran, and of the path's records only the throughput matters, and only where a roulette test will want its probability
	const bool rec = A_REPLAY == 1, rec_thr = !rec || rp.bounces - 1 > rp.rr_min_bounces;
	const Col total = rec ? mkc(0.f, 0.f, 0.f) : c3(HGET(18));
	const int l_end = rec ? 0 : (int)((ubits(HGET(14).w) >> 8) & 0xffu);
	if(!rec) hot_zero_tot(a, slot, h);              // taken: nothing reads it again before the next vertex zeroes it
	if(c.stage == kStPrimary)
	{
		const uint32_t bsdfs0 = ubits(VGET(5).w);
		if(bsdfs0 & kDiffuse) c.col = c.col + total;                                            // :156
		const uint32_t path_flags = rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse;
		if(rp.integrator != YAFGPU_INTEGRATOR_PATH || !(bsdfs0 & path_flags)) return W_RECURSE;
		c.path_i = 0;
		if(beside) { c.stage = kStFirst; return W_NEXT_VERTEX; }      // st_start_path's work was done by st_beside
		return W_START_PATH;
	}
	const yafgpu_material &pm_rec = sc.mats[(int)ubits(VGET(7).w)];
	BsdfDat dat_n;
	const uint32_t mb = mat_init_bsdf(pm_rec, dat_n);      // the flags do not depend on the nodes
	yafgpu_material pm_tmp; (void)pm_tmp;
	const yafgpu_material *pm_p = &pm_rec;
	const bool caustic = YAFGPU_FEAT_CAUSTIC && c.stage == kStDepth && rp.trace_caustics && c.incl;      // the segment that ended here: :252
#if YAFGPU_FEAT_TEXTURE
	if((c.stage == kStFirst || caustic) && (mb & kEmit) && pm_rec.n_nodes > 0 && sc.tex.nodes != nullptr)
	{	// its own emission reads the diffuse shader (emit(), material_shiny_diffuse.cc:295-306)
		const float4 p7 = REC(7);
		SurfPt hp; hp.p = v3(p7); hp.n = v3(REC(8)); hp.ng = v3(REC(9)); hp.mat = (int)ubits(p7.w);
		pm_p = &wf_mat_parked(a, slot, 1, hp, pm_tmp);
	}
#endif
	const yafgpu_material &pm = *pm_p;
	Col lcol = mkc(0.f, 0.f, 0.f);
	if(l_end > 0) lcol = total * (float)sc.n_lights;
	const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
	float4 r11 = rec_thr ? HGET(11) : z4, r12 = rec ? z4 : HGET(12);
	Col throughput = c3(r11), path_col = c3(r12);
	if(c.stage == kStFirst)
	{
		if(mb & kEmit)
		{	// :226 — include_lights_ is false here, so only shinydiffuse's own emission can contribute
			SurfPt dummy; dummy.n = mk(0.f, 0.f, 0.f);
			lcol = lcol + mat_emit(pm, dummy, mk(0.f, 0.f, 0.f), false);
		}
		path_col = path_col + lcol * throughput;                                                // :228
		if(!rec) HSET(12, f4(path_col, r12.w));
		c.depth = 1;
		if(beside)
		{	// st_extend's work was done by st_beside: throughput *= scol (:251), include_lights_ = caustic (:253)
			const float4 r27 = REC(27);
			HSET(11, f4(throughput * c3(r27), r11.w));
			c.incl = (int)(ubits(r27.w) & 1u);
			c.stage = kStDepth;
			return W_NEXT_VERTEX;
		}
		if(c.depth < rp.bounces) return W_EXTEND;
		++c.path_i;
		return W_START_PATH;
	}
	bool alive = true;
	if(c.depth > rp.rr_min_bounces)
	{	// Russian roulette :282-288
		const float probability = smax(throughput.r, smax(throughput.g, throughput.b));
		if(A_REPLAY != 0)
		{
			const uint32_t e = wf_event(a, slot, a.ev_m > 1 ? ubits(REC(19).z) : 0u, c.path_i);
			if(A_REPLAY == 1)
			{	// record: the test and its probability; the draw is the tile stream's, made by wf_replay_scan in sample order
				a.ev_p[(size_t)e * (size_t)max(rp.bounces - 1, 1) + (size_t)(c.depth - 1)] = probability;
				a.ev_flags[e] |= 1u << (16 + c.depth);
				alive = probability > 0.f;
			}
			else alive = (int)a.ev_kill[e] != c.depth;
			if(alive) throughput = throughput * (1.f / probability);
		}
		else
		{
			Mwc rr; rr.x = ubits(r11.w); rr.c = ubits(r12.w);
			const float random_value = (float)rr.next();
			r11.w = fbits(rr.x); r12.w = fbits(rr.c);
			if(probability <= 0.f || probability < random_value) alive = false;
			else throughput = throughput * (1.f / probability);
		}
	}
	if(alive)
	{
		if(caustic && (mb & kEmit) && !rec)
		{	// :290 a vertex reached through a caustic lobe adds what it emits, its lights included (include_lights_ is set)
			const float4 p7 = REC(7);
			SurfPt hp; hp.p = v3(p7); hp.n = v3(REC(8)); hp.ng = v3(REC(9)); hp.mat = (int)ubits(p7.w);
			lcol = lcol + mat_emit(pm, hp, v3(REC(10)), true);
		}
		path_col = path_col + lcol * throughput;                                                // :292
		++c.depth;
	}
	if(beside)
	{	// (no roulette test at this vertex, or st_beside would not have gone ahead)  throughput *= scol (:251), include_lights_ = caustic (:253)
		const float4 r27 = REC(27);
		HSET(11, f4(throughput * c3(r27), r11.w)); HSET(12, f4(path_col, r12.w));
		c.incl = (int)(ubits(r27.w) & 1u);
		c.stage = kStDepth;
		return W_NEXT_VERTEX;
	}
	if(rec_thr) HSET(11, f4(throughput, r11.w));
	if(!rec) HSET(12, f4(path_col, r12.w));
	if(alive && c.depth < rp.bounces) return W_EXTEND;
	++c.path_i;
	return W_START_PATH;
}

// next segment from the current vertex, :232-257
YG_DEV int st_extend(const WfArgs &a, uint32_t slot, Hot &h, Ctl &c)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc;
	const float4 p = VGET(7);
	SurfPt hit; make_sp(v3(p), v3(VGET(8)), v3(VGET(9)), (int)ubits(p.w), hit);
	wf_frame_parked(a, slot, 1, hit);
	const float4 r10 = VGET(10);
	const V3 pwo = v3(r10);
	yafgpu_material pm_tmp;
	const yafgpu_material &pm = wf_mat_parked(a, slot, 1, hit, pm_tmp);
	BsdfDat dat_n; mat_init_bsdf(pm, dat_n);
	const uint32_t offs = ubits(REC(19).x);
	const int d_4 = 4 * c.depth;
	BsdfSample bs;
	bs.s_1 = (float)scr_halton(sc, d_4 + 3, offs);
	bs.s_2 = (float)scr_halton(sc, d_4 + 4, offs);
	bs.pdf = 0.f; bs.sampled = kNone; bs.flags = kAll;
	// `w` and `p_ray.dir_` are variables of integrate() that sample() may leave untouched (a material with no lobe to
	// sample returns Rgb(1) and nothing else, material_shiny_diffuse.cc sample()): the path then carries on straight
	// through with the previous weight (integrator_path_tracer.cc:243-249).  They live in REC(6).w and in 8..10.w.
	float w = REC(6).w;
	V3 p_dir = mk(VGET(8).w, VGET(9).w, r10.w);
	const Col scol = mat_sample(pm, dat_n, hit, pwo, p_dir, bs, w) * w;
	REC(6).w = w;
	if(is_black(scol)) { ++c.path_i; return W_START_PATH; }                                      // :249 `break`
	if(A_REPLAY != 1 || ra.rp.bounces - 1 > ra.rp.rr_min_bounces)      // (a record pass without a roulette test to record has no use for the throughput)
	{
		const float4 r11 = HGET(11);
		HSET(11, f4(c3(r11) * scol, r11.w));
	}
	// :252-253 caustic = trace_caustics_ && the lobe sampled is specular, glossy or a filter; state.include_lights_ = caustic
	c.incl = (ra.rp.trace_caustics && (bs.sampled & (kSpecular | kGlossy | kFilter))) ? 1 : 0;
	REC(0) = f4(hit.p, ra.ray_min_dist); REC(1) = f4(p_dir, -1.f);
	c.stage = kStDepth;
	return W_PARK_CLOSEST;
}

// first segment of path sample `path_i` from the camera hit, :186-216 — or the end of the sample
YG_DEV int st_start_path(const WfArgs &a, uint32_t slot, Hot &h, Ctl &c, uint32_t pixel_sample, uint32_t sampling_offs, uint32_t ordinal)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc; const yafgpu_render_params &rp = ra.rp;
	const DivState dv = wf_div(a, slot, c.level);
	const int n_paths = max(1, rp.path_samples / dv.division);                                     // :182 n_samples
	if(c.path_i >= n_paths) { if(A_REPLAY != 1) c.col = c.col + c3(HGET(12)) / (float)n_paths; return W_RECURSE; } // :297
	c.incl = 0;                                                                                   // :211 state.include_lights_ = false
	const float4 p = VGET(3);
	SurfPt sp0; make_sp(v3(p), v3(VGET(4)), v3(VGET(5)), (int)ubits(p.w), sp0);
	wf_frame_parked(a, slot, 0, sp0);
	const V3 wo0 = v3(VGET(6));
	yafgpu_material m_tmp;
	const yafgpu_material &m = wf_mat_parked(a, slot, 0, sp0, m_tmp);
	BsdfDat dat0; mat_init_bsdf(m, dat0);
	const uint32_t offs = (uint32_t)rp.path_samples * pixel_sample + sampling_offs + (uint32_t)c.path_i;
	BsdfSample bs;
	bs.s_1 = ri_vdc(offs, 0u);
	bs.s_2 = (float)scr_halton(sc, 2, offs);
	if(dv.division > 1) { bs.s_1 = add_mod_1(bs.s_1, dv.dc_1); bs.s_2 = add_mod_1(bs.s_2, dv.dc_2); }   // :201-205
	bs.pdf = 0.f; bs.sampled = kNone;
	bs.flags = (rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse) | kDiffuse | kReflect | kTransmit;
	const float4 r6 = VGET(6);
	float w = r6.w;                                   // integrate()'s `w`: 0 at its start, then whatever the last sample() left
	V3 p_dir = mk(0.f, 0.f, 0.f);
	const Col scol = mat_sample(m, dat0, sp0, wo0, p_dir, bs, w) * w;
	REC(6).w = w;
	if(bs.sampled == kNone || !YAFGPU_ACC_ZERO_FLAG) REC(10) = f4(wo0, 0.f);      // pwo = wo: only a segment that sampled nothing keeps it (:224, st_after_closest)
	if(A_REPLAY == 1)
	{	// (record pass: the throughput alone, and only for a roulette test's probability)
		if(rp.bounces - 1 > rp.rr_min_bounces) HSET(11, f4(scol, 0.f));
	}
	else if(c.path_i == 0)
	{	// the level's first path sample: path colour 0 and the roulette stream of the per-sample mode start here
		Mwc rr; rr.init(fnv32a(ordinal) + 123u);
		HSET(11, f4(scol, fbits(rr.x)));
		HSET(12, make_float4(0.f, 0.f, 0.f, fbits(rr.c)));
	}
	else HSET(11, f4(scol, HGET(11).w));              // throughput = scol
	float4 misc = REC(19);
	misc.x = fbits(offs); misc.y = fbits(bs.sampled);
	REC(19) = misc;
	REC(0) = f4(sp0.p, ra.ray_min_dist); REC(1) = f4(p_dir, -1.f);
	c.stage = kStFirst;
	return W_PARK_CLOSEST;
}

// The path is about to park for a shadow pair.  If it is the LAST pair of this vertex's light estimate and the path goes on from here
// whatever the answers are, take the next segment now — st_start_path's sample from the camera hit, or st_extend's from the path
// vertex — and park for both (WfArgs::speculate).  Everything written here is what those steps would write after the answers, except
// what the pending estimate still reads: the ray goes to record 26 (records 0 / 1 hold the shadow ray), st_extend's colour to
// record 27 (record 11 must keep the throughput the estimate is booked with), and the control word's stage is advanced on resume.
// w_last: the (li, l_end, is) word of the last pair of this park — record 14's, or the second pair's (WfArgs::multi)
YG_DEV int st_beside(const WfArgs &a, uint32_t slot, Hot &h, Ctl &c, uint32_t pixel_sample, uint32_t sampling_offs, uint32_t ordinal, uint32_t w_last)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc; const yafgpu_render_params &rp = ra.rp;
	if(!a.speculate || A_REPLAY == 1 || a.frames != 0 || rp.integrator != YAFGPU_INTEGRATOR_PATH) return W_PARK_SHADOW;      // (a record pass has no shadow parks anyway)
	{	// the last pair of the estimate?
		const uint32_t w14 = w_last;
		const int li = (int)(w14 & 0xffu), l_end = (int)((w14 >> 8) & 0xffu), is = (int)(w14 >> 20);
		if(li + 1 != l_end) return W_PARK_SHADOW;
		const yafgpu_light &light = sc.lights[li];
		if(light.type != YAFGPU_LIGHT_POINT && is + 1 != dl_area_samples(ra, light, 1)) return W_PARK_SHADOW;
	}
	if(c.stage == kStPrimary)
	{	// st_dl_done will go to st_start_path(path_i = 0): :186-216
		const float4 r5 = REC(5);
		const uint32_t path_flags = rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse;
		if(!(ubits(r5.w) & path_flags)) return W_PARK_SHADOW;
		c.incl = 0;
		const float4 p = REC(3);
		SurfPt sp0; make_sp(v3(p), v3(REC(4)), v3(r5), (int)ubits(p.w), sp0);
		wf_frame_parked(a, slot, 0, sp0);
		const float4 r6 = REC(6);
		const V3 wo0 = v3(r6);
		yafgpu_material m_tmp;
		const yafgpu_material &m = wf_mat_parked(a, slot, 0, sp0, m_tmp);
		BsdfDat dat0; mat_init_bsdf(m, dat0);
		const uint32_t offs = (uint32_t)rp.path_samples * pixel_sample + sampling_offs;
		BsdfSample bs;
		bs.s_1 = ri_vdc(offs, 0u);
		bs.s_2 = (float)scr_halton(sc, 2, offs);
		bs.pdf = 0.f; bs.sampled = kNone;
		bs.flags = path_flags | kDiffuse | kReflect | kTransmit;
		float w = r6.w;
		V3 p_dir = mk(0.f, 0.f, 0.f);
		const Col scol = mat_sample(m, dat0, sp0, wo0, p_dir, bs, w) * w;
		REC(6).w = w;
		if(bs.sampled == kNone || !YAFGPU_ACC_ZERO_FLAG) REC(10) = f4(wo0, 0.f);
		Mwc rr; rr.init(fnv32a(ordinal) + 123u);
		HSET(11, f4(scol, fbits(rr.x)));
		HSET(12, make_float4(0.f, 0.f, 0.f, fbits(rr.c)));
		float4 misc = REC(19);
		misc.x = fbits(offs); misc.y = fbits(bs.sampled);
		REC(19) = misc;
		REC(26) = f4(p_dir, ra.ray_min_dist);
		return W_PARK_BOTH;
	}
	// st_dl_done will book the estimate and go to st_extend: only if no roulette test stands in between and a bounce is left
	const int next_depth = c.stage == kStFirst ? 1 : c.depth + 1;
	if(next_depth >= rp.bounces) return W_PARK_SHADOW;
	if(c.stage == kStDepth && c.depth > rp.rr_min_bounces)
	{	// a roulette test stands between the estimate and the next segment (:282-288).  With the serial-state replay its outcome is
		// already on the table (ev_kill, from the tile's stream); a per-sample stream would have to be drawn from here: no shortcut then
		if(A_REPLAY != 2) return W_PARK_SHADOW;
		const uint32_t e = wf_event(a, slot, a.ev_m > 1 ? ubits(REC(19).z) : 0u, c.path_i);
		if((int)a.ev_kill[e] == c.depth) return W_PARK_SHADOW;
	}
	const float4 p = REC(7);
	SurfPt hit; make_sp(v3(p), v3(REC(8)), v3(REC(9)), (int)ubits(p.w), hit);
	wf_frame_parked(a, slot, 1, hit);
	const float4 r10 = REC(10);
	const V3 pwo = v3(r10);
	yafgpu_material pm_tmp;
	const yafgpu_material &pm = wf_mat_parked(a, slot, 1, hit, pm_tmp);
	BsdfDat dat_n; mat_init_bsdf(pm, dat_n);
	const uint32_t offs = ubits(REC(19).x);
	const int d_4 = 4 * next_depth;
	BsdfSample bs;
	bs.s_1 = (float)scr_halton(sc, d_4 + 3, offs);
	bs.s_2 = (float)scr_halton(sc, d_4 + 4, offs);
	bs.pdf = 0.f; bs.sampled = kNone; bs.flags = kAll;
	float w = REC(6).w;
	V3 p_dir = mk(REC(8).w, REC(9).w, r10.w);
	const Col scol = mat_sample(pm, dat_n, hit, pwo, p_dir, bs, w) * w;
	if(is_black(scol)) return W_PARK_SHADOW;          // the path sample ends here (:249): nothing was written, st_extend finds it again
	REC(6).w = w;
	REC(26) = f4(p_dir, ra.ray_min_dist);
	// (.w: st_extend's include_lights_ for the segment — the vertex being estimated still needs the one it was reached with)
	REC(27) = f4(scol, fbits((rp.trace_caustics && (bs.sampled & (kSpecular | kGlossy | kFilter))) ? 1u : 0u));
	return W_PARK_BOTH;
}

// recursiveRaytrace, integrator_montecarlo.cc:782-1028 — the perfect specular branch (:971-1025): the level's own
// radiance is complete; follow the reflected and the (filtered, straight-through) transmitted ray with a full
// integrate() each, one level deeper.  The call stack is a frame per level behind the working records:
//   F0 colour so far | alpha without a transmitted ray     F1 transmission weight | material alpha
//   F2 hit point | flags (1: transmitted ray still to go, 2: it is the one out now)
//   F3 transmitted direction                                 F4 reflection weight
YG_DEV void wf_start_level(const WfArgs &a, uint32_t slot, Ctl &c, V3 p, V3 dir)
{
	REC(0) = f4(p, a.ra.ray_min_dist); REC(1) = f4(dir, -1.f);      // DiffRay(sp.p_, dir, scene_->ray_min_dist_)
	c.stage = kStPrimary; c.depth = 0; c.path_i = 0; c.dl_on_sp0 = 0;
}
// the transmitted ray's origin: pushed along the ray by the material's transparent bias (integrator_montecarlo.cc:1003-1011)
YG_DEV V3 wf_transp_origin(const yafgpu_material &m, V3 p, V3 dir, int raylevel)
{
	float f = m.transp_bias_factor;
	if(!(f > 0.f)) return p;
	if(m.transp_bias_mult) f *= (float)raylevel;
	return p + dir * f;
}
// recursiveRaytrace's perfect specular branch (:971-1025) at level c.level, whose working records hold the level's hit
YG_DEV int st_recurse_spec(const WfArgs &a, uint32_t slot, Ctl &c)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc; const yafgpu_render_params &rp = ra.rp; (void)sc; (void)rp;
	if(!YAFGPU_FEAT_RECURSE) return W_RETURN;
	const float4 r5 = REC(5);
	if(!(ubits(r5.w) & (kSpecular | kFilter))) return W_RETURN;
	const float4 p = REC(3);
	SurfPt sp0; make_sp(v3(p), v3(REC(4)), v3(r5), (int)ubits(p.w), sp0);
	wf_frame_parked(a, slot, 0, sp0);
	const V3 wo0 = v3(REC(6));
	yafgpu_material m_tmp;
	const yafgpu_material &m = wf_mat_parked(a, slot, 0, sp0, m_tmp);
	BsdfDat dat0; mat_init_bsdf(m, dat0);
	c.incl = 1;                                                                       // :973
	bool refl, refr; V3 d_refl, d_refr; Col c_refl, c_refr;
	mat_get_specular(m, dat0, sp0, wo0, c.level + 1, refl, refr, d_refl, c_refl, d_refr, c_refr);
	if(!refl && !refr) return W_RETURN;
	const float m_alpha = mat_alpha(m, dat0, sp0, wo0);
	const int L = c.level;
	if(a.has_glossy)
	{	// the rays below keep this level's trajectory-splitting state
		const DivState dv = wf_div(a, slot, L);
		FREC(L, 5) = make_float4(fbits((uint32_t)dv.division), fbits((uint32_t)dv.offset), dv.dc_1, dv.dc_2);
	}
	FREC(L, 0) = f4(c.col, REC(19).w);
	FREC(L, 1) = f4(c_refr, m_alpha);
	// :991, :1016 vol = material->getVolumeHandler(sp.ng_ * ref_ray.dir_ < 0): flags 4 / 8 = the reflected / transmitted ray
	// runs inside the absorbing material
	uint32_t vol = 0u;
	if((ubits(r5.w) & kVolumetric) && m.has_vol_i)
		vol = (refl && dot(sp0.ng, d_refl) < 0.f ? 4u : 0u) | (refr && dot(sp0.ng, d_refr) < 0.f ? 8u : 0u);
	FREC(L, 2) = f4(sp0.p, fbits((refl ? (refr ? 1u : 0u) : 2u) | vol | ((uint32_t)c.add << 8)));      // bits 8-11: this level's additional depth (a level below may raise its own)
	FREC(L, 3) = f4(d_refr, 0.f);                              // .w: the tmax_ of the ray that is out (st_after_closest)
	FREC(L, 4) = f4(c_refl, fbits((uint32_t)sp0.mat));
	wf_start_level(a, slot, c, refl ? sp0.p : wf_transp_origin(sc.mats[sp0.mat], sp0.p, d_refr, L + 1), refl ? d_refl : d_refr);
	c.level = L + 1;
	return W_PARK_CLOSEST;
}
// recursiveRaytrace's glossy branch (:861-972): gsam trajectories through the glossy lobe, each a full integrate() one level down
// under the trajectory-splitting state of frame record 5 — one per trajectory for materials that reflect only (:897-918), two for
// those that reflect and transmit (rough glass, :919-959).  The loop lives in the frame (flag 16 in F2.w):
// F6 normal | ns, gsam, kGl* bits   F7 geometric normal | bsdf flags   F8 wo | integrate()'s w   F9 gcol | material
// F10 the (first) sample's colour | weight   F11 texture coordinates of the hit (tri, bu, bv)   F12 (bump mapping) the hit's nu | bumped
// two directions: F1.x the material's alpha   F3 the second direction | the tmax_ of the ray that is out   F4 its colour | weight
enum : uint32_t { kGlSecond = 1u << 16,      // a second direction waits in F3 / F4
                  kGlOnSecond = 1u << 17,    // the ray that is out is the second one
                  kGlVolFirst = 1u << 18, kGlVolSecond = 1u << 19,      // that ray runs inside the absorbing material (:935, :949)
                  kGlTwo = 1u << 20,         // the two-direction form (factor association :940, :954)
                  kGlBits = 0x1fu << 16 };
YG_DEV int st_glossy_begin(const WfArgs &a, uint32_t slot, Ctl &c)
{
	const int L = c.level;
	const float4 r3 = REC(3), r5 = REC(5);
	const DivState dv = wf_div(a, slot, L);
	const int gsam = dv.division > 1 ? max(1, 8 / dv.division) : 8;
	FREC(L, 0) = f4(c.col, REC(19).w);
	FREC(L, 2) = f4(v3(r3), fbits(16u | ((uint32_t)c.add << 8)));
	FREC(L, 6) = f4(v3(REC(4)), fbits((uint32_t)gsam << 8));
	FREC(L, 7) = r5;
	FREC(L, 8) = REC(6);
	FREC(L, 9) = make_float4(0.f, 0.f, 0.f, r3.w);
	if(YAFGPU_FEAT_TEXTURE && a.ra.sc.tex.nodes != nullptr) FREC(L, 11) = REC(22);
	if(YAFGPU_FEAT_TEXTURE && a.ra.sc.tex.has_bump) FREC(L, 12) = REC(24);
	c.incl = 1;            // :863 state.include_lights_ = true, once: a later trajectory starts with what the one before left (its path samples clear it, :211)
	return W_GLOSSY_NEXT;
}
YG_DEV int st_glossy_next(const WfArgs &a, uint32_t slot, Ctl &c, uint32_t pixel_sample, uint32_t sampling_offs)
{
	const RenderArgs &ra = a.ra; const DevScene &sc = ra.sc;
	const int L = c.level;
	const float4 f2 = FREC(L, 2), f6 = FREC(L, 6), f7 = FREC(L, 7);
	const int ns = (int)(ubits(f6.w) & 0xffu), gsam = (int)((ubits(f6.w) >> 8) & 0xffu);
	SurfPt sp0; make_sp(v3(f2), v3(f6), v3(f7), (int)ubits(FREC(L, 9).w), sp0);
	if(YAFGPU_FEAT_TEXTURE && sc.tex.has_bump) wf_frame_set(sp0, FREC(L, 12));
	const V3 wo0 = v3(FREC(L, 8));
	yafgpu_material m_tmp;
	const yafgpu_material *mp = &sc.mats[sp0.mat];
#if YAFGPU_FEAT_TEXTURE
	if(sc.tex.nodes != nullptr) { const float4 t = FREC(L, 11); mp = &wf_mat_hit(sc, sp0, (int)ubits(t.x), t.y, t.z, m_tmp); }
#endif
	(void)m_tmp;
	BsdfDat dat0; mat_init_bsdf(*mp, dat0);
	const DivState old = wf_div(a, slot, L);
	const int division = old.division * gsam;
	const int branch = division * old.offset + ns;
	const float dc_1 = (float)scr_halton(sc, 2 * (L + 1) + 1, (uint32_t)branch + sampling_offs);           // :886-887
	const float dc_2 = (float)scr_halton(sc, 2 * (L + 1) + 2, (uint32_t)branch + sampling_offs);
	FREC(L, 5) = make_float4(fbits((uint32_t)division), fbits((uint32_t)branch), dc_1, dc_2);
	const uint32_t offs = (uint32_t)gsam * pixel_sample + sampling_offs;
	Halton hal_2, hal_3;
	hal_2.init(2u); hal_3.init(3u);
	hal_2.set_start(offs); hal_3.set_start(offs);
	float s_1 = 0.f, s_2 = 0.f;
	for(int k = 0; k <= ns; ++k) { s_1 = hal_2.next(); s_2 = hal_3.next(); }      // the incremental sequence, replayed
	BsdfSample bs; bs.s_1 = s_1; bs.s_2 = s_2; bs.pdf = 0.f; bs.sampled = kNone; bs.flags = kGlossy | kReflect;
	float w = 0.f;
	V3 wi = mk(0.f, 0.f, 0.f);
	if(YG_IS(*mp, YAFGPU_MAT_ROUGH_GLASS))
	{	// :919-959 Reflect and Transmit: dir[0] goes out under the "reflect" test, dir[1] under the "transmit" one (both as the reference has them)
		bs.flags = kGlossy | kReflect | kTransmit;
		V3 d1 = mk(0.f, 0.f, 0.f); Col tcol = mkc(0.f, 0.f, 0.f); float w1 = 0.f;
		const Col mcol = rough_glass_sample(*mp, sp0, wo0, bs, true, wi, w, d1, tcol, w1);
		const bool vol = (ubits(f7.w) & kVolumetric) && mp->has_vol_i;
		uint32_t bits = kGlTwo;
		if(vol && dot(sp0.ng, wi) < 0.f) bits |= kGlVolFirst;
		if(bs.sampled & kTransmit) { bits |= kGlSecond; if(vol && dot(sp0.ng, d1) < 0.f) bits |= kGlVolSecond; }
		FREC(L, 6) = f4(v3(f6), fbits((ubits(f6.w) & ~kGlBits) | bits));
		FREC(L, 10) = f4(mcol, w);
		FREC(L, 3) = f4(d1, 0.f);
		FREC(L, 4) = f4(tcol, w1);
		if(ns == 0) FREC(L, 1) = make_float4(mat_alpha(*mp, dat0, sp0, wo0), 0.f, 0.f, 0.f);
		wf_start_level(a, slot, c, sp0.p, wi);      // (sampled_flags_ has Reflect on every way out of the sample: the first ray always goes)
		c.level = L + 1;
		return W_PARK_CLOSEST;
	}
	const Col mcol = mat_sample(*mp, dat0, sp0, wo0, wi, bs, w);
	FREC(L, 10) = f4(mcol, w);
	wf_start_level(a, slot, c, sp0.p, wi);
	c.level = L + 1;
	return W_PARK_CLOSEST;
}
// the level's own radiance is complete: recursiveRaytrace (:782-1028; the dispersive branch is refused on the host)
YG_DEV int st_recurse(const WfArgs &a, uint32_t slot, Ctl &c)
{
	const yafgpu_render_params &rp = a.ra.rp;
	if(!YAFGPU_FEAT_RECURSE || c.level + 1 > rp.raydepth + c.add || c.level >= a.frames) return W_RETURN;          // :791
	if(a.has_glossy && (ubits(REC(5).w) & kGlossy)) return st_glossy_begin(a, slot, c);
	return W_RECURSE_SPEC;
}
// an integrate() ends with (c.col, alpha): hand it to the level above, which either sends its transmitted ray or ends too
YG_DEV int st_return(const WfArgs &a, uint32_t slot, Ctl &c, float &alpha_out)
{
	const yafgpu_render_params &rp = a.ra.rp;
	float alpha = REC(19).w;
	for(;;)
	{
		if(rp.bg_transp) alpha = smax(alpha, 0.f);      // EmptyVolumeIntegrator: transmittance 1 (integrator_path_tracer.cc:336-344)
		if(!YAFGPU_FEAT_RECURSE || c.level == 0) { alpha_out = alpha; return W_FINISH; }
		const int P = c.level - 1;
		const float4 f0 = FREC(P, 0), f2 = FREC(P, 2);
		const uint32_t flags = ubits(f2.w);
		Col integ = c.col;
		c.add = (int)((flags >> 8) & 0xfu);            // back in the level above: its own additional depth
		if(flags & 16u)
		{	// a trajectory of the glossy loop is back: gcol += integ * mcol * w (:918), then the next one or the loop's end (:958)
			const float4 f9 = FREC(P, 9), f10 = FREC(P, 10), f6 = FREC(P, 6);
			const uint32_t gl = ubits(f6.w);
			Col gcol;
			float alpha_p = f0.w;
			if(gl & kGlTwo)
			{	// :932-956: integ *= vcol inside the absorbing material, gcol += integ * (mcol * w); the second ray's alpha is the level's
				const bool second = (gl & kGlOnSecond) != 0u;
				if(gl & (second ? kGlVolSecond : kGlVolFirst)) integ = integ * beer_transmittance(a.ra.sc.mats[ubits(f9.w)].beer_sigma, FREC(P, 3).w);
				const float4 fs = second ? FREC(P, 4) : f10;
				gcol = c3(f9) + integ * (c3(fs) * fs.w);
				if(!second && (gl & kGlSecond))
				{
					FREC(P, 9) = f4(gcol, f9.w);
					FREC(P, 6) = f4(v3(f6), fbits((gl & ~kGlSecond) | kGlOnSecond));
					wf_start_level(a, slot, c, v3(f2), v3(FREC(P, 3)));
					return W_PARK_CLOSEST;
				}
				if(second)
				{	// :957 alpha = integ.a_, then integrator_path_tracer.cc:321-326 / integrator_direct_light.cc:165-170
					const float m_alpha = FREC(P, 1).x;
					alpha_p = rp.bg_transp_refract ? m_alpha + (1.f - m_alpha) * alpha : 1.f;
				}
			}
			else gcol = c3(f9) + (integ * c3(f10)) * f10.w;
			const int ns = (int)(gl & 0xffu) + 1, gsam = (int)((gl >> 8) & 0xffu);
			c.level = P;
			if(ns < gsam)
			{
				FREC(P, 9) = f4(gcol, f9.w);
				FREC(P, 6) = f4(v3(f6), fbits((uint32_t)ns | ((uint32_t)gsam << 8)));
				if(alpha_p != f0.w) FREC(P, 0) = f4(c3(f0), alpha_p);
				return W_GLOSSY_NEXT;
			}
			c.col = c3(f0) + gcol * (1.f / (float)gsam);
			// the level's hit again, for the specular branch that follows
			REC(3) = f4(v3(f2), f9.w); REC(4) = f4(v3(f6), 0.f); REC(5) = FREC(P, 7); REC(6) = FREC(P, 8);
			if(YAFGPU_FEAT_TEXTURE && a.ra.sc.tex.nodes != nullptr) REC(22) = FREC(P, 11);
			if(YAFGPU_FEAT_TEXTURE && a.ra.sc.tex.has_bump) REC(24) = FREC(P, 12);
			REC(19) = make_float4(0.f, 0.f, a.ev_m > 1 ? REC(19).z : 0.f, alpha_p);
			return W_RECURSE_SPEC;
		}
		if(flags & ((flags & 2u) ? 8u : 4u))
		{	// the ray ran inside absorbing glass: integ *= vcol (:991-994, :1016-1019)
			const yafgpu_material &pm = a.ra.sc.mats[ubits(FREC(P, 4).w)];
			integ = integ * beer_transmittance(pm.beer_sigma, FREC(P, 3).w);
		}
		if(!(flags & 2u))
		{	// :980-990 the reflected ray is back
			const Col col_p = c3(f0) + integ * c3(FREC(P, 4));
			if(flags & 1u)
			{	// :991-1023 now the transmitted one, at the same level
				FREC(P, 0) = f4(col_p, f0.w);
				FREC(P, 2) = f4(v3(f2), fbits(2u | (flags & 8u) | (flags & 0xf00u)));
				wf_start_level(a, slot, c, wf_transp_origin(a.ra.sc.mats[ubits(FREC(P, 4).w)], v3(f2), v3(FREC(P, 3)), P + 1), v3(FREC(P, 3)));
				return W_PARK_CLOSEST;
			}
			c.col = col_p; alpha = f0.w;
		}
		else
		{
			const float4 f1 = FREC(P, 1);
			c.col = c3(f0) + integ * c3(f1);
			alpha = rp.bg_transp_refract ? f1.w + (1.f - f1.w) * alpha : 1.f;             // :1022 alpha = integ.a_, then integrator_path_tracer.cc:321-326
		}
		c.level = P;
	}
}
#undef FREC

// Resume a parked path and run it to its next kd-tree query (or to its end).
YG_DEV int wf_advance(const WfArgs &a, uint32_t slot, uint32_t pixel_sample, uint32_t sampling_offs, uint32_t ordinal, const float4 ans, float result[4], int &out_mask)
{
	Ctl c = load_ctl(a, slot);
	// (a record pass's kernel: every park is for a closest hit)
	const bool beside = YAFGPU_FEAT_LIGHTS && c.pc == kPcAfterBoth;        // the shadow pair of a vertex and the next segment's closest hit came back together
	int where = (YAFGPU_FEAT_LIGHTS && (c.pc == kPcAfterShadow || beside)) ? W_AFTER_SHADOW : W_AFTER_CLOSEST;
	// The records the light-estimate bookkeeping passes from step to step (throughput, path colour, the estimate
	// in flight and its accumulators) live in registers for the duration of the advance: a resumed shadow answer
	// loads them in ONE round of loads instead of one dependent round per step (each step used to re-read what the
	// previous one had just stored), and they are written back once, when the path parks.  (Also forwarding the
	// vertex st_after_closest writes to st_dl_eval in registers costs more in spills than the round trip it saves:
	// 6.9 -> 7.7 ms on C2.)
	Hot h; h.valid = 0u; h.dirty = 0u; h.acc_zero = 0u; h.tot_zero = 0u;
#if !YAFGPU_FEAT_LIGHTS
	h.vvalid = 0u;
#endif
	uint32_t w_last = 0u;      // the (li, l_end, is) word of the last pair the path is about to park for (st_beside)
	float alpha = 0.f;
	uint2 verdict = make_uint2(0u, 0u), verdict2 = verdict;
	if(where == W_AFTER_SHADOW)
	{	// bits 4*slot + 2*pair + which of the verdict bit array (set by the any-hit kernel for an occluded ray)
		const uint32_t w = a.verdict[slot >> 3], sh = (slot & 7u) << 2;
		verdict = make_uint2((w >> sh) & 1u, (w >> (sh + 1u)) & 1u);
		verdict2 = make_uint2((w >> (sh + 2u)) & 1u, (w >> (sh + 3u)) & 1u);
		hot_preload(a, slot, h, c.stage != kStPrimary);
	}
	// The step graph has no backward edge except NEXT <-> EVAL, so the program is written out once in topological
	// order (a dispatch loop makes the optimizer thread the transitions, duplicate the steps and keep the union
	// of their registers alive: 163 VGPRs against 81 for the widest single step).
	// ... with one exception.  A path that parked its next segment beside a vertex's LAST shadow pair (st_beside) first closes that
	// estimate — the small steps, a second time, ahead of everything: the answers, the one light still open, the booking — and then
	// runs the vertex steps below for the segment's hit.  (A second turn through the big steps instead, as a loop or by calling the
	// program again, made the optimizer keep the union of their registers alive: 65-90 spilled VGPRs.)
	// a SECOND pair parked with the first (WfArgs::multi): booked where the path would have evaluated it, at its W_DL_EVAL
	int pend2 = (YAFGPU_FEAT_LIGHTS && YAFGPU_FEAT_MULTI) ? c.mask2 : 0;
	c.mask2 = 0;
	if(beside)
	{
		where = st_after_shadow(a, slot, h, verdict);
		if(where == W_DL_NEXT) where = st_dl_next(a, slot, h, c.level);       // closes the light: st_beside only goes ahead at the last pair
		if(YAFGPU_FEAT_MULTI && where == W_DL_EVAL && pend2)
		{	// ... which was the second pair of the park
			where = st_after_shadow(a, slot, h, verdict2, true, pend2);
			pend2 = 0;
			if(where == W_DL_NEXT) where = st_dl_next(a, slot, h, c.level);
		}
		if(where == W_DL_DONE) where = st_dl_done(a, slot, h, c, true);
		if(where == W_NEXT_VERTEX) where = W_AFTER_CLOSEST;
	}
	if(where == W_AFTER_CLOSEST) where = st_after_closest(a, slot, h, c, ordinal, ans, beside);
	if(YAFGPU_FEAT_LIGHTS)
	{
		if(where == W_AFTER_SHADOW) where = st_after_shadow(a, slot, h, verdict);
		bool second = false;      // the pair being evaluated is the one AFTER the pair the path is about to park for
		uint32_t w2 = 0u;
		while(where == W_DL_NEXT || where == W_DL_EVAL)
		{
			if(where == W_DL_NEXT) where = st_dl_next(a, slot, h, c.level);
			else if(YAFGPU_FEAT_MULTI && pend2) { where = st_after_shadow(a, slot, h, verdict2, true, pend2); pend2 = 0; }
			else
			{
				int m = 0;
				const int r = st_dl_eval(a, slot, h, c, pixel_sample, sampling_offs, m, second, w2);
				if(!second)
				{
					where = r;
					if(r == W_PARK_SHADOW)
					{
						out_mask = m;
						w_last = ubits(HGET(14).w);
						if(YAFGPU_FEAT_MULTI && a.multi)
						{	// the pair st_dl_next would name after this one: the light's next sample, or the next light's first
							const int li = (int)(w_last & 0xffu), l_end = (int)((w_last >> 8) & 0xffu), is = (int)(w_last >> 20);
							const yafgpu_light &light = a.ra.sc.lights[li];
							const int n = light.type == YAFGPU_LIGHT_POINT ? 1 : dl_area_samples(a.ra, light, wf_div(a, slot, c.level).division);
							const bool same = is + 1 < n;
							if(same || li + 1 < l_end) { second = true; w2 = pack_dlc(same ? li : li + 1, l_end, 0, same ? is + 1 : 0); where = W_DL_EVAL; }
						}
					}
				}
				else
				{
					if(r == W_PARK_SHADOW) { c.mask2 = m; out_mask |= m << 2; w_last = w2; }
					where = W_PARK_SHADOW;
				}
			}
		}
	}
	if(where == W_DL_DONE) where = st_dl_done(a, slot, h, c, false);
	if(YAFGPU_FEAT_LIGHTS && where == W_PARK_SHADOW) where = st_beside(a, slot, h, c, pixel_sample, sampling_offs, ordinal, w_last);
	if(where == W_EXTEND) where = st_extend(a, slot, h, c);
	if(where == W_START_PATH) where = st_start_path(a, slot, h, c, pixel_sample, sampling_offs, ordinal);
	if(where == W_RECURSE) where = st_recurse(a, slot, c);
#if YAFGPU_FEAT_RECURSE
	// recursion: return -> (next glossy trajectory | the specular branch of the level above) -> park or return again
	for(;;)
	{
		if(where == W_GLOSSY_NEXT) where = st_glossy_next(a, slot, c, pixel_sample, sampling_offs);
		if(where == W_RECURSE_SPEC) where = st_recurse_spec(a, slot, c);
		if(where != W_RETURN) break;
		where = st_return(a, slot, c, alpha);
		if(where != W_GLOSSY_NEXT && where != W_RECURSE_SPEC) break;
	}
#else
	if(where == W_RETURN) where = st_return(a, slot, c, alpha);
#endif
	if(where != W_FINISH) hot_flush(a, slot, h, where == W_PARK_CLOSEST);      // a path that ends needs none of them again
	if(where == W_PARK_CLOSEST) { c.pc = kPcAfterClosest; REC(13) = f4(c.col, fbits(pack_ctl(c))); return kReqClosest; }
	if(where == W_PARK_SHADOW) { c.pc = kPcAfterShadow; REC(13) = f4(c.col, fbits(pack_ctl(c))); return kReqShadow; }
	if(where == W_PARK_BOTH) { c.pc = kPcAfterBoth; REC(13) = f4(c.col, fbits(pack_ctl(c))); return kReqBoth; }
	// W_FINISH
	if(a.ra.rp.bg_transp) alpha = smax(alpha, 0.f);   // EmptyVolumeIntegrator: transmittance 1 (integrator_empty_volume.cc:32-38)
	result[0] = c.col.r; result[1] = c.col.g; result[2] = c.col.b; result[3] = alpha;
	return kReqDone;
}
#ifdef YAFGPU_STEP_PROBE
// compile-time probe (not built by default): register footprint of each step in isolation and of the whole program —
//   hipcc ... -DYAFGPU_STEP_PROBE -Rpass-analysis=kernel-resource-usage
#ifndef PROBE_WAVES
#define PROBE_WAVES 1
#endif
__global__ __launch_bounds__(kBlock, PROBE_WAVES) void probe_advance(const WfArgs a, int *out)
{
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; float r[4] = {0.f, 0.f, 0.f, 0.f}; int m = 0;
	out[slot] = wf_advance(a, slot, slot * 3u, slot * 5u, slot * 7u, a.state[2 * (size_t)a.cap + slot], r, m) + m + (int)r[0] + (int)r[3];
}
#define PROBE(name, call) __global__ __launch_bounds__(kBlock, PROBE_WAVES) void name(const WfArgs a, int *out) { \
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; Ctl c = load_ctl(a, slot); int m = 0; (void)m; \
	Hot h; hot_preload(a, slot, h, true); const int w = call; hot_flush(a, slot, h, false); \
	a.state[(size_t)13 * a.cap + slot] = f4(c.col, fbits(pack_ctl(c))); out[slot] = w + m; }
PROBE(probe_after_closest, st_after_closest(a, slot, h, c, slot * 7u, a.state[2 * (size_t)a.cap + slot], false))
PROBE(probe_after_shadow, st_after_shadow(a, slot, h, make_uint2(slot & 1u, slot & 2u)))
PROBE(probe_dl_next, st_dl_next(a, slot, h, c.level))
PROBE(probe_dl_eval, st_dl_eval(a, slot, h, c, slot * 3u, slot * 5u, m))
PROBE(probe_dl_done, st_dl_done(a, slot, h, c, false))
PROBE(probe_extend, st_extend(a, slot, h, c))
PROBE(probe_start_path, st_start_path(a, slot, h, c, slot * 3u, slot * 5u, slot * 7u))
#undef PROBE
#endif

#undef REC

// identity of a path slot: slot = pixel_local * spp + sample
template<bool kTable>
YG_DEV void wf_identity(const WfArgs &a, uint32_t slot, int &px, int &py, int &sample, uint32_t &pixel_sample, uint32_t &sampling_offs, uint32_t &ordinal)
{
	const yafgpu_render_params &rp = a.ra.rp;
	const uint32_t spp = (uint32_t)rp.aa_minsamples;
	const uint32_t pixel_local = slot / spp;
	sample = (int)(slot - pixel_local * spp);
	if(kTable) { const uint32_t xy = a.pix_xy[pixel_local]; px = (int)(xy & 0xffffu); py = (int)(xy >> 16); }
	else wf_pixel_of(a, pixel_local, px, py);
	sampling_offs = fnv32a((uint32_t)py * fnv32a((uint32_t)px));
	pixel_sample = rp.base_sampling_offset + rp.pass_offset + (uint32_t)sample;     // pass_offs + sample, integrator_tiled.cc:389
	ordinal = ((uint32_t)(py - rp.ystart) * (uint32_t)rp.width + (uint32_t)(px - rp.xstart)) * spp + (uint32_t)sample;
}

YG_DEV void wf_sample_offsets(const WfArgs &a, int sample, uint32_t sampling_offs, float &dx, float &dy)
{
	const int n_samples = a.ra.rp.aa_minsamples;
	dx = 0.5f; dy = 0.5f;
	if(a.ra.rp.multi_pass)
	{	// integrator_tiled.cc:394-398: scrambled van der Corput / Sobol, good for any total sample count
		const uint32_t pixel_sample = a.ra.rp.base_sampling_offset + a.ra.rp.pass_offset + (uint32_t)sample;
		dx = ri_vdc(pixel_sample, sampling_offs);
		dy = ri_s(pixel_sample, sampling_offs);
	}
	else if(n_samples > 1)
	{
		const float d_1 = (float)(1.0 / (double)(float)n_samples);
		dx = (float)((0.5 + (double)(float)sample) * (double)d_1);
		dy = ri_lp((uint32_t)sample + sampling_offs, 0u);
	}
}

// append slot to a queue with one atomic per wave
YG_DEV void wf_push(uint32_t *queue, uint32_t *count, bool pred, uint32_t slot)
{
	const unsigned long long mask = __ballot(pred);
	if(mask == 0ull) return;
	const int lane = (int)(threadIdx.x & (kWave - 1));
	const int leader = __ffsll((long long)mask) - 1;
	uint32_t base = 0u;
	if(lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
	base = (uint32_t)__shfl((int)base, leader, kWave);
	if(pred) queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

#ifndef YAFGPU_VARIANT_TU      // a shade-kernel variant unit compiles wf_shade only
// camera rays: TiledIntegrator::renderTile :378-410
__global__ __launch_bounds__(kBlock) void wf_generate(const WfArgs a)
{
	for(uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < a.n_paths; slot += gridDim.x * blockDim.x)
	{
		int px, py, sample; uint32_t pixel_sample, sampling_offs, ordinal;
		if(a.pix_listed) wf_identity<true>(a, slot, px, py, sample, pixel_sample, sampling_offs, ordinal);
		else
		{
			wf_identity<false>(a, slot, px, py, sample, pixel_sample, sampling_offs, ordinal);
			if(sample == 0) a.pix_xy[slot / (uint32_t)a.ra.rp.aa_minsamples] = (uint32_t)px | ((uint32_t)py << 16);
		}
		float dx, dy;
		wf_sample_offsets(a, sample, sampling_offs, dx, dy);
		V3 from, dir; float tmin, tmax;
		float lens_u = 0.5f, lens_v = 0.5f;
		if(a.ra.sc.cam.aperture != 0.f)
		{	// integrator_tiled.cc:382-383,405-409: Halton(3) / Halton(5) started at pass_offs + sampling_offs per pixel and
			// advanced once per sample; the incremental sequence is replayed up to this sample (its roundings are its own)
			Halton hal_u, hal_v;
			hal_u.init(3u); hal_v.init(5u);
			const uint32_t start = a.ra.rp.base_sampling_offset + a.ra.rp.pass_offset + sampling_offs;
			hal_u.set_start(start); hal_v.set_start(start);
			for(int k = 0; k <= sample; ++k) { lens_u = hal_u.next(); lens_v = hal_v.next(); }
		}
		camera_shoot(a.ra.sc.cam, (float)px + dx, (float)py + dy, lens_u, lens_v, from, dir, tmin, tmax);
		float4 *b = a.state + slot; const size_t c = a.cap;
		b[0 * c] = f4(from, tmin);
		b[1 * c] = f4(dir, tmax);
		b[13 * c] = make_float4(0.f, 0.f, 0.f, fbits((uint32_t)kPcAfterClosest | ((uint32_t)kStPrimary << 2)));
	}
	if(blockIdx.x == 0 && threadIdx.x == 0) { a.cnt_in[0] = a.n_paths; a.cnt_in[1] = 0u; a.cnt_in[2] = 0u; a.cnt_in[3] = 0u; a.cnt_in[4] = 0u; }
}

// The traversal kernels: Scene::intersect (scene.cc:896-927) / Scene::isShadowed (:962-994) over a queue.
//
// Persistent waves with ray refill: a lane whose ray is finished does not idle until the slowest ray
// of its wave ends; when at least kRefill lanes are free the wave fetches that many new rays from the
// queue with one atomic (ballot + prefix rank) and the freed lanes start them while the others carry
// on.  Traversal state (current node, [tmin,tmax], best hit, short stack in LDS) is per lane, so lanes
// of one wave can be at any point of any ray.  The walk itself is kd_trace's, cut at leaf granularity.
#ifndef YAFGPU_NODE_WINDOW
#define YAFGPU_NODE_WINDOW 1   // C2 sweep: 1 -> 22.1, 2 -> 23.7, 4 -> 27.2, 8 -> 37.0 ms of traversal per pass: extra node fetches cost more than the latency they hide
#endif
#ifndef YAFGPU_VOTE_NUM
#define YAFGPU_VOTE_NUM 1      // triangle round when n_tri * NUM >= n_node * DEN
#endif
#ifndef YAFGPU_VOTE_DEN
#define YAFGPU_VOTE_DEN 1
#endif
#ifndef YAFGPU_NODE_BURST
#define YAFGPU_NODE_BURST 4    // node steps per vote
#endif
#ifndef YAFGPU_TRACE_BATCH
#define YAFGPU_TRACE_BATCH 512
#endif
constexpr int kTraceBatch = YAFGPU_TRACE_BATCH;
#ifndef YAFGPU_REFILL
#define YAFGPU_REFILL 24               // C2 sweep (voted rounds): 8..32 within 2 %, 48 -> -5 %, 56 -> -12 %
#endif
// Postponed leaves.  A lane that reaches a non-empty leaf does not stop there: it notes the leaf as PENDING (its
// reference range and the exit distance of its cell) and walks on at once, as if the leaf held no terminating hit; only
// at a second non-empty leaf does it wait.  A lane then usually has both kinds of work on offer — node steps of the
// walk ahead and triangle tests of the pending leaf — so whichever kind the wave votes for, more lanes take part (the
// voted rounds alone left half of the lanes idle in every round: PMC lane utilisation 38-44 %).  Nothing about the
// answer changes: the triangles tested, and their order per ray, are those of TriKdTree::intersect — the pending leaf
// is always tested before a later one is even noted, a hit inside its cell ends the ray exactly where :822 does, and
// the walk ahead is thrown away then (its node steps are the price: counted apart in the stats build).  A walk may end
// early on the hit known so far (`hit && z <= tmax`, `z < tmin`): z only shrinks, so the sequential walk ends there too.
#ifndef YAFGPU_TRACE_POSTPONE
#define YAFGPU_TRACE_POSTPONE 1
#endif
#ifndef YAFGPU_TRACE_FUSED
#define YAFGPU_TRACE_FUSED 0
#endif
#ifndef YAFGPU_TRACE_BELOW_MASKS
#define YAFGPU_TRACE_BELOW_MASKS 1
#endif
#ifndef YAFGPU_TRACE_TRIPF
#define YAFGPU_TRACE_TRIPF 0     // measured, off: the pending leaf's next triangle record fetched a round ahead (10 more VGPRs, loads for tests that never run, a wait for the reference where the fetch is issued): 2035 against 2245 Mrays/s on the 1 M-triangle scene
#endif
#ifndef YAFGPU_TRACE_PAIR
#define YAFGPU_TRACE_PAIR 1      // closest-hit launches -6.5 % on the 1 M-triangle scenes, -4.5 % on the 100 k one (pair4 in profiles/r02_ab_pair.txt)
#endif
#ifndef YAFGPU_TRACE_BLOCKS
#define YAFGPU_TRACE_BLOCKS 0    // measured, off: 1 = the any-hit launches walk the block layout of the tree (DevScene::nodes_blk): +4 %; 2 = the closest-hit ones too, instead of the pair array: +10 % (profiles/r02_ab_blocks.txt)
#endif
// Leaves apart.  A node burst used to handle a leaf the moment a lane fetched one: the empty-leaf pop (21 vector instructions) and
// the non-empty leaf's note-and-go-on (30) sat as divergent branches inside EVERY node step, entered by the 15 % of the walking
// lanes that stood at a leaf — with 35 lanes walking some lane nearly always does, so a step cost 28 + 21 + 30 instructions for
// 28 useful ones.  Now a lane that fetches a leaf stands still for the rest of the burst (kAtLeaf) and all leaves of a burst are
// handled together after it, once.  Per ray the steps and their order are what they were.  0: the old placement (comparison build).
#ifndef YAFGPU_TRACE_LEAF_APART
#define YAFGPU_TRACE_LEAF_APART 0      // measured (profiles/r03_ab_leaf.txt): fewer instructions, fewer lanes per step — closest-hit launches +6 %, any-hit launches equal
#endif
// Triangle records (48 MB at 1 M triangles, each read a few times per pass by unrelated rays) loaded with the non-temporal hint, so that
// they do not push the node lines out of the 32 KB L1 / 4 MB L2 (measured: profiles/r03_ab_toptree.txt).
#ifndef YAFGPU_TRACE_NT
#define YAFGPU_TRACE_NT 0
#endif
typedef float yg_f4v __attribute__((ext_vector_type(4)));
YG_DEV float4 ld_tri(const float4 *p)
{
#if YAFGPU_TRACE_NT
	const yg_f4v v = __builtin_nontemporal_load((const yg_f4v *)p);
	return make_float4(v.x, v.y, v.z, v.w);
#else
	return *p;
#endif
}
#ifndef YAFGPU_TRACE_WAVES
#define YAFGPU_TRACE_WAVES 7     // waves per SIMD the register allocation must leave room for (22.5 KB of LDS per block allow 7): 70 / 68 VGPRs; without the bound the any-hit kernel took 81 (5 waves)
#endif
template<bool kAny, bool kStats>
__global__ __launch_bounds__(kBlock, YAFGPU_TRACE_WAVES) void wf_trace(const WfArgs a)
{
	__shared__ uint2 s_stack[kWavesPerBlock][kStack][kWave];
	// (origin, 1/direction) of the lane's ray per axis: a node step fetches the pair of its split axis with one
	// ds_read_b64 instead of selecting it out of six registers (the kernel is VALU-bound, the LDS port is idle)
	__shared__ float2 s_axis[kWavesPerBlock][3][kWave];
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	const DevScene &sc = a.ra.sc;
#if YAFGPU_TRACE_TOP == 2
	__shared__ uint4 s_top[kTopN];
	for(int i = (int)threadIdx.x; i < kTopN; i += (int)blockDim.x) s_top[i] = sc.top[i];
	__syncthreads();
#endif
#ifdef YAFGPU_TRACE_LDS_PAD      // (A/B aid: the LDS the staged top would take, without it — the occupancy alone)
	__shared__ uint32_t s_pad[YAFGPU_TRACE_LDS_PAD / 4];
	if(a.cap == 0xffffffffu) s_pad[threadIdx.x] = 1u;
#endif
	constexpr uint32_t kRoot = YAFGPU_TRACE_TOP ? kTopTag : 0u;
	LaneStack stk;
	stk.col = &s_stack[wave][0][lane];
	float2 *const axis_col = &s_axis[wave][0][lane];      // axis k at axis_col[k * kWave]
	LaneCounters cn = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
	const uint32_t n = kAny ? a.cnt_in[1] : a.cnt_in[0];
	uint32_t *cursor = kAny ? &a.cnt_in[3] : &a.cnt_in[2];
	const uint32_t *q = kAny ? a.q_shadow_in : a.q_closest_in;
	const size_t c = a.cap;
	bool exhausted = (n == 0u);
	uint32_t slot = 0u, node = 0u, which = 0u, pair = 0u, qi = 0u;      // qi: the ray's position in the queue (where a closest-hit answer goes)
	uint32_t w_next = 0u, w_end = 0u;      // this wave's reserved queue range (wave-uniform)
	// reservation size: large enough to keep the counter word off the critical path, small enough that a short queue still spreads over all waves
	const uint32_t batch = min((uint32_t)kTraceBatch, max((uint32_t)kWave, (n / (gridDim.x * (uint32_t)kWavesPerBlock * 2u)) & ~63u));
	V3 from = mk(0.f, 0.f, 0.f), dir = from;
	uint32_t dneg = 0u;                    // bit k: direction component k <= 0 (the tie rule at o == split)
	float ray_tmin = 0.f, dist = 0.f, t_exit = 0.f, tmin = 0.f, tmax = 0.f, z = 0.f, bu = 0.f, bv = 0.f;
	int tri = -1; bool hit = false;
	// Answers.  Any-hit: one BIT per ray in a zeroed array, set only for occluded rays (a 4-byte word per ray scattered by
	// path cost 9x the verdicts' size in HBM writes).  Closest-hit: 16 B at the ray's QUEUE position — waves own contiguous
	// queue ranges, so the stores of a wave fall into a few lines that complete while still in L2 (by path they were 2.4x).
	auto answer_any = [&](bool occluded) {
		const uint32_t bit = 4u * slot + 2u * pair + which;
		if(occluded) atomicOr(&a.verdict[bit >> 5], 1u << (bit & 31u));
	};
	// the walk: at a node | at a non-empty leaf, waiting for the pending slot | no node left | no ray
	enum : uint32_t { kWalk = 0u, kBlocked = 1u, kWalkEnd = 2u, kNoRay = 3u, kAtLeaf = 4u };
	uint32_t lf_first = 0u, lf_np = 0u;              // the leaf a lane in kAtLeaf stands at: first reference, reference count
	constexpr int kVoteNum = YAFGPU_VOTE_NUM, kVoteDen = YAFGPU_VOTE_DEN, kNodeBurst = YAFGPU_NODE_BURST;
	constexpr bool kBlk = (YAFGPU_TRACE_BLOCKS == 2) || (YAFGPU_TRACE_BLOCKS == 1 && kAny);
	uint32_t ws = kNoRay;
	uint32_t p_cur = 0u, p_end = 0u, ti = 0u;       // pending leaf: references [p_cur, p_end) still to test, ti = refs[p_cur] (in flight)
	float p_tmax = 0.f;                              // exit distance of the pending leaf's cell
#if YAFGPU_TRACE_TRIPF
	float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;      // record of triangle `ti`, fetched ahead of its test
	bool q_ok = false;
#endif
	bool done = false;
	uint32_t spec = 0u;                              // kStats: node steps + leaves of the walk ahead of the pending leaf
	uint32_t spec_leaves = 0u;
	uint32_t rounds_node = 0u, rounds_tri = 0u;     // wave-uniform (kStats)
	for(;;)
	{
		const unsigned long long idle = __ballot(ws == kNoRay);
		const int n_idle = __popcll(idle);
		if(!exhausted && (n_idle >= YAFGPU_REFILL || n_idle == kWave))
		{
			// the wave owns [w_next, w_end) of the queue; one atomic reserves kTraceBatch entries at a time (a single
			// counter word sustains only ~90 returning atomics per microsecond: MI355X_MICROARCH.md, row "dequeue")
			if(w_next >= w_end)
			{
				const int leader = __ffsll((long long)idle) - 1;
				uint32_t base = 0u;
				if(lane == leader) base = atomicAdd(cursor, batch);
				base = (uint32_t)__shfl((int)base, leader, kWave);
				w_next = base; w_end = min(base + batch, n);
				if(base >= n) { exhausted = true; w_next = w_end = 0u; }
			}
			const uint32_t avail = w_end - w_next;
			const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			const uint32_t first_i = w_next;
			w_next += min(avail, (uint32_t)n_idle);
			if(ws == kNoRay && rank < avail)
			{
				const uint32_t i = first_i + rank;
				slot = q ? q[i] : i;
				qi = i;
				which = slot >> 31;                            // any-hit: the second ray of a pair; closest-hit: a segment parked beside a shadow pair
				pair = kAny ? (slot >> 30) & 1u : 0u;          // any-hit: the second pair of a park (records 28..31)
				slot &= kAny ? 0x3fffffffu : 0x7fffffffu;
				float4 r0 = a.state[slot], r1 = a.state[c + slot];
				if(!kAny && which)
				{	// origin: the vertex in record 0; direction and tmin in record 26 (records 0.w / 1 hold the pair's first shadow ray)
					const float4 r26 = a.state[26 * c + slot];
					r1 = make_float4(r26.x, r26.y, r26.z, -1.f);
					r0.w = r26.w;
				}
				if(kAny && pair)
				{	// the second pair of the park: the same origin; 28 direction | tmax and 29.w tmin of its first ray, 30 direction | tmin and 31.w tmax of its second
					const float4 ra_ = a.state[(which ? 30 : 28) * c + slot];
					const float rb_ = a.state[(which ? 31 : 29) * c + slot].w;
					r1 = make_float4(ra_.x, ra_.y, ra_.z, which ? rb_ : ra_.w);
					r0.w = which ? ra_.w : rb_;
				}
				else if(kAny && which)
				{	// second ray of the pair: same origin, direction/tmin in r20, tmax in r21.w
					const float4 r20 = a.state[20 * c + slot];
					r1 = make_float4(r20.x, r20.y, r20.z, a.state[21 * c + slot].w);
					r0.w = r20.w;
				}
				from = v3(r0); dir = v3(r1);
				if(kAny)
				{
					from = from + dir * r0.w;
					dist = (r1.w < 0.f) ? INFINITY : r1.w - 2.f * r0.w;
					// intersectS accepts hits from 0 on, intersectTs (transpShad) from the ray's tmin_ on — measured from
					// the origin already moved by tmin_ (kdtree_triangle.cc:936 / :1099, scene.cc isShadowed)
					ray_tmin = a.ra.rp.transp_shad ? r0.w : 0.f;
					++cn.shadow;
				}
				else
				{
					dist = (r1.w < 0.f) ? INFINITY : r1.w;
					ray_tmin = r0.w;
					++cn.closest;
				}
				float ea, eb;
				tri = -1; hit = false; z = dist; bu = 0.f; bv = 0.f; done = false; p_cur = p_end = 0u;
				const V3 inv_dir = mk(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
				if(sc.n_nodes != 0u && bound_cross(sc, from, dir, inv_dir, dist, ea, eb) && !(dist < smax(ea, 0.f)))   // :717 on entry
				{
					axis_col[0] = make_float2(from.x, inv_dir.x);
					axis_col[kWave] = make_float2(from.y, inv_dir.y);
					axis_col[2 * kWave] = make_float2(from.z, inv_dir.z);
					dneg = (dir.x <= 0.f ? 1u : 0u) | (dir.y <= 0.f ? 2u : 0u) | (dir.z <= 0.f ? 4u : 0u);
					t_exit = eb; tmin = smax(ea, 0.f); tmax = t_exit; node = kRoot; ws = kWalk;
					stk.reset();
				}
				else
				{	// misses the scene bound: answer at once
					if(!kAny) a.state[2 * c + qi] = make_float4(fbits(0xffffffffu), dist, 0.f, 0.f);
				}
			}
		}
		const unsigned long long m_act = __ballot(ws != kNoRay);
		if(m_act == 0ull) { if(exhausted) break; else continue; }
		// The kernel is bound by instruction issue, not by memory, so what counts is how many lanes share each
		// instruction.  Each round the WAVE does one kind of work, chosen by vote: a round of triangle tests (lanes with a
		// pending leaf) when at least as many lanes can take part in it as in node steps, else a burst of node steps (lanes
		// whose walk stands at a node).  Per ray the steps and their order are kd_trace's.
		const bool has_pend = p_cur < p_end;
		const unsigned long long m_tri = __ballot(has_pend);
		const int n_tri = __popcll(m_tri), n_node = __popcll(__ballot(ws == kWalk));
		// what follows a leaf in the walk (kdtree_triangle.cc:822-835 / :936-960): the nearest pending far child, or a restart
		// at the cell exit if the short stack lost it, or the end.  Written with selects: the wave pays for every branch
		// some lane takes.  `hit && z <= tmax` on the hit known so far ends the walk (not the ray: a pending leaf is still
		// tested); the leaf's own hits are looked at when its tests are through (see the triangle round).
		auto leaf_end = [&]() {
			const uint2 top = stk.col[((stk.sp - 1) & (kStack - 1)) * kWave];        // garbage when empty: unused then
			const bool hit_here = !kAny && hit && z <= tmax;
			const bool emp = stk.empty();
			const bool restart = emp && stk.lost() && !(tmax >= t_exit);
			if(kStats && restart && !hit_here) ++cn.restarts;
			tmin = restart ? restart_from(tmin, tmax) : tmax;                          // see restart_from: progress on degenerate trees
			node = emp ? kRoot : top.x;
			tmax = emp ? t_exit : __uint_as_float(top.y);
			stk.sp = emp ? 0 : stk.sp - 1;
			stk.lo = emp ? 0 : stk.lo;
			const bool fin = hit_here || (emp && !restart) || z < tmin;                // z < tmin: :717
			ws = fin ? kWalkEnd : kWalk;
		};
		// one triangle test of the pending leaf, on the fetched record (r0, r1, r2) and the next reference
		auto tri_step = [&](const float4 r0, const float4 r1, const float4 r2, const uint32_t ref_v) {
			float t, u, v;
			if(kStats) ++cn.tests;
			// Triangle::intersect without its early returns: the same operations in the same order, every lane to the
			// end (a wave of 30 rays almost never leaves early as a whole), the rejections folded into one predicate
			const bool ok = tri_test_flat(r0, r1, r2, from, dir, t, u, v);
			const uint32_t vis = __float_as_uint(r1.w) >> 30;
			if(kAny)
			{
				const bool found = ok && t < dist && t >= ray_tmin && (vis == 0u || vis == 2u);
				hit = hit || found; done = done || found;
			}
			else
			{
				const bool better = ok && t < z && t >= ray_tmin && (vis == 0u || vis == 1u);
				z = better ? t : z; tri = better ? (int)ti : tri; bu = better ? u : bu; bv = better ? v : bv; hit = hit || better;
			}
			++p_cur; ti = ref_v;
			if(p_cur >= p_end)
			{	// the leaf is through: :822 (a hit inside its cell ends the ray), else the walk ahead stands
				if(YAFGPU_TRACE_POSTPONE)
				{
					const bool ends = !kAny && hit && z <= p_tmax;
					done = done || ends;
					if(kStats) { if(done) { cn.interior -= spec; cn.leaves -= spec_leaves; } spec = 0u; spec_leaves = 0u; }
					// the walk ahead was led by the hit known then; it may be over by what this leaf found (:717)
					if(!kAny && ws != kWalkEnd && z < tmin) ws = kWalkEnd;
					ws = (ws == kBlocked) ? kWalk : ws;      // a leaf it waited at can be noted now
				}
				else if(!done) leaf_end();                  // (comparison build) the lane stood at the leaf: go on from it
			}
		};
		// one step of the walk on the fetched node
		auto node_step = [&](const uint2 nd, const uint32_t left_c, const uint32_t right_c) {
			if((nd.y & 3u) != 3u)
			{
				const uint32_t axis = nd.y & 3u;
				const float split = __uint_as_float(nd.x);
				const float2 oi = axis_col[axis * kWave];
				const float o = oi.x;
				const float tplane = (split - o) * oi.y;
				// (o < split) || (o == split && d <= 0), kdtree_triangle.cc:725-760, without the short-circuit branches
				const bool dn = ((dneg >> axis) & 1u) != 0u;
#if YAFGPU_TRACE_BELOW_MASKS
				// the same predicate as lane-mask logic: three compares, combined on the scalar unit (the kernel is VALU-bound)
				const bool below = __builtin_amdgcn_inverse_ballot_w64(__ballot(o < split) | (__ballot(o == split) & __ballot(dn)));
#else
				const bool below = dn ? (o <= split) : (o < split);
#endif
				uint32_t left = left_c, right = right_c;
				if(kBlk)
				{	// block layout: children inside the block for slots 0..2, the roots of the two child blocks for slots 3..6
					const uint32_t sl = node & 7u;
					const bool inside = sl < 3u;
					left = inside ? node + sl + 1u : (nd.y >> 2) << 3;
					right = inside ? left + 1u : left + 8u;
				}
				const uint32_t near_c = below ? left : right, far_c = below ? right : left;
				if(kStats) { ++cn.interior; if(p_cur < p_end) ++spec; }
				const bool near_only = !(tplane <= tmax) || tplane <= 0.f;        // plane beyond the cell or behind the origin (also NaN)
				const bool far_only = !near_only && tplane < tmin;
				const bool both = !near_only && !far_only;
				// the slot above the top is always free (at most kStack-1 live entries), so the far child is written
				// unconditionally and only the stack pointer says whether it was a push
				stk.col[(stk.sp & (kStack - 1)) * kWave] = make_uint2(far_c, __float_as_uint(tmax));
				stk.sp += both ? 1 : 0;
				stk.lo = max(stk.lo, stk.sp - (kStack - 1));
				node = far_only ? far_c : near_c;
				tmax = both ? tplane : tmax;
			}
			else if(YAFGPU_TRACE_LEAF_APART) { lf_first = nd.x; lf_np = nd.y >> 2; ws = kAtLeaf; }      // handled after the burst (leaf_visit)
			else
			{
				const uint32_t np = nd.y >> 2;
				if(np == 0u) { if(kStats) { ++cn.leaves; if(p_cur < p_end) ++spec_leaves; } leaf_end(); }
				else if(!YAFGPU_TRACE_POSTPONE)
				{	// (comparison build) stop at every non-empty leaf until its tests are through
					if(kStats) ++cn.leaves;
					p_cur = nd.x; p_end = nd.x + np; p_tmax = tmax; ti = sc.refs[nd.x]; ws = kBlocked;
				}
				else if(p_cur < p_end) ws = kBlocked;          // a second non-empty leaf: wait for the pending one (the node is read again then)
				else
				{
					if(kStats) ++cn.leaves;
					p_cur = nd.x; p_end = nd.x + np; p_tmax = tmax;
					ti = sc.refs[nd.x];                           // in flight while the lane walks on
					leaf_end();
				}
			}
		};
		// one step on the heap-ordered copy of the tree's top (YAFGPU_TRACE_TOP): children by index arithmetic, and by the node's own
		// place in the depth-first array on the copy's last level, below which the walk goes on in `nodes`
		auto top_step = [&]() {
#if YAFGPU_TRACE_TOP
			const uint32_t hh = node & ~kTopTag;
#if YAFGPU_TRACE_TOP == 2
			const uint4 e = s_top[hh];
#else
			const uint4 e = sc.top[hh];
#endif
			const bool bottom = hh >= (uint32_t)kTopBottom;
			node_step(make_uint2(e.x, e.y), bottom ? e.z + 1u : (kTopTag | (2u * hh + 1u)), bottom ? (e.y >> 2) : (kTopTag | (2u * hh + 2u)));
#endif
		};
		// the leaves of a burst, together (YAFGPU_TRACE_LEAF_APART): an empty leaf is left at once, a non-empty one becomes the pending
		// leaf (its first reference in flight while the lane walks on) — or the lane waits at it while another leaf is pending
		auto leaf_visit = [&]() {
			const bool empty = lf_np == 0u, pend = p_cur < p_end;
			if(!YAFGPU_TRACE_POSTPONE && !empty)
			{	// (comparison build) stop at every non-empty leaf until its tests are through
				if(kStats) ++cn.leaves;
				p_cur = lf_first; p_end = lf_first + lf_np; p_tmax = tmax; ti = sc.refs[lf_first]; ws = kBlocked;
			}
			else if(!empty && pend) ws = kBlocked;           // the node is read again when the pending leaf is through
			else
			{
				if(kStats) { ++cn.leaves; if(empty && pend) ++spec_leaves; }
				if(!empty)
				{
					p_cur = lf_first; p_end = lf_first + lf_np; p_tmax = tmax;
					ti = sc.refs[lf_first];
				}
				leaf_end();
			}
		};
#if YAFGPU_TRACE_TOP
		// lanes that stand in the top copy (new rays; a far child popped off the stack) walk down it first, together: these steps wait for
		// the L1 or LDS only, and afterwards the node rounds below find (nearly) every lane in the depth-first array
#pragma unroll 1
		for(int s = 0; s < kTopDepth + 2; ++s)
		{
			if(__ballot(ws == kWalk && (node & kTopTag)) == 0ull) break;
			if(ws == kWalk && (node & kTopTag)) top_step();
			if(YAFGPU_TRACE_LEAF_APART && ws == kAtLeaf) leaf_visit();
		}
#endif
#if YAFGPU_TRACE_FUSED
		// Fused rounds: every lane fetches what it can use — the node its walk stands at AND the next triangle of its pending
		// leaf — the wave waits once, then runs the node section and the triangle section one after the other.  More
		// instructions per round than a voted round (both sections are always issued), half as many memory round trips per
		// ray: the better trade when the tree does not fit the L2s and waves spend most of their time in s_waitcnt
		// (1M triangles: PMC SQ_WAIT_ANY 65 % of wave cycles, L2 hit rate 75-83 %).
		{
			const bool walking = ws == kWalk;
			uint2 nd = make_uint2(0u, 3u);
			float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
			uint32_t ref_v = 0u;
			if(walking) nd = sc.nodes[node];
			if(has_pend)
			{
				if(p_cur + 1u < p_end) ref_v = sc.refs[p_cur + 1u];
				r0 = sc.tri[3u * ti]; r1 = sc.tri[3u * ti + 1u]; r2 = sc.tri[3u * ti + 2u];
			}
			if(kStats) { ++rounds_node; ++rounds_tri; }
			if(walking) node_step(nd, node + 1u, nd.y >> 2);
			if(YAFGPU_TRACE_LEAF_APART && ws == kAtLeaf) leaf_visit();
			if(has_pend) tri_step(r0, r1, r2, ref_v);
			(void)n_tri; (void)n_node;
		}
#else
		if(n_tri * kVoteNum >= n_node * kVoteDen && n_tri > 0)
		{
			if(kStats) ++rounds_tri;
			if(has_pend)
			{
				uint32_t ref_v = 0u;
				if(p_cur + 1u < p_end) ref_v = sc.refs[p_cur + 1u];
#if YAFGPU_TRACE_TRIPF
				if(!q_ok) { q0 = sc.tri[3u * ti]; q1 = sc.tri[3u * ti + 1u]; q2 = sc.tri[3u * ti + 2u]; }
				tri_step(q0, q1, q2, ref_v);
				// the next triangle of the leaf (its reference has arrived meanwhile): its record is in flight until the next
				// triangle round — behind whatever the wave waits for in between
				q_ok = p_cur < p_end;
				if(q_ok) { q0 = sc.tri[3u * ti]; q1 = sc.tri[3u * ti + 1u]; q2 = sc.tri[3u * ti + 2u]; }
#else
				const float4 r0 = ld_tri(&sc.tri[3u * ti]), r1 = ld_tri(&sc.tri[3u * ti + 1u]), r2 = ld_tri(&sc.tri[3u * ti + 2u]);
				tri_step(r0, r1, r2, ref_v);
#endif
			}
		}
		else
		{
#if YAFGPU_TRACE_PAIR
			// Two node steps per memory round trip: nodes2[i] holds node i and a copy of its right child, the left child is node
			// i + 1, so one 32-byte fetch at i brings both children along and the step after an interior node needs no fetch.
			// Closest-hit rays only: the doubled node footprint costs the any-hit rays (L2 hit rate 75 %) more than the saved
			// round trips give them (1 M triangles: closest-hit launches -7 %, any-hit launches +8.5 %).
#pragma unroll 1
			for(int s = 0; s < ((kAny || kBlk) ? 0 : kNodeBurst / 2); ++s)
			{
				if(kStats && __ballot(ws == kWalk) != 0ull) ++rounds_node;
#if YAFGPU_TRACE_TRIPF
				if(p_cur < p_end && !q_ok) { q0 = sc.tri[3u * ti]; q1 = sc.tri[3u * ti + 1u]; q2 = sc.tri[3u * ti + 2u]; q_ok = true; }
#endif
				if(YAFGPU_TRACE_TOP && ws == kWalk && (node & kTopTag)) { top_step(); if(ws == kWalk && (node & kTopTag)) top_step(); }
				else if(ws == kWalk)
				{
					const uint32_t n0 = node;
					const uint4 a0 = sc.nodes2[n0], a1 = sc.nodes2[n0 + 1u];
					node_step(make_uint2(a0.x, a0.y), n0 + 1u, a0.y >> 2);
					if((a0.y & 3u) != 3u)          // an interior node hands the walk to one of its two children
					{
						const bool to_left = node == n0 + 1u;
						const uint2 n1 = to_left ? make_uint2(a1.x, a1.y) : make_uint2(a0.z, a0.w);
						node_step(n1, node + 1u, n1.y >> 2);
					}
				}
			}
#endif
#pragma unroll 1
			for(int s = 0; s < ((YAFGPU_TRACE_PAIR && !kAny && !kBlk) ? 0 : kNodeBurst); ++s)
			{
				if(kStats && __ballot(ws == kWalk) != 0ull) ++rounds_node;
#if YAFGPU_TRACE_TRIPF
				if(p_cur < p_end && !q_ok) { q0 = sc.tri[3u * ti]; q1 = sc.tri[3u * ti + 1u]; q2 = sc.tri[3u * ti + 2u]; q_ok = true; }
#endif
				if(YAFGPU_TRACE_TOP && ws == kWalk && (node & kTopTag)) top_step();
				else if(ws == kWalk) { const uint2 nd = kBlk ? sc.nodes_blk[node] : sc.nodes[node]; node_step(nd, node + 1u, nd.y >> 2); }
			}
			if(YAFGPU_TRACE_LEAF_APART && ws == kAtLeaf) leaf_visit();
		}
#endif
		if(done || (ws == kWalkEnd && p_cur >= p_end))
		{
			if(kAny) answer_any(hit);
			else a.state[2 * c + qi] = make_float4(fbits((uint32_t)(hit ? tri : -1)), z, bu, bv);
			ws = kNoRay; done = false; p_cur = p_end = 0u;
#if YAFGPU_TRACE_TRIPF
			q_ok = false;
#endif
		}
	}
	if(a.ra.counters != nullptr)
	{
		const uint32_t v0 = wave_sum(cn.closest), v1 = wave_sum(cn.shadow), v2 = wave_sum(cn.interior), v3_ = wave_sum(cn.leaves),
		               v4 = wave_sum(cn.tests), v6 = wave_sum(cn.restarts);
		if(lane == 0 && (v0 | v1))
		{
			if(v0) atomicAdd((unsigned long long *)&a.ra.counters->rays_closest, (unsigned long long)v0);
			if(v1) atomicAdd((unsigned long long *)&a.ra.counters->rays_shadow, (unsigned long long)v1);
			if(kStats)
			{
				atomicAdd((unsigned long long *)&a.ra.counters->interior_steps, (unsigned long long)v2);
				atomicAdd((unsigned long long *)&a.ra.counters->leaves, (unsigned long long)v3_);
				atomicAdd((unsigned long long *)&a.ra.counters->tri_tests, (unsigned long long)v4);
				atomicAdd((unsigned long long *)&a.ra.counters->restarts, (unsigned long long)v6);
				atomicAdd((unsigned long long *)&a.ra.counters->wave_rounds, (unsigned long long)rounds_node | ((unsigned long long)rounds_tri << 32));
			}
		}
	}
}

// Scene::isShadowed with transparent shadows (scene.cc:996-1035) over the shadow queue: one lane per ray, no refill —
// the feature path for scenes with transparent materials and transpShad, not the benchmark path.

__global__ __launch_bounds__(kBlock) void wf_trace_ts(const WfArgs a)
{
	__shared__ uint2 s_stack[kWavesPerBlock][kStack][kWave];
	__shared__ uint32_t s_seen[kWavesPerBlock][kTsMaxDepth + 1][kWave];
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	const DevScene &sc = a.ra.sc;
#if YAFGPU_TRACE_TOP == 2
	__shared__ uint4 s_top[kTopN];
	for(int i = (int)threadIdx.x; i < kTopN; i += (int)blockDim.x) s_top[i] = sc.top[i];
	__syncthreads();
#endif
#ifdef YAFGPU_TRACE_LDS_PAD      // (A/B aid: the LDS the staged top would take, without it — the occupancy alone)
	__shared__ uint32_t s_pad[YAFGPU_TRACE_LDS_PAD / 4];
	if(a.cap == 0xffffffffu) s_pad[threadIdx.x] = 1u;
#endif
	constexpr uint32_t kRoot = YAFGPU_TRACE_TOP ? kTopTag : 0u;
	LaneStack stk;
	stk.col = &s_stack[wave][0][lane];
	uint32_t *seen = &s_seen[wave][0][lane];
	const uint32_t n = a.cnt_in[1];
	const size_t c = a.cap;
	uint32_t count = 0u;
	for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		uint32_t slot = a.q_shadow_in[i];
		const uint32_t which = slot >> 31; slot &= 0x7fffffffu;
		float4 r0 = a.state[slot], r1 = a.state[c + slot];
		if(which)
		{
			const float4 r20 = a.state[20 * c + slot];
			r1 = make_float4(r20.x, r20.y, r20.z, a.state[21 * c + slot].w);
			r0.w = r20.w;
		}
		const V3 dir = v3(r1);
		const V3 from = v3(r0) + dir * r0.w;
		const float dist = (r1.w < 0.f) ? INFINITY : r1.w - 2.f * r0.w;
		Col filt;
		const bool sh = kd_trace_ts(sc, stk, seen, from, dir, r0.w, dist, a.ra.rp.shadow_depth, filt);     // sray keeps ray.tmin_ (:998-999)
		if(sh) { const uint32_t bit = 4u * slot + which; atomicOr(&a.verdict[bit >> 5], 1u << (bit & 31u)); }      // (a park has one pair under transparent shadows)
		a.shadow_filt[2u * slot + which] = f4(filt, 0.f);
		++count;
	}
	if(a.ra.counters != nullptr)
	{
		const uint32_t v = wave_sum(count);
		if(lane == 0 && v) atomicAdd((unsigned long long *)&a.ra.counters->rays_shadow, (unsigned long long)v);
	}
}

#endif // YAFGPU_VARIANT_TU
// resume every answered path: entries [0, n_closest) come from the closest queue, the rest from the resume queue
#ifndef YAFGPU_SHADE_WAVES
#define YAFGPU_SHADE_WAVES 3     // 168 VGPRs, 44 B of scratch; C2: 3 -> 7.25, 4 -> 8.5 ms per pass (4: 224 B of scratch)
#endif
// Material and light tables staged in LDS: the path program reads dozens of their fields per step through per-lane indices,
// i.e. as vector loads, each a trip to the vector cache the wave then waits for (PMC: 92 VMEM reads per wave item, 79 % of
// wave cycles waiting at 3 waves per SIMD).  Scenes whose tables fit kShadeTabBytes read them from LDS instead.
#ifndef YAFGPU_SHADE_LDS_TABLES
#define YAFGPU_SHADE_LDS_TABLES 1
#endif
constexpr int kShadeTabBytes = 16384;     // 3 blocks per CU at 3 waves per SIMD
// ... and the Faure permutations: scrHalton__ looks one digit up per loop turn (13 dependent loads for a base-5 sample index
// of 2^30), two to four calls per path segment
#ifndef YAFGPU_SHADE_LDS_FAURE
#define YAFGPU_SHADE_LDS_FAURE 1
#endif
constexpr int kShadeFaureBytes = YAFGPU_SHADE_LDS_FAURE ? 20480 : 0;      // 5117 ints for the 50 dimensions
__global__ __launch_bounds__(kBlock, YAFGPU_SHADE_WAVES) void wf_shade(const WfArgs a_in)
{
#if YAFGPU_SHADE_LDS_TABLES
	__shared__ uint4 s_tab[(kShadeTabBytes + kShadeFaureBytes) / 16];
	WfArgs a = a_in;
	if(YAFGPU_SHADE_LDS_FAURE && (uint32_t)a_in.ra.sc.n_faure * 4u <= (uint32_t)kShadeFaureBytes)
	{
		int *dst = (int *)s_tab + kShadeTabBytes / 4;
		for(uint32_t w = threadIdx.x; w < (uint32_t)a_in.ra.sc.n_faure; w += blockDim.x) dst[w] = a_in.ra.sc.faure[w];
		a.ra.sc.faure = dst;        // (the barrier below, or the one of the first queue round, orders the copy before any use)
		__syncthreads();
	}
	{
		const uint32_t mat_bytes = (uint32_t)a_in.ra.sc.n_mats * (uint32_t)sizeof(yafgpu_material);      // multiples of 8
		const uint32_t light_off = (mat_bytes + 15u) & ~15u;
		const uint32_t light_bytes = (uint32_t)a_in.ra.sc.n_lights * (uint32_t)sizeof(yafgpu_light);
		if(light_off + light_bytes <= (uint32_t)kShadeTabBytes)
		{
			uint32_t *dst = (uint32_t *)s_tab;
			const uint32_t *src_m = (const uint32_t *)a_in.ra.sc.mats, *src_l = (const uint32_t *)a_in.ra.sc.lights;
			for(uint32_t w = threadIdx.x; w < mat_bytes / 4u; w += blockDim.x) dst[w] = src_m[w];
			for(uint32_t w = threadIdx.x; w < light_bytes / 4u; w += blockDim.x) dst[light_off / 4u + w] = src_l[w];
			__syncthreads();
			a.ra.sc.mats = (const yafgpu_material *)s_tab;
			a.ra.sc.lights = (const yafgpu_light *)((const char *)s_tab + light_off);
		}
	}
#else
	const WfArgs &a = a_in;
#endif
	// Queue appends are aggregated per workgroup over kItems items per thread: one returning atomic per queue
	// per 1024 paths.  (A single counter word sustains ~90 returning atomics per microsecond; one per wave and
	// queue — 2 M per pass on C2 — was the whole cost of this kernel.)
	constexpr int kItems = 4;
	__shared__ uint32_t s_tot[kWavesPerBlock][3];
	__shared__ uint32_t s_base[kWavesPerBlock][3];
	__shared__ uint32_t s_item[kItems][kBlock];
	__shared__ uint8_t s_more[kItems][kBlock];      // bit0 / bit1: the first / second ray of a SECOND shadow pair (WfArgs::multi)
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	const uint32_t nc = a.cnt_in[0], nr = a.cnt_in[4];
	const uint32_t total = nc + nr;
	// (the closest-hit queries a final pass answers from the record pass's cache count as the rays they stand for)
	if(YAFGPU_FEAT_LIGHTS && a.replay == 2 && a.hit_cache != nullptr && a.ra.counters != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && nc)
		atomicAdd((unsigned long long *)&a.ra.counters->rays_closest, (unsigned long long)nc);
	const uint32_t per_block = (uint32_t)kBlock * kItems;
	for(uint32_t base = blockIdx.x * per_block; base < total; base += gridDim.x * per_block)
	{
		// what each item asked for, kept in LDS between the two loops: slot | code << 27
		// (code: bit0 closest, bit1 resume, bit2 shadow ray A, bit3 shadow ray B, bit4 the closest ray is in record 26); the item loop is not unrolled, so
		// the register footprint is that of one item
		uint32_t wc = 0u, wr = 0u, ws = 0u;            // this wave's totals per queue
#pragma unroll 1
		for(int k = 0; k < kItems; ++k)
		{
			const uint32_t i = base + (uint32_t)k * kBlock + threadIdx.x;
			const bool live = i < total;
			int code = 0, more = 0;
			uint32_t slot = 0u;
			if(live)
			{
				slot = (i < nc) ? (a.q_closest_in ? (a.q_closest_in[i] & 0x7fffffffu) : i) : a.q_resume_in[i - nc];      // (bit 31: parked beside a shadow pair — the control word says so too)
				// the closest-hit answer of queue entry i (the traversal kernel wrote it at the ray's queue position)
				float4 ans = make_float4(0.f, 0.f, 0.f, 0.f);
				if(i < nc)
				{
					if(YAFGPU_FEAT_LIGHTS && a.replay == 2 && a.hit_cache != nullptr)
					{	// the final pass of a serial-state replay: the record pass kept this query's answer (wf_hit_key; a segment parked beside a shadow
						// pair is filed under the control word st_start_path / st_extend would have parked it with: wf_ctl_of_beside)
						uint32_t ctl = ubits(wf_rec(a, 13, slot).w);
						if(a.q_closest_in && (a.q_closest_in[i] >> 31)) ctl = wf_ctl_of_beside(ctl);
						const uint32_t z19 = a.ev_m > 1 ? ubits(wf_rec(a, 19, slot).z) : 0u;
						ans = a.hit_cache[wf_hit_key(a, slot, ctl, z19)];
					}
					else ans = a.state[2 * (size_t)a.cap + i];
				}
				int px, py, sample; uint32_t pixel_sample, sampling_offs, ordinal;
				wf_identity<true>(a, slot, px, py, sample, pixel_sample, sampling_offs, ordinal);
				float res[4];
				int m = 0;
				const int req = wf_advance(a, slot, pixel_sample, sampling_offs, ordinal, ans, res, m);
				if(req == kReqDone)
				{
					if(res[3] > 1.f) res[3] = 1.f;    // integrator_tiled.cc:459
					a.results[slot] = make_float4(res[0], res[1], res[2], res[3]);
				}
				else if(req == kReqClosest) code = 1;
				else if(req == kReqShadow) code = 2 | ((m & 1) ? 4 : 0) | ((m & 2) ? 8 : 0);
				else if(req == kReqBoth) code = 1 | 16 | ((m & 1) ? 4 : 0) | ((m & 2) ? 8 : 0);      // the closest queue resumes it: no resume entry
				if(req == kReqShadow || req == kReqBoth) more = (m >> 2) & 3;
			}
			s_item[k][threadIdx.x] = slot | ((uint32_t)code << 27);
			s_more[k][threadIdx.x] = (uint8_t)more;
			wc += (uint32_t)__popcll(__ballot(code & 1));
			wr += (uint32_t)__popcll(__ballot(code & 2));
			ws += (uint32_t)__popcll(__ballot(code & 4)) + (uint32_t)__popcll(__ballot(code & 8)) + (uint32_t)__popcll(__ballot(more & 1)) + (uint32_t)__popcll(__ballot(more & 2));
		}
		if(lane == 0) { s_tot[wave][0] = wc; s_tot[wave][1] = wr; s_tot[wave][2] = ws; }
		__syncthreads();
		if(threadIdx.x < 3)
		{
			const int qi = (int)threadIdx.x;
			uint32_t sum = 0u;
			for(int w = 0; w < kWavesPerBlock; ++w) sum += s_tot[w][qi];
			uint32_t b = 0u;
			if(sum) b = atomicAdd(&a.cnt_out[qi == 0 ? 0 : (qi == 1 ? 4 : 1)], sum);
			for(int w = 0; w < kWavesPerBlock; ++w) { s_base[w][qi] = b; b += s_tot[w][qi]; }
		}
		__syncthreads();
		uint32_t oc = s_base[wave][0], orr = s_base[wave][1], os = s_base[wave][2];
		const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
		for(int k = 0; k < kItems; ++k)
		{
			const uint32_t item = s_item[k][threadIdx.x];
			const int code = (int)(item >> 27);
			const uint32_t slot = item & 0x07ffffffu;
			const unsigned long long bc = __ballot(code & 1), br = __ballot(code & 2), ba = __ballot(code & 4), bb = __ballot(code & 8);
			if(code & 1) a.q_closest_out[oc + (uint32_t)__popcll(bc & below)] = slot | ((code & 16) ? 0x80000000u : 0u);
			if(code & 2) a.q_resume_out[orr + (uint32_t)__popcll(br & below)] = slot;
			if(code & 4) a.q_shadow_out[os + (uint32_t)__popcll(ba & below)] = slot;
			const uint32_t na = (uint32_t)__popcll(ba);
			if(code & 8) a.q_shadow_out[os + na + (uint32_t)__popcll(bb & below)] = slot | 0x80000000u;
			os += na + (uint32_t)__popcll(bb);
			// (a second pair's rays: bit 30 of the entry)
			const int more = (int)s_more[k][threadIdx.x];
			const unsigned long long b2a = __ballot(more & 1), b2b = __ballot(more & 2);
			if(b2a | b2b)
			{
				if(more & 1) a.q_shadow_out[os + (uint32_t)__popcll(b2a & below)] = slot | 0x40000000u;
				const uint32_t n2a = (uint32_t)__popcll(b2a);
				if(more & 2) a.q_shadow_out[os + n2a + (uint32_t)__popcll(b2b & below)] = slot | 0xc0000000u;
				os += n2a + (uint32_t)__popcll(b2b);
			}
			oc += (uint32_t)__popcll(bc); orr += (uint32_t)__popcll(br);
		}
		__syncthreads();   // s_tot / s_base are reused by the next round
	}
}

// Rgb::clampProportionalRgb (color.h:412-445), applied by addSample to every sample (imagefilm.cc:975)
YG_DEV float4 wf_clamped(float4 c, float max_value)
{
	if(max_value > 0.f)
	{
		const float max_rgb = smax(c.x, smax(c.y, c.z));
		const float adj = max_value / max_rgb;
		if(max_rgb > max_value)
		{
			if(c.x >= max_rgb) { c.x = max_value; c.y *= adj; c.z *= adj; }
			else if(c.y >= max_rgb) { c.y = max_value; c.x *= adj; c.z *= adj; }
			else { c.z = max_value; c.x *= adj; c.y *= adj; }
		}
	}
	return c;
}

#ifndef YAFGPU_VARIANT_TU      // 
// ImageFilm::addSample (imagefilm.cc:925-1015), box filter half-width 0.501: one thread per pixel of the
// chunk adds its samples in index order into the own / right / down / diagonal planes
__global__ __launch_bounds__(kBlock) void wf_accumulate(const WfArgs a)
{
	const yafgpu_render_params &rp = a.ra.rp;
	const int spp = rp.aa_minsamples;
	const int cx1 = rp.xstart + rp.width, cy1 = rp.ystart + rp.height;
	const size_t plane_stride = (size_t)rp.width * (size_t)rp.height * YAFGPU_FILM_CHANNELS;
	for(uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x; pl < a.n_pixels; pl += gridDim.x * blockDim.x)
	{
		const uint32_t xy = a.pix_xy[pl];
		const int px = (int)(xy & 0xffffu), py = (int)(xy >> 16);
		const uint32_t sampling_offs = fnv32a((uint32_t)py * fnv32a((uint32_t)px));
		if(a.ra.wide_filter)
		{	// general footprint: table weights, neighbours through float atomics on plane 0 (their order is the only
			// non-deterministic part: last-bit noise); the pixel's own share is summed in sample order first
			const int cx0 = rp.xstart, cy0 = rp.ystart;
			const double fw = (double)a.ra.filterw, ts = (double)a.ra.table_scale;
			float own[YAFGPU_FILM_CHANNELS] = {0.f, 0.f, 0.f, 0.f, 0.f};
			for(int s = 0; s < spp; ++s)
			{
				const float4 r = wf_clamped(a.results[(size_t)pl * (size_t)spp + (size_t)s], rp.aa_clamp_samples);
				float dx, dy;
				wf_sample_offsets(a, s, sampling_offs, dx, dy);
				const int dx_0 = max(cx0 - px, round2int((double)dx - fw)), dx_1 = min(cx1 - px - 1, round2int((double)dx + fw - 1.0));
				const int dy_0 = max(cy0 - py, round2int((double)dy - fw)), dy_1 = min(cy1 - py - 1, round2int((double)dy + fw - 1.0));
				const double x_offs = (double)dx - 0.5, y_offs = (double)dy - 0.5;
				for(int j = dy_0; j <= dy_1; ++j)
				{
					const int yi = (int)floor(fabs(((double)j - y_offs) * ts));
					for(int i = dx_0; i <= dx_1; ++i)
					{
						const int xi = (int)floor(fabs(((double)i - x_offs) * ts));
						const float w = a.ra.filter_table[yi * 16 + xi];
						if(i == 0 && j == 0)
						{
							own[0] += r.x * w; own[1] += r.y * w; own[2] += r.z * w; own[3] += r.w * w; own[4] += w;
						}
						else
						{
							float *dst = a.ra.planes + ((size_t)(py + j - cy0) * (size_t)rp.width + (size_t)(px + i - cx0)) * YAFGPU_FILM_CHANNELS;
							atomicAdd(dst + 0, r.x * w); atomicAdd(dst + 1, r.y * w); atomicAdd(dst + 2, r.z * w); atomicAdd(dst + 3, r.w * w);
							atomicAdd(dst + 4, w);
						}
					}
				}
			}
			float *dst = a.ra.planes + ((size_t)(py - cy0) * (size_t)rp.width + (size_t)(px - cx0)) * YAFGPU_FILM_CHANNELS;
#pragma unroll
			for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) atomicAdd(dst + c, own[c]);
			continue;
		}
		const size_t pix = ((size_t)(py - rp.ystart) * (size_t)rp.width + (size_t)(px - rp.xstart)) * YAFGPU_FILM_CHANNELS;
		float acc[YAFGPU_FILM_PLANES][YAFGPU_FILM_CHANNELS];
#pragma unroll
		for(int k = 0; k < YAFGPU_FILM_PLANES; ++k)
#pragma unroll
			for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) acc[k][c] = rp.accumulate ? a.ra.planes[(size_t)k * plane_stride + pix + c] : 0.f;   // a later pass carries the running sums on
		for(int s = 0; s < spp; ++s)
		{
			const float4 r = wf_clamped(a.results[(size_t)pl * (size_t)spp + (size_t)s], rp.aa_clamp_samples);
			float dx, dy;
			wf_sample_offsets(a, s, sampling_offs, dx, dy);
			const int dx_1 = min(cx1 - px - 1, round2int((double)dx + (double)a.ra.filterw - 1.0));
			const int dy_1 = min(cy1 - py - 1, round2int((double)dy + (double)a.ra.filterw - 1.0));
			acc[0][0] += r.x; acc[0][1] += r.y; acc[0][2] += r.z; acc[0][3] += r.w; acc[0][4] += 1.f;
			if(dx_1 >= 1) { acc[1][0] += r.x; acc[1][1] += r.y; acc[1][2] += r.z; acc[1][3] += r.w; acc[1][4] += 1.f; }
			if(dy_1 >= 1) { acc[2][0] += r.x; acc[2][1] += r.y; acc[2][2] += r.z; acc[2][3] += r.w; acc[2][4] += 1.f; }
			if(dx_1 >= 1 && dy_1 >= 1) { acc[3][0] += r.x; acc[3][1] += r.y; acc[3][2] += r.z; acc[3][3] += r.w; acc[3][4] += 1.f; }
		}
#pragma unroll
		for(int k = 0; k < YAFGPU_FILM_PLANES; ++k)
		{
			if(k == 0 || acc[k][4] != 0.f)
			{
				float *dst = a.ra.planes + (size_t)k * plane_stride + pix;
#pragma unroll
				for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) dst[c] = acc[k][c];
			}
		}
	}
	if(a.ra.counters != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
		atomicAdd((unsigned long long *)&a.ra.counters->camera_samples, (unsigned long long)a.n_paths);
}

// ---- serial-state replay between the record pass and the final pass (see WfArgs::replay) ----------------------------
struct ReplayArgs
{
	const uint32_t *seg_begin;     // n_seg + 1: first pixel (chunk-local) of every tile segment of the chunk, in the reference's tile order
	const uint32_t *seg_seed;      // n_seg: seed of the tile's Random (integrator_tiled.cc:319)
	uint32_t n_seg, spp, n_paths, n_prob;   // path samples per camera sample; probabilities per entry (bounces - 1, at least 1)
	uint32_t bounces;
	const uint32_t *ev_flags; const float *ev_p; uint8_t *ev_kill; uint8_t *ev_calls; uint32_t *lc_base;
	uint32_t *seg_total;           // n_seg: light calls per segment, then (wf_replay_bases) the counter value the segment starts with
	uint32_t *lc_counter;          // correlative_sample_number_ so far (carried over chunks and passes)
	const uint32_t *seg_base_in;   // sharded frames: the counter value every segment starts with, from the ranks' exchange (else nullptr: wf_replay_bases left it in seg_total)
};

// One wave per tile.  Entries (camera sample x path sample) of a tile are contiguous and already in the reference's
// order (pixels row by row, samples, path samples); lanes fetch 64 of them at a time, lane 0 walks them with the tile's
// MWC stream: one draw per roulette test of a path that is still alive, exactly as integrate() draws them
// (integrator_path_tracer.cc:282-288: `probability <= 0 || probability < random_value` kills), and counts the
// estimateOneDirectLight calls made up to the kill (the call of the killing depth comes before the test, :273 / :282).
__global__ __launch_bounds__(kWave) void wf_replay_tiles(const ReplayArgs r)
{
	__shared__ uint32_t s_flags[kWave];
	__shared__ float s_p[kWave][12];
	__shared__ uint8_t s_kill[kWave], s_calls[kWave];
	const int lane = (int)threadIdx.x;
	for(uint32_t seg = blockIdx.x; seg < r.n_seg; seg += gridDim.x)
	{
		const uint32_t per_px = r.spp * r.n_paths;
		const uint32_t e0 = r.seg_begin[seg] * per_px, n_ent = (r.seg_begin[seg + 1u] - r.seg_begin[seg]) * per_px;
		Mwc rr; rr.init(r.seg_seed[seg]);
		uint32_t total = 0u;
		// (four groups of 64 entries per turn: their flags are asked for together — the walk itself stays in entry order)
		constexpr uint32_t kGroups = 4u;
		for(uint32_t base0 = 0u; base0 < n_ent; base0 += kGroups * kWave)
		{
			uint32_t fl[kGroups];
#pragma unroll
			for(uint32_t g = 0u; g < kGroups; ++g)
			{
				const uint32_t off = base0 + g * kWave + (uint32_t)lane;
				fl[g] = off < n_ent ? r.ev_flags[e0 + off] : 0u;
			}
#pragma unroll
			for(uint32_t g = 0u; g < kGroups; ++g)
			{
				const uint32_t base = base0 + g * kWave;
				if(base >= n_ent) break;
				const uint32_t e = e0 + base + (uint32_t)lane;
				const bool live = base + (uint32_t)lane < n_ent;
				const uint32_t flags = fl[g];
				const uint32_t tests = flags >> 16;
				if(__ballot(tests != 0u) == 0ull)
				{	// no roulette test among these 64: nothing serial to do
					const uint32_t calls = (uint32_t)__popc(flags & 0xffffu);
					if(live) { r.ev_kill[e] = 255u; r.ev_calls[e] = (uint8_t)calls; }
					total += wave_sum(calls);
					continue;
				}
				s_flags[lane] = flags;
				for(uint32_t d = 1u; d < r.bounces; ++d)
					if((tests >> d) & 1u) s_p[lane][d] = r.ev_p[(size_t)e * r.n_prob + (d - 1u)];
				__syncthreads();
				if(lane == 0)
				{
					const uint32_t cnt = min((uint32_t)kWave, n_ent - base);
					for(uint32_t j = 0u; j < cnt; ++j)
					{
						const uint32_t f = s_flags[j];
						uint32_t kill = 255u, calls = f & 1u;
						for(uint32_t d = 1u; d < r.bounces; ++d)
						{
							calls += (f >> d) & 1u;
							if((f >> (16u + d)) & 1u)
							{
								const float random_value = (float)rr.next();
								const float probability = s_p[j][d];
								if(probability <= 0.f || probability < random_value) { kill = d; break; }
							}
						}
						s_kill[j] = (uint8_t)kill; s_calls[j] = (uint8_t)calls;
						total += calls;
					}
				}
				__syncthreads();
				if(live) { r.ev_kill[e] = s_kill[lane]; r.ev_calls[e] = s_calls[lane]; }
				__syncthreads();
			}
		}
		if(lane == 0) r.seg_total[seg] = total;
	}
}

// exclusive scan of the segments' call counts in tile order, on top of the counter so far (a few hundred to a few thousand values)
__global__ void wf_replay_bases(const ReplayArgs r)
{
	if(blockIdx.x != 0 || threadIdx.x != 0) return;
	uint32_t run = *r.lc_counter;
	for(uint32_t k = 0u; k < r.n_seg; ++k) { const uint32_t t = r.seg_total[k]; r.seg_total[k] = run; run += t; }
	*r.lc_counter = run;
}

// the counter value every camera sample starts with: the segment's base + the calls of the samples before it in the tile
__global__ __launch_bounds__(kWave) void wf_replay_samples(const ReplayArgs r)
{
	const int lane = (int)threadIdx.x;
	for(uint32_t seg = blockIdx.x; seg < r.n_seg; seg += gridDim.x)
	{
		const uint32_t s0 = r.seg_begin[seg] * r.spp, n_slots = (r.seg_begin[seg + 1u] - r.seg_begin[seg]) * r.spp;
		uint32_t run = r.seg_base_in ? r.seg_base_in[seg] : r.seg_total[seg];
		constexpr uint32_t kGroups = 4u;      // (as in wf_replay_tiles: four groups' loads in flight, the scan in slot order)
		for(uint32_t base0 = 0u; base0 < n_slots; base0 += kGroups * kWave)
		{
			uint32_t cl[kGroups];
#pragma unroll
			for(uint32_t g = 0u; g < kGroups; ++g)
			{
				const uint32_t off = base0 + g * kWave + (uint32_t)lane;
				uint32_t calls = 0u;
				if(off < n_slots) for(uint32_t i = 0u; i < r.n_paths; ++i) calls += r.ev_calls[(size_t)(s0 + off) * r.n_paths + i];
				cl[g] = calls;
			}
#pragma unroll
			for(uint32_t g = 0u; g < kGroups; ++g)
			{
				const uint32_t base = base0 + g * kWave;
				if(base >= n_slots) break;
				const uint32_t slot = s0 + base + (uint32_t)lane;
				const bool live = base + (uint32_t)lane < n_slots;
				const uint32_t calls = cl[g];
				uint32_t incl = calls;
#pragma unroll
				for(int o = 1; o < kWave; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o, kWave); if(lane >= o) incl += v; }
				if(live) r.lc_base[slot] = run + incl - calls;
				run += (uint32_t)__shfl((int)incl, kWave - 1, kWave);
			}
		}
	}
}

#endif // YAFGPU_VARIANT_TU
} // namespace yafgpu
