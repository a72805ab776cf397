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

constexpr int kWfRecs = 20;   // float4 records of parked state per path (320 B)

struct WfArgs
{
	RenderArgs ra;
	float4 *state; uint32_t cap;      // record k of path s lives at state[k * cap + s]
	float4 *results;                  // final rgba per path
	uint32_t n_paths, pixel_begin, n_pixels;
	const uint32_t *pix_prefix;       // n_tiles+1 prefix sums of pixels per tile of this shard
	const uint32_t *q_closest_in, *q_shadow_in;   // nullptr closest queue = identity (first iteration)
	uint32_t *q_closest_out, *q_shadow_out;
	uint32_t *cnt_in;                 // [0] closest count, [1] shadow count, [2] closest fetch cursor, [3] shadow fetch cursor
	uint32_t *cnt_out;                // same layout, filled by wf_shade for the next iteration
};

enum : int { kPcAfterClosest = 1, kPcAfterShadow = 2 };
enum : int { kReqDone = 0, kReqClosest = 1, kReqShadow = 2 };

struct PathRegs
{
	V3 r_from, r_dir; float r_tmin, r_tmax;          // r0, r1: the pending ray (closest: as is; shadow: dir/tmin/tmax of the light ray)
	int tri; float t, bu, bv;                        // r2: answer of the closest-hit query
	V3 sp0_p, sp0_n, sp0_ng; int mat0; uint32_t bsdfs0; V3 wo0; float alpha;   // r3..r6
	V3 hit_p, hit_n, hit_ng; int hit_mat; V3 pwo;    // r7..r10
	uint32_t sampled_flags, offs, one_light_calls;
	Col throughput, path_col, col; Mwc rr;           // r11..r13
	int pc, stage, path_i, depth, dl_on_sp0;
	Col pending, ccol, ccol_2, col_dirac, total;     // r14..r18
	int li, l_end, phase, is, shadowed;
	uint32_t have, dirty;                            // record groups present in registers / modified
};

YG_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
YG_DEV float4 f4(Col c, float w) { return make_float4(c.r, c.g, c.b, w); }
YG_DEV V3 v3(float4 f) { return mk(f.x, f.y, f.z); }
YG_DEV Col c3(float4 f) { return mkc(f.x, f.y, f.z); }
YG_DEV float fbits(uint32_t u) { return __uint_as_float(u); }
YG_DEV uint32_t ubits(float f) { return __float_as_uint(f); }

// Parked state is read and written by record group, on demand: a path that resumes after a shadow
// ray and parks again for the next one touches ~10 of the 20 records, not all of them.
enum : uint32_t {
	G_RAY = 1u << 0,    // r0, r1   pending ray
	G_ANS = 1u << 1,    // r2       answer of the last query
	G_SP0 = 1u << 2,    // r3..r6   camera hit: p|mat0, n, ng|bsdfs0, wo0
	G_HIT = 1u << 3,    // r7..r10  current path vertex: p|mat, n, ng, pwo
	G_PATH = 1u << 4,   // r11, r12 throughput|rr.x, path_col|rr.c
	G_CTRL = 1u << 5,   // r13      col | pc, stage, depth, path_i
	G_DLC = 1u << 6,    // r14      pending contribution | li, l_end, phase, is
	G_ACC = 1u << 7,    // r15..r18 ccol, ccol_2, col_dirac, total
	G_MISC = 1u << 8,   // r19      offs, sampled_flags, one_light_calls, alpha
};

YG_DEV void wf_need(const WfArgs &a, uint32_t s, PathRegs &p, uint32_t groups)
{
	groups &= ~p.have;
	if(!groups) return;
	p.have |= groups;
	const float4 *b = a.state + s; const size_t c = a.cap;
	float4 r;
	if(groups & G_RAY) { r = b[0 * c]; p.r_from = v3(r); p.r_tmin = r.w; r = b[1 * c]; p.r_dir = v3(r); p.r_tmax = r.w; }
	if(groups & G_ANS) { r = b[2 * c]; p.tri = (int)ubits(r.x); p.t = r.y; p.bu = r.z; p.bv = r.w; p.shadowed = (p.tri != 0); }
	if(groups & G_SP0)
	{
		r = b[3 * c]; p.sp0_p = v3(r); p.mat0 = (int)ubits(r.w);
		r = b[4 * c]; p.sp0_n = v3(r);
		r = b[5 * c]; p.sp0_ng = v3(r); p.bsdfs0 = ubits(r.w);
		r = b[6 * c]; p.wo0 = v3(r);
	}
	if(groups & G_HIT)
	{
		r = b[7 * c]; p.hit_p = v3(r); p.hit_mat = (int)ubits(r.w);
		r = b[8 * c]; p.hit_n = v3(r);
		r = b[9 * c]; p.hit_ng = v3(r);
		r = b[10 * c]; p.pwo = v3(r);
	}
	if(groups & G_PATH) { r = b[11 * c]; p.throughput = c3(r); p.rr.x = ubits(r.w); r = b[12 * c]; p.path_col = c3(r); p.rr.c = ubits(r.w); }
	if(groups & G_CTRL)
	{
		r = b[13 * c]; p.col = c3(r);
		const uint32_t w = ubits(r.w);
		p.pc = (int)(w & 3u); p.stage = (int)((w >> 2) & 3u); p.dl_on_sp0 = (int)((w >> 4) & 1u);
		p.depth = (int)((w >> 8) & 0xffu); p.path_i = (int)(w >> 16);
	}
	if(groups & G_DLC)
	{
		r = b[14 * c]; p.pending = c3(r);
		const uint32_t w = ubits(r.w);
		p.li = (int)(w & 0xffu); p.l_end = (int)((w >> 8) & 0xffu); p.phase = (int)((w >> 16) & 0xfu); p.is = (int)(w >> 20);
	}
	if(groups & G_ACC) { p.ccol = c3(b[15 * c]); p.ccol_2 = c3(b[16 * c]); p.col_dirac = c3(b[17 * c]); p.total = c3(b[18 * c]); }
	if(groups & G_MISC) { r = b[19 * c]; p.offs = ubits(r.x); p.sampled_flags = ubits(r.y); p.one_light_calls = ubits(r.z); p.alpha = r.w; }
}

YG_DEV void wf_store(const WfArgs &a, uint32_t s, const PathRegs &p)
{
	float4 *b = a.state + s; const size_t c = a.cap;
	const uint32_t d = p.dirty;
	if(d & G_RAY) { b[0 * c] = f4(p.r_from, p.r_tmin); b[1 * c] = f4(p.r_dir, p.r_tmax); }
	if(d & G_SP0)
	{
		b[3 * c] = f4(p.sp0_p, fbits((uint32_t)p.mat0));
		b[4 * c] = f4(p.sp0_n, 0.f);
		b[5 * c] = f4(p.sp0_ng, fbits(p.bsdfs0));
		b[6 * c] = f4(p.wo0, 0.f);
	}
	if(d & G_HIT)
	{
		b[7 * c] = f4(p.hit_p, fbits((uint32_t)p.hit_mat));
		b[8 * c] = f4(p.hit_n, 0.f);
		b[9 * c] = f4(p.hit_ng, 0.f);
		b[10 * c] = f4(p.pwo, 0.f);
	}
	if(d & G_PATH) { b[11 * c] = f4(p.throughput, fbits(p.rr.x)); b[12 * c] = f4(p.path_col, fbits(p.rr.c)); }
	if(d & G_CTRL)
		b[13 * c] = f4(p.col, fbits((uint32_t)p.pc | ((uint32_t)p.stage << 2) | ((uint32_t)p.dl_on_sp0 << 4) | ((uint32_t)p.depth << 8) | ((uint32_t)p.path_i << 16)));
	if(d & G_DLC) b[14 * c] = f4(p.pending, fbits((uint32_t)p.li | ((uint32_t)p.l_end << 8) | ((uint32_t)p.phase << 16) | ((uint32_t)p.is << 20)));
	if(d & G_ACC) { b[15 * c] = f4(p.ccol, 0.f); b[16 * c] = f4(p.ccol_2, 0.f); b[17 * c] = f4(p.col_dirac, 0.f); b[18 * c] = f4(p.total, 0.f); }
	if(d & G_MISC) b[19 * c] = make_float4(fbits(p.offs), fbits(p.sampled_flags), fbits(p.one_light_calls), p.alpha);
}

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

// One candidate of MonteCarloIntegrator::doLightEstimation (integrator_montecarlo.cc:78-345): light
// `li`, half `phase` of the MIS pair (0 light sampling :161-262, 1 BSDF sampling :285-333; Dirac lights
// have a single half :94-148), sample `is`.  Returns whether a shadow ray is wanted and, if so, the ray
// and the radiance it would carry if unoccluded — identical arithmetic to direct_light().
YG_DEV bool dl_candidate(const RenderArgs &ra, const yafgpu_light &light, int li, int phase, int is, const SurfPt &sp, const yafgpu_material &mat,
                         const BsdfDat &dat, V3 wo, uint32_t pixel_sample, uint32_t sampling_offs, V3 &r_dir, float &r_tmin, float &r_tmax, Col &contrib)
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
	const int n = (int)ceilf((float)light.samples * ra.rp.aa_light_sample_multiplier);
	const uint32_t offs = (uint32_t)n * pixel_sample + sampling_offs + (uint32_t)li * 4567u;
	Halton hal_2, hal_3;
	hal_2.init(2u); hal_3.init(3u);
	hal_2.set_start(offs - 1u); hal_3.set_start(offs - 1u);
	float s_1 = 0.f, s_2 = 0.f;
	for(int k = 0; k <= is; ++k) { s_1 = hal_2.next(); s_2 = hal_3.next(); }   // the incremental sequence, replayed
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

// Resume a parked path and run it to its next kd-tree query (or to its end).
YG_DEV int wf_advance(const WfArgs &a, uint32_t slot, PathRegs &P, uint32_t pixel_sample, uint32_t sampling_offs, uint32_t ordinal, float result[4])
{
	const RenderArgs &ra = a.ra;
	const DevScene &sc = ra.sc;
	const yafgpu_render_params &rp = ra.rp;
	const int n_paths = rp.path_samples > 1 ? rp.path_samples : 1;
	enum { W_AFTER_CLOSEST, W_AFTER_SHADOW, W_DL_NEXT, W_DL_DONE, W_EXTEND, W_START_PATH, W_FINISH };
#define NEED(g) wf_need(a, slot, P, (g))
#define DIRTY(g) do { P.have |= (g); P.dirty |= (g); } while(0)
	NEED(G_CTRL | G_ANS);
	int where = (P.pc == kPcAfterShadow) ? W_AFTER_SHADOW : W_AFTER_CLOSEST;
	for(;;)
	{
		switch(where)
		{
			case W_AFTER_CLOSEST:
			{
				const bool got = P.tri >= 0;
				NEED(G_RAY);
				if(P.stage == kStPrimary)
				{
					DIRTY(G_CTRL | G_MISC);
					P.col = mkc(0.f, 0.f, 0.f);
					P.alpha = rp.bg_transp ? 0.f : 1.f;
					if(!got)
					{
						if(rp.has_background && !rp.bg_transp_refract) P.col = P.col + mkc(rp.background[0], rp.background[1], rp.background[2]);
						where = W_FINISH; break;
					}
					DIRTY(G_SP0 | G_PATH | G_ACC | G_DLC);
					SurfPt sp0;
					get_surface(sc, P.tri, P.r_from + P.r_dir * P.t, P.bu, P.bv, sp0);
					P.sp0_p = sp0.p; P.sp0_n = sp0.n; P.sp0_ng = sp0.ng; P.mat0 = sp0.mat;
					const yafgpu_material &m = sc.mats[sp0.mat];
					BsdfDat dat0;
					P.bsdfs0 = mat_init_bsdf(m, dat0);
					P.wo0 = -P.r_dir;
					if(P.bsdfs0 & kEmit) P.col = P.col + mat_emit(m, sp0, P.wo0, true);
					P.alpha = 1.f;
					if(rp.bg_transp_refract)
					{
						const float m_alpha = (m.type == YAFGPU_MAT_SHINYDIFFUSE) ? sd_alpha(m, dat0, sp0, P.wo0) : 1.f;
						P.alpha = m_alpha + (1.f - m_alpha) * (rp.bg_transp ? 0.f : 1.f);
					}
					P.path_col = mkc(0.f, 0.f, 0.f); P.throughput = mkc(1.f, 1.f, 1.f);
					P.rr.init(fnv32a(ordinal) + 123u);   // see DESIGN.md: Russian-roulette stream (row N4)
					P.path_i = 0; P.depth = 0; P.one_light_calls = 0u; P.sampled_flags = kNone; P.offs = 0u;
					P.total = mkc(0.f, 0.f, 0.f);
					if((P.bsdfs0 & kDiffuse) && sc.n_lights > 0)
					{
						P.li = 0; P.l_end = sc.n_lights; P.phase = 0; P.is = 0; P.dl_on_sp0 = 1;
						P.ccol = P.ccol_2 = P.col_dirac = mkc(0.f, 0.f, 0.f);
						where = W_DL_NEXT; break;
					}
					where = W_DL_DONE; break;
				}
				if(!got) { DIRTY(G_CTRL); ++P.path_i; where = W_START_PATH; break; }
				NEED(G_MISC);
				DIRTY(G_HIT | G_ACC | G_DLC | G_MISC | G_CTRL);
				SurfPt hit;
				get_surface(sc, P.tri, P.r_from + P.r_dir * P.t, P.bu, P.bv, hit);
				P.hit_p = hit.p; P.hit_n = hit.n; P.hit_ng = hit.ng; P.hit_mat = hit.mat;
				const yafgpu_material &pm = sc.mats[hit.mat];
				BsdfDat dat_n;
				const uint32_t mb = mat_init_bsdf(pm, dat_n);
				if(P.stage == kStFirst)
				{
					if(P.sampled_flags != kNone) P.pwo = -P.r_dir;
					else { const float4 r10 = a.state[10 * (size_t)a.cap + slot]; P.pwo = v3(r10); }   // keeps the pwo of the first segment (:224)
				}
				else P.pwo = -P.r_dir;
				P.total = mkc(0.f, 0.f, 0.f);
				const bool want_dl = sc.n_lights > 0 && (P.stage == kStFirst || (mb & kDiffuse));
				if(want_dl)
				{
					int lnum = 0;
					if(sc.n_lights > 1)
					{
						Halton h2; h2.init(2u);
						h2.set_start(rp.base_sampling_offset + (ordinal * 16u + P.one_light_calls) - 1u);
						lnum = min((int)(h2.next() * (float)sc.n_lights), sc.n_lights - 1);
					}
					++P.one_light_calls;
					P.li = lnum; P.l_end = lnum + 1; P.phase = 0; P.is = 0; P.dl_on_sp0 = 0;
					P.ccol = P.ccol_2 = P.col_dirac = mkc(0.f, 0.f, 0.f);
					where = W_DL_NEXT; break;
				}
				P.li = 0; P.l_end = 0;   // marks "no light estimate ran" for W_DL_DONE
				where = W_DL_DONE; break;
			}
			case W_AFTER_SHADOW:
			{
				NEED(G_DLC | G_ACC);
				DIRTY(G_DLC | G_ACC);
				if(!P.shadowed)
				{
					const bool dirac = sc.lights[P.li].type == YAFGPU_LIGHT_POINT;
					if(dirac) P.col_dirac = P.col_dirac + P.pending;
					else if(P.phase == 0) P.ccol = P.ccol + P.pending;
					else P.ccol_2 = P.ccol_2 + P.pending;
				}
				++P.is;
				where = W_DL_NEXT; break;
			}
			case W_DL_NEXT:
			{
				// iterate (li, phase, is) in the order of direct_light(): for li { for phase { for is } }
				NEED(P.dl_on_sp0 ? G_SP0 : G_HIT);
				SurfPt sp;
				if(P.dl_on_sp0) make_sp(P.sp0_p, P.sp0_n, P.sp0_ng, P.mat0, sp);
				else make_sp(P.hit_p, P.hit_n, P.hit_ng, P.hit_mat, sp);
				const yafgpu_material &mat = sc.mats[sp.mat];
				BsdfDat dat; mat_init_bsdf(mat, dat);
				const V3 wo = P.dl_on_sp0 ? P.wo0 : P.pwo;
				bool parked = false;
				while(P.li < P.l_end)
				{
					const yafgpu_light &light = sc.lights[P.li];
					const bool dirac = light.type == YAFGPU_LIGHT_POINT;
					const int n = dirac ? 1 : (int)ceilf((float)light.samples * rp.aa_light_sample_multiplier);
					const int n_phase = dirac ? 1 : 2;
					if(P.is >= n) { P.is = 0; ++P.phase; }
					if(P.phase >= n_phase)
					{
						const float inv_ns = 1.f / (float)n;
						Col col = mkc(0.f, 0.f, 0.f);
						if(dirac) col = col + P.col_dirac;
						else { col = col + P.ccol * inv_ns; col = col + P.ccol_2 * inv_ns; }
						P.total = P.total + col;
						P.ccol = P.ccol_2 = P.col_dirac = mkc(0.f, 0.f, 0.f);
						P.phase = 0; P.is = 0; ++P.li;
						continue;
					}
					V3 d; float tmin, tmax; Col contrib;
					if(dl_candidate(ra, light, P.li, P.phase, P.is, sp, mat, dat, wo, pixel_sample, sampling_offs, d, tmin, tmax, contrib))
					{
						const bool cast_shadows = light.cast_shadows && mat.receive_shadows;
						if(cast_shadows)
						{
							P.pending = contrib;
							P.r_from = sp.p; P.r_dir = d; P.r_tmin = tmin; P.r_tmax = tmax;
							parked = true;
							break;
						}
						if(dirac) P.col_dirac = P.col_dirac + contrib;
						else if(P.phase == 0) P.ccol = P.ccol + contrib;
						else P.ccol_2 = P.ccol_2 + contrib;
					}
					++P.is;
				}
				if(parked) { DIRTY(G_RAY | G_CTRL | G_DLC | G_ACC); P.pc = kPcAfterShadow; return kReqShadow; }
				where = W_DL_DONE; break;
			}
			case W_DL_DONE:
			{
				NEED(G_ACC | G_DLC);
				DIRTY(G_CTRL);
				if(P.stage == kStPrimary)
				{
					NEED(G_SP0);
					if(P.bsdfs0 & kDiffuse) P.col = P.col + P.total;                                  // :156
					const uint32_t path_flags = rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse;
					if(rp.integrator != YAFGPU_INTEGRATOR_PATH || !(P.bsdfs0 & path_flags)) { where = W_FINISH; break; }
					P.path_i = 0;
					where = W_START_PATH; break;
				}
				NEED(G_HIT | G_PATH);
				DIRTY(G_PATH);
				const yafgpu_material &pm = sc.mats[P.hit_mat];
				BsdfDat dat_n;
				const uint32_t mb = mat_init_bsdf(pm, dat_n);
				SurfPt hit; make_sp(P.hit_p, P.hit_n, P.hit_ng, P.hit_mat, hit);
				Col lcol = mkc(0.f, 0.f, 0.f);
				if(P.l_end > 0) lcol = P.total * (float)sc.n_lights;
				if(P.stage == kStFirst)
				{
					if(mb & kEmit) lcol = lcol + mat_emit(pm, hit, P.pwo, false);                    // :226
					P.path_col = P.path_col + lcol * P.throughput;                                    // :228
					P.depth = 1;
					where = (P.depth < rp.bounces) ? W_EXTEND : W_START_PATH;
					if(where == W_START_PATH) ++P.path_i;
					break;
				}
				bool alive = true;
				if(P.depth > rp.rr_min_bounces)
				{
					const float random_value = (float)P.rr.next();
					const float probability = smax(P.throughput.r, smax(P.throughput.g, P.throughput.b));
					if(probability <= 0.f || probability < random_value) alive = false;
					else P.throughput = P.throughput * (1.f / probability);
				}
				if(alive)
				{
					P.path_col = P.path_col + lcol * P.throughput;                                    // :292
					++P.depth;
					if(P.depth < rp.bounces) { where = W_EXTEND; break; }
				}
				++P.path_i;
				where = W_START_PATH; break;
			}
			case W_EXTEND:
			{
				NEED(G_HIT | G_PATH | G_MISC | G_RAY);
				DIRTY(G_PATH | G_RAY | G_CTRL);
				const yafgpu_material &pm = sc.mats[P.hit_mat];
				BsdfDat dat_n; mat_init_bsdf(pm, dat_n);
				SurfPt hit; make_sp(P.hit_p, P.hit_n, P.hit_ng, P.hit_mat, hit);
				const int d_4 = 4 * P.depth;
				BsdfSample bs;
				bs.s_1 = (float)scr_halton(sc, d_4 + 3, P.offs);
				bs.s_2 = (float)scr_halton(sc, d_4 + 4, P.offs);
				bs.pdf = 0.f; bs.sampled = kNone; bs.flags = kAll;
				float w = 0.f;
				V3 p_dir = P.r_dir;
				const Col scol = mat_sample(pm, dat_n, hit, P.pwo, p_dir, bs, w) * w;
				if(is_black(scol)) { ++P.path_i; where = W_START_PATH; break; }
				P.throughput = P.throughput * scol;
				P.r_from = hit.p; P.r_dir = p_dir; P.r_tmin = ra.ray_min_dist; P.r_tmax = -1.f;
				P.stage = kStDepth; P.pc = kPcAfterClosest;
				return kReqClosest;
			}
			case W_START_PATH:
			{
				NEED(G_PATH);
				if(P.path_i >= n_paths) { P.col = P.col + P.path_col / (float)n_paths; where = W_FINISH; break; }
				NEED(G_SP0 | G_MISC);
				DIRTY(G_PATH | G_MISC | G_RAY | G_CTRL);
				const yafgpu_material &m = sc.mats[P.mat0];
				BsdfDat dat0; mat_init_bsdf(m, dat0);
				SurfPt sp0; make_sp(P.sp0_p, P.sp0_n, P.sp0_ng, P.mat0, sp0);
				P.offs = (uint32_t)rp.path_samples * pixel_sample + sampling_offs + (uint32_t)P.path_i;
				BsdfSample bs;
				bs.s_1 = ri_vdc(P.offs, 0u);
				bs.s_2 = (float)scr_halton(sc, 2, P.offs);
				bs.pdf = 0.f; bs.sampled = kNone;
				bs.flags = (rp.no_recursive ? (uint32_t)kAll : (uint32_t)kDiffuse) | kDiffuse | kReflect | kTransmit;
				float w = 0.f;
				V3 p_dir = mk(0.f, 0.f, 0.f);
				P.pwo = P.wo0;
				a.state[10 * (size_t)a.cap + slot] = f4(P.pwo, 0.f);
				const Col scol = mat_sample(m, dat0, sp0, P.pwo, p_dir, bs, w) * w;
				P.throughput = scol;
				P.sampled_flags = bs.sampled;
				P.r_from = sp0.p; P.r_dir = p_dir; P.r_tmin = ra.ray_min_dist; P.r_tmax = -1.f;
				P.stage = kStFirst; P.pc = kPcAfterClosest;
				return kReqClosest;
			}
			default: // W_FINISH
			{
				NEED(G_MISC);
				float alpha = P.alpha;
				if(rp.bg_transp) alpha = smax(alpha, 0.f);
				result[0] = P.col.r; result[1] = P.col.g; result[2] = P.col.b; result[3] = alpha;
				return kReqDone;
			}
		}
	}
#undef NEED
#undef DIRTY
}

// identity of a path slot: slot = pixel_local * spp + sample
YG_DEV void wf_identity(const WfArgs &a, uint32_t slot, int &px, int &py, int &sample, uint32_t &pixel_sample, uint32_t &sampling_offs, uint32_t &ordinal)
{
	const yafgpu_render_params &rp = a.ra.rp;
	const uint32_t spp = (uint32_t)rp.aa_minsamples;
	const uint32_t pixel_local = slot / spp;
	sample = (int)(slot - pixel_local * spp);
	wf_pixel_of(a, pixel_local, px, py);
	sampling_offs = fnv32a((uint32_t)py * fnv32a((uint32_t)px));
	pixel_sample = rp.base_sampling_offset + (uint32_t)sample;
	ordinal = ((uint32_t)(py - rp.ystart) * (uint32_t)rp.width + (uint32_t)(px - rp.xstart)) * spp + (uint32_t)sample;
}

YG_DEV void wf_sample_offsets(const WfArgs &a, int sample, uint32_t sampling_offs, float &dx, float &dy)
{
	const int n_samples = a.ra.rp.aa_minsamples;
	dx = 0.5f; dy = 0.5f;
	if(n_samples > 1)
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

// camera rays: TiledIntegrator::renderTile :378-410
__global__ __launch_bounds__(kBlock) void wf_generate(const WfArgs a)
{
	for(uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < a.n_paths; slot += gridDim.x * blockDim.x)
	{
		int px, py, sample; uint32_t pixel_sample, sampling_offs, ordinal;
		wf_identity(a, slot, px, py, sample, pixel_sample, sampling_offs, ordinal);
		float dx, dy;
		wf_sample_offsets(a, sample, sampling_offs, dx, dy);
		V3 from, dir; float tmin, tmax;
		camera_shoot(a.ra.sc.cam, (float)px + dx, (float)py + dy, from, dir, tmin, tmax);
		float4 *b = a.state + slot; const size_t c = a.cap;
		b[0 * c] = f4(from, tmin);
		b[1 * c] = f4(dir, tmax);
		b[13 * c] = make_float4(0.f, 0.f, 0.f, fbits((uint32_t)kPcAfterClosest | ((uint32_t)kStPrimary << 2)));
	}
	if(blockIdx.x == 0 && threadIdx.x == 0) { a.cnt_in[0] = a.n_paths; a.cnt_in[1] = 0u; a.cnt_in[2] = 0u; a.cnt_in[3] = 0u; }
}

// The traversal kernels: Scene::intersect (scene.cc:896-927) / Scene::isShadowed (:962-994) over a queue.
//
// Persistent waves with ray refill: a lane whose ray is finished does not idle until the slowest ray
// of its wave ends; when at least kRefill lanes are free the wave fetches that many new rays from the
// queue with one atomic (ballot + prefix rank) and the freed lanes start them while the others carry
// on.  Traversal state (current node, [tmin,tmax], best hit, short stack in LDS) is per lane, so lanes
// of one wave can be at any point of any ray.  The walk itself is kd_trace's, cut at leaf granularity.
#ifndef YAFGPU_REFILL
#define YAFGPU_REFILL 32               // C2 sweep: 16 -> 1134, 32 -> 1408, 48 -> 1360 Mrays/s
#endif
template<bool kAny, bool kStats>
__global__ __launch_bounds__(kBlock) void wf_trace(const WfArgs a)
{
	__shared__ uint2 s_stack[kWavesPerBlock][kStack][kWave];
	const int lane = (int)(threadIdx.x & (kWave - 1)), wave = (int)(threadIdx.x >> 6);
	const DevScene &sc = a.ra.sc;
	LaneStack stk;
	stk.col = &s_stack[wave][0][lane];
	LaneCounters cn = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
	const uint32_t n = kAny ? a.cnt_in[1] : a.cnt_in[0];
	uint32_t *cursor = kAny ? &a.cnt_in[3] : &a.cnt_in[2];
	const uint32_t *q = kAny ? a.q_shadow_in : a.q_closest_in;
	const size_t c = a.cap;
	// per-lane ray + traversal state
	bool active = false, exhausted = (n == 0u) || (sc.n_nodes == 0u && false);
	uint32_t slot = 0u, node = 0u;
	V3 from = mk(0.f, 0.f, 0.f), dir = from, inv_dir = from;
	float ray_tmin = 0.f, dist = 0.f, t_exit = 0.f, tmin = 0.f, tmax = 0.f, z = 0.f, bu = 0.f, bv = 0.f;
	int tri = -1; bool hit = false;
	for(;;)
	{
		const unsigned long long idle = __ballot(!active);
		const int n_idle = __popcll(idle);
		if(!exhausted && (n_idle >= YAFGPU_REFILL || n_idle == kWave))
		{
			const int leader = __ffsll((long long)idle) - 1;
			uint32_t base = 0u;
			if(lane == leader) base = atomicAdd(cursor, (uint32_t)n_idle);
			base = (uint32_t)__shfl((int)base, leader, kWave);
			if(base + (uint32_t)n_idle >= n) exhausted = true;
			if(!active)
			{
				const uint32_t i = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
				if(i < n)
				{
					slot = q ? q[i] : i;
					const float4 r0 = a.state[slot], r1 = a.state[c + slot];
					from = v3(r0); dir = v3(r1);
					if(kAny)
					{
						from = from + dir * r0.w;
						dist = (r1.w < 0.f) ? INFINITY : r1.w - 2.f * r0.w;
						ray_tmin = 0.f;
						++cn.shadow;
					}
					else
					{
						dist = (r1.w < 0.f) ? INFINITY : r1.w;
						ray_tmin = r0.w;
						++cn.closest;
					}
					float ea, eb;
					tri = -1; hit = false; z = dist; bu = 0.f; bv = 0.f;
					if(sc.n_nodes != 0u && bound_cross(sc, from, dir, dist, ea, eb))
					{
						inv_dir = mk(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
						t_exit = eb; tmin = smax(ea, 0.f); tmax = t_exit; node = 0u;
						stk.reset();
						active = true;
					}
					else
					{	// misses the scene bound: answer at once
						a.state[2 * c + slot] = kAny ? make_float4(fbits(0u), 0.f, 0.f, 0.f) : make_float4(fbits(0xffffffffu), dist, 0.f, 0.f);
					}
				}
			}
		}
		if(__ballot(active) == 0ull) { if(exhausted) break; else continue; }
		if(active)
		{
			bool done = false, found = false;
			if(z < tmin) done = true;                     // kdtree_triangle.cc:717
			else
			{
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
					if(!(tplane <= tmax) || tplane <= 0.f) node = near_c;
					else if(tplane < tmin) node = far_c;
					else { stk.push(far_c, tmax); node = near_c; tmax = tplane; }
					nd = sc.nodes[node];
				}
				const uint32_t np = nd.y >> 2, first = nd.x;
				if(kStats) ++cn.leaves;
				for(uint32_t k = 0; k < np; ++k)
				{
					const uint32_t ti = sc.refs[first + k];
					const float4 r0 = sc.tri[3u * ti], r1 = sc.tri[3u * ti + 1u], r2 = sc.tri[3u * ti + 2u];
					float t, u, v;
					if(kStats) ++cn.tests;
					if(tri_test(r0, r1, r2, from, dir, t, u, v))
					{
						const uint32_t vis = __float_as_uint(r1.w) >> 30;
						if(kAny)
						{
							if(t < dist && t >= 0.f && (vis == 0u || vis == 2u)) { found = true; break; }
						}
						else if(t < z && t >= ray_tmin && (vis == 0u || vis == 1u)) { z = t; tri = (int)ti; bu = u; bv = v; hit = true; }
					}
				}
				if(kAny ? found : (hit && z <= tmax)) done = true;         // :822 / :936-945
				else if(stk.count == 0)
				{
					if(!stk.dropped || tmax >= t_exit) done = true;
					else { tmin = tmax; tmax = t_exit; node = 0u; stk.dropped = false; if(kStats) ++cn.restarts; }
				}
				else { tmin = tmax; stk.pop(node, tmax); }
			}
			if(done)
			{
				a.state[2 * c + slot] = kAny ? make_float4(fbits(found ? 1u : 0u), 0.f, 0.f, 0.f)
				                             : make_float4(fbits((uint32_t)(hit ? tri : -1)), z, bu, bv);
				active = false;
			}
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
			}
		}
	}
}

// resume every answered path; entries [0, n_closest) come from the closest queue, the rest from the shadow queue
#ifndef YAFGPU_SHADE_WAVES
#define YAFGPU_SHADE_WAVES 1
#endif
__global__ __launch_bounds__(kBlock, YAFGPU_SHADE_WAVES) void wf_shade(const WfArgs a)
{
	const uint32_t nc = a.cnt_in[0], ns = a.cnt_in[1];
	const uint32_t total = nc + ns;
	// every lane of a wave runs the same number of iterations so that wf_push's ballots see whole waves
	const uint32_t stride = gridDim.x * blockDim.x;
	for(uint32_t base = blockIdx.x * blockDim.x; base < total; base += stride)
	{
		const uint32_t i = base + threadIdx.x;
		const bool live = i < total;
		int req = kReqDone;
		uint32_t slot = 0u;
		if(live)
		{
			slot = (i < nc) ? (a.q_closest_in ? a.q_closest_in[i] : i) : a.q_shadow_in[i - nc];
			int px, py, sample; uint32_t pixel_sample, sampling_offs, ordinal;
			wf_identity(a, slot, px, py, sample, pixel_sample, sampling_offs, ordinal);
			PathRegs P;
			P.have = 0u; P.dirty = 0u;
			float res[4];
			req = wf_advance(a, slot, P, pixel_sample, sampling_offs, ordinal, res);
			if(req == kReqDone)
			{
				if(res[3] > 1.f) res[3] = 1.f;    // integrator_tiled.cc:459
				a.results[slot] = make_float4(res[0], res[1], res[2], res[3]);
			}
			else wf_store(a, slot, P);
		}
		wf_push(a.q_closest_out, &a.cnt_out[0], live && req == kReqClosest, slot);
		wf_push(a.q_shadow_out, &a.cnt_out[1], live && req == kReqShadow, slot);
	}
}

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
		int px, py;
		wf_pixel_of(a, pl, px, py);
		const uint32_t sampling_offs = fnv32a((uint32_t)py * fnv32a((uint32_t)px));
		float acc[YAFGPU_FILM_PLANES][YAFGPU_FILM_CHANNELS];
#pragma unroll
		for(int k = 0; k < YAFGPU_FILM_PLANES; ++k)
#pragma unroll
			for(int c = 0; c < YAFGPU_FILM_CHANNELS; ++c) acc[k][c] = 0.f;
		for(int s = 0; s < spp; ++s)
		{
			const float4 r = a.results[(size_t)pl * (size_t)spp + (size_t)s];
			float dx, dy;
			wf_sample_offsets(a, s, sampling_offs, dx, dy);
			const int dx_1 = min(cx1 - px - 1, round2int((double)dx + (double)a.ra.filterw - 1.0));
			const int dy_1 = min(cy1 - py - 1, round2int((double)dy + (double)a.ra.filterw - 1.0));
			acc[0][0] += r.x; acc[0][1] += r.y; acc[0][2] += r.z; acc[0][3] += r.w; acc[0][4] += 1.f;
			if(dx_1 >= 1) { acc[1][0] += r.x; acc[1][1] += r.y; acc[1][2] += r.z; acc[1][3] += r.w; acc[1][4] += 1.f; }
			if(dy_1 >= 1) { acc[2][0] += r.x; acc[2][1] += r.y; acc[2][2] += r.z; acc[2][3] += r.w; acc[2][4] += 1.f; }
			if(dx_1 >= 1 && dy_1 >= 1) { acc[3][0] += r.x; acc[3][1] += r.y; acc[3][2] += r.z; acc[3][3] += r.w; acc[3][4] += 1.f; }
		}
		const size_t pix = ((size_t)(py - rp.ystart) * (size_t)rp.width + (size_t)(px - rp.xstart)) * YAFGPU_FILM_CHANNELS;
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

} // namespace yafgpu
