// Inter coding of a CU for P slices with one reference picture (BASELINE configs 2-4; B slices are outside the built rows):
// motion search, compensation, vector prediction, merge evaluation, the inter transform tree.
// Restates hmr_motion_inter.c: encode_inter_cu / _chroma :40-230, select_mv_candidate :975-1031, hmr_motion_estimation :1404-1775,
// hmr_motion_compensation_luma / _chroma :1779-1907, get_merge_mvp_candidates :1937-2180, get_amvp_candidates :2342-2448,
// hmr_cu_motion_estimation :2471-2880, predict_inter :2924-3067, encode_inter :3071-3294, check_rd_cost_merge_2nx2n :3493-3742.
#pragma once
#include "enc_common.h"

namespace henc {

// ---- motion compensation ------------------------------------------------------------------------------------------
// Device: the reference picture exists as 8-bit phase planes (FrameCtx::sub_y / sub_c, built once per frame by k_subpel.hip with the stage rules of
// hmr_motion_inter.c:240-391 / inter_prediction.c:796,818), so the prediction of a block is a copy.  Checker build: the interpolation itself, from ref[].
#if !defined(__HIPCC__)
// `ref` points at the co-located block (mv = 0) in the padded reference plane
template <class G>
HENC_HD void mc_luma_interp(const G g, Enc &__restrict__ e, const int16_t *ref, int rs, int16_t *pred, int ps, int n, int mvx, int mvy)
{
	HENC_ENC_IN_LDS(e);
	const int xf = mvx & 3, yf = mvy & 3;
	const int16_t *src = ref + (mvy >> 2) * rs + (mvx >> 2);
	if (xf == 0) interp_stage<8>(g, src, rs, pred, ps, yf, n, n, 1, 1, 1);
	else if (yf == 0) interp_stage<8>(g, src, rs, pred, ps, xf, n, n, 0, 1, 1);
	else {
		int16_t *tmp = e.mc_tmp_y;
		const int ts = e.mc_tmp_y_stride;
		interp_stage<8>(g, src - 3 * rs, rs, tmp, ts, xf, n, n + 7, 0, 1, 0);
		interp_stage<8>(g, tmp + 3 * ts, ts, pred, ps, yf, n, n, 1, 0, 1);
	}
}
template <class G>
HENC_HD void mc_chroma_interp(const G g, Enc &__restrict__ e, const int16_t *ref, int rs, int16_t *pred, int ps, int n, int mvx, int mvy)
{
	HENC_ENC_IN_LDS(e);
	const int xf = mvx & 7, yf = mvy & 7;
	const int16_t *src = ref + (mvy >> 3) * rs + (mvx >> 3);
	if (xf == 0) interp_stage<4>(g, src, rs, pred, ps, yf, n, n, 1, 1, 1);
	else if (yf == 0) interp_stage<4>(g, src, rs, pred, ps, xf, n, n, 0, 1, 1);
	else {
		int16_t *tmp = e.mc_tmp_c;
		interp_stage<4>(g, src - rs, rs, tmp, 40, xf, n, n + 3, 0, 1, 0);
		interp_stage<4>(g, tmp + 40, 40, pred, ps, yf, n, n, 1, 0, 1);
	}
}
#endif

// prediction of the node's three blocks for the vector mv (hmr_motion_compensation_luma / _chroma :1779-1907, uni-directional)
template <class G>
HENC_HD void motion_compensate_cu(const G g, Enc &__restrict__ e, int ni, MV mv)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[ni];
	const Seq &S = *e.seq;
	const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y, gxc = (e.ctu_x >> 1) + q.xc, gyc = (e.ctu_y >> 1) + q.yc;
#if defined(__HIPCC__)
	PRIM_T0();
	// (row-interleaved planes, k_subpel.hip: row y of phase f starts at (y * phases + f) * stride)
	const int sy = 16 * S.stride_y, sc = 64 * S.stride_c;
	const uint8_t *py = e.f->sub_y + (((mv.y & 3) << 2) | (mv.x & 3)) * S.stride_y + (ptrdiff_t)(gy + (mv.y >> 2)) * sy + gx + (mv.x >> 2);
	const ptrdiff_t oc = (((mv.y & 7) << 3) | (mv.x & 7)) * S.stride_c + (ptrdiff_t)(gyc + (mv.y >> 3)) * sc + gxc + (mv.x >> 3);
	if (q.size <= 32) {
		// every load of the three blocks is issued before the first store waits for one: the copy costs one trip to the planes, not three
		const int n = q.size, nc = q.size_chroma, lw = ilog2i(n) - 2, lc = ilog2i(nc) - 2, ychunks = (n * n) >> 2, cchunks = (nc * nc) >> 2;
		const uint8_t *pu = e.f->sub_c[0] + oc, *pv = e.f->sub_c[1] + oc;
		pred_t *dy = w.pred_y + q.y * 64 + q.x, *du = w.pred_c[0] + q.yc * 32 + q.xc, *dv = w.pred_c[1] + q.yc * 32 + q.xc;
		uint32_t vy[4] = {0, 0, 0, 0}, vu = 0, vv = 0;
		const int ic = g.tid, rc = ic >> lc, cc = (ic & ((1 << lc) - 1)) << 2;
		if (ic < cchunks) { vu = ld32u(pu + rc * sc + cc); vv = ld32u(pv + rc * sc + cc); }
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int i = g.tid + 64 * k;
			if (i < ychunks) vy[k] = ld32u(py + (i >> lw) * sy + ((i & ((1 << lw) - 1)) << 2));
		}
		// four prediction bytes per lane and step, as they are
		if (ic < cchunks) {
			*(uint32_t *)(du + rc * 32 + cc) = vu;
			*(uint32_t *)(dv + rc * 32 + cc) = vv;
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int i = g.tid + 64 * k;
			if (i < ychunks) *(uint32_t *)(dy + (i >> lw) * 64 + ((i & ((1 << lw) - 1)) << 2)) = vy[k];
		}
	} else {
		blk_from_u8(g, e.f->sub_c[0] + oc, sc, w.pred_c[0] + q.yc * 32 + q.xc, 32, q.size_chroma);
		blk_from_u8(g, e.f->sub_c[1] + oc, sc, w.pred_c[1] + q.yc * 32 + q.xc, 32, q.size_chroma);
		blk_from_u8(g, py, sy, w.pred_y + q.y * 64 + q.x, 64, q.size);
	}
	g.sync();
	PRIM_END(PP_INTERP);
#else
	mc_luma_interp(g, e, e.f->ref[0] + gy * S.stride_y + gx, S.stride_y, w.pred_y + q.y * 64 + q.x, 64, q.size, mv.x, mv.y);
	mc_chroma_interp(g, e, e.f->ref[1] + gyc * S.stride_c + gxc, S.stride_c, w.pred_c[0] + q.yc * 32 + q.xc, 32, q.size_chroma, mv.x, mv.y);
	mc_chroma_interp(g, e, e.f->ref[2] + gyc * S.stride_c + gxc, S.stride_c, w.pred_c[1] + q.yc * 32 + q.xc, 32, q.size_chroma, mv.x, mv.y);
#endif
}

// SADs of the source block at (ox, oy) of the CTU against the reference displaced by (qx[k], qy[k]) quarter samples from the co-located block at picture
// position (gx, gy), for the candidates with ok[k]: a round of the motion search in one go
template <int MAXC, class G>
HENC_HD void cand_sads(const G g, Enc &__restrict__ e, int ox, int oy, int gx, int gy, int size, const int (&qx)[MAXC], const int (&qy)[MAXC], const bool (&ok)[MAXC],
		       uint32_t (&out)[MAXC])
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
#if defined(__HIPCC__)
	const int sy = 16 * S.stride_y;      // (row-interleaved planes)
	const uint8_t *p0 = e.f->sub_y + (ptrdiff_t)gy * sy + gx;
	const uint8_t *cand[MAXC];
#pragma unroll
	for (int k = 0; k < MAXC; k++)
		cand[k] = ok[k] ? p0 + (((qy[k] & 3) << 2) | (qx[k] & 3)) * S.stride_y + (ptrdiff_t)(qy[k] >> 2) * sy + (qx[k] >> 2) : nullptr;
	{ PRIM_T0(); multi_sad_u8<MAXC>(g, e.w->curr_y + oy * 64 + ox, size, cand, sy, out); PRIM_END(PP_SAD); }
#else
	const int16_t *orig = e.w->curr_y + oy * 64 + ox, *ref = e.f->ref[0] + gy * S.stride_y + gx;
	for (int k = 0; k < MAXC; k++) {
		if (!ok[k]) { out[k] = 0; continue; }
		if (((qx[k] | qy[k]) & 3) == 0) out[k] = blk_sad(g, orig, CTU_STRIDE_Y, ref + (qy[k] >> 2) * S.stride_y + (qx[k] >> 2), S.stride_y, size);
		else {
			mc_luma_interp(g, e, ref, S.stride_y, e.w->pred_aux, 64, size, qx[k], qy[k]);   // scratch: no TU is in flight during the search
			out[k] = blk_sad(g, orig, CTU_STRIDE_Y, e.w->pred_aux, 64, size);
			g.sync();
		}
	}
#endif
}
// a[idx] with the array kept in registers (the indices of the search walk are data dependent)
template <int N>
HENC_INLINE uint32_t pick(const uint32_t (&a)[N], int idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// (the elements go through an empty asm: a plain chain of selects is turned back into a[idx], and an array indexed at run time lives in private memory;
	// the values are a search round's SADs - wavefront sums, uniform - hence scalar registers)
	uint32_t v[N];
#pragma unroll
	for (int k = 0; k < N; k++) { v[k] = a[k]; asm("" : "+s"(v[k])); }
	uint32_t r = v[0];
#pragma unroll
	for (int k = 1; k < N; k++) r = idx == k ? v[k] : r;
	return r;
#else
	uint32_t r = a[0];
	for (int k = 1; k < N; k++) r = idx == k ? a[k] : r;
	return r;
#endif
}

// ---- vector cost ----------------------------------------------------------------------------------------------------
// select_mv_candidate_fast :1004
HENC_INLINE uint32_t mv_cost_fast(const MvCandList &l, double corr, int mvx, int mvy, int *best_idx)
{
	uint32_t best = 0x7fffffff;
	int bi = 0;
	for (int i = 0; i < l.num; i++) {
		const double cx = corr * ((float)habs(l.mv[i].x - mvx)), cy = corr * ((float)habs(l.mv[i].y - mvy));
		const uint32_t c = (uint32_t)(cx + cy + .5);
		if (best > c) { best = c; bi = i; }
	}
	*best_idx = bi;
	return best;
}
// squareRoot :938: the reference's exponent-halving approximation on the float's bits
HENC_INLINE float bit_sqrt(float x)
{
	union { float f; uint32_t u; } v;
	v.f = x;
	v.u += 127u << 23;
	v.u >>= 1;
	return v.f;
}
// select_mv_candidate :975 (xCheckBestMVP)
HENC_INLINE uint32_t mv_cost_sqrt(const MvCandList &l, uint32_t qp, int mvx, int mvy, int *best_idx)
{
	uint32_t best = 0x7fffffff;
	int bi = 0;
	for (int i = 0; i < l.num; i++) {
		const double cx = (float)qp * bit_sqrt((float)habs(l.mv[i].x - mvx));
		const double cy = (float)qp * bit_sqrt((float)habs(l.mv[i].y - mvy));
		const uint32_t c = (uint32_t)(3. + cx + cy + .5);
		if (best > c) { best = c; bi = i; }
	}
	*best_idx = bi;
	return best;
}

// ---- hmr_motion_estimation :1404-1775 --------------------------------------------------------------------------------
// The source block sits at (ox, oy) of the CTU, (gx, gy) in the picture.  Returns the best SAD.
// The candidates of a search round do not depend on each other - only the comparisons do - so every round asks for all its SADs at once (cand_sads) and
// then walks them in the reference's order, with its loop bounds that move while the loop runs.
template <class G>
HENC_HD uint32_t motion_estimation(const G g, Enc &__restrict__ e, int ox, int oy, int gx, int gy, int size, const MvCandList &amvp, const MvCandList &search, double corr,
				   int action, MV *mv_io, MV *subpix_out)
{
	HENC_ENC_IN_LDS(e);
	static constexpr int ds[4][2] = {{-1, 0}, {0, -1}, {1, 0}, {0, 1}};
	static constexpr int db[8][2] = {{-2, 0}, {-1, -1}, {0, -2}, {1, -1}, {2, 0}, {1, 1}, {0, 2}, {-1, 1}};
	static constexpr int ref_h[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
	static constexpr int ref_q[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}};
	const int fw = e.seq->width, fh = e.seq->height;
	const int xlow = (gx - SEARCH_RANGE_X) < 0 ? -gx : -SEARCH_RANGE_X, xhigh = (gx + SEARCH_RANGE_X) > (fw - size) ? fw - gx - size : SEARCH_RANGE_X;
	const int ylow = (gy - SEARCH_RANGE_Y) < 0 ? -gy : -SEARCH_RANGE_Y, yhigh = (gy + SEARCH_RANGE_Y) > (fh - size) ? fh - gy - size : SEARCH_RANGE_Y;
	uint32_t cur_sad = 0, cur_rd = 0, best_sad = 0xffffffffu;
	int cur_x = 0, cur_y = 0, best_x = 0, best_y = 0, mvx = 0, mvy = 0, subx = 0, suby = 0, dummy;
#define HENC_IN_WIN(x, y) ((x) >= xlow && (x) <= xhigh && (y) >= ylow && (y) <= yhigh)
#define HENC_TRY(x, y, sad_, on_better)                                                            \
	do {                                                                                       \
		if (HENC_IN_WIN(x, y)) {                                                           \
			const uint32_t s_ = (sad_);                                                \
			const uint32_t rd_ = s_ + mv_cost_fast(amvp, corr, (x) << 2, (y) << 2, &dummy); \
			if (rd_ < cur_rd) { on_better; cur_sad = s_; cur_rd = rd_; cur_x = (x); cur_y = (y); } \
		}                                                                                  \
	} while (0)
	HENC_PROF_T0();
	if (action & ME_PEL) {
		bool early = false;
		cur_x = hclip(0, xlow, xhigh);
		cur_y = hclip(0, ylow, yhigh);
		{
			// the start position and the predictor positions
			int qx[4], qy[4];
			bool ok[4];
			uint32_t s4[4];
			qx[0] = cur_x << 2; qy[0] = cur_y << 2; ok[0] = true;
			for (int i = 0; i < 3; i++) {
				const int x = i < search.num ? search.mv[i].x >> 2 : 0, y = i < search.num ? search.mv[i].y >> 2 : 0;
				qx[i + 1] = x << 2; qy[i + 1] = y << 2;
				ok[i + 1] = i < search.num && !(x == 0 && y == 0) && HENC_IN_WIN(x, y);
			}
			cand_sads<4>(g, e, ox, oy, gx, gy, size, qx, qy, ok, s4);
			cur_sad = s4[0];
			cur_rd = cur_sad + mv_cost_fast(amvp, corr, cur_x << 2, cur_y << 2, &dummy);
			best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
			if (best_sad <= 0) early = true;
			if (!early) {
				for (int i = 0; i < search.num; i++) {
					const int x = search.mv[i].x >> 2, y = search.mv[i].y >> 2;
					if (x == 0 && y == 0) continue;
					HENC_TRY(x, y, pick(s4, i + 1), (void)0);
				}
				best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
				if (best_sad <= 0) early = true;
			}
		}
		if (!early) {
			int qx[4], qy[4];
			bool ok[4];
			uint32_t s4[4];
			for (int i = 0; i < 4; i++) {
				const int x = best_x + ds[i][0], y = best_y + ds[i][1];
				qx[i] = x << 2; qy[i] = y << 2; ok[i] = HENC_IN_WIN(x, y);
			}
			cand_sads<4>(g, e, ox, oy, gx, gy, size, qx, qy, ok, s4);
			for (int i = 0; i < 4; i++) {
				const int x = best_x + ds[i][0], y = best_y + ds[i][1];
				HENC_TRY(x, y, pick(s4, i), (void)0);
			}
			if (best_sad <= 0) early = true;
		}
		if (!early) {
			int dist = 2;
			const int end = (best_x != 0 && best_y != 0) ? 4 : 8;
			int next_start = 0, search_size = 8;
			best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
			while (dist < end) {
				// the eight directions at this distance around the (fixed) centre; the walk below visits a data-dependent subset of them
				int qx[8], qy[8];
				bool ok[8];
				uint32_t s8[8];
				for (int i = 0; i < 8; i++) {
					const int x = best_x + db[i][0] * dist, y = best_y + db[i][1] * dist;
					qx[i] = x << 2; qy[i] = y << 2; ok[i] = HENC_IN_WIN(x, y);
				}
				cand_sads<8>(g, e, ox, oy, gx, gy, size, qx, qy, ok, s8);
				for (int i = next_start; i < next_start + search_size; i++) {
					const int idx = i % 8, x = best_x + db[idx][0] * dist, y = best_y + db[idx][1] * dist;
					HENC_TRY(x, y, pick(s8, idx), (next_start = (idx - 2 + 8) % 8, search_size = 5));
				}
				dist *= 2;
			}
		}
		// last search: small-diamond descent
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		{
			int next_start = 0, search_size = 4;
			for (;;) {
				int qx[4], qy[4];
				bool ok[4];
				uint32_t s4[4];
				for (int i = 0; i < 4; i++) {
					const int x = best_x + ds[i][0], y = best_y + ds[i][1];
					qx[i] = x << 2; qy[i] = y << 2; ok[i] = HENC_IN_WIN(x, y);
				}
				cand_sads<4>(g, e, ox, oy, gx, gy, size, qx, qy, ok, s4);
				for (int i = next_start; i < next_start + search_size; i++) {
					const int idx = i % 4, x = best_x + ds[idx][0], y = best_y + ds[idx][1];
					HENC_TRY(x, y, pick(s4, idx), (next_start = (idx - 1 + 4) % 4, search_size = 3));
				}
				if (best_x == cur_x && best_y == cur_y) break;
				best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
			}
		}
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		mvx = best_x << 2; mvy = best_y << 2;
	} else {
		mvx = mv_io->x; mvy = mv_io->y;
	}
	HENC_PROF_ADD(e, PF_ME_INT);
	if (action & ME_HALF) {
		int bidx = 0, bx = 0, by = 0;
		best_x = mvx >> 2; best_y = mvy >> 2;
		int qx[9], qy[9];
		bool ok[9];
		uint32_t sads[9];
		for (int i = 0; i < 9; i++) { qx[i] = (best_x << 2) + ref_h[i][0] * 2; qy[i] = (best_y << 2) + ref_h[i][1] * 2; ok[i] = true; }
		cand_sads<9>(g, e, ox, oy, gx, gy, size, qx, qy, ok, sads);
		if (!(action & ME_PEL)) cur_sad = sads[0];     // candidate 0 of the round is the integer position itself
		for (int i = 0; i < 9; i++) {
			const int cx = ref_h[i][0] * 2, cy = ref_h[i][1] * 2;
			const uint32_t s = pick(sads, i);
			if (s < cur_sad) { cur_sad = s; bx = cx; by = cy; bidx = i; }
		}
		mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
		best_sad = cur_sad;
		if (action & ME_QUARTER) {
			const int hx = ref_h[bidx][0], hy = ref_h[bidx][1];
			bx = hx * 2; by = hy * 2;
			for (int i = 0; i < 9; i++) { qx[i] = (best_x << 2) + hx * 2 + ref_q[i][0]; qy[i] = (best_y << 2) + hy * 2 + ref_q[i][1]; }
			cand_sads<9>(g, e, ox, oy, gx, gy, size, qx, qy, ok, sads);
			for (int i = 0; i < 9; i++) {
				const int cx = hx * 2 + ref_q[i][0], cy = hy * 2 + ref_q[i][1];
				const uint32_t s = pick(sads, i);
				if (s < cur_sad) { cur_sad = s; bx = cx; by = cy; }
			}
			best_sad = cur_sad;
			mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
		}
	}
#undef HENC_IN_WIN
#undef HENC_TRY
	HENC_PROF_ADD(e, PF_ME_SUB);   // includes the integer part; the report subtracts
	mv_io->x = mvx; mv_io->y = mvy;
	subpix_out->x = subx; subpix_out->y = suby;
	return best_sad;
}

// ---- candidate derivation ---------------------------------------------------------------------------------------------
struct CornerNodes { int lb, tl, tr; };
HENC_INLINE CornerNodes corner_nodes(Enc &__restrict__ e, int ni)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	const int np = q.size >> 2, base = cfg_depth_start(CFG_MAX_CU_DEPTH);
	CornerNodes c;
	c.lb = base + raster2abs(q.raster_index + 16 * (np - 1));
	c.tl = base + raster2abs(q.raster_index);
	c.tr = base + raster2abs(q.raster_index + np - 1);
	return c;
}
// What the candidate derivations read of a neighbouring unit, fetched in one go.  The side-info records live in HBM (the CTU's own too): read field by field as
// the reference's logic asks for them, a derivation was five or six dependent trips to L2 (4800 cycles per call in the profile); the derivations below fetch the
// units they are certain to look at side by side, before the logic starts.  A missing neighbour reads the CTU's own first unit (a valid address) and is marked.
struct NbUnit {
	int there, pred_mode, inter_mode, ref_idx;
	MV mv;
};
HENC_INLINE NbUnit nb_unit(Enc &__restrict__ e, const CtuPublic *c, uint32_t idx)
{
	HENC_ENC_IN_LDS(e);
	const CtuPublic *p = c ? c : e.ctu;
	const uint32_t i = c ? idx : 0;
	NbUnit u;
	u.pred_mode = p->pred_mode[i];
	u.inter_mode = p->inter_mode[i];
	u.ref_idx = p->mv_ref_idx[i];
	u.mv = p->mv_ref[i];
	u.there = c != nullptr;
	return u;
}
HENC_INLINE int equal_motion(const NbUnit &a, const NbUnit &b)      // equal_motion :1913 on fetched units
{
	if (a.inter_mode != b.inter_mode) return 0;
	if (a.inter_mode & 1) {
		if (a.mv.x != b.mv.x || a.mv.y != b.mv.y || a.ref_idx != b.ref_idx) return 0;
	}
	return 1;
}

// add_amvp_cand :2182 (one reference picture: list 0 index 0 is the only picture a neighbour can point to; the scaled variant
// add_amvp_cand_order :2229 then adds the same unscaled vector under the same condition)
HENC_INLINE int add_amvp_cand(MvCandList &l, const NbUnit &u)
{
	if (u.there && u.ref_idx >= 0) {
		l.mv[l.num++] = u.mv;
		return 1;
	}
	return 0;
}
// get_amvp_candidates :2342.  The five neighbour units are fetched together (the two copies of the CU's corner flags the look-ups need are unconditional in the
// reference too, and nothing in between reads them), then the reference's order of questions runs on the fetched values.
HENC_INLINE void get_amvp_candidates(Enc &__restrict__ e, int ni, MvCandList &l)
{
	HENC_ENC_IN_LDS(e);
	const CornerNodes cn = corner_nodes(e, ni);
	uint32_t idx_lb = 0, idx_l = 0, idx_tr = 0, idx_t = 0, idx_tl = 0;
	l.num = 0;
	const uint8_t has_lb = node_of(e, ni).left_bottom_nb, has_tr = node_of(e, ni).top_right_nb;
	corner_set_left_bottom(e, cn.lb, has_lb);
	corner_set_top_right(e, cn.tr, has_tr);
	CtuPublic *c_lb = pu_left_bottom(e, cn.lb, has_lb, &idx_lb);
	CtuPublic *c_l = pu_left(e, cn.lb, &idx_l);
	CtuPublic *c_tr = pu_top_right(e, cn.tr, has_tr, &idx_tr);
	CtuPublic *c_t = pu_top(e, cn.tr, &idx_t, 0);
	CtuPublic *c_tl = pu_top_left(e, cn.tl, &idx_tl);
	const NbUnit u_lb = nb_unit(e, c_lb, idx_lb), u_l = nb_unit(e, c_l, idx_l), u_tr = nb_unit(e, c_tr, idx_tr), u_t = nb_unit(e, c_t, idx_t), u_tl = nb_unit(e, c_tl, idx_tl);
	int added_smvp = u_lb.there && u_lb.pred_mode != PM_INTRA;
	if (!added_smvp) added_smvp = u_l.there && u_l.pred_mode != PM_INTRA;
	int added = add_amvp_cand(l, u_lb);
	if (!added) added = add_amvp_cand(l, u_l);
	if (!added) {
		added = add_amvp_cand(l, u_lb);
		if (!added) added = add_amvp_cand(l, u_l);
	}
	added = add_amvp_cand(l, u_tr);
	if (!added) added = add_amvp_cand(l, u_t);
	if (!added) added = add_amvp_cand(l, u_tl);
	if (!added_smvp) {
		added = add_amvp_cand(l, u_tr);
		if (!added) added = add_amvp_cand(l, u_t);
		if (!added) added = add_amvp_cand(l, u_tl);
	}
	if (l.num == 2 && l.mv[0].x == l.mv[1].x && l.mv[0].y == l.mv[1].y) l.num = 1;
	if (l.num > 2) l.num = 2;
	while (l.num < 2) {
		l.mv[l.num].x = 0;
		l.mv[l.num].y = 0;
		l.num++;
	}
}

// equal_motion :1913 (list 0 only carries vectors in P slices)
HENC_INLINE int equal_motion(const CtuPublic *a, uint32_t ia, const CtuPublic *b, uint32_t ib)
{
	if (a->inter_mode[ia] != b->inter_mode[ib]) return 0;
	if (a->inter_mode[ia] & 1) {
		if (a->mv_ref[ia].x != b->mv_ref[ib].x || a->mv_ref[ia].y != b->mv_ref[ib].y || a->mv_ref_idx[ia] != b->mv_ref_idx[ib]) return 0;
	}
	return 1;
}
// get_merge_mvp_candidates :1937, P slice.  inter_modes[k] = inter_mode of candidate k's source unit.
HENC_INLINE void get_merge_candidates(Enc &__restrict__ e, int ni, MvCandList &l, uint8_t *inter_modes)
{
	HENC_ENC_IN_LDS(e);
	const int max_cand = CFG_NUM_MERGE_CAND;
	const CornerNodes cn = corner_nodes(e, ni);
	uint32_t i_l = 0, i_t = 0, i_tr = 0, i_lb = 0, i_tl = 0;
	int cnt = 0;
	for (int k = 0; k < max_cand; k++) l.ref_idx[k] = -1;
	l.num = 0;
	// A1 (left) and B1 (top) are always looked at: both fetched before the logic
	CtuPublic *c_l = pu_left(e, cn.lb, &i_l);
	CtuPublic *c_t = pu_top(e, cn.tr, &i_t, 0);
	const NbUnit u_l = nb_unit(e, c_l, i_l), u_t = nb_unit(e, c_t, i_t);
	const int a1 = u_l.there && u_l.pred_mode != PM_INTRA;
	if (a1) {
		inter_modes[cnt] = (uint8_t)u_l.inter_mode;
		l.mv[cnt] = u_l.mv;
		l.ref_idx[cnt] = u_l.ref_idx;
		cnt++;
	}
	if (cnt >= max_cand) { l.num = cnt; return; }
	const int b1 = u_t.there && u_t.pred_mode != PM_INTRA;
	if (b1 && (!a1 || !equal_motion(u_l, u_t))) {
		inter_modes[cnt] = (uint8_t)u_t.inter_mode;
		l.mv[cnt] = u_t.mv;
		l.ref_idx[cnt] = u_t.ref_idx;
		cnt++;
	}
	if (cnt >= max_cand) { l.num = cnt; return; }
	const uint8_t has_tr = node_of(e, ni).top_right_nb;
	corner_set_top_right(e, cn.tr, has_tr);
	CtuPublic *c_tr = pu_top_right(e, cn.tr, has_tr, &i_tr);
	const int b0 = c_tr && c_tr->pred_mode[i_tr] != PM_INTRA;
	if (b0 && (!b1 || !equal_motion(c_t, i_t, c_tr, i_tr))) {
		inter_modes[cnt] = c_tr->inter_mode[i_tr];
		l.mv[cnt] = c_tr->mv_ref[i_tr];
		l.ref_idx[cnt] = c_tr->mv_ref_idx[i_tr];
		cnt++;
	}
	if (cnt >= max_cand) { l.num = cnt; return; }
	const uint8_t has_lb = node_of(e, ni).left_bottom_nb;
	corner_set_left_bottom(e, cn.lb, has_lb);
	CtuPublic *c_lb = pu_left_bottom(e, cn.lb, has_lb, &i_lb);
	const int a0 = c_lb && c_lb->pred_mode[i_lb] != PM_INTRA;
	if (a0 && (!a1 || !equal_motion(c_l, i_l, c_lb, i_lb))) {
		inter_modes[cnt] = c_lb->inter_mode[i_lb];
		l.mv[cnt] = c_lb->mv_ref[i_lb];
		l.ref_idx[cnt] = c_lb->mv_ref_idx[i_lb];
		cnt++;
	}
	if (cnt >= max_cand) { l.num = cnt; return; }
	if (cnt < 4) {
		CtuPublic *c_tl = pu_top_left(e, cn.tl, &i_tl);
		const int b2 = c_tl && c_tl->pred_mode[i_tl] != PM_INTRA;
		if (b2 && (!a1 || !equal_motion(c_l, i_l, c_tl, i_tl)) && (!b1 || !equal_motion(c_t, i_t, c_tl, i_tl))) {
			inter_modes[cnt] = c_tl->inter_mode[i_tl];
			l.mv[cnt] = c_tl->mv_ref[i_tl];
			l.ref_idx[cnt] = c_tl->mv_ref_idx[i_tl];
			cnt++;
		}
	}
	if (cnt >= max_cand) { l.num = cnt; return; }
	int addr = cnt;
	while (addr < max_cand) {       // zero candidates (one reference picture: index 0)
		inter_modes[addr] = 1;
		l.mv[addr].x = 0;
		l.mv[addr].y = 0;
		l.ref_idx[addr] = 0;
		addr++;
	}
	l.num = addr;
}

// ---- inter TUs ---------------------------------------------------------------------------------------------------------
// encode_inter_cu :40 (comp 0) / encode_inter_cu_chroma :133: DCT + quant, keep-or-drop decision in the residual domain, reconstruction
template <class G>
HENC_HD uint32_t encode_inter_tu(const G g, Enc &__restrict__ e, int ni, int comp, int depth, int part_size_type, int *curr_sum, uint32_t *raw_ssq = nullptr, int scratch_off = 0)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	Node &nd = node_of(e, ni);
	const int original_depth = e.geo[ni].depth;
	const int pi = (comp == COMP_Y || e.geo[ni].size_chroma != 2) ? ni : e.geo[ni].parent;     // 2x2 chroma: coded as the parent's 4x4
	const Geo &q = e.geo[pi];
	const int is_y = comp == COMP_Y;
	const int curr_depth = q.depth, n = is_y ? q.size : q.size_chroma, x = is_y ? q.x : q.xc, y = is_y ? q.y : q.yc;
	const int cs = ctu_stride(comp), ds = dec_stride(comp);
	const int wnd = original_depth + 1 + (part_size_type != PART_2Nx2N);
	const int qp = is_y ? (int)nd.qp : chroma_qp_table((int)nd.qp + e.seq->chroma_qp_offset);
	const int per = qp / 6, rem = qp % 6;
	const double weight = is_y ? 1.0 : e.f->chroma_weight;
	const int off = is_y ? (q.abs_index << 4) : ((q.abs_index << 4) >> 2);
	const pred_t *pred = pred_ptr(w, comp) + y * cs + x;
	const src_t *orig = curr_ptr(w, comp) + y * cs + x;      // (the residual source - prediction is formed where it is read: the worker keeps no residual window)
	// (scratch_off: the group's share of the scratch buffers when two groups of a wavefront run a TU each - PairGrp, both chroma planes of a small TU on the helper)
	int16_t *quant = tq_ptr(w, wnd, comp) + off, *iquant = iq_slot(w, comp, off, e.on_helper) + scratch_off;
	int16_t *const scratch_a = e.scratch_a + scratch_off, *const scratch_b = e.scratch_b + scratch_off;
	int16_t *dec = dec_ptr(w, wnd, comp) + y * ds + x;
	// The chain runs in the worker's fast memory: coefficients in scratch_a, rounding remainders in scratch_b, the levels in the block's slot of the
	// dequantised-coefficient buffer (dequantised in place afterwards), the reconstructed residual in scratch_b (the reference's separate window, which
	// nothing else reads); only the final levels and the reconstruction go to the windows in HBM.
	int16_t *rdec = scratch_b;
	tr_forward(g, HENC_FT(e), e.T, orig, cs, pred, cs, scratch_a, scratch_b, n, 0);
	int sum = quantize(g, HENC_FT(e), e.T, scratch_a, iquant, scratch_b, SCAN_DIAG, curr_depth, comp, 0, e.f->slice_type == SLICE_I, e.seq->sign_hiding, n, per, rem);
	nd.inter_cbf[comp] = (sum ? 1 : 0) << (original_depth - depth);
	if (is_y) nd.inter_tr_idx = original_depth - depth;
	uint32_t ssd;
	const bool coded = sum > 0;
	uint32_t raw_zero = 0;
	if (coded) {
		lin_copy_nosync(g, iquant, quant, n * n);
		raw_zero = blk_ssd(g, orig, cs, pred, cs, n);
		if (raw_ssq) *raw_ssq = raw_zero;
		dequantize(g, HENC_FT(e), e.T, iquant, iquant, curr_depth, comp, 0, n, per, rem);
	}
#if defined(HENC_MFMA_TRANSFORM) && !defined(HENC_MFMA_NO_PAIR)
	// two halves of a wavefront with a block each: their inverse transforms share a matrix-core tile, which all lanes have to run together - outside the halves' own branches
	if constexpr (G::n == 32) {
		if (__ballot(coded) != 0 && n <= 8) tr_inverse_pair(g, coded, e.T, rdec, n, iquant, n);
		else if (coded) tr_inverse(g, HENC_FT(e), e.T, rdec, n, iquant, scratch_a, n, 0);
	} else
#endif
	if (coded) tr_inverse(g, HENC_FT(e), e.T, rdec, n, iquant, scratch_a, n, 0);
	if (coded) {
		const uint32_t raw = blk_ssd_diff(g, orig, cs, pred, cs, rdec, n, n);
		uint32_t ssd_zero;
		if (is_y) { ssd_zero = raw_zero; ssd = raw; }
		else { ssd_zero = (uint32_t)(weight * raw_zero); ssd = (uint32_t)(weight * raw); }
		const double thr = hclip(e.f->avg_dist / 2.5 - 5., 1., 20000.);
		if (is_y ? ((double)ssd_zero <= (double)(int)ssd + thr * sum) : ((double)ssd_zero <= (double)ssd + thr * sum)) {
			lin_zero(g, quant, n * n);
			sum = 0;
			nd.inter_cbf[comp] = 0;
			blk_reconst(g, pred, cs, (const int16_t *)nullptr, 0, dec, ds, n);
		} else {
			blk_reconst(g, pred, cs, rdec, n, dec, ds, n);
		}
	} else {
		lin_zero_nosync(g, quant, n * n);
		const uint32_t raw = blk_ssd(g, orig, cs, pred, cs, n);
		if (raw_ssq) *raw_ssq = raw;
		ssd = is_y ? raw : (uint32_t)(weight * raw);
		blk_reconst(g, pred, cs, (const int16_t *)nullptr, 0, dec, ds, n);
	}
	*curr_sum = sum;      // the caller adds the three components up (nd.sum; the reference accumulates it here, :128,:222)
	return ssd;
}

// the three components of one TU, one after the other
template <class G>
HENC_HD void inter_tu_all_comps(const G g, Enc &__restrict__ e, int curr, int depth, int part_size_type, int has_chroma, uint32_t *dist, int *sums)
{
	HENC_ENC_IN_LDS(e);
	dist[0] = encode_inter_tu(g, e, curr, COMP_Y, depth, part_size_type, &sums[0]);
	dist[1] = dist[2] = 0;
	sums[1] = sums[2] = 0;
	if (has_chroma) {
		dist[1] = encode_inter_tu(g, e, curr, COMP_U, depth, part_size_type, &sums[1]);
		dist[2] = encode_inter_tu(g, e, curr, COMP_V, depth, part_size_type, &sums[2]);
	}
	g.sync();
}

// SET_ENC_INFO_BUFFS :2451
template <class G>
HENC_HD void set_enc_info_buffs(const G g, Enc &__restrict__ e, int ni, int depth)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	Work &w = *e.w;
	for (int i = g.tid; i < q.num_part; i += g.n) {
		w.cbf_buffs[COMP_Y][depth][q.abs_index + i] = (uint8_t)nd.inter_cbf[COMP_Y];
		w.cbf_buffs[COMP_U][depth][q.abs_index + i] = (uint8_t)nd.inter_cbf[COMP_U];
		w.cbf_buffs[COMP_V][depth][q.abs_index + i] = (uint8_t)nd.inter_cbf[COMP_V];
		w.tr_idx_buffs[depth][q.abs_index + i] = (uint8_t)nd.inter_tr_idx;
	}
	g.sync();
}

// encode_inter :3071 - the transform tree of an inter CU; referenced by the prediction depth
template <class G>
HENC_WALK_FN HENC_HD uint32_t encode_inter(const G g, Enc &__restrict__ e, int depth, int part_position, int part_size_type)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	Work &w = *e.w;
#if !defined(__HIPCC__) && defined(HENC_TRACE_ENABLE)
	if (henc_trace_file) {      // what the prediction window holds for this CU when the evaluation starts (oracle/ref_ctudump.c prints the same sums)
		const Geo &tq = e.geo[node_at(e, depth, part_position)];
		unsigned a[3] = {0, 0, 0};
		for (int c = 0; c < 3; c++) {
			const int n = c ? tq.size_chroma : tq.size, st = c ? 32 : 64;
			const pred_t *p = c ? w.pred_c[c - 1] + tq.yc * 32 + tq.xc : w.pred_y + tq.y * 64 + tq.x;
			for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) a[c] += (unsigned)(p[y * st + x] & 0xffff) * (unsigned)(1 + ((x + 3 * y) & 7));
		}
		HENC_TRACE("EIN ctu=%d d=%d abs=%d pred=%u,%u,%u\n", e.ctu->ctu_number, depth, tq.abs_index, a[0], a[1], a[2]);
	}
#endif
	const int nxn = part_size_type != PART_2Nx2N;
	int parent, curr, initial_state, end_state;
	uint32_t qp;
	if (depth == 0 && CFG_MAX_CU_SIZE == 64) {
		parent = cfg_depth_start(0);
		curr = e.geo[parent].child[0];
		node_of(e, parent).cost = 0x7fffffff;
		initial_state = part_position & 3;
		end_state = initial_state;
		qp = node_of(e, parent).qp;
	} else {
		curr = node_at(e, depth, part_position);
		parent = e.geo[curr].parent;
		initial_state = part_position & 3;
		end_state = initial_state + 1;
		qp = node_of(e, curr).qp;
	}
	int curr_depth = e.geo[curr].depth;
	const int log2cu_size = CFG_MAX_CU_SHIFT - depth;
	const int one_level_nxn = (S.max_inter_tr_depth == 1 && nxn);
	int cu_min_tu_size_shift;
	if (log2cu_size < S.min_tu_size_shift + S.max_inter_tr_depth - 1 + one_level_nxn) cu_min_tu_size_shift = S.min_tu_size_shift;
	else {
		cu_min_tu_size_shift = log2cu_size - (S.max_inter_tr_depth - 1 + one_level_nxn);
		if (cu_min_tu_size_shift > S.max_tu_size_shift) cu_min_tu_size_shift = S.max_tu_size_shift;
	}
	int max_tr_processing_depth = CFG_MAX_CU_SHIFT - cu_min_tu_size_shift;
	if (S.perf_mode >= 1) max_tr_processing_depth = depth == 0 ? 1 : (depth + (nxn ? 1 : 0));
	if (S.max_inter_tr_depth == 1 && nxn && curr_depth == depth && log2cu_size > max_tr_processing_depth) {
		parent = curr;
		curr = e.geo[parent].child[0];
		node_of(e, parent).distortion = node_of(e, parent).cost = MAX_COST;
		initial_state = part_position & 3;
		end_state = initial_state;
	}
	DepthState depth_state;
	depth_state.set(curr_depth, initial_state);
	curr_depth = e.geo[curr].depth;
	int curr_sum_y = 0, curr_sum_u = 0, curr_sum_v = 0;
	// the squared residual of every component over the CU's TUs - the distortion of coding nothing, which the TU decisions compute anyway and the merge evaluation's
	// no-residual pass would compute again as SSD(source, prediction) (check_rd_cost_merge)
	e.inter_ssq[0] = e.inter_ssq[1] = e.inter_ssq[2] = 0;
	e.inter_ssq_valid = 1;
	int luma_covered = 0;      // (a tree that visits a parent AND its children would count samples twice: checked at the end)
	while (curr_depth != depth || depth_state.get(curr_depth) != end_state) {
		curr = parent < 0 ? curr : e.geo[parent].child[depth_state.get(curr_depth)];
		if (e.geo[curr].depth >= 1) nodes_select_quad(g, e, e.geo[curr].abs_index >> 6);      // (the transform tree of the 64 x 64 CU goes through all four quadrants)
		Node &cn = node_of(e, curr);
		cn.qp = qp;
		curr_depth = e.geo[curr].depth;
		uint32_t dist_y, dist_u, dist_v;
		const bool has_chroma = e.geo[curr].size_chroma != 2 || depth_state.get(curr_depth) == 0;
		if (has_chroma && HENC_HELPERS(e)) {
			// the three components of a TU are independent: the helpers take U and V
			uint32_t raw = 0;
			if (NHELP >= 2) {
				helper_post(g, e, 0, HJOB_INTER_TU, curr, COMP_U, depth, part_size_type);
				helper_post(g, e, NHELP - 1, HJOB_INTER_TU, curr, COMP_V, depth, part_size_type);
			} else helper_post(g, e, 0, HJOB_INTER_TU, curr, COMP_UV, depth, part_size_type);
			dist_y = encode_inter_tu(g, e, curr, COMP_Y, depth, part_size_type, &curr_sum_y, &raw);
			for (int h = 0; h < NHELP; h++) helper_wait(g, e, h);
			const uint32_t *ru = e.box->r[0], *rv = NHELP >= 2 ? e.box->r[NHELP - 1] : e.box->r[0] + 3;
			dist_u = ru[0]; curr_sum_u = (int)ru[1];
			dist_v = rv[0]; curr_sum_v = (int)rv[1];
			e.inter_ssq[0] += raw; e.inter_ssq[1] += ru[2]; e.inter_ssq[2] += rv[2];
		} else {
			uint32_t raw = 0;
			dist_y = encode_inter_tu(g, e, curr, COMP_Y, depth, part_size_type, &curr_sum_y, &raw);
			e.inter_ssq[0] += raw;
			dist_u = dist_v = 0;
		}
		if (has_chroma && !HENC_HELPERS(e)) {
			uint32_t raw_u = 0, raw_v = 0;
			dist_u = encode_inter_tu(g, e, curr, COMP_U, depth, part_size_type, &curr_sum_u, &raw_u);
			dist_v = encode_inter_tu(g, e, curr, COMP_V, depth, part_size_type, &curr_sum_v, &raw_v);
			e.inter_ssq[1] += raw_u; e.inter_ssq[2] += raw_v;
		} else if (!has_chroma) {
			dist_u = dist_v = 0;
			cn.inter_cbf[COMP_U] = node_of(e, curr - 1).inter_cbf[COMP_U];
			cn.inter_cbf[COMP_V] = node_of(e, curr - 1).inter_cbf[COMP_V];
		}
		luma_covered += e.geo[curr].size * e.geo[curr].size;
		cn.distortion = dist_y + dist_u + dist_v;
		cn.cost = cn.distortion;
		cn.sum = (uint32_t)(curr_sum_y + curr_sum_u + curr_sum_v);
		depth_state.inc(curr_depth);
		if (curr_depth < max_tr_processing_depth) {
			curr_depth++;
			parent = curr;
		} else if (depth_state.get(curr_depth) == 4) {
			while (depth_state.get(curr_depth) == 4 && curr_depth > depth) {
				const Geo &pq = e.geo[parent];
				Node &pn = node_of(e, parent);
				Node &c0 = node_of(e, pq.child[0]), &c1 = node_of(e, pq.child[1]), &c2 = node_of(e, pq.child[2]), &c3 = node_of(e, pq.child[3]);
				const double distortion = (double)c0.distortion + c1.distortion + c2.distortion + c3.distortion;
				const uint32_t sum = c0.sum + c1.sum + c2.sum + c3.sum;
				const double cost = distortion;
				const int buff_depth = depth + nxn;
				depth_state.set(curr_depth, 0);
				if (cost < pn.cost) {
					pn.cost = (uint32_t)cost;
					pn.distortion = (uint32_t)distortion;
					pn.sum = sum;
					const int tr_mask = 1 << (curr_depth - depth);
					if (curr_depth == max_tr_processing_depth) {
						uint32_t sp[3];
						for (int c = 0; c < 3; c++)
							sp[c] = ((c0.inter_cbf[c] & tr_mask) | (c1.inter_cbf[c] & tr_mask) | (c2.inter_cbf[c] & tr_mask) | (c3.inter_cbf[c] & tr_mask)) >> 1;
						for (int k = 0; k < 4; k++) {
							Node &ck = node_of(e, pq.child[k]);
							ck.inter_cbf[0] |= (int32_t)sp[0];
							ck.inter_cbf[1] |= (int32_t)sp[1];
							ck.inter_cbf[2] |= (int32_t)sp[2];
							set_enc_info_buffs(g, e, pq.child[k], buff_depth);
						}
					} else {
						uint32_t cb[3];
						for (int c = 0; c < 3; c++) {
							const uint8_t *b = w.cbf_buffs[c][buff_depth];
							cb[c] = ((b[e.geo[pq.child[0]].abs_index] & tr_mask) | (b[e.geo[pq.child[1]].abs_index] & tr_mask) | (b[e.geo[pq.child[2]].abs_index] & tr_mask) |
								 (b[e.geo[pq.child[3]].abs_index] & tr_mask)) >> 1;
						}
						g.sync();
						for (int i = g.tid; i < pq.num_part; i += g.n)
							for (int c = 0; c < 3; c++) w.cbf_buffs[c][buff_depth][pq.abs_index + i] |= (uint8_t)cb[c];
						g.sync();
					}
					sync_motion_buffers(g, e, parent, curr_depth + 1 + nxn, curr_depth + nxn, curr_depth + 1 + nxn, curr_depth + nxn);
				} else {
					set_enc_info_buffs(g, e, parent, buff_depth);
				}
				curr_depth--;
				parent = e.geo[parent].parent;
			}
		}
	}
	const int top = node_at(e, depth, part_position);
	if (luma_covered != e.geo[top].size * e.geo[top].size) e.inter_ssq_valid = 0;
	if (depth == max_tr_processing_depth) set_enc_info_buffs(g, e, top, depth + nxn);
	{
		const Geo &tq = e.geo[top];
		const int8_t ri = (int8_t)node_of(e, top).inter_ref_index;
		for (int i = g.tid; i < tq.num_part; i += g.n) e.ctu->mv_ref_idx[tq.abs_index + i] = ri;
		g.sync();
	}
	return node_of(e, top).cost;
}

// SET_INTER_MV_BUFFS :2460 + the reference-index memsets that follow it in predict_inter :3042-3044
template <class G>
HENC_HD void set_inter_mv_buffs(const G g, Enc &__restrict__ e, int ni)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	for (int i = g.tid; i < q.num_part; i += g.n) {
		e.ctu->mv_ref[q.abs_index + i] = nd.inter_mv;
		e.ctu->mv_ref_idx[q.abs_index + i] = (int8_t)nd.inter_ref_index;
	}
	g.sync();
}

// predict_inter :2924 (uni-directional): vector predictor choice, motion compensation, residual.  Returns the vector cost.
template <class G>
HENC_HD int predict_inter(const G g, Enc &__restrict__ e, int depth, int part_position, int part_size_type)
{	depth = uni(depth); part_position = uni(part_position); part_size_type = uni(part_size_type);

	HENC_ENC_IN_LDS(e);
	int curr = node_at(e, depth, part_position), num_partitions = 1;
	if (part_size_type == PART_NxN) {
		curr = e.geo[e.geo[curr].parent].child[0];
		num_partitions = 4;
	}
	int mv_cost = 0;
	for (int np = 0; np < num_partitions; np++, curr++) {
		Node &nd = node_of(e, curr);
		const MV mv = nd.inter_mv;
		// (the list motion estimation has just derived for this node is still in place: nothing in between changes what it is derived from)
		if (e.amvp_node != curr) { PRIM_T0(); get_amvp_candidates(e, curr, e.w->amvp); PRIM_END(PP_CAND); }
		e.amvp_node = -1;
		int best_idx = 0;
		mv_cost += (int)mv_cost_sqrt(e.w->amvp, nd.qp, mv.x, mv.y, &best_idx);
		nd.best_candidate_idx = (int8_t)best_idx;
		nd.best_dif_mv.x = mv.x - e.w->amvp.mv[nd.best_candidate_idx].x;
		nd.best_dif_mv.y = mv.y - e.w->amvp.mv[nd.best_candidate_idx].y;
		set_inter_mv_buffs(g, e, curr);
		motion_compensate_cu(g, e, curr, mv);      // (the residual the reference writes here, :3046-3055, is formed by the TUs that read it)
	}
	return mv_cost;
}

// hmr_cu_motion_estimation :2471 (list 0, one reference).  Returns SAD + vector cost.
template <class G>
HENC_HD int cu_motion_estimation(const G g, Enc &__restrict__ e, int depth, int part_position, int part_size_type, int action)
{	depth = uni(depth); part_position = uni(part_position); part_size_type = uni(part_size_type); action = uni(action);

	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Seq &S = *e.seq;
	int curr = node_at(e, depth, part_position), num_partitions = 1;
	if (part_size_type == PART_NxN) {
		curr = e.geo[e.geo[curr].parent].child[0];
		num_partitions = 4;
	}
	uint32_t sad = 0, mv_total_cost = 0;
	for (int np = 0; np < num_partitions; np++, curr++) {
		const Geo &q = e.geo[curr];
		Node &nd = node_of(e, curr);
		const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y;
		MvCandList &amvp = w.amvp;      // (built where predict_inter will read it: as a local its dynamically indexed lists lived in private memory)
		{ PRIM_T0(); get_amvp_candidates(e, curr, amvp); PRIM_END(PP_CAND); }
		w.search_cands.num = 0;
		for (int i = 0; i < amvp.num; i++)
			if (amvp.mv[i].x != 0 && amvp.mv[i].y != 0) w.search_cands.mv[w.search_cands.num++] = amvp.mv[i];
		if (q.parent >= 0) {
			const MV pm = node_of(e, q.parent).inter_mv;
			if (pm.x != 0 && pm.y != 0) w.search_cands.mv[w.search_cands.num++] = pm;
		}
		MV mv = {0, 0}, subpix = {0, 0};
		if ((action & (ME_HALF | ME_QUARTER)) && !(action & ME_PEL)) {
			mv = nd.inter_mv;
			subpix = nd.subpix_mv;
		}
		const double corr = calc_mv_correction(nd.qp, e.f->avg_dist);
		const uint32_t cost = motion_estimation(g, e, q.x, q.y, gx, gy, q.size, amvp, w.search_cands, corr, action, &mv, &subpix);
		int mvp_idx = 0;
		const uint32_t mv_cost = mv_cost_fast(amvp, corr, mv.x, mv.y, &mvp_idx);
		nd.subpix_mv = subpix;
		// best_cost starts at MAX_COST with a zero vector cost, so the single reference always wins (:2652)
		nd.inter_mv = mv;
		nd.inter_ref_index = 0;
		nd.inter_mode = 1;
		e.amvp_node = curr;      // (predict_inter, if it follows for this node, finds the list in w.amvp)
		nd.best_candidate_idx = mvp_idx;
		nd.best_dif_mv.x = mv.x - amvp.mv[mvp_idx].x;
		nd.best_dif_mv.y = mv.y - amvp.mv[mvp_idx].y;
		sad += cost;
		mv_total_cost += mv_cost;
		nd.inter_mode = 1;
		if (part_size_type == PART_NxN) {
			set_inter_mv_buffs(g, e, curr);
			for (int i = g.tid; i < q.num_part; i += g.n) {
				e.ctu->inter_mode[q.abs_index + i] = (uint8_t)nd.inter_mode;
				e.ctu->pred_mode[q.abs_index + i] = PM_INTER;
			}
			g.sync();
		}
	}
	return (int)(sad + mv_total_cost);
}

}  // namespace henc
