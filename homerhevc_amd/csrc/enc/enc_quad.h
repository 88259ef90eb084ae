// The merge evaluation of an 8 x 8 CU with all its candidates side by side (device only).
//
// check_rd_cost_merge_2nx2n (hmr_motion_inter.c:3493-3742) evaluates up to five candidate vectors one after the other: motion compensation, then encode_inter
// (:3071: forward transform, quantisation + sign hiding, and for blocks with levels dequantisation, inverse transform, the keep-or-drop test of
// encode_inter_cu :40-230, reconstruction) for the three components.  What an evaluation computes is a pure function of the source block, the candidate's
// prediction and the QP; only the bookkeeping between the evaluations (which candidate is the best so far, whether the best is a skip, which evaluations are
// left out) is sequential.  An 8 x 8 CU keeps 16 lanes of a wavefront busy for its luma block and 8 for its two chroma blocks, and 8 x 8 CUs are where the
// walk spends most of its merge evaluations (22 % of a P-CTU's time on the 1080p bench clips, profiles/r06_history.md).  So:
//
//   1. quad_prepare: the (at most four) DISTINCT candidate vectors of the CU get a slot each; the worker runs the luma chain of all slots in ONE pass
//      (quad_chain<8>: four 8 x 8 blocks are the four quadrants of one 16 x 16 matrix-core tile, block-diagonal basis DevTables::fragp), its helper the chroma
//      chains of all slots in one pass (quad_chain<4>: eight 4 x 4 blocks in one tile, DevTables::fragq); the results - prediction, final levels,
//      reconstruction, distortion, level sum, no-residual distortion, cbf per slot and component - stay in the worker's LDS (QuadScratch, in the TU scratch).
//   2. the reference's candidate loop runs unchanged (enc_ctu.h check_rd_cost_merge), but its two expensive steps are copies now: motion compensation puts
//      the slot's prediction into the prediction window (quad_put_pred), the coded evaluation puts the slot's levels and reconstruction into the windows of the
//      CU's depth and sets the node's fields (quad_encode_inter) - every side effect of the sequential evaluation, none of its arithmetic.
//
// Not taken (the sequential evaluation runs as before): CUs of another size, performance_mode 0 (the 8 x 8 CU's transform tree then has a 4 x 4 level), more
// than four distinct candidates, a candidate whose vector points outside the padded reference (quirk Q12: evaluated on a stale window).
// The arithmetic of a chain is the one of encode_inter_tu (enc_inter.h) and the primitives it calls (enc_prims.h: tr_forward_mfma, quantize, sbh_pass,
// dequantize, tr_inverse_mfma, blk_ssd, blk_ssd_diff, blk_reconst), element for element; the checker build (one lane) has no such path, so every device / checker
// comparison and every stream fixture compares this path with the sequential one.
#pragma once
#include "enc_inter.h"

namespace henc {

#if defined(__HIPCC__) && defined(HENC_MFMA_TRANSFORM) && !defined(HENC_NO_QUAD)
#define HENC_QUAD 1

struct QuadRes {
	uint32_t dist[3];      // what encode_inter_tu returns per component (chroma: weighted)
	int32_t sum[3];        // its *curr_sum
	uint32_t raw[3];       // its *raw_ssq: SSD(source, prediction), unweighted
	uint32_t cbf[3];       // the node's inter_cbf after it
};
struct QuadScratch {      // in Work::pred_aux + Work::delta_u (the TU scratch: no TU is in flight while the candidate loop runs on the slots' results)
	uint8_t pred_y[4][64], rec_y[4][64];
	int16_t lv_y[4][64];
	uint8_t pred_c[4][2][16], rec_c[4][2][16];
	int16_t lv_c[4][2][16];
	QuadRes res[4];
	MV mv[4];
	int32_t acs[4];                              // level sums of the luma blocks (what the sign-hiding lanes ask for)
	int16_t wk_lv[256], wk_cf[256], wk_du[256];  // the luma chain's exchange buffers (levels / coefficients / remainders for the sign-hiding walk; the dequantised coefficients for the inverse's operand)
};
static_assert(sizeof(QuadScratch) <= 2 * TU_SCRATCH * sizeof(int16_t), "the slots' results live in the TU scratch");
static_assert(offsetof(Work, delta_u) == offsetof(Work, pred_aux) + TU_SCRATCH * sizeof(int16_t), "pred_aux and delta_u are one area");
HENC_INLINE QuadScratch &quad_scratch(Enc &__restrict__ e) { return *(QuadScratch *)(int16_t *)e.w->pred_aux; }

// sum over the lanes of a block of the tile: N = 4: the block's rows are the four lanes of a quad; N = 8: eight lanes of a half row and the eight 16 lanes on
template <int N>
__device__ __forceinline__ uint32_t quad_blk_sum(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);       // quad_perm [1, 0, 3, 2]
	x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);       // quad_perm [2, 3, 0, 1]
	if constexpr (N == 8) {
		x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: the other quad of the eight
		x += __builtin_amdgcn_ds_swizzle(x, 0x401F);                      // lane ^ 16
	}
	return (uint32_t)x;
}

// The chain of all slots for one component class.  N = 8: luma, slot = 2 (tile row / 8) + tile column / 8; N = 4: chroma, slot = tile row / 4, plane = tile
// column / 4 (columns 8 .. 15 of the tile stay empty).  A lane owns four consecutive elements of one block row all the way: source, prediction, residual,
// coefficients, levels, reconstruction; memory is only touched for the sign-hiding walk (a coefficient group per lane) and for the inverse transform's
// operand (the transposed block).  ALL 64 lanes run this in uniform control flow.
template <int N>
__device__ __forceinline__ void quad_chain(const int lane, Enc &__restrict__ e, const int ni, QuadScratch &qs, int16_t *wk_lv, int16_t *wk_cf, int16_t *wk_du, int32_t *acs)
{
	HENC_ENC_IN_LDS(e);
	constexpr int L = N == 8 ? 3 : 2, NN = N * N, GPB = NN / 16, NBLK = N == 8 ? 4 : 8;
	Work &w = *e.w;
	const Seq &S = *e.seq;
	const DevTables *T = e.T;
	const Geo &q = e.geo[ni];
	const int row = lane & 15, k0 = (lane >> 4) * 4;
	const bool is_y = N == 8;
	const int slot = is_y ? 2 * (row >> 3) + (k0 >> 3) : (row >> 2);
	const int plane = is_y ? 0 : ((lane >> 4) & 1);
	const bool active = is_y || lane < 32;
	const int blk = is_y ? slot : slot * 2 + plane;
	const int r = row & (N - 1), c0 = k0 & (N - 1), pos0 = r * N + c0;
	const int comp = is_y ? COMP_Y : COMP_U + plane;
	// source and prediction (motion compensation from the phase planes: motion_compensate_cu, enc_inter.h)
	const MV mv = qs.mv[slot];
	uint32_t o4, p4;
	if constexpr (N == 8) {
		const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y, sy = 16 * S.stride_y;
		const uint8_t *py = e.f->sub_y + (((mv.y & 3) << 2) | (mv.x & 3)) * S.stride_y + (ptrdiff_t)(gy + (mv.y >> 2)) * sy + gx + (mv.x >> 2);
		p4 = ld32u(py + r * sy + c0);
		o4 = *(const uint32_t *)(w.curr_y + (q.y + r) * 64 + q.x + c0);
		*(uint32_t *)(qs.pred_y[slot] + pos0) = p4;
	} else {
		const int gxc = (e.ctu_x >> 1) + q.xc, gyc = (e.ctu_y >> 1) + q.yc, sc = 64 * S.stride_c;
		const ptrdiff_t oc = (((mv.y & 7) << 3) | (mv.x & 7)) * S.stride_c + (ptrdiff_t)(gyc + (mv.y >> 3)) * sc + gxc + (mv.x >> 3);
		p4 = ld32u(e.f->sub_c[plane] + oc + r * sc);
		o4 = *(const uint32_t *)(w.curr_c[plane] + (q.yc + r) * 32 + q.xc);
		if (active) *(uint32_t *)(qs.pred_c[slot][plane] + pos0) = p4;
	}
	int x[4], pv[4];
#pragma unroll
	for (int k = 0; k < 4; k++) { pv[k] = (int)((p4 >> (8 * k)) & 255u); x[k] = active ? (int)((o4 >> (8 * k)) & 255u) - pv[k] : 0; }
	// quantiser values of the lane's four positions (quantize / dequantize, enc_prims.h: inter lists; 4 x 4: the flat value)
	const Node &nd = node_of(e, ni);
	const int qp = is_y ? (int)nd.qp : chroma_qp_table((int)nd.qp + S.chroma_qp_offset);
	const int per = qp / 6, rem = qp % 6;
	uint32_t qv[4], iv[4];
	if constexpr (N == 8) {
		const int32_t *qc = T->quant[1][3][rem] + pos0, *ic = T->dequant[1][3][rem] + pos0;
#pragma unroll
		for (int k = 0; k < 4; k++) { qv[k] = (uint32_t)qc[k]; iv[k] = (uint32_t)ic[k]; }
	} else {
		const uint32_t qf = (uint32_t)(uint16_t)T->quant[0][0][rem][0], ifl = (uint32_t)(uint16_t)T->dequant[0][0][rem][0];
#pragma unroll
		for (int k = 0; k < 4; k++) { qv[k] = qf; iv[k] = ifl; }
	}
	// forward transform (tr_forward_mfma with the block-diagonal basis)
	const mf_f4 z = {0, 0, 0, 0};
	int y[4];
	{
		constexpr int sh1 = L - 1, sh2 = L + 6, rnd1 = 1 << (sh1 - 1), rnd2 = 1 << (sh2 - 1);
		const mf_h4 m = mf_frag(N == 8 ? T->fragp[0][1] : T->fragq[0], lane);
		mf_h4 xh;
#pragma unroll
		for (int k = 0; k < 4; k++) xh[k] = (_Float16)(short)x[k];
		const mf_f4 d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(xh, m, z, 0, 0, 0);
		int t[4];
#pragma unroll
		for (int k = 0; k < 4; k++) t[k] = (int)sat16(((int)d1[k] + rnd1) >> sh1);
		mf_h4 hi, lo;
		mf_split(t, hi, lo);
		const mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, m, z, 0, 0, 0);
		const mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, m, z, 0, 0, 0);
#pragma unroll
		for (int k = 0; k < 4; k++) y[k] = mf_stage(dh[k], dl[k], rnd2, sh2);
	}
	// quantisation (quantize: inter block, the slice's rounding offset)
	const int qbits = 14 + per + (15 - 8 - L), qbits8 = qbits - 8;
	const int32_t qadd = (int32_t)((uint32_t)(e.f->slice_type == SLICE_I ? 171 : 85) << (qbits - 9));
	int lv[4], du[4];
	uint32_t lsum = 0, lraw = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int sv = y[k];
		const uint32_t a = (uint16_t)(sv < 0 ? -sv : sv);
		const int32_t aux = (int32_t)(a * qv[k]);
		const int32_t c = (int32_t)((uint32_t)aux + (uint32_t)qadd) >> qbits;
		const int32_t d = (int32_t)((uint32_t)aux - ((uint32_t)c << qbits)) >> qbits8;
		const int sgn = sv > 0 ? 1 : (sv < 0 ? -1 : 0);
		lsum += (uint32_t)c;
		lv[k] = (int16_t)(sgn * sat16(c));
		du[k] = sat16(d);
		const int32_t dd = (int16_t)x[k];
		lraw += (uint32_t)(dd * dd);
	}
	int sum = (int)quad_blk_sum<N>(lsum);                    // (the level sum BEFORE sign hiding, as quantize reports it)
	const uint32_t raw_zero = quad_blk_sum<N>(lraw);         // blk_ssd(source, prediction)
	// sign hiding (sbh_pass): a coefficient group per lane, over the blocks' buffers
	if (S.sign_hiding) {      // (uniform)
		if (active) {
			S4 a, b, c;
#pragma unroll
			for (int k = 0; k < 4; k++) { a.v[k] = (int16_t)lv[k]; b.v[k] = (int16_t)y[k]; c.v[k] = (int16_t)du[k]; }
			st4(wk_lv + blk * NN + pos0, a);
			st4(wk_cf + blk * NN + pos0, b);
			st4(wk_du + blk * NN + pos0, c);
			if (pos0 == 0) acs[blk] = sum;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		SbhGroup sg;
		const int gb = lane / GPB, cg = lane % GPB;
		const bool nz = lane < NBLK * GPB && acs[gb] >= 2 && sbh_gather(sg, wk_lv + gb * NN, wk_cf + gb * NN, wk_du + gb * NN, T->scan[SCAN_DIAG][L], cg);
		const uint64_t mask = __ballot(nz);
		if (nz) {
			const uint32_t mine = (uint32_t)(mask >> (gb * GPB)) & ((1u << GPB) - 1u);
			sbh_apply(sg, wk_lv + gb * NN, cg == 31 - __builtin_clz(mine));
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (active) {
			const S4 a = ld4(wk_lv + blk * NN + pos0);
#pragma unroll
			for (int k = 0; k < 4; k++) lv[k] = a.v[k];
		}
	}
	const bool coded = sum > 0;
	// dequantisation (dequantize), to the exchange buffer: the inverse transform's operand is the transposed tile
	{
		const int iq_shift = 20 - 14 - (15 - 8 - L) + 4;
		const int32_t iadd = iq_shift > per ? 1 << (iq_shift - per - 1) : 0;
		const int sh = iq_shift > per ? iq_shift - per : per - iq_shift;
		S4 o;
#pragma unroll
		for (int k = 0; k < 4; k++)
			o.v[k] = iq_shift > per ? sat16((int32_t)((uint32_t)(int32_t)lv[k] * iv[k] + (uint32_t)iadd) >> sh) : sat16((int32_t)(((uint32_t)(int32_t)lv[k] * iv[k]) << sh));
		if (active) st4(wk_cf + blk * NN + pos0, o);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	// inverse transform (tr_inverse_mfma): operand element e of the lane is tile element (k0 + e, row)
	int rd[4];
	{
		const mf_h4 mt = mf_frag(N == 8 ? T->fragp[1][1] : T->fragq[1], lane);
		int c[4] = {0, 0, 0, 0};
		// the block that holds tile rows k0 .. k0 + 3 and tile column `row`
		const int tb = is_y ? 2 * (k0 >> 3) + (row >> 3) : (k0 >> 2) * 2 + (row >> 2);
		const bool there = is_y || row < 8;
		if (there) {
#pragma unroll
			for (int k = 0; k < 4; k++) c[k] = wk_cf[tb * NN + (c0 + k) * N + r];
		}
		mf_h4 hi, lo;
		mf_split(c, hi, lo);
		mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, mt, z, 0, 0, 0);
		mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, mt, z, 0, 0, 0);
		int t[4];
#pragma unroll
		for (int k = 0; k < 4; k++) t[k] = mf_stage(dh[k], dl[k], 64, 7);
		mf_split(t, hi, lo);
		dh = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, hi, z, 0, 0, 0);
		dl = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, lo, z, 0, 0, 0);
#pragma unroll
		for (int k = 0; k < 4; k++) rd[k] = mf_stage(dh[k], dl[k], 2048, 12);
	}
	// blk_ssd_diff(source, prediction, reconstructed residual), then the keep-or-drop test of encode_inter_tu
	uint32_t lrec = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int32_t dd = (int16_t)((int16_t)x[k] - (int16_t)rd[k]);
		lrec += (uint32_t)(dd * dd);
	}
	const uint32_t raw = quad_blk_sum<N>(lrec);
	const double weight = e.f->chroma_weight;
	uint32_t ssd;
	bool keep = false;
	if (coded) {
		uint32_t ssd_zero;
		if (is_y) { ssd_zero = raw_zero; ssd = raw; }
		else { ssd_zero = (uint32_t)(weight * raw_zero); ssd = (uint32_t)(weight * raw); }
		const double thr = hclip(e.f->avg_dist / 2.5 - 5., 1., 20000.);
		const bool drop = is_y ? ((double)ssd_zero <= (double)(int)ssd + thr * sum) : ((double)ssd_zero <= (double)ssd + thr * sum);
		keep = !drop;
		if (drop) sum = 0;
	} else {
		ssd = is_y ? raw_zero : (uint32_t)(weight * raw_zero);
	}
	// reconstruction (blk_reconst) and the block's final levels
	uint32_t rec4 = 0;
	S4 fl;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int v = hclip((int)sat16(pv[k] + (keep ? rd[k] : 0)), 0, 255);
		rec4 |= (uint32_t)v << (8 * k);
		fl.v[k] = keep ? (int16_t)lv[k] : (int16_t)0;
	}
	if (active) {
		if constexpr (N == 8) {
			*(uint32_t *)(qs.rec_y[slot] + pos0) = rec4;
			st4(qs.lv_y[slot] + pos0, fl);
		} else {
			*(uint32_t *)(qs.rec_c[slot][plane] + pos0) = rec4;
			st4(qs.lv_c[slot][plane] + pos0, fl);
		}
		if (pos0 == 0) {
			QuadRes &res = qs.res[slot];
			res.dist[comp] = ssd;
			res.sum[comp] = sum;
			res.raw[comp] = raw_zero;
			res.cbf[comp] = sum ? 1u : 0u;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the helper's half of quad_prepare: the chroma blocks of all slots (k_encode.hip, HJOB_QUAD_C); its exchange buffers are its own scratch
__device__ __forceinline__ void quad_chroma_job(const WaveGrp g, Enc &__restrict__ e, int ni)
{
	HENC_ENC_IN_LDS(e);
	int16_t *s = e.scratch_a;
	quad_chain<4>(g.tid, e, uni(ni), quad_scratch(e), s, s + 128, s + 256, (int32_t *)(s + 384));
}

// Slots for the CU's merge candidates and the chains of all slots.  Returns -1 when the CU is evaluated the sequential way (see the head of the file), else
// the slot of candidate cand in bits 4 cand .. 4 cand + 3 (quad_slot).
HENC_INLINE int quad_slot(int slots, int cand) { return (slots >> (4 * cand)) & 15; }
__device__ __forceinline__ int quad_prepare(const WaveGrp g, Enc &__restrict__ e, int ni)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Seq &S = *e.seq;
	const Geo &q = e.geo[ni];
	if (q.size != 8 || S.perf_mode < 1) return -1;
	QuadScratch &qs = quad_scratch(e);
	const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y, n = 8;
	const int xlow = -S.margin_y, xhigh = S.width + S.margin_y, ylow = -S.margin_y, yhigh = S.height + S.margin_y;
	int nslots = 0, slots = 0;
	MV smv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
	int sref[4] = {0, 0, 0, 0};
	for (int cand = 0; cand < CFG_NUM_MERGE_CAND; cand++) {
		const MV mv = w.merge_cands.mv[cand];
		const int ref = w.merge_cands.ref_idx[cand];
		const int spx = gx + mv.x / 4, spy = gy + mv.y / 4;
		if (spx < xlow || spx + n > xhigh || spy < ylow || spy + n > yhigh) return -1;      // Q12: the sequential evaluation reads a stale window
		int s = -1;
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (s < 0 && k < nslots && smv[k].x == mv.x && smv[k].y == mv.y && sref[k] == ref) s = k;
		if (s < 0) {
			if (nslots == 4) return -1;
			s = nslots;
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (k == nslots) { smv[k] = mv; sref[k] = ref; }
			nslots++;
		}
		slots |= s << (4 * cand);
	}
	// (slots nobody uses repeat slot 0: every block of the tile then holds valid data)
#pragma unroll
	for (int k = 0; k < 4; k++) qs.mv[k] = k < nslots ? smv[k] : smv[0];
	g.sync();
	helper_post(g, e, 0, HJOB_QUAD_C, ni);
	quad_chain<8>(g.tid, e, ni, qs, qs.wk_lv, qs.wk_cf, qs.wk_du, qs.acs);
	helper_wait(g, e, 0);
	return slots;
}

// What the candidate loop leaves behind, written once: `cons` = the evaluation put_consolidated_info saw last (the best: into window 0 and the CTU's record),
// `wnd` = the evaluation that wrote the windows and per-depth buffers of the CU's depth last, `last_pred` = the slot whose prediction the window holds.
// An evaluation is a slot and a kind: 1 = coded (the slot's levels and reconstruction), 2 = a winning no-residual evaluation (zero levels, the prediction).
__device__ __forceinline__ void quad_commit(const WaveGrp g, Enc &__restrict__ e, int ni, int depth, int cons_slot, int cons_kind, int wnd_slot, int wnd_kind, int last_pred, int coded_any)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	QuadScratch &qs = quad_scratch(e);
	const S4 zero = {{0, 0, 0, 0}};
	if (g.tid < 16) {
		const int r = g.tid >> 1, c0 = (g.tid & 1) * 4;
		if (cons_kind) {
			st4(tq_ptr(w, 0, COMP_Y) + (q.abs_index << 4) + g.tid * 4, cons_kind == 1 ? ld4(qs.lv_y[cons_slot] + g.tid * 4) : zero);
			st4(dec_ptr(w, 0, COMP_Y) + (q.y + r) * DEC_STRIDE_Y + q.x + c0, ld4((cons_kind == 1 ? qs.rec_y[cons_slot] : qs.pred_y[cons_slot]) + r * 8 + c0));
		}
		if (wnd_kind) {
			st4(tq_ptr(w, depth + 1, COMP_Y) + (q.abs_index << 4) + g.tid * 4, wnd_kind == 1 ? ld4(qs.lv_y[wnd_slot] + g.tid * 4) : zero);
			st4(dec_ptr(w, depth + 1, COMP_Y) + (q.y + r) * DEC_STRIDE_Y + q.x + c0, ld4((wnd_kind == 1 ? qs.rec_y[wnd_slot] : qs.pred_y[wnd_slot]) + r * 8 + c0));
		}
		if (last_pred >= 0) *(uint32_t *)(w.pred_y + (q.y + r) * 64 + q.x + c0) = *(const uint32_t *)(qs.pred_y[last_pred] + r * 8 + c0);
	} else if (g.tid < 24) {
		const int p = (g.tid - 16) >> 2, r = g.tid & 3;
		if (cons_kind) {
			st4(tq_ptr(w, 0, COMP_U + p) + ((q.abs_index << 4) >> 2) + r * 4, cons_kind == 1 ? ld4(qs.lv_c[cons_slot][p] + r * 4) : zero);
			st4(dec_ptr(w, 0, COMP_U + p) + (q.yc + r) * DEC_STRIDE_C + q.xc, ld4((cons_kind == 1 ? qs.rec_c[cons_slot][p] : qs.pred_c[cons_slot][p]) + r * 4));
		}
		if (wnd_kind) {
			st4(tq_ptr(w, depth + 1, COMP_U + p) + ((q.abs_index << 4) >> 2) + r * 4, wnd_kind == 1 ? ld4(qs.lv_c[wnd_slot][p] + r * 4) : zero);
			st4(dec_ptr(w, depth + 1, COMP_U + p) + (q.yc + r) * DEC_STRIDE_C + q.xc, ld4((wnd_kind == 1 ? qs.rec_c[wnd_slot][p] : qs.pred_c[wnd_slot][p]) + r * 4));
		}
		if (last_pred >= 0) *(uint32_t *)(w.pred_c[p] + (q.yc + r) * 32 + q.xc) = *(const uint32_t *)(qs.pred_c[last_pred][p] + r * 4);
	} else if (g.tid < 24 + 4) {
		// the side info of the CU's four units: put_consolidated_info's copy of the per-depth buffers into the CTU's record (with what the buffers held for `cons`),
		// then the buffers as `wnd` left them (set_enc_info_buffs), and encode_inter's reference-index write
		const int k = q.abs_index + (g.tid - 24);
		if (cons_kind) {
			const QuadRes &cr = qs.res[cons_slot];
			c.cbf[0][k] = cons_kind == 1 ? (uint8_t)cr.cbf[0] : (uint8_t)0;
			c.cbf[1][k] = cons_kind == 1 ? (uint8_t)cr.cbf[1] : (uint8_t)0;
			c.cbf[2][k] = cons_kind == 1 ? (uint8_t)cr.cbf[2] : (uint8_t)0;
			c.tr_idx[k] = 0;
			c.intra_mode[0][k] = w.intra_mode_buffs[0][depth][k];
			c.intra_mode[1][k] = w.intra_mode_buffs[1][depth][k];
		}
		if (wnd_kind) {
			const QuadRes &wr = qs.res[wnd_slot];
			w.cbf_buffs[COMP_Y][depth][k] = wnd_kind == 1 ? (uint8_t)wr.cbf[0] : (uint8_t)0;
			w.cbf_buffs[COMP_U][depth][k] = wnd_kind == 1 ? (uint8_t)wr.cbf[1] : (uint8_t)0;
			w.cbf_buffs[COMP_V][depth][k] = wnd_kind == 1 ? (uint8_t)wr.cbf[2] : (uint8_t)0;
			w.tr_idx_buffs[depth][k] = 0;
		}
		if (coded_any) c.mv_ref_idx[k] = (int8_t)nd.inter_ref_index;
	}
	g.sync();
}

// check_rd_cost_merge (enc_ctu.h) on the slots' results: the same loop, statement for statement, with every evaluation a look-up and every copy deferred to
// quad_commit.  `slots` from quad_prepare; inter_modes from get_merge_candidates.
__device__ __forceinline__ uint32_t quad_merge_loop(const WaveGrp g, Enc &__restrict__ e, int ni, int slots, const uint8_t *inter_modes)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[ni];
	Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	QuadScratch &qs = quad_scratch(e);
	const int abs_index = q.abs_index, curr_depth = q.depth;
	uint32_t no_nores_mask = 0;      // merge_cand_buffer: bit cand
	int best_is_skip = 0, best_candidate = 0, have_ctu_cbf = 0, prev_nores_ran = 0;
	uint32_t dist, best_dist = MAX_COST, cost, best_cost = MAX_COST, best_sum = 0, ctu_cbf = 0;
	MV best_mv = {0, 0};
	int best_ref_idx = 0;
	int cons_slot = 0, cons_kind = 0, wnd_slot = 0, wnd_kind = 0, last_pred = -1, coded_any = 0;
	const double weight = e.f->chroma_weight;
	for (int cand = 0; cand < CFG_NUM_MERGE_CAND; cand++) {
		const int slot = quad_slot(slots, cand);
		int mc_done = 0;
		if (cand >= 1 && w.merge_cands.mv[cand].x == w.merge_cands.mv[cand - 1].x && w.merge_cands.mv[cand].y == w.merge_cands.mv[cand - 1].y &&
		    w.merge_cands.ref_idx[cand] == w.merge_cands.ref_idx[cand - 1]) {
			const int coded_runs = !best_is_skip;
			if (coded_runs) no_nores_mask = (no_nores_mask & ~(1u << cand)) | (((no_nores_mask >> (cand - 1)) & 1u) << cand);
			const int nores_runs = !(coded_runs && ((no_nores_mask >> cand) & 1u));
			if (!nores_runs || prev_nores_ran) {
				if (nores_runs) {
					nd.inter_cbf[0] = nd.inter_cbf[1] = nd.inter_cbf[2] = 0;
					nd.inter_tr_idx = 0;
					nd.sum = 0;
				}
				prev_nores_ran = nores_runs;
				continue;
			}
		}
		prev_nores_ran = 0;
		const QuadRes &qr = qs.res[slot];
		for (int no_res = 0; no_res < 2; no_res++) {
			if (no_res == 1 && ((no_nores_mask >> cand) & 1u)) continue;
			if (best_is_skip && no_res == 0) continue;
			if (no_res == 1) prev_nores_ran = 1;
			if (!mc_done) { last_pred = slot; mc_done = 1; }
			if (no_res == 0) {
				// encode_inter: the node's fields, the windows and buffers of the CU's depth (deferred: wnd), the squared residuals
				nd.inter_cbf[0] = (int32_t)qr.cbf[0];
				nd.inter_cbf[1] = (int32_t)qr.cbf[1];
				nd.inter_cbf[2] = (int32_t)qr.cbf[2];
				nd.inter_tr_idx = 0;
				dist = qr.dist[0] + qr.dist[1] + qr.dist[2];
				nd.distortion = dist;
				nd.cost = dist;
				nd.sum = (uint32_t)(qr.sum[0] + qr.sum[1] + qr.sum[2]);
				e.inter_ssq[0] = qr.raw[0]; e.inter_ssq[1] = qr.raw[1]; e.inter_ssq[2] = qr.raw[2];
				e.inter_ssq_valid = 1;
				wnd_slot = slot; wnd_kind = 1; coded_any = 1;
				cost = dist;
				cost = (uint32_t)((double)cost + cost_rd(e.f->avg_dist, nd.sum));
			} else {
				dist = qr.raw[0];
				dist += (uint32_t)(weight * qr.raw[1]);
				dist += (uint32_t)(weight * qr.raw[2]);
				nd.inter_cbf[0] = nd.inter_cbf[1] = nd.inter_cbf[2] = 0;
				nd.inter_tr_idx = 0;
				nd.sum = 0;
				cost = dist;
			}
			if (cost < best_cost) {
				best_mv = w.merge_cands.mv[cand];
				best_ref_idx = w.merge_cands.ref_idx[cand];
				best_candidate = cand;
				best_dist = dist;
				best_cost = cost;
				best_sum = nd.sum;
				if (no_res == 1) { wnd_slot = slot; wnd_kind = 2; }      // the prediction is the reconstruction, the levels are zero
				cons_slot = wnd_slot; cons_kind = wnd_kind;               // put_consolidated_info
				ctu_cbf = wnd_kind == 1 ? (qs.res[wnd_slot].cbf[0] | qs.res[wnd_slot].cbf[1] | qs.res[wnd_slot].cbf[2]) : 0u;
				have_ctu_cbf = 1;
				best_is_skip = (ctu_cbf & 1) == 0;
			}
			if (no_res == 0) {
				if (!have_ctu_cbf) { ctu_cbf = (uint32_t)c.cbf[0][abs_index] | c.cbf[1][abs_index] | c.cbf[2][abs_index]; have_ctu_cbf = 1; }
				if (((ctu_cbf >> curr_depth) & 1) == 0) no_nores_mask |= 1u << cand;
			}
		}
	}
	quad_commit(g, e, ni, curr_depth, cons_slot, cons_kind, wnd_slot, wnd_kind, last_pred, coded_any);
	nd.skipped = best_is_skip;
	nd.inter_mv = best_mv;
	nd.inter_ref_index = best_ref_idx;
	nd.cost = nd.distortion = best_dist;
	nd.merge_flag = 1;
	nd.merge_idx = best_candidate;
	nd.inter_mode = inter_modes[best_candidate];
	nd.sum = best_sum;
	return best_dist;
}

#endif

}  // namespace henc
