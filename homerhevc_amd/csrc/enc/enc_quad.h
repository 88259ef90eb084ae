// The merge evaluation of an 8 x 8 or 16 x 16 CU with all its candidates side by side (device only).
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
// A 16 x 16 CU (round 6, second step) goes the same way with other tile shapes: its luma block IS a tile (the slots' luma chains run one after the other in one
// piece of code, DevTables::frag16), its four chroma blocks (two slots x two planes, 8 x 8) are the four quadrants of the helper's tile.
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
// in Work::pred_aux + Work::delta_u (the TU scratch: no TU is in flight while the candidate loop runs on the slots' results).  CU 8: up to four slots and the luma
// chain's exchange buffers; CU 16: up to two slots (the luma chain's exchange buffers are the worker's level slot, Work::iq_y)
struct QuadScratch {
	union {
		struct {
			uint8_t pred_y[4][64], rec_y[4][64];
			int16_t lv_y[4][64];
			uint8_t pred_c[4][2][16], rec_c[4][2][16];
			int16_t lv_c[4][2][16];
			int16_t wk_lv[256], wk_cf[256], wk_du[256];      // levels / coefficients / remainders for the sign-hiding walk; the dequantised coefficients for the inverse's operand
		} s8;
		struct {
			uint8_t pred_y[2][256], rec_y[2][256];
			int16_t lv_y[2][256];
			uint8_t pred_c[2][2][64], rec_c[2][2][64];
			int16_t lv_c[2][2][64];
		} s16;
	};
	QuadRes res[4];
	MV mv[4];
	int32_t acs[4];      // level sums of the luma blocks (what the sign-hiding lanes ask for)
};
static_assert(sizeof(QuadScratch) <= 2 * TU_SCRATCH * sizeof(int16_t), "the slots' results live in the TU scratch");
static_assert(offsetof(Work, delta_u) == offsetof(Work, pred_aux) + TU_SCRATCH * sizeof(int16_t), "pred_aux and delta_u are one area");
HENC_INLINE QuadScratch &quad_scratch(Enc &__restrict__ e) { return *(QuadScratch *)(int16_t *)e.w->pred_aux; }
template <int CU> __device__ __forceinline__ uint8_t *quad_pred_y(QuadScratch &qs, int slot) { if constexpr (CU == 8) return qs.s8.pred_y[slot]; else return qs.s16.pred_y[slot]; }
template <int CU> __device__ __forceinline__ uint8_t *quad_rec_y(QuadScratch &qs, int slot) { if constexpr (CU == 8) return qs.s8.rec_y[slot]; else return qs.s16.rec_y[slot]; }
template <int CU> __device__ __forceinline__ int16_t *quad_lv_y(QuadScratch &qs, int slot) { if constexpr (CU == 8) return qs.s8.lv_y[slot]; else return qs.s16.lv_y[slot]; }
template <int CU> __device__ __forceinline__ uint8_t *quad_pred_c(QuadScratch &qs, int slot, int p) { if constexpr (CU == 8) return qs.s8.pred_c[slot][p]; else return qs.s16.pred_c[slot][p]; }
template <int CU> __device__ __forceinline__ uint8_t *quad_rec_c(QuadScratch &qs, int slot, int p) { if constexpr (CU == 8) return qs.s8.rec_c[slot][p]; else return qs.s16.rec_c[slot][p]; }
template <int CU> __device__ __forceinline__ int16_t *quad_lv_c(QuadScratch &qs, int slot, int p) { if constexpr (CU == 8) return qs.s8.lv_c[slot][p]; else return qs.s16.lv_c[slot][p]; }

// sum over the lanes of a block of the tile: N = 4: the block's rows are the four lanes of a quad; N = 8: eight lanes of a half row and the eight 16 lanes on;
// N = 16: the whole wavefront
template <int N>
__device__ __forceinline__ uint32_t quad_blk_sum(uint32_t v)
{
	int x = (int)v;
	if constexpr (N == 16) {
		x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
		return (uint32_t)(__builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63));
	} else {
		x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);       // quad_perm [1, 0, 3, 2]
		x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);       // quad_perm [2, 3, 0, 1]
		if constexpr (N == 8) {
			x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror: the other quad of the eight
			x += __builtin_amdgcn_ds_swizzle(x, 0x401F);                      // lane ^ 16
		}
		return (uint32_t)x;
	}
}

// The chain of the slots' blocks of one component class in ONE matrix-core tile.  Block size N and what the tile holds:
//   CU 8 luma:    N = 8,  four blocks, slot = 2 (tile row / 8) + tile column / 8            (block-diagonal basis DevTables::fragp)
//   CU 8 chroma:  N = 4,  eight blocks, slot = tile row / 4, plane = tile column / 4 (columns 8 .. 15 of the tile stay empty; DevTables::fragq)
//   CU 16 luma:   N = 16, the block of slot `one_slot` (DevTables::frag16)
//   CU 16 chroma: N = 8,  four blocks, quadrant b = 2 (tile row / 8) + tile column / 8: slot = b / 2, plane = b % 2
// A lane owns four consecutive elements of one block row all the way: source, prediction, residual, coefficients, levels, reconstruction; memory is only touched
// for the sign-hiding walk (a coefficient group per lane) and for the inverse transform's operand (the transposed block).  ALL 64 lanes run this in uniform control flow.
template <int CU, bool CHROMA>
__device__ __forceinline__ void quad_chain(const int lane, Enc &__restrict__ e, const int ni, QuadScratch &qs, const int one_slot, int16_t *wk_lv, int16_t *wk_cf, int16_t *wk_du, int32_t *acs)
{
	HENC_ENC_IN_LDS(e);
	constexpr int N = CHROMA ? CU / 2 : CU;
	constexpr int L = N == 16 ? 4 : (N == 8 ? 3 : 2), NN = N * N, GPB = NN / 16, NBLK = N == 16 ? 1 : (N == 8 ? 4 : 8);
	Work &w = *e.w;
	const Seq &S = *e.seq;
	const DevTables *T = e.T;
	const Geo &q = e.geo[ni];
	const int row = lane & 15, k0 = (lane >> 4) * 4;
	const int quadrant = 2 * (row >> 3) + (k0 >> 3);
	const int slot = N == 16 ? one_slot : (N == 8 ? (CHROMA ? quadrant >> 1 : quadrant) : (row >> 2));
	const int plane = !CHROMA ? 0 : (N == 8 ? quadrant & 1 : ((lane >> 4) & 1));
	const bool active = N != 4 || lane < 32;
	const int blk = N == 16 ? 0 : (N == 8 ? quadrant : slot * 2 + plane);      // the block's place in the exchange buffers
	const int r = row & (N - 1), c0 = k0 & (N - 1), pos0 = r * N + c0;
	const int comp = CHROMA ? COMP_U + plane : COMP_Y;
	uint8_t *const st_pred = CHROMA ? quad_pred_c<CU>(qs, slot, plane) : quad_pred_y<CU>(qs, slot);
	uint8_t *const st_rec = CHROMA ? quad_rec_c<CU>(qs, slot, plane) : quad_rec_y<CU>(qs, slot);
	int16_t *const st_lv = CHROMA ? quad_lv_c<CU>(qs, slot, plane) : quad_lv_y<CU>(qs, slot);
	// source and prediction (motion compensation from the phase planes: motion_compensate_cu, enc_inter.h)
	const MV mv = qs.mv[slot];
	uint32_t o4, p4;
	if constexpr (!CHROMA) {
		const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y, sy = 16 * S.stride_y;
		const uint8_t *py = e.f->sub_y + (((mv.y & 3) << 2) | (mv.x & 3)) * S.stride_y + (ptrdiff_t)(gy + (mv.y >> 2)) * sy + gx + (mv.x >> 2);
		p4 = ld32u(py + r * sy + c0);
		o4 = *(const uint32_t *)(w.curr_y + (q.y + r) * 64 + q.x + c0);
	} else {
		const int gxc = (e.ctu_x >> 1) + q.xc, gyc = (e.ctu_y >> 1) + q.yc, sc = 64 * S.stride_c;
		const ptrdiff_t oc = (((mv.y & 7) << 3) | (mv.x & 7)) * S.stride_c + (ptrdiff_t)(gyc + (mv.y >> 3)) * sc + gxc + (mv.x >> 3);
		p4 = ld32u(e.f->sub_c[plane] + oc + r * sc + c0);
		o4 = *(const uint32_t *)(w.curr_c[plane] + (q.yc + r) * 32 + q.xc + c0);
	}
	if (active) *(uint32_t *)(st_pred + pos0) = p4;
	int x[4], pv[4];
#pragma unroll
	for (int k = 0; k < 4; k++) { pv[k] = (int)((p4 >> (8 * k)) & 255u); x[k] = active ? (int)((o4 >> (8 * k)) & 255u) - pv[k] : 0; }
	// quantiser values of the lane's four positions (quantize / dequantize, enc_prims.h: the inter lists as 8 x 8 cells; 4 x 4: the flat value)
	const Node &nd = node_of(e, ni);
	const int qp = !CHROMA ? (int)nd.qp : chroma_qp_table((int)nd.qp + S.chroma_qp_offset);
	const int per = qp / 6, rem = qp % 6;
	uint32_t qv[4], iv[4];
	ft_list_value4(T->quant[1][3][rem], (uint32_t)(uint16_t)T->quant[0][0][rem][0], pos0, L, qv);
	ft_list_value4(T->dequant[1][3][rem], (uint32_t)(uint16_t)T->dequant[0][0][rem][0], pos0, L, iv);
	// the scan positions of the lane's coefficient group for the sign-hiding walk (lanes below NBLK * GPB: block gb, group cg), fetched with everything else that
	// comes from memory
	const int gb = lane / GPB, cg = lane % GPB;
	SbhGroup sg;
	if (S.sign_hiding && lane < NBLK * GPB) {
		const uint32_t *sc = T->scan[SCAN_DIAG][L] + (cg << 4);
#pragma unroll
		for (int n = 0; n < 16; n++) sg.pos[n] = sc[n];
	}
	const uint16_t *const frag_f = N == 16 ? T->frag16[0][2] : (N == 8 ? T->fragp[0][1] : T->fragq[0]);
	const uint16_t *const frag_i = N == 16 ? T->frag16[1][2] : (N == 8 ? T->fragp[1][1] : T->fragq[1]);
	// forward transform (tr_forward_mfma with the block-diagonal basis)
	const mf_f4 z = {0, 0, 0, 0};
	int y[4];
	{
		constexpr int sh1 = L - 1, sh2 = L + 6, rnd1 = 1 << (sh1 - 1), rnd2 = 1 << (sh2 - 1);
		const mf_h4 m = mf_frag(frag_f, lane);
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
	// sign hiding (sbh_pass): a coefficient group per lane, over the blocks' buffers - when any block has two levels to hide a sign in
	if (S.sign_hiding && __ballot(active && sum >= 2) != 0) {      // (uniform)
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
		bool nz = false;
		if (lane < NBLK * GPB && acs[gb] >= 2) {
			const int16_t *dl = wk_lv + gb * NN, *sl = wk_cf + gb * NN, *ul = wk_du + gb * NN;
			int any = 0;
#pragma unroll
			for (int n = 0; n < 16; n++) { sg.lv[n] = dl[sg.pos[n]]; sg.sv[n] = sl[sg.pos[n]]; sg.du[n] = ul[sg.pos[n]]; any |= sg.lv[n]; }      // (sbh_gather)
			nz = any != 0;
		}
		const uint64_t mask = __ballot(nz);
		if (nz) {
			const uint32_t mine = GPB == 16 ? (uint32_t)mask : ((uint32_t)(mask >> (gb * GPB)) & ((1u << (GPB & 15)) - 1u));
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
	int rd[4] = {0, 0, 0, 0};
	uint32_t raw = 0;
	if (__ballot(active && coded) != 0) {      // (uniform: some block has levels)
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
		{
			const mf_h4 mt = mf_frag(frag_i, lane);
			int c[4] = {0, 0, 0, 0};
			// the block that holds tile rows k0 .. k0 + 3 and tile column `row`
			const int tb = N == 16 ? 0 : (N == 8 ? 2 * (k0 >> 3) + (row >> 3) : (k0 >> 2) * 2 + (row >> 2));
			const bool there = N != 4 || row < 8;
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
		// blk_ssd_diff(source, prediction, reconstructed residual)
		uint32_t lrec = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int32_t dd = (int16_t)((int16_t)x[k] - (int16_t)rd[k]);
			lrec += (uint32_t)(dd * dd);
		}
		raw = quad_blk_sum<N>(lrec);
	}
	// the keep-or-drop test of encode_inter_tu
	const double weight = e.f->chroma_weight;
	uint32_t ssd;
	bool keep = false;
	if (coded) {
		uint32_t ssd_zero;
		if (!CHROMA) { ssd_zero = raw_zero; ssd = raw; }
		else { ssd_zero = (uint32_t)(weight * raw_zero); ssd = (uint32_t)(weight * raw); }
		const double thr = hclip(e.f->avg_dist / 2.5 - 5., 1., 20000.);
		const bool drop = !CHROMA ? ((double)ssd_zero <= (double)(int)ssd + thr * sum) : ((double)ssd_zero <= (double)ssd + thr * sum);
		keep = !drop;
		if (drop) sum = 0;
	} else {
		ssd = !CHROMA ? raw_zero : (uint32_t)(weight * raw_zero);
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
		*(uint32_t *)(st_rec + pos0) = rec4;
		st4(st_lv + pos0, fl);
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

// the helper's half of quad_prepare: the chroma blocks of all slots (k_encode.hip, HJOB_QUAD_C; cu: the CU's size); its exchange buffers are its own scratch
__device__ __forceinline__ void quad_chroma_job(const WaveGrp g, Enc &__restrict__ e, int ni, int cu)
{
	HENC_ENC_IN_LDS(e);
	int16_t *s = e.scratch_a;
	if (uni(cu) == 8) quad_chain<8, true>(g.tid, e, uni(ni), quad_scratch(e), 0, s, s + 128, s + 256, (int32_t *)(s + 384));
	else quad_chain<16, true>(g.tid, e, uni(ni), quad_scratch(e), 0, s, s + 256, s + 512, (int32_t *)(s + 768));
}

// the CU's merge candidates in registers (one look at the list for the slot assignment and the candidate loop; indexed with constants only: the loops over
// the candidates are unrolled)
struct QuadCands {
	int x[5], y[5], ref[5];
};
__device__ __forceinline__ QuadCands quad_load_cands(Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	const MvCandList &l = e.w->merge_cands;
	QuadCands c;
#pragma unroll
	for (int k = 0; k < 5; k++) {
		if (k < CFG_NUM_MERGE_CAND) { c.x[k] = uni(l.mv[k].x); c.y[k] = uni(l.mv[k].y); c.ref[k] = uni(l.ref_idx[k]); }
		else c.x[k] = c.y[k] = c.ref[k] = 0;
	}
	return c;
}

// Slots for the CU's merge candidates and the chains of all slots.  Returns -1 when the CU is evaluated the sequential way (see the head of the file), else
// the slot of candidate cand in bits 4 cand .. 4 cand + 3 (quad_slot).
HENC_INLINE int quad_slot(int slots, int cand) { return (slots >> (4 * cand)) & 15; }
template <int CU>
__device__ __forceinline__ int quad_prepare(const WaveGrp g, Enc &__restrict__ e, int ni, const QuadCands &mc HENC_QPROF_ARG)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	const Geo &q = e.geo[ni];
	if (S.perf_mode < 1) return -1;
	static_assert(CFG_NUM_MERGE_CAND <= 5, "at most five merge candidates (QuadCands, the slot bits)");
	constexpr int MAXS = CU == 8 ? 4 : 2;
	QuadScratch &qs = quad_scratch(e);
	const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y, n = CU;
	const int xlow = -S.margin_y, xhigh = S.width + S.margin_y, ylow = -S.margin_y, yhigh = S.height + S.margin_y;
	int nslots = 0, slots = 0, bad = 0;
	int sx[4] = {0, 0, 0, 0}, sy[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0};
#pragma unroll
	for (int cand = 0; cand < CFG_NUM_MERGE_CAND; cand++) {
		const int spx = gx + mc.x[cand] / 4, spy = gy + mc.y[cand] / 4;
		if (spx < xlow || spx + n > xhigh || spy < ylow || spy + n > yhigh) bad = 1;      // Q12: the sequential evaluation reads a stale window
		int s = -1;
#pragma unroll
		for (int k = 3; k >= 0; k--)
			if (k < nslots && sx[k] == mc.x[cand] && sy[k] == mc.y[cand] && sr[k] == mc.ref[cand]) s = k;
		if (s < 0) {
			if (nslots == MAXS) bad = 1;
			s = nslots & 3;
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (k == s && nslots < MAXS) { sx[k] = mc.x[cand]; sy[k] = mc.y[cand]; sr[k] = mc.ref[cand]; }
			nslots++;
		}
		slots |= s << (4 * cand);
	}
	if (bad) return -1;
	// (slots nobody uses repeat slot 0: every block of the tile then holds valid data)
	if (g.tid < 4) {
		int mx = sx[0], my = sy[0];
#pragma unroll
		for (int k = 1; k < 4; k++)
			if (g.tid == k && k < nslots) { mx = sx[k]; my = sy[k]; }
		qs.mv[g.tid].x = mx;
		qs.mv[g.tid].y = my;
	}
	g.sync();
	HENC_QPROF_MARK(e, 2);      // (the slots)
	helper_post(g, e, 0, HJOB_QUAD_C, ni, CU);
	if constexpr (CU == 8) quad_chain<8, false>(g.tid, e, ni, qs, 0, qs.s8.wk_lv, qs.s8.wk_cf, qs.s8.wk_du, qs.acs);
	else {
		// the slots' luma blocks one after the other (a block is a tile); the exchange buffers: the worker's level slot
		int16_t *x0 = e.w->iq_y;
		quad_chain<16, false>(g.tid, e, ni, qs, 0, x0, x0 + 256, x0 + 512, qs.acs);
		if (nslots > 1) quad_chain<16, false>(g.tid, e, ni, qs, 1, x0, x0 + 256, x0 + 512, qs.acs);
	}
	HENC_QPROF_MARK(e, 3);      // (the luma chain)
	helper_wait(g, e, 0);
	HENC_QPROF_MARK(e, 4);      // (the rest of the helper's chroma chain)
	return slots;
}

// What the candidate loop leaves behind, written once: `cons` = the evaluation put_consolidated_info saw last (the best: into window 0 and the CTU's record),
// `wnd` = the evaluation that wrote the windows and per-depth buffers of the CU's depth last, `last_pred` = the slot whose prediction the window holds.
// An evaluation is a slot and a kind: 1 = coded (the slot's levels and reconstruction), 2 = a winning no-residual evaluation (zero levels, the prediction).
template <int CU>
__device__ __forceinline__ void quad_commit(const WaveGrp g, Enc &__restrict__ e, int ni, int depth, int cons_slot, int cons_kind, int wnd_slot, int wnd_kind, int last_pred, int coded_any)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	QuadScratch &qs = quad_scratch(e);
	const S4 zero = {{0, 0, 0, 0}};
	constexpr int NC = CU / 2, YL = CU * CU / 4, CL = 2 * NC * NC / 4;      // lanes (four samples each) of the luma block and of the two chroma blocks
	// luma: lanes 0 .. YL - 1; chroma: lanes YL .. YL + CL - 1 (CU 8) or, after the luma lanes have had the whole wavefront, lanes 0 .. CL - 1 (CU 16)
	auto luma = [&](int l) {
		const int r = l / (CU / 4), c0 = (l % (CU / 4)) * 4;
		if (cons_kind) {
			st4(tq_ptr(w, 0, COMP_Y) + (q.abs_index << 4) + l * 4, cons_kind == 1 ? ld4(quad_lv_y<CU>(qs, cons_slot) + l * 4) : zero);
			st4(dec_ptr(w, 0, COMP_Y) + (q.y + r) * DEC_STRIDE_Y + q.x + c0, ld4((cons_kind == 1 ? quad_rec_y<CU>(qs, cons_slot) : quad_pred_y<CU>(qs, cons_slot)) + r * CU + c0));
		}
		if (wnd_kind) {
			st4(tq_ptr(w, depth + 1, COMP_Y) + (q.abs_index << 4) + l * 4, wnd_kind == 1 ? ld4(quad_lv_y<CU>(qs, wnd_slot) + l * 4) : zero);
			st4(dec_ptr(w, depth + 1, COMP_Y) + (q.y + r) * DEC_STRIDE_Y + q.x + c0, ld4((wnd_kind == 1 ? quad_rec_y<CU>(qs, wnd_slot) : quad_pred_y<CU>(qs, wnd_slot)) + r * CU + c0));
		}
		if (last_pred >= 0) *(uint32_t *)(w.pred_y + (q.y + r) * 64 + q.x + c0) = *(const uint32_t *)(quad_pred_y<CU>(qs, last_pred) + r * CU + c0);
	};
	auto chroma = [&](int l) {
		const int p = l / (NC * NC / 4), j = l % (NC * NC / 4), r = j / (NC / 4), c0 = (j % (NC / 4)) * 4;
		if (cons_kind) {
			st4(tq_ptr(w, 0, COMP_U + p) + ((q.abs_index << 4) >> 2) + j * 4, cons_kind == 1 ? ld4(quad_lv_c<CU>(qs, cons_slot, p) + j * 4) : zero);
			st4(dec_ptr(w, 0, COMP_U + p) + (q.yc + r) * DEC_STRIDE_C + q.xc + c0, ld4((cons_kind == 1 ? quad_rec_c<CU>(qs, cons_slot, p) : quad_pred_c<CU>(qs, cons_slot, p)) + r * NC + c0));
		}
		if (wnd_kind) {
			st4(tq_ptr(w, depth + 1, COMP_U + p) + ((q.abs_index << 4) >> 2) + j * 4, wnd_kind == 1 ? ld4(quad_lv_c<CU>(qs, wnd_slot, p) + j * 4) : zero);
			st4(dec_ptr(w, depth + 1, COMP_U + p) + (q.yc + r) * DEC_STRIDE_C + q.xc + c0, ld4((wnd_kind == 1 ? quad_rec_c<CU>(qs, wnd_slot, p) : quad_pred_c<CU>(qs, wnd_slot, p)) + r * NC + c0));
		}
		if (last_pred >= 0) *(uint32_t *)(w.pred_c[p] + (q.yc + r) * 32 + q.xc + c0) = *(const uint32_t *)(quad_pred_c<CU>(qs, last_pred, p) + r * NC + c0);
	};
	// the side info of the CU's units: put_consolidated_info's copy of the per-depth buffers into the CTU's record (with what the buffers held for `cons`), then the
	// buffers as `wnd` left them (set_enc_info_buffs), and encode_inter's reference-index write
	auto info = [&](int u) {
		const int k = q.abs_index + u;
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
	};
	constexpr int UNITS = CU * CU / 16;
	if constexpr (CU == 8) {
		if (g.tid < YL) luma(g.tid);
		else if (g.tid < YL + CL) chroma(g.tid - YL);
		else if (g.tid < YL + CL + UNITS) info(g.tid - YL - CL);
	} else {
		luma(g.tid);
		if (g.tid < CL) chroma(g.tid);
		else if (g.tid < CL + UNITS) info(g.tid - CL);
	}
	g.sync();
}

// check_rd_cost_merge (enc_ctu.h) on the slots' results: the same loop, statement for statement, with every evaluation a look-up and every copy deferred to
// quad_commit.  The slots' figures are fetched once (a slot per lane, then scalar registers) and the node's fields are written once at the end: the loop itself
// touches no memory.  `slots` from quad_prepare; inter_modes from get_merge_candidates.
template <int CU>
__device__ __forceinline__ uint32_t quad_merge_loop(const WaveGrp g, Enc &__restrict__ e, int ni, int slots, const QuadCands &mc, const uint8_t *inter_modes HENC_QPROF_ARG)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	QuadScratch &qs = quad_scratch(e);
	const int abs_index = q.abs_index, curr_depth = q.depth;
	const double weight = e.f->chroma_weight, avg_dist = e.f->avg_dist;
	// per slot: the coded evaluation's distortion, level sum, cost and cbf bits, the no-residual evaluation's distortion
	uint32_t s_dist[4], s_sum[4], s_cost[4], s_cbf[4], s_nores[4];
	{
		const QuadRes &r = qs.res[g.tid & 3];
		const uint32_t d = r.dist[0] + r.dist[1] + r.dist[2], sm = (uint32_t)(r.sum[0] + r.sum[1] + r.sum[2]);
		const uint32_t cst = (uint32_t)((double)d + cost_rd(avg_dist, sm));
		uint32_t nr = r.raw[0];
		nr += (uint32_t)(weight * r.raw[1]);
		nr += (uint32_t)(weight * r.raw[2]);
		const uint32_t cb = r.cbf[0] | (r.cbf[1] << 1) | (r.cbf[2] << 2);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			s_dist[k] = (uint32_t)__builtin_amdgcn_readlane((int)d, k);
			s_sum[k] = (uint32_t)__builtin_amdgcn_readlane((int)sm, k);
			s_cost[k] = (uint32_t)__builtin_amdgcn_readlane((int)cst, k);
			s_cbf[k] = (uint32_t)__builtin_amdgcn_readlane((int)cb, k);
			s_nores[k] = (uint32_t)__builtin_amdgcn_readlane((int)nr, k);
		}
	}
	auto of_slot = [](const uint32_t (&a)[4], int s) -> uint32_t { return s == 0 ? a[0] : (s == 1 ? a[1] : (s == 2 ? a[2] : a[3])); };
	uint32_t no_nores_mask = 0;      // merge_cand_buffer: bit cand
	int best_is_skip = 0, best_candidate = 0, have_ctu_cbf = 0, prev_nores_ran = 0;
	uint32_t dist, best_dist = MAX_COST, cost, best_cost = MAX_COST, best_sum = 0, ctu_cbf = 0;
	int best_x = 0, best_y = 0, best_ref_idx = 0;
	int cons_slot = 0, cons_kind = 0, wnd_slot = 0, wnd_kind = 0, last_pred = -1, coded_slot = -1;
	uint32_t nd_cbf = 0, nd_sum = 0;      // the node's inter_cbf bits and level sum as the evaluations leave them
#pragma unroll
	for (int cand = 0; cand < CFG_NUM_MERGE_CAND; cand++) {
		const int slot = quad_slot(slots, cand);
		int mc_done = 0;
		if (cand >= 1 && mc.x[cand] == mc.x[cand >= 1 ? cand - 1 : 0] && mc.y[cand] == mc.y[cand >= 1 ? cand - 1 : 0] && mc.ref[cand] == mc.ref[cand >= 1 ? cand - 1 : 0]) {
			const int coded_runs = !best_is_skip;
			if (coded_runs) no_nores_mask = (no_nores_mask & ~(1u << cand)) | (((no_nores_mask >> (cand >= 1 ? cand - 1 : 0)) & 1u) << cand);
			const int nores_runs = !(coded_runs && ((no_nores_mask >> cand) & 1u));
			if (!nores_runs || prev_nores_ran) {
				if (nores_runs) { nd_cbf = 0; nd_sum = 0; }
				prev_nores_ran = nores_runs;
				continue;
			}
		}
		prev_nores_ran = 0;
#pragma unroll
		for (int no_res = 0; no_res < 2; no_res++) {
			if (no_res == 1 && ((no_nores_mask >> cand) & 1u)) continue;
			if (best_is_skip && no_res == 0) continue;
			if (no_res == 1) prev_nores_ran = 1;
			if (!mc_done) { last_pred = slot; mc_done = 1; }
			if (no_res == 0) {
				// encode_inter: the node's fields, the windows and buffers of the CU's depth (deferred: wnd), the squared residuals (coded_slot)
				nd_cbf = of_slot(s_cbf, slot);
				nd_sum = of_slot(s_sum, slot);
				dist = of_slot(s_dist, slot);
				wnd_slot = slot; wnd_kind = 1; coded_slot = slot;
				cost = of_slot(s_cost, slot);
			} else {
				dist = of_slot(s_nores, slot);
				nd_cbf = 0;
				nd_sum = 0;
				cost = dist;
			}
			if (cost < best_cost) {
				best_x = mc.x[cand]; best_y = mc.y[cand];
				best_ref_idx = mc.ref[cand];
				best_candidate = cand;
				best_dist = dist;
				best_cost = cost;
				best_sum = nd_sum;
				if (no_res == 1) { wnd_slot = slot; wnd_kind = 2; }      // the prediction is the reconstruction, the levels are zero
				cons_slot = wnd_slot; cons_kind = wnd_kind;               // put_consolidated_info
				const uint32_t cb = wnd_kind == 1 ? of_slot(s_cbf, wnd_slot) : 0u;
				ctu_cbf = (cb | (cb >> 1) | (cb >> 2)) & 1u;               // cbf[0] | cbf[1] | cbf[2] of the per-depth buffers (each 0 or 1)
				have_ctu_cbf = 1;
				best_is_skip = (ctu_cbf & 1) == 0;
			}
			if (no_res == 0) {
				if (!have_ctu_cbf) { ctu_cbf = (uint32_t)c.cbf[0][abs_index] | c.cbf[1][abs_index] | c.cbf[2][abs_index]; have_ctu_cbf = 1; }
				if (((ctu_cbf >> curr_depth) & 1) == 0) no_nores_mask |= 1u << cand;
			}
		}
	}
	HENC_QPROF_MARK(e, 5);      // (the candidate loop)
	// what the evaluations left in the node and the context, then the windows, buffers and the record
	nd.inter_cbf[0] = (int32_t)(nd_cbf & 1u);
	nd.inter_cbf[1] = (int32_t)((nd_cbf >> 1) & 1u);
	nd.inter_cbf[2] = (int32_t)((nd_cbf >> 2) & 1u);
	nd.inter_tr_idx = 0;
	if (coded_slot >= 0) {
		const QuadRes &lr = qs.res[coded_slot];
		e.inter_ssq[0] = lr.raw[0]; e.inter_ssq[1] = lr.raw[1]; e.inter_ssq[2] = lr.raw[2];
		e.inter_ssq_valid = 1;
	}
	quad_commit<CU>(g, e, ni, curr_depth, cons_slot, cons_kind, wnd_slot, wnd_kind, last_pred, coded_slot >= 0);
	HENC_QPROF_MARK(e, 6);      // (the commit)
	nd.skipped = best_is_skip;
	nd.inter_mv.x = best_x;
	nd.inter_mv.y = best_y;
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
