// RD_FULL (rd_mode = 1): the bit estimates of the intra decisions - the mode search prices its candidates and the transform-tree walk prices every node with
// the CABAC bit counter (hmr_arithmetic_encoding.c:2139-2375: rd_encode_intra_dir_luma_ang, fast_rd_estimate_bits_intra_luma_mode, rd_estimate_bits_intra_mode,
// rd_est_intra_header, rd_transform_tree, rd_get_intra_bits_qt; the counter: hmr_binary_encoding.c:306-352, Cabac in counter mode).
//
// What the counter starts from.  Every estimate copies context states from `et->ee`, the real coder object the WPP thread selected last
// (wfpp_encode_select_bitstream, hmr_encoder_lib.c:2299) - the sub-stream of the CTU the thread entropy coded last, usually two rows up and three CTUs back
// (hmr_deblock_sao_pad_sync_ctu :2386), in whatever state that coder is when the decision runs.  enc_rdctx.h replays which sub-stream and how many of its CTUs
// that is for every CTU of a frame; the post-decision stage keeps the contexts after every coded CTU (PostPic::ctx_after); Enc::rd_ctx points at the states the
// CTU being decided has to copy.
//
// The shadow CTU.  The estimates walk `ctu_rd`, a copy of the CTU's descriptor whose side-info pointers the decision code re-aims at the per-depth buffers of the
// CU under evaluation before each call (hmr_motion_intra.c:1351-1361, :1457-1492, hmr_motion_intra_chroma.c:228-236, :417-432) - and leaves where they were
// otherwise: the chroma estimates read the luma direction through the pointer the LAST luma estimate left (Enc::rd_luma_depth).  Its prediction modes are all
// INTRA (motion_intra :2000), its partition sizes and prediction depths are written by encode_intra_luma (:1287) and consolidate_prediction_info.  One context
// crosses calls: rd_estimate_bits_intra_mode copies only the luma-direction context, so the chroma-direction context of the counter is the one the last full copy
// brought - until a counted bin sends it to state 0 (the counter's transition table is all zero): Enc::rd_chroma_state.
#pragma once
#include "enc_entropy.h"

#if defined(__HIPCC__)
#define HENC_RD_NOINLINE __attribute__((noinline))      // one compiled body of the counter's walk, not one per call site
#else
#define HENC_RD_NOINLINE
#endif

namespace henc {

struct RdViews {
	EntView v;
	CtuView c, l, t;
};
static_assert(sizeof(RdViews) <= sizeof(WorkRd::rd_views), "RdViews outgrew its place in the worker's RD area");
// the worker's one set of views (in its fast memory): filled by rd_make_views before every estimate
HENC_INLINE RdViews &rd_views_of(Enc &e) { return *(RdViews *)e.wrd->rd_views; }

// the shadow CTU's view: luma cbf / transform index buffers of `y_depth`, the luma directions where the last luma estimate left the pointer, the chroma cbf and
// direction buffers given
template <class G>
HENC_FI void rd_make_views(const G g, Enc &__restrict__ e, RdViews &r, int y_depth, int tr_depth_buf, const uint8_t *cbf_u, const uint8_t *cbf_v, int chroma_mode_depth,
			   const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	CtuView &c = r.c;
	c.cbf[0] = w.cbf_buffs[COMP_Y][y_depth];
	c.cbf[1] = cbf_u;
	c.cbf[2] = cbf_v;
	// the luma directions: the depth buffer the shadow CTU's pointer was left at - through the tokens that stand for values inherited from the CTU before
	// (enc_intra.h read_mode_buff) - or the CTU's own array
	if (e.rd_luma_depth >= 0) {
		for (int i = g.tid; i < NPART; i += g.n) {
			const int m = w.intra_mode_buffs[COMP_Y][e.rd_luma_depth][i];
			e.wrd->rd_luma_modes[i] = (uint8_t)((m & MODE_TOKEN) ? w.mode_in[COMP_Y][m & 7][i] : m);
		}
		g.sync();
		c.intra_mode[0] = e.wrd->rd_luma_modes;
	} else c.intra_mode[0] = e.ctu->intra_mode[0];
	c.intra_mode[1] = w.intra_mode_buffs[COMP_CHR][chroma_mode_depth];
	c.tr_idx = w.tr_idx_buffs[tr_depth_buf];
	c.pred_depth = e.wrd->rd_pred_depth;
	c.part_size_type = e.wrd->rd_part_size;
	c.pred_mode = e.wrd->rd_pred_mode;
	c.inter_mode = c.skipped = c.merge = c.merge_idx = c.mv_diff_ref_idx = nullptr;
	c.qp = nullptr;
	c.mv_diff = nullptr;
	c.x = e.ctu_x; c.y = e.ctu_y;
	EntView &v = r.v;
	v.seq = &*e.seq; v.f = &*e.f; v.T = e.T; v.geo = e.geo;
	v.c = &r.c;
	CtuPublic *lp = ctu_left_of(e), *tp = ctu_top_of(e);
	if (lp) r.l = view_of(*lp);
	if (tp) r.t = view_of(*tp);
	v.left = lp ? &r.l : nullptr;
	v.top = tp ? &r.t : nullptr;
	v.coeff[0] = coef_y; v.coeff[1] = coef_u; v.coeff[2] = coef_v;
	v.n = 0;
	v.prev_last_qp = -1;
}

// fast_rd_estimate_bits_intra_luma_mode :2186 for a direction that is one of the three candidates (the caller charges 6 bits otherwise): the flag on the luma
// direction context as et->ee has it, then one or two bypass bins for the FIRST candidate that matches (rd_encode_intra_dir_luma_ang :2139)
HENC_INLINE uint32_t rd_bits_luma_mode_in_preds(const Enc &e, int dir, const int *preds)
{
	int idx = -1;
	for (int i = 0; i < 3; i++)
		if (dir == preds[i]) { idx = i; break; }
	uint64_t frac = kEntropyBits[e.rd_ctx[CTX_INTRA_PRED] ^ (idx != -1 ? 1u : 0u)];
	if (idx != -1) frac += idx ? 2 * 32768u : 32768u;
	else frac += 5 * 32768u;
	return (uint32_t)(frac >> 15);
}

// encode_intra_dir_luma_ang :838 for one partition (is_multiple = FALSE): the LAST matching candidate counts
HENC_FI void rd_code_luma_dir(Cabac &ec, const EntView &v, int ni)
{
	int dir = uni(v.c->intra_mode[0][v.geo[ni].abs_index]), preds[3], pred_idx = -1;
	ent_intra_preds(v, ni, preds);
	for (int i = 0; i < 3; i++)
		if (dir == preds[i]) pred_idx = i;
	ec.encode_bin(CTX_INTRA_PRED, pred_idx != -1 ? 1 : 0);
	if (pred_idx != -1) {
		ec.encode_ep(pred_idx ? 1 : 0);
		if (pred_idx) ec.encode_ep(pred_idx - 1);
	} else {
		if (preds[0] > preds[1]) { const int t = preds[0]; preds[0] = preds[1]; preds[1] = t; }
		if (preds[0] > preds[2]) { const int t = preds[0]; preds[0] = preds[2]; preds[2] = t; }
		if (preds[1] > preds[2]) { const int t = preds[1]; preds[1] = preds[2]; preds[2] = t; }
		for (int i = 2; i >= 0; i--) dir = dir > preds[i] ? dir - 1 : dir;
		ec.encode_bins_ep(dir, 5);
	}
}
// encode_intra_dir_chroma :907; `state`: the chroma direction context of the counter (see the head of the file), nullptr: ec's own
HENC_FI void rd_code_chroma_dir(Cabac &ec, const EntView &v, int ni)
{
	const int abs_index = v.geo[ni].abs_index;
	uint32_t chroma = uni(v.c->intra_mode[1][abs_index]);
	if (chroma == DM_CHROMA_IDX) { ec.encode_bin(CTX_CHROMA_PRED, 0); return; }
	int list[5];
	const int luma = uni(v.c->intra_mode[0][abs_index]);
	list[0] = PLANAR_IDX; list[1] = VER_IDX; list[2] = HOR_IDX; list[3] = DC_IDX; list[4] = DM_CHROMA_IDX;
	for (int i = 0; i < 4; i++)
		if (luma == list[i]) { list[i] = 34; break; }
	for (int i = 0; i < 4; i++)
		if ((int)chroma == list[i]) { chroma = i; break; }
	ec.encode_bin(CTX_CHROMA_PRED, 1);
	ec.encode_bins_ep(chroma, 2);
}

// rd_transform_tree :2239: the luma OR the chroma syntax of the transform tree under `top_ni`
template <class G>
HENC_FI void rd_transform_tree(const G g, Cabac &ec, const EntView &v, EntScratch &sc, int top_ni, int is_luma)
{
	const Seq &S = *v.seq;
	const CtuView *c = v.c;
	const int depth = v.geo[top_ni].depth;
	DepthState depth_state;
	int curr = top_ni, parent = top_ni, curr_depth = depth;
	while (curr_depth != depth || depth_state.get(curr_depth) != 1) {
		const Geo &q = v.geo[curr];
		curr_depth = q.depth;
		const int abs_index = q.abs_index;
		const int shift = CFG_MAX_CU_SHIFT - curr_depth;
		const int pred_depth = uni(c->pred_depth[abs_index]), tr_depth = curr_depth - pred_depth, first = tr_depth == 0;
		const int tr_idx = uni(c->tr_idx[abs_index]);
		const int is_intra = uni(c->pred_mode[abs_index]) == PM_INTRA, part = uni(c->part_size_type[abs_index]);
		const int split_flag = (tr_idx + pred_depth) > curr_depth;
		const int log2_tr = shift, log2_cu = CFG_MAX_CU_SHIFT - pred_depth;
		const int max_tr = is_intra ? uni(S.max_intra_tr_depth) : uni(S.max_inter_tr_depth), nxn = is_intra && part == PART_NxN;
		int tu_min_in_cu;
		if (log2_cu < uni(S.min_tu_size_shift) + max_tr - 1 + nxn) tu_min_in_cu = uni(S.min_tu_size_shift);
		else {
			tu_min_in_cu = log2_cu - (max_tr - 1 + nxn);
			if (tu_min_in_cu > uni(S.max_tu_size_shift)) tu_min_in_cu = uni(S.max_tu_size_shift);
		}
		if (is_luma && !(is_intra && part == PART_NxN && curr_depth == pred_depth) && !(!is_intra && part != PART_2Nx2N && curr_depth == pred_depth) &&
		    !(log2_tr > uni(S.max_tu_size_shift)) && !(log2_tr == uni(S.min_tu_size_shift)) && !(log2_tr == tu_min_in_cu))
			ec.encode_bin(CTX_TRANS_SUBDIV + 5 - shift, split_flag);
		if (!is_luma && (first || shift > 2)) {
			if (first || HENC_CBF(c, abs_index, 1, tr_depth - 1)) encode_qt_cbf(ec, 1, tr_depth, HENC_CBF(c, abs_index, 1, tr_depth));
			if (first || HENC_CBF(c, abs_index, 2, tr_depth - 1)) encode_qt_cbf(ec, 2, tr_depth, HENC_CBF(c, abs_index, 2, tr_depth));
		}
		depth_state.inc(curr_depth);
		if (split_flag) {
			parent = curr;
			curr_depth++;
		} else {
			if (is_luma) {
				const uint32_t cbf_y = HENC_CBF(c, abs_index, 0, tr_depth);
				encode_qt_cbf(ec, 0, tr_depth, cbf_y);
				HENC_TRACE("  CBF d=%d abs=%d comp=0 trd=%d cbf=%d frac=%llu\n", q.depth, abs_index, tr_depth, (int)cbf_y, (unsigned long long)ec.frac_bits);
				if (cbf_y) { encode_residual(g, ec, v, sc, curr, 0); HENC_TRACE("  RES d=%d abs=%d comp=0 frac=%llu\n", q.depth, abs_index, (unsigned long long)ec.frac_bits); }
			} else {
				const bool here = shift > 2 || q.list_index == v.geo[v.geo[q.parent].child[0]].list_index + 3;      // (4 x 4 luma: the chroma of the four with the last one)
				if (here)
					for (int comp = 1; comp < 3; comp++)
						if (HENC_CBF(c, abs_index, comp, tr_depth)) encode_residual(g, ec, v, sc, curr, comp);
			}
			while (depth_state.get(curr_depth) == 4) {
				depth_state.set(curr_depth, 0);
				parent = v.geo[parent].parent;
				curr_depth--;
			}
			if (curr_depth == 0 && depth_state.get(curr_depth) == 1) break;
		}
		curr = v.geo[parent].child[depth_state.get(curr_depth)];
	}
}

// rd_get_intra_bits_qt :2362: all contexts from et->ee, the header of the partition and the luma or chroma part of its transform tree
template <class G>
HENC_RD_NOINLINE HENC_HD uint32_t rd_get_intra_bits_qt(const G g, Enc &__restrict__ e, const RdViews &r, int ni, int is_luma)
{	ni = uni(ni); is_luma = uni(is_luma);

	HENC_ENC_IN_LDS(e);
	Cabac ec;
	ec.counter = true;
	ec.ctx = e.wrd->rd_ctx_work;
	for (int i = g.tid; i < CTX_TOTAL; i += g.n) e.wrd->rd_ctx_work[i] = e.rd_ctx[i];
	g.sync();
	ec.load_ctx(g);
	e.rd_chroma_state = e.rd_ctx[CTX_CHROMA_PRED];      // (a full copy: the counter's chroma direction context is et->ee's again)
	const EntView &v = r.v;
	if (is_luma) {
		// rd_est_intra_header :2214: the partition size only with the CTU's first partition
		if (v.geo[ni].abs_index == 0) {
			const int min_cu_depth = uni(v.seq->max_cu_depth) - uni(v.seq->mincu_mintr_shift_diff);
			if (v.geo[ni].depth == min_cu_depth) ec.encode_bin(CTX_PART_SIZE, uni(v.c->part_size_type[0]) == PART_2Nx2N ? 1 : 0);
		}
		rd_code_luma_dir(ec, v, ni);
		HENC_TRACE("  DIR d=%d abs=%d dir=%d frac=%llu\n", v.geo[ni].depth, v.geo[ni].abs_index, (int)v.c->intra_mode[0][v.geo[ni].abs_index], (unsigned long long)ec.frac_bits);
	} else {
		rd_code_chroma_dir(ec, v, ni);
		e.rd_chroma_state = 0;
	}
	rd_transform_tree(g, ec, v, e.wrd->rd_ent, ni, is_luma);
	HENC_TRACE("RDQT ctu=%d d=%d abs=%d pd=%d luma=%d bits=%u", e.ctu->ctu_number, v.geo[ni].depth, v.geo[ni].abs_index, (int)v.c->pred_depth[v.geo[ni].abs_index], is_luma, ec.bitcnt());
#if !defined(__HIPCC__) && defined(HENC_TRACE_ENABLE)
	if (getenv("HOMER_RDTRACE_CTX")) {
		HENC_TRACE(" ctx ");
		for (int i = 0; i < CTX_TOTAL; i++) HENC_TRACE("%02x", e.rd_ctx[i]);
	}
#endif
	HENC_TRACE("\n");
	return ec.bitcnt();
}

// rd_estimate_bits_intra_mode :2198 for chroma: only the LUMA direction context is copied from et->ee; the chroma direction flag is counted on what the counter
// holds (Enc::rd_chroma_state), which then falls to state 0
HENC_FI uint32_t rd_bits_chroma_mode(Enc &__restrict__ e, const EntView &v, int ni)
{
	HENC_ENC_IN_LDS(e);
	const int abs_index = v.geo[ni].abs_index;
	const uint32_t chroma = uni(v.c->intra_mode[1][abs_index]);
	uint64_t frac = kEntropyBits[e.rd_chroma_state ^ (chroma == DM_CHROMA_IDX ? 0u : 1u)];
	if (chroma != DM_CHROMA_IDX) frac += 2 * 32768u;
	e.rd_chroma_state = 0;
	HENC_TRACE("RDCM ctu=%d d=%d abs=%d luma=0 bits=%u\n", e.ctu->ctu_number, v.geo[ni].depth, abs_index, (uint32_t)(frac >> 15));
	return (uint32_t)(frac >> 15);
}

}  // namespace henc
