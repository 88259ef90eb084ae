// Intra coding of a CU: mode search, transform tree, chroma.
// Restates hmr_motion_intra.c:970-1630 (encode_intra_cu, homer_loop1_motion_intra, encode_intra_luma),
// hmr_motion_intra_chroma.c:92-469 (encode_intra_chroma) and hmr_motion_intra.c:1731-1757 (encode_intra)
// for rd_mode != RD_FULL (BASELINE configs 1-4; the full-RDO bit estimators are a later row).
#pragma once
#include "enc_common.h"
#include "enc_rdo.h"

namespace henc {

HENC_INLINE int intra_is_filtered(int mode, int inv_depth)
{
	static constexpr int intra_filter[5] = {10, 7, 1, 0, 10};   // hmr_motion_intra.c:148
	const int diff = hmin(habs(mode - HOR_IDX), habs(mode - VER_IDX));
	return (mode != DC_IDX) && (diff > intra_filter[inv_depth - 2]);
}

// fill_reference_samples for the node's block of component class `comp` in decoded window `wnd` (+ smoothing when asked)
template <class G>
HENC_HD void node_fill_refs(const G g, Enc &__restrict__ e, int ni, int wnd, int comp, int want_filtered)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	const int is_y = comp == COMP_Y;
	const int n = is_y ? q.size : q.size_chroma, x = is_y ? q.x : q.xc, y = is_y ? q.y : q.yc;
	const int pw = is_y ? e.seq->width : e.seq->width >> 1, ph = is_y ? e.seq->height : e.seq->height >> 1;
	const int cx = is_y ? e.ctu_x : e.ctu_x >> 1, cy = is_y ? e.ctu_y : e.ctu_y >> 1;
	const int bl_size = hmin(n, ph - (cy + y + n)), tr_size = hmin(n, pw - (cx + x + n));
	const int st = dec_stride(comp);
	const int16_t *corner = dec_ptr(*e.w, wnd, comp) + (y - 1) * st + (x - 1);
	intra_fill_refs(g, corner, st, n, nd.left_nb, nd.top_nb, nd.left_bottom_nb, nd.top_right_nb, bl_size, tr_size, is_y ? e.w->adi : e.adi_c);
	if (want_filtered) intra_adi_filter(g, e.w->adi, e.w->adi_f, n, e.seq->strong_intra);
}

// most probable modes as the DECISION code sees them (homer_loop1_motion_intra, hmr_motion_intra.c:1102-1104 through
// get_intra_dir_luma_predictor, hmr_arithmetic_encoding.c:545): the left / top unit inside the CTU is looked up in the
// worker's mode buffer of the PU's depth and always counts as intra (the worker's shadow CTU keeps pred_mode == INTRA
// from the first frame on); units of other CTUs use those CTUs' final arrays.  The worker's buffer is NOT reset between
// CTUs: for a neighbour that ended up inter-coded it holds whatever earlier CUs left there, which is part of the
// reference's single-thread behaviour and therefore of the bitstream.
// A worker that runs one CTU row cannot have that state (it comes from the CTU before in raster order, which for a row start is the
// end of the row above).  ctu_begin therefore fills the buffers with tokens that stand for "the inherited value", the copies between
// buffers and CTU arrays carry tokens along, and a look-up that lands on a token uses the scheduler's guess (Work::mode_in).  A mode
// search that used such a guess logs where its two neighbour directions came from and the SAD of every mode it tried
// (SearchLog); enc_sched.h replays the search walk on that table with the true directions and re-encodes the CTU only when
// the winner, its cost or its bit cost would have been different.
HENC_INLINE int read_mode_buff(Enc &__restrict__ e, int depth, uint32_t idx, uint16_t *src)
{
	HENC_ENC_IN_LDS(e);
	const int v = e.w->intra_mode_buffs[COMP_Y][depth][idx];
	if (!(v & MODE_TOKEN)) { *src = (uint16_t)v; return v; }
	const int d = v & 7;
	*src = (uint16_t)(0x8000 | (d << 8) | idx);
	return e.w->mode_in[COMP_Y][d][idx];
}
// get_intra_dir_luma_predictor, hmr_arithmetic_encoding.c:545-590: the candidate list from the two neighbour directions
HENC_INLINE void mpm_from_dirs(int left_dir, int top_dir, int *preds)
{
	if (left_dir == top_dir) {
		if (left_dir > 1) {
			preds[0] = left_dir;
			preds[1] = ((left_dir + 29) % 32) + 2;
			preds[2] = ((left_dir - 1) % 32) + 2;
		} else {
			preds[0] = PLANAR_IDX; preds[1] = DC_IDX; preds[2] = VER_IDX;
		}
	} else {
		preds[0] = left_dir;
		preds[1] = top_dir;
		if (left_dir && top_dir) preds[2] = PLANAR_IDX;
		else preds[2] = (left_dir + top_dir) < 2 ? VER_IDX : DC_IDX;
	}
}
HENC_INLINE void intra_neighbour_dirs(Enc &__restrict__ e, int ni, int depth, int *dirs, uint16_t *src)
{
	HENC_ENC_IN_LDS(e);
	uint32_t idx = 0;
	CtuPublic *cl = pu_left(e, ni, &idx);
	dirs[0] = dirs[1] = DC_IDX;
	src[0] = src[1] = DC_IDX;
	if (cl == e.ctu) { if (e.w->thread_seen_intra) dirs[0] = read_mode_buff(e, depth, idx, &src[0]); }   // (else: the shadow CTU says "not intra" -> DC)
	else if (cl) src[0] = (uint16_t)(dirs[0] = cl->pred_mode[idx] == PM_INTRA ? cl->intra_mode[COMP_Y][idx] : DC_IDX);
	CtuPublic *ct = pu_top(e, ni, &idx, 1);
	if (ct == e.ctu) { if (e.w->thread_seen_intra) dirs[1] = read_mode_buff(e, depth, idx, &src[1]); }
	else if (ct) src[1] = (uint16_t)(dirs[1] = ct->pred_mode[idx] == PM_INTRA ? ct->intra_mode[COMP_Y][idx] : DC_IDX);
}

// The walk of homer_loop1_motion_intra (hmr_motion_intra.c:1084-1180) over the prediction directions: planar / DC, five coarse
// angles, +-2 / +-4 around the best, +-1 around that.  sad_of(mode) returns the SAD of a direction or a negative value when it is not
// available (only the replay in enc_sched.h can fail).  Returns the bit cost of the winner, or -1.
template <class SadsFn, class BitsFn>
HENC_INLINE int intra_search_walk_batched(const int *preds, int rd_fast, double sqrt_lambda, SadsFn &&sads_of, int *best_mode_out, double *best_cost_out, BitsFn &&full_bits_of,
					  int *modes_buf = nullptr, int64_t *sads_buf = nullptr)
{
	static constexpr int search_points[4][5] = {{0, 1, 0, 8, 16}, {2, 10, 16, 22, 30}, {-4, -2, 2, 4, 0}, {-1, 1, 0, 0, 0}};
	static constexpr int num_search_points[4] = {2, 5, 4, 2};
	int best_cu_mode = 0, new_best = 0, min_mode = 0, max_mode = 1, best_bit_cost = 0;
	double best_cost = MAX_COST;
	for (int loop = 0; loop < 4; loop++) {
		if (loop == 1) { best_cu_mode = 2; min_mode = 2; max_mode = 34; }
		// the candidates of a round do not depend on each other: their SADs may be computed side by side, the comparison below keeps the reference's order
		int modes_local[5], cnt = 0;
		int64_t sads_local[5];
		int *modes = modes_buf ? modes_buf : modes_local;      // (the worker passes a place in its fast memory: the lists are indexed at run time)
		int64_t *sads = sads_buf ? sads_buf : sads_local;
		for (int k = 0; k < num_search_points[loop]; k++) {
			const int mode = best_cu_mode + search_points[loop][k];
			if (mode < min_mode || mode > max_mode) continue;
			modes[cnt++] = mode;
		}
		if (!sads_of(modes, cnt, sads)) return -1;
		for (int k = 0; k < cnt; k++) {
			const int mode = modes[k];
			double cost = (double)(uint32_t)sads[k];
			int bit_cost = 0;
			if (rd_fast == 1) {
				bit_cost = (preds[0] == mode || preds[1] == mode || preds[2] == mode) ? 1 : 12;
				cost += bit_cost * sqrt_lambda;
			} else if (rd_fast == 2) {      // RD_FULL :1140: the counter's bits for a candidate direction, 6 for any other ("introduces some error but gives good results")
				bit_cost = (preds[0] == mode || preds[1] == mode || preds[2] == mode) ? (int)full_bits_of(mode) : 6;
				cost += bit_cost * sqrt_lambda;
			}
			if (cost < best_cost) { best_cost = cost; new_best = mode; best_bit_cost = bit_cost; }
		}
		best_cu_mode = new_best;
	}
	*best_mode_out = best_cu_mode;
	*best_cost_out = best_cost;
	return best_bit_cost;
}
template <class SadFn>
HENC_INLINE int intra_search_walk(const int *preds, int rd_fast, double sqrt_lambda, SadFn &&sad_of, int *best_mode_out, double *best_cost_out)
{
	return intra_search_walk_batched(preds, rd_fast, sqrt_lambda, [&](const int *modes, int cnt, int64_t *sads) -> bool {
		for (int k = 0; k < cnt; k++) {
			sads[k] = sad_of(modes[k]);
			if (sads[k] < 0) return false;
		}
		return true;
	}, best_mode_out, best_cost_out, [](int) { return 0u; });
}

// what encode_intra_luma returns for rd_mode != RD_FULL: the transform tree's cost plus the mode bits (hmr_motion_intra.c:1625)
HENC_INLINE uint32_t intra_luma_cost(uint32_t tu_cost, int mode_bits, double correction) { return (uint32_t)(tu_cost + mode_bits * correction + .5); }

// homer_loop1_motion_intra, hmr_motion_intra.c:1084-1180.  Returns the bit cost of the winner; *best_mode / *best_cost out.
template <class G>
HENC_HD int intra_mode_search(const G g, Enc &__restrict__ e, int ni, int depth, int *best_mode_out, double *best_cost_out)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	Work &w = *e.w;
	const int n = q.size, curr_depth = q.depth, inv_depth = CFG_MAX_CU_SHIFT - curr_depth;
	node_fill_refs(g, e, ni, depth + 1, COMP_Y, 1);
	int preds[3], dirs[2];
	uint16_t src[2];
	e.rd_luma_depth = curr_depth;      // (homer_loop1_motion_intra :1103 aims the shadow CTU's luma directions at this depth's buffer "for rd")
	intra_neighbour_dirs(e, ni, curr_depth, dirs, src);
	mpm_from_dirs(dirs[0], dirs[1], preds);
	const int rd_fast = e.seq->rd_mode == RDM_FAST ? 1 : (e.seq->rd_mode == RDM_FULL ? 2 : 0);      // (how the walk prices a direction)
	// a search whose candidate list rests on a guess is logged (every lane writes the same values)
	SearchLog *lg = nullptr;
	e.last_slog = -1;
	if (rd_fast == 1 && ((src[0] | src[1]) & 0x8000)) {
		if (e.n_spec_reads < MAX_SEARCH_LOGS) {
			lg = e.ctu_g->slog + e.n_spec_reads;
			lg->src[0] = src[0]; lg->src[1] = src[1];
			lg->used[0] = (uint8_t)dirs[0]; lg->used[1] = (uint8_t)dirs[1];
			lg->n = 0;
			lg->has_cmp = 0;
			lg->tu_cost = 0;
			e.last_slog = e.n_spec_reads;
		}
		e.n_spec_reads++;
	}
	const src_t *orig = w.curr_y + q.y * CTU_STRIDE_Y + q.x;
	return intra_search_walk_batched(preds, rd_fast, e.f->sqrt_lambda, [&](const int *modes, int cnt, int64_t *sads) -> bool {
		// with helper wavefronts: rounds of 1 + NHELP candidates, the worker always taking the last one of the round (so that the prediction
		// left in the window is the one the serial order leaves there); the helpers only return the SAD
		for (int k0 = 0; k0 < cnt;) {
			const int take = HENC_HELPERS(e) ? hmin(1 + NHELP, cnt - k0) : 1, mine = k0 + take - 1;
			for (int j = 0; j < take - 1; j++) helper_post(g, e, j, HJOB_INTRA_SAD, ni, n, modes[k0 + j], intra_is_filtered(modes[k0 + j], inv_depth));
			{
				const int mode = modes[mine], filt = intra_is_filtered(mode, inv_depth);
				// (the prediction is not stored: the transform tree that follows predicts every sample of this block again before anything reads the window)
				sads[mine] = (int64_t)intra_predict_sad(g, (pred_t *)nullptr, 0, orig, CTU_STRIDE_Y, filt ? w.adi_f : w.adi, n, mode, 1);
			}
			for (int j = 0; j < take - 1; j++) {
				helper_wait(g, e, j);
				sads[k0 + j] = (int64_t)e.box->r[j][0];
			}
			k0 += take;
		}
		if (lg)
			for (int k = 0; k < cnt; k++) {
				lg->mode[lg->n] = (uint8_t)modes[k];
				lg->sad[lg->n] = (uint32_t)sads[k];
				lg->n++;
			}
		return true;
	}, best_mode_out, best_cost_out, [&](int mode) { return rd_bits_luma_mode_in_preds(e, mode, preds); }, w.srch_modes, w.srch_sads);
}

template <class G>
HENC_HD void set_intra_info_buffs(const G g, Enc &__restrict__ e, int depth, int ni)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	Work &w = *e.w;
	for (int i = g.tid; i < q.num_part; i += g.n) {
		w.cbf_buffs[COMP_Y][depth][q.abs_index + i] = (uint8_t)nd.intra_cbf[COMP_Y];
		w.tr_idx_buffs[depth][q.abs_index + i] = (uint8_t)nd.intra_tr_idx;
		w.intra_mode_buffs[COMP_Y][depth][q.abs_index + i] = (uint8_t)nd.intra_mode[COMP_Y];
	}
	g.sync();
}

// encode_intra_cu, hmr_motion_intra.c:973-1071: one luma TU.  depth = prediction depth.  Returns the SSD, *curr_sum the level sum.
template <class G>
HENC_HD uint32_t encode_intra_tu(const G g, Enc &__restrict__ e, int ni, int depth, int cu_mode, int part_size_type, int *curr_sum)
{
	HENC_ENC_IN_LDS(e);
	const Geo &q = e.geo[ni];
	Node &nd = node_of(e, ni);
	Work &w = *e.w;
	const int curr_depth = q.depth, n = q.size;
	const int scan_mode = find_scan_mode(1, 1, n, cu_mode, 0);
	const int per = nd.qp / 6, rem = nd.qp % 6;
	const int wnd = curr_depth + 1;
	pred_t *pred = w.pred_y + q.y * CTU_STRIDE_Y + q.x;
	const src_t *orig = w.curr_y + q.y * CTU_STRIDE_Y + q.x;
	int16_t *quant = tq_ptr(w, wnd, COMP_Y) + (q.abs_index << 4), *iquant = iq_slot(w, COMP_Y, q.abs_index << 4, e.on_helper);
	int16_t *dec = dec_ptr(w, wnd, COMP_Y) + q.y * DEC_STRIDE_Y + q.x;
	const int inv_depth = CFG_MAX_CU_SHIFT - curr_depth;
	const int filt = intra_is_filtered(cu_mode, inv_depth);
	node_fill_refs(g, e, ni, wnd, COMP_Y, filt);
	intra_predict(g, pred, CTU_STRIDE_Y, filt ? w.adi_f : w.adi, n, cu_mode, 1);
	// transform chain in fast memory (see encode_inter_tu): the residual source - prediction formed by the transform's first stage, levels in the block's slot of
	// the dequantised-coefficient buffer (to the window in HBM when final), the reconstructed residual where the rounding remainders were
	tr_forward(g, HENC_FT(e), e.T, orig, CTU_STRIDE_Y, pred, CTU_STRIDE_Y, w.pred_aux, w.delta_u, n, cu_mode != REG_DCT);
	const int sum = quantize(g, HENC_FT(e), e.T, w.pred_aux, iquant, w.delta_u, scan_mode, curr_depth, COMP_Y, 1, e.f->slice_type == SLICE_I, e.seq->sign_hiding, n, per, rem);
	*curr_sum = sum;
	const int tr = curr_depth - depth + (part_size_type == PART_NxN);
	nd.sum = (uint32_t)sum;
	nd.intra_cbf[COMP_Y] = (sum ? 1 : 0) << tr;
	nd.intra_tr_idx = tr;
	nd.intra_mode[COMP_Y] = cu_mode;
	if (e.seq->rd_mode == RDM_FULL) set_intra_info_buffs(g, e, curr_depth, ni);      // :1041: what the bit estimate of this node reads
	if (sum) {
		lin_copy_nosync(g, iquant, quant, n * n);
		dequantize(g, HENC_FT(e), e.T, iquant, iquant, curr_depth, COMP_Y, 1, n, per, rem);
		tr_inverse(g, HENC_FT(e), e.T, w.delta_u, n, iquant, w.pred_aux, n, cu_mode != REG_DCT);
		return blk_reconst_ssd(g, pred, CTU_STRIDE_Y, w.delta_u, n, orig, CTU_STRIDE_Y, dec, DEC_STRIDE_Y, n);
	}
	lin_zero_nosync(g, quant, n * n);
	return blk_reconst_ssd(g, pred, CTU_STRIDE_Y, (const int16_t *)nullptr, 0, orig, CTU_STRIDE_Y, dec, DEC_STRIDE_Y, n);
}

// encode_intra_luma, hmr_motion_intra.c:1229-1630 (non-HM path): search, then the transform tree of the winner.
template <class G>
HENC_WALK_FN HENC_HD uint32_t encode_intra_luma(const G g, Enc &__restrict__ e, int depth, int part_position, int part_size_type)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	Work &w = *e.w;
	const int top_ni = node_at(e, depth, part_position);
	const uint32_t qp = node_of(e, top_ni).qp;
	int cu_mode;
	double search_cost;
	int bitcost_cu_mode;
	const bool rd_full = S.rd_mode == RDM_FULL;
	if (rd_full) {      // :1287: the shadow CTU's partition size and prediction depth of this CU
		const Geo &tq = e.geo[top_ni];
		bytes_set(g, &e.wrd->rd_part_size[tq.abs_index], part_size_type, tq.num_part);
		bytes_set(g, &e.wrd->rd_pred_depth[tq.abs_index], depth - (part_size_type == PART_NxN), tq.num_part);
	}
	{
		HENC_PROF_T0();
		// (the P-slice walk may have had the helper run this search under the CU's inter evaluation: enc_common.h bg_post; a search of another node is stopped first - the
		// neighbour arrays it fills are the ones the search here and the TUs below use)
		bool taken = false;
		if constexpr (G::bg) taken = bg_take(g, e, top_ni, &cu_mode, &bitcost_cu_mode);
		if (taken) { search_cost = 0; e.last_slog = -1; }
		else bitcost_cu_mode = intra_mode_search(g, e, top_ni, depth, &cu_mode, &search_cost);
		HENC_PROF_ADD(e, PF_INTRA_SEARCH);
	}

	int parent, curr, initial_state, end_state;
	if (depth == 0 && CFG_MAX_CU_SIZE == 64) {
		parent = cfg_depth_start(0);
		curr = e.geo[parent].child[0];
		node_of(e, parent).cost = 0x7fffffff;
		initial_state = part_position & 3;
		end_state = initial_state;
	} else {
		curr = top_ni;
		parent = e.geo[curr].parent;
		initial_state = part_position & 3;
		end_state = initial_state + 1;
	}
	int curr_depth = e.geo[curr].depth;
	const int log2cu_size = CFG_MAX_CU_SHIFT - (depth - (part_size_type == PART_NxN));
	int cu_min_tu_size_shift;
	if (log2cu_size < S.min_tu_size_shift + S.max_intra_tr_depth - 1 + (part_size_type == PART_NxN)) cu_min_tu_size_shift = S.min_tu_size_shift;
	else {
		cu_min_tu_size_shift = log2cu_size - (S.max_intra_tr_depth - 1 + (part_size_type == PART_NxN));
		if (cu_min_tu_size_shift > 5) cu_min_tu_size_shift = 5;
	}
	int max_tr_processing_depth = CFG_MAX_CU_SHIFT - cu_min_tu_size_shift;
	if (S.perf_mode >= 1)
		max_tr_processing_depth = (depth + 2 <= max_tr_processing_depth) ? depth + 2 : ((depth + 1 <= max_tr_processing_depth) ? depth + 1 : max_tr_processing_depth);

	DepthState depth_state;
	depth_state.set(curr_depth, initial_state);
	// The four TUs of a split at the tree's last level are compared with their parent only as sums (distortion and level sum, below), and both only grow from child
	// to child: once the children evaluated so far cannot beat the parent any more the rest of them cannot change the decision.  The reference evaluates them anyway;
	// what that leaves behind - their nodes' fields, the inside of the deeper window - is overwritten before anything reads it (the losing branch below restores the
	// window's border from the parent).  Not with RD_FULL: its bit estimates have side effects of their own (enc_rdo.h).
	const bool split_early_out = !rd_full;
	double split_cost = 0;
	uint32_t split_sum = 0;
	bool split_lost = false;
	while (curr_depth != depth || depth_state.get(curr_depth) != end_state) {
		curr = parent < 0 ? curr : e.geo[parent].child[depth_state.get(curr_depth)];
		if (e.geo[curr].depth >= 1) nodes_select_quad(g, e, e.geo[curr].abs_index >> 6);      // (the transform tree of the 64 x 64 CU goes through all four quadrants)
		Node &cn = node_of(e, curr);
		cn.qp = qp;
		curr_depth = e.geo[curr].depth;
		int curr_sum = 0;
		cn.distortion = encode_intra_tu(g, e, curr, depth, cu_mode, part_size_type, &curr_sum);
		cn.sum = (uint32_t)curr_sum;
		cn.cost = cn.distortion;
		if (split_early_out && curr_depth == max_tr_processing_depth && curr_depth > depth && parent >= 0 && e.geo[parent].depth == curr_depth - 1) {
			if (depth_state.get(curr_depth) == 0) { split_cost = 0; split_sum = 0; }
			split_cost += (double)cn.distortion;
			split_sum += cn.sum;
			const Node &spn = node_of(e, parent);
			const bool lost = S.rd_mode != RDM_FAST ? !(split_cost < (double)spn.cost)
								  : !(1.25 * (split_cost + (double)(uint32_t)(45u * split_sum)) < (double)(uint32_t)(spn.cost + 45u * spn.sum));
			if (lost && depth_state.get(curr_depth) < 3) {
				split_lost = true;
				depth_state.set(curr_depth, 3);      // (the increment below makes it 4: the split is closed)
			}
		}
		if (rd_full && (curr_depth < max_tr_processing_depth || curr_depth == depth)) {      // :1457: the node's syntax priced by the bit counter
			RdViews &rv = rd_views_of(e);
			e.rd_luma_depth = curr_depth;
			rd_make_views(g, e, rv, curr_depth, curr_depth, nullptr, nullptr, 0, tq_ptr(w, curr_depth + 1, COMP_Y), nullptr, nullptr);
			const uint32_t bit_cost = rd_get_intra_bits_qt(g, e, rv, curr, 1);
			cn.cost += (uint32_t)(bit_cost * e.f->lambda + .5);
		}
		depth_state.inc(curr_depth);
		if (curr_depth < max_tr_processing_depth) {
			curr_depth++;
			parent = curr;
		} else if (depth_state.get(curr_depth) == 4) {
			while (depth_state.get(curr_depth) == 4 && curr_depth > depth) {
				const Geo &pq = e.geo[parent];
				Node &pn = node_of(e, parent);
				Node &c0 = node_of(e, pq.child[0]), &c1 = node_of(e, pq.child[1]), &c2 = node_of(e, pq.child[2]), &c3 = node_of(e, pq.child[3]);
				const uint32_t sum = c0.sum + c1.sum + c2.sum + c3.sum;
				const double distortion = (double)c0.distortion + c1.distortion + c2.distortion + c3.distortion;
				double cost = distortion;
				depth_state.set(curr_depth, 0);
				if (rd_full) {      // :1486: the parent's syntax with its four children as the transform split
					RdViews &rv = rd_views_of(e);
					rd_make_views(g, e, rv, curr_depth, curr_depth, nullptr, nullptr, 0, tq_ptr(w, curr_depth + 1, COMP_Y), nullptr, nullptr);
					const uint32_t bit_cost = rd_get_intra_bits_qt(g, e, rv, parent, 1);
					cost += (uint32_t)(bit_cost * e.f->lambda + .5);
				}
				bool take_children;
				if (S.rd_mode != RDM_FAST) take_children = cost < pn.cost;
				else take_children = 1.25 * (cost + (double)(uint32_t)(45u * sum)) < (double)(uint32_t)(pn.cost + 45u * pn.sum);
				if (split_lost) { take_children = false; split_lost = false; }      // (decided above on the children evaluated: the others' fields are not theirs)
				if (take_children) {
					pn.cost = (uint32_t)cost;
					pn.distortion = (uint32_t)distortion;
					pn.sum = sum;
					if (curr_depth == max_tr_processing_depth) {
						const int tr_mask = 1 << (curr_depth - depth + (part_size_type == PART_NxN));
						uint32_t cbf_split = (c0.intra_cbf[0] & tr_mask) | (c1.intra_cbf[0] & tr_mask) | (c2.intra_cbf[0] & tr_mask) | (c3.intra_cbf[0] & tr_mask);
						cbf_split >>= 1;
						for (int k = 0; k < 4; k++) {
							node_of(e, pq.child[k]).intra_cbf[0] |= (int32_t)cbf_split;
							set_intra_info_buffs(g, e, depth, pq.child[k]);
						}
					} else {
						const int tr_mask = 1 << (curr_depth - depth + (part_size_type == PART_NxN));
						uint8_t *cb = w.cbf_buffs[COMP_Y][depth];
						uint32_t cbf_y = (cb[e.geo[pq.child[0]].abs_index] & tr_mask) | (cb[e.geo[pq.child[1]].abs_index] & tr_mask) |
								 (cb[e.geo[pq.child[2]].abs_index] & tr_mask) | (cb[e.geo[pq.child[3]].abs_index] & tr_mask);
						cbf_y >>= 1;
						g.sync();
						for (int i = g.tid; i < pq.num_part; i += g.n) cb[pq.abs_index + i] |= (uint8_t)cbf_y;
						g.sync();
					}
					sync_motion_buffers_luma(g, e, parent, curr_depth + 1, curr_depth, curr_depth + 1, curr_depth);
				} else {
					set_intra_info_buffs(g, e, depth, parent);
					sync_reference_buffs(g, e, parent, curr_depth, curr_depth + 1);
				}
				curr_depth--;
				parent = e.geo[parent].parent;
			}
			if (curr_depth + 2 <= max_tr_processing_depth) {
				const int aux = parent >= 0 ? e.geo[parent].child[(depth_state.get(curr_depth) + 3) & 3] : 0;
				for (int aux_depth = curr_depth + 2; aux_depth <= max_tr_processing_depth; aux_depth++) {
					sync_reference_buffs(g, e, aux, curr_depth + 1, aux_depth + 1);
					if (rd_full) {      // :1577
						const Geo &aq = e.geo[aux];
						bytes_copy(g, &w.intra_mode_buffs[COMP_Y][depth][aq.abs_index], &w.intra_mode_buffs[COMP_Y][aux_depth][aq.abs_index], aq.num_part);
						bytes_copy(g, &w.cbf_buffs[COMP_Y][depth][aq.abs_index], &w.cbf_buffs[COMP_Y][aux_depth][aq.abs_index], aq.num_part);
						bytes_copy(g, &w.tr_idx_buffs[depth][aq.abs_index], &w.tr_idx_buffs[aux_depth][aq.abs_index], aq.num_part);
					}
				}
			}
		}
	}
	Node &tn = node_of(e, top_ni);
	if (depth == max_tr_processing_depth) set_intra_info_buffs(g, e, depth, top_ni);
	if (part_size_type == PART_NxN && (part_position & 3) == 3) {
		const int par = e.geo[top_ni].parent;
		const int nsub = e.geo[e.geo[par].child[0]].num_part, abs_index = e.geo[par].abs_index;
		uint8_t *cb = w.cbf_buffs[COMP_Y][curr_depth];
		uint32_t split = (cb[abs_index] & 2) | (cb[abs_index + nsub] & 2) | (cb[abs_index + 2 * nsub] & 2) | (cb[abs_index + 3 * nsub] & 2);
		if (split) {
			split >>= 1;
			g.sync();
			for (int i = g.tid; i < 4 * nsub; i += g.n) cb[abs_index + i] |= (uint8_t)split;
			g.sync();
		}
	}
	if (S.rd_mode != RDM_FULL) {
		const double correction = calc_mv_correction(tn.qp, e.f->avg_dist);
		if (e.last_slog >= 0) e.ctu_g->slog[e.last_slog].tu_cost = tn.cost;
		return intra_luma_cost(tn.cost, bitcost_cu_mode, correction);
	}
	return tn.cost;
}

HENC_INLINE void chroma_dir_list(int *list, int luma_mode)
{
	list[0] = PLANAR_IDX; list[1] = VER_IDX; list[2] = HOR_IDX; list[3] = DC_IDX; list[4] = DM_CHROMA_IDX;
	// (the first match only - the four entries differ, so at most one matches; no break, so that the loop unrolls and the list stays in registers)
#pragma unroll
	for (int i = 0; i < 4; i++)
		if (luma_mode == list[i]) list[i] = 34;
}

// one chroma plane of the candidate search of encode_intra_chroma: SAD of the five candidates on the unfiltered neighbours of the auxiliary window
template <class G>
HENC_WALK_FN HENC_HD void chroma_search_comp(const G g, Enc &__restrict__ e, int curr, int c, const int *cand, uint32_t *sads)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[curr];
	const int n = q.size_chroma;
	pred_t *pred = pred_ptr(w, c) + q.yc * CTU_STRIDE_C + q.xc;
	const src_t *orig = curr_ptr(w, c) + q.yc * CTU_STRIDE_C + q.xc;
	// (the reference fills the neighbour array before every candidate, hmr_motion_intra_chroma.c:196; nothing between two candidates changes what it is filled from)
	node_fill_refs(g, e, curr, NWND - 1, c, 0);
	for (int mi = 0; mi < 5; mi++) sads[mi] = intra_predict_sad(g, pred, CTU_STRIDE_C, orig, CTU_STRIDE_C, e.adi_c, n, cand[mi], 0);
}
// one chroma TU of the winner: neighbours, prediction, residual, transform chain, reconstruction into the auxiliary window.  Returns the weighted SSD.
template <class G>
HENC_HD int chroma_tu_comp(const G g, Enc &__restrict__ e, int curr, int c, int cu_mode, int scan_mode, int shifts, int per, int rem, int *curr_sum_out)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Geo &q = e.geo[curr];
	const int n = q.size_chroma, curr_depth = q.depth, qwnd = NWND - 1, dwnd = NWND - 1;
	pred_t *pred = pred_ptr(w, c) + q.yc * CTU_STRIDE_C + q.xc;
	const src_t *orig = curr_ptr(w, c) + q.yc * CTU_STRIDE_C + q.xc;
	int16_t *quant = tq_ptr(w, qwnd, c) + ((q.abs_index << 4) >> 2), *iquant = iq_slot(w, c, (q.abs_index << 4) >> 2, e.on_helper);
	int16_t *dec = dec_ptr(w, dwnd, c) + q.yc * DEC_STRIDE_C + q.xc;
	node_fill_refs(g, e, curr, dwnd, c, 0);
	intra_predict(g, pred, CTU_STRIDE_C, e.adi_c, n, cu_mode, 0);
	tr_forward(g, HENC_FT(e), e.T, orig, CTU_STRIDE_C, pred, CTU_STRIDE_C, e.scratch_a, e.scratch_b, n, 0);
	const int curr_sum = quantize(g, HENC_FT(e), e.T, e.scratch_a, iquant, e.scratch_b, scan_mode, curr_depth, c, 1, e.f->slice_type == SLICE_I, e.seq->sign_hiding, n, per, rem);
	const int cbfv = ((curr_sum ? 1 : 0) << (shifts & 255)) | ((curr_sum ? 1 : 0) << (shifts >> 8));
	bytes_set(g, &w.cbf_chroma[c - 1][q.abs_index], cbfv, q.num_part);
	uint32_t raw;
	if (curr_sum) {
		lin_copy_nosync(g, iquant, quant, n * n);
		dequantize(g, HENC_FT(e), e.T, iquant, iquant, curr_depth, c, 1, n, per, rem);
		tr_inverse(g, HENC_FT(e), e.T, e.scratch_b, n, iquant, e.scratch_a, n, 0);
		raw = blk_reconst_ssd(g, pred, CTU_STRIDE_C, e.scratch_b, n, orig, CTU_STRIDE_C, dec, DEC_STRIDE_C, n);
	} else {
		lin_zero_nosync(g, quant, n * n);
		raw = blk_reconst_ssd(g, pred, CTU_STRIDE_C, (const int16_t *)nullptr, 0, orig, CTU_STRIDE_C, dec, DEC_STRIDE_C, n);
	}
	*curr_sum_out = curr_sum;
	return (int)(e.f->chroma_weight * raw);
}

// both chroma planes of a TU, one after the other
template <class G>
HENC_HD void chroma_tu_both(const G g, Enc &__restrict__ e, int curr, int cu_mode, int scan_mode, int shifts, int per, int rem, int *pc, int *cs)
{
	HENC_ENC_IN_LDS(e);
	pc[0] = chroma_tu_comp(g, e, curr, COMP_U, cu_mode, scan_mode, shifts, per, rem, &cs[0]);
	pc[1] = chroma_tu_comp(g, e, curr, COMP_V, cu_mode, scan_mode, shifts, per, rem, &cs[1]);
	g.sync();
}

// encode_intra_chroma, hmr_motion_intra_chroma.c:114-469 (non-HM path, rd_mode != RD_FULL)
template <class G>
HENC_WALK_FN HENC_HD uint32_t encode_intra_chroma(const G g, Enc &__restrict__ e, int depth, int part_position, int part_size_type)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	Work &w = *e.w;
	const int nxn = part_size_type == PART_NxN;
	int curr = node_at(e, depth, part_position), parent;
	const double weight = e.f->chroma_weight;
	const int qp_chroma = chroma_qp_table((int)node_of(e, curr).qp + S.chroma_qp_offset);
	const int per = qp_chroma / 6, rem = qp_chroma % 6;
	if (depth == 0 && CFG_MAX_CU_SIZE == 64) {
		parent = cfg_depth_start(0);
		curr = e.geo[parent].child[0];
	} else parent = e.geo[curr].parent;
	const int luma_mode = w.intra_mode_buffs[COMP_Y][depth][e.geo[curr].abs_index];   // written by the luma pass just before: never a token
#if !defined(__HIPCC__)
	if (luma_mode & MODE_TOKEN) { fprintf(stderr, "encode_intra_chroma: inherited luma mode\n"); abort(); }
#endif
	int mode_list[5];
	chroma_dir_list(mode_list, luma_mode);
	if (e.geo[curr].size_chroma == 2) {
		curr = parent;
		parent = e.geo[curr].parent;
	}
	// candidate search on the unfiltered neighbours of the last decoded window
	int best_modes[3] = {0, 0, 0};
	double best_costs[3] = {1.7e+308, 1.7e+308, 1.7e+308};
	uint32_t best_bits[3] = {0, 0, 0};
	{
		const Geo &q = e.geo[curr];
		const int n = q.size_chroma;
		// the SADs of the five candidates, U and V side by side when there is a helper wavefront (the reference alternates U and V per candidate;
		// the neighbour arrays do not change during the search, so the grouping does not matter)
		uint32_t sad_u[5], sad_v[5];
		int cand[5];
#pragma unroll
		for (int mi = 0; mi < 5; mi++) cand[mi] = mode_list[mi] == DM_CHROMA_IDX ? luma_mode : mode_list[mi];
		if (HENC_HELPERS(e)) {
			helper_post(g, e, 0, HJOB_CHROMA_SEARCH, curr, COMP_U, cand[0] | (cand[1] << 8) | (cand[2] << 16) | (cand[3] << 24), cand[4]);
			chroma_search_comp(g, e, curr, COMP_V, cand, sad_v);
			helper_wait(g, e, 0);
#pragma unroll
			for (int mi = 0; mi < 5; mi++) sad_u[mi] = e.box->r[0][mi];
		} else {
			chroma_search_comp(g, e, curr, COMP_U, cand, sad_u);
			chroma_search_comp(g, e, curr, COMP_V, cand, sad_v);
		}
#pragma unroll
		for (int mi = 0; mi < 5; mi++) {
			uint32_t distortion = sad_u[mi], cost = distortion;
			distortion += sad_v[mi];
			cost += distortion;
			uint32_t bit_cost = mode_list[mi] == DM_CHROMA_IDX ? 1 : 12;
			if (S.rd_mode == RDM_FULL) {      // hmr_motion_intra_chroma.c:228: the candidate goes into the depth's direction buffer and is priced by the counter
				const Geo &sq = e.geo[curr];
				bytes_set(g, &w.intra_mode_buffs[COMP_CHR][depth][sq.abs_index], mode_list[mi], sq.num_part);
				RdViews &rv = rd_views_of(e);
				rd_make_views(g, e, rv, depth, depth, nullptr, nullptr, depth, nullptr, nullptr, nullptr);
				bit_cost = rd_bits_chroma_mode(e, rv.v, curr);
			}
			cost += (uint32_t)(bit_cost * e.f->sqrt_lambda + .5);
			// homer_update_cand_list, hmr_motion_intra.c:893
			int um = mode_list[mi];
			double uc = cost;
			uint32_t ub = bit_cost;
#pragma unroll
			for (int i = 0; i < 3; i++)
				if (best_costs[i] > uc) {
					const int am = best_modes[i]; const double ac = best_costs[i]; const uint32_t ab = best_bits[i];
					best_costs[i] = uc; best_modes[i] = um; best_bits[i] = ub;
					uc = ac; um = am; ub = ab;
				}
		}
	}
	uint32_t best_cost = MAX_COST, best_sum = 0, sum = 0, distortion = 0, cost;
	int best_mode = 0;
	int top;
	{
		int cu_mode = best_modes[0];
		const uint32_t bit_cost = best_bits[0];
		DepthState depth_state;
		DepthState cbf_split0, cbf_split1;      // (0 / 1 per depth, one word per chroma component)
		DepthInts4 partition_cost;
		if (cu_mode == DM_CHROMA_IDX) cu_mode = luma_mode;
		if (depth == 0 && CFG_MAX_CU_SIZE == 64) {
			parent = cfg_depth_start(0);
			curr = e.geo[parent].child[0];
		} else {
			curr = node_at(e, depth, part_position);
			parent = e.geo[curr].parent;
		}
		int curr_depth = e.geo[curr].depth;
		depth_state.set(curr_depth, part_position & 3);
		const int qwnd = NWND - 1, dwnd = NWND - 1;
		bool broke = false;
		while (!(curr_depth == (depth - nxn) && depth_state.get(curr_depth) == (part_position & 3) + 1)) {
			curr = parent < 0 ? curr : e.geo[parent].child[depth_state.get(curr_depth)];
			const int tr_depth_luma = w.tr_idx_buffs[depth][e.geo[curr].abs_index] + depth - nxn;
			while (curr_depth < tr_depth_luma) {
				parent = curr;
				curr_depth++;
				curr = e.geo[parent].child[depth_state.get(curr_depth)];
			}
			if (e.geo[curr].depth >= 1) nodes_select_quad(g, e, e.geo[curr].abs_index >> 6);
			const int scan_mode = find_scan_mode(1, 0, e.geo[curr].size_chroma, cu_mode, 0);
			const int original_depth = e.geo[curr].depth;
			if (e.geo[curr].size_chroma == 2) {
				curr = parent;
				parent = e.geo[curr].parent;
			}
			const Geo &q = e.geo[curr];
			curr_depth = q.depth;
			const int n = q.size_chroma;
			partition_cost.set(depth_state.get(curr_depth), 0);
			{
				int cs[2], pc[2];
				const int shifts = (original_depth - depth + nxn) | ((curr_depth - depth + nxn) << 8);
				if (HENC_HELPERS(e)) {
					helper_post(g, e, 0, HJOB_CHROMA_TU, curr, COMP_U, cu_mode, scan_mode, shifts, per | (rem << 8));
					pc[1] = chroma_tu_comp(g, e, curr, COMP_V, cu_mode, scan_mode, shifts, per, rem, &cs[1]);
					helper_wait(g, e, 0);
					pc[0] = (int)e.box->r[0][0];
					cs[0] = (int)e.box->r[0][1];
				} else {
					pc[0] = chroma_tu_comp(g, e, curr, COMP_U, cu_mode, scan_mode, shifts, per, rem, &cs[0]);
					pc[1] = chroma_tu_comp(g, e, curr, COMP_V, cu_mode, scan_mode, shifts, per, rem, &cs[1]);
				}
				for (int k = 0; k < 2; k++) {
					sum += (uint32_t)cs[k];
					if (cs[k]) { if (k == 0) cbf_split0.set(curr_depth, 1); else cbf_split1.set(curr_depth, 1); }
					partition_cost.add(depth_state.get(curr_depth), pc[k]);
				}
			}
			node_of(e, curr).sum += sum;
			distortion += (uint32_t)partition_cost.get(depth_state.get(curr_depth));
			if (distortion > best_cost) {
				distortion = best_cost + 1;
				broke = true;
				break;
			}
			depth_state.inc(curr_depth);
			if (depth_state.get(curr_depth) == 4) {
				while (depth_state.get(curr_depth) == 4 && curr_depth > (depth - nxn)) {
					const Geo &pq = e.geo[parent];
					const int sh = curr_depth - 1 - depth + nxn;
					g.sync();
					for (int i = g.tid; i < pq.num_part; i += g.n) {
						w.cbf_chroma[0][pq.abs_index + i] |= (uint8_t)(cbf_split0.get(curr_depth) << sh);
						w.cbf_chroma[1][pq.abs_index + i] |= (uint8_t)(cbf_split1.get(curr_depth) << sh);
					}
					g.sync();
					cbf_split0.set(curr_depth - 1, cbf_split0.get(curr_depth - 1) | cbf_split0.get(curr_depth));
					cbf_split1.set(curr_depth - 1, cbf_split1.get(curr_depth - 1) | cbf_split1.get(curr_depth));
					cbf_split0.set(curr_depth, 0);
					cbf_split1.set(curr_depth, 0);
					depth_state.set(curr_depth, 0);
					curr_depth--;
					depth_state.inc(curr_depth);
					if (curr_depth != 0) parent = e.geo[parent].parent;
				}
			}
		}
		(void)broke;
		if (depth == 0) top = cfg_depth_start(0);
		else {
			top = node_at(e, depth, part_position);
			if (nxn) top = e.geo[top].parent;
		}
		cost = distortion;
		if (S.rd_mode == RDM_FULL && cost < best_cost) {      // :417: the chroma syntax of the CU with the winner of the search
			const Geo &tq = e.geo[top];
			bytes_set(g, &w.intra_mode_buffs[COMP_CHR][depth][tq.abs_index], best_modes[0], tq.num_part);
			RdViews &rv = rd_views_of(e);
			rd_make_views(g, e, rv, depth, depth, w.cbf_chroma[0], w.cbf_chroma[1], depth, nullptr, tq_ptr(w, qwnd, COMP_U), tq_ptr(w, qwnd, COMP_V));
			const uint32_t bits = rd_get_intra_bits_qt(g, e, rv, top, 0);
			cost += (uint32_t)(bits * e.f->lambda + .5);
		} else if (S.rd_mode != RDM_FULL && cost < best_cost) {
			const double correction = calc_mv_correction(node_of(e, top).qp, e.f->avg_dist);
			cost += (uint32_t)(bit_cost * correction + .5);
		}
		if (cost < best_cost) {
			best_cost = cost;
			best_sum = sum;
			best_mode = best_modes[0];
			sync_motion_buffers_chroma(g, e, top, qwnd, depth + 1, dwnd, depth + 1);
			const Geo &tq = e.geo[top];
			bytes_copy(g, &w.cbf_chroma[0][tq.abs_index], &w.cbf_buffs[COMP_U][depth][tq.abs_index], tq.num_part);
			bytes_copy(g, &w.cbf_chroma[1][tq.abs_index], &w.cbf_buffs[COMP_V][depth][tq.abs_index], tq.num_part);
		}
	}
	const Geo &tq = e.geo[top];
	bytes_set(g, &w.intra_mode_buffs[COMP_CHR][depth][tq.abs_index], best_mode, tq.num_part);
	node_of(e, top).sum += best_sum;
	return best_cost;
}

template <class G>
HENC_HD uint32_t encode_intra(const G g, Enc &__restrict__ e, int curr_depth, int position, int part_size_type);

}  // namespace henc
