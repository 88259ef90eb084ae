// The CTU decision drivers: the depth-first walk over the coding quadtree for P slices (motion_inter_full,
// hmr_motion_inter.c:3746-4263) and I slices (motion_intra_cu, hmr_motion_intra.c:1759-1990), the merge evaluation
// (check_rd_cost_merge_2nx2n :3493-3742), the buffer consolidation between depths (:3298-3487), and the CTU set-up /
// tear-down the WPP thread does around them (wfpp_encoder_thread hmr_encoder_lib.c:2897-2945, init_ctu :2254,
// create_partition_ctu_neighbours hmr_motion_intra.c:658, hmr_mem_transfer.c:284-419).
#pragma once
#include "enc_intra.h"
#include "enc_inter.h"
#include "enc_quad.h"

namespace henc {

// ---- info-buffer shuffles ----------------------------------------------------------------------------------------------
// CONSOLIDATE_ENC_INFO_BUFFS :3298 (dir = 0: CTU arrays <- worker buffers of `depth`) and its inverse (get_back :3461-3466,
// consolidate_info_buffers_for_rd hmr_motion_intra.c:1632)
template <class G>
HENC_HD void info_buffs_copy(const G g, Enc &__restrict__ e, int depth, int abs_idx, int num, int to_ctu)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	Work &w = *e.w;
	CtuPublic &c = *e.ctu;
	for (int i = g.tid; i < num; i += g.n) {
		const int k = abs_idx + i;
		if (to_ctu) {
			c.cbf[0][k] = w.cbf_buffs[0][depth][k];
			c.cbf[1][k] = w.cbf_buffs[1][depth][k];
			c.cbf[2][k] = w.cbf_buffs[2][depth][k];
			c.tr_idx[k] = w.tr_idx_buffs[depth][k];
			c.intra_mode[0][k] = w.intra_mode_buffs[0][depth][k];
			c.intra_mode[1][k] = w.intra_mode_buffs[1][depth][k];
		} else {
			// (the record is in HBM behind a pointer the compiler cannot tell from the worker's buffers: read everything first - one trip to memory, not six)
			const uint8_t v0 = c.cbf[0][k], v1 = c.cbf[1][k], v2 = c.cbf[2][k], v3 = c.tr_idx[k], v4 = c.intra_mode[0][k], v5 = c.intra_mode[1][k];
			w.cbf_buffs[0][depth][k] = v0;
			w.cbf_buffs[1][depth][k] = v1;
			w.cbf_buffs[2][depth][k] = v2;
			w.tr_idx_buffs[depth][k] = v3;
			w.intra_mode_buffs[0][depth][k] = v4;
			w.intra_mode_buffs[1][depth][k] = v5;
		}
	}
	g.sync();
	PRIM_END(PP_INFO);
}

// SET_INTER_INFO_BUFFS :3309
template <class G>
HENC_HD void set_inter_info_buffs(const G g, Enc &__restrict__ e, int ni)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	const int a = q.abs_index, n = q.num_part;
	if (nd.prediction_mode == PM_INTER) {
		for (int i = g.tid; i < n; i += g.n) {
			c.inter_mode[a + i] = (uint8_t)nd.inter_mode;
			c.skipped[a + i] = (uint8_t)nd.skipped;
			c.merge[a + i] = (uint8_t)nd.merge_flag;
			c.merge_idx[a + i] = (uint8_t)nd.merge_idx;
			c.mv_ref_idx[a + i] = (int8_t)nd.inter_ref_index;
			if (nd.inter_mode & 1) c.mv_ref[a + i] = nd.inter_mv;
		}
		if (nd.inter_mode & 1) {
			c.mv_diff_ref_idx[a] = (uint8_t)nd.best_candidate_idx;
			c.mv_diff[a] = nd.best_dif_mv;
		}
	} else {
		for (int i = g.tid; i < n; i += g.n) {
			c.mv_ref_idx[a + i] = -1;
			c.skipped[a + i] = 0;
			c.merge[a + i] = 0;
		}
	}
	for (int i = g.tid; i < n; i += g.n) c.pred_mode[a + i] = (uint8_t)nd.prediction_mode;
	g.sync();
	PRIM_END(PP_INFO);
}

// get_back_consolidated_info :3456 / put_consolidated_info :3472
template <class G>
HENC_HD void get_back_consolidated_info(const G g, Enc &__restrict__ e, int ni, int depth)
{	ni = uni(ni); depth = uni(depth);

	HENC_ENC_IN_LDS(e);
	info_buffs_copy(g, e, depth, e.geo[ni].abs_index, e.geo[ni].num_part, 0);
	sync_motion_buffers(g, e, ni, 0, depth + 1, 0, depth + 1);
}
template <class G>
HENC_HD void put_consolidated_info(const G g, Enc &__restrict__ e, int ni, int depth)
{	ni = uni(ni); depth = uni(depth);

	HENC_ENC_IN_LDS(e);
	info_buffs_copy(g, e, depth, e.geo[ni].abs_index, e.geo[ni].num_part, 1);
	sync_motion_buffers(g, e, ni, depth + 1, 0, depth + 1, 0);
}

// consolidate_prediction_info :3372.  Returns true when the children were taken (the caller's running cost of the parent's depth then changes by
// children_cost - parent_cost, as the reference does inside).
template <class G>
HENC_HD bool consolidate_prediction_info(const G g, Enc &__restrict__ e, int pi, uint32_t parent_cost, uint32_t children_cost, int is_max_depth)
{	pi = uni(pi); parent_cost = uni(parent_cost); children_cost = uni(children_cost); is_max_depth = uni(is_max_depth);

	HENC_ENC_IN_LDS(e);
	const Geo &pq = e.geo[pi];
	Node &pn = node_of(e, pi);
	CtuPublic &c = *e.ctu;
	const int abs_index = pq.abs_index, num = pq.num_part, curr_depth = pq.depth + 1;
	uint32_t children_sum = 0;
	if (pq.child[0] >= 0)
		children_sum = node_of(e, pq.child[0]).sum + node_of(e, pq.child[1]).sum + node_of(e, pq.child[2]).sum + node_of(e, pq.child[3]).sum;
	if (children_cost < parent_cost || !(pn.b_inside && pn.r_inside)) {
		const int part2 = curr_depth < CFG_MAX_PRED_DEPTH ? PART_2Nx2N : PART_NxN;
		pn.cost = children_cost;
		pn.distortion = node_of(e, pq.child[0]).distortion + node_of(e, pq.child[1]).distortion + node_of(e, pq.child[2]).distortion + node_of(e, pq.child[3]).distortion;
		pn.sum = children_sum;
		if (is_max_depth) {
			sync_motion_buffers(g, e, pi, curr_depth + 1, 0, curr_depth + 1, 0);
			info_buffs_copy(g, e, curr_depth, abs_index, num, 1);
			for (int k = 0; k < 4; k++) {
				const int ci = pq.child[k];
				set_inter_info_buffs(g, e, ci);
				const Geo &cq = e.geo[ci];
				const uint8_t qp = (uint8_t)node_of(e, ci).qp;
				for (int i = g.tid; i < cq.num_part; i += g.n) c.qp[cq.abs_index + i] = qp;
			}
			for (int i = g.tid; i < num; i += g.n) {
				c.pred_depth[abs_index + i] = (uint8_t)(curr_depth - (part2 == PART_NxN));
				c.part_size_type[abs_index + i] = (uint8_t)part2;
			}
			g.sync();
		}
		return true;
	} else {
		const int part2 = pq.depth < CFG_MAX_PRED_DEPTH ? PART_2Nx2N : PART_NxN;
		const int parent_depth = pq.depth;
		sync_motion_buffers(g, e, pi, parent_depth + 1, 0, parent_depth + 1, 0);
		info_buffs_copy(g, e, parent_depth, abs_index, num, 1);
		set_inter_info_buffs(g, e, pi);
		const uint8_t qp = (uint8_t)pn.qp;
		for (int i = g.tid; i < num; i += g.n) {
			c.qp[abs_index + i] = qp;
			c.pred_depth[abs_index + i] = (uint8_t)(pq.depth - (part2 == PART_NxN));
			c.part_size_type[abs_index + i] = (uint8_t)part2;
		}
		g.sync();
	}
	return false;
}

// the reference-sample refresh after a CU (sub)tree is final: bottom row / right column of the consolidated reconstruction
// into the deeper windows (hmr_motion_inter.c:3985-4001 and :4222-4230, hmr_motion_intra.c:1899-1916, 1956-1974)
template <class G>
HENC_HD void refresh_deeper_windows(const G g, Enc &__restrict__ e, int aux_ni, int from_depth, int with_info)
{	aux_ni = uni(aux_ni); from_depth = uni(from_depth); with_info = uni(with_info);

	HENC_ENC_IN_LDS(e);
	const int max_processing_depth = hmin(CFG_MAX_PRED_DEPTH + e.seq->max_intra_tr_depth - 1, NDEPTH - 1);
	if (from_depth > max_processing_depth) return;
	const Geo &q = e.geo[aux_ni];
	// (the reference copies window by window, each followed by the depth's side-info copy; source and destinations are distinct buffers, so the order is free)
	sync_reference_buffs_range(g, e, aux_ni, 0, from_depth + 1, max_processing_depth + 1);
	if (with_info && e.seq->rd_mode != RDM_DIST_ONLY)
		for (int aux_depth = from_depth; aux_depth <= max_processing_depth; aux_depth++) info_buffs_copy(g, e, aux_depth, q.abs_index, q.num_part, 0);
	sync_reference_buffs_chroma(g, e, aux_ni, 0, NWND - 1);
}

// encode_intra, hmr_motion_intra.c:1731
template <class G>
HENC_WALK_FN HENC_HD uint32_t encode_intra(const G g, Enc &__restrict__ e, int curr_depth, int position, int part_size_type)
{
	HENC_ENC_IN_LDS(e);
	uint32_t cost = 0;
	if (part_size_type == PART_2Nx2N) {
		uint32_t cl, cc;
		{ HENC_PROF_T0(); cl = encode_intra_luma(g, e, curr_depth, position, part_size_type); HENC_PROF_ADD(e, PF_INTRA_TU); } HENC_TRACE_PW(e, "iluma");
		{ HENC_PROF_T0(); cc = encode_intra_chroma(g, e, curr_depth, position, part_size_type); HENC_PROF_ADD(e, PF_INTRA_CHROMA); } HENC_TRACE_PW(e, "ichroma");
		cost = cl + cc;
	} else {
		for (int n = 0; n < 4; n++) {
			node_of(e, node_at(e, curr_depth, position) + n).qp = (uint32_t)e.ctu_qp;
			cost += encode_intra_luma(g, e, curr_depth, position + n, part_size_type); HENC_TRACE_PW(e, "iluma");
		}
		cost += encode_intra_chroma(g, e, curr_depth, position, part_size_type); HENC_TRACE_PW(e, "ichroma");
	}
	return cost;
}

// check_rd_cost_merge_2nx2n :3493 (P slice)
template <class G>
HENC_WALK_FN HENC_HD uint32_t check_rd_cost_merge(const G g, Enc &__restrict__ e, int depth, int position)
{
	HENC_ENC_IN_LDS(e);
	Work &w = *e.w;
	const Seq &S = *e.seq;
	const int ni = node_at(e, depth, position);
	const Geo &q = e.geo[ni];
	Node &nd = node_of(e, ni);
	CtuPublic &c = *e.ctu;
	const int abs_index = q.abs_index, curr_depth = q.depth, n = q.size, nc = q.size_chroma;
	const int gx = e.ctu_x + q.x, gy = e.ctu_y + q.y;
	int merge_cand_buffer[5] = {0, 0, 0, 0, 0};
	int best_is_skip = 0, best_candidate = 0, have_ctu_cbf = 0, prev_nores_ran = 0;      // prev_nores_ran: has the candidate before this one been through a no-residual evaluation?
	uint32_t dist, best_dist = MAX_COST, cost, best_cost = MAX_COST, best_sum = 0, ctu_cbf = 0;      // ctu_cbf: cbf[0] | cbf[1] | cbf[2] of the CTU record at this CU's first unit
	MV best_mv = {0, 0};
	int best_ref_idx = 0;
	uint8_t inter_modes[5] = {255, 255, 255, 255, 255};
	const double weight = e.f->chroma_weight;
	HENC_QPROF_T0();
	{ PRIM_T0(); get_merge_candidates(e, ni, w.merge_cands, inter_modes); PRIM_END(PP_CAND); }
	if (q.size == 8) HENC_QPROF_MARK(e, 1);
#if defined(HENC_QUAD)
	{	// an 8 x 8 CU: the evaluations of all candidates in one pass, then this loop on their results (enc_quad.h); -1: the sequential way
		const QuadCands mc = quad_load_cands(e);
		const int slots = q.size == 8 ? quad_prepare<8>(g, e, ni, mc HENC_QPROF_PASS) : (q.size == 16 ? quad_prepare<16>(g, e, ni, mc HENC_QPROF_PASS) : -1);
		if (slots >= 0) {
			const uint32_t best = q.size == 8 ? quad_merge_loop<8>(g, e, ni, slots, mc, inter_modes HENC_QPROF_PASS) : quad_merge_loop<16>(g, e, ni, slots, mc, inter_modes HENC_QPROF_PASS);
			HENC_QPROF_MARK(e, 7);      // (the node's fields)
			if (g.tid == 0 && e.prof) e.prof[8] += 1;      // (CUs taken this way)
			return best;
		}
	}
#endif
	for (int cand = 0; cand < CFG_NUM_MERGE_CAND; cand++) {
		int mc_done = 0;
		// A candidate that repeats the one before it (same vector, same reference - the usual case under coherent motion, and always for the zero candidates that
		// fill the list: with one reference picture they are all alike) cannot win: its two evaluations recompute what its predecessor's did, bit for bit, into
		// the same buffers, and "<" keeps the earlier one.  The reference runs them anyway; what they leave behind is what the predecessor left, except that
		// the no-residual evaluation, where it runs (the candidate's coded evaluation - the predecessor's - had levels, or is itself skipped because the best so
		// far is a skip), resets the node's cbf / transform index / level sum.  Do exactly that and nothing else.  (best_is_skip only ever goes from 0 to 1, so a
		// coded evaluation that would run here has run for the predecessor.)
		// One case is NOT a repeat: the predecessor's coded evaluation came out without levels (so its own no-residual evaluation was left out) and became the best
		// (so this candidate's coded evaluation is left out and its flag stays 0): the no-residual evaluation of this prediction then runs here for the first time, and
		// it can win - a transform block whose levels were zeroed by the coded evaluation's rate test is charged the distortion of the levels it dropped
		// (encode_inter, hmr_motion_inter.c:207-219), not that of the prediction.  Found by tools/encoder_fuzz.py (392x136, clip 931814, QP 22).
		if (cand >= 1 && w.merge_cands.mv[cand].x == w.merge_cands.mv[cand - 1].x && w.merge_cands.mv[cand].y == w.merge_cands.mv[cand - 1].y &&
		    w.merge_cands.ref_idx[cand] == w.merge_cands.ref_idx[cand - 1]) {
			const int coded_runs = !best_is_skip;                                       // (then it rewrites the predecessor's coded result over itself)
			if (coded_runs) merge_cand_buffer[cand] = merge_cand_buffer[cand - 1];
			const int nores_runs = !(coded_runs && merge_cand_buffer[cand] == 1);
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
		int coded_ran = 0;
		prev_nores_ran = 0;
		for (int no_res = 0; no_res < 2; no_res++) {
			if (no_res == 1 && merge_cand_buffer[cand] == 1) continue;
			if (best_is_skip && no_res == 0) continue;
			if (no_res == 1) prev_nores_ran = 1;
			if (!mc_done) {
				const MV mv = w.merge_cands.mv[cand];
				const int xlow = -S.margin_y, xhigh = S.width + S.margin_y, ylow = -S.margin_y, yhigh = S.height + S.margin_y;
				const int spx = gx + mv.x / 4, spy = gy + mv.y / 4;
				if (!(spx < xlow || spx + n > xhigh || spy < ylow || spy + n > yhigh)) motion_compensate_cu(g, e, ni, mv);
				else e.n_stale_pred++;      // Q12: the candidate is evaluated on what the prediction window holds (see include/homer_gpu.h, hmr_gpu_enc_stale_predictions)
				mc_done = 1;
			}
			if (no_res == 0) {
				cost = dist = encode_inter(g, e, curr_depth, position, PART_2Nx2N); HENC_TRACE_PW(e, "einter");
				cost = (uint32_t)((double)cost + cost_rd(e.f->avg_dist, nd.sum));
				coded_ran = e.inter_ssq_valid;
			} else if (coded_ran) {
				// SSD(source, prediction) of the three blocks is the squared residual the coded evaluation of this same prediction has just summed up
				dist = e.inter_ssq[0];
				dist += (uint32_t)(weight * e.inter_ssq[1]);
				dist += (uint32_t)(weight * e.inter_ssq[2]);
				nd.inter_cbf[0] = nd.inter_cbf[1] = nd.inter_cbf[2] = 0;
				nd.inter_tr_idx = 0;
				nd.sum = 0;
				cost = dist;
			} else {
				if (HENC_HELPERS(e)) {
					if (NHELP >= 2) {
						helper_post(g, e, 0, HJOB_SSD, ni, COMP_U);
						helper_post(g, e, NHELP - 1, HJOB_SSD, ni, COMP_V);
					} else helper_post(g, e, 0, HJOB_SSD, ni, COMP_UV);
				}
				dist = blk_ssd(g, w.curr_y + q.y * 64 + q.x, 64, w.pred_y + q.y * 64 + q.x, 64, n);
				if (HENC_HELPERS(e)) {
					for (int h = 0; h < NHELP; h++) helper_wait(g, e, h);
					dist += (uint32_t)(weight * e.box->r[0][0]);
					dist += (uint32_t)(weight * (NHELP >= 2 ? e.box->r[NHELP - 1][0] : e.box->r[0][1]));
				} else {
					dist += (uint32_t)(weight * blk_ssd(g, w.curr_c[0] + q.yc * 32 + q.xc, 32, w.pred_c[0] + q.yc * 32 + q.xc, 32, nc));
					dist += (uint32_t)(weight * blk_ssd(g, w.curr_c[1] + q.yc * 32 + q.xc, 32, w.pred_c[1] + q.yc * 32 + q.xc, 32, nc));
				}
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
				if (no_res == 1) {
					// skipped: the prediction is the reconstruction, the levels are zero
					blk_copy(g, w.pred_y + q.y * 64 + q.x, 64, dec_ptr(w, curr_depth + 1, COMP_Y) + q.y * DEC_STRIDE_Y + q.x, DEC_STRIDE_Y, n, n);
					for (int k = 0; k < 2; k++)
						blk_copy(g, w.pred_c[k] + q.yc * 32 + q.xc, 32, dec_ptr(w, curr_depth + 1, COMP_U + k) + q.yc * DEC_STRIDE_C + q.xc, DEC_STRIDE_C, nc, nc);
					lin_zero(g, tq_ptr(w, curr_depth + 1, COMP_Y) + (abs_index << 4), n * n);
					lin_zero(g, tq_ptr(w, curr_depth + 1, COMP_U) + ((abs_index << 4) >> 2), nc * nc);
					lin_zero(g, tq_ptr(w, curr_depth + 1, COMP_V) + ((abs_index << 4) >> 2), nc * nc);
					set_enc_info_buffs(g, e, ni, curr_depth);
				}
				put_consolidated_info(g, e, ni, curr_depth);
				// (the CTU record's cbf of this CU is what put_consolidated_info has just copied there from the worker's buffers of this depth: read from
				// those - the record lives in HBM, and a store followed by a load of it is a round trip to L2)
				ctu_cbf = (uint32_t)w.cbf_buffs[0][curr_depth][abs_index] | w.cbf_buffs[1][curr_depth][abs_index] | w.cbf_buffs[2][curr_depth][abs_index];
				have_ctu_cbf = 1;
				best_is_skip = (ctu_cbf & 1) == 0;
			}
			if (no_res == 0) {
				if (!have_ctu_cbf) { ctu_cbf = (uint32_t)c.cbf[0][abs_index] | c.cbf[1][abs_index] | c.cbf[2][abs_index]; have_ctu_cbf = 1; }
				if (((ctu_cbf >> curr_depth) & 1) == 0) merge_cand_buffer[cand] = 1;
			}
		}
	}
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

// the comparison of the P-slice walk that uses the running intra ratio (:4018-4021).  The ratio is the one input of a CTU that depends on
// every CTU before it in raster order; enc_sched.h re-evaluates the logged comparisons with the true value through this same function.
HENC_INLINE double intra_ratio(uint32_t total_intra_partitions, uint32_t total_partitions)
{
	const uint32_t tp = total_partitions == 0 ? 1 : total_partitions;
	return hclip((double)total_intra_partitions / (double)tp, .0, .15);
}
HENC_INLINE double intra_cost_with_ratio(double intra_dist, double ratio, double add, double rd)
{
	double intra_cost = intra_dist * (1.275 - ratio) + add;
	intra_cost += rd;
	return intra_cost;
}

// which walk a CTU takes (wfpp_encoder_thread, hmr_encoder_lib.c:2916): intra for I slices and for the CTUs after a scene cut
HENC_INLINE int ctu_takes_intra_walk(const FrameCtx &f, int ctu_num)
{
	if (f.slice_type == SLICE_I) return 1;
	if (f.scene_cut_ctu < 0) return 0;
	if (!f.lockstep) return ctu_num > f.scene_cut_ctu;
	// synchronous wavefront: the detecting CTU (thread 0, row 0) decides first in its step; everything else from that step on is intra
	return ctu_num != f.scene_cut_ctu && ctu_num % f.wctu + 2 * (ctu_num / f.wctu) >= f.scene_cut_ctu % f.wctu + 2 * (f.scene_cut_ctu / f.wctu);
}
// the detection itself, evaluated when CTU `ctu_num` enters the inter walk with the counters of the CTUs before it (:3791-3793)
HENC_INLINE int scene_cut_fires(const Seq &S, const FrameCtx &f, uint32_t intra_before, uint32_t parts_before)
{
	return f.scene_cut_allowed && parts_before > (uint32_t)(S.nctu * NPART / 10) && (double)intra_before > (double)parts_before * .7;
}

// motion_inter_full :3746
template <class G>
HENC_HD uint32_t motion_inter_ctu(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	CtuPublic &c = *e.ctu;
	const double avg_distortion = e.f->avg_dist;
	const int perf_min_depth = S.perf_min_depth, perf_fast_skip = S.perf_fast_skip;
	DepthState depth_state;
	DepthCosts cost_sum;
	int curr_depth = 0, parent = -1, curr = 0;
	double dist = 0, best_cost;
	const int root = 0;
	while (curr_depth != 0 || depth_state.get(curr_depth) != 1) {
		double cost = 0, intra_cost = 0;
		int stop_recursion = 0, is_skipped = 0;
		const Geo &q = e.geo[curr];
		if constexpr (G::bg) bg_quiesce(g, e);      // (a background intra search of the CU before that nobody took: it reads the partition nodes, which the next line may replace)
		if (q.depth >= 1) nodes_select_quad(g, e, q.abs_index >> 6);
		Node &nd = node_of(e, curr);
		curr_depth = q.depth;
		const int part_size_type = curr_depth < CFG_MAX_PRED_DEPTH ? PART_2Nx2N : PART_NxN;
		const int num_part_in_cu = q.num_part;
		const int position = q.list_index - cfg_depth_start(curr_depth);
		nd.qp = (uint32_t)e.ctu_qp;   // hmr_rc_get_cu_qp (hmr_rate_control.c:337): the slice QP, or under rate control the CTU's (qp_depth 0: a deeper CU takes its parent's)
		if (nd.b_inside && nd.r_inside) {
			int mv_cost = 0;
			if (part_size_type == PART_2Nx2N) {
				uint32_t sad = 0, merge_dist = MAX_COST, merge_cost = MAX_COST, merge_sum = 0;
				MV merge_mv = {0, 0};
				int merge_ref_idx = 0, merge_inter_mode = 0;
				const int action = S.me_precision * 2 - 1;
				nd.prediction_mode = PM_INTER;
				if (curr_depth >= perf_min_depth) {
					{ HENC_PROF_T0(); merge_dist = check_rd_cost_merge(g, e, curr_depth, position); HENC_PROF_ADD(e, PF_MERGE); }
					merge_mv = nd.inter_mv;
					merge_ref_idx = nd.inter_ref_index;
					merge_inter_mode = nd.inter_mode;
					merge_sum = nd.sum;
					merge_cost = merge_dist;
					merge_cost += (uint32_t)nd.merge_idx;
					merge_cost = (uint32_t)(merge_cost * 1.1 + depth_term(avg_distortion, curr_depth));
					merge_cost = (uint32_t)((double)merge_cost + cost_rd(e.f->avg_dist, nd.sum));
					cost = merge_cost;
					dist = merge_dist;
					if (nd.skipped) merge_cost = (uint32_t)(merge_cost * .95);
					if (nd.skipped && (double)merge_cost < avg_distortion * NPART / 2.5) is_skipped = 1;
				} else {
					nd.inter_mv.x = nd.inter_mv.y = 0;
					nd.merge_flag = 0;
					nd.skipped = 0;
					cost = nd.cost = merge_cost = MAX_COST;
					dist = nd.distortion = merge_dist = MAX_COST;
					is_skipped = 0;
				}
				// the intra evaluation below starts with a mode search that depends on nothing the motion search and the inter transform tree change: the helper
				// starts on it now (enc_common.h bg_post), for the CU sizes whose intra evaluation does not depend on the motion search's SAD
				if constexpr (G::bg)
					if (NHELP == 1 && perf_fast_skip && curr_depth >= perf_min_depth && !is_skipped && q.size < 32 && S.rd_mode != RDM_FULL && e.f->lockstep) bg_post(g, e, curr, curr_depth);
				if (curr_depth >= perf_min_depth) {
					if (!is_skipped) sad = (uint32_t)cu_motion_estimation(g, e, curr_depth, position, PART_2Nx2N, action);   // timed inside (PF_ME_INT / PF_ME_SUB)
					if (!is_skipped && (q.size < 64 || sad < 100u * num_part_in_cu)) {
						{ HENC_PROF_T0(); mv_cost = predict_inter(g, e, curr_depth, position, PART_2Nx2N); HENC_PROF_ADD(e, PF_PRED_INTER); } HENC_TRACE_PW(e, "pred");
						{ HENC_PROF_T0(); dist = (double)(int)encode_inter(g, e, curr_depth, position, PART_2Nx2N); HENC_PROF_ADD(e, PF_ENC_INTER); }
					} else {
						mv_cost = 0;
						dist = MAX_COST;
						nd.sum = 0;
						nd.inter_mv.x = nd.inter_mv.y = 0;
					}
					cost = dist;
					cost += 2 * mv_cost;
					cost = cost * 1.1 + depth_term(avg_distortion, curr_depth);
					cost += cost_rd(e.f->avg_dist, nd.sum);
					if (cost < merge_cost) {
						nd.merge_flag = 0;
						nd.skipped = 0;
					} else {
						nd.inter_mv = merge_mv;
						nd.inter_ref_index = merge_ref_idx;
						nd.inter_mode = merge_inter_mode;
						nd.sum = merge_sum;
						cost = merge_cost;
						dist = merge_dist;
					}
				}
				nd.cost = (uint32_t)cost;
				nd.distortion = (uint32_t)dist;
				HENC_TRACE("CU ctu=%d d=%d abs=%d inter: merge=%d idx=%d skip=%d mv=(%d,%d) cost=%u dist=%u sum=%u\n", c.ctu_number, curr_depth, q.abs_index,
					   nd.merge_flag, nd.merge_idx, nd.skipped, nd.inter_mv.x, nd.inter_mv.y, nd.cost, nd.distortion, nd.sum);
				if (perf_fast_skip && (dist == 0 || (nd.sum < (uint32_t)num_part_in_cu && dist < .25 * avg_distortion * num_part_in_cu) ||
						       (nd.sum == 0 && dist < avg_distortion * num_part_in_cu))) {
					if (nd.merge_flag) get_back_consolidated_info(g, e, curr, curr_depth);
					stop_recursion = 1;
					(void)consolidate_prediction_info(g, e, curr, nd.cost, MAX_COST, 0);
					refresh_deeper_windows(g, e, curr, curr_depth, 1);
				}
				if (perf_fast_skip && (curr_depth >= perf_min_depth && !stop_recursion && !is_skipped && (q.size < 32 || sad > 400u * num_part_in_cu))) {
					const uint32_t inter_sum = nd.sum;
					if (!nd.merge_flag) put_consolidated_info(g, e, curr, curr_depth);
					e.last_slog = -1;
					// encode_intra (:1731): luma, then chroma.  The comparison below only grows with what chroma adds - its distortion, and its levels through cost_rd - so when
					// luma alone already loses against the inter cost for every value the intra ratio can take (it is clipped to 0 .. 0.15, and enc_sched.h re-evaluates
					// the comparison with the true one), chroma cannot change the outcome: the evaluation is left out.  What it would have left behind - the auxiliary
					// chroma window, the depth's chroma buffers, the node's level sum - is restored from the consolidated inter result by the losing branch below or
					// rewritten before it is read.  Not with RD_FULL (its bit estimates have side effects of their own) and not while a CTU is logged for replay by
					// a schedule that needs the true distortion (lockstep = 0 keeps the full evaluation).
					uint32_t intra_dist;
					{
						uint32_t cl;
						{ HENC_PROF_T0(); cl = encode_intra_luma(g, e, curr_depth, position, PART_2Nx2N); HENC_PROF_ADD(e, PF_INTRA_TU); } HENC_TRACE_PW(e, "iluma");
						const double add_lb = hclip(avg_distortion - 400, 40., avg_distortion) / 1.75 * curr_depth;
						const double lb = intra_cost_with_ratio((double)cl, .15, add_lb, cost_rd(e.f->avg_dist, nd.sum));
						if (S.rd_mode != RDM_FULL && e.f->lockstep && lb > cost + 1.) intra_dist = cl;
						else {
							uint32_t cc;
							{ HENC_PROF_T0(); cc = encode_intra_chroma(g, e, curr_depth, position, PART_2Nx2N); HENC_PROF_ADD(e, PF_INTRA_CHROMA); } HENC_TRACE_PW(e, "ichroma");
							intra_dist = cl + cc;
						}
					}
#if defined(__HIPCC__) && defined(HENC_PROFILE)
					if (g.tid == 0 && e.timeline && e.timeline[2] == 0) e.timeline[2] = wall_clock64();
#endif
					const double ratio = intra_ratio(e.total_intra_partitions, e.total_partitions);
					const double add = hclip(avg_distortion - 400, 40., avg_distortion) / 1.75 * curr_depth;
					const double rd = cost_rd(e.f->avg_dist, nd.sum);
					intra_cost = intra_cost_with_ratio((double)intra_dist, ratio, add, rd);
					const int take_intra = intra_cost < cost;
					if (e.n_ratio_cmp < MAX_RATIO_CMP) {
						double *lg = e.ctu_g->ratio_cmp + 4 * e.n_ratio_cmp;
						lg[0] = (double)intra_dist; lg[1] = add; lg[2] = rd; lg[3] = cost;
						e.ctu_g->ratio_out[e.n_ratio_cmp] = (uint8_t)take_intra;
						e.ctu_g->ratio_slog[e.n_ratio_cmp] = (int16_t)e.last_slog;
						if (e.last_slog >= 0) e.ctu_g->slog[e.last_slog].has_cmp = 1;
					}
					e.n_ratio_cmp++;
					HENC_TRACE("CU ctu=%d d=%d abs=%d intra: dist=%u cost=%.3f vs %.3f\n", c.ctu_number, curr_depth, q.abs_index, intra_dist, intra_cost, cost);
					if (take_intra) {
						nd.cost = (uint32_t)intra_cost;
						nd.distortion = intra_dist;
						nd.prediction_mode = PM_INTRA;
						nd.merge_flag = 0;
						nd.skipped = 0;
					} else {
						get_back_consolidated_info(g, e, curr, curr_depth);
						nd.cost = (uint32_t)cost;
						nd.distortion = (uint32_t)dist;
						nd.sum = inter_sum;
						nd.prediction_mode = PM_INTER;
					}
				} else if (curr_depth >= perf_min_depth && !stop_recursion) {
					if (nd.merge_flag) get_back_consolidated_info(g, e, curr, curr_depth);
				}
			} else {
				// NxN level.  Inter NxN needs a parent larger than 8x8 (:4061), which the 8x8 minimum CU of the built configurations excludes.
				cost = dist = nd.cost = nd.distortion = MAX_COST;
				depth_state.set(curr_depth, 3);
			}
		} else {
			nd.cost = MAX_COST;
		}
		cost_sum.add(curr_depth, nd.cost);
		depth_state.inc(curr_depth);
		if (curr_depth < CFG_MAX_PRED_DEPTH && nd.tl_inside && !stop_recursion) {
			curr_depth++;
			parent = curr;
		} else if (depth_state.get(curr_depth) == 4) {
			while (depth_state.get(curr_depth) == 4 && curr_depth > 0) {
				const int is_max_depth = curr_depth == CFG_MAX_PRED_DEPTH;
				const Geo &pq = e.geo[parent];
				const uint32_t ccost = node_of(e, pq.child[0]).cost + node_of(e, pq.child[1]).cost + node_of(e, pq.child[2]).cost + node_of(e, pq.child[3]).cost;
				cost = ccost;
				depth_state.set(curr_depth, 0);
				best_cost = node_of(e, parent).cost;
				if (consolidate_prediction_info(g, e, parent, (uint32_t)best_cost, (uint32_t)cost, is_max_depth)) cost_sum.add(e.geo[parent].depth, (uint32_t)cost - (uint32_t)best_cost);
				cost_sum.set(curr_depth, 0);
				curr_depth--;
				parent = e.geo[parent].parent;
				if (curr_depth > perf_min_depth && cost_sum.get(curr_depth) > node_of(e, parent).cost && depth_state.get(curr_depth) < 4 && node_of(e, root).b_inside &&
				    node_of(e, root).r_inside)
					depth_state.set(curr_depth, 4);
			}
			const int aux = parent >= 0 ? e.geo[parent].child[(depth_state.get(curr_depth) + 3) & 3] : root;
			refresh_deeper_windows(g, e, aux, curr_depth, 0);
		}
		if (parent >= 0) curr = e.geo[parent].child[depth_state.get(curr_depth)];
	}
	if constexpr (G::bg) bg_quiesce(g, e);
	return node_of(e, root).cost;
}

// analyse_recursive_info_cu + calc_variance_cu, hmr_motion_intra.c:1645-1727 (performance_mode 3, called on the CTU's root: :1788): the variance of every partition
// (luma + 1.25 x both chroma components, per sample) and, bottom up, "recursive_split": a partition whose variance exceeds one of its children's by the reference's
// measure - or one of whose children is itself split, or which reaches outside the picture - is not evaluated as a whole, one that is not split ends the recursion.
// The reference walks the tree depth first; what it computes depends on a partition and its children only, so the variances are taken partition by partition and the
// flags level by level, the partitions of a level side by side.  Variances sit in the TU scratch (free before the walk), the flags as a bit per node in Work::rsplit.
HENC_INLINE bool rsplit_of(const Work &w, int ni) { return (w.rsplit[ni >> 5] >> (ni & 31)) & 1u; }
template <class G>
HENC_HD void analyse_recursive_info(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	Work &w = *e.w;
	uint32_t *var = (uint32_t *)(int16_t *)w.pred_aux;       // [NNODES]
	uint8_t *split = (uint8_t *)(int16_t *)w.delta_u;         // [NNODES]
	static_assert(NNODES * 4 <= TU_SCRATCH * 2 && NNODES <= TU_SCRATCH * 2, "the analysis works in the TU scratch");
	for (int ni = 0; ni < NNODES; ni++) {
		const Geo &q = e.geo[ni];
		const bool inside = (e.ctu_y + q.y + q.size <= S.height) && (e.ctu_x + q.x + q.size <= S.width);
		uint32_t v = 0;
		if (inside) {
			const int nc = q.size_chroma;
			const uint32_t vy = blk_modified_variance(g, w.curr_y + q.y * CTU_STRIDE_Y + q.x, CTU_STRIDE_Y, q.size, 1) / (uint32_t)(q.size * q.size);
			uint32_t vc = (uint32_t)(1.25 * blk_modified_variance(g, w.curr_c[0] + q.yc * CTU_STRIDE_C + q.xc, CTU_STRIDE_C, nc, 2) / (nc * nc));
			vc += (uint32_t)(1.25 * blk_modified_variance(g, w.curr_c[1] + q.yc * CTU_STRIDE_C + q.xc, CTU_STRIDE_C, nc, 2) / (nc * nc));
			v = vy + vc;
		}
		if (g.tid == 0) { var[ni] = v; split[ni] = inside ? 0 : 1; }
	}
	g.sync();
	for (int depth = CFG_MAX_PRED_DEPTH - 1; depth >= 0; depth--) {
		const int first = cfg_depth_start(depth), count = 1 << (2 * depth);
		for (int k = g.tid; k < count; k += g.n) {
			const int pi = first + k;
			const Geo &pq = e.geo.lane(pi);
			const uint32_t parent_variance = (uint32_t)(.5 + hsqrt((double)var[pi]));
			for (int l = 0; l < 4; l++) {
				const int ci = pq.child[l];
				const uint32_t child_variance = (uint32_t)(.5 + ((double)(depth + 1) / 4.) * hsqrt((double)var[ci]) + 3 * (depth + 1));
				if (parent_variance > child_variance || split[ci]) { split[pi] = 1; break; }
			}
		}
		g.sync();
	}
	for (int base = 0; base < 11 * 32; base += g.n) {
		const int ni = base + g.tid;
		if (G::n >= 64) {
			const uint64_t m = g.ballot(ni < NNODES && split[ni]);
			if (g.tid == 0) { w.rsplit[base >> 5] = (uint32_t)m; if ((base >> 5) + 1 < 11) w.rsplit[(base >> 5) + 1] = (uint32_t)(m >> 32); }
		} else {
			if ((ni & 31) == 0) w.rsplit[ni >> 5] = 0;
			if (ni < NNODES && split[ni]) w.rsplit[ni >> 5] |= 1u << (ni & 31);
		}
	}
	g.sync();
}

// motion_intra_cu, hmr_motion_intra.c:1759
template <class G>
HENC_HD uint32_t motion_intra_ctu(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	CtuPublic &c = *e.ctu;
	DepthState depth_state;
	DepthCosts cost_sum;
	int curr_depth = 0, parent = 0, curr = 0;
	const int initial_depth = 0, initial_position = 0;
	depth_state.set(0, initial_position);
	e.w->thread_seen_intra = 1;   // (every lane stores the same value) hmr_motion_intra.c:1783: from now on this thread's shadow CTU reads "intra"
	if (S.rd_mode == RDM_FULL) {      // motion_intra :1993: the shadow CTU starts as a copy of the CTU's descriptor (its side-info pointers at the CTU's arrays), all INTRA
		bytes_set(g, e.wrd->rd_pred_mode, PM_INTRA, NPART);
		e.rd_luma_depth = -1;
		e.rd_chroma_state = 0;
	}
	const bool fastest = S.perf_mode > 2;      // PERF_FASTEST_COMPUTATION: the variance pre-analysis steers the recursion (:1788, :1824, :1891)
	if (fastest) analyse_recursive_info(g, e);
	while (curr_depth != initial_depth || depth_state.get(curr_depth) != initial_position + 1) {
		const Geo &q = e.geo[curr];
		if (q.depth >= 1) nodes_select_quad(g, e, q.abs_index >> 6);
		Node *nd = &node_of(e, curr);
		curr_depth = q.depth;
		const int part_size_type = curr_depth < CFG_MAX_PRED_DEPTH ? PART_2Nx2N : PART_NxN;
		int position = q.list_index - cfg_depth_start(curr_depth);
		double cost_luma = 0, cost_chroma = 0;
		int stop_recursion = 0;
		nd->qp = (uint32_t)e.ctu_qp;
		if (nd->b_inside && nd->r_inside) {
			if (fastest && rsplit_of(*e.w, curr)) {
				nd->cost = MAX_COST;      // (:1826-1828: not evaluated as a whole; nothing goes into the depth's running cost)
			} else if (part_size_type == PART_2Nx2N) {
				{ HENC_PROF_T0(); cost_luma = encode_intra_luma(g, e, curr_depth, position, part_size_type); HENC_PROF_ADD(e, PF_INTRA_TU); } HENC_TRACE_PW(e, "iluma");
				{ HENC_PROF_T0(); cost_chroma = encode_intra_chroma(g, e, curr_depth, position, part_size_type); HENC_PROF_ADD(e, PF_INTRA_CHROMA); } HENC_TRACE_PW(e, "ichroma");
				nd->cost = (uint32_t)(cost_luma + cost_chroma);
				cost_sum.add(curr_depth, nd->cost);
				nd->prediction_mode = PM_INTRA;
				HENC_TRACE("ICU ctu=%d d=%d abs=%d cost=%u (l %.0f c %.0f) mode=%d\n", c.ctu_number, curr_depth, q.abs_index, nd->cost, cost_luma, cost_chroma, nd->intra_mode[0]);
			} else {
				cost_luma = 0;
				for (int n = 0; n < 4; n++) {
					Node &sn = node_of(e, curr + n);
					sn.qp = (uint32_t)e.ctu_qp;
					sn.cost = encode_intra_luma(g, e, curr_depth, position + n, part_size_type); HENC_TRACE_PW(e, "iluma");
					cost_luma += sn.cost;
					cost_sum.add(curr_depth, sn.cost);
					sn.prediction_mode = PM_INTRA;
				}
				if (cost_luma < node_of(e, parent).cost && (nd->b_inside && nd->r_inside)) {
					position = e.geo[e.geo[parent].child[0]].list_index - cfg_depth_start(curr_depth);
					cost_chroma = encode_intra_chroma(g, e, curr_depth, position, part_size_type); HENC_TRACE_PW(e, "ichroma");
					nd->cost += (uint32_t)cost_chroma;
					cost_sum.add(curr_depth, (uint32_t)cost_chroma);
				}
				HENC_TRACE("ICU ctu=%d d=%d abs=%d NxN luma=%.0f chroma=%.0f\n", c.ctu_number, curr_depth, q.abs_index, cost_luma, cost_chroma);
				depth_state.set(curr_depth, 3);
			}
		}
		depth_state.inc(curr_depth);
		// :1891: a partition the analysis does not split (or that cost nothing) ends the recursion here
		if (fastest && (!rsplit_of(*e.w, curr) || nd->cost == 0) && part_size_type != PART_NxN && nd->b_inside && nd->r_inside) {
			(void)consolidate_prediction_info(g, e, curr, nd->cost, MAX_COST, 0);
			stop_recursion = 1;
			const int aux = parent >= 0 ? e.geo[parent].child[(depth_state.get(curr_depth) + 3) & 3] : 0;
			refresh_deeper_windows(g, e, aux, curr_depth, 1);
		}
		if (!stop_recursion && curr_depth < CFG_MAX_PRED_DEPTH && nd->tl_inside) {
			curr_depth++;
			parent = curr;
		} else if (depth_state.get(curr_depth) == 4) {
			while (depth_state.get(curr_depth) == 4 && curr_depth > initial_depth) {
				const Geo &pq = e.geo[parent];
				const uint32_t ccost = node_of(e, pq.child[0]).cost + node_of(e, pq.child[1]).cost + node_of(e, pq.child[2]).cost + node_of(e, pq.child[3]).cost;
				const double cost = ccost;
				depth_state.set(curr_depth, 0);
				const double best_cost = node_of(e, parent).cost;
				if (consolidate_prediction_info(g, e, parent, (uint32_t)best_cost, (uint32_t)cost, curr_depth == CFG_MAX_PRED_DEPTH)) cost_sum.add(e.geo[parent].depth, (uint32_t)cost - (uint32_t)best_cost);
				cost_sum.set(curr_depth, 0);
				curr_depth--;
				parent = e.geo[parent].parent;
				if (S.perf_mode <= 2 && curr_depth > 0 && curr_depth < CFG_MAX_PRED_DEPTH && depth_state.get(curr_depth) < 4 && node_of(e, 0).b_inside && node_of(e, 0).r_inside) {
					double totalcost = 0;
					for (int h = 0; h < depth_state.get(curr_depth); h++) totalcost += node_of(e, e.geo[parent].child[h]).cost;
					if (totalcost > node_of(e, parent).cost) depth_state.set(curr_depth, 4);
				}
			}
			const int aux = parent >= 0 ? e.geo[parent].child[(depth_state.get(curr_depth) + 3) & 3] : 0;
			refresh_deeper_windows(g, e, aux, curr_depth, 1);
		}
		if (parent >= 0) curr = e.geo[parent].child[depth_state.get(curr_depth)];
	}
	for (int i = g.tid; i < NPART; i += g.n) {
		c.mv_ref_idx[i] = -1;
		c.pred_mode[i] = PM_INTRA;
		c.skipped[i] = 0;
	}
	g.sync();
	return node_of(e, 0).cost;
}

// ---- CTU set-up and tear-down ---------------------------------------------------------------------------------------------
// create_partition_ctu_neighbours, hmr_motion_intra.c:658 + cu_partition_get_neighbours :629
// The reference walks the tree depth-first; a node only reads its parent, so the levels are done one after the other with the nodes of a level side by side.
// A node is reached when its parent's top-left corner lies inside the picture (then so do the corners of the parent's ancestors).
template <class G>
HENC_HD void create_partition_neighbours(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *e.seq;
	CtuPublic &c = *e.ctu;
	const int cu_min_tu_size_shift = hmax(CFG_MAX_CU_SHIFT - (CFG_MAX_PRED_DEPTH + hmax(S.max_intra_tr_depth, S.max_inter_tr_depth) - 1), 2);
	const int max_processing_depth = CFG_MAX_CU_SHIFT - cu_min_tu_size_shift;
	const int valid_lines = (c.y + 64) > S.height ? S.height - c.y : 64, valid_cols = (c.x + 64) > S.width ? S.width - c.x : 64;
	const int cx = c.x, cy = c.y, has_left = c.has_left, has_top = c.has_top, has_top_right = c.has_top_right;
	for (int depth = 0; depth <= max_processing_depth && depth < NDEPTH; depth++) {
		const int first = cfg_depth_start(depth), count = 1 << (2 * depth);
		for (int k = g.tid; k < count; k += g.n) {
			const int curr = first + k;
			const Geo &q = e.geo.lane(curr);
			if (depth > 0) {
				const Geo &pq = e.geo.lane(q.parent);
				if (!((cy + pq.y < S.height) && (cx + pq.x < S.width))) continue;
			}
			Node &nd = depth < 3 ? e.nodes[curr] : e.ctu_g->nodes[curr];      // (depths 3 and 4: the record in HBM - no quadrant is resident yet)
			nd.tl_inside = (cy + q.y < S.height) && (cx + q.x < S.width);
			nd.b_inside = (cy + q.y + q.size <= S.height);
			nd.r_inside = (cx + q.x + q.size <= S.width);
			if (nd.tl_inside) {
				if (depth == 0) {
					nd.left_nb = has_left;
					nd.top_nb = has_top;
					nd.left_bottom_nb = 0;
					nd.top_right_nb = has_top_right;
				} else {
					const Geo &pq = e.geo.lane(q.parent);
					const Node &pn = depth < 4 ? e.nodes[q.parent] : e.ctu_g->nodes[q.parent];      // (the parent of a depth-4 node was written to the record by the pass before)
					nd.left_nb = (pn.left_nb || q.x) ? 1 : 0;
					nd.top_nb = (pn.top_nb || q.y) ? 1 : 0;
					nd.left_bottom_nb = ((pn.left_bottom_nb && q.x == pq.x) || (pn.left_nb && q.x == pq.x && q.y == pq.y && valid_lines > q.y + q.size)) ? 1 : 0;
					nd.top_right_nb = ((pn.top_right_nb && q.y == pq.y) || (pn.top_nb && q.x == pq.x && q.y == pq.y && valid_cols > q.x + q.size) ||
							   (q.x == pq.x && q.y != pq.y && valid_cols > q.x + q.size))
								  ? 1
								  : 0;
				}
			}
		}
		g.sync();
	}
}

// init_ctu :2254 + CuGetNeighbors :2160, mem_transfer_move_curr_ctu_group / mem_transfer_intra_refs (hmr_mem_transfer.c:284,351)
template <class G>
HENC_HD void ctu_begin(const G g, Enc &__restrict__ e, int ctu_num)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Seq &S = *e.seq;
	Work &w = *e.w;
	e.ctu_g = e.ctus + ctu_num;
	// the side-info record and the partition nodes are read and written all through the walk: a worker with fast memory of its own works on copies
	if (e.ctu_fast) {
		lin_copy_words(g, (const uint32_t *)(const CtuPublic *)e.ctu_g, (uint32_t *)e.ctu_fast, (int)(sizeof(CtuPublic) / 4));
		e.ctu = e.ctu_fast;
	} else e.ctu = e.ctu_g;
#if !defined(__HIPCC__)
	e.nodes_fast = (Node *)(((uintptr_t)w.nodes_fast_store + 15) & ~(uintptr_t)15);
	e.wrd = w.rd_store;
#endif
	lin_copy_words(g, (const uint32_t *)e.ctu_g->nodes, (uint32_t *)e.nodes_fast, (int)(sizeof(Node) * NODES_RESIDENT / 4));
	e.nodes = e.nodes_fast;
	e.node_quad = -1;
	CtuPublic &c = *e.ctu;
	e.scratch_a = w.pred_aux;
	e.scratch_b = w.delta_u;
#if !defined(__HIPCC__)
	e.mc_tmp_c = w.sub_tmp;
	e.mc_tmp_y = w.sub_tmp;
	e.mc_tmp_y_stride = 72;
#endif
	e.adi_c = w.adi;
	const int cx = ctu_num % S.wctu, cy = ctu_num / S.wctu;
	c.ctu_number = ctu_num;
	c.x = cx * 64;
	c.y = cy * 64;
	e.ctu_x = cx * 64;
	e.ctu_y = cy * 64;
	const int ctu_w = (c.x + 64) < S.width ? 64 : S.width - c.x, ctu_h = (c.y + 64) < S.height ? 64 : S.height - c.y;
	if (ctu_w != 64 || ctu_h != 64) c.last_valid_partition = raster2abs(((ctu_h >> 2) - 1) * 16 + (ctu_w >> 2) - 1);
	else c.last_valid_partition = NPART - 1;
	c.has_left = cx > 0;
	c.has_top = cy > 0;
	c.has_top_left = cx > 0 && cy > 0;
	c.has_top_right = cy > 0 && cx != S.wctu - 1;
	e.nb_ctus = (uint32_t)((cx > 0) | ((cy > 0) << 1) | ((cy > 0 && cx != S.wctu - 1) << 2) | ((cx > 0 && cy > 0) << 3));
	e.n_spec_reads = e.n_ratio_cmp = 0;
	e.n_stale_pred = 0;
	e.amvp_node = -1;
	// the worker's mode buffers start as "inherited" everywhere (see read_mode_buff, enc_intra.h)
	for (int i = g.tid; i < 2 * NDEPTH * NPART; i += g.n) (&w.intra_mode_buffs[0][0][0])[i] = (uint8_t)(MODE_TOKEN | ((i / NPART) % NDEPTH));
	// source CTU
	for (int comp = 0; comp < 3; comp++) {
		const int sz = comp ? 32 : 64, px = comp ? c.x >> 1 : c.x, py = comp ? c.y >> 1 : c.y;
		const int pw = comp ? S.width >> 1 : S.width, ph = comp ? S.height >> 1 : S.height;
		const int ss = comp ? S.src_stride_c : S.src_stride_y;
		const int ww = (px + sz) < pw ? sz : pw - px, hh = (py + sz) < ph ? sz : ph - py;
		blk_copy_to_src(g, e.f->src[comp] + py * ss + px, ss, curr_ptr(w, comp), sz, hh, ww);
	}
	// neighbour samples of the picture under reconstruction into every decoded window
	if (c.has_left || c.has_top) {
		for (int comp = 0; comp < 3; comp++) {
			const int sz = comp ? 32 : 64, px = comp ? c.x >> 1 : c.x, py = comp ? c.y >> 1 : c.y;
			const int pw = comp ? S.width >> 1 : S.width, ph = comp ? S.height >> 1 : S.height;
			const int rs = comp ? S.stride_c : S.stride_y, ds = dec_stride(comp);
			int left_copy = 0, top_copy = 0;
			if (c.has_left) left_copy += sz;
			if (c.has_top) top_copy += sz;
			if (c.has_top_right) top_copy += sz;
			if (left_copy > ph - py) left_copy = ph - py;
			if (top_copy > pw - px) top_copy = pw - px;
			const int16_t *src = e.f->rec[comp] + py * rs + px;
			for (int l = 0; l < NWND; l++) {
				int16_t *dst = dec_ptr(w, l, comp);
				for (int i = g.tid; i < 1 + top_copy + left_copy; i += g.n) {
					if (i == 0) {
						if (c.has_left && c.has_top) dst[-ds - 1] = src[-rs - 1];
					} else if (i <= top_copy) dst[-ds + (i - 1)] = src[-rs + (i - 1)];
					else {
						const int j = i - 1 - top_copy;
						dst[j * ds - 1] = src[j * rs - 1];
					}
				}
			}
		}
		g.sync();
	}
	create_partition_neighbours(g, e);
	if (HENC_HELPERS(e)) {     // the helpers work on this CTU from now on
		g.sync();      // (the helpers copy this context - the worker's, at its fixed place in LDS - when they take the job)
		for (int h = 0; h < NHELP; h++) helper_post(g, e, h, HJOB_NEW_CTU);
		for (int h = 0; h < NHELP; h++) helper_wait(g, e, h);
	}
	PRIM_END(PP_CTU_IO);
}

// mem_transfer_decoded_blocks :312 + the coefficient copy (hmr_encoder_lib.c:2942-2945) + the thread counters (:2924-2940)
template <class G>
HENC_HD void ctu_end(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Seq &S = *e.seq;
	Work &w = *e.w;
	CtuPublic &c = *e.ctu;
	for (int comp = 0; comp < 3; comp++) {
		const int sz = comp ? 32 : 64, px = comp ? c.x >> 1 : c.x, py = comp ? c.y >> 1 : c.y;
		const int pw = comp ? S.width >> 1 : S.width, ph = comp ? S.height >> 1 : S.height;
		const int rs = comp ? S.stride_c : S.stride_y;
		const int ww = (px + sz) < pw ? sz : pw - px, hh = (py + sz) < ph ? sz : ph - py;
		blk_copy(g, dec_ptr(w, 0, comp), dec_stride(comp), e.f->rec[comp] + py * rs + px, rs, hh, ww);
	}
	lin_copy(g, tq_ptr(w, 0, COMP_Y), e.coeff, 4096);
	lin_copy(g, tq_ptr(w, 0, COMP_U), e.coeff + 4096, 1024);
	lin_copy(g, tq_ptr(w, 0, COMP_V), e.coeff + 5120, 1024);
	uint32_t cnt = 0;
	for (int i = g.tid; i < NPART; i += g.n) cnt += c.pred_mode[i] == PM_INTRA;
	cnt = g.sum(cnt);
	c.intra_parts = ctu_takes_intra_walk(*e.f, c.ctu_number) ? NPART : cnt;
	c.distortion = node_of(e, 0).distortion;
	e.ctu_g->n_spec_reads = e.n_spec_reads;
	e.ctu_g->n_stale_pred = e.n_stale_pred;
	e.ctu_g->n_ratio_cmp = e.n_ratio_cmp;
	g.sync();
	if (e.ctu_fast) lin_copy_words(g, (const uint32_t *)e.ctu_fast, (uint32_t *)(CtuPublic *)e.ctu_g, (int)(sizeof(CtuPublic) / 4));
	nodes_write_back(g, e);
	PRIM_END(PP_CTU_IO);
}

// tokens -> values, for the worker buffers and the CTU's mode arrays, once the values behind the tokens (Work::mode_in) are the true ones
template <class G>
HENC_HD void resolve_mode_tokens(const G g, Work &w, CtuPublic &c)
{
	for (int i = g.tid; i < 2 * NDEPTH * NPART; i += g.n) {
		uint8_t &v = (&w.intra_mode_buffs[0][0][0])[i];
		if (v & MODE_TOKEN) v = w.mode_in[i / (NDEPTH * NPART)][v & 7][i % NPART];
	}
	for (int i = g.tid; i < 2 * NPART; i += g.n) {
		uint8_t &v = (&c.intra_mode[0][0])[i];
		if (v & MODE_TOKEN) v = w.mode_in[i / NPART][v & 7][i % NPART];
	}
	g.sync();
}

template <class G>
#if defined(__HIPCC__)
__attribute__((noinline))   // one compiled body for every kernel that encodes CTUs
#endif
HENC_HD void encode_ctu(const G g, Enc &__restrict__ e, int ctu_num)
{
	HENC_ENC_IN_LDS(e);
	{ HENC_PROF_T0(); ctu_begin(g, e, ctu_num); HENC_PROF_ADD(e, PF_SETUP); }
	const int intra_walk = ctu_takes_intra_walk(*e.f, ctu_num);
	e.ctu_g->walk_intra = intra_walk;
	if (!intra_walk) motion_inter_ctu(g, e);
	else motion_intra_ctu(g, e);
	ctu_end(g, e);
}

}  // namespace henc
