// Host-side control of the frame encoder: configuration clamps, the static partition geometry, frame typing and the
// per-frame scalars the device code takes as inputs (doubles that come out of pow / sqrt stay on the host).
// Restates HOMER_enc_control(HOMER_SETCFG) hmr_encoder_lib.c:704-1650 (the fields the hot path reads), init_partition_info
// hmr_motion_intra.c:758, put_frame_to_encode :262 (frame typing), hmr_rd_init hmr_tables.c:315 and the per-frame
// statistics of encoder_engine_thread :3217-3238.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <vector>
#include "enc_types.h"
#include "enc_rc.h"

namespace henc {

// mirrors HVENC_Cfg (homer_hevc_enc_api.h:138-167), same field names
struct HostCfg {
	int32_t size, profile, width, height;
	float frame_rate;
	int32_t cu_size, max_pred_partition_depth, max_intra_tr_depth, max_inter_tr_depth, intra_period, gop_size, num_b, num_ref_frames;
	int32_t motion_estimation_precision, qp, chroma_qp_offset, num_enc_engines, wfpp_enable, wfpp_num_threads, sign_hiding, sample_adaptive_offset;
	int32_t bitrate_mode, bitrate, vbv_size, vbv_init, reinit_gop_on_scene_change, rd_mode, performance_mode;
};

inline int host_raster2abs(int r)
{
	const int x = r & 15, y = r >> 4;
	int a = 0;
	for (int b = 0; b < 4; b++) a |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
	return a;
}
inline int host_abs2raster(int a)
{
	int x = 0, y = 0;
	for (int b = 0; b < 4; b++) {
		x |= ((a >> (2 * b)) & 1) << b;
		y |= ((a >> (2 * b + 1)) & 1) << b;
	}
	return y * 16 + x;
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// returns false for configurations outside the rows built so far
inline bool make_seq(HostCfg cfg, Seq &s, const char **why)
{
	memset(&s, 0, sizeof s);
	*why = "";
	cfg.num_b = clampi(cfg.num_b, 0, 1);
	cfg.gop_size = clampi(cfg.gop_size, 1, cfg.num_b + 1);
	cfg.intra_period = clampi(cfg.intra_period, cfg.gop_size + 1, ((cfg.intra_period - 1) / cfg.gop_size) * cfg.gop_size + 1);
	cfg.num_ref_frames = (cfg.gop_size == cfg.num_b) ? 1 : clampi(cfg.num_ref_frames, 0, 16);
	if (cfg.cu_size != 64) { *why = "cu_size != 64"; return false; }
	if (cfg.num_b != 0 || cfg.gop_size != 1) { *why = "B frames"; return false; }
	if (cfg.num_ref_frames != 1) { *why = "num_ref_frames != 1"; return false; }
	if (cfg.bitrate_mode < 0 || cfg.bitrate_mode > 2) { *why = "bitrate_mode"; return false; }
	if (cfg.bitrate_mode != 0 && (cfg.bitrate <= 0 || cfg.vbv_size <= 0 || cfg.frame_rate <= 0)) { *why = "rate control needs bitrate, vbv_size and frame_rate"; return false; }
	// RD_FULL: the bit estimates copy the real coder's contexts as the schedule leaves them (enc_rdo.h, enc_rc.h RdCtxSim: the synchronous wavefront, or one thread in
	// raster order; every engine has coder objects of its own) - fixed QP
	if (cfg.rd_mode == RDM_FULL && cfg.bitrate_mode != 0) {
		*why = "rd_mode RD_FULL needs fixed QP";
		return false;
	}
	s.max_cu_size = 64;
	s.max_cu_size_shift = 6;
	s.max_pred_depth = cfg.max_pred_partition_depth > 4 ? 4 : cfg.max_pred_partition_depth;
	if (s.max_pred_depth != 4) { *why = "max_pred_partition_depth != 4"; return false; }
	if (cfg.width % (64 >> (s.max_pred_depth - 1)) || cfg.height % (64 >> (s.max_pred_depth - 1))) { *why = "size not a multiple of the minimum CU"; return false; }
	// Two corners of the reference's CTU-lagged filter pipeline (hmr_encoder_lib.c:2386-2843) that the frame-level passes here do not reproduce
	// (found by sweeping picture grids against the compiled reference, tools/stream_diff.py): pictures one CTU wide, and - with SAO on - pictures of at
	// most five CTU columns with at least as many rows as columns (and at least four): the in-row lag conditions (`ctu_num_index >= 3 / 4 / 5`)
	// then hardly ever hold, the stages run in the row-end flushes instead, and SAO statistics see offsets already applied above them.  Observed:
	// 3x4, 3x5, 4x4, 5x5, 5x6 differ; 3x3, 4x3, 5x4, 6x5, 6x6, 6x17, 7x4 and everything wider are identical.  Refused rather than approximated.
	// WPP threads: 1 = the reference's deterministic single-thread order; N > 1 = its synchronous-wavefront schedule (enc_sched.h), which is one of the
	// reference's own interleavings only if thread k is free again when its next row (k + N) reaches its first step: 2 N >= CTU columns.  The reference has room
	// for 32 threads (hmr_private.h:1234), so a 2160p picture (34 rows, 60 columns) runs with 32 and rows 32 / 33 continue threads 0 / 1.
	{
		const int wc = (cfg.width + 63) / 64, hc = (cfg.height + 63) / 64, n = cfg.wfpp_num_threads;
		if (n > 32) { *why = "wfpp_num_threads above the reference's 32"; return false; }
		if (n > hc) { *why = "more WPP threads than CTU rows"; return false; }
		if (n > 1 && n < hc && 2 * n < wc) { *why = "wfpp_num_threads between 1 and the number of CTU rows needs 2 x threads >= CTU columns"; return false; }
	}
	{
		const int wc = (cfg.width + 63) / 64, hc = (cfg.height + 63) / 64;
		if (hc > POST_MAX_ROWS) { *why = "more than 128 CTU rows"; return false; }
		if (cfg.wfpp_num_threads > 1 && wc + 2 * (hc - 1) > HENC_MAX_STEPS) { *why = "more than 192 wavefront steps (CTU columns + 2 x (CTU rows - 1))"; return false; }
	}
	if ((cfg.width + 63) / 64 < 2) { *why = "picture narrower than two CTUs"; return false; }
	// (two CTU columns and more than one CTU row: the compiled reference crashes - 120x88, 128x128, 128x192, any configuration; 128x64 runs - so no stream exists to pin one on)
	if ((cfg.width + 63) / 64 < 3 && (cfg.height + 63) / 64 > 1) { *why = "picture of two CTU columns and more than one CTU row (the reference crashes there)"; return false; }
	// Several engines on a narrow picture: the reference hands a reference row on to the next engine from the in-row stage of its lagged filter pipeline only
	// from column index 5 + 2 (the search window) on (hmr_deblock_sao_pad_sync_ctu, hmr_encoder_lib.c:2437); on pictures of fewer than nine CTU columns the
	// semaphore counts then do not add up once there are more than four CTU rows and its engines wait for each other for ever (observed with ref_lockstep: 3 ... 8
	// columns x 5, 6 or 9 rows, SAO on or off, any thread count; with two columns it crashes).  There is nothing to pin a stream on there: refused.
	{
		const int wc = (cfg.width + 63) / 64, hc = (cfg.height + 63) / 64;
		if (cfg.num_enc_engines > 1 && (wc < 3 || (wc < 9 && hc > 4))) { *why = "num_enc_engines > 1 on a picture of fewer than nine CTU columns and more than four CTU rows (the reference's engines deadlock there)"; return false; }
		// RD_FULL with several engines: every engine has coder objects of its own, and a per-engine replay of them (RdCtxSim) reproduces the compiled reference on the
		// fixtures tried and on most random configurations - but not all (round 6, tools/encoder_fuzz.py --combos: 4 of about 75 differ, on 6 / 8 / 13 / 14 CTU columns,
		// checker build and device alike).  Not pinned, so refused.
		if (cfg.num_enc_engines > 1 && cfg.rd_mode == RDM_FULL) { *why = "rd_mode RD_FULL with num_enc_engines > 1"; return false; }
	}
	{
		const int wc = (cfg.width + 63) / 64, hc = (cfg.height + 63) / 64;
		if (cfg.sample_adaptive_offset && wc <= 5 && hc >= (wc > 4 ? wc : 4)) { *why = "SAO on a picture of at most five CTU columns that has at least as many CTU rows"; return false; }
		// (three columns: the same corner without SAO too - 3x4 ... 3x9 differ, 3x3 and every 4 ... 7 column grid up to 9 rows are identical; tools/encoder_fuzz.py)
		if (wc == 3 && hc >= 4) { *why = "picture of three CTU columns and four or more CTU rows"; return false; }
	}
	s.max_intra_tr_depth = cfg.max_intra_tr_depth > 5 ? 5 : cfg.max_intra_tr_depth;
	s.max_inter_tr_depth = cfg.max_inter_tr_depth > 5 ? 5 : cfg.max_inter_tr_depth;
	s.min_tu_size_shift = 2;
	s.max_tu_size_shift = 5;
	s.mincu_mintr_shift_diff = (6 - s.max_pred_depth) - 2;
	s.max_cu_depth = s.max_pred_depth + s.mincu_mintr_shift_diff;
	s.mincu_mintr_shift_diff++;
	s.perf_mode = clampi(cfg.performance_mode, 0, 3);
	s.perf_fast_skip = s.perf_mode >= 1;
	s.perf_min_depth = s.perf_mode == 2 ? 1 : (s.perf_mode == 3 ? 2 : 0);
	s.rd_mode = clampi(cfg.rd_mode, 0, 2);
	s.me_precision = cfg.motion_estimation_precision == 0 ? ME_PEL : (cfg.motion_estimation_precision == 1 ? ME_HALF : ME_QUARTER);
	s.num_merge_cand = 2;
	static_assert(CFG_MAX_CU_SIZE == 64 && CFG_MAX_CU_SHIFT == 6 && CFG_MAX_PRED_DEPTH == 4 && CFG_NUM_MERGE_CAND == 2, "the decision code's constants (enc_types.h)");
	if (s.max_cu_depth != CFG_MAX_CU_DEPTH || s.max_pred_depth != CFG_MAX_PRED_DEPTH) { *why = "partition depths other than the built ones"; return false; }
	s.sign_hiding = cfg.sign_hiding;
	s.strong_intra = 1;
	s.chroma_qp_offset = cfg.chroma_qp_offset;
	s.sao = cfg.sample_adaptive_offset;
	s.wpp = cfg.wfpp_enable;
	s.bitrate_mode = cfg.bitrate_mode;
	s.qp = cfg.qp;
	s.intra_period = cfg.intra_period;
	s.gop_size = cfg.gop_size;
	s.num_ref_frames = cfg.num_ref_frames;
	s.reinit_gop = cfg.reinit_gop_on_scene_change;
	s.width = cfg.width;
	s.height = cfg.height;
	s.wctu = (s.width + 63) >> 6;
	s.hctu = (s.height + 63) >> 6;
	s.nctu = s.wctu * s.hctu;
	s.depth_start[0] = 0; s.depth_start[1] = 1; s.depth_start[2] = 5; s.depth_start[3] = 21; s.depth_start[4] = 85;
	// wnd_alloc(width, height, 64 + 16, 64 + 16), hmr_mem_transfer.c:48: strides are rounded to 16 bytes
	s.margin_y = 80;
	s.margin_c = 40;
	s.stride_y = ((s.width * 2 + 15) / 16 * 16) / 2 + 2 * s.margin_y;
	s.stride_c = (((s.width / 2) * 2 + 15) / 16 * 16) / 2 + 2 * s.margin_c;
	s.src_stride_y = ((s.width * 2 + 15) / 16 * 16) / 2;
	s.src_stride_c = (((s.width / 2) * 2 + 15) / 16 * 16) / 2;
	s.plane_elems_y = s.stride_y * (s.height + 2 * s.margin_y);
	s.plane_elems_c = s.stride_c * (s.height / 2 + 2 * s.margin_c);
	return true;
}

// init_partition_info + get_partition_neigbours
inline void make_geo(Geo *geo)
{
	memset(geo, 0, sizeof(Geo) * NNODES);
	geo[0].size = 64; geo[0].size_chroma = 32; geo[0].num_part = 256; geo[0].parent = -1;
	for (int k = 0; k < 4; k++) geo[0].child[k] = -1;
	int next = 1;
	for (int p = 0; p < 85; p++) {
		Geo &par = geo[p];
		const int depth = par.depth + 1, size = 64 >> depth, np = (size * size) >> 4;
		for (int k = 0; k < 4; k++, next++) {
			Geo &c = geo[next];
			par.child[k] = (int32_t)next;
			c.parent = (int32_t)p;
			c.list_index = (int32_t)next;
			c.depth = (int32_t)depth;
			c.size = (int32_t)size;
			c.size_chroma = (int32_t)(size >> 1);
			c.x = (int32_t)(par.x + (k & 1) * size);
			c.y = (int32_t)(par.y + (k >> 1) * size);
			c.xc = c.x >> 1;
			c.yc = c.y >> 1;
			c.abs_index = (int32_t)(par.abs_index + k * np);
			c.num_part = (int32_t)np;
			for (int j = 0; j < 4; j++) c.child[j] = -1;
		}
	}
	for (int i = 0; i < NNODES; i++) {
		Geo &c = geo[i];
		c.raster_index = (int32_t)host_abs2raster(c.abs_index);
		const int r = c.raster_index;
		const int left = (r & 15) == 0 ? r + 15 : r - 1;
		const int left_bottom = (left + 16) & 255, top = (r + 240) & 255;
		const int top_right = (top & 15) == 15 ? top - 15 : top + 1;
		const int top_left = (left + 240) & 255;
		c.abs_left = (int32_t)host_raster2abs(left);
		c.abs_left_bottom = (int32_t)host_raster2abs(left_bottom);
		c.abs_top = (int32_t)host_raster2abs(top);
		c.abs_top_right = (int32_t)host_raster2abs(top_right);
		c.abs_top_left = (int32_t)host_raster2abs(top_left);
	}
}

inline int host_chroma_qp(int qpi)
{
	static const uint8_t t[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
				      29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};
	return t[clampi(qpi, 0, 57)];
}

// sequence-level state the engine thread carries from frame to frame
// Engines (num_enc_engines = E > 1, encoder_engine_thread hmr_encoder_lib.c:3043-3330): frames are dealt to E engines that overlap in the reference, with a
// timing-dependent result; what is built is the interleaving oracle/ref_ctudump.c's engine turnstile pins on it - frame n sees the complete reconstruction
// of frame n - 1 and the frame-typing state of the single-engine run, starts from the avg_dist frame n - E left behind (zero for the first E frames), and
// works on the persistent state (CTU records, WPP thread mode buffers) of engine n mod E.  The frames may overlap as far as reference rows allow: none of
// that changes.  HostState is everything that travels from frame to frame besides the pictures; with one engine per GPU it is sent along with them.
struct HostState {
	int poc = 0, last_intra = 0, last_gop_reinit = 0, num_encoded_frames = 0;
	double avg_dist = 0.0;          // hvenc->avg_dist: calloc'ed, so the first frame runs with 0 (hmr_encoder_lib.c:3191)
	double avg_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // hvenc->avg_dist after each of the last eight frames
	int engines = 1, last_idr = 0;  // last_idr: picture order count of the last IDR picture
	RcState rc = {};                // rate control (enc_rc.h): hvenc->rc, pict_qp - with several engines the rc object of the engine whose frame is in flight
	// Rate control with several engines, as the engine turnstile of oracle/ref_ctudump.c pins it (hmr_encoder_lib.c:3195-3202, :3262-3279, hmr_rate_control.c:89-265):
	// a frame's engine copies hvenc->rc when the frame is fed - right after the engine's last frame n - E has ended - and makes its CTU decisions with that
	// (vbv_fullness E frames old, the engine's own pict_qp and avg_qp); the end sections run in frame order, each pushing vbv_fullness / acc_avg / acc_rate into
	// the engines ahead after their CTU work, so hmr_rc_end_pic(n) continues from what hmr_rc_end_pic(n - 1) left.  rc_hist[k & 7]: the state after frame k's end.
	RcState rc_hist[8] = {};
	RcState rc_start = {};          // what hmr_rc_init leaves (the first E frames are fed before any frame has ended)
};
// HVENC_Cfg rates -> the rate control's sequence state (HOMER_SETCFG hmr_encoder_lib.c:949-963 + hmr_rc_init)
inline void host_rc_init(const HostCfg &cfg, const Seq &s, HostState &st)
{
	double vbv_size = cfg.vbv_size, vbv_init = cfg.vbv_init;
	if (cfg.bitrate_mode == BR_VBR) {
		vbv_size = cfg.vbv_size * 20;
		vbv_init = ((double)cfg.vbv_init / (double)cfg.vbv_size) * vbv_size;
	}
	rc_init(st.rc, (double)cfg.bitrate, vbv_size, vbv_init, cfg.frame_rate, s.nctu, cfg.qp);
	st.rc_start = st.rc;
}
constexpr int MAX_ENGINES = 8;       // hmr_private.h:1232

enum { IMG_AUTO = 0, IMG_B = 1, IMG_P = 2, IMG_I = 3 };

// frame typing (put_frame_to_encode :311-331) and the per-frame scalars (hmr_slice_init :1986, hmr_rd_init hmr_tables.c:315)
inline void begin_frame(const Seq &s, HostState &st, int image_type, FrameCtx &f)
{
	memset(&f, 0, sizeof f);
	const int poc = st.poc++;
	const bool intra = poc == 0 || (s.intra_period != 0 && poc == st.last_intra + s.intra_period && image_type == IMG_AUTO) || image_type == IMG_I;
	if (intra) {
		st.last_intra = poc;
		st.last_gop_reinit = poc;
		st.last_idr = poc;
	}
	f.slice_type = intra ? SLICE_I : SLICE_P;
	f.poc = poc;
	f.last_idr = st.last_idr;
	f.qp = s.qp;
	if (s.bitrate_mode != BR_FIXED_QP) {
		// hmr_slice_init :1990 (the slice QP is the frame QP the engine's last frame left), hmr_rc_init_pic
		if (st.engines > 1) st.rc = st.num_encoded_frames >= st.engines ? st.rc_hist[(st.num_encoded_frames - st.engines) & 7] : st.rc_start;
		f.qp = st.rc.pict_qp;
		rc_init_pic(st.rc, f.slice_type, s.intra_period);
		rc_frame_view(st.rc, s.nctu, s.bitrate_mode, f.rc);
		f.rc.sqrt_clipped_intra_period = sqrt((double)rc_clipped_intra_period(s.intra_period));
	}
	f.num_encoded_frames = st.num_encoded_frames;
	f.is_scene_change = 0;
	f.scene_cut_ctu = -1;
	f.lockstep = 0;
	f.wctu = s.wctu;
	f.scene_cut_allowed = f.slice_type == SLICE_P && st.num_encoded_frames > 1 && 20 < poc - st.last_gop_reinit;
	f.ref_poc = poc - 1;
	f.avg_dist = st.num_encoded_frames >= st.engines ? st.avg_hist[(st.num_encoded_frames - st.engines) & 7] : 0.0;   // one engine: the frame before
	const double qp_temp = (double)f.qp - 12;      // hmr_rd_init: enc_engine->pict_qp
	const double lambda_scale = 1.0 - fmin(fmax(0.05 * (double)(s.gop_size - 1), 0.0), 0.5);
	double qp_factor = 0.4624;
	if (f.slice_type == SLICE_I) qp_factor = 0.57 * lambda_scale;
	double lambda = qp_factor * pow(2.0, qp_temp / 3.0);
	if (f.slice_type != SLICE_I) lambda *= .95;
	const double weight = pow(2.0, (f.qp - host_chroma_qp(f.qp + s.chroma_qp_offset)) / 3.0);
	f.lambda = lambda;
	f.sqrt_lambda = sqrt(lambda);
	f.chroma_weight = weight;
	f.sao_lambda[0] = lambda;
	f.sao_lambda[1] = f.sao_lambda[2] = lambda / weight;
}

// The frame's distortion total as the reference forms it (:3217-3228, hmr_private.h:1217): every WPP thread adds up the root distortions of the CTUs of
// its rows (row r belongs to thread r mod T) in a uint32 that may wrap, the engine adds the threads' totals in a double.  dist_of(n) = CTU n's distortion.
template <class DistFn>
inline double frame_acc_dist(const Seq &s, int threads, DistFn &&dist_of)
{
	const int T = threads < 1 ? 1 : threads;
	double total = 0;
	for (int t = 0; t < T && t < s.hctu; t++) {
		uint32_t acc = 0;
		for (int r = t; r < s.hctu; r += T)
			for (int c = 0; c < s.wctu; c++) acc += dist_of(r * s.wctu + c);
		total += acc;
	}
	return total;
}

// what the rate control takes from a finished frame: the sum of the CTUs' QPs (acc_qp, hmr_encoder_lib.c:2938), the bits of all its CTUs, and the picture target as
// the frame left it (hmr_rc_change_pic_mode moves it when a scene change is found)
struct FrameRcOut {
	int sum_qp;
	double consumed_bits, target_pict_size;
};
// :3217-3262 after the CTUs of a frame: acc_dist = frame_acc_dist
inline void end_frame(const Seq &s, HostState &st, const FrameCtx &f, double acc_dist, const FrameRcOut *rc = nullptr)
{
	const bool scene_change = f.scene_cut_ctu >= 0;     // :3796-3800: the frame was found to be a new scene while it was encoded
	if (scene_change) {
		if (s.reinit_gop) st.last_intra = f.poc;
		st.last_gop_reinit = f.poc;
	}
	if (st.num_encoded_frames == 0 || f.slice_type != SLICE_I || s.intra_period == 1) {
		double a = acc_dist;
		a /= s.nctu * NPART;
		a = a < .1 ? .1 : a;
		if (f.slice_type == SLICE_I) a *= 1.5;
		else if (scene_change) a *= 1.375;
		st.avg_dist = a;
	}
	if (s.bitrate_mode != BR_FIXED_QP && rc) {
		if (st.engines > 1 && st.num_encoded_frames >= 1) {      // (the push of the frames that ended before this one: the last one's counts)
			const RcState &prev = st.rc_hist[(st.num_encoded_frames - 1) & 7];
			st.rc.vbv_fullness = prev.vbv_fullness;
			st.rc.acc_avg = prev.acc_avg;
			st.rc.acc_rate = prev.acc_rate;
		}
		rc_end_pic(st.rc, f.slice_type, s.intra_period, s.bitrate_mode, s.nctu, st.num_encoded_frames == 0 || f.slice_type != SLICE_I || s.intra_period == 1, rc->sum_qp, st.avg_dist,
			   scene_change, rc->consumed_bits, rc->target_pict_size);
		st.rc_hist[st.num_encoded_frames & 7] = st.rc;
	}
	st.avg_hist[st.num_encoded_frames & 7] = st.avg_dist;    // (an I frame inside the sequence keeps the value pushed by the frame before it, :3268-3279)
	st.num_encoded_frames++;
}

}  // namespace henc
