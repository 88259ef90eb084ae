// Rate control (bitrate_mode CBR / VBR): hmr_rate_control.c restated.
//   * per sequence / per frame on the host: hmr_rc_init :31, hmr_rc_init_pic :89, hmr_rc_end_pic :152, the frame QP of encoder_engine_thread
//     (hmr_encoder_lib.c:3217-3262: pict_qp = the rounded mean of the CTUs' QPs, compensated after an intra frame);
//   * per CTU in the decision walk (device): hmr_rc_calc_cu_qp :266 - with qp_depth = 0 (:983 of hmr_encoder_lib.c: "if rc enabled qp_depth == 0") the QP is
//     computed at the CTU's root and every deeper CU takes its parent's (hmr_rc_get_cu_qp :337);
//   * hmr_rc_change_pic_mode :52 when a scene change is detected inside a P frame.
// What hmr_rc_calc_cu_qp reads of the frame so far is the bits and the number of the CTUs that have been ENTROPY CODED (henc_thread_t num_bits / num_encoded_ctus,
// summed over the WPP threads, :276-282) - and the reference codes a CTU inside its CTU-lagged filter pipeline, a fixed distance behind the CTU being decided
// (hmr_deblock_sao_pad_sync_ctu, hmr_encoder_lib.c:2386-2843).  rc_coded_by() replays that lag arithmetic: which CTU's post-decision section codes which CTU.
// In the synchronous wavefront (enc_sched.h; the interleaving oracle/ref_ctudump.c pins on the reference) the post-decision sections of a step's CTUs run after
// all of the step's decisions, so the CTUs of step t all see the CTUs coded by the sections of the steps before t: rc_need_table().
#pragma once
#include "enc_types.h"
#include "enc_platform.h"

namespace henc {

enum { BR_FIXED_QP = 0, BR_CBR = 1, BR_VBR = 2 };

// rate_control_t (hmr_private.h:977-990) + what the engine thread carries beside it
struct RcState {
	double vbv_size, average_pict_size, average_bits_per_ctu, vbv_fullness, target_pict_size, target_bits_per_ctu, acc_rate, acc_avg;
	int32_t extra_bits;
	int32_t pict_qp;              // enc_engine->pict_qp: the slice QP of the next frame
	int32_t avg_qp_carry;         // encoder_engine_thread's `avg_qp`: a local that survives from frame to frame and is only reset when the frame's statistics are taken (:3219)
	int32_t pad_;
};

// hmr_rc_calc_cu_qp :266.  consumed_bits / consumed_ctus: the CTUs entropy coded so far (without extra_bits); is_scene_change: the frame has been found to be a new scene
HENC_INLINE int rc_calc_cu_qp(const RcFrame &rc, double consumed_bits, int consumed_ctus, int slice_type, int is_scene_change, int reinit_gop, int intra_period, double avg_dist,
			      int num_encoded_frames)
{
	double qp, pic_corrector = 0.0, vbv_corrector;
	const double entropy = 3;
	double consumed_bitrate = consumed_bits + rc.extra_bits;
	if (consumed_bitrate > 1.5 * rc.target_bits_per_ctu * consumed_ctus) {
		if (slice_type == SLICE_I) pic_corrector = 2.5 * .0125 * (consumed_bitrate / (rc.target_bits_per_ctu * consumed_ctus));
		else pic_corrector = .0125 * (consumed_bitrate / (rc.target_bits_per_ctu * consumed_ctus));
	}
	pic_corrector = hclip(pic_corrector, 0., .5);
	const double min_vbv_size = hclip(rc.vbv_fullness, rc.vbv_fullness, rc.vbv_size * .95);
	if (consumed_bitrate > rc.target_bits_per_ctu * consumed_ctus)
		vbv_corrector = 1.0 - hclip((min_vbv_size - consumed_bitrate + rc.target_bits_per_ctu * consumed_ctus) / rc.vbv_size, 0.0, 1.0);
	else
		vbv_corrector = 1.0 - hclip((min_vbv_size) / rc.vbv_size, 0.0, 1.0);
	qp = ((pic_corrector + vbv_corrector) / 1.) * (51) + (entropy - 3.);
	if (rc.is_vbr) {
		if (qp < rc.qp_min) qp = rc.qp_min;
	}
	if (intra_period > 1) {
		if (slice_type == SLICE_I || (is_scene_change && reinit_gop)) qp /= hclip(1.5 - (avg_dist / 15000.), 1.15, 1.5);
		else if (is_scene_change) qp /= 1.1;      // (the B-slice branch :313 is outside the built configurations)
	}
	if (is_scene_change && qp <= 5) qp = 5;
	if (num_encoded_frames == 0) qp += 4;
	else if (slice_type == SLICE_I && consumed_bitrate > 1. * (rc.target_bits_per_ctu * consumed_ctus) && rc.vbv_fullness < .5 * rc.vbv_size) qp += 2;
	return (int)hclip(qp + .5, 1.0, 51.);
}

// hmr_rc_change_pic_mode :52 (a scene change detected inside a P frame): new picture target from what has been coded so far
HENC_INLINE void rc_change_pic_mode(RcFrame &rc, int reinit_gop, int intra_period, int nctu, double sqrt_clipped_intra_period, uint32_t consumed_bits, int consumed_ctus)
{
	(void)intra_period;
	double pic_size_new;
	if (reinit_gop && rc.vbv_fullness < .5 * rc.vbv_size) pic_size_new = 1. * rc.average_pict_size * sqrt_clipped_intra_period;
	else pic_size_new = .75 * rc.average_pict_size * sqrt_clipped_intra_period;
	rc.target_pict_size = hmin(pic_size_new, rc.vbv_fullness);
	rc.target_bits_per_ctu = rc.target_pict_size / nctu;
	rc.extra_bits = (int32_t)(rc.target_pict_size * ((double)consumed_ctus / nctu) - (int)consumed_bits);
}

}  // namespace henc
#include <math.h>
#include <vector>
namespace henc {

// hmr_rc_init :31
inline void rc_init(RcState &rc, double bitrate, double vbv_size, double vbv_init, float frame_rate, int nctu, int qp)
{
	memset(&rc, 0, sizeof rc);
	rc.vbv_size = vbv_size * 1000;
	rc.vbv_fullness = vbv_init * 1000;
	rc.average_pict_size = bitrate * 1000 / frame_rate;
	rc.average_bits_per_ctu = rc.average_pict_size / nctu;
	rc.pict_qp = qp;
}
inline int rc_clipped_intra_period(int intra_period) { return intra_period == 0 ? 20 : intra_period; }

// hmr_rc_init_pic :89 (I and P slices; num_b = 0)
inline void rc_init_pic(RcState &rc, int slice_type, int intra_period)
{
	const int cip = rc_clipped_intra_period(intra_period);
	const double intra_avg_size = 2.25 * rc.average_pict_size * sqrt((double)cip);
	rc.extra_bits = 0;
	if (slice_type == SLICE_I) rc.target_pict_size = hmin(intra_avg_size, rc.vbv_fullness);
	else rc.target_pict_size = (rc.average_pict_size * cip - intra_avg_size) / (cip - 1);
}
inline void rc_frame_view(const RcState &rc, int nctu, int bitrate_mode, RcFrame &f)
{
	f.vbv_size = rc.vbv_size; f.vbv_fullness = rc.vbv_fullness; f.target_pict_size = rc.target_pict_size; f.target_bits_per_ctu = rc.target_pict_size / nctu;
	f.average_pict_size = rc.average_pict_size;
	f.extra_bits = rc.extra_bits;
	f.on = bitrate_mode != BR_FIXED_QP;
	f.is_vbr = bitrate_mode == BR_VBR;
	f.qp_min = bitrate_mode == BR_VBR ? 15 : 0;      // hmr_encoder_lib.c:957
}

inline double rc_compensate_qp_from_intra(double avg_dist, double qp) { return hclip(qp * (hclip(1.5 - (avg_dist / 15000.), 1.15, 1.5)), 1.0, 51.); }

// end of a frame: the frame QP (hmr_encoder_lib.c:3217-3247) and hmr_rc_end_pic :152.  sum_qp: the sum of the CTUs' QPs when the frame's statistics are taken
// (`stats_taken`: the first frame, a P frame, or intra_period 1), consumed_bits: all CTUs' bits; target_pict_size: as the frame left it (a scene change moves it)
inline void rc_end_pic(RcState &rc, int slice_type, int intra_period, int bitrate_mode, int nctu, bool stats_taken, int sum_qp, double avg_dist, int is_scene_change, double consumed_bits,
		       double target_pict_size)
{
	int avg_qp = rc.avg_qp_carry;
	if (stats_taken) avg_qp = sum_qp;
	avg_qp = (avg_qp + (nctu >> 1)) / nctu;
	rc.avg_qp_carry = avg_qp;
	rc.pict_qp = hclip(avg_qp, 1, 51);
	if (slice_type == SLICE_I && intra_period != 1) rc.pict_qp = (int32_t)rc_compensate_qp_from_intra(avg_dist, rc.pict_qp);
	rc.target_pict_size = target_pict_size;
	double consumed_bitrate = consumed_bits;
	const int avg_rate_period = intra_period == 0 ? 100 : intra_period;
	rc.vbv_fullness += rc.average_pict_size;
	if (slice_type == SLICE_I && intra_period != 1) {
		const double aux = 3. * consumed_bitrate / 5.;
		rc.acc_rate += aux;
		consumed_bitrate -= aux;
		rc.acc_avg = rc.acc_rate / avg_rate_period;
		rc.vbv_fullness -= consumed_bitrate + rc.acc_avg;
		rc.acc_rate -= rc.acc_avg;
	} else if (is_scene_change && intra_period != 1) {
		if (rc.vbv_fullness < .5 * rc.vbv_size) {
			rc.acc_rate += consumed_bitrate - rc.average_pict_size;
			consumed_bitrate = rc.average_pict_size;
		} else {
			rc.acc_rate += consumed_bitrate / 3;
			consumed_bitrate = 2 * consumed_bitrate / 3;
		}
		rc.acc_avg = rc.acc_rate / avg_rate_period;
		rc.vbv_fullness -= consumed_bitrate + rc.acc_avg;
		rc.acc_rate -= rc.acc_avg;
	} else {
		if (bitrate_mode == BR_VBR && slice_type != SLICE_I) {
			if (consumed_bitrate < .45 * rc.target_pict_size && rc.vbv_fullness < .75 * rc.vbv_size) {
				rc.acc_rate += (.005 * rc.vbv_size);
				consumed_bitrate -= (.005 * rc.vbv_size);
				rc.acc_avg = rc.acc_rate / avg_rate_period;
			} else if (consumed_bitrate > 1.55 * rc.target_pict_size && rc.vbv_fullness > .1 * rc.vbv_size) {
				rc.acc_rate -= (.005 * rc.vbv_size);
				consumed_bitrate += (.005 * rc.vbv_size);
				rc.acc_avg = rc.acc_rate / avg_rate_period;
			}
		}
		rc.vbv_fullness -= consumed_bitrate;
		rc.vbv_fullness -= rc.acc_avg;
		rc.acc_rate -= rc.acc_avg;
	}
	if (rc.vbv_fullness > rc.vbv_size) rc.vbv_fullness = rc.vbv_size;
	if (rc.vbv_fullness < 0) rc.vbv_fullness = 0;
}

// Which CTU's post-decision section entropy codes CTU m: hmr_deblock_sao_pad_sync_ctu :2386 for an inter GOP (intra_period != 1: always, the period is clamped to
// at least 2) - the regular lag, the flush at the end of each CTU row and the flush at the end of the picture.  coded_by[m] = n.
inline void rc_coded_by(int W, int H, int sao, std::vector<int> &coded_by)
{
	const int total = W * H;
	coded_by.assign(total, -1);
	auto code = [&](int m, int n) { if (m >= 0 && m < total && coded_by[m] < 0) coded_by[m] = n; };
	for (int n = 0; n < total; n++) {
		const int idx = n % W;
		int v = n - (W + 1);
		const int vi = v % W;               // (C remainder: negative for a negative v, as in the reference)
		int h = v - 1;
		const int pad = sao ? h - (W + 1) : h - W;
		int s = pad;
		if (!sao) {
			code(n, n);
			continue;
		}
		if (v >= 0 && idx >= 1 && h >= 0 && idx >= 2 && s >= 0 && idx >= 3) code(s, n);
		if ((vi + 1) == W - 1 && (n + 1) != total) {
			int max_filter = ((v / W) + 1) * W;
			if (s > 0) {
				max_filter -= W;
				for (int a = s + 1; a < max_filter; a++) code(a, n);
			}
		}
		if ((n + 1) == total) {
			if (s < 0) s = -1;
			for (int a = s + 1; a < total; a++) code(a, n);
		}
	}
}
// ---- RD_FULL: which contexts a CTU's bit estimates copy (enc_rdo.h) -------------------------------------------------------------------------------------------
// The estimates copy from et->ee, the coder object the WPP thread selected last (wfpp_encode_select_bitstream, hmr_encoder_lib.c:2299-2344): there are 2 T objects
// in ee_list (T threads; :1096, created with all-zero states :1116-1121), thread i starts with ee_list[2 i] (:1439) and keeps its pointer across frames.  Selecting
// CTU m of row R for coding: at the row's first CTU (R > 0) the slots 2 (R mod T) and 2 (R mod T) - 1 swap their objects - the second one holds the states the
// row above saved after ITS second CTU (wfpp_encode_ctu :2368-2373, into slot 2 ((R - 1) mod T) + 1) - then the thread's pointer is the object in slot 2 (R mod T);
// CTU 0 resets the object to the slice's initial states.  Coding CTU (R, x) leaves the object with the sub-stream's states after x + 1 CTUs.  So the content of any
// object is "the states of sub-stream R of frame f after k CTUs" (or all-zero) - a version.  RdCtxSim replays a frame's sections in the synchronous-wavefront order
// (decisions of a step, then the steps' sections by rows) and says which version every CTU's decisions see.
struct RdCtxVersion {
	int frame, row, k;      // frame < 0: the all-zero states the objects are created with; else sub-stream `row` of `frame` after `k` CTUs; k = 0: the slice's initial states
};
struct RdCtxSim {
	int T = 0, W = 0, H = 0, sao = 0;
	std::vector<int> slot, thread_ee;
	std::vector<RdCtxVersion> obj;
	void init(int threads, int wctu, int hctu, int sao_on)
	{
		T = threads; W = wctu; H = hctu; sao = sao_on;
		slot.resize(2 * T);
		thread_ee.resize(T);
		obj.assign(2 * T, RdCtxVersion{-1, 0, 0});
		for (int i = 0; i < 2 * T; i++) slot[i] = i;
		for (int i = 0; i < T; i++) thread_ee[i] = 2 * i;
	}
	// src[n]: what CTU n's decisions copy in frame f.  raster: one thread in raster order (wfpp_num_threads = 1: T = 1) - a CTU's decisions see what the sections
	// behind all the CTUs before it have left; else the synchronous wavefront.
	void frame(int f, std::vector<RdCtxVersion> &src, bool raster = false)
	{
		const int total = W * H, num_ee = 2 * T, last = W + 2 * (H - 1);
		src.assign(total, RdCtxVersion{-1, 0, 0});
		std::vector<char> coded(total, 0);
		// wfpp_encoder_thread :2865-2877: every frame, thread 0 takes the object in slot 0 and resets it to the slice's initial states before its first CTU
		thread_ee[0] = slot[0];
		obj[slot[0]] = RdCtxVersion{f, 0, 0};
		auto code = [&](int th, int m) {
			if (m < 0 || m >= total || coded[m]) return;
			coded[m] = 1;
			const int R = m / W, x = m % W, idx = R % T;
			if (m != 0 && R > 0 && x == 0) std::swap(slot[2 * idx], slot[(2 * idx + num_ee - 1) % num_ee]);
			thread_ee[th] = slot[2 * idx];
			obj[thread_ee[th]] = RdCtxVersion{f, R, x + 1};
			if (x == 1 && R + 1 != H) obj[slot[(2 * idx + 1) % num_ee]] = RdCtxVersion{f, R, 2};
		};
		// the entropy coding calls of the section behind CTU n = (r, c) (hmr_deblock_sao_pad_sync_ctu; the same lag arithmetic as rc_coded_by)
		auto section = [&](int r, int c) {
			const int n = r * W + c, th = r % T, idx = c;
			if (!sao) { code(th, n); return; }
			const int v = n - (W + 1), vi = v % W, h = v - 1;
			int s = h - (W + 1);
			if (v >= 0 && idx >= 1 && h >= 0 && idx >= 2 && s >= 0 && idx >= 3) code(th, s);
			if ((vi + 1) == W - 1 && (n + 1) != total) {
				int max_filter = ((v / W) + 1) * W;
				if (s > 0) {
					max_filter -= W;
					for (int a = s + 1; a < max_filter; a++) code(th, a);
				}
			}
			if ((n + 1) == total) {
				if (s < 0) s = -1;
				for (int a = s + 1; a < total; a++) code(th, a);
			}
		};
		if (raster) {
			for (int n = 0; n < total; n++) {
				src[n] = obj[thread_ee[(n / W) % T]];
				section(n / W, n % W);
			}
			return;
		}
		for (int t = 0; t < last; t++) {
			for (int r = 0; r < H; r++) {
				const int c = t - 2 * r;
				if (c >= 0 && c < W) src[r * W + c] = obj[thread_ee[r % T]];
			}
			for (int r = 0; r < H; r++) {
				const int c = t - 2 * r;
				if (c < 0 || c >= W) continue;
				section(r, c);
			}
		}
	}
};

// need[k * H + r] = the CTUs of row r that are coded when the decisions with index k start - synchronous wavefront: k = the step, 0 .. steps (the last entry is
// the whole picture); raster order (one thread): k = the CTU, 0 .. nctu.  False when a row's coded CTUs are not a prefix of the row (no picture grid the encoder
// accepts does that).
inline bool rc_need_table(int W, int H, int sao, bool wavefront, std::vector<uint16_t> &need)
{
	std::vector<int> by;
	rc_coded_by(W, H, sao, by);
	const int last = wavefront ? W + 2 * (H - 1) : W * H;
	need.assign((size_t)(last + 1) * H, 0);
	bool prefix = true;
	for (int r = 0; r < H; r++) {
		// index of the decisions that first see CTU (r, c) coded: one past its coder's
		std::vector<int> seen(W);
		for (int c = 0; c < W; c++) {
			const int n = by[r * W + c];
			seen[c] = n < 0 ? last + 1 : (wavefront ? n % W + 2 * (n / W) : n) + 1;
			if (c > 0 && seen[c] < seen[c - 1]) prefix = false;
		}
		for (int k = 0; k <= last; k++) {
			int cnt = 0;
			while (cnt < W && seen[cnt] <= k) cnt++;
			need[(size_t)k * H + r] = (uint16_t)cnt;
		}
	}
	return prefix;
}

}  // namespace henc
