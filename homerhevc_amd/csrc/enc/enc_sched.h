// Running the CTU rows of a frame side by side and still producing, bit for bit, what the reference produces with one worker
// thread that walks the picture in raster order (wfpp_num_threads = 1, the deterministic configuration every fixture is minted from).
//
// Two inputs of a CTU come from ALL the CTUs before it in raster order, not just from its neighbours:
//   (a) the running share of intra partitions, used when an intra candidate is compared with the inter one (hmr_motion_inter.c:3767-3776, :4018);
//   (b) the worker thread's intra-mode buffers, which are never reset: a most-probable-mode look-up for a neighbour inside the CTU can
//       land on a value some earlier CTU left there (hmr_motion_intra.c:1102 -> get_intra_dir_luma_predictor).
// A row worker does not have them when it starts (the row above is only two CTUs ahead).  The scheme here:
//   * encode with guesses, logging every use: the comparisons (enc_ctu.h) and the inherited modes looked at (enc_intra.h);
//   * afterwards walk the frame in raster order, rebuild the true chain of (a) and (b) from the CTUs' own outputs and check every
//     log against it (sched_chain_step / sched_guesses_hold below);
//   * re-encode the CTUs whose guesses were wrong with the true values, and the CTUs that read a neighbour whose output changed;
//     repeat until a walk finds nothing wrong.  The first wrong CTU in raster order always gets exact inputs, so this terminates,
//     and when it does every CTU has seen exactly what the single thread would have shown it.
// With one WPP thread per CTU row (wfpp_num_threads = CTU rows) none of this runs: each row worker owns its buffers for the whole sequence and
// the counters of (a) are read as of the end of the previous wavefront step - the synchronous-wavefront schedule (k_encode.hip, lockstep path; the checker:
// oracle/enc_cpu.cpp frame_ctus_lockstep; the pin: oracle/ref_ctudump.c HOMER_TURNSTILE).
// The functions are shared by the gfx950 kernels (k_encode.hip) and the one-lane checker build (oracle/enc_cpu.cpp).
#pragma once
#include "enc_ctu.h"

#if !defined(__HIPCC__) && defined(HENC_SCHED_CAUSES)
extern "C" int henc_sched_causes[8];   // checker build: why CTUs failed (0 replay impossible, 1 winner changed, 2 outcome flipped by mode bits, 3 by the ratio, 4 intra kept, bits changed)
#define SCHED_CAUSE(k) (henc_sched_causes[k]++)
#else
#define SCHED_CAUSE(k) ((void)0)
#endif

namespace henc {

constexpr int MODE_STATE_BYTES = 2 * NDEPTH * NPART;   // one snapshot of Work::intra_mode_buffs

// the chain (b) for one unit column k: st[comp][depth] before the CTU -> after it, given the CTU's buffers at its end (values or tokens)
HENC_INLINE void sched_chain_step(uint8_t st[2][NDEPTH], const uint8_t *out_tokens, int k)
{
	uint8_t nx[2][NDEPTH];
	for (int comp = 0; comp < 2; comp++)
		for (int d = 0; d < NDEPTH; d++) {
			const uint8_t v = out_tokens[(comp * NDEPTH + d) * NPART + k];
			nx[comp][d] = (v & MODE_TOKEN) ? st[comp][v & 7] : v;
		}
	for (int comp = 0; comp < 2; comp++)
		for (int d = 0; d < NDEPTH; d++) st[comp][d] = nx[comp][d];
}

// one logged search replayed with the true neighbour directions: 1 when the walk could be replayed, *mode / *bits = its outcome
HENC_INLINE int sched_replay_search(const SearchLog &lg, const uint8_t *true_in, double sqrt_lambda, int *mode_g, int *bits_g, int *mode_t, int *bits_t)
{
	int td[2], pg[3], pt[3];
	double cg = 0, ct = 0;
	for (int k = 0; k < 2; k++) td[k] = (lg.src[k] & 0x8000) ? true_in[((lg.src[k] >> 8) & 7) * NPART + (lg.src[k] & 255)] : lg.used[k];
	mpm_from_dirs(lg.used[0], lg.used[1], pg);
	mpm_from_dirs(td[0], td[1], pt);
	auto table = [&](int mode) -> int64_t {
		for (int k = 0; k < lg.n; k++)
			if (lg.mode[k] == mode) return (int64_t)lg.sad[k];
		return -1;
	};
	*bits_g = intra_search_walk(pg, 1, sqrt_lambda, table, mode_g, &cg);
	*bits_t = intra_search_walk(pt, 1, sqrt_lambda, table, mode_t, &ct);
	return *bits_t >= 0;
}

// 1 when CTU `c` would have come out the same with the true inputs: true_in = the mode buffers as the single thread would have had
// them at the CTU's start, used_in = what the CTU was given, intra_before / parts_before = the true counters of (a).
//   * a search outside the P-slice walk must keep its winner and its mode bits;
//   * a search inside it feeds exactly one intra / inter comparison: the winner must stay, the mode bits may change as long as the
//     inter candidate wins either way (the intra cost is then discarded; what the evaluation left in the buffers is the winner);
//   * every comparison must keep its outcome under the true ratio, and where intra wins, the cost it leaves behind.
template <class G>
HENC_HD int sched_guesses_hold(const G g, const CtuInfo &c, const FrameCtx &f, const uint8_t *true_in, const uint8_t *used_in, uint32_t intra_before,
			       uint32_t parts_before, uint32_t used_intra, uint32_t used_parts, int uses_ratio)
{
	int bad = 0;
	if (c.walk_intra != ctu_takes_intra_walk(f, c.ctu_number)) return 0;   // the scene cut moved: the CTU took the wrong walk
	if (c.walk_intra) uses_ratio = 0;
	if (c.n_spec_reads > MAX_SEARCH_LOGS || c.n_ratio_cmp > MAX_RATIO_CMP) {
		// log overflow: everything the CTU was given has to be right
		for (int i = g.tid; i < NDEPTH * NPART; i += g.n) bad |= true_in[i] != used_in[i];
		if (uses_ratio) bad |= intra_ratio(intra_before, parts_before) != intra_ratio(used_intra, used_parts);
		return !g.any(bad);
	}
	for (int i = g.tid; i < c.n_spec_reads; i += g.n) {
		const SearchLog &lg = c.slog[i];
		if (lg.has_cmp) continue;
		int mg, bg, mt, bt;
		bad |= !sched_replay_search(lg, true_in, f.sqrt_lambda, &mg, &bg, &mt, &bt) || mg != mt || bg != bt;
	}
	if (uses_ratio) {
		const double ratio = intra_ratio(intra_before, parts_before), used_ratio = intra_ratio(used_intra, used_parts);
		const double correction = calc_mv_correction((uint32_t)f.qp, f.avg_dist);
		for (int i = g.tid; i < c.n_ratio_cmp; i += g.n) {
			const double *lg = c.ratio_cmp + 4 * i;
			double intra_dist = lg[0];
			int same_bits = 1;
			if (c.ratio_slog[i] >= 0) {
				const SearchLog &sl = c.slog[c.ratio_slog[i]];
				int mg, bg, mt, bt;
				if (!sched_replay_search(sl, true_in, f.sqrt_lambda, &mg, &bg, &mt, &bt) || mg != mt) { bad = 1; SCHED_CAUSE(bt < 0 ? 0 : 1); continue; }
				same_bits = bg == bt;
				intra_dist = (double)((uint32_t)lg[0] - intra_luma_cost(sl.tu_cost, bg, correction) + intra_luma_cost(sl.tu_cost, bt, correction));
			}
			// inter wins either way: the intra cost is discarded.  Intra wins either way: the cost it leaves in the node ((uint32_t), :4024) must be the same.
			const double ic = intra_cost_with_ratio(intra_dist, ratio, lg[1], lg[2]), ic_used = intra_cost_with_ratio(lg[0], used_ratio, lg[1], lg[2]);
			const int take = ic < lg[3];
			if (take != (c.ratio_out[i] != 0)) SCHED_CAUSE(c.ratio_slog[i] >= 0 && !same_bits ? 2 : 3);
			else if (take && (!same_bits || (uint32_t)ic != (uint32_t)ic_used)) SCHED_CAUSE(4);
			bad |= take != (c.ratio_out[i] != 0) || (take && (!same_bits || (uint32_t)ic != (uint32_t)ic_used));
		}
	}
	return !g.any(bad);
}

// what other CTUs can see of a CTU: its side-info arrays and its reconstruction.  Two independent 32-bit sums of products.
template <class G>
HENC_HD uint64_t sched_output_hash(const G g, const Seq &S, const FrameCtx &f, const CtuInfo &c)
{
	uint32_t h1 = 0, h2 = 0;
	const uint32_t *p = (const uint32_t *)(const CtuPublic *)&c;
	const int words = (int)(offsetof(CtuPublic, sao_recon) / 4);
	for (int i = g.tid; i < words; i += g.n) {
		h1 += p[i] * (2654435761u + 2u * (uint32_t)i);
		h2 += (p[i] ^ 0x9e3779b9u) * (40503u + 2u * (uint32_t)i + 1u);
	}
	for (int comp = 0; comp < 3; comp++) {
		const int sz = comp ? 32 : 64, px = comp ? c.x >> 1 : c.x, py = comp ? c.y >> 1 : c.y;
		const int pw = comp ? S.width >> 1 : S.width, ph = comp ? S.height >> 1 : S.height, rs = comp ? S.stride_c : S.stride_y;
		const int ww = (px + sz) < pw ? sz : pw - px, hh = (py + sz) < ph ? sz : ph - py;
		const int16_t *r = f.rec[comp] + py * rs + px;
		for (int i = g.tid; i < ww * hh; i += g.n) {
			const uint32_t v = (uint16_t)r[(i / ww) * rs + (i % ww)];
			h1 += v * (2246822519u + 2u * (uint32_t)(i + comp * 4096));
			h2 += (v + 0x632be5abu) * (3266489917u + 2u * (uint32_t)(i + comp * 4096));
		}
	}
	h1 = g.sum(h1);
	h2 = g.sum(h2);
	return ((uint64_t)h1 << 32) | h2;
}

// the intra statistics the wavefront order guarantees to be known when CTU (row, col) starts: row - k has finished col + 2k CTUs
// (prefix[r * (W + 1) + i] = intra partitions of the first i CTUs of row r).  The first guess of the ratio is the share among those.
HENC_INLINE void sched_known_intra(const uint32_t *prefix, int W, int row, int col, uint32_t *intra, uint32_t *parts)
{
	uint32_t ti = prefix[(size_t)row * (W + 1) + col], tp = (uint32_t)col;
	for (int k = 1; k <= row; k++) {
		const int have = col + 2 * k < W ? col + 2 * k : W;
		ti += prefix[(size_t)(row - k) * (W + 1) + have];
		tp += (uint32_t)have;
	}
	*intra = ti;
	*parts = tp * NPART;
}

}  // namespace henc
