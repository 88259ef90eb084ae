// Device building blocks shared by the intra kernels (k_intra.hip: prediction / reference build as separate batches,
// k_intrasearch.hip: the fused mode search).  Reference semantics: hmr_sse42_functions_prediction.c:199,926 (scalar spec
// hmr_motion_intra.c:408-625), fill_reference_samples hmr_motion_intra.c:246, adi_filter :189.
//
// `adi` is the 4N+1 neighbour array (bottom-left .. top-left .. top-right), mid = adi + 2N the corner sample; `mainr` is the
// projected main reference of an angular mode with origin at index 0 (valid -N+1 .. 2N).  G lanes (l = 0..G-1) cooperate.
#pragma once
#include "common.h"

namespace {

// ang_table {0, 2, 5, 9, 13, 17, 21, 26, 32} and inv_ang_table {0, 4096, 1638, 910, 630, 482, 390, 315, 256} (hmr_encoder_lib.c:35-36) as
// immediates: a table in memory would put a dependent load in front of every prediction
__device__ __forceinline__ int intra_ang(int aa)
{
	const unsigned long long t = 0ull | (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) | (26ull << 42) | (32ull << 48);
	return (int)((t >> (6 * aa)) & 63);
}
__device__ __forceinline__ int intra_inv_ang(int aa)
{
	const unsigned long long lo = 0ull | (4096ull << 16) | (1638ull << 32) | (910ull << 48), hi = 630ull | (482ull << 16) | (390ull << 32) | (315ull << 48);
	return aa >= 8 ? 256 : (int)(((aa < 4 ? lo : hi) >> (16 * (aa & 3))) & 0xffff);
}

struct IntraMode {
	int mode, angle, inv_angle, sm;   // sm: main[idx] = mid[sm*idx], side[k] = mid[-sm*k]; +1 for vertical modes
	bool is_ver, is_hor;
};

__device__ __forceinline__ IntraMode intra_mode_setup(int mode)
{
	IntraMode m;
	m.mode = mode;
	m.is_hor = mode >= 2 && mode < 18;
	m.is_ver = mode >= 18;
	m.angle = m.is_ver ? mode - 26 : m.is_hor ? -(mode - 10) : 0;
	m.inv_angle = 0;
	if (mode >= 2) {
		const int aa = m.angle < 0 ? -m.angle : m.angle;
		m.inv_angle = intra_inv_ang(aa);
		m.angle = m.angle < 0 ? -intra_ang(aa) : intra_ang(aa);
	}
	m.sm = m.is_ver ? 1 : -1;
	return m;
}

// main reference of an angular mode, built in closed form (no serial running sum); the caller synchronises afterwards
template <int N, int G>
__device__ __forceinline__ void intra_fill_main(const IntraMode &m, const int16_t *mid, int16_t *mainr, int l)
{
	if (m.mode < 2) return;
	for (int idx = l; idx <= 2 * N; idx += G) mainr[idx] = mid[m.sm * idx];
	if (m.angle < 0) {
		const int last = (N * m.angle) >> 5;   // projected entries idx = -1 .. last+1
		for (int t = 1 + l; -t > last; t += G) mainr[-t] = mid[-m.sm * ((128 + t * m.inv_angle) >> 8)];
	}
}

// DC value; every lane of the group calls it (shuffle reduction)
template <int N, int G>
__device__ __forceinline__ int intra_dc(const int16_t *mid, int l, bool active)
{
	int s = 0;
	if (active)
		for (int i = 1 + l; i <= N; i += G) s += mid[i] + mid[-i];
	s = group_sum<G>(s);
	return ((s + N) / (2 * N)) & 0xff;
}

template <int N>
__device__ __forceinline__ int intra_pixel(const IntraMode &m, const int16_t *mid, const int16_t *mainr, int dc, bool edge, int x, int y)
{
	constexpr int l2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
	if (m.mode == 0) {
		const int left = mid[-(y + 1)], top = mid[x + 1], bl = mid[-(N + 1)], tr = mid[N + 1];
		return ((N - 1 - x) * left + (x + 1) * tr + (N - 1 - y) * top + (y + 1) * bl + N) >> (l2 + 1);
	}
	if (m.mode == 1) {
		int v = dc;
		if (edge) {
			if (x == 0 && y == 0) v = (mid[-1] + mid[1] + 2 * dc + 2) >> 2;
			else if (y == 0) v = (mid[1 + x] + 3 * dc + 2) >> 2;
			else if (x == 0) v = (mid[-1 - y] + 3 * dc + 2) >> 2;
		}
		return v;
	}
	// (line, pos) in the mode's own orientation: vertical modes line = row, horizontal modes line = column
	const int line = m.is_ver ? y : x, i = m.is_ver ? x : y;
	if (m.angle == 0) {
		int v = mainr[i + 1] & 0xff;
		if (edge && i == 0) v = clip3i(v + ((mid[-m.sm * (line + 1)] - mid[0]) >> 1), 0, 255);
		return v;
	}
	const int pos = (line + 1) * m.angle, delta = pos >> 5, fract = pos & 31, idx = i + delta + 1;
	return fract ? (((32 - fract) * mainr[idx] + fract * mainr[idx + 1] + 16) >> 5) & 0xff : mainr[idx] & 0xff;
}

// Neighbour array from the reconstructed plane; d addresses the corner sample (-1,-1).  Flags must come from the partition tree
// (bottom_left implies left, top_right implies top).  Closed-form availability / substitution per entry.
template <int N, int G>
__device__ __forceinline__ void intra_build_refs(int16_t *adi, const int16_t *d, int st, bool left, bool top, int bl_size, int tr_size, int l)
{
	constexpr int total = 4 * N + 1;
	// substitution samples (hmr_motion_intra.c:277,301,324-338)
	int first_sample = 128, last_sample = 128;
	if (left) first_sample = d[(size_t)(N + bl_size) * st];          // lowest available left / bottom-left sample
	else if (top) first_sample = d[1];                               // top[0]
	if (top) last_sample = d[N + tr_size];                           // right-most available top / top-right sample
	else if (left) last_sample = d[(size_t)st];                      // top of the left column
	for (int i = l; i < total; i += G) {
		int v;
		if (!left && !top) v = 128;
		else if (i < N) {                         // bottom-left, adi[N-1-r] = row N+1+r
			const int r = N - 1 - i;
			v = (r < bl_size) ? d[(size_t)(N + 1 + r) * st] : first_sample;
		} else if (i < 2 * N) {                   // left, adi[N+r'] = row N-r'
			v = left ? d[(size_t)(2 * N - i) * st] : first_sample;
		} else if (i == 2 * N) {
			v = (left && top) ? d[0] : (left ? last_sample : first_sample);
		} else if (i <= 3 * N) {
			v = top ? d[i - 2 * N] : last_sample;
		} else {
			v = (i - 3 * N - 1 < tr_size) ? d[i - 2 * N] : last_sample;
		}
		adi[i] = (int16_t)v;
	}
}

// [1 2 1] / 4 smoothing, or the strong bilinear filter for N >= 32 when enabled and both edges are flat (adi_filter, :189)
template <int N, int G>
__device__ __forceinline__ void intra_filter_refs(const int16_t *adi, int16_t *f, bool strong_enabled, int l)
{
	constexpr int total = 4 * N + 1;
	const int bls = adi[0], tl = adi[2 * N], trs = adi[total - 1];
	bool strong = false;
	if (strong_enabled && N >= 32) {
		const int dl = bls + tl - 2 * adi[N], dt = tl + trs - 2 * adi[3 * N];
		strong = (dl < 0 ? -dl : dl) < 8 && (dt < 0 ? -dt : dt) < 8;
	}
	constexpr int l2n = N == 32 ? 6 : 7;   // log2(2N) for the sizes that reach the strong branch
	for (int i = l; i < total; i += G) {
		int v;
		if (i == 0 || i == total - 1 || (strong && i == 2 * N)) v = adi[i];
		else if (strong) {
			v = i < 2 * N ? ((2 * N - i) * bls + i * tl + N) >> l2n : ((4 * N - i) * tl + (i - 2 * N) * trs + N) >> l2n;
		} else v = (adi[i - 1] + 2 * adi[i] + adi[i + 1] + 2) >> 2;
		f[i] = (int16_t)v;
	}
}

}  // namespace
