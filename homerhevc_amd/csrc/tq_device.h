// Device pieces shared by the transform / quantisation kernels and the fused TU chain.
#pragma once
#include "common.h"

namespace {

template <int N> struct Geo {
	static constexpr int E = N * N;
	// lanes per TU: at least 16 (sign hiding walks 16-coefficient groups with 16 lanes), about 4 coefficients per lane
	static constexpr int G = E / 4 < 16 ? 16 : (E / 4 < HMR_WAVE ? E / 4 : HMR_WAVE);
	static constexpr int JPW = HMR_WAVE / G;                 // TUs per wave
	static constexpr int JPB = JPW * HMR_WAVES_PER_BLOCK;    // TUs per workgroup iteration
	static constexpr int P = N + 2;                          // LDS row pitch (samples)
	static constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
};

template <int N>
__device__ __forceinline__ void load_basis(int16_t (*sM)[N * N], const DevTables *t)
{
	for (int i = threadIdx.x; i < N * N; i += HMR_BLOCK) {
		sM[0][i] = t->dct[Geo<N>::L2 - 2][i];
		sM[1][i] = N == 4 ? t->dst4[i] : (int16_t)0;
	}
}

// Sign-data hiding (hmr_quant.c:61-169) with 16 lanes per coefficient group: lane n owns scan position n of the group.
// The serial "walk n downwards, keep the strictly smallest cost" becomes a min-reduction over the key (cost, 15 - n);
// the only cross-group dependency - which group is the last one holding a level - is resolved beforehand.
// Must be called by all 64 lanes (it shuffles); `active` predicates the update.
__device__ __forceinline__ void sbh_group16(int16_t *dst, const int16_t *src, const int16_t *du, const uint32_t *scan, int cg, bool is_last_cg, bool active)
{
	const int lane = lane_id(), n = lane & 15, gbase = lane & 48;
	const unsigned pos = scan[cg * 16 + n];
	const int lv = active ? dst[pos] : 0, d = active ? du[pos] : 0, sv = active ? src[pos] : 0;
	const unsigned mask = (unsigned)((__ballot(lv != 0) >> gbase) & 0xffffu);
	const int last_nz = mask ? 31 - __clz((int)mask) : -1, first_nz = mask ? __ffs((int)mask) - 1 : 16;
	const int abs_sum = group_sum<16>(lv);
	const int first_val = __shfl(lv, gbase + (first_nz & 15), HMR_WAVE);
	const unsigned signbit = first_val > 0 ? 0u : 1u;
	const bool hide = active && (last_nz - first_nz >= 4) && (signbit != (unsigned)(abs_sum & 1));
	const int start = is_last_cg ? last_nz : 15;
	int cost = 0x7fffffff, change = 0;
	if (n <= start) {
		if (lv != 0) {
			if (d > 0) { cost = -d; change = 1; }
			else if (!(n == first_nz && (lv == 1 || lv == -1))) { cost = d; change = -1; }
		} else if (n < first_nz) {
			if ((sv >= 0 ? 0u : 1u) == signbit) { cost = -d; change = 1; }
		} else { cost = -d; change = 1; }
	}
	// smallest cost wins, ties go to the larger n (the reference meets it first)
	unsigned long long key = ((unsigned long long)(unsigned)(cost ^ 0x80000000) << 8) | (unsigned)(15 - n);
#pragma unroll
	for (int m = 8; m >= 1; m >>= 1) {
		const unsigned lo = __shfl_xor((unsigned)key, m, HMR_WAVE), hi = __shfl_xor((unsigned)(key >> 32), m, HMR_WAVE);
		const unsigned long long other = ((unsigned long long)hi << 32) | lo;
		key = other < key ? other : key;
	}
	const int win_n = 15 - (int)(key & 0xff);
	if (hide && n == win_n && cost != 0x7fffffff) {
		if (lv == 32767 || lv == -32768) change = -1;
		dst[pos] = (int16_t)(sv >= 0 ? lv + change : lv - change);
	}
}

}  // namespace
