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

// The same decision with ONE lane per coefficient group (the fused TU chain: a wave holds 4N groups, so every group of every TU
// of the wave is decided at once instead of group after group).  The reference's walk "n from start down to 0, keep the strictly
// smallest cost" (hmr_quant.c:101-160) is taken literally; levels are read once into registers, deltaU and the source signs only
// by the lanes whose group actually has to change a level.
__device__ __forceinline__ void sbh_group_serial(int16_t *dst, const int16_t *src, const int16_t *du, const uint32_t *scan, int cg, bool is_last_cg)
{
	unsigned pos[16];
	int lv[16];
#pragma unroll
	for (int n = 0; n < 16; n++) pos[n] = scan[cg * 16 + n];
	unsigned mask = 0;
	int sum = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) {
		lv[n] = dst[pos[n]];
		mask |= (lv[n] != 0 ? 1u : 0u) << n;
		sum += lv[n];
	}
	if (!mask) return;
	const int last_nz = 31 - __clz((int)mask), first_nz = __ffs((int)mask) - 1;
	int first_val = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) first_val = n == first_nz ? lv[n] : first_val;
	const unsigned signbit = first_val > 0 ? 0u : 1u;
	if (last_nz - first_nz < 4 || signbit == (unsigned)(sum & 1)) return;
	const int start = is_last_cg ? last_nz : 15;
	int min_cost = 0x7fffffff, win_change = 0, win_lv = 0, win_sv = 0;
	unsigned win_pos = 0;
	bool found = false;
#pragma unroll
	for (int n = 15; n >= 0; n--) {
		if (n > start) continue;
		const int d = du[pos[n]], sv = src[pos[n]];
		int cost = 0x7fffffff, change = 0;
		if (lv[n] != 0) {
			if (d > 0) { cost = -d; change = 1; }
			else if (!(n == first_nz && (lv[n] == 1 || lv[n] == -1))) { cost = d; change = -1; }
		} else if (n < first_nz) {
			if ((sv >= 0 ? 0u : 1u) == signbit) { cost = -d; change = 1; }
		} else { cost = -d; change = 1; }
		if (cost < min_cost) {
			min_cost = cost; win_change = change; win_lv = lv[n]; win_sv = sv; win_pos = pos[n];
			found = true;
		}
	}
	if (found) {
		if (win_lv == 32767 || win_lv == -32768) win_change = -1;
		dst[win_pos] = (int16_t)(win_sv >= 0 ? win_lv + win_change : win_lv - win_change);
	}
}

// Same decision, reading the group as the 4x4 block it is.  Inside a group the three scans the encoder uses are fixed 4x4 patterns
// (hmr_tables.c scan construction: horizontal = raster, vertical = column-major, diagonal = up-right); only the group's origin comes
// from the scan table (its first entry).  Four 8-byte LDS reads fetch the levels; deltaU and the source coefficients are fetched
// the same way only when a level has to change.  PAT[n] = 4 * y + x of scan position n.
template <int MODE> struct CgPattern;
template <> struct CgPattern<1> { static constexpr int at(int n) { return n; } };
template <> struct CgPattern<2> { static constexpr int at(int n) { return (n & 3) * 4 + (n >> 2); } };
template <> struct CgPattern<3> {
	static constexpr int at(int n)
	{
		constexpr int t[16] = {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15};
		return t[n];
	}
};

// sample k (0..15, raster inside the group) of four packed rows
__device__ __forceinline__ int cg_get(const int (&w)[8], int k) { return (w[k >> 1] << (16 * (1 - (k & 1)))) >> 16; }
__device__ __forceinline__ void cg_load(int (&w)[8], const int16_t *p, int pitch)
{
#pragma unroll
	for (int r = 0; r < 4; r++) __builtin_memcpy(&w[2 * r], p + r * pitch, 8);
}

template <int MODE, int N>
__device__ __forceinline__ void sbh_group_block(int16_t *dst, const int16_t *src, const int16_t *du, const uint32_t *scan, int cg, bool is_last_cg)
{
	using PT = CgPattern<MODE>;
	const unsigned org = scan[cg * 16];
	int lw[8];           // levels, two per register, raster inside the group
	cg_load(lw, dst + org, N);
	unsigned mask = 0;
	int sum = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) {
		const int lv = cg_get(lw, PT::at(n));
		mask |= (lv != 0 ? 1u : 0u) << n;
		sum += lv;
	}
	if (!mask) return;
	const int last_nz = 31 - __clz((int)mask), first_nz = __ffs((int)mask) - 1;
	int first_val = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) first_val = n == first_nz ? cg_get(lw, PT::at(n)) : first_val;
	const unsigned signbit = first_val > 0 ? 0u : 1u;
	if (last_nz - first_nz < 4 || signbit == (unsigned)(sum & 1)) return;
	int dw[8], sw[8];
	cg_load(dw, du + org, N);
	cg_load(sw, src + org, N);
	const int start = is_last_cg ? last_nz : 15;
	int min_cost = 0x7fffffff, win_change = 0, win_lv = 0, win_sv = 0, win_off = 0;
	bool found = false;
#pragma unroll
	for (int n = 15; n >= 0; n--) {
		if (n > start) continue;
		const int ri = PT::at(n);
		const int lv = cg_get(lw, ri), d = cg_get(dw, ri), sv = cg_get(sw, ri);
		int cost = 0x7fffffff, change = 0;
		if (lv != 0) {
			if (d > 0) { cost = -d; change = 1; }
			else if (!(n == first_nz && (lv == 1 || lv == -1))) { cost = d; change = -1; }
		} else if (n < first_nz) {
			if ((sv >= 0 ? 0u : 1u) == signbit) { cost = -d; change = 1; }
		} else { cost = -d; change = 1; }
		if (cost < min_cost) {
			min_cost = cost; win_change = change; win_lv = lv; win_sv = sv; win_off = (ri >> 2) * N + (ri & 3);
			found = true;
		}
	}
	if (found) {
		if (win_lv == 32767 || win_lv == -32768) win_change = -1;
		dst[org + win_off] = (int16_t)(win_sv >= 0 ? win_lv + win_change : win_lv - win_change);
	}
}

}  // namespace
