// Motion search driver and motion compensation.
// Reference semantics: hmr_motion_estimation hmr_motion_inter.c:1404-1775 (integer diamond search with AMVP-relative
// vector cost, 9-point half- and 9-point quarter-sample refinement on SAD), the sub-pel plane recipes
// hmr_half/quarter_pixel_estimation_luma_hm :395,442, and hmr_motion_compensation_luma/chroma :1779,1860.
//
// One wave runs the whole search of one PU: the source block is staged in LDS once, every candidate SAD is a
// wave-wide reduction, and the search state (best vector, restart arc, ring radius) is wave-uniform scalar code, so the
// strict-'<' tie-breaking order of the reference is reproduced exactly.  The reference materialises 16 sub-pel planes
// per PU in memory (18 interpolation calls); here the horizontal first stage of the (at most three) candidate
// columns is kept in LDS tiles and each candidate's vertical stage feeds the SAD directly - the planes never exist
// in HBM.  The vector cost is the reference's double arithmetic with the host-supplied factor
// calc_mv_correction(qp, avg_dist) (hmr_common.h:53); the build uses -ffp-contract=off so no FMA is formed.
#include "common.h"
#include "vec.h"

namespace {

__constant__ int16_t cLumaTaps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int16_t cChromaTaps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
					  {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
__constant__ int cDs[4][2] = {{-1, 0}, {0, -1}, {1, 0}, {0, 1}};
__constant__ int cDb[8][2] = {{-2, 0}, {-1, -1}, {0, -2}, {1, -1}, {2, 0}, {1, 1}, {0, 2}, {-1, 1}};

// Packed-sample arithmetic shared by the search driver and motion compensation: samples stay two per register as loaded and
// every multiply-accumulate is a v_dot2_i32_i16 against a coefficient pair; sums are exact in 32 bits.
typedef short mc_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mc_dot2(int a, int b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_short2, a), __builtin_bit_cast(mc_short2, b), c, false); }
__device__ __forceinline__ int mc_pair(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
struct i32x2 { int v[2]; };
__device__ __forceinline__ i32x2 ld2w(const int16_t *p)
{
	i32x2 r;
	__builtin_memcpy(&r, p, 8);
	return r;
}

// four horizontal outputs at p[TAPS/2-1 ...]: p addresses the first sample of the footprint
template <int TAPS>
__device__ __forceinline__ void mc_hor4(const int16_t *p, const int (&c)[TAPS], int (&out)[4])
{
	if constexpr (TAPS == 8) {
		const i32x2 a = ld2w(p), b = ld2w(p + 4), d = ld2w(p + 7);          // s0..s3, s4..s7, s7..s10 (no sample beyond the footprint)
		const int r0 = a.v[0], r1 = a.v[1], r2 = b.v[0], r3 = b.v[1], q78 = d.v[0], q9a = d.v[1];
		const int r89 = (int)__builtin_amdgcn_alignbit((unsigned)q9a, (unsigned)q78, 16);   // (s8, s9)
		const int c01 = mc_pair(c[0], c[1]), c23 = mc_pair(c[2], c[3]), c45 = mc_pair(c[4], c[5]), c67 = mc_pair(c[6], c[7]);
		const int c12 = mc_pair(c[1], c[2]), c34 = mc_pair(c[3], c[4]), c56 = mc_pair(c[5], c[6]), z0 = c[0] << 16, z7 = c[7] << 16, c7l = c[7] & 0xffff;
		out[0] = mc_dot2(r3, c67, mc_dot2(r2, c45, mc_dot2(r1, c23, mc_dot2(r0, c01, 0))));
		out[1] = mc_dot2(q78, z7, mc_dot2(r3, c56, mc_dot2(r2, c34, mc_dot2(r1, c12, mc_dot2(r0, z0, 0)))));
		out[2] = mc_dot2(r89, c67, mc_dot2(r3, c45, mc_dot2(r2, c23, mc_dot2(r1, c01, 0))));
		out[3] = mc_dot2(q9a, z7, mc_dot2(r89, c56, mc_dot2(r3, c34, mc_dot2(r2, c12, mc_dot2(r1, z0, 0)))));
		(void)c7l;
	} else {
		const i32x2 a = ld2w(p), d = ld2w(p + 3);                           // s0..s3, s3..s6
		const int r0 = a.v[0], r1 = a.v[1], q34 = d.v[0], q56 = d.v[1];
		const int r45 = (int)__builtin_amdgcn_alignbit((unsigned)q56, (unsigned)q34, 16);   // (s4, s5)
		const int c01 = mc_pair(c[0], c[1]), c23 = mc_pair(c[2], c[3]), c12 = mc_pair(c[1], c[2]), z0 = c[0] << 16, z3 = c[3] << 16;
		out[0] = mc_dot2(r1, c23, mc_dot2(r0, c01, 0));
		out[1] = mc_dot2(q34, z3, mc_dot2(r1, c12, mc_dot2(r0, z0, 0)));
		out[2] = mc_dot2(r45, c23, mc_dot2(r1, c01, 0));
		out[3] = mc_dot2(q56, c23, mc_dot2(q34, c01, 0));
	}
}

// four vertical outputs of one row: p addresses the first tap row, `step` elements between rows
template <int TAPS>
__device__ __forceinline__ void mc_ver4(const int16_t *p, int step, const int (&c)[TAPS], int (&out)[4])
{
	out[0] = out[1] = out[2] = out[3] = 0;
#pragma unroll
	for (int k = 0; k < TAPS; k++) {
		const i32x2 v = ld2w(p + (ptrdiff_t)k * step);
		const int lo = c[k] & 0xffff, hi = c[k] << 16;
		out[0] = mc_dot2(v.v[0], lo, out[0]);
		out[1] = mc_dot2(v.v[0], hi, out[1]);
		out[2] = mc_dot2(v.v[1], lo, out[2]);
		out[3] = mc_dot2(v.v[1], hi, out[3]);
	}
}

// |a0 - b0| + |a1 - b1| + acc on packed unsigned 16-bit pairs (picture samples are 0..255)
__device__ __forceinline__ uint32_t sad2(int a, int b, uint32_t acc) { return __builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, acc); }

// luma_filter_coeffs (hmr_motion_inter.c:240-246) and the nine-point refinement patterns as immediates, for lanes that work on
// different candidates at the same time (a table in memory would be a divergent load)
__device__ __forceinline__ void luma_taps(int f, int (&c)[8])
{
	// one byte per tap, taps 0..3 in the low word
	const unsigned lo = f == 0 ? 0x40000000u : f == 1 ? 0x3af604ffu : f == 2 ? 0x28f504ffu : 0x11fb0100u;
	const unsigned hi = f == 0 ? 0x00000000u : f == 1 ? 0x0001fb11u : f == 2 ? 0xff04f528u : 0xff04f63au;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		c[k] = (int)(lo << (24 - 8 * k)) >> 24;
		c[4 + k] = (int)(hi << (24 - 8 * k)) >> 24;
	}
}
// i-th refinement point (x, y) in -1..1: half-pel order (hmr_motion_inter.c:1693) and quarter-pel order (:1738)
__device__ __forceinline__ int ref_pt(unsigned packed, int i) { return (int)((packed >> (2 * i)) & 3u) - 1; }
constexpr unsigned kRefX = 1u | (1u << 2) | (1u << 4) | (0u << 6) | (2u << 8) | (0u << 10) | (2u << 12) | (0u << 14) | (2u << 16);    // 0 0 0 -1 1 -1 1 -1 1
constexpr unsigned kRefHY = 1u | (0u << 2) | (2u << 4) | (1u << 6) | (1u << 8) | (0u << 10) | (0u << 12) | (2u << 14) | (2u << 16);   // 0 -1 1 0 0 -1 -1 1 1
constexpr unsigned kRefQY = 1u | (0u << 2) | (2u << 4) | (0u << 6) | (0u << 8) | (1u << 10) | (1u << 12) | (2u << 14) | (2u << 16);   // 0 -1 1 -1 -1 0 0 1 1

template <int N> struct MeGeo {
	static constexpr int WPB = N == 64 ? 1 : 4;          // waves per workgroup (LDS budget: 36 KB per wave at N = 64)
	static constexpr int TR = N + 8;                     // tile rows: reference rows best_y-4 .. best_y+N+3
};

// wave-uniform vector cost (select_mv_candidate_fast, hmr_motion_inter.c:1004)
__device__ __forceinline__ uint32_t mv_cost(const hmr_gpu_me_job &jb, int mvx, int mvy)
{
	uint32_t best = 0x7fffffffu;
	for (int i = 0; i < jb.n_amvp; i++) {
		const int dx = jb.amvp[i][0] - mvx, dy = jb.amvp[i][1] - mvy;
		const double cx = jb.corr * (double)(float)(dx < 0 ? -dx : dx), cy = jb.corr * (double)(float)(dy < 0 ? -dy : dy);
		const uint32_t c = (uint32_t)(cx + cy + .5);
		if (best > c) best = c;
	}
	return best;
}

template <int N>
__global__ __launch_bounds__(MeGeo<N>::WPB * 64) void k_motion_estimation(const hmr_gpu_me_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ O,
									     const int16_t *__restrict__ R, int range_x, int range_y, int frame_w, int frame_h,
									     hmr_gpu_me_result *__restrict__ out)
{
	using g = MeGeo<N>;
	constexpr int CH = N * N / 4, CPR = N / 4;
	__shared__ __attribute__((aligned(16))) int16_t sOrig[g::WPB][N * N];
	__shared__ __attribute__((aligned(16))) int16_t sTile[g::WPB][3][g::TR * N];
	const int lane = lane_id(), w = threadIdx.x >> 6;
	int16_t *orig = sOrig[w];
	const JobRange jr = xcd_job_range(njobs, g::WPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w;
		if (j >= jr.end) continue;       // whole wave leaves together; no block-level barrier is used below
		const hmr_gpu_me_job jb = jobs[j];
		const int16_t *ref = R + jb.ref_off;
		const int rs = (int)jb.ref_stride;
		{
			const int16_t *o = O + jb.orig_off;
			for (int e = lane; e < CH; e += 64) {
				const int y = e / CPR, x = (e % CPR) * 4;
				st4(orig + y * N + x, ld4(o + (size_t)y * jb.orig_stride + x));
			}
		}
		wave_sync();
		const int gx = jb.gx, gy = jb.gy;
		const int xlow = (gx - range_x) < 0 ? -gx : -range_x, xhigh = (gx + range_x) > (frame_w - N) ? frame_w - gx - N : range_x;
		const int ylow = (gy - range_y) < 0 ? -gy : -range_y, yhigh = (gy + range_y) > (frame_h - N) ? frame_h - gy - N : range_y;

		auto sad_at = [&](int x, int y) -> uint32_t {
			const int16_t *p = ref + (ptrdiff_t)y * rs + x;
			uint32_t acc = 0;
#pragma unroll 4
			for (int e = lane; e < CH; e += 64) {
				const int yy = e / CPR, xx = (e % CPR) * 4;
				const i32x2 a = ld2w(orig + yy * N + xx), b = ld2w(p + (size_t)yy * rs + xx);
				acc = sad2(a.v[1], b.v[1], sad2(a.v[0], b.v[0], acc));
			}
			return wave_sum(acc);
		};

		uint32_t cur_sad = 0, cur_rd = 0;
		int cur_x = 0, cur_y = 0, best_x = 0, best_y = 0, mvx, mvy, subx = 0, suby = 0;
		const unsigned action = jb.action;
		if (action & 1) {
			int next_start, search_size;
			bool better;
#define TRY(x_, y_)                                                                                          \
	better = false;                                                                                      \
	if ((x_) >= xlow && (x_) <= xhigh && (y_) >= ylow && (y_) <= yhigh) {                                \
		const uint32_t s_ = sad_at((x_), (y_));                                                      \
		const uint32_t rd_ = s_ + mv_cost(jb, (x_) << 2, (y_) << 2);                                 \
		if (rd_ < cur_rd) { cur_sad = s_; cur_rd = rd_; cur_x = (x_); cur_y = (y_); better = true; } \
	}
			cur_x = clip3i(jb.init_x, xlow, xhigh);
			cur_y = clip3i(jb.init_y, ylow, yhigh);
			cur_sad = sad_at(cur_x, cur_y);
			cur_rd = cur_sad + mv_cost(jb, cur_x << 2, cur_y << 2);
			uint32_t best_sad = cur_sad;
			best_x = cur_x; best_y = cur_y;
			bool skip = best_sad == 0;
			if (!skip) {
				for (int i = 0; i < jb.n_search; i++) {
					const int x = jb.search[i][0] >> 2, y = jb.search[i][1] >> 2;
					if (x == 0 && y == 0) continue;
					TRY(x, y)
				}
				best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
				skip = best_sad == 0;
			}
			if (!skip) {
				for (int i = 0; i < 4; i++) {
					const int x = best_x + cDs[i][0], y = best_y + cDs[i][1];
					TRY(x, y)
				}
				skip = best_sad == 0;   // (best_sad is not refreshed after the small diamond, as in the reference)
			}
			if (!skip) {
				int dist = 2;
				const int end = (best_x != 0 && best_y != 0) ? 4 : 8;
				next_start = 0; search_size = 8;
				best_x = cur_x; best_y = cur_y;
				while (dist < end) {
					for (int i = next_start; i < next_start + search_size; i++) {
						const int idx = i % 8, x = best_x + cDb[idx][0] * dist, y = best_y + cDb[idx][1] * dist;
						TRY(x, y)
						if (better) { next_start = (idx - 2 + 8) % 8; search_size = 5; }
					}
					dist *= 2;
				}
			}
			best_x = cur_x; best_y = cur_y;
			next_start = 0; search_size = 4;
			for (;;) {
				for (int i = next_start; i < next_start + search_size; i++) {
					const int idx = i % 4, x = best_x + cDs[idx][0], y = best_y + cDs[idx][1];
					TRY(x, y)
					if (better) { next_start = (idx - 1 + 4) % 4; search_size = 3; }
				}
				if (best_x == cur_x && best_y == cur_y) break;
				best_x = cur_x; best_y = cur_y;
			}
#undef TRY
			mvx = best_x << 2; mvy = best_y << 2;
		} else {
			mvx = jb.init_x << 2; mvy = jb.init_y << 2;
		}
		uint32_t best_sad = cur_sad;
		if (action & 2) {
			best_x = mvx >> 2; best_y = mvy >> 2;
			if (!(action & 1)) cur_sad = sad_at(best_x, best_y);
			// first-stage tile t holds column offset cxs[t] (quarter samples): rows best_y-4 .. best_y+N+3
			auto build_tiles = [&](int cx0, int cx1, int cx2) {
				const int cxs[3] = {cx0, cx1, cx2};
				for (int t = 0; t < 3; t++) {
					const int qx = (best_x << 2) + cxs[t], ix = qx >> 2, fx = qx & 3;
					const int16_t *p0 = ref + (ptrdiff_t)(best_y - 4) * rs + ix - 3;
					int c[8];
#pragma unroll
					for (int k = 0; k < 8; k++) c[k] = cLumaTaps[fx][k];
					if (fx == 0) {       // integer column: the filter is the single tap 64 on the centre sample - 64 * x - 8192 without the seven zero products
						for (int e = lane; e < g::TR * CPR; e += 64) {
							const int r = e / CPR, x = (e % CPR) * 4;
							const i16x4 v = ld4(p0 + (size_t)r * rs + x + 3);
							i16x4 o;
#pragma unroll
							for (int k = 0; k < 4; k++) o.v[k] = (int16_t)(64 * v.v[k] - 8192);
							st4(&sTile[w][t][r * N + x], o);
						}
						continue;
					}
					for (int e = lane; e < g::TR * CPR; e += 64) {
						const int r = e / CPR, x = (e % CPR) * 4;
						int sm[4];
						mc_hor4<8>(p0 + (size_t)r * rs + x, c, sm);
						i16x4 o;
#pragma unroll
						for (int k = 0; k < 4; k++) o.v[k] = (int16_t)(sm[k] - 8192);   // first, not last: shift 0, offset -8192 (for fx = 0: 64*x - 8192)
						st4(&sTile[w][t][r * N + x], o);
					}
				}
				wave_sync();
			};
			// one four-sample item of candidate (tile t, vertical quarter offset cy)
			auto item_sub = [&](int t, int cy, int e) -> uint32_t {
				const int qy = (best_y << 2) + cy, iy = (qy >> 2) - best_y, fy = qy & 3;   // iy in {-1, 0}
				int c[8];
				luma_taps(fy, c);
				const int16_t *tile = &sTile[w][t][(4 + iy - 3) * N];
				const int y = e / CPR, x = (e % CPR) * 4;
				int sm[4];
				if (fy == 0) {          // integer row: single tap 64 on the centre row
					const i16x4 v = ld4(tile + (y + 3) * N + x);
#pragma unroll
					for (int k = 0; k < 4; k++) sm[k] = 64 * v.v[k];
				} else
					mc_ver4<8>(tile + y * N + x, N, c, sm);
				int pv[4];
#pragma unroll
				for (int k = 0; k < 4; k++) pv[k] = clip3i(sat16i((sm[k] + 2048 + (8192 << 6)) >> 12), 0, 255);   // not first, last
				const i32x2 a = ld2w(orig + y * N + x);
				return sad2(a.v[1], pv[2] | (pv[3] << 16), sad2(a.v[0], pv[0] | (pv[1] << 16), 0));
			};
			// SADs of the nine candidates of a refinement round; candidate i = (tile px(i) + 1, vertical offset oy + sy * py(i)).
			// A block of fewer than 64 items (8x8) is evaluated 64 / CH candidates at a time, each by its own lane group.
			auto round9 = [&](unsigned ypat, int oy, int sy, uint32_t (&sd)[9]) {
				if constexpr (CH >= HMR_WAVE) {
#pragma unroll
					for (int i = 0; i < 9; i++) {
						uint32_t acc = 0;
						for (int e = lane; e < CH; e += HMR_WAVE) acc += item_sub(ref_pt(kRefX, i) + 1, oy + sy * ref_pt(ypat, i), e);
						sd[i] = wave_sum(acc);
					}
				} else {
					constexpr int PAR = HMR_WAVE / CH;
#pragma unroll
					for (int b = 0; b < (9 + PAR - 1) / PAR; b++) {
						const int i = b * PAR + lane / CH, ii = i < 9 ? i : 8;
						uint32_t acc = i < 9 ? item_sub(ref_pt(kRefX, ii) + 1, oy + sy * ref_pt(ypat, ii), lane % CH) : 0u;
						acc = group_sum<CH>(acc);
#pragma unroll
						for (int q = 0; q < PAR; q++)
							if (b * PAR + q < 9) sd[b * PAR + q] = (uint32_t)__builtin_amdgcn_readlane((int)acc, q * CH);
					}
				}
			};
			uint32_t sd[9];
			int bidx = 0, bx = 0, by = 0;
			build_tiles(-2, 0, 2);
			round9(kRefHY, 0, 2, sd);
#pragma unroll
			for (int i = 0; i < 9; i++)
				if (sd[i] < cur_sad) { cur_sad = sd[i]; bx = ref_pt(kRefX, i) * 2; by = ref_pt(kRefHY, i) * 2; bidx = i; }
			mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
			best_sad = cur_sad;
			if (action & 4) {
				const int hx = ref_pt(kRefX, bidx), hy = ref_pt(kRefHY, bidx);
				wave_sync();
				build_tiles(hx * 2 - 1, hx * 2, hx * 2 + 1);
				bx = hx * 2; by = hy * 2;
				round9(kRefQY, hy * 2, 1, sd);
#pragma unroll
				for (int i = 0; i < 9; i++)
					if (sd[i] < cur_sad) { cur_sad = sd[i]; bx = hx * 2 + ref_pt(kRefX, i); by = hy * 2 + ref_pt(kRefQY, i); }
				best_sad = cur_sad;
				mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
			}
		} else {
			best_sad = cur_sad;
		}
		if (lane == 0) {
			hmr_gpu_me_result r;
			r.mvx = mvx; r.mvy = mvy; r.subx = subx; r.suby = suby; r.sad = best_sad;
			out[j] = r;
		}
		wave_sync();
	}
}

// Motion compensation: p0 = mv.x, p1 = mv.y (quarter samples for luma, eighth samples for chroma), w/h extent,
// a = co-located block in the reference, c = prediction.  `lanes_per_job` lanes share a job (small chroma blocks fill a
// wave together); two-stage vectors keep the first stage in the job's share of an LDS tile.
//
// A work item is four horizontally adjacent outputs.  Samples stay packed two per register as they are loaded (8-byte
// loads at 2-byte alignment) and every multiply-accumulate is a v_dot2_i32_i16 against a packed coefficient pair:
//   horizontal: the TAPS+3 samples of the footprint in 2 / 3 loads; even outputs pair (c[k], c[k+1]) with aligned sample
//               pairs, odd outputs use the same registers with the pairs (0, c0), (c1, c2), ... - no unpacking, no shuffles
//               beyond one v_alignbit for the pair that straddles the overlapped last load;
//   vertical  : one 8-byte load per tap row; each register (two columns) is hit with (c, 0) and (0, c).
// The sums are exact in 32 bits, so the result is that of the scalar definition (hmr_motion_inter.c:262-391).
template <int TAPS>
__global__ __launch_bounds__(HMR_BLOCK) void k_mc(const hmr_gpu_job *__restrict__ jobs, int njobs, int is_bi, int lanes_per_job, const int16_t *__restrict__ A,
						     int16_t *__restrict__ Cc)
{
	constexpr int FM = TAPS == 8 ? 3 : 7, FS = TAPS == 8 ? 2 : 3, HT = TAPS / 2 - 1;
	// first-stage tile: outputs are produced in tiles of at most 32 x 32, so a wave needs (32 + TAPS - 1) x 32 intermediates however
	// large the block is - a small tile keeps eight waves per SIMD resident
	constexpr int TS = 32, TILE = (TS + TAPS - 1) * TS;
	__shared__ __attribute__((aligned(16))) int16_t sTile[HMR_WAVES_PER_BLOCK][TILE];
	const int G = lanes_per_job, JPW = HMR_WAVE / G, share = TILE / JPW;
	const int sub = lane_id() / G, lane = lane_id() % G, w = wave_in_block();
	const bool last = !is_bi;
	int16_t *tile = sTile[w] + sub * share;
	const JobRange jr = xcd_job_range(njobs, JPW * HMR_WAVES_PER_BLOCK);
	for (long j0 = jr.begin + w * JPW; j0 < jr.end; j0 += jr.stride) {
		const long j = j0 + sub;
		if (j >= jr.end) continue;
		const hmr_gpu_job jb = JPW == 1 ? load_job_uniform(jobs, j) : jobs[j];
		const int bw = jb.w, bh = jb.h, mvx = (int)jb.p0, mvy = (int)jb.p1;
		const int xf = mvx & FM, yf = mvy & FM, rs = (int)jb.a_stride, ds = (int)jb.c_stride;
		const int16_t *src = A + jb.a_off + (ptrdiff_t)(mvy >> FS) * rs + (mvx >> FS);
		int16_t *dst = Cc + jb.c_off;
		int cx[TAPS], cy[TAPS];
#pragma unroll
		for (int k = 0; k < TAPS; k++) {
			cx[k] = TAPS == 8 ? cLumaTaps[xf][k] : cChromaTaps[xf][k];
			cy[k] = TAPS == 8 ? cLumaTaps[yf][k] : cChromaTaps[yf][k];
		}
		const bool vec = (bw & 3) == 0;       // every HEVC PU except 2-wide chroma
		const int cpr = bw >> 2, lcpr = __ffs(cpr) - 1;   // four-output items per row (a power of two: PU widths are)
		if (xf == 0 || yf == 0) {
			// one stage, first: (sum + 32) >> 6 clipped when last, sum - 8192 when feeding a bi-prediction average
			const bool vert = xf == 0;
			const int f = vert ? yf : xf;
			if (TAPS == 4 && f == 0 && bw < 4) continue;   // hmr_sse42_functions_inter_prediction.c:822
			if (vec) {
				for (int e = lane; e < cpr * bh; e += G) {
					const int y = e >> lcpr, x = (e & (cpr - 1)) * 4;
					int o[4];
					if (f == 0) {
						const i16x4 v = ld4(src + (size_t)y * rs + x);
#pragma unroll
						for (int q = 0; q < 4; q++) o[q] = last ? v.v[q] : (int16_t)((int16_t)(v.v[q] << 6) - 8192);
					} else {
						if (vert) mc_ver4<TAPS>(src + (ptrdiff_t)(y - HT) * rs + x, rs, cy, o);
						else mc_hor4<TAPS>(src + (size_t)y * rs + x - HT, cx, o);
#pragma unroll
						for (int q = 0; q < 4; q++) o[q] = last ? clip3i(sat16i((o[q] + 32) >> 6), 0, 255) : sat16i(o[q] - 8192);
					}
					i16x4 r;
#pragma unroll
					for (int q = 0; q < 4; q++) r.v[q] = (int16_t)o[q];
					st4(dst + (size_t)y * ds + x, r);
				}
			} else {
				const int step = vert ? rs : 1;
				const int16_t *s0 = src - HT * step;
				for (int e = lane; e < bw * bh; e += G) {
					const int y = e / bw, x = e - y * bw;
					int v;
					if (f == 0) {
						const int p = src[(size_t)y * rs + x];
						v = last ? p : (int16_t)((int16_t)(p << 6) - 8192);
					} else {
						int sm = 0;
#pragma unroll
						for (int k = 0; k < TAPS; k++) sm += s0[(size_t)y * rs + x + k * step] * (vert ? cy[k] : cx[k]);
						v = last ? clip3i(sat16i((sm + 32) >> 6), 0, 255) : sat16i(sm - 8192);
					}
					dst[(size_t)y * ds + x] = (int16_t)v;
				}
			}
			continue;
		}
		const int tw = bw < TS ? bw : TS, tb = bh < TS ? bh : TS;      // output tile
		if (vec && tw * (tb + TAPS - 1) <= share) {
			const int tcpr = tw >> 2, ltcpr = __ffs(tcpr) - 1;
			for (int ty = 0; ty < bh; ty += tb)
				for (int tx = 0; tx < bw; tx += tw) {
					const int16_t *ts = src + (ptrdiff_t)ty * rs + tx;
					for (int e = lane; e < tcpr * (tb + TAPS - 1); e += G) {
						const int y = e >> ltcpr, x = (e & (tcpr - 1)) * 4;
						int o[4];
						mc_hor4<TAPS>(ts + (ptrdiff_t)(y - HT) * rs + x - HT, cx, o);
						i16x4 r;
#pragma unroll
						for (int q = 0; q < 4; q++) r.v[q] = (int16_t)sat16i(o[q] - 8192);
						st4(tile + y * tw + x, r);
					}
					wave_sync();
					for (int e = lane; e < tcpr * tb; e += G) {
						const int y = e >> ltcpr, x = (e & (tcpr - 1)) * 4;
						int o[4];
						mc_ver4<TAPS>(tile + y * tw + x, tw, cy, o);
						i16x4 r;
#pragma unroll
						for (int q = 0; q < 4; q++) r.v[q] = (int16_t)(last ? clip3i(sat16i((o[q] + 2048 + (8192 << 6)) >> 12), 0, 255) : sat16i(o[q] >> 6));
						st4(dst + (size_t)(ty + y) * ds + tx + x, r);
					}
					wave_sync();
				}
		} else {
			// 2-wide chroma, or the job does not fit its LDS share (hint too small for this block): first stage recomputed per output
			for (int e = lane; e < bw * bh; e += G) {
				const int y = e / bw, x = e - y * bw;
				int s2 = 0;
				for (int r = 0; r < TAPS; r++) {
					const int16_t *p = src + (ptrdiff_t)(y + r - HT) * rs + x - HT;
					int sm = 0;
#pragma unroll
					for (int k = 0; k < TAPS; k++) sm += p[k] * cx[k];
					s2 += sat16i(sm - 8192) * cy[r];
				}
				const int v = last ? clip3i(sat16i((s2 + 2048 + (8192 << 6)) >> 12), 0, 255) : sat16i(s2 >> 6);
				dst[(size_t)y * ds + x] = (int16_t)v;
			}
		}
	}
}

}  // namespace

extern "C" int hmr_gpu_motion_estimation_batch(hmr_gpu_ctx *ctx, const hmr_gpu_me_job *jobs, int njobs, int size, const int16_t *orig_base,
						const int16_t *ref_base, int range_x, int range_y, int frame_w, int frame_h, hmr_gpu_me_result *out)
{
	if (njobs <= 0) return HMR_GPU_OK;
#define ME_LAUNCH(N)                                                                                                                       \
	hipLaunchKernelGGL((k_motion_estimation<N>), dim3(hmr_grid_for_units(((long)njobs + MeGeo<N>::WPB - 1) / MeGeo<N>::WPB)), dim3(MeGeo<N>::WPB * 64), 0, \
			   ctx->stream, jobs, njobs, orig_base, ref_base, range_x, range_y, frame_w, frame_h, out)
	switch (size) {
	case 8: ME_LAUNCH(8); break;
	case 16: ME_LAUNCH(16); break;
	case 32: ME_LAUNCH(32); break;
	case 64: ME_LAUNCH(64); break;
	default: hmr_set_error("motion_estimation_batch: PU size must be 8, 16, 32 or 64"); return HMR_GPU_ERR_ARG;
	}
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_mc_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int flags, int is_bi, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	const int is_luma = flags & 1;
	int g = (flags >> 8) & 0xff;   // lanes per job hint: 4 / 8 / 16 / 32 / 64 (0 = 64)
	if (g != 4 && g != 8 && g != 16 && g != 32) g = 64;
	const int jpw = HMR_WAVE / g;
	dim3 grid(hmr_grid_for_waves(((long)njobs + jpw - 1) / jpw)), block(HMR_BLOCK);
	if (is_luma) hipLaunchKernelGGL((k_mc<8>), grid, block, 0, ctx->stream, jobs, njobs, is_bi, g, a, c);
	else hipLaunchKernelGGL((k_mc<4>), grid, block, 0, ctx->stream, jobs, njobs, is_bi, g, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
