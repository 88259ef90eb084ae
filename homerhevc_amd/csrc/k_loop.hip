// Frame-level in-loop filters: deblocking (two passes), SAO statistics, SAO offset, reference border padding.
// Reference semantics: hmr_deblocking_filter.c:737 (+:138,287,478), hmr_sse42_sao.c:35, hmr_sao.c:960,1210,
// hmr_encoder_lib.c:1723.  The reference runs these CTU by CTU in a lagged pipeline (hmr_encoder_lib.c:2386);
// every sample it reads is final when read, so whole-picture passes give identical results and expose the
// picture's full parallelism: these are the HBM-streaming kernels of the path.
//
// Deblock: one thread per 4-sample edge segment.  For vertical edges consecutive lanes own consecutive edges of
// one 4-row band, so each row is read as 16-byte pieces at a 16-byte pitch - a fully coalesced 1 KiB row segment
// per wave; for horizontal edges one lane owns one column and the per-segment decisions are exchanged inside
// 4-lane groups with shuffles.  SAO statistics: one workgroup per (CTU, component), the (w+2) x (h+2) deblocked
// tile staged in LDS, per-lane class counters reduced with wave shuffles, then one LDS atomic per wave.
#include "common.h"

namespace {

constexpr int F_INTRA = 1, F_CBF = 2, F_EDGE_VER = 4, F_EDGE_HOR = 8;

__constant__ uint8_t cTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
				2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t cBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
				  16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t cChromaMid[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int chroma_qp(int q)
{
	q = clip3i(q, 0, 57);
	return q < 30 ? q : (q < 44 ? cChromaMid[q - 30] : q - 6);
}

struct UnitInfo {
	int units_stride;
	const int16_t *mvx, *mvy;
	const int8_t *ref_idx;
	const uint8_t *qp;
	const uint8_t *flags;
};

__device__ __forceinline__ int boundary_strength(const UnitInfo &ui, int q, int p)
{
	const int fq = ui.flags[q], fp = ui.flags[p];
	if ((fq | fp) & F_INTRA) return 2;
	if ((fq | fp) & F_CBF) return 1;
	const int rq = ui.ref_idx[q], rp = ui.ref_idx[p];
	const int mqx = rq < 0 ? 0 : ui.mvx[q], mqy = rq < 0 ? 0 : ui.mvy[q];
	const int mpx = rp < 0 ? 0 : ui.mvx[p], mpy = rp < 0 ? 0 : ui.mvy[p];
	if ((rp < 0 ? -1 : rp) != (rq < 0 ? -1 : rq)) return 1;
	return (iabs(mqx - mpx) >= 4 || iabs(mqy - mpy) >= 4) ? 1 : 0;
}

// units [ux0, ux0 + w4) x [uy0, uy0 + h4) of the picture (the whole picture, or one CTU)
__global__ __launch_bounds__(HMR_BLOCK) void k_edge_flags(const uint8_t *__restrict__ pred_depth, const uint8_t *__restrict__ tr_idx, int ux0, int uy0, int w4, int h4,
							     int units_stride, uint8_t *__restrict__ flags)
{
	const int i = blockIdx.x * HMR_BLOCK + threadIdx.x;
	if (i >= w4 * h4) return;
	const int uy = uy0 + i / w4, ux = ux0 + i % w4, o = uy * units_stride + ux;
	int leaf = 64 >> (pred_depth[o] + tr_idx[o]);
	if (leaf < 8) leaf = 8;
	int f = flags[o] & ~(F_EDGE_VER | F_EDGE_HOR);
	if (ux && (ux * 4) % leaf == 0) f |= F_EDGE_VER;
	if (uy && (uy * 4) % leaf == 0) f |= F_EDGE_HOR;
	flags[o] = (uint8_t)f;
}

// luma filter of one line m[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 (filter_luma, hmr_deblocking_filter.c:287)
__device__ __forceinline__ void luma_line(int (&m)[8], int tc, bool strong, int thr_cut, bool fp, bool fq)
{
	const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7];
	if (strong) {
		m[3] = clip3i((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3, m3 - 2 * tc, m3 + 2 * tc);
		m[4] = clip3i((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3, m4 - 2 * tc, m4 + 2 * tc);
		m[2] = clip3i((m1 + m2 + m3 + m4 + 2) >> 2, m2 - 2 * tc, m2 + 2 * tc);
		m[5] = clip3i((m3 + m4 + m5 + m6 + 2) >> 2, m5 - 2 * tc, m5 + 2 * tc);
		m[1] = clip3i((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - 2 * tc, m1 + 2 * tc);
		m[6] = clip3i((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3, m6 - 2 * tc, m6 + 2 * tc);
	} else {
		int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
		if (iabs(delta) < thr_cut) {
			const int tc2 = tc >> 1;
			delta = clip3i(delta, -tc, tc);
			m[3] = clip3i(m3 + delta, 0, 255);
			m[4] = clip3i(m4 - delta, 0, 255);
			if (fp) m[2] = clip3i(m2 + clip3i((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1, -tc2, tc2), 0, 255);
			if (fq) m[5] = clip3i(m5 + clip3i((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1, -tc2, tc2), 0, 255);
		}
	}
}

__device__ __forceinline__ bool strong_decision(const int (&m)[8], int d, int beta, int tc)
{
	return (iabs(m[0] - m[3]) + iabs(m[7] - m[4]) < (beta >> 3)) && (d < (beta >> 2)) && (iabs(m[3] - m[4]) < ((tc * 5 + 1) >> 1));
}

__device__ __forceinline__ void chroma_edge(int16_t *e, int s, int t, int tc)
{
#pragma unroll
	for (int i = 0; i < 2; i++) {
		int16_t *x = e + i * t;
		const int m4 = x[0], m3 = x[-s], m5 = x[s], m2 = x[-2 * s];
		const int delta = clip3i((((m4 - m3) << 2) + m2 - m5 + 4) >> 3, -tc, tc);
		x[-s] = (int16_t)clip3i(m3 + delta, 0, 255);
		x[0] = (int16_t)clip3i(m4 - delta, 0, 255);
	}
}

struct DeblockArgs {
	int16_t *y, *u, *v;
	int ys, cs, w4, h4;
	int ux0, uy0;        // region of interest in 4x4 units: [ux0, ux0 + w4) x [uy0, uy0 + h4); ux0 / uy0 even (CTU aligned)
	UnitInfo ui;
	int cb_off, cr_off, beta_off, tc_off;
	uint8_t *bs_out;
};

// vertical edges: thread = (unit row uy, even unit column ux)
__global__ __launch_bounds__(HMR_BLOCK) void k_deblock_ver(DeblockArgs a)
{
	const int ew = (a.w4 + 1) / 2;                 // edges per unit row (even ux)
	const int i = blockIdx.x * HMR_BLOCK + threadIdx.x;
	if (i >= ew * a.h4) return;
	const int uy = a.uy0 + i / ew, ux = a.ux0 + (i % ew) * 2;
	const int q = uy * a.ui.units_stride + ux;
	if (a.bs_out) { a.bs_out[q] = 0; if (ux + 1 < a.ux0 + a.w4) a.bs_out[q + 1] = 0; }
	if (!(a.ui.flags[q] & F_EDGE_VER)) return;
	const int p = q - 1;
	const int bs = boundary_strength(a.ui, q, p);
	if (a.bs_out) a.bs_out[q] = (uint8_t)(0x80 | bs);
	if (!bs) return;
	const int qpa = (a.ui.qp[p] + a.ui.qp[q] + 1) >> 1;
	const int tc = cTc[clip3i(qpa + 2 * (bs - 1) + (a.tc_off << 1), 0, 53)];
	const int beta = cBeta[clip3i(qpa + (a.beta_off << 1), 0, 51)];
	int16_t *e = a.y + (size_t)uy * 4 * a.ys + ux * 4;
	int m[4][8];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		// 8 samples p3..q3 start 8-byte aligned (edge on the 8-sample grid, strides are multiples of 8 samples)
		const short4 lo = *reinterpret_cast<const short4 *>(e + (size_t)r * a.ys - 4);
		const short4 hi = *reinterpret_cast<const short4 *>(e + (size_t)r * a.ys);
		m[r][0] = lo.x; m[r][1] = lo.y; m[r][2] = lo.z; m[r][3] = lo.w;
		m[r][4] = hi.x; m[r][5] = hi.y; m[r][6] = hi.z; m[r][7] = hi.w;
	}
	const int dp0 = iabs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = iabs(m[0][4] - 2 * m[0][5] + m[0][6]);
	const int dp3 = iabs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = iabs(m[3][4] - 2 * m[3][5] + m[3][6]);
	const int d0 = dp0 + dq0, d3 = dp3 + dq3;
	if (d0 + d3 < beta) {
		const int side = (beta + (beta >> 1)) >> 3;
		const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
		const bool sw = strong_decision(m[0], 2 * d0, beta, tc) && strong_decision(m[3], 2 * d3, beta, tc);
#pragma unroll
		for (int r = 0; r < 4; r++) {
			luma_line(m[r], tc, sw, tc * 10, fp, fq);
			short4 lo, hi;
			lo.x = (short)m[r][0]; lo.y = (short)m[r][1]; lo.z = (short)m[r][2]; lo.w = (short)m[r][3];
			hi.x = (short)m[r][4]; hi.y = (short)m[r][5]; hi.z = (short)m[r][6]; hi.w = (short)m[r][7];
			*reinterpret_cast<short4 *>(e + (size_t)r * a.ys - 4) = lo;
			*reinterpret_cast<short4 *>(e + (size_t)r * a.ys) = hi;
		}
	}
	if (bs > 1 && (ux & 3) == 0) {
		const size_t co = (size_t)uy * 2 * a.cs + ux * 2;
		chroma_edge(a.u + co, 1, a.cs, cTc[clip3i(chroma_qp(qpa + a.cb_off) + 2 * (bs - 1) + (a.tc_off << 1), 0, 53)]);
		chroma_edge(a.v + co, 1, a.cs, cTc[clip3i(chroma_qp(qpa + a.cr_off) + 2 * (bs - 1) + (a.tc_off << 1), 0, 53)]);
	}
}

// horizontal edges: thread = one luma column of one even unit row; 4-lane groups share the segment decision
__global__ __launch_bounds__(HMR_BLOCK) void k_deblock_hor(DeblockArgs a)
{
	const int width = a.w4 * 4, eh = (a.h4 + 1) / 2;
	const int i = blockIdx.x * HMR_BLOCK + threadIdx.x;
	const bool in = i < width * eh;
	const int er = in ? i / width : 0, x = a.ux0 * 4 + (in ? i - er * width : 0);
	const int uy = a.uy0 + er * 2, ux = x >> 2, col = x & 3;
	const int q = uy * a.ui.units_stride + ux;
	if (in && a.bs_out && col == 0) { a.bs_out[q] = 0; if (uy + 1 < a.uy0 + a.h4) a.bs_out[q + a.ui.units_stride] = 0; }
	const bool edge = in && (a.ui.flags[q] & F_EDGE_HOR);
	int bs = 0, qpa = 0;
	if (edge) {
		const int p = q - a.ui.units_stride;
		bs = boundary_strength(a.ui, q, p);
		qpa = (a.ui.qp[p] + a.ui.qp[q] + 1) >> 1;
		if (a.bs_out && col == 0) a.bs_out[q] = (uint8_t)(0x80 | bs);
	}
	int m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int16_t *e = a.y + (size_t)uy * 4 * a.ys + x;
	if (bs) {
#pragma unroll
		for (int k = 0; k < 8; k++) m[k] = e[(ptrdiff_t)(k - 4) * a.ys];
	}
	// width is a multiple of 8, so 4-lane groups never straddle rows or the `in` boundary: whole groups share bs
	const int dp = iabs(m[1] - 2 * m[2] + m[3]), dq = iabs(m[4] - 2 * m[5] + m[6]);
	const int lane = threadIdx.x & 63, g0 = lane & ~3;
	const int dp0 = __shfl(dp, g0, 64), dq0 = __shfl(dq, g0, 64), dp3 = __shfl(dp, g0 + 3, 64), dq3 = __shfl(dq, g0 + 3, 64);
	const int tc = cTc[clip3i(qpa + 2 * (bs ? bs - 1 : 0) + (a.tc_off << 1), 0, 53)];
	const int beta = cBeta[clip3i(qpa + (a.beta_off << 1), 0, 51)];
	const int d0 = dp0 + dq0, d3 = dp3 + dq3;
	const bool my_strong = strong_decision(m, 2 * (col == 0 ? d0 : d3), beta, tc);
	const bool s0 = __shfl((int)my_strong, g0, 64), s3 = __shfl((int)my_strong, g0 + 3, 64);
	if (bs && d0 + d3 < beta) {
		const int side = (beta + (beta >> 1)) >> 3;
		luma_line(m, tc, s0 && s3, tc * 10, (dp0 + dp3) < side, (dq0 + dq3) < side);
#pragma unroll
		for (int k = 1; k < 7; k++) e[(ptrdiff_t)(k - 4) * a.ys] = (int16_t)m[k];
	}
	if (bs > 1 && (uy & 3) == 0 && col < 2) {
		// two chroma columns per unit: lanes col 0 and 1 take one each
		const size_t co = (size_t)uy * 2 * a.cs + ux * 2 + col;
		const int tcb = cTc[clip3i(chroma_qp(qpa + a.cb_off) + 2 * (bs - 1) + (a.tc_off << 1), 0, 53)];
		const int tcr = cTc[clip3i(chroma_qp(qpa + a.cr_off) + 2 * (bs - 1) + (a.tc_off << 1), 0, 53)];
		for (int pl = 0; pl < 2; pl++) {
			int16_t *xp = (pl ? a.v : a.u) + co;
			const int tcc = pl ? tcr : tcb, s = a.cs;
			const int m4 = xp[0], m3 = xp[-s], m5 = xp[s], m2 = xp[-2 * s];
			const int delta = clip3i((((m4 - m3) << 2) + m2 - m5 + 4) >> 3, -tcc, tcc);
			xp[-s] = (int16_t)clip3i(m3 + delta, 0, 255);
			xp[0] = (int16_t)clip3i(m4 - delta, 0, 255);
		}
	}
}

struct Planes {
	const int16_t *p[3];
	int stride[2];   // luma, chroma
};

__device__ __forceinline__ int sgn(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// grid = (ctus, 3).  stats[ctu][comp][type][diff|count][32]
__global__ __launch_bounds__(HMR_BLOCK) void k_sao_stats(Planes org, Planes rec, int width, int height, int ctus_x, int ctu_first, int32_t *__restrict__ stats)
{
	__shared__ int16_t tile[66 * 66];
	__shared__ int sAcc[5][2][32];
	// ctu_first lets the drop-in entry run a single CTU of the picture (stats then holds that CTU only)
	const int ctu = blockIdx.x + ctu_first, comp = blockIdx.y, cx = ctu % ctus_x, cy = ctu / ctus_x;
	const int sh = comp ? 1 : 0;
	const int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
	const int h = hl >> sh, w = wl >> sh;
	const bool la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
	const int rs = rec.stride[sh], os = org.stride[sh];
	const int16_t *r0 = rec.p[comp] + (size_t)(cy * 64 >> sh) * rs + (cx * 64 >> sh);
	const int16_t *o0 = org.p[comp] + (size_t)(cy * 64 >> sh) * os + (cx * 64 >> sh);
	const int tw = w + 2;
	for (int i = threadIdx.x; i < 5 * 2 * 32; i += HMR_BLOCK) (&sAcc[0][0][0])[i] = 0;
	// stage the tile with a 1-sample ring; ring samples outside the picture are never used by the regions below
	for (int i = threadIdx.x; i < tw * (h + 2); i += HMR_BLOCK) {
		const int ty = i / tw, tx = i - ty * tw, y = ty - 1, x = tx - 1;
		const bool ok = (x >= 0 || la) && (x < w || ra) && (y >= 0 || ta) && (y < h || ba);
		tile[ty * 66 + tx] = ok ? r0[(ptrdiff_t)y * rs + x] : (int16_t)0;
	}
	__syncthreads();
	const int skr = comp ? 3 : 5, skb = comp ? 2 : 4;
	// per-lane class accumulators as bit fields (a lane sees at most 16 samples): counts 5 x 6 bits in one register per edge type;
	// differences biased by +256 (9 bits, sums < 2^13) in 16-bit fields - classes 0..3 in a 64-bit pair, class 4 on its own -
	// so a sample costs one shift-add per type instead of five select-accumulates
	unsigned cnt[4] = {0, 0, 0, 0}, d4[4] = {0, 0, 0, 0};
	unsigned long long d03[4] = {0, 0, 0, 0};
	const int ex_eo = ra ? w - skr : w - 1, ex_full = ra ? w - skr : w, sx_eo = la ? 0 : 1;
	const int ey_eo = ba ? h - skb : h - 1, ey_full = ba ? h - skb : h, sy_eo = ta ? 0 : 1;
	for (int i = threadIdx.x; i < w * h; i += HMR_BLOCK) {
		const int y = i / w, x = i - y * w;
		const int16_t *c = &tile[(y + 1) * 66 + x + 1];
		const int v = c[0], d = o0[(size_t)y * os + x] - v;
		const int sl = sgn(v - c[-1]), sr = sgn(v - c[1]), su = sgn(v - c[-66]), sd = sgn(v - c[66]);
		const int sul = sgn(v - c[-67]), sdr = sgn(v - c[67]), sur = sgn(v - c[-65]), sdl = sgn(v - c[65]);
		const bool in_x_eo = x >= sx_eo && x < ex_eo, in_y_eo = y >= sy_eo && y < ey_eo;
		const bool in[4] = {in_x_eo && y < ey_full,                // EO 0: all rows (bottom skip only)
				    x < ex_full && in_y_eo,                 // EO 90
				    in_x_eo && in_y_eo, in_x_eo && in_y_eo}; // EO 135 / 45
		const int k[4] = {2 + sl + sr, 2 + su + sd, 2 + sul + sdr, 2 + sur + sdl};
		const unsigned e = (unsigned)(d + 256);
#pragma unroll
		for (int t = 0; t < 4; t++) {
			cnt[t] += in[t] ? 1u << (6 * k[t]) : 0u;
			d03[t] += (in[t] && k[t] < 4) ? (unsigned long long)e << (16 * k[t]) : 0ull;
			d4[t] += (in[t] && k[t] == 4) ? e : 0u;
		}
		if (x < ex_full && y < ey_full) {                        // BO
			atomicAdd(&sAcc[4][0][v >> 3], d);
			atomicAdd(&sAcc[4][1][v >> 3], 1);
		}
	}
#pragma unroll
	for (int t = 0; t < 4; t++)
#pragma unroll
		for (int kk = 0; kk < 5; kk++) {
			const int cl = (int)((cnt[t] >> (6 * kk)) & 63u);
			const int el = kk < 4 ? (int)((d03[t] >> (16 * kk)) & 0xffffu) : (int)d4[t];
			const int ds = wave_sum(el - 256 * cl), cs = wave_sum(cl);
			if ((threadIdx.x & 63) == 0) {
				atomicAdd(&sAcc[t][0][kk], ds);
				atomicAdd(&sAcc[t][1][kk], cs);
			}
		}
	__syncthreads();
	int32_t *out = stats + ((size_t)blockIdx.x * 3 + comp) * 5 * 2 * 32;
	for (int i = threadIdx.x; i < 5 * 2 * 32; i += HMR_BLOCK) out[i] = (&sAcc[0][0][0])[i];
}

// grid = (ctus, 3).  dst must already hold a copy of src.
__global__ __launch_bounds__(HMR_BLOCK) void k_sao_apply(Planes src, int16_t *dy, int16_t *du, int16_t *dv, int width, int height, int ctus_x, int ctu_first,
							    const int32_t *__restrict__ params)
{
	const int ctu = ctu_first + blockIdx.x, comp = blockIdx.y, cx = ctu % ctus_x, cy = ctu / ctus_x;
	const int32_t *pc = params + (size_t)blockIdx.x * 3 * 34, *p = pc + comp * 34;   // params[0] belongs to CTU ctu_first
	if (!p[0]) return;
	const int type = p[1], sh = comp ? 1 : 0;
	const int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
	const int h = hl >> sh, w = wl >> sh, st = src.stride[sh];
	const bool la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
	const size_t base = (size_t)(cy * 64 >> sh) * st + (cx * 64 >> sh);
	const int16_t *s0 = src.p[comp] + base;
	int16_t *d0 = (comp == 0 ? dy : comp == 1 ? du : dv) + base;
	__shared__ int off[32];
	if (threadIdx.x < 32) off[threadIdx.x] = p[2 + threadIdx.x];
	__syncthreads();
	const int dx0 = type == 1 ? 0 : (type == 3 ? 1 : -1), dy0 = type == 0 ? 0 : -1;   // second neighbour is the mirror image
	for (int i = threadIdx.x; i < w * h; i += HMR_BLOCK) {
		const int y = i / w, x = i - y * w;
		const int c = s0[(size_t)y * st + x];
		int k;
		if (type == 4) k = c >> 3;
		else {
			if (type != 1 && ((x == 0 && !la) || (x == w - 1 && !ra))) continue;
			if (type != 0 && ((y == 0 && !ta) || (y == h - 1 && !ba))) continue;
			k = 2 + sgn(c - s0[(ptrdiff_t)(y + dy0) * st + x + dx0]) + sgn(c - s0[(ptrdiff_t)(y - dy0) * st + x - dx0]);
		}
		d0[(size_t)y * st + x] = (int16_t)clip3i(c + off[k], 0, 255);
	}
}

// replicate the picture edge into the margins; pic = sample (0,0); one thread per margin-or-picture sample of the padded window
__global__ __launch_bounds__(HMR_BLOCK) void k_pad(int16_t *pic, int stride, int width, int height, int pad_x, int pad_y)
{
	const int pw = width + 2 * pad_x;
	const long i = (long)blockIdx.x * HMR_BLOCK + threadIdx.x;
	if (i >= (long)pw * (height + 2 * pad_y)) return;
	const int y = (int)(i / pw) - pad_y, x = (int)(i % pw) - pad_x;
	if (x >= 0 && x < width && y >= 0 && y < height) return;
	pic[(ptrdiff_t)y * stride + x] = pic[(ptrdiff_t)clip3i(y, 0, height - 1) * stride + clip3i(x, 0, width - 1)];
}

// per-CTU z-order side-info (ctu_info_t, hmr_private.h:792-843) -> raster arrays over the picture; abs2raster_table
// (hmr_encoder_lib.c:95-100) is the Morton de-interleave of the unit index.  One thread per unit.
__global__ __launch_bounds__(HMR_BLOCK) void k_units_from_ctus(hmr_gpu_ctu_units src, int ctus_x, int n_units, int units_stride, int16_t *__restrict__ mvx,
								  int16_t *__restrict__ mvy, int8_t *__restrict__ ref_idx, uint8_t *__restrict__ qp,
								  uint8_t *__restrict__ flags, uint8_t *__restrict__ pred_depth, uint8_t *__restrict__ tr_idx)
{
	const int s = blockIdx.x * HMR_BLOCK + threadIdx.x;
	if (s >= n_units) return;
	const int c = s >> 8, a = s & 255;
	int x = 0, y = 0;
#pragma unroll
	for (int b = 0; b < 4; b++) {
		x |= ((a >> (2 * b)) & 1) << b;
		y |= ((a >> (2 * b + 1)) & 1) << b;
	}
	const size_t o = (size_t)((c / ctus_x) * 16 + y) * units_stride + (c % ctus_x) * 16 + x;
	mvx[o] = src.mvx[s]; mvy[o] = src.mvy[s]; ref_idx[o] = src.ref_idx[s]; qp[o] = src.qp[s];
	const int tr = src.tr_idx[s];
	flags[o] = (uint8_t)((src.pred_mode[s] == 1 ? F_INTRA : 0) | (((src.cbf_y[s] >> tr) & 1) ? F_CBF : 0));   // INTRA_MODE = 1 (hmr_private.h:221), CBF() hmr_common.h:73
	if (pred_depth) pred_depth[o] = src.pred_depth[s];
	if (tr_idx) tr_idx[o] = (uint8_t)tr;
}

}  // namespace

extern "C" int hmr_gpu_units_from_ctus(hmr_gpu_ctx *ctx, const hmr_gpu_ctu_units *src, int ctus_x, int ctus_y, const hmr_gpu_units *dst, uint8_t *pred_depth,
				       uint8_t *tr_idx)
{
	if (!src || !dst || ctus_x <= 0 || ctus_y <= 0 || dst->units_stride < ctus_x * 16) return HMR_GPU_ERR_ARG;
	const int n = ctus_x * ctus_y * 256;
	hipLaunchKernelGGL(k_units_from_ctus, dim3((n + HMR_BLOCK - 1) / HMR_BLOCK), dim3(HMR_BLOCK), 0, ctx->stream, *src, ctus_x, n, dst->units_stride,
			   const_cast<int16_t *>(dst->mvx), const_cast<int16_t *>(dst->mvy), const_cast<int8_t *>(dst->ref_idx), const_cast<uint8_t *>(dst->qp), dst->flags,
			   pred_depth, tr_idx);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_edge_flags_frame(hmr_gpu_ctx *ctx, const uint8_t *pred_depth, const uint8_t *tr_idx, int width, int height, int units_stride,
					uint8_t *flags)
{
	const int n = (width / 4) * (height / 4);
	hipLaunchKernelGGL(k_edge_flags, dim3((n + HMR_BLOCK - 1) / HMR_BLOCK), dim3(HMR_BLOCK), 0, ctx->stream, pred_depth, tr_idx, 0, 0, width / 4, height / 4,
			   units_stride, flags);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

static int deblock_args(DeblockArgs &a, const hmr_gpu_frame *f, const hmr_gpu_units *info, int cb_qp_offset, int cr_qp_offset, int beta_offset_div2,
			int tc_offset_div2)
{
	if (!f || !info || (f->width & 7) || (f->height & 7) || (f->stride_y & 3) || (f->stride_c & 1) || ((uintptr_t)f->y & 7)) {
		hmr_set_error("deblock: width/height must be multiples of 8, luma stride a multiple of 4 and the luma plane 8-byte aligned");
		return HMR_GPU_ERR_ARG;
	}
	a.y = f->y; a.u = f->u; a.v = f->v; a.ys = f->stride_y; a.cs = f->stride_c;
	a.ux0 = a.uy0 = 0;
	a.w4 = f->width / 4; a.h4 = f->height / 4;
	a.ui.units_stride = info->units_stride; a.ui.mvx = info->mvx; a.ui.mvy = info->mvy; a.ui.ref_idx = info->ref_idx; a.ui.qp = info->qp; a.ui.flags = info->flags;
	a.cb_off = cb_qp_offset; a.cr_off = cr_qp_offset; a.beta_off = beta_offset_div2; a.tc_off = tc_offset_div2;
	a.bs_out = nullptr;
	return HMR_GPU_OK;
}

static void deblock_launch(hmr_gpu_ctx *ctx, const DeblockArgs &a, int dir)
{
	if (dir == 0) {
		const int nver = ((a.w4 + 1) / 2) * a.h4;
		hipLaunchKernelGGL(k_deblock_ver, dim3((nver + HMR_BLOCK - 1) / HMR_BLOCK), dim3(HMR_BLOCK), 0, ctx->stream, a);
	} else {
		const int nhor = a.w4 * 4 * ((a.h4 + 1) / 2);
		hipLaunchKernelGGL(k_deblock_hor, dim3((nhor + HMR_BLOCK - 1) / HMR_BLOCK), dim3(HMR_BLOCK), 0, ctx->stream, a);
	}
}

extern "C" int hmr_gpu_deblock_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *f, const hmr_gpu_units *info, int cb_qp_offset, int cr_qp_offset,
				     int beta_offset_div2, int tc_offset_div2, uint8_t *bs_ver, uint8_t *bs_hor)
{
	DeblockArgs a;
	const int rc = deblock_args(a, f, info, cb_qp_offset, cr_qp_offset, beta_offset_div2, tc_offset_div2);
	if (rc != HMR_GPU_OK) return rc;
	a.bs_out = bs_ver;
	deblock_launch(ctx, a, 0);
	a.bs_out = bs_hor;
	deblock_launch(ctx, a, 1);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

// the reference's own call granularity (hmr_deblock_filter_cu, hmr_deblocking_filter.c:737): the edges of one direction inside one CTU; the EDGE bits
// of the CTU's units are derived first when the coding-tree arrays are given
extern "C" int hmr_gpu_deblock_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *f, const hmr_gpu_units *info, const uint8_t *pred_depth, const uint8_t *tr_idx,
				   int cb_qp_offset, int cr_qp_offset, int beta_offset_div2, int tc_offset_div2, int ctu_x, int ctu_y, int ctu_size, int dir)
{
	DeblockArgs a;
	const int rc = deblock_args(a, f, info, cb_qp_offset, cr_qp_offset, beta_offset_div2, tc_offset_div2);
	if (rc != HMR_GPU_OK) return rc;
	if ((ctu_x & 7) || (ctu_y & 7) || ctu_x < 0 || ctu_y < 0 || ctu_x >= f->width || ctu_y >= f->height || (dir != 0 && dir != 1)) {
		hmr_set_error("deblock_ctu: bad CTU position / direction");
		return HMR_GPU_ERR_ARG;
	}
	const int x1 = ctu_x + ctu_size < f->width ? ctu_x + ctu_size : f->width, y1 = ctu_y + ctu_size < f->height ? ctu_y + ctu_size : f->height;
	a.ux0 = ctu_x / 4; a.uy0 = ctu_y / 4;
	a.w4 = (x1 - ctu_x) / 4; a.h4 = (y1 - ctu_y) / 4;
	if (pred_depth && tr_idx) {
		const int n = a.w4 * a.h4;
		hipLaunchKernelGGL(k_edge_flags, dim3((n + HMR_BLOCK - 1) / HMR_BLOCK), dim3(HMR_BLOCK), 0, ctx->stream, pred_depth, tr_idx, a.ux0, a.uy0, a.w4, a.h4,
				   info->units_stride, info->flags);
	}
	deblock_launch(ctx, a, dir);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

static Planes planes_of(const hmr_gpu_frame *f)
{
	Planes p;
	p.p[0] = f->y; p.p[1] = f->u; p.p[2] = f->v;
	p.stride[0] = f->stride_y; p.stride[1] = f->stride_c;
	return p;
}

extern "C" int hmr_gpu_sao_stats_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *orig, const hmr_gpu_frame *recon, int32_t *stats)
{
	if (!orig || !recon || orig->width != recon->width || orig->height != recon->height) return HMR_GPU_ERR_ARG;
	const int ctus_x = (recon->width + 63) / 64, ctus_y = (recon->height + 63) / 64;
	hipLaunchKernelGGL(k_sao_stats, dim3(ctus_x * ctus_y, 3), dim3(HMR_BLOCK), 0, ctx->stream, planes_of(orig), planes_of(recon), recon->width, recon->height,
			   ctus_x, 0, stats);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

// one CTU of the picture (the granularity of low_level_funcs_t.get_sao_stats): stats[3][5][2][32]
extern "C" int hmr_gpu_sao_stats_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *orig, const hmr_gpu_frame *recon, int ctu_index, int32_t *stats)
{
	if (!orig || !recon || orig->width != recon->width || orig->height != recon->height) return HMR_GPU_ERR_ARG;
	const int ctus_x = (recon->width + 63) / 64;
	hipLaunchKernelGGL(k_sao_stats, dim3(1, 3), dim3(HMR_BLOCK), 0, ctx->stream, planes_of(orig), planes_of(recon), recon->width, recon->height, ctus_x, ctu_index,
			   stats);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_sao_apply_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *src, const hmr_gpu_frame *dst, const int32_t *params)
{
	if (!src || !dst || src->width != dst->width || src->height != dst->height || src->stride_y != dst->stride_y || src->stride_c != dst->stride_c)
		return HMR_GPU_ERR_ARG;
	const int ctus_x = (src->width + 63) / 64, ctus_y = (src->height + 63) / 64;
	hipLaunchKernelGGL(k_sao_apply, dim3(ctus_x * ctus_y, 3), dim3(HMR_BLOCK), 0, ctx->stream, planes_of(src), dst->y, dst->u, dst->v, src->width, src->height,
			   ctus_x, 0, params);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

// one CTU (sao_offset_ctu's own granularity); params = this CTU's [3][34]
extern "C" int hmr_gpu_sao_apply_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *src, const hmr_gpu_frame *dst, int ctu_index, const int32_t *params)
{
	if (!src || !dst || src->width != dst->width || src->height != dst->height || src->stride_y != dst->stride_y || src->stride_c != dst->stride_c)
		return HMR_GPU_ERR_ARG;
	const int ctus_x = (src->width + 63) / 64;
	hipLaunchKernelGGL(k_sao_apply, dim3(1, 3), dim3(HMR_BLOCK), 0, ctx->stream, planes_of(src), dst->y, dst->u, dst->v, src->width, src->height, ctus_x, ctu_index,
			   params);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

// the three planes of a 4:2:0 picture in one launch (blockIdx.y = plane; the grid is the luma plane's)
__global__ __launch_bounds__(HMR_BLOCK) void k_pad420(int16_t *y, int16_t *u, int16_t *v, int stride_y, int stride_c, int width, int height, int pad_x, int pad_y)
{
	const int c = (int)blockIdx.y;
	int16_t *pic = c == 0 ? y : (c == 1 ? u : v);
	const int stride = c ? stride_c : stride_y, w = c ? width / 2 : width, h = c ? height / 2 : height, px = c ? pad_x / 2 : pad_x, py = c ? pad_y / 2 : pad_y;
	const int pw = w + 2 * px;
	const long i = (long)blockIdx.x * HMR_BLOCK + threadIdx.x;
	if (i >= (long)pw * (h + 2 * py)) return;
	const int yy = (int)(i / pw) - py, xx = (int)(i % pw) - px;
	if (xx >= 0 && xx < w && yy >= 0 && yy < h) return;
	pic[(ptrdiff_t)yy * stride + xx] = pic[(ptrdiff_t)clip3i(yy, 0, h - 1) * stride + clip3i(xx, 0, w - 1)];
}

extern "C" int hmr_gpu_pad_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *f, int pad_x, int pad_y)
{
	if (!f) return HMR_GPU_ERR_ARG;
	const long n = (long)(f->width + 2 * pad_x) * (f->height + 2 * pad_y);
	hipLaunchKernelGGL(k_pad420, dim3((unsigned)((n + HMR_BLOCK - 1) / HMR_BLOCK), 3), dim3(HMR_BLOCK), 0, ctx->stream, f->y, f->u, f->v, f->stride_y, f->stride_c, f->width, f->height,
			   pad_x, pad_y);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
