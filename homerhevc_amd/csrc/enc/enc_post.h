// The post-decision stage of a CTU: what the reference's WPP thread does behind the CTU decisions in its CTU-lagged pipeline (hmr_deblock_sao_pad_sync_ctu,
// hmr_encoder_lib.c:2386-2843) - deblocking (hmr_deblock_filter_cu, hmr_deblocking_filter.c:737), SAO statistics (sse_sao_get_ctu_stats, hmr_sse42_sao.c:35),
// SAO parameter decision (hmr_wpp_sao_ctu, hmr_sao.c:1410 -> sao_decide_blk_params :1295), entropy coding of the CTU (wfpp_encode_ctu :2347: ee_encode_sao +
// ee_encode_ctu, with the bit count the rate control reads, :2366), SAO offset (sao_offset_ctu, hmr_sao.c:1210) and border padding
// (reference_picture_border_padding_ctu :1723) - as SPMD code like the decision core, run by the same workers as TASKS of the CTU kernel:
//
//   D(r, c)  copy CTU (r, c) of the reconstruction into the deblocking picture, its side-info into the raster unit arrays, vertical edges of (r, c), then the
//            horizontal edges of (r, c - 1) (and of (r, c) when it is the row's last);
//   P(r, c)  SAO statistics -> candidate offsets -> decision -> SAO syntax + CTU syntax into the row's CABAC sub-stream -> SAO offset into the final picture
//            -> border padding.
//
// The reference orders these by a fixed lag behind the CTU being decided and filters in place; its results do not depend on that order (every sample a stage
// reads is final when read; the picture grids where that does not hold are refused, enc_host.h).  Here three pictures take the place of the one - reconstruction
// (what intra prediction of later CTUs reads: never filtered), deblocked, final - and a task runs as soon as what it reads is final:
//   D(r, c):  CTU (r, c) decided; D(r, c - 1) done; D(r - 1, c) done (its vertical edges: the horizontal edges of row r read the rows above them);
//   P(r, c):  the horizontal edges of (r, c + 1) and (r + 1, c + 1) done (every deblocked sample of the CTU and of the ring around it is final);
//             P(r, c - 1) done (sub-stream order, SAO merge-left); P(r - 1, c + 1) done (SAO merge-up; the WPP context hand-over after the row's second CTU).
// Without SAO the reference codes a CTU right behind its decisions (:2611-2612) and the rate control may read its bits one step later: P(r, c) then waits for the
// CTU's decisions only, and the copy of the deblocked CTU into the final picture + padding is a task of its own, F(r, c), with P's filter conditions.
// When the frames of a sequence overlap (hmr_gpu_enc_encode_chain) a fourth task S(r, c) makes the phase planes of the final picture around CTU (r, c) as soon as
// P (or F) has run on the CTUs around it; the next frame's CTUs wait for the S tasks of the part of this picture their vectors can reach (k_encode.hip).
// Progress is kept in per-row counters; any worker may run any task whose conditions hold (claimed with a compare-and-swap on the row's task ticket).
// Under rate control the CTU decisions of wavefront step t read the bits of the CTUs the reference has coded when step t starts: CodedSchedule replays the
// reference's lag arithmetic once per picture size and the step does not open before those P tasks are done (enc_rc.h).
#pragma once
#include "enc_entropy.h"
// The post-decision tasks stay functions of their own on the device: inlined into the pool's drain they made IT save eighty registers to private memory on every call -
// and an idle worker calls it for every picture of the launch to find out that nothing is ready.
#if defined(__HIP_DEVICE_COMPILE__)
#define HENC_TASK_FN __attribute__((noinline))
#else
#define HENC_TASK_FN
#endif
#include "enc_rc.h"
#if defined(__HIPCC__)
#include "../subpel_task.h"      // (device only) the phase planes of the final picture, CTU by CTU: task S
#endif

#if defined(__HIPCC__)
#define HENC_NOINLINE __attribute__((noinline))      // one compiled body for every task that uses the function (and: inlined into k_post_frame's copy of the F task, the
                                                     // offset pass of ROCm 7.2's compiler stored nothing - seen on the MI355X, profiles/r04_history.md)
#else
#define HENC_NOINLINE
#endif

namespace henc {

// ---- memory-order helpers: agent scope on the device, plain on the CPU (one thread) ------------------------------------------------------------------------
HENC_INLINE int post_ld(const int *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	return *p;
#endif
}
HENC_INLINE void post_st_release(int *p, int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	__hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
	*p = v;
#endif
}
// the store after a post_release() of the whole wavefront: the fence has published what the store announces (a release STORE would write the L2 back a second time)
HENC_INLINE void post_st_fenced(int *p, int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	*p = v;
#endif
}
HENC_INLINE void post_acquire()
{
#if defined(__HIP_DEVICE_COMPILE__)
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
HENC_INLINE void post_release()
{
#if defined(__HIP_DEVICE_COMPILE__)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}
// one lane tries to move *p from `expect` to expect + 1; every lane learns the outcome
template <class G>
HENC_HDX bool post_claim(const G g, int *p, int expect)
{
#if defined(__HIP_DEVICE_COMPILE__)
	int ok = 0;
	if (g.tid == 0) {
		int e = expect;
		ok = __hip_atomic_compare_exchange_strong(p, &e, expect + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
	}
	return __builtin_amdgcn_readfirstlane(ok) != 0;
#else
	(void)g;
	if (*p != expect) return false;
	*p = expect + 1;
	return true;
#endif
}
template <class G>
HENC_HDX void post_add_fast(const G g, int32_t *p, int32_t v)      // an accumulator in the worker's fast memory that several lanes add to
{
#if defined(__HIP_DEVICE_COMPILE__)
	(void)g;
	__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
	(void)g;
	*p += v;
#endif
}

// ---- per-picture state of the stage ----------------------------------------------------------------------------------------------------------------------------
struct PostRow {               // progress of one CTU row: CTUs decided, D / P / F / S tasks claimed and done
	int dec, d_claim, d_done, p_claim, p_done, f_claim, f_done, s_claim, s_done, pad_[7];
};
struct RowEnt {                // the CABAC coder of a CTU row's sub-stream between two CTUs, and the contexts the next row starts from
	uint32_t low, range, buffered_byte;
	int32_t num_buffered, bits_left, bytecnt;
	uint8_t ctx[CTX_TOTAL + 5], saved[CTX_TOTAL + 5];
};
static_assert(sizeof(RowEnt) % 4 == 0, "word copies");
struct PostPic {
	int16_t *dbk[3], *fin[3];  // the deblocked picture and the final one (padded planes, first valid sample; the reconstruction is FrameCtx::rec)
	int units_stride;
	int16_t *mvx, *mvy;        // side-info of the picture's 4x4 units in raster order (what the deblocking filter reads)
	int8_t *ref;
	uint8_t *uqp, *flags;
	PostRow *rows;             // [hctu]
	RowEnt *ent;               // [hctu]
	uint8_t *bs;               // sub-stream of row r at bs + r * row_cap
	int row_cap;
	uint32_t *cumbits;         // [nctu] bits of the CTU's sub-stream up to and including the CTU (what hmr_bitstream_bitcount has grown by, :2366)
	const double *sao_lambda;  // [52][2] the SAO Lagrange multipliers by the CTU's QP: luma, chroma (hmr_wpp_sao_ctu :1415-1430; pow() stays on the host)
	int *errors;               // [0] a sub-stream ran out of room
	const uint16_t *rc_need;   // rate control / RD_FULL: [steps + 1][hctu] P tasks of each row the reference has run when the step starts (nullptr: neither)
	uint8_t *ctx_after;        // RD_FULL: [nctu][RD_CTX_BYTES] the context states of the CTU's sub-stream after the CTU (what later decisions' bit estimates copy), or nullptr
	unsigned long long *prof;  // profiling build (-DHENC_POST_PROFILE): ticks per part of the stage (PostProf), else unused
	uint8_t *planes[3];        // device, overlapping frames of a sequence: the phase planes of the final picture (allocation start: k_subpel.hip's layout), written by
	                           // task S(r, c) when the final samples around CTU (r, c) are there; nullptr: the planes are made by the frame kernels before the next frame
};
enum PostProf { PPF_D = 0, PPF_P_LOAD, PPF_P_STATS, PPF_P_CAND, PPF_P_DECIDE, PPF_P_SYNTAX, PPF_P_APPLY, PPF_SCAN, PPF_D_COUNT, PPF_P_COUNT, PPF_N };
#if defined(__HIP_DEVICE_COMPILE__) && defined(HENC_POST_PROFILE)
#define PPF_T0() unsigned long long ppf_t_ = __builtin_amdgcn_s_memtime()
#define PPF_LAP(P, cat) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && (P).prof) atomicAdd((P).prof + (cat), n_ - ppf_t_); ppf_t_ = n_; } while (0)
#define PPF_COUNT(P, cat) do { if (threadIdx.x == 0 && (P).prof) atomicAdd((P).prof + (cat), 1ull); } while (0)
#else
#define PPF_T0() do { } while (0)
#define PPF_LAP(P, cat) do { } while (0)
#define PPF_COUNT(P, cat) do { } while (0)
#endif

constexpr int UF_INTRA = 1, UF_CBF = 2, UF_EDGE_VER = 4, UF_EDGE_HOR = 8;      // unit flags (the frame-level kernels' bits, k_loop.hip)

// samples of the deblocked CTU in the scratch tiles: bytes on the device (the scratch has to fit what a worker's fast memory leaves idle between two CTUs)
#if defined(__HIPCC__)
typedef uint8_t tile_t;
#else
typedef int16_t tile_t;
#endif
constexpr int POST_TS_Y = 72, POST_TS_C = 40;      // row pitches of the tiles: the CTU's first column is 8-byte aligned behind the ring column
// scratch of a task in the worker's fast memory (on the device the worker's Work area, idle between two CTUs)
struct alignas(16) PostScratch {
	alignas(16) CtuPublic c;                       // the CTU's record while it is coded
	EntScratch ent;
	uint8_t ctx[CTX_TOTAL + 5];
	uint8_t t_range[256], t_next[128];   // the coder's tables next to it
	alignas(16) int16_t coef[6144];    // the CTU's levels
	alignas(16) tile_t tile_y[66 * POST_TS_Y], tile_c[2][34 * POST_TS_C];   // the deblocked CTU with a one-sample ring, per component (sample (x, y) at (y + 1) * pitch + x + 4)
	int32_t acc[5][2][32];             // statistics of the component being counted
	SaoStats stats;
	int32_t cand_off[3][5][32], cand_aux[3][5];
	int64_t cand_dist[3][5];
	double cost[32];
	int64_t dist[64];
	// what the syntax coder reads the CTU and its neighbours through (handed on by address: as locals they lived in private memory)
	EntView view;
	CtuView view_c, view_l, view_t;
	SaoOffset sao_ws[SAO_DECIDE_WS];   // sao_decide's candidates under test
};
#if defined(__HIPCC__)
// (k_encode.hip checks that the scratch fits the part of a worker's fast memory that is idle between two CTUs: its Work and the CTU's partition nodes)
// (task S works in the same place: k_encode.hip checks SubpelScratch against it too)
#endif

static constexpr uint8_t kDbkTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
				       2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static constexpr uint8_t kDbkBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
					 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};

// ---- D task ------------------------------------------------------------------------------------------------------------------------------------------------------
// the CTU's side-info (z-order) into the raster unit arrays, with the transform / prediction edge flags (k_units_from_ctuinfo + k_edge_flags)
template <class G>
HENC_HDX void post_units_ctu(const G g, const PostPic &P, const CtuPublic &c, int cx, int cy)
{
	for (int a = g.tid; a < NPART; a += g.n) {
		const int r = abs2raster(a), ux = cx * 16 + (r & 15), uy = cy * 16 + (r >> 4);
		const size_t o = (size_t)uy * P.units_stride + ux;
		P.mvx[o] = (int16_t)c.mv_ref[a].x;
		P.mvy[o] = (int16_t)c.mv_ref[a].y;
		P.ref[o] = c.mv_ref_idx[a];
		P.uqp[o] = c.qp[a];
		int leaf = 64 >> (c.pred_depth[a] + c.tr_idx[a]);
		if (leaf < 8) leaf = 8;
		int f = (c.pred_mode[a] == PM_INTRA ? UF_INTRA : 0) | (((c.cbf[0][a] >> c.tr_idx[a]) & 1) ? UF_CBF : 0);
		if (ux && (ux * 4) % leaf == 0) f |= UF_EDGE_VER;
		if (uy && (uy * 4) % leaf == 0) f |= UF_EDGE_HOR;
		P.flags[o] = (uint8_t)f;
	}
}

// the CTU's samples from the reconstruction into the deblocking picture
template <class G>
HENC_HDX void post_copy_ctu(const G g, const Seq &S, const FrameCtx &f, const PostPic &P, int cx, int cy)
{
	for (int comp = 0; comp < 3; comp++) {
		const int sz = comp ? 32 : 64, px = cx * sz, py = cy * sz, pw = comp ? S.width >> 1 : S.width, ph = comp ? S.height >> 1 : S.height;
		const int st = comp ? S.stride_c : S.stride_y;
		const int ww = px + sz < pw ? sz : pw - px, hh = py + sz < ph ? sz : ph - py;
		const int16_t *s = f.rec[comp] + (size_t)py * st + px;
		int16_t *d = P.dbk[comp] + (size_t)py * st + px;
		const int w4 = ww >> 2;                              // (widths are multiples of 4: the picture is a multiple of the 8 x 8 minimum CU)
		for (int i = g.tid; i < w4 * hh; i += g.n) {
			const int y = i / w4, x = (i - y * w4) << 2;
			st4(d + (size_t)y * st + x, ld4(s + (size_t)y * st + x));
		}
	}
}

HENC_INLINE int dbk_chroma_qp(int q)
{
	q = hclip(q, 0, 57);
	return chroma_qp_table(q);
}
// boundary strength of the edge between units q and p (get_bs, hmr_deblocking_filter.c:138; P slices with one reference picture)
HENC_INLINE int dbk_bs(const PostPic &P, size_t q, size_t p)
{
	const int fq = P.flags[q], fp = P.flags[p];
	if ((fq | fp) & UF_INTRA) return 2;
	if ((fq | fp) & UF_CBF) return 1;
	const int rq = P.ref[q], rp = P.ref[p];
	const int mqx = rq < 0 ? 0 : P.mvx[q], mqy = rq < 0 ? 0 : P.mvy[q];
	const int mpx = rp < 0 ? 0 : P.mvx[p], mpy = rp < 0 ? 0 : P.mvy[p];
	if ((rp < 0 ? -1 : rp) != (rq < 0 ? -1 : rq)) return 1;
	return (habs(mqx - mpx) >= 4 || habs(mqy - mpy) >= 4) ? 1 : 0;
}
// luma filter of one line m[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 (filter_luma :287)
HENC_INLINE void dbk_luma_line(int *m, int tc, bool strong, int thr_cut, bool fp, bool fq)
{
	const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7];
	if (strong) {
		m[3] = hclip((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3, m3 - 2 * tc, m3 + 2 * tc);
		m[4] = hclip((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3, m4 - 2 * tc, m4 + 2 * tc);
		m[2] = hclip((m1 + m2 + m3 + m4 + 2) >> 2, m2 - 2 * tc, m2 + 2 * tc);
		m[5] = hclip((m3 + m4 + m5 + m6 + 2) >> 2, m5 - 2 * tc, m5 + 2 * tc);
		m[1] = hclip((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - 2 * tc, m1 + 2 * tc);
		m[6] = hclip((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3, m6 - 2 * tc, m6 + 2 * tc);
	} else {
		int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
		if (habs(delta) < thr_cut) {
			const int tc2 = tc >> 1;
			delta = hclip(delta, -tc, tc);
			m[3] = hclip(m3 + delta, 0, 255);
			m[4] = hclip(m4 - delta, 0, 255);
			if (fp) m[2] = hclip(m2 + hclip((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1, -tc2, tc2), 0, 255);
			if (fq) m[5] = hclip(m5 + hclip((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1, -tc2, tc2), 0, 255);
		}
	}
}
HENC_INLINE bool dbk_strong(const int *m, int d, int beta, int tc)
{
	return (habs(m[0] - m[3]) + habs(m[7] - m[4]) < (beta >> 3)) && (d < (beta >> 2)) && (habs(m[3] - m[4]) < ((tc * 5 + 1) >> 1));
}
// two chroma lines across an edge: e = first q sample, s = step across the edge, t = step along it
HENC_INLINE void dbk_chroma_edge(int16_t *e, int s, int t, int tc)
{
	for (int i = 0; i < 2; i++) {
		int16_t *x = e + i * t;
		const int m4 = x[0], m3 = x[-s], m5 = x[s], m2 = x[-2 * s];
		const int delta = hclip((((m4 - m3) << 2) + m2 - m5 + 4) >> 3, -tc, tc);
		x[-s] = (int16_t)hclip(m3 + delta, 0, 255);
		x[0] = (int16_t)hclip(m4 - delta, 0, 255);
	}
}

// edges of one direction inside CTU (cx, cy) and on its left / top border (hmr_deblock_filter_cu): a lane owns a four-sample edge segment
template <class G>
HENC_HDX void post_deblock_ctu(const G g, const Seq &S, const PostPic &P, int cx, int cy, int dir)
{	cx = uni(cx); cy = uni(cy); dir = uni(dir);      // (arguments of a function of its own arrive in vector registers)

	const int w4 = hmin(16, (S.width >> 2) - cx * 16), h4 = hmin(16, (S.height >> 2) - cy * 16);
	const int ys = S.stride_y, cs = S.stride_c, us = P.units_stride;
	const int cb_off = S.chroma_qp_offset, cr_off = S.chroma_qp_offset;
	const int along = dir == 0 ? h4 : w4, across = dir == 0 ? (w4 + 1) / 2 : (h4 + 1) / 2;
	for (int i = g.tid; i < along * across; i += g.n) {
		// vertical edges: unit row i / across, even unit column; horizontal edges: even unit row, unit column i % along
		const int ux = cx * 16 + (dir == 0 ? (i % across) * 2 : i % along), uy = cy * 16 + (dir == 0 ? i / across : (i / along) * 2);
		const size_t q = (size_t)uy * us + ux;
		if (!(P.flags[q] & (dir == 0 ? UF_EDGE_VER : UF_EDGE_HOR))) continue;
		const size_t p = dir == 0 ? q - 1 : q - us;
		const int bs = dbk_bs(P, q, p);
		if (!bs) continue;
		const int qpa = (P.uqp[p] + P.uqp[q] + 1) >> 1;
		const int tc = kDbkTc[hclip(qpa + 2 * (bs - 1), 0, 53)];
		const int beta = kDbkBeta[hclip(qpa, 0, 51)];
		int16_t *e = P.dbk[0] + (size_t)uy * 4 * ys + ux * 4;
		const int sa = dir == 0 ? 1 : ys, sl = dir == 0 ? ys : 1;      // step across the edge / along it
		int m[4][8];
		for (int l = 0; l < 4; l++)
			for (int k = 0; k < 8; k++) m[l][k] = e[(ptrdiff_t)l * sl + (ptrdiff_t)(k - 4) * sa];
		const int dp0 = habs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = habs(m[0][4] - 2 * m[0][5] + m[0][6]);
		const int dp3 = habs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = habs(m[3][4] - 2 * m[3][5] + m[3][6]);
		const int d0 = dp0 + dq0, d3 = dp3 + dq3;
		if (d0 + d3 < beta) {
			const int side = (beta + (beta >> 1)) >> 3;
			const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
			const bool sw = dbk_strong(m[0], 2 * d0, beta, tc) && dbk_strong(m[3], 2 * d3, beta, tc);
			for (int l = 0; l < 4; l++) {
				dbk_luma_line(m[l], tc, sw, tc * 10, fp, fq);
				for (int k = 1; k < 7; k++) e[(ptrdiff_t)l * sl + (ptrdiff_t)(k - 4) * sa] = (int16_t)m[l][k];
			}
		}
		if (bs > 1 && ((dir == 0 ? ux : uy) & 3) == 0) {
			const size_t co = (size_t)uy * 2 * cs + ux * 2;
			const int ca = dir == 0 ? 1 : cs, cl = dir == 0 ? cs : 1;
			dbk_chroma_edge(P.dbk[1] + co, ca, cl, kDbkTc[hclip(dbk_chroma_qp(qpa + cb_off) + 2 * (bs - 1), 0, 53)]);
			dbk_chroma_edge(P.dbk[2] + co, ca, cl, kDbkTc[hclip(dbk_chroma_qp(qpa + cr_off) + 2 * (bs - 1), 0, 53)]);
		}
	}
}

// ---- P task: SAO ------------------------------------------------------------------------------------------------------------------------------------------------
HENC_INLINE int sgn3(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// the three components of the deblocked CTU with a one-sample ring into the scratch tiles (what the statistics and the offset pass read): ring samples outside
// the picture are never used
template <class G>
HENC_NOINLINE HENC_HDX void post_stage_tiles(const G g, const Seq &S, const PostPic &P, PostScratch &sc, int cx, int cy)
{	cx = uni(cx); cy = uni(cy);

	const int width = S.width, height = S.height;
	const bool la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
	const int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, h = hl >> sh, w = wl >> sh, w4 = w >> 2;
		const int rs = comp ? S.stride_c : S.stride_y, ts = comp ? POST_TS_C : POST_TS_Y;
		const int16_t *r0 = P.dbk[comp] + (size_t)((cy * 64) >> sh) * rs + ((cx * 64) >> sh);
		tile_t *t0 = (comp ? sc.tile_c[comp - 1] : sc.tile_y) + ts + 4;
		const int y0 = ta ? -1 : 0, y1 = ba ? h + 1 : h;
		// rows (with the ring rows above / below), four samples per lane
		for (int i = g.tid; i < w4 * (y1 - y0); i += g.n) {
			const int y = y0 + i / w4, x = (i % w4) << 2;
			st4(t0 + y * ts + x, ld4(r0 + (ptrdiff_t)y * rs + x));
		}
		// ring columns
		for (int i = g.tid; i < 2 * (y1 - y0); i += g.n) {
			const int y = y0 + (i >> 1), right = i & 1;
			if (right ? ra : la) t0[y * ts + (right ? w : -1)] = (tile_t)r0[(ptrdiff_t)y * rs + (right ? w : -1)];
		}
	}
	g.sync();
}

// statistics of CTU (cx, cy) (sse_sao_get_ctu_stats; the scalar form hmr_sao.c:75-348): for the three components the differences and counts of the five edge
// classes of the four edge types and of the 32 bands, over the CTU minus the margins the reference leaves out because they were not deblocked yet in its pipeline
template <class G>
HENC_HDX void post_sao_stats(const G g, const Seq &S, const FrameCtx &f, PostScratch &sc, int cx, int cy)
{	cx = uni(cx); cy = uni(cy);

	const int width = S.width, height = S.height;
	const bool la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
	const int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, h = hl >> sh, w = wl >> sh, w4 = w >> 2;
		const int os = comp ? S.src_stride_c : S.src_stride_y, ts = comp ? POST_TS_C : POST_TS_Y;
		const int16_t *o0 = f.src[comp] + (size_t)((cy * 64) >> sh) * os + ((cx * 64) >> sh);
		const tile_t *t0 = (comp ? sc.tile_c[comp - 1] : sc.tile_y) + ts + 4;
		for (int i = g.tid; i < 5 * 2 * 32; i += g.n) (&sc.acc[0][0][0])[i] = 0;
		g.sync();
		const int skr = comp ? 3 : 5, skb = comp ? 2 : 4;
		const int ex_eo = ra ? w - skr : w - 1, ex_full = ra ? w - skr : w, sx_eo = la ? 0 : 1;
		const int ey_eo = ba ? h - skb : h - 1, ey_full = ba ? h - skb : h, sy_eo = ta ? 0 : 1;
		// per-lane class accumulators as bit fields, emptied every 16 samples: counts 5 x 6 bits per edge type; differences biased by +256 (sums < 2^13) in
		// 16-bit fields - classes 0..3 in a 64-bit word, class 4 on its own.  A lane takes four samples of a row per step.
		unsigned cnt[4] = {0, 0, 0, 0}, d4[4] = {0, 0, 0, 0};
		unsigned long long d03[4] = {0, 0, 0, 0};
		int held = 0;
		const int total = w4 * h, rounds = (total + g.n - 1) / g.n;
		for (int it = 0; it < rounds; it++) {
			const int i = it * g.n + g.tid;
			if (i < total) {
				const int y = i / w4, x0 = (i - y * w4) << 2;
				const S4 org = ld4(o0 + (size_t)y * os + x0);
				const tile_t *c = t0 + y * ts + x0;
				const bool in_y_eo = y >= sy_eo && y < ey_eo;
				for (int j = 0; j < 4; j++) {
					const int x = x0 + j;
					const tile_t *cc = c + j;
					const int v = cc[0], d = org.v[j] - v;
					const int sl = sgn3(v - cc[-1]), sr = sgn3(v - cc[1]), su = sgn3(v - cc[-ts]), sd = sgn3(v - cc[ts]);
					const int sul = sgn3(v - cc[-ts - 1]), sdr = sgn3(v - cc[ts + 1]), sur = sgn3(v - cc[-ts + 1]), sdl = sgn3(v - cc[ts - 1]);
					const bool in_x_eo = x >= sx_eo && x < ex_eo;
					const bool in[4] = {in_x_eo && y < ey_full, x < ex_full && in_y_eo, in_x_eo && in_y_eo, in_x_eo && in_y_eo};
					const int k[4] = {2 + sl + sr, 2 + su + sd, 2 + sul + sdr, 2 + sur + sdl};
					const unsigned e = (unsigned)(d + 256);
					for (int t = 0; t < 4; t++) {
						cnt[t] += in[t] ? 1u << (6 * k[t]) : 0u;
						d03[t] += (in[t] && k[t] < 4) ? (unsigned long long)e << (16 * k[t]) : 0ull;
						d4[t] += (in[t] && k[t] == 4) ? e : 0u;
					}
					if (x < ex_full && y < ey_full) {
						post_add_fast(g, &sc.acc[4][0][v >> 3], d);
						post_add_fast(g, &sc.acc[4][1][v >> 3], 1);
					}
				}
			}
			held += 4;
			if (held == 16 || it + 1 == rounds) {
				for (int t = 0; t < 4; t++)
					for (int kk = 0; kk < 5; kk++) {
						const int cl = (int)((cnt[t] >> (6 * kk)) & 63u);
						const int el = kk < 4 ? (int)((d03[t] >> (16 * kk)) & 0xffffu) : (int)d4[t];
						const int ds = (int)g.sum((uint32_t)(el - 256 * cl)), cs2 = (int)g.sum((uint32_t)cl);
						if (g.tid == 0) { sc.acc[t][0][kk] += ds; sc.acc[t][1][kk] += cs2; }
					}
				for (int t = 0; t < 4; t++) { cnt[t] = 0; d4[t] = 0; d03[t] = 0; }
				held = 0;
			}
		}
		g.sync();
		for (int i = g.tid; i < 5 * 2 * 32; i += g.n) (&sc.stats[comp][0][0][0])[i] = (&sc.acc[0][0][0])[i];
		g.sync();
	}
}

// candidate offsets of SAO_MODE_NEW for every (component, type) of the CTU: sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion
// (hmr_sao.c:480-659, est_iter_offset :445).  "Lane" l owns a class: 0-31 the bands, 32-51 the 4 x 5 edge classes.
template <class G>
HENC_HDX void post_sao_candidates(const G g, PostScratch &sc, const double *lambdas)
{
	for (int comp = 0; comp < 3; comp++) {
		const double lambda = lambdas[comp];
		for (int l = g.tid; l < 64; l += g.n) {
			const bool bo = l < 32, eo = l >= 32 && l < 52;
			const int type = bo ? 4 : (l - 32) / 5, cls = bo ? l : (l - 32) % 5;
			int q = 0;
			long long d = 0;
			double cost = lambda;
			if (bo || eo) {
				const long long df = sc.stats[comp][type][0][cls], cn = sc.stats[comp][type][1][cls];
				if (cn != 0 && (bo || cls != 2)) {
					const double x = (double)df / (double)cn;
					int v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
					v = v < -7 ? -7 : v > 7 ? 7 : v;
					if (eo && ((cls < 2 && v < 0) || (cls > 2 && v > 0))) v = 0;
					// est_iter_offset: towards zero, keep the cheapest; an offset that never beats lambda alone becomes 0
					double min_cost = lambda;
					for (int it = v; it != 0; it = it > 0 ? it - 1 : it + 1) {
						const int a = it < 0 ? -it : it;
						const long long rate = (bo ? a + 2 : a + 1) - (a == 7 ? 1 : 0);
						const long long dd = cn * it * it - df * it * 2;
						const double c = (double)dd + lambda * (double)rate;
						if (c < min_cost) { min_cost = c; q = it; d = dd; cost = c; }
					}
				}
			}
			if (bo) sc.cost[l] = cost;
			sc.dist[l] = d;
			if (bo) sc.cand_off[comp][4][l] = q;              // (the bands outside the chosen four are cleared below)
			else if (eo) sc.cand_off[comp][type][cls] = q;
		}
		for (int e = g.tid; e < 4 * 27; e += g.n) sc.cand_off[comp][e / 27][5 + e % 27] = 0;      // entries 5..31 of the edge types
		g.sync();
		// the band position: the first minimum of the four-band cost sums, added in the reference's order
		double min_cost = (double)MAX_COST;
		int band = 0;
		for (int i = 0; i < 29; i++) {
			double s = sc.cost[i];
			s += sc.cost[i + 1]; s += sc.cost[i + 2]; s += sc.cost[i + 3];
			if (s < min_cost) { min_cost = s; band = i; }
		}
		for (int l = g.tid; l < 32; l += g.n)
			if (!(l >= band && l < band + 4)) sc.cand_off[comp][4][l] = 0;
		for (int t = g.tid; t < 5; t += g.n) {
			long long s = 0;
			if (t == 4) for (int i = band; i < band + 4; i++) s += sc.dist[i];
			else for (int c = 0; c < 5; c++) s += sc.dist[32 + t * 5 + c];
			sc.cand_dist[comp][t] = s;
			sc.cand_aux[comp][t] = t == 4 ? band : 0;
		}
		g.sync();
	}
}
struct SaoCandFromScratch {
	const PostScratch *sc;
	HENC_HDX int64_t get(int comp, int type, SaoOffset &t) const
	{
		for (int k = 0; k < 32; k++) t.offset[k] = sc->cand_off[comp][type][k];
		t.type_aux = sc->cand_aux[comp][type];
		return sc->cand_dist[comp][type];
	}
};

// sao_offset_ctu (hmr_sao.c:1210, offset_block :960) from the deblocked CTU (the scratch tiles) into the final picture, then reference_picture_border_padding_ctu
// (:1723) of the final one
template <class G>
HENC_NOINLINE HENC_HDX void post_sao_apply_pad(const G g, const Seq &S, const PostPic &P, const PostScratch &sc, const SaoOffset *params, int cx, int cy)
{	cx = uni(cx); cy = uni(cy);

	const int width = S.width, height = S.height;
	const bool la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
	const int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, h = hl >> sh, w = wl >> sh, w4 = w >> 2, st = comp ? S.stride_c : S.stride_y, ts = comp ? POST_TS_C : POST_TS_Y;
		const size_t base = (size_t)((cy * 64) >> sh) * st + ((cx * 64) >> sh);
		const tile_t *t0 = (comp ? sc.tile_c[comp - 1] : sc.tile_y) + ts + 4;
		int16_t *d0 = P.fin[comp] + base;
		const SaoOffset &p = params[comp];
		const int on = S.sao && p.mode_idc != SAO_OFF, type = p.type_idc;
		const int dx0 = type == 1 ? 0 : (type == 3 ? 1 : -1), dy0 = type == 0 ? 0 : -1;   // second neighbour is the mirror image
		const int nb = dy0 * ts + dx0;
		for (int i = g.tid; i < w4 * h; i += g.n) {
			const int y = i / w4, x0 = (i - y * w4) << 2;
			const tile_t *c = t0 + y * ts + x0;
			S4 out = ld4(c);
			if (on) {
				const bool row_ok = !(type != 0 && ((y == 0 && !ta) || (y == h - 1 && !ba)));
				for (int j = 0; j < 4; j++) {
					const int x = x0 + j, v = c[j];
					int k = -1;
					if (type == SAO_BO) k = v >> 3;
					else if (row_ok && !(type != 1 && ((x == 0 && !la) || (x == w - 1 && !ra)))) k = 2 + sgn3(v - c[j + nb]) + sgn3(v - c[j - nb]);
					if (k >= 0) out.v[j] = (int16_t)hclip(v + p.offset[k], 0, 255);
				}
			}
			st4(d0 + (size_t)y * st + x0, out);
		}
	}
	if (la && ta && ra && ba) return;
	g.sync();
	// margins next to a border CTU: every margin sample takes the nearest picture sample, which lies in this CTU
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, st = comp ? S.stride_c : S.stride_y, m = comp ? S.margin_c : S.margin_y;
		const int pw = width >> sh, ph = height >> sh;
		const int x0 = (cx * 64) >> sh, y0 = (cy * 64) >> sh, x1 = x0 + (wl >> sh), y1 = y0 + (hl >> sh);
		const int ex0 = la ? x0 : -m, ex1 = ra ? x1 : pw + m, ey0 = ta ? y0 : -m, ey1 = ba ? y1 : ph + m;
		int16_t *pic = P.fin[comp];
		const int ew = ex1 - ex0;
		for (int i = g.tid; i < ew * (ey1 - ey0); i += g.n) {
			const int y = ey0 + i / ew, x = ex0 + i % ew;
			if (x >= 0 && x < pw && y >= 0 && y < ph) continue;
			pic[(ptrdiff_t)y * st + x] = pic[(ptrdiff_t)hclip(y, 0, ph - 1) * st + hclip(x, 0, pw - 1)];
		}
	}
}

// ---- the tasks ---------------------------------------------------------------------------------------------------------------------------------------------------
struct PostCtx {               // what a task needs of the picture it belongs to
	const Seq *seq;
	const FrameCtx *f;
	const DevTables *T;
	GeoTable geo;
	CtuInfo *ctus;
	const int16_t *coeff;      // [nctu][6144]
	const PostPic *pic;
};

template <class G>
HENC_TASK_FN HENC_HDX void post_task_d(const G g, const PostCtx &x, int r, int c)
{	r = uni(r); c = uni(c);

	const Seq &S = *x.seq;
	const PostPic &P = *x.pic;
	const int W = S.wctu;
	PPF_T0();
	PPF_COUNT(P, PPF_D_COUNT);
	post_units_ctu(g, P, x.ctus[r * W + c], c, r);
	post_copy_ctu(g, S, *x.f, P, c, r);
	g.sync();
	post_deblock_ctu(g, S, P, c, r, 0);
	g.sync();
	if (c > 0) post_deblock_ctu(g, S, P, c - 1, r, 1);
	if (c == W - 1) {
		g.sync();
		post_deblock_ctu(g, S, P, c, r, 1);
	}
	g.sync();
	PPF_LAP(P, PPF_D);
}

template <class G>
HENC_TASK_FN HENC_HDX void post_task_p(const G g, const PostCtx &x, PostScratch &sc, int r, int c)
{	r = uni(r); c = uni(c);

	const Seq &S = *x.seq;
	const FrameCtx &f = *x.f;
	const PostPic &P = *x.pic;
	const int W = S.wctu, H = S.hctu, n = r * W + c;
	CtuInfo *home = x.ctus + n;
	PPF_T0();
	PPF_COUNT(P, PPF_P_COUNT);
	lin_copy_words(g, (const uint32_t *)(const CtuPublic *)home, (uint32_t *)&sc.c, (int)(sizeof(CtuPublic) / 4));
	lin_copy_words(g, (const uint32_t *)(x.coeff + (size_t)n * 6144), (uint32_t *)sc.coef, 6144 / 2);
	const int row = S.wpp ? r : 0;
	RowEnt &re = P.ent[row];
	Cabac ee;
	BitWriter &bw = ee.bw;
	ee.ctx = sc.ctx;
	bw.attach(P.bs + (size_t)row * P.row_cap, S.wpp ? P.row_cap : P.row_cap * S.hctu);      // (without WPP the picture is ONE sub-stream: it has the whole allocation, hctu x row_cap)
	// wfpp_encode_select_bitstream :2299
	const bool fresh = n == 0 || (S.wpp && c == 0);
	if (n == 0) ee.init_contexts(g, f.slice_type, f.qp);
	else {
		const uint8_t *from = (S.wpp && c == 0) ? P.ent[r - 1].saved : re.ctx;
		for (int i = g.tid; i < CTX_TOTAL; i += g.n) sc.ctx[i] = from[i];
	}
	if (fresh) { ee.start(); ee.reset_bits(); }
	else {
		ee.low = uni(re.low); ee.range = uni(re.range); ee.buffered_byte = uni(re.buffered_byte); ee.num_buffered = uni(re.num_buffered); ee.bits_left = uni(re.bits_left);
		bw.bytecnt = uni(re.bytecnt);
	}
	g.sync();
	PPF_LAP(P, PPF_P_LOAD);
	const int bits_before = bw.bitcount();
	if (S.sao) {
		post_stage_tiles(g, S, P, sc, c, r);
		const double *lam2 = P.sao_lambda + 2 * hclip((int)sc.c.qp[0], 0, 51);
		const double lambdas[3] = {lam2[0], lam2[1], lam2[1]};
		post_sao_stats(g, S, f, sc, c, r);
		PPF_LAP(P, PPF_P_STATS);
		post_sao_candidates(g, sc, lambdas);
		PPF_LAP(P, PPF_P_CAND);
		const SaoTables T = {kEntropyBits, kNextStateLps};
		const SaoCandFromScratch cand = {&sc};
		sao_decide(T, sc.ctx[CTX_SAO_MERGE], sc.ctx[CTX_SAO_TYPE], cand, sc.stats, c > 0 ? (home - 1)->sao_recon : nullptr, r > 0 ? (home - W)->sao_recon : nullptr, lambdas,
			   sc.c.sao_coded, sc.c.sao_recon, sc.sao_ws);
		g.sync();
		// the neighbours' decisions read the parameters from the record
		lin_copy_words(g, (const uint32_t *)sc.c.sao_recon, (uint32_t *)home->sao_recon, (int)(2 * 3 * sizeof(SaoOffset) / 4));
		ee.load_ctx(g);
		code_sao_blk_param(ee, sc.c.sao_coded, c > 0, r > 0);
		PPF_LAP(P, PPF_P_DECIDE);
	} else ee.load_ctx(g);
	EntView &v = sc.view;
	v.seq = x.seq; v.f = x.f; v.T = x.T; v.geo = x.geo;
	sc.view_c = view_of(sc.c); sc.view_l = view_of(*(c > 0 ? home - 1 : home)); sc.view_t = view_of(*(r > 0 ? home - W : home));
	v.c = &sc.view_c;
	v.left = c > 0 ? &sc.view_l : nullptr;
	v.top = r > 0 ? &sc.view_t : nullptr;
	v.coeff[0] = sc.coef; v.coeff[1] = sc.coef + 4096; v.coeff[2] = sc.coef + 5120;
	v.n = n;
	v.prev_last_qp = (n > 0 && !(S.wpp && c == 0)) ? uni((int)(home - 1)->qp[(home - 1)->last_valid_partition]) : -1;
	g.sync();
	encode_ctu_syntax(g, ee, v, sc.ent);
	ee.store_ctx(g);
	if (P.ctx_after)
		for (int i = g.tid; i < CTX_TOTAL; i += g.n) P.ctx_after[(size_t)n * RD_CTX_BYTES + i] = sc.ctx[i];
	const uint32_t bits = (uint32_t)(bw.bitcount() - bits_before);
	if (S.bitrate_mode != 0) {      // the QPs the delta-QP rules rewrote (ee_encode_ctu :2091-2104): the next CTU's predictor reads them
		g.sync();
		lin_copy_words(g, (const uint32_t *)sc.c.qp, (uint32_t *)home->qp, NPART / 4);
	}
	if (c == 1 && r + 1 != H && S.wpp)
		for (int i = g.tid; i < CTX_TOTAL; i += g.n) re.saved[i] = sc.ctx[i];
	if ((S.wpp && c + 1 == W) || (!S.wpp && n + 1 == S.nctu)) {
		ee.encode_trm(1);
		ee.finish();
		bw.trailing_bits();
	}
	for (int i = g.tid; i < CTX_TOTAL; i += g.n) re.ctx[i] = sc.ctx[i];
	if (g.tid == 0) {
		re.low = ee.low; re.range = ee.range; re.buffered_byte = ee.buffered_byte; re.num_buffered = ee.num_buffered; re.bits_left = ee.bits_left;
		re.bytecnt = bw.bytecnt;
		P.cumbits[n] = (c > 0 ? P.cumbits[n - 1] : 0u) + bits;
		if (bw.overflow) P.errors[0] = 1;
	}
	PPF_LAP(P, PPF_P_SYNTAX);
#if defined(HENC_POST_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
	ENT_PROF_ADD(2, ee.nbins);
#endif
	// the filter output of this CTU: SAO offsets applied to the deblocked samples, margins of border CTUs (without SAO: task F)
	if (S.sao) post_sao_apply_pad(g, S, P, sc, sc.c.sao_recon, c, r);
	g.sync();
	PPF_LAP(P, PPF_P_APPLY);
}

// without SAO: the deblocked CTU into the final picture, margins of border CTUs
template <class G>
HENC_TASK_FN HENC_HDX void post_task_f(const G g, const PostCtx &x, PostScratch &sc, int r, int c)
{	r = uni(r); c = uni(c);

	post_stage_tiles(g, *x.seq, *x.pic, sc, c, r);
	post_sao_apply_pad(g, *x.seq, *x.pic, sc, sc.c.sao_recon, c, r);
	g.sync();
}

// ---- the scheduler -----------------------------------------------------------------------------------------------------------------------------------------------
HENC_INLINE int post_min(int a, int b) { return a < b ? a : b; }

// which task of the picture could run now: every lane looks at one row (which of its chains could move?), the first row with one wins.  false: nothing is ready
template <class G>
HENC_HDX inline bool post_scan(const G g, const PostCtx &x, int &pick_r, int &pick_c, int &pick_kind)
{
	const Seq &S = *x.seq;
	const PostPic &P = *x.pic;
	const int W = S.wctu, H = S.hctu;
	pick_r = -1; pick_c = 0; pick_kind = 0;
		for (int base = 0; base < H && pick_r < 0; base += g.n) {
			const int r = base + g.tid;
			bool d_ok = false, p_ok = false, f_ok = false;
			int dc = 0, pc = 0, fc = 0;
			if (r < H) {
				const PostRow &me = P.rows[r];
				dc = post_ld(&me.d_done);
				pc = post_ld(&me.p_done);
				const int below = r + 1 < H ? post_ld(&P.rows[r + 1].d_done) : W;
				if (dc < W && post_ld(&me.d_claim) == dc && post_ld(&me.dec) >= dc + 1 && (r == 0 || post_ld(&P.rows[r - 1].d_done) >= post_min(dc + 1, W))) d_ok = true;
				// rate control without SAO: the reference codes a CTU before anything deblocks it (:2611), and coding rewrites the QP of CUs without levels
				// (ee_encode_ctu :2091-2104) - the filter must see those
				if (d_ok && !S.sao && P.rc_need && pc < dc + 1) d_ok = false;
				if (pc < W && post_ld(&me.p_claim) == pc) {
					// horizontal edges of (r, c + 1) and (r + 1, c + 1) done: D(., c + 2) has run - or D of the row's last CTU.  Without SAO: the CTU decided
					const int need = post_min(pc + 3, W);
					p_ok = (S.sao ? dc >= need && below >= need : post_ld(&me.dec) >= pc + 1) && (r == 0 || post_ld(&P.rows[r - 1].p_done) >= post_min(pc + 2, W));
					if (p_ok && !S.wpp && r > 0 && pc == 0) p_ok = post_ld(&P.rows[r - 1].p_done) >= W;      // one sub-stream: raster order
				}
				if (!S.sao) {
					fc = post_ld(&me.f_done);
					const int need = post_min(fc + 3, W);
					f_ok = fc < W && post_ld(&me.f_claim) == fc && dc >= need && below >= need && (r == 0 || post_ld(&P.rows[r - 1].f_done) >= post_min(fc + 2, W));
				}
			}
			bool s_ok = false;
			int sc_ = 0;
#if defined(__HIP_DEVICE_COMPILE__)
			if (r < H && P.planes[0]) {
				// S(r, c): the final samples of the CTUs around (r, c) - the filters reach four samples out
				const PostRow &me = P.rows[r];
				sc_ = post_ld(&me.s_done);
				const int need = post_min(sc_ + 2, W);
				const int *fin_done = S.sao ? &me.p_done : &me.f_done;
				const ptrdiff_t row_ints = (ptrdiff_t)(sizeof(PostRow) / sizeof(int));
				s_ok = sc_ < W && post_ld(&me.s_claim) == sc_ && post_ld(fin_done) >= need && (r == 0 || post_ld(fin_done - row_ints) >= need) && (r + 1 >= H || post_ld(fin_done + row_ints) >= need);
			}
#endif
			const uint64_t pm = g.ballot(p_ok), dm = g.ballot(d_ok), fm = g.ballot(f_ok), sm = g.ballot(s_ok);
			if (pm) {
				const int lane = __builtin_ctzll(pm);
				pick_r = base + lane; pick_kind = 1;
#if defined(__HIP_DEVICE_COMPILE__)
				pick_c = __builtin_amdgcn_readlane(pc, lane);
#else
				pick_c = pc;
#endif
			} else if (dm) {
				const int lane = __builtin_ctzll(dm);
				pick_r = base + lane; pick_kind = 0;
#if defined(__HIP_DEVICE_COMPILE__)
				pick_c = __builtin_amdgcn_readlane(dc, lane);
#else
				pick_c = dc;
#endif
			} else if (fm) {
				const int lane = __builtin_ctzll(fm);
				pick_r = base + lane; pick_kind = 2;
#if defined(__HIP_DEVICE_COMPILE__)
				pick_c = __builtin_amdgcn_readlane(fc, lane);
#else
				pick_c = fc;
#endif
			} else if (sm) {
				const int lane = __builtin_ctzll(sm);
				pick_r = base + lane; pick_kind = 3;
#if defined(__HIP_DEVICE_COMPILE__)
				pick_c = __builtin_amdgcn_readlane(sc_, lane);
#else
				pick_c = sc_;
#endif
			}
		}
	return pick_r >= 0;
}
// Runs tasks of the picture until none is ready (other workers may be running some: whoever finishes a task looks again).  Returns the number of tasks it ran.
template <class G>
HENC_HDX int post_drain(const G g, const PostCtx &x, PostScratch &sc)
{
	const Seq &S = *x.seq;
	const PostPic &P = *x.pic;
	int ran = 0;
	PPF_T0();
	(void)S;
	for (;;) {
		int pick_r, pick_c, pick_kind;
		post_scan(g, x, pick_r, pick_c, pick_kind);
		PPF_LAP(P, PPF_SCAN);
		if (pick_r < 0) return ran;
		PostRow &row = P.rows[pick_r];
		int *claim = pick_kind == 1 ? &row.p_claim : (pick_kind == 2 ? &row.f_claim : (pick_kind == 3 ? &row.s_claim : &row.d_claim));
		int *done = pick_kind == 1 ? &row.p_done : (pick_kind == 2 ? &row.f_done : (pick_kind == 3 ? &row.s_done : &row.d_done));
		if (!post_claim(g, claim, pick_c)) continue;      // somebody else took it: look again
		post_acquire();
		if (pick_kind == 1) post_task_p(g, x, sc, pick_r, pick_c);
		else if (pick_kind == 2) post_task_f(g, x, sc, pick_r, pick_c);
#if defined(__HIP_DEVICE_COMPILE__)
		else if (pick_kind == 3) subpel_task_ctu(g.tid, *(SubpelScratch *)&sc, S, P.fin, P.planes[0], P.planes[1], P.planes[2], pick_c, pick_r);
#endif
		else post_task_d(g, x, pick_r, pick_c);
		g.sync();      // (every lane's stores of the task before the fence: on a wavefront the fence is wave-wide, a group of another width has to say so)
		post_release();
		if (g.tid == 0) post_st_fenced(done, pick_c + 1);
		g.sync();
		ran++;
#if defined(__HIP_DEVICE_COMPILE__) && defined(HENC_POST_PROFILE)
		ppf_t_ = __builtin_amdgcn_s_memtime();
#endif
	}
}

// ---- rate control: what the decisions with index k (the wavefront step; in raster order the CTU) see of the frame so far (enc_rc.h) ---------------------------
// the bits and the number of the CTUs the reference has entropy coded when they start
template <class G>
HENC_HDX void rc_consumed(const G g, const PostPic &P, int W, int H, int k, uint32_t *bits, int *ctus)
{
	uint32_t b = 0, n = 0;
	for (int r = g.tid; r < H; r += g.n) {
		const int cnt = P.rc_need[(size_t)k * H + r];
		n += (uint32_t)cnt;
		if (cnt) b += P.cumbits[r * W + cnt - 1];
	}
	*bits = g.sum(b);
	*ctus = (int)g.sum(n);
}
// have those CTUs been coded here?
template <class G>
HENC_HDX bool rc_ready(const G g, const PostPic &P, int H, int k)
{
	bool ok = true;
	for (int base = 0; base < H; base += g.n) {
		const int r = base + g.tid;
		const bool miss = r < H && post_ld(&P.rows[r].p_done) < (int)P.rc_need[(size_t)k * H + r];
		if (g.any(miss)) ok = false;
	}
	return ok;
}

// all P tasks of the picture done?
HENC_INLINE bool post_finished(const Seq &S, const PostPic &P)
{
	if (!(post_ld(&P.rows[S.hctu - 1].p_done) >= S.wctu && (S.sao || post_ld(&P.rows[S.hctu - 1].f_done) >= S.wctu))) return false;
	if (P.planes[0])      // (S tasks of different rows are not ordered: every row)
		for (int r = 0; r < S.hctu; r++)
			if (post_ld(&P.rows[r].s_done) < S.wctu) return false;
	return true;
}

}  // namespace henc
