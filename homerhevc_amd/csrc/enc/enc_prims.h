// Block primitives of the CTU encoder in SPMD form: the lanes of the group share one block.
// Each primitive cites the reference kernel it restates; the arithmetic is the one pinned in round 1
// (oracle/hmr_oracle.c against the compiled reference, tests/test_oracle_vs_ref.py).
// Convention: inputs must be visible to the group on entry (caller synced), outputs are visible on return.
#pragma once
#include "enc_types.h"
#include "../tables_layout.h"

// Primitive-level timers of the profiling build (-DHENC_PROFILE, device only): lane 0 adds s_memtime ticks and a call count per primitive class
// to a small table at the end of the worker's LDS; k_encode_ctus folds it into the per-row profile.
enum { PP_SAD = 0, PP_SSD, PP_BLK, PP_FILLREF, PP_ADIFILT, PP_INTRAPRED, PP_INTERP, PP_TRF, PP_TRI, PP_QUANT, PP_DEQUANT, PP_CAND, PP_SYNC, PP_INFO, PP_CTU_IO, PP_HWAIT, PP_COUNT };
#if defined(__HIPCC__) && defined(HENC_PROFILE)
extern __shared__ __align__(16) unsigned char henc_lds[];
#define HENC_LDS_PROF_OFFSET 43008      // (behind the worker state: the profiling build holds three workers per CU where the product holds four; k_encode.hip checks the place)
#define PRIM_T0() const unsigned long long prim_t0_ = __builtin_amdgcn_s_memtime()
#define PRIM_END(cat) do { if (threadIdx.x == 0) { unsigned long long *pp_ = (unsigned long long *)(henc_lds + HENC_LDS_PROF_OFFSET); pp_[cat] += __builtin_amdgcn_s_memtime() - prim_t0_; pp_[PP_COUNT + (cat)]++; } } while (0)
#else
#define PRIM_T0() do { } while (0)
#define PRIM_END(cat) do { } while (0)
#endif

namespace henc {

HENC_INLINE int ilog2i(int n)   // smallest s with (1 << s) >= n
{
	return n <= 1 ? 0 : 32 - __builtin_clz((unsigned)(n - 1));
}
// row / column of element k of a block w wide without a division when w is a power of two (it is, except at picture edges)
HENC_INLINE void split_rc(int k, int w, int lw, int *r, int *c)
{
	if ((w & (w - 1)) == 0) { *r = k >> lw; *c = k & (w - 1); }
	else { *r = k / w; *c = k - *r * w; }
}

// ---- pixel kernels (hmr_sse42_functions_pixel.c:462,728,817,919) -------------------------------------------------
// A CU's memory pipeline is shared by its wavefronts and a 2-byte-per-lane access costs it as much as an 8-byte one, so every block loop moves
// FOUR samples per lane and step (one 8-byte access; blocks are at least 4 wide and start at multiples of 4 samples in rows whose pitch is a
// multiple of 4, except the motion search's reference operand, which may start anywhere: gfx950 runs with unaligned global access enabled).
struct S4 { int16_t v[4]; };
HENC_INLINE S4 ld4(const int16_t *p) { S4 r; __builtin_memcpy(&r, p, 8); return r; }
HENC_INLINE void st4(int16_t *p, const S4 &v) { __builtin_memcpy(p, &v, 8); }
// four source samples (enc_types.h src_t: bytes on the device) as 16-bit values
HENC_INLINE S4 ld4(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	S4 r;
	r.v[0] = (int16_t)(v & 255); r.v[1] = (int16_t)((v >> 8) & 255); r.v[2] = (int16_t)((v >> 16) & 255); r.v[3] = (int16_t)(v >> 24);
	return r;
}

// four 16-bit sample values (0 .. 255) into a byte window
HENC_INLINE void st4(uint8_t *p, const S4 &v)
{
	const uint32_t o = (uint32_t)(v.v[0] & 255) | ((uint32_t)(v.v[1] & 255) << 8) | ((uint32_t)(v.v[2] & 255) << 16) | ((uint32_t)(v.v[3] & 255) << 24);
	__builtin_memcpy(p, &o, 4);
}

template <class G, class S>
HENC_PRIM uint32_t blk_sad(const G g, const S *a, int as, const int16_t *b, int bs, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
	uint32_t acc = 0;
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 va = ld4(a + y * as + x), vb = ld4(b + y * bs + x);
#pragma unroll
		for (int k = 0; k < 4; k++) acc += (uint32_t)habs((int16_t)(va.v[k] - vb.v[k]));
	}
	{ const auto prim_ret_ = g.sum(acc); PRIM_END(PP_SAD); return prim_ret_; }
}

#if defined(__HIPCC__)
// ---- 8-bit reference planes (k_subpel.hip) ----------------------------------------------------------------------------
// The reference picture is kept as sixteen 8-bit luma planes (one per quarter-sample phase) and sixty-four chroma planes per component, built once per
// frame by a bandwidth-bound kernel: inside the CTU walk a motion-compensated block is a copy and a motion-search candidate is a SAD against a plane -
// no interpolation on the serial path.  Blocks start at any byte of a plane (gfx950 runs with unaligned global access enabled).
HENC_INLINE uint32_t ld32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// SADs of up to MAXC candidate blocks (global, 8 bit, row pitch `stride`) against one source block (8 bit, row pitch 64); cand[k] == nullptr: skipped.
// Every candidate's loads are issued before the first result is needed, so a round of candidates costs one memory latency.
// (The profiling build times it at the call, enc_inter.h cand_sads: with the timer in here the compiler fails with "illegal VGPR to SGPR copy".)
template <int MAXC>
__device__ __forceinline__ void multi_sad_u8(const WaveGrp g, const uint8_t *orig8, int n, const uint8_t *const (&cand)[MAXC], int stride, uint32_t (&out)[MAXC])
{
	if (n == 8) {
		// 16 four-sample chunks per block: four candidates side by side, one per 16-lane row; the row totals are what WaveGrp::sum adds up last
		const int sub = g.tid >> 4, r = (g.tid & 15) >> 1, c = (g.tid & 1) << 2;
		const uint32_t a = *(const uint32_t *)(orig8 + r * 64 + c);
#pragma unroll
		for (int k0 = 0; k0 < MAXC; k0 += 4) {
			// (each candidate pointer goes through an empty asm: the compiler otherwise turns this chain of selects into ONE load at cand[k0 + sub] - an array
			// indexed at run time lives in private memory, a store and a dependent load through L2 per round of the search)
			uint64_t cp[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				cp[j] = k0 + j < MAXC ? (uint64_t)(uintptr_t)cand[k0 + j] : 0;
				asm("" : "+v"(cp[j]));
			}
			uint64_t ps = cp[0];
#pragma unroll
			for (int j = 1; j < 4; j++)
				if (k0 + j < MAXC) ps = sub == j ? cp[j] : ps;
			const uint8_t *p = (const uint8_t *)(uintptr_t)ps;
			if (k0 + 4 > MAXC && sub >= MAXC - k0) p = nullptr;
			int x = p ? (int)__builtin_amdgcn_sad_u8(a, ld32u(p + r * stride + c), 0u) : 0;
			x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
			x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
#pragma unroll
			for (int j = 0; j < 4; j++)
				if (k0 + j < MAXC) out[k0 + j] = (uint32_t)__builtin_amdgcn_readlane(x, 16 * j + 15);
		}
		return;
	}
	const int lw = ilog2i(n) - 2, chunks = (n * n) >> 2;       // n >= 16: at least one chunk per lane
	uint32_t acc[MAXC];
#pragma unroll
	for (int k = 0; k < MAXC; k++) acc[k] = 0;
	for (int i = g.tid; i < chunks; i += 64) {
		const int r = i >> lw, c = (i & ((1 << lw) - 1)) << 2;
		const uint32_t a = *(const uint32_t *)(orig8 + r * 64 + c);
		uint32_t b[MAXC];
#pragma unroll
		for (int k = 0; k < MAXC; k++) b[k] = cand[k] ? ld32u(cand[k] + r * stride + c) : a;
#pragma unroll
		for (int k = 0; k < MAXC; k++) acc[k] = __builtin_amdgcn_sad_u8(a, b[k], acc[k]);
	}
#pragma unroll
	for (int k = 0; k < MAXC; k++) out[k] = g.sum(acc[k]);
}

// n x n samples of an 8-bit plane into the (8-bit) prediction window (motion compensation from the phase planes); the caller syncs
__device__ __forceinline__ void blk_from_u8(const WaveGrp g, const uint8_t *s, int ss, uint8_t *d, int ds, int n)
{
	const int lw = ilog2i(n) - 2, chunks = (n * n) >> 2;
#pragma unroll 4
	for (int i = g.tid; i < chunks; i += 64) {
		const int r = i >> lw, c = (i & ((1 << lw) - 1)) << 2;
		*(uint32_t *)(d + r * ds + c) = ld32u(s + r * ss + c);
	}
}
#endif

template <class G, class S, class P>
HENC_PRIM uint32_t blk_ssd(const G g, const S *a, int as, const P *b, int bs, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
	uint32_t acc = 0;
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 va = ld4(a + y * as + x), vb = ld4(b + y * bs + x);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int32_t d = (int16_t)(va.v[k] - vb.v[k]);
			acc += (uint32_t)(d * d);
		}
	}
	{ const auto prim_ret_ = g.sum(acc); PRIM_END(PP_SSD); return prim_ret_; }
}

// SSD of the residual source - prediction against a reconstructed residual, the residual formed on the way (16-bit wrap like the reference's predict kernel
// writes it): what ssd16b(residual window, reconstructed residual) gives, without the window
template <class G, class S, class P>
HENC_PRIM uint32_t blk_ssd_diff(const G g, const S *o, int os, const P *p, int ps, const int16_t *r, int rs, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
	uint32_t acc = 0;
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 vo = ld4(o + y * os + x), vp = ld4(p + y * ps + x), vr = ld4(r + y * rs + x);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int16_t res = (int16_t)(vo.v[k] - vp.v[k]);
			const int32_t d = (int16_t)(res - vr.v[k]);
			acc += (uint32_t)(d * d);
		}
	}
	{ const auto prim_ret_ = g.sum(acc); PRIM_END(PP_SSD); return prim_ret_; }
}

// modified_variance (sse_modified_variance, hmr_sse42_functions_pixel.c:1123; quirk Q2 of SURVEY.md: the SSE code loads 16-bit rows and zero-extends their BYTES - per
// row it consumes `size` bytes: the low and high bytes of half of the samples, for size >= 16 bytes [32 g, 32 g + 8) and [32 g + 16, 32 g + 24) of every
// 16-sample group g; oracle/hmr_oracle.c ora_modified_variance).  The source window holds the samples themselves: byte b of a row is sample b / 2 for even b, 0 for odd b
// (a sample's high byte).
template <class G, class S>
HENC_PRIM uint32_t blk_modified_variance(const G g, const S *p, int stride, int size, int modif)
{
	PRIM_T0();
	const int l = ilog2i(size), total = size * size;
	auto byte_at = [&](int i) -> int {
		const int j = i >> l, k = i & (size - 1);
		const int off = size < 16 ? k : (k >> 4) * 32 + ((k >> 3) & 1) * 16 + (k & 7);
		const int v = (int)p[j * stride + (off >> 1)];
		return (off & 1) ? ((v >> 8) & 255) : (v & 255);
	};
	uint32_t acc = 0;
	for (int i = g.tid; i < total; i += g.n) acc += (uint32_t)byte_at(i);
	const int avg = (int)(g.sum(acc) / (uint32_t)total);
	acc = 0;
	for (int i = g.tid; i < total; i += g.n) {
		const int16_t d = (int16_t)(1 + (int16_t)((int16_t)(byte_at(i) - avg) * (int16_t)modif));
		acc += (uint32_t)((int32_t)d * d);
	}
	{ const auto prim_ret_ = g.sum(acc); PRIM_END(PP_SSD); return prim_ret_; }
}

// sum of squares of a block (ssd16b against the reference's zero row, hmr_motion_inter.c:94)
template <class G>
HENC_PRIM uint32_t blk_ssq(const G g, const int16_t *a, int as, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
	uint32_t acc = 0;
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const S4 va = ld4(a + (i >> l) * as + (i & (n - 1)));
#pragma unroll
		for (int k = 0; k < 4; k++) acc += (uint32_t)((int32_t)va.v[k] * va.v[k]);
	}
	{ const auto prim_ret_ = g.sum(acc); PRIM_END(PP_SSD); return prim_ret_; }
}

template <class G, class S, class P>
HENC_PRIM void blk_predict(const G g, const S *o, int os, const P *p, int ps, int16_t *r, int rs, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 vo = ld4(o + y * os + x), vp = ld4(p + y * ps + x);
		S4 vr;
#pragma unroll
		for (int k = 0; k < 4; k++) vr.v[k] = (int16_t)(vo.v[k] - vp.v[k]);
		st4(r + y * rs + x, vr);
	}
	g.sync();
	PRIM_END(PP_BLK);
}

// res == nullptr: the all-zero residual (the reference passes a zeroed row with stride 0, hmr_motion_intra.c:1065)
template <class G, class P>
HENC_PRIM void blk_reconst(const G g, const P *p, int ps, const int16_t *res, int rs, int16_t *d, int ds, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 vp = ld4(p + y * ps + x);
		S4 vr = {{0, 0, 0, 0}}, vd;
		if (res) vr = ld4(res + y * rs + x);
#pragma unroll
		for (int k = 0; k < 4; k++) vd.v[k] = (int16_t)hclip((int)sat16(vp.v[k] + vr.v[k]), 0, 255);
		st4(d + y * ds + x, vd);
	}
	g.sync();
	PRIM_END(PP_BLK);
}

// reconstruction and its distance from the source in one pass (the reference reconstructs, then reads the window back for ssd16b, hmr_motion_intra.c:1061-1068)
template <class G, class S, class P>
HENC_PRIM uint32_t blk_reconst_ssd(const G g, const P *p, int ps, const int16_t *res, int rs, const S *o, int os, int16_t *d, int ds, int n)
{
	PRIM_T0();
	const int l = ilog2i(n);
	uint32_t acc = 0;
#pragma unroll 2
	for (int i = g.tid * 4; i < n * n; i += g.n * 4) {
		const int y = i >> l, x = i & (n - 1);
		const S4 vp = ld4(p + y * ps + x), vo = ld4(o + y * os + x);
		S4 vr = {{0, 0, 0, 0}}, vd;
		if (res) vr = ld4(res + y * rs + x);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			vd.v[k] = (int16_t)hclip((int)sat16(vp.v[k] + vr.v[k]), 0, 255);
			const int32_t df = (int16_t)(vo.v[k] - vd.v[k]);
			acc += (uint32_t)(df * df);
		}
		st4(d + y * ds + x, vd);
	}
	const uint32_t s = g.sum(acc);
	g.sync();
	{ const auto prim_ret_ = s; PRIM_END(PP_BLK); return prim_ret_; }
}

template <class G, class P>
HENC_PRIM void blk_copy(const G g, const P *s, int ss, int16_t *d, int ds, int h, int w)
{
	PRIM_T0();
	if ((w & 3) == 0 && (w & (w - 1)) == 0) {
		const int lw = ilog2i(w);
		// up to four steps in flight: source or destination may be a window in HBM, and a load cannot pass a store the compiler cannot tell apart from it
		constexpr int BATCH = 4;
		for (int i0 = g.tid * 4; i0 < h * w; i0 += g.n * 4 * BATCH) {
			S4 v[BATCH];
#pragma unroll
			for (int u = 0; u < BATCH; u++) {
				const int i = i0 + u * g.n * 4;
				if (i < h * w) v[u] = ld4(s + (i >> lw) * ss + (i & (w - 1)));
			}
#pragma unroll
			for (int u = 0; u < BATCH; u++) {
				const int i = i0 + u * g.n * 4;
				if (i < h * w) st4(d + (i >> lw) * ds + (i & (w - 1)), v[u]);
			}
		}
	} else {
		const int lw = ilog2i(w);
		for (int i = g.tid; i < h * w; i += g.n) {
			int y, x;
			split_rc(i, w, lw, &y, &x);
			d[y * ds + x] = s[y * ss + x];
		}
	}
	g.sync();
	PRIM_END(PP_BLK);
}

// a block of a 16-bit picture plane into the CTU's source buffer (bytes on the device); w a multiple of 4
template <class G>
HENC_PRIM void blk_copy_to_src(const G g, const int16_t *s, int ss, src_t *d, int ds, int h, int w)
{
	PRIM_T0();
	const int cw = w >> 2;
	for (int i = g.tid; i < h * cw; i += g.n) {
		const int y = i / cw, x = (i - y * cw) << 2;
		const S4 v = ld4(s + y * ss + x);
#if defined(__HIPCC__)
		*(uint32_t *)(d + y * ds + x) = (uint32_t)(v.v[0] & 255) | ((uint32_t)(v.v[1] & 255) << 8) | ((uint32_t)(v.v[2] & 255) << 16) | ((uint32_t)(v.v[3] & 255) << 24);
#else
		st4(d + y * ds + x, v);
#endif
	}
	g.sync();
	PRIM_END(PP_BLK);
}

// linear copies: 16 bytes per lane and step where both ends allow it, four steps in flight (the CTU's record, its 40 KB of partition nodes and its levels
// move between HBM and the worker's fast memory at every CTU start and end)
typedef uint32_t Q16 __attribute__((vector_size(16), may_alias, aligned(16)));      // (a vector, not a struct of four words: struct temporaries of the copy loops ended up in private memory)
template <class G>
HENC_HD void lin_copy_bytes(const G g, const void *s, void *d, int bytes)
{
	const uintptr_t both = (uintptr_t)s | (uintptr_t)d | (uintptr_t)bytes;
	if ((both & 15) == 0) {
		const Q16 *sq = (const Q16 *)s;
		Q16 *dq = (Q16 *)d;
		const int nq = bytes >> 4;
		int i = g.tid;
		for (; i + 3 * g.n < nq; i += 4 * g.n) {
			const Q16 a = sq[i], b = sq[i + g.n], c = sq[i + 2 * g.n], e = sq[i + 3 * g.n];
			dq[i] = a; dq[i + g.n] = b; dq[i + 2 * g.n] = c; dq[i + 3 * g.n] = e;
		}
		for (; i < nq; i += g.n) dq[i] = sq[i];
	} else if ((both & 3) == 0) {
		const uint32_t *sw = (const uint32_t *)s;
		uint32_t *dw = (uint32_t *)d;
		const int nw = bytes >> 2;
		int i = g.tid;
		for (; i + 3 * g.n < nw; i += 4 * g.n) {
			const uint32_t a = sw[i], b = sw[i + g.n], c = sw[i + 2 * g.n], e = sw[i + 3 * g.n];
			dw[i] = a; dw[i + g.n] = b; dw[i + 2 * g.n] = c; dw[i + 3 * g.n] = e;
		}
		for (; i < nw; i += g.n) dw[i] = sw[i];
	} else {
		const uint16_t *sh = (const uint16_t *)s;
		uint16_t *dh = (uint16_t *)d;
		for (int i = g.tid; i < (bytes >> 1); i += g.n) dh[i] = sh[i];
	}
}
template <class G>
HENC_PRIM void lin_copy(const G g, const int16_t *s, int16_t *d, int count)
{
	PRIM_T0();
	lin_copy_bytes(g, s, d, count * 2);
	g.sync();
	PRIM_END(PP_BLK);
}

// the same without the closing sync: stores to a window nobody reads before the chain's next sync (the levels of a TU on their way to HBM)
template <class G>
HENC_PRIM void lin_copy_nosync(const G g, const int16_t *s, int16_t *d, int count)
{
	lin_copy_bytes(g, s, d, count * 2);
}
template <class G>
HENC_PRIM void lin_zero_nosync(const G g, int16_t *d, int count)
{
	if ((((uintptr_t)d | (uintptr_t)(count * 2)) & 7) == 0) {
		const S4 z = {{0, 0, 0, 0}};
		for (int i = g.tid * 4; i < count; i += g.n * 4) st4(d + i, z);
	} else
		for (int i = g.tid; i < count; i += g.n) d[i] = 0;
}

template <class G>
HENC_PRIM void lin_copy_words(const G g, const uint32_t *s, uint32_t *d, int count)
{
	PRIM_T0();
	lin_copy_bytes(g, s, d, count * 4);
	g.sync();
	PRIM_END(PP_BLK);
}

template <class G>
HENC_PRIM void lin_zero(const G g, int16_t *d, int count)
{
	PRIM_T0();
	for (int i = g.tid; i < count; i += g.n) d[i] = 0;
	g.sync();
	PRIM_END(PP_BLK);
}

template <class G>
HENC_PRIM void bytes_set(const G g, uint8_t *d, int v, int count)
{
	PRIM_T0();
	for (int i = g.tid; i < count; i += g.n) d[i] = (uint8_t)v;
	g.sync();
	PRIM_END(PP_BLK);
}

template <class G>
HENC_PRIM void bytes_copy(const G g, const uint8_t *s, uint8_t *d, int count)
{
	PRIM_T0();
	for (int i = g.tid; i < count; i += g.n) d[i] = s[i];
	g.sync();
	PRIM_END(PP_BLK);
}

// ---- intra reference samples (fill_reference_samples hmr_motion_intra.c:246-404, adi_filter :189-244) --------------
// `corner` points at sample (-1,-1) of the block in the window under reconstruction.
template <class G>
HENC_PRIM void intra_fill_refs(const G g, const int16_t *corner, int stride, int n, int left, int top, int bottom_left, int top_right,
			     int bl_size, int tr_size, int16_t *adi)
{
	PRIM_T0();
	const int adi_size = 4 * n + 1;
	if (!left && !top) {
		for (int i = g.tid; i < adi_size; i += g.n) adi[i] = 128;
		g.sync();
		{ PRIM_END(PP_FILLREF); return; }
	}
	int pl_ptr = 0, pl_size = 0, pt_ptr = 0, pt_size = 0, first_idx = 0, last_idx = 0;
	if (left) { first_idx = n; last_idx = 2 * n - 1; }
	else { pl_ptr = n; pl_size = n; }
	if (bottom_left) {
		first_idx = n - bl_size;
		if (bl_size != n) { pl_ptr = 0; pl_size = n - bl_size; }
	} else {
		pl_ptr = 0;
		if (left) pl_size = n;
		else pl_size += n;
	}
	if (top) {
		if (!left) first_idx = 2 * n + 1;
		last_idx = 3 * n;
	} else { pt_ptr = 2 * n + 1; pt_size = n; }
	if (top_right) {
		last_idx = 3 * n + tr_size;
		if (tr_size != n) { pt_ptr = 3 * n + 1 + tr_size; pt_size = n - tr_size; }
	} else {
		if (top) { pt_ptr = 3 * n + 1; pt_size = n; }
		else pt_size += n;
	}
	const bool corner_copy = left && top;
	if (!corner_copy) {
		if (left) { pt_ptr--; pt_size++; }
		else pl_size++;
	}
	// entry k of the array straight from the window, for the entries that are available (first_idx and last_idx always are)
	auto sample = [&](int k) -> int16_t {
		if (k < n) return corner[(n + 1 + (n - 1 - k)) * stride];        // bottom-left: adi[n-1-i] = row n+1+i
		if (k < 2 * n) return corner[(n - (k - n)) * stride];            // left: adi[n+i] = row n-i
		if (k == 2 * n) return corner[0];
		if (k <= 3 * n) return corner[k - 2 * n];                        // top
		return corner[1 + n + (k - 3 * n - 1)];                          // top-right
	};
	const int16_t first_sample = sample(first_idx), last_sample = sample(last_idx);
	// one pass: available samples, and the two substitution runs (the reference copies first, then pads: same result, two barriers fewer)
	for (int k = g.tid; k < adi_size; k += g.n) {
		bool avail;
		if (k < n) avail = bottom_left && (n - 1 - k) < bl_size;
		else if (k < 2 * n) avail = left;
		else if (k == 2 * n) avail = corner_copy;
		else if (k <= 3 * n) avail = top;
		else avail = top_right && (k - 3 * n - 1) < tr_size;
		if (k >= pl_ptr && k < pl_ptr + pl_size) adi[k] = first_sample;
		else if (k >= pt_ptr && k < pt_ptr + pt_size) adi[k] = last_sample;
		else if (avail) adi[k] = sample(k);
	}
	g.sync();
	PRIM_END(PP_FILLREF);
}

template <class G>
HENC_PRIM void intra_adi_filter(const G g, const int16_t *adi, int16_t *out, int n, int strong_enabled)
{
	PRIM_T0();
	const int adi_size = 4 * n + 1;
	bool strong = false;
	int bl = 0, tl = 0, tr = 0;
	if (strong_enabled) {
		bl = adi[0]; tl = adi[2 * n]; tr = adi[adi_size - 1];
		const bool lin_left = habs(bl + tl - 2 * adi[n]) < 8, lin_top = habs(tl + tr - 2 * adi[3 * n]) < 8;
		strong = n >= 32 && lin_left && lin_top;
	}
	if (strong) {
		const int shift = ilog2i(2 * n);
		#pragma unroll 4
		for (int i = g.tid; i < adi_size; i += g.n) {
			int v;
			if (i == 0 || i == 2 * n || i == adi_size - 1) v = adi[i];
			else if (i < 2 * n) v = ((2 * n - i) * bl + i * tl + n) >> shift;
			else { const int k = i - 2 * n; v = ((2 * n - k) * tl + k * tr + n) >> shift; }
			out[i] = (int16_t)v;
		}
	} else {
		for (int i = g.tid; i < adi_size; i += g.n)
			out[i] = (i == 0 || i == adi_size - 1) ? adi[i] : (int16_t)((adi[i - 1] + 2 * adi[i] + adi[i + 1] + 2) >> 2);
	}
	g.sync();
	PRIM_END(PP_ADIFILT);
}

// ---- intra prediction (planar hmr_motion_intra.c:408-439; DC / angular :482-625; SSE twins prediction.c:199,926) -------
// The value of one prediction sample in closed form, so that a search can compare it against the source without storing it.
struct IntraPredictor {
	const int16_t *mid;
	int n, shift, mode, is_luma;
	int kind;                // 0 planar, 1 dc, 2 angular
	int dc, bl, tr;
	int is_ver, angle, inv_angle, edge_filter;
};

// (the DC value: the 2 n samples beside the block, one per lane and a sum over the group - every lane adding all of them itself is 2 n dependent reads)
template <class G>
HENC_INLINE IntraPredictor intra_setup(const G g, const int16_t *adi, int n, int mode, int is_luma)
{
	IntraPredictor p;
	p.mid = adi + 2 * n;
	p.n = n; p.shift = ilog2i(n); p.mode = mode; p.is_luma = is_luma;
	p.dc = p.bl = p.tr = p.is_ver = p.angle = p.inv_angle = p.edge_filter = 0;
	if (mode == PLANAR_IDX) {
		p.kind = 0;
		p.bl = p.mid[-(n + 1)];
		p.tr = p.mid[n + 1];
	} else if (mode == DC_IDX) {
		p.kind = 1;
		int acc = 0;
		for (int i = g.tid; i < 2 * n; i += g.n) acc += i < n ? p.mid[i + 1] : p.mid[-(i - n + 1)];
		acc = (int)g.sum((uint32_t)acc);
		p.dc = (uint8_t)(uint16_t)((acc + n) / (2 * n));
		p.edge_filter = n <= 16 && is_luma;
	} else {
		p.kind = 2;
		static constexpr int ang_table[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
		static constexpr int inv_ang_table[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
		const int is_hor = mode < 18;
		p.is_ver = !is_hor;
		int pa = p.is_ver ? mode - 26 : -(mode - 10);
		const int aa = habs(pa), sign = pa < 0 ? -1 : (pa > 0 ? 1 : 0);
		p.inv_angle = inv_ang_table[aa];
		p.angle = sign * ang_table[aa];
		p.edge_filter = is_luma ? (n <= 16) : 0;
	}
	return p;
}

HENC_INLINE int intra_ref_main(const IntraPredictor &p, int idx)
{
	if (idx >= 0) return p.is_ver ? p.mid[idx] : p.mid[-idx];
	const int k = (128 + (-idx) * p.inv_angle) >> 8;
	return p.is_ver ? p.mid[-k] : p.mid[k];
}

// sample at row j, column i of the prediction block
HENC_INLINE int intra_sample(const IntraPredictor &p, int j, int i)
{
	const int n = p.n;
	if (p.kind == 0) {
		const int left = p.mid[-(j + 1)], top = p.mid[i + 1];
		const int hor = (left << p.shift) + n + (i + 1) * (p.tr - left);
		const int ver = (top << p.shift) + (j + 1) * (p.bl - top);
		return (int16_t)((hor + ver) >> (p.shift + 1));
	}
	if (p.kind == 1) {
		if (p.edge_filter) {
			if (j == 0 && i == 0) return (int16_t)((p.mid[-1] + p.mid[1] + 2 * p.dc + 2) >> 2);
			if (j == 0) return (int16_t)((p.mid[1 + i] + 3 * p.dc + 2) >> 2);
			if (i == 0) return (int16_t)((p.mid[-1 - j] + 3 * p.dc + 2) >> 2);
		}
		return p.dc;
	}
	// angular: horizontal modes are the vertical construction transposed (a = index along the main reference, b = line)
	const int a = p.is_ver ? i : j, b = p.is_ver ? j : i;
	if (p.angle == 0) {
		int v = (uint8_t)intra_ref_main(p, a + 1);
		if (p.edge_filter && a == 0) {
			const int side_b = p.is_ver ? p.mid[-(b + 1)] : p.mid[b + 1], side_0 = p.mid[0];
			v = hclip(v + ((side_b - side_0) >> 1), 0, 255);
		}
		return v;
	}
	const int pos = (b + 1) * p.angle, delta = pos >> 5, fract = pos & 31, idx = a + delta + 1;
	if (fract) return (uint8_t)(((32 - fract) * intra_ref_main(p, idx) + fract * intra_ref_main(p, idx + 1) + 16) >> 5);
	return (uint8_t)intra_ref_main(p, idx);
}

template <class G, class P>
HENC_PRIM void intra_predict(const G g, P *pred, int ps, const int16_t *adi, int n, int mode, int is_luma)
{
	PRIM_T0();
	if constexpr (G::n == 64) { mode = uni(mode); n = uni(n); is_luma = uni(is_luma); }      // (a search's winner arrives in a vector register: what depends on it is scalar again)
	const IntraPredictor p = intra_setup(g, adi, n, mode, is_luma);
	const int l = p.shift;
	#pragma unroll 4
	for (int k = g.tid; k < n * n; k += g.n) {
		const int j = k >> l, i = k & (n - 1);
		pred[j * ps + i] = (P)intra_sample(p, j, i);
	}
	g.sync();
	PRIM_END(PP_INTRAPRED);
}

// prediction + SAD against the source in one pass; the prediction is also stored (later stages of the reference read it)
template <class G, class S, class P>
HENC_PRIM uint32_t intra_predict_sad(const G g, P *pred, int ps, const S *orig, int os, const int16_t *adi, int n, int mode, int is_luma)
{
	PRIM_T0();
	if constexpr (G::n == 64) { mode = uni(mode); n = uni(n); is_luma = uni(is_luma); }
	const IntraPredictor p = intra_setup(g, adi, n, mode, is_luma);
	const int l = p.shift;
	uint32_t acc = 0;
	#pragma unroll 4
	for (int k = g.tid; k < n * n; k += g.n) {
		const int j = k >> l, i = k & (n - 1);
		const int v = intra_sample(p, j, i);
		if (pred) pred[j * ps + i] = (P)v;                // pred == nullptr: the SAD alone (a helper wavefront's candidate)
		acc += (uint32_t)habs((int16_t)(orig[j * os + i] - (int16_t)v));
	}
	const uint32_t s = g.sum(acc);
	g.sync();
	{ const auto prim_ret_ = s; PRIM_END(PP_INTRAPRED); return prim_ret_; }
}

// ---- interpolation (hmr_motion_inter.c:240-391,878-936; SSE twins inter_prediction.c:796,818) -----------------------
HENC_INLINE void luma_tap_row(int f, int *c)
{
	const int t[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
	for (int k = 0; k < 8; k++) c[k] = t[f][k];
}
HENC_INLINE void chroma_tap_row(int f, int *c)
{
	const int t[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
	for (int k = 0; k < 4; k++) c[k] = t[f][k];
}

// one separable stage; NT = 8 (luma) / 4 (chroma).  fraction 0 = the reference's filter_copy variants.  Four outputs per lane and step: a horizontal
// stage reads its 4 + NT - 1 consecutive samples with 8-byte loads (the source may start at any sample: unaligned global access), a vertical one reads
// NT rows of four.
template <int NT, class G>
HENC_PRIM void interp_stage(const G g, const int16_t *src, int ss, int16_t *dst, int ds, int fraction, int w, int h, int vert, int first, int last)
{
	PRIM_T0();
	if (NT == 4 && w < 4 && fraction == 0) { g.sync(); PRIM_END(PP_INTERP); return; }    // chroma no-op (inter_prediction.c:822-825)
	const int lw = ilog2i(w);
	if ((w & 3) != 0 || (fraction != 0 && w * h < 4 * G::n && G::n > 1)) {
		// blocks too small to give every lane four outputs (and 2-wide chroma blocks): one output per lane
		int c8[8];
		if (NT == 8) luma_tap_row(fraction, c8);
		else chroma_tap_row(fraction, c8);
		const int rs = vert ? ss : 1;
		int shift = 6, offset;
		if (last) { shift += first ? 0 : 6; offset = 1 << (shift - 1); offset += first ? 0 : 8192 << 6; }
		else { shift -= first ? 6 : 0; offset = first ? -(8192 << shift) : 0; }
		const int16_t *s0 = src - (NT / 2 - 1) * rs;
		for (int k = g.tid; k < w * h; k += g.n) {
			int r, c;
			split_rc(k, w, lw, &r, &c);
			int sum = 0;
			for (int t = 0; t < NT; t++) sum += s0[r * ss + c + t * rs] * c8[t];
			int16_t v = sat16((sum + offset) >> shift);
			if (last) v = (int16_t)hclip((int)v, 0, 255);
			dst[r * ds + c] = v;
		}
		g.sync();
		PRIM_END(PP_INTERP);
		return;
	}
	if (fraction == 0) {
#pragma unroll 2
		for (int k = g.tid * 4; k < w * h; k += g.n * 4) {
			int r, c;
			split_rc(k, w, lw, &r, &c);
			const S4 v = ld4(src + r * ss + c);
			S4 o;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (first == last) o.v[j] = v.v[j];
				else if (first) o.v[j] = (int16_t)((int16_t)(v.v[j] << 6) - 8192);
				else o.v[j] = (int16_t)hclip((v.v[j] + 8192 + 32) >> 6, 0, 255);
			}
			st4(dst + r * ds + c, o);
		}
		g.sync();
		PRIM_END(PP_INTERP);
		return;
	}
	int c8[8];
	if (NT == 8) luma_tap_row(fraction, c8);
	else chroma_tap_row(fraction, c8);
	int shift = 6, offset;
	if (last) {
		shift += first ? 0 : 6;
		offset = 1 << (shift - 1);
		offset += first ? 0 : 8192 << 6;
	} else {
		shift -= first ? 6 : 0;
		offset = first ? -(8192 << shift) : 0;
	}
	if (!vert) {
		constexpr int NS = NT == 8 ? 12 : 8;                  // samples a lane reads: 4 outputs + NT - 1 taps, rounded up to whole loads
		const int16_t *s0 = src - (NT / 2 - 1);
#pragma unroll 2
		for (int k = g.tid * 4; k < w * h; k += g.n * 4) {
			int r, c;
			split_rc(k, w, lw, &r, &c);
			int sv[NS];
#pragma unroll
			for (int q = 0; q < NS / 4; q++) {
				const S4 v = ld4(s0 + r * ss + c + 4 * q);
#pragma unroll
				for (int j = 0; j < 4; j++) sv[4 * q + j] = v.v[j];
			}
			S4 o;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				int sum = 0;
#pragma unroll
				for (int t = 0; t < NT; t++) sum += sv[j + t] * c8[t];
				int16_t v = sat16((sum + offset) >> shift);
				if (last) v = (int16_t)hclip((int)v, 0, 255);
				o.v[j] = v;
			}
			st4(dst + r * ds + c, o);
		}
	} else {
		const int16_t *s0 = src - (NT / 2 - 1) * ss;
#pragma unroll 2
		for (int k = g.tid * 4; k < w * h; k += g.n * 4) {
			int r, c;
			split_rc(k, w, lw, &r, &c);
			int sum[4] = {0, 0, 0, 0};
#pragma unroll
			for (int t = 0; t < NT; t++) {
				const S4 v = ld4(s0 + (r + t) * ss + c);
#pragma unroll
				for (int j = 0; j < 4; j++) sum[j] += v.v[j] * c8[t];
			}
			S4 o;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				int16_t v = sat16((sum[j] + offset) >> shift);
				if (last) v = (int16_t)hclip((int)v, 0, 255);
				o.v[j] = v;
			}
			st4(dst + r * ds + c, o);
		}
	}
	g.sync();
	PRIM_END(PP_INTERP);
}

// ---- the constant tables a TU needs, next to the worker ------------------------------------------------------------------
// The transform bases, the coefficient scans and the quantiser lists are read by every TU of every candidate; DevTables keeps them in HBM (1.2 MB with every
// list at every size and QP remainder), where each use costs an L2 round trip in the middle of a dependent chain.  A row worker keeps what it can use in LDS:
//   * the four DCT bases + DST, plain and transposed, densely packed;
//   * the scans as 16-bit positions: horizontal / vertical / diagonal for 4x4 and 8x8, diagonal only for 16x16 and 32x32 (find_scan_mode gives nothing else there);
//   * the quantiser / dequantiser lists of ONE QP (the frame's, luma and chroma remainder): the default scaling lists are 8x8 matrices replicated over larger
//     blocks (tables.cpp, HOMER_enc_init), so two 64-entry lists (intra, inter) per remainder class, the flat value of 4x4 blocks / DC of the replicated sizes.
// A TU whose QP is not the cached one falls back to DevTables.
struct alignas(16) FastTables {
	int16_t dct[16 + 64 + 256 + 1024], dct_t[16 + 64 + 256 + 1024];   // [log2N - 2] at offsets 0, 16, 80, 336
	int16_t dst4[16], dst4_t[16];
	uint16_t scan4[3][16], scan8[3][64], scan16[256], scan32[1024];    // [scan_mode - 1]
	uint16_t q8[2][2][64], iq8[2][2][64];                               // [luma / chroma remainder][intra / inter list][8x8 cell]
	uint16_t q_flat[2], iq_flat[2];
	int32_t rem[2], valid;
};
HENC_INLINE int ft_dct_offset(int log2n) { return log2n == 2 ? 0 : (log2n == 3 ? 16 : (log2n == 4 ? 80 : 336)); }
HENC_INLINE const uint16_t *ft_scan(const FastTables &F, int scan_mode, int log2n)
{
	return log2n == 2 ? F.scan4[scan_mode - 1] : (log2n == 3 ? F.scan8[scan_mode - 1] : (log2n == 4 ? F.scan16 : F.scan32));
}
// the list (0 intra, 1 inter) behind DevTables::quant[log2n - 2][list]: tables.cpp, hmr_tables.c default scaling lists (32x32 has its own rule)
HENC_INLINE int ft_list_kind(int log2n, int list) { return log2n == 5 ? (list != 0) : (list >= 3); }
// element i of an n x n list from its 8x8 cells (n >= 8; DC of the replicated sizes and every 4x4 entry: the flat value)
HENC_INLINE uint32_t ft_list_value(const uint16_t *cells, uint32_t flat, int i, int log2n)
{
	if (log2n == 2 || (i == 0 && log2n > 3)) return flat;
	const int s = log2n - 3, y = i >> log2n, x = i & ((1 << log2n) - 1);
	return cells[((y >> s) << 3) | (x >> s)];
}
// the same for the four elements i0 .. i0 + 3 of a row (i0 a multiple of 4): one, two or four cells
template <class CellT>      // uint16_t: the cells cached in FastTables; int32_t: the 8 x 8 lists of DevTables themselves
HENC_INLINE void ft_list_value4(const CellT *cells, uint32_t flat, int i0, int log2n, uint32_t *v)
{
	if (log2n == 2) { v[0] = v[1] = v[2] = v[3] = flat; return; }
	const int s = log2n - 3, y = i0 >> log2n, x = i0 & ((1 << log2n) - 1);
	const CellT *c = cells + (((y >> s) << 3) | (x >> s));
	if (s == 0) { v[0] = (uint32_t)c[0]; v[1] = (uint32_t)c[1]; v[2] = (uint32_t)c[2]; v[3] = (uint32_t)c[3]; }
	else if (s == 1) { v[0] = v[1] = (uint32_t)c[0]; v[2] = v[3] = (uint32_t)c[1]; }
	else v[0] = v[1] = v[2] = v[3] = (uint32_t)c[0];
	if (i0 == 0 && log2n > 3) v[0] = flat;
}
template <class G>
HENC_HD void fast_tables_fill(const G g, FastTables &F, const DevTables *T, int rem_y, int rem_c)
{
	for (int l = 2; l <= 5; l++) {
		const int n = 1 << l, o = ft_dct_offset(l);
		for (int i = g.tid; i < n * n; i += g.n) { F.dct[o + i] = T->dct[l - 2][i]; F.dct_t[o + i] = T->dct_t[l - 2][i]; }
	}
	for (int i = g.tid; i < 16; i += g.n) { F.dst4[i] = T->dst4[i]; F.dst4_t[i] = T->dst4_t[i]; }
	for (int m = 1; m <= 3; m++) {
		for (int i = g.tid; i < 16; i += g.n) F.scan4[m - 1][i] = (uint16_t)T->scan[m][2][i];
		for (int i = g.tid; i < 64; i += g.n) F.scan8[m - 1][i] = (uint16_t)T->scan[m][3][i];
	}
	for (int i = g.tid; i < 256; i += g.n) F.scan16[i] = (uint16_t)T->scan[SCAN_DIAG][4][i];
	for (int i = g.tid; i < 1024; i += g.n) F.scan32[i] = (uint16_t)T->scan[SCAN_DIAG][5][i];
	for (int c = 0; c < 2; c++) {
		const int rem = c ? rem_c : rem_y;
		for (int k = 0; k < 2; k++)
			for (int i = g.tid; i < 64; i += g.n) {
				F.q8[c][k][i] = (uint16_t)T->quant[1][k ? 3 : 0][rem][i];       // the 8x8 lists themselves
				F.iq8[c][k][i] = (uint16_t)T->dequant[1][k ? 3 : 0][rem][i];
			}
		if (g.tid == 0) {
			F.q_flat[c] = (uint16_t)T->quant[0][0][rem][0];
			F.iq_flat[c] = (uint16_t)T->dequant[0][0][rem][0];
			F.rem[c] = rem;
		}
	}
	if (g.tid == 0) F.valid = 1;
	g.sync();
}

// ---- transforms (hmr_sse42_functions_transform.c:1670,1700; spec hmr_transform.c:133-549) ---------------------------
// Every stage is out[k][j] = sat16((sum_i B[k][i] * in[j][i] + rnd) >> shift) with both operand rows contiguous: a lane keeps "its" input
// row in registers as packed pairs and walks the basis rows with 16-byte loads and two-way dot products (v_dot2_i32_i16 on the device).
// The sums are exact in 32 bits (<= 32 products of 16-bit values), so the order of accumulation does not matter.
HENC_INLINE int32_t dot2_acc(int32_t a, int32_t b, int32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	typedef short short2_t __attribute__((ext_vector_type(2)));
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c, false);
#else
	return c + (int32_t)(int16_t)(a & 0xffff) * (int32_t)(int16_t)(b & 0xffff) + (int32_t)(int16_t)(a >> 16) * (int32_t)(int16_t)(b >> 16);
#endif
}
HENC_INLINE int32_t pack_pair(int lo, int hi) { return (int32_t)(((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16)); }
// N 16-bit values from an address that is a multiple of 2N bytes (at most 16) -> N / 2 packed pairs
template <int N>
HENC_INLINE void load_pairs(const int16_t *p, int32_t (&r)[N / 2])
{
	__builtin_memcpy(r, __builtin_assume_aligned(p, N >= 8 ? 16 : 8), N * 2);
}

// where a stage's input rows come from: a 16-bit block, or the residual source - prediction formed on the way (what the reference's predict kernel writes into
// its residual window - 16-bit wrap included - before the forward transform reads it; the worker keeps no such window)
template <int N>
struct RowPlain {
	const int16_t *in;
	int is;
	HENC_INLINE void load(int j, int32_t (&row)[N / 2]) const { load_pairs<N>(in + j * is, row); }
};
template <int N, class S, class P>
struct RowDiff {
	const S *o;
	int os;
	const P *p;
	int ps;
	HENC_INLINE void load(int j, int32_t (&row)[N / 2]) const
	{
		S ov[N];
		P pv[N];
		__builtin_memcpy(ov, __builtin_assume_aligned(o + j * os, (N * sizeof(S)) >= 16 ? 16 : N * sizeof(S)), N * sizeof(S));
		__builtin_memcpy(pv, __builtin_assume_aligned(p + j * ps, (N * sizeof(P)) >= 16 ? 16 : N * sizeof(P)), N * sizeof(P));
#pragma unroll
		for (int h = 0; h < N / 2; h++) {
			const int16_t lo = (int16_t)((int)ov[2 * h] - (int)pv[2 * h]), hi = (int16_t)((int)ov[2 * h + 1] - (int)pv[2 * h + 1]);
			row[h] = pack_pair(lo, hi);
		}
	}
};

// one stage with the input row held by the lane: row j of the source `in`
template <int N, class G, class Rows>
HENC_HD void tr_stage_rows_from(const G g, const int16_t *B, const Rows &in, int16_t *out, int os_k, int os_j, int shift)
{
	constexpr int H = N / 2;
	const int rnd = shift > 0 ? 1 << (shift - 1) : 0;
	if (G::n % N == 0) {
		const int j = g.tid % N;
		int32_t row[H], m[H];
		in.load(j, row);
#pragma unroll 4
		for (int k = g.tid / N; k < N; k += (G::n / N ? G::n / N : 1)) {
			load_pairs<N>(B + k * N, m);
			int32_t s = 0;
#pragma unroll
			for (int h = 0; h < H; h++) s = dot2_acc(m[h], row[h], s);
			out[k * os_k + j * os_j] = sat16((s + rnd) >> shift);
		}
	} else {
		for (int o = g.tid; o < N * N; o += G::n) {
			const int j = o % N, k = o / N;
			int32_t row[H], m[H];
			in.load(j, row);
			load_pairs<N>(B + k * N, m);
			int32_t s = 0;
			for (int h = 0; h < H; h++) s = dot2_acc(m[h], row[h], s);
			out[k * os_k + j * os_j] = sat16((s + rnd) >> shift);
		}
	}
}
// ... in[j][0..N) contiguous at in + j * is
template <int N, class G>
HENC_HD void tr_stage_rows(const G g, const int16_t *B, const int16_t *in, int is, int16_t *out, int os_k, int os_j, int shift)
{
	tr_stage_rows_from<N>(g, B, RowPlain<N>{in, is}, out, os_k, os_j, shift);
}
// the same with the input COLUMN j of a linear N x N array (first inverse stage: the levels come row-major)
template <int N, class G>
HENC_HD void tr_stage_cols(const G g, const int16_t *B, const int16_t *in, int16_t *out, int shift)
{
	constexpr int H = N / 2;
	const int rnd = 1 << (shift - 1);
	const int step = G::n % N == 0 ? G::n : 1;                   // lanes keep their column when the group is a multiple of N wide
	for (int o = g.tid; o < N * N; o += step) {
		const int j = o % N;
		int32_t col[H], m[H];
#pragma unroll
		for (int h = 0; h < H; h++) col[h] = pack_pair(in[(2 * h) * N + j], in[(2 * h + 1) * N + j]);
#pragma unroll 4
		for (int k = o / N; k < N; k += (G::n % N == 0 ? G::n / N : N)) {
			load_pairs<N>(B + k * N, m);
			int32_t s = 0;
#pragma unroll
			for (int h = 0; h < H; h++) s = dot2_acc(m[h], col[h], s);
			out[k * N + j] = sat16((s + rnd) >> shift);
		}
		if (G::n % N == 0) break;
	}
}

template <int N, class G, class Rows>
HENC_HD void tr_forward_n(const G g, const int16_t *M, const Rows &block, int16_t *coeff, int16_t *tmp)
{
	constexpr int L = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	tr_stage_rows_from<N>(g, M, block, tmp, N, 1, L - 1);    // tmp[k][j] = sum_i M[k][i] * block[j][i]
	g.sync();
	tr_stage_rows<N>(g, M, tmp, N, coeff, N, 1, L + 6);      // coeff[k][j] = sum_i M[k][i] * tmp[j][i]
	g.sync();
}
template <int N, class G>
HENC_HD void tr_inverse_n(const G g, const int16_t *Mt, int16_t *block, int bs, const int16_t *coeff, int16_t *tmp)
{
	tr_stage_cols<N>(g, Mt, coeff, tmp, 7);                   // tmp[k][j] = sum_i M[i][k] * coeff[i][j]  (the reference's tmp, transposed)
	g.sync();
	tr_stage_rows<N>(g, Mt, tmp, N, block, 1, bs, 12);        // block[j][k] = sum_i M[i][k] * tmp[k'= i][j]
	g.sync();
}

#if defined(__HIPCC__) && !defined(HENC_NO_MFMA_TRANSFORM)
// ---- the transforms on the matrix cores (device, one whole wavefront) -------------------------------------------------------------------------------
// A stage is a matrix product, and the product of two matrices of small integers is exact in the matrix cores' binary16 x binary16 -> binary32 arithmetic
// as long as every partial sum stays below 2^24: the bases are integers of at most 90, a residual is within +-255, and a 16-bit stage input is split
// into its high byte (-128 .. 127) and low byte (0 .. 255), each with an accumulator of its own (at most 32 x 255 x 90 < 2^20), put together as integers.
// The rounding, shift and 16-bit saturation between the stages are the reference's, on integers.  v_mfma_f32_16x16x16_f16 holds A[lane % 16][4 (lane / 16) + e],
// B[4 (lane / 16) + e][lane % 16] and D[4 (lane / 16) + r][lane % 16]: a stage's result IS the next stage's operand (as B, or - read as the transposed
// matrix - as A), so the intermediate never leaves the registers, and the last result has four consecutive outputs of one row per lane (one 8-byte store).
// Sizes 4 and 8 (and the DST) run in a 16 x 16 tile padded with zeros; 32 x 32 runs as 16 x 16 quarter tiles (as a v_mfma_f32_32x32x8_f16 chain the accumulators
// held 64 registers, paid for with spills around the transform).  The bases come as ready fragments (DevTables::frag16 / frag32t / fragp).  ALL 64 lanes must be active: the caller is the whole wavefront in uniform control flow.
typedef _Float16 mf_h4 __attribute__((ext_vector_type(4)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));
typedef float mf_f16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mf_h4 mf_frag(const uint16_t *frag, int lane) { mf_h4 r; __builtin_memcpy(&r, __builtin_assume_aligned(frag + lane * 4, 8), 8); return r; }
__device__ __forceinline__ int mf_stage(float d_hi, float d_lo, int rnd, int shift) { return (int)sat16(((((int)d_hi) << 8) + (int)d_lo + rnd) >> shift); }
__device__ __forceinline__ void mf_split(const int (&t)[4], mf_h4 &hi, mf_h4 &lo)
{
#pragma unroll
	for (int r = 0; r < 4; r++) { hi[r] = (_Float16)(short)(t[r] >> 8); lo[r] = (_Float16)(short)(t[r] & 255); }
}
template <int N, class S, class P>
__device__ __forceinline__ void tr_forward_mfma(int lane, const DevTables *T, int basis, const S *orig, int os, const P *pred, int ps, int16_t *coeff)
{
	constexpr int L = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	constexpr int sh1 = L - 1, sh2 = L + 6, rnd1 = sh1 > 0 ? 1 << (sh1 - 1) : 0, rnd2 = 1 << (sh2 - 1);
	if constexpr (N <= 16) {
		const int row = lane & 15, k0 = (lane >> 4) * 4;
		const bool in = row < N && k0 < N;
		const mf_h4 m = mf_frag(T->frag16[0][basis], lane);                 // M[row][k0 + e]
		mf_h4 x = {0, 0, 0, 0};
		if (in) {
			S ov[4]; P pv[4];
			__builtin_memcpy(ov, __builtin_assume_aligned(orig + row * os + k0, 4 * sizeof(S)), 4 * sizeof(S));
			__builtin_memcpy(pv, __builtin_assume_aligned(pred + row * ps + k0, 4 * sizeof(P)), 4 * sizeof(P));
#pragma unroll
			for (int e = 0; e < 4; e++) x[e] = (_Float16)(short)((int)ov[e] - (int)pv[e]);
		}
		const mf_f4 z = {0, 0, 0, 0};
		const mf_f4 d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, m, z, 0, 0, 0);      // T[j = k0 + r][k1 = row]
		int t[4];
#pragma unroll
		for (int r = 0; r < 4; r++) t[r] = (int)sat16(((int)d1[r] + rnd1) >> sh1);
		mf_h4 hi, lo;
		mf_split(t, hi, lo);
		const mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, m, z, 0, 0, 0);     // (as A: the transposed intermediate) -> Y[k2 = row][k1 = k0 + r]
		const mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, m, z, 0, 0, 0);
		if (in) {
			S4 o;
#pragma unroll
			for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(dh[r], dl[r], rnd2, sh2);
			st4(coeff + row * N + k0, o);
		}
	} else {
		// 32 x 32 as 16 x 16 quarter tiles (see tr_inverse_mfma): eight accumulator registers in flight
		const int r16 = lane & 15, g4 = (lane >> 4) * 4;
		mf_h4 m[2][2];
#pragma unroll
		for (int R = 0; R < 2; R++)
#pragma unroll
			for (int K = 0; K < 2; K++) m[R][K] = mf_frag(T->frag32t[0][R][K], lane);        // M[16 R + r16][16 K + g4 + e]
		const mf_f4 z = {0, 0, 0, 0};
		mf_h4 th[2][2], tl[2][2];                                                               // the intermediate: tile (J, K1) = rows j of block J, columns k1 of block K1
#pragma unroll
		for (int J = 0; J < 2; J++) {
			mf_h4 x[2];
#pragma unroll
			for (int C = 0; C < 2; C++) {
				S ov[4]; P pv[4];
				__builtin_memcpy(ov, __builtin_assume_aligned(orig + (16 * J + r16) * os + 16 * C + g4, 4 * sizeof(S)), 4 * sizeof(S));
				__builtin_memcpy(pv, __builtin_assume_aligned(pred + (16 * J + r16) * ps + 16 * C + g4, 4 * sizeof(P)), 4 * sizeof(P));
#pragma unroll
				for (int e = 0; e < 4; e++) x[C][e] = (_Float16)(short)((int)ov[e] - (int)pv[e]);
			}
#pragma unroll
			for (int K1 = 0; K1 < 2; K1++) {
				mf_f4 d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x[0], m[K1][0], z, 0, 0, 0);
				d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x[1], m[K1][1], d1, 0, 0, 0);        // T[j = 16 J + g4 + r][k1 = 16 K1 + r16]
				int t[4];
#pragma unroll
				for (int r = 0; r < 4; r++) t[r] = (int)sat16(((int)d1[r] + rnd1) >> sh1);
				mf_split(t, th[J][K1], tl[J][K1]);
			}
		}
#pragma unroll
		for (int K2 = 0; K2 < 2; K2++)
#pragma unroll
			for (int K1 = 0; K1 < 2; K1++) {
				mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(th[0][K1], m[K2][0], z, 0, 0, 0);
				mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(tl[0][K1], m[K2][0], z, 0, 0, 0);
				dh = __builtin_amdgcn_mfma_f32_16x16x16f16(th[1][K1], m[K2][1], dh, 0, 0, 0);
				dl = __builtin_amdgcn_mfma_f32_16x16x16f16(tl[1][K1], m[K2][1], dl, 0, 0, 0);
				S4 o;                                                                               // Y[k2 = 16 K2 + r16][k1 = 16 K1 + g4 + r]
#pragma unroll
				for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(dh[r], dl[r], rnd2, sh2);
				st4(coeff + (16 * K2 + r16) * 32 + 16 * K1 + g4, o);
			}
	}
}
// inverse: tmp[a][b] = sat16((sum_i Mt[a][i] coeff[i][b] + 64) >> 7), block[a][k] = sat16((sum_b Mt[k][b] tmp[a][b] + 2048) >> 12)   (tr_inverse_n)
template <int N>
__device__ __forceinline__ void tr_inverse_mfma(int lane, const DevTables *T, int basis, int16_t *block, int bs, const int16_t *coeff)
{
	if constexpr (N <= 16) {
		const int row = lane & 15, k0 = (lane >> 4) * 4;
		const bool in = row < N && k0 < N;
		const mf_h4 mt = mf_frag(T->frag16[1][basis], lane);                // Mt[row][k0 + e]
		int c[4] = {0, 0, 0, 0};
		if (in) {
#pragma unroll
			for (int e = 0; e < 4; e++) c[e] = coeff[(k0 + e) * N + row];         // coeff[i = k0 + e][b = row]: the operand's K runs down a column
		}
		mf_h4 hi, lo;
		mf_split(c, hi, lo);
		const mf_f4 z = {0, 0, 0, 0};
		mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, mt, z, 0, 0, 0);          // tmp[a = row][b = k0 + r]  (D rows b, columns a)
		mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, mt, z, 0, 0, 0);
		int t[4];
#pragma unroll
		for (int r = 0; r < 4; r++) t[r] = mf_stage(dh[r], dl[r], 64, 7);
		mf_split(t, hi, lo);
		dh = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, hi, z, 0, 0, 0);                // (as B) -> block[a = row][k = k0 + r]
		dl = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, lo, z, 0, 0, 0);
		if (in) {
			S4 o;
#pragma unroll
			for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(dh[r], dl[r], 2048, 12);
			st4(block + row * bs + k0, o);
		}
	} else {
		// 32 x 32 as 16 x 16 quarter tiles (DevTables::frag32t): the two stages as sixteen small products each, with eight accumulator registers in flight (as one
		// 32 x 32 x 8 chain the four accumulators took 64 registers, and what they displaced was spilled in the loops around the transform: 17 % more store instructions)
		const int r16 = lane & 15, g4 = (lane >> 4) * 4;
		mf_h4 mt[2][2];
#pragma unroll
		for (int R = 0; R < 2; R++)
#pragma unroll
			for (int K = 0; K < 2; K++) mt[R][K] = mf_frag(T->frag32t[1][R][K], lane);      // Mt[16 R + r16][16 K + g4 + e]
		const mf_f4 z = {0, 0, 0, 0};
		mf_h4 th[2][2], tl[2][2];                                                               // tmp, transposed: tile (B, A) = rows b of block B, columns a of block A
#pragma unroll
		for (int B = 0; B < 2; B++) {
			mf_h4 ch[2], cl[2];
#pragma unroll
			for (int I = 0; I < 2; I++) {
				int c[4];
#pragma unroll
				for (int e = 0; e < 4; e++) c[e] = coeff[(16 * I + g4 + e) * 32 + 16 * B + r16];       // coeff[i][b]
				mf_split(c, ch[I], cl[I]);
			}
#pragma unroll
			for (int A = 0; A < 2; A++) {
				mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(ch[0], mt[A][0], z, 0, 0, 0);
				mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(cl[0], mt[A][0], z, 0, 0, 0);
				dh = __builtin_amdgcn_mfma_f32_16x16x16f16(ch[1], mt[A][1], dh, 0, 0, 0);
				dl = __builtin_amdgcn_mfma_f32_16x16x16f16(cl[1], mt[A][1], dl, 0, 0, 0);
				int t[4];
#pragma unroll
				for (int r = 0; r < 4; r++) t[r] = mf_stage(dh[r], dl[r], 64, 7);                 // tmp[a = 16 A + r16][b = 16 B + g4 + r]
				mf_split(t, th[B][A], tl[B][A]);
			}
		}
#pragma unroll
		for (int K = 0; K < 2; K++)
#pragma unroll
			for (int A = 0; A < 2; A++) {
				mf_f4 eh = __builtin_amdgcn_mfma_f32_16x16x16f16(mt[K][0], th[0][A], z, 0, 0, 0);
				mf_f4 el = __builtin_amdgcn_mfma_f32_16x16x16f16(mt[K][0], tl[0][A], z, 0, 0, 0);
				eh = __builtin_amdgcn_mfma_f32_16x16x16f16(mt[K][1], th[1][A], eh, 0, 0, 0);
				el = __builtin_amdgcn_mfma_f32_16x16x16f16(mt[K][1], tl[1][A], el, 0, 0, 0);
				S4 o;                                                                               // block[a = 16 A + r16][k = 16 K + g4 + r]
#pragma unroll
				for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(eh[r], el[r], 2048, 12);
				st4(block + (16 * A + r16) * bs + 16 * K + g4, o);
			}
	}
}
// Two blocks of size 4 or 8 at once, one per half of the wavefront (PairGrp: the helper's two chroma planes of a small TU): block h lies in rows / columns
// 8 h .. 8 h + N - 1 of the 16 x 16 tile (DevTables::fragp is the block-diagonal basis), its operands come from the lanes of half h whose tile row is the block's,
// and its results arrive in those lanes.  Both halves run this together (the caller's control flow is uniform here); `live` says whether the half has work.
template <int N, class S, class P>
__device__ __forceinline__ void tr_forward_mfma_pair(int lane, const DevTables *T, const S *orig, int os, const P *pred, int ps, int16_t *coeff)
{
	constexpr int L = N == 4 ? 2 : 3;
	constexpr int sh1 = L - 1, sh2 = L + 6, rnd1 = 1 << (sh1 - 1), rnd2 = 1 << (sh2 - 1);
	const int row = lane & 7, k0 = ((lane >> 4) & 1) * 4;
	const bool in = ((lane & 15) >> 3) == (lane >> 5) && row < N && k0 < N;
	const mf_h4 m = mf_frag(T->fragp[0][N == 8], lane);
	mf_h4 x = {0, 0, 0, 0};
	if (in) {
		S ov[4]; P pv[4];
		__builtin_memcpy(ov, __builtin_assume_aligned(orig + row * os + k0, 4 * sizeof(S)), 4 * sizeof(S));
		__builtin_memcpy(pv, __builtin_assume_aligned(pred + row * ps + k0, 4 * sizeof(P)), 4 * sizeof(P));
#pragma unroll
		for (int e = 0; e < 4; e++) x[e] = (_Float16)(short)((int)ov[e] - (int)pv[e]);
	}
	const mf_f4 z = {0, 0, 0, 0};
	const mf_f4 d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, m, z, 0, 0, 0);
	int t[4];
#pragma unroll
	for (int r = 0; r < 4; r++) t[r] = (int)sat16(((int)d1[r] + rnd1) >> sh1);
	mf_h4 hi, lo;
	mf_split(t, hi, lo);
	const mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, m, z, 0, 0, 0);
	const mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, m, z, 0, 0, 0);
	if (in) {
		S4 o;
#pragma unroll
		for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(dh[r], dl[r], rnd2, sh2);
		st4(coeff + row * N + k0, o);
	}
}
template <int N>
__device__ __forceinline__ void tr_inverse_mfma_pair(int lane, bool live, const DevTables *T, int16_t *block, int bs, const int16_t *coeff)
{
	const int row = lane & 7, k0 = ((lane >> 4) & 1) * 4;
	const bool in = live && ((lane & 15) >> 3) == (lane >> 5) && row < N && k0 < N;
	const mf_h4 mt = mf_frag(T->fragp[1][N == 8], lane);
	int c[4] = {0, 0, 0, 0};
	if (in) {
#pragma unroll
		for (int e = 0; e < 4; e++) c[e] = coeff[(k0 + e) * N + row];
	}
	mf_h4 hi, lo;
	mf_split(c, hi, lo);
	const mf_f4 z = {0, 0, 0, 0};
	mf_f4 dh = __builtin_amdgcn_mfma_f32_16x16x16f16(hi, mt, z, 0, 0, 0);
	mf_f4 dl = __builtin_amdgcn_mfma_f32_16x16x16f16(lo, mt, z, 0, 0, 0);
	int t[4];
#pragma unroll
	for (int r = 0; r < 4; r++) t[r] = mf_stage(dh[r], dl[r], 64, 7);
	mf_split(t, hi, lo);
	dh = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, hi, z, 0, 0, 0);
	dl = __builtin_amdgcn_mfma_f32_16x16x16f16(mt, lo, z, 0, 0, 0);
	if (in) {
		S4 o;
#pragma unroll
		for (int r = 0; r < 4; r++) o.v[r] = (int16_t)mf_stage(dh[r], dl[r], 2048, 12);
		st4(block + row * bs + k0, o);
	}
}
#define HENC_MFMA_TRANSFORM 1
#endif

// forward transform of the residual source - prediction (hmr_motion_intra.c:1036-1040 / hmr_motion_inter.c:57-60: predict, then transform of the residual window)
template <class G, class S, class P>
HENC_PRIM void tr_forward(const G g, const FastTables *F, const DevTables *T, const S *orig, int os, const P *pred, int ps, int16_t *coeff, int16_t *tmp, int n, int is_dst)
{
	PRIM_T0();
	if constexpr (G::n == 64) {      // (one block for the whole wavefront: the arguments are uniform; a pair of groups has a block per half)
		orig = uni_ptr(orig); pred = uni_ptr(pred); coeff = uni_ptr(coeff); tmp = uni_ptr(tmp); os = uni(os); ps = uni(ps);
	}
	T = uni_ptr(T); n = uni(n); is_dst = uni(is_dst);
	HENC_OP_IN_LDS(orig); HENC_OP_IN_LDS(pred); HENC_OP_IN_LDS(coeff); HENC_OP_IN_LDS(tmp);
#if defined(HENC_MFMA_TRANSFORM)
#if !defined(HENC_MFMA_NO_FWD)
	if constexpr (G::n == 64 && sizeof(S) == 1 && sizeof(P) == 1) {      // (bytes: the residual is within +-255)
		switch (n) {
		case 4: tr_forward_mfma<4>(g.tid, T, is_dst ? 3 : 0, orig, os, pred, ps, coeff); break;
		case 8: tr_forward_mfma<8>(g.tid, T, 1, orig, os, pred, ps, coeff); break;
		case 16: tr_forward_mfma<16>(g.tid, T, 2, orig, os, pred, ps, coeff); break;
		default: tr_forward_mfma<32>(g.tid, T, 0, orig, os, pred, ps, coeff); break;
		}
		g.sync();
		PRIM_END(PP_TRF);
		return;
	}
#endif
#if !defined(HENC_MFMA_NO_PAIR)
	if constexpr (G::n == 32 && sizeof(S) == 1 && sizeof(P) == 1) {      // two blocks of 4 x 4 or 8 x 8, a half of the wavefront each, in one tile (the caller's flow is uniform)
		if (n <= 8 && !is_dst) {
			const int lane = g.half * 32 + g.tid;
			if (n == 4) tr_forward_mfma_pair<4>(lane, T, orig, os, pred, ps, coeff);
			else tr_forward_mfma_pair<8>(lane, T, orig, os, pred, ps, coeff);
			g.sync();
			PRIM_END(PP_TRF);
			return;
		}
	}
#endif
#endif
	if (!F) {
		switch (n) {
		case 4: tr_forward_n<4>(g, is_dst ? T->dst4 : T->dct[0], RowDiff<4, S, P>{orig, os, pred, ps}, coeff, tmp); break;
		case 8: tr_forward_n<8>(g, T->dct[1], RowDiff<8, S, P>{orig, os, pred, ps}, coeff, tmp); break;
		case 16: tr_forward_n<16>(g, T->dct[2], RowDiff<16, S, P>{orig, os, pred, ps}, coeff, tmp); break;
		default: tr_forward_n<32>(g, T->dct[3], RowDiff<32, S, P>{orig, os, pred, ps}, coeff, tmp); break;
		}
		PRIM_END(PP_TRF);
		return;
	}
	switch (n) {
	case 4: tr_forward_n<4>(g, is_dst ? F->dst4 : F->dct, RowDiff<4, S, P>{orig, os, pred, ps}, coeff, tmp); break;
	case 8: tr_forward_n<8>(g, F->dct + 16, RowDiff<8, S, P>{orig, os, pred, ps}, coeff, tmp); break;
	case 16: tr_forward_n<16>(g, F->dct + 80, RowDiff<16, S, P>{orig, os, pred, ps}, coeff, tmp); break;
	default: tr_forward_n<32>(g, F->dct + 336, RowDiff<32, S, P>{orig, os, pred, ps}, coeff, tmp); break;
	}
	PRIM_END(PP_TRF);
}

template <class G>
HENC_PRIM void tr_inverse(const G g, const FastTables *F, const DevTables *T, int16_t *block, int bs, const int16_t *coeff, int16_t *tmp, int n, int is_dst)
{
	PRIM_T0();
	if constexpr (G::n == 64) { block = uni_ptr(block); coeff = uni_ptr(coeff); tmp = uni_ptr(tmp); bs = uni(bs); }
	T = uni_ptr(T); n = uni(n); is_dst = uni(is_dst);
	HENC_OP_IN_LDS(block); HENC_OP_IN_LDS(coeff); HENC_OP_IN_LDS(tmp);
#if defined(HENC_MFMA_TRANSFORM) && !defined(HENC_MFMA_NO_INV)
#if !defined(HENC_MFMA_INV_MASK)
#define HENC_MFMA_INV_MASK 15
#endif
	if constexpr (G::n == 64) if ((HENC_MFMA_INV_MASK >> (n == 4 ? 0 : n == 8 ? 1 : n == 16 ? 2 : 3)) & 1) {
		switch (n) {
		case 4: tr_inverse_mfma<4>(g.tid, T, is_dst ? 3 : 0, block, bs, coeff); break;
		case 8: tr_inverse_mfma<8>(g.tid, T, 1, block, bs, coeff); break;
		case 16: tr_inverse_mfma<16>(g.tid, T, 2, block, bs, coeff); break;
		default: tr_inverse_mfma<32>(g.tid, T, 0, block, bs, coeff); break;
		}
		g.sync();
		PRIM_END(PP_TRI);
		return;
	}
#endif
	if (!F) {
		switch (n) {
		case 4: tr_inverse_n<4>(g, is_dst ? T->dst4_t : T->dct_t[0], block, bs, coeff, tmp); break;
		case 8: tr_inverse_n<8>(g, T->dct_t[1], block, bs, coeff, tmp); break;
		case 16: tr_inverse_n<16>(g, T->dct_t[2], block, bs, coeff, tmp); break;
		default: tr_inverse_n<32>(g, T->dct_t[3], block, bs, coeff, tmp); break;
		}
		PRIM_END(PP_TRI);
		return;
	}
	switch (n) {
	case 4: tr_inverse_n<4>(g, is_dst ? F->dst4_t : F->dct_t, block, bs, coeff, tmp); break;
	case 8: tr_inverse_n<8>(g, F->dct_t + 16, block, bs, coeff, tmp); break;
	case 16: tr_inverse_n<16>(g, F->dct_t + 80, block, bs, coeff, tmp); break;
	default: tr_inverse_n<32>(g, F->dct_t + 336, block, bs, coeff, tmp); break;
	}
	PRIM_END(PP_TRI);
}

#if defined(HENC_MFMA_TRANSFORM)
// the inverse transform of the two halves' blocks together (see tr_forward_mfma_pair): called by BOTH halves from uniform control flow, `live` = this half has levels
template <class G>
__device__ __forceinline__ void tr_inverse_pair(const G g, bool live, const DevTables *T, int16_t *block, int bs, const int16_t *coeff, int n)
{
	PRIM_T0();
	HENC_OP_IN_LDS(block); HENC_OP_IN_LDS(coeff);
	const int lane = g.half * 32 + g.tid;
	if (n == 4) tr_inverse_mfma_pair<4>(lane, live, T, block, bs, coeff);
	else tr_inverse_mfma_pair<8>(lane, live, T, block, bs, coeff);
	g.sync();
	PRIM_END(PP_TRI);
}
#endif

// ---- quantisation (hmr_sse42_functions_quant.c:34-131 + sign_bit_hidding hmr_quant.c:61-169; inverse :135-246) -------
// sign hiding of one 16-coefficient group (sign_bit_hidding, hmr_quant.c:61-169).  The group's positions, levels, source coefficients and
// rounding remainders are gathered first (independent loads), the reference's walk then runs on them with fixed trip counts:
//   first / last non-zero level in scan order, the parity test, and the search for the cheapest +-1 change from the start position down
//   to 0 (strict "<" on a descending walk = the LARGEST position wins ties, hence "<=" on the ascending loop here).
struct SbhGroup {
	uint32_t pos[16];
	int16_t lv[16], sv[16], du[16];
};
template <class ScanT>
HENC_INLINE bool sbh_gather(SbhGroup &q, const int16_t *dst, const int16_t *src, const int16_t *du, const ScanT *scan, int cg)
{
	int any = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) q.pos[n] = scan[(cg << 4) + n];
#pragma unroll
	for (int n = 0; n < 16; n++) { q.lv[n] = dst[q.pos[n]]; q.sv[n] = src[q.pos[n]]; q.du[n] = du[q.pos[n]]; any |= q.lv[n]; }
	return any != 0;
}
HENC_INLINE void sbh_apply(const SbhGroup &q, int16_t *dst, bool is_last_cg)
{
	int first_nz = 16, last_nz = -1, abs_sum = 0, first_val = 0;
#pragma unroll
	for (int n = 0; n < 16; n++)
		if (q.lv[n]) last_nz = n;
#pragma unroll
	for (int n = 15; n >= 0; --n)
		if (q.lv[n]) { first_nz = n; first_val = q.lv[n]; }
	if (last_nz - first_nz < 4) return;
#pragma unroll
	for (int n = 0; n < 16; n++) abs_sum += q.lv[n];          // zero outside first_nz..last_nz
	const unsigned signbit = first_val > 0 ? 0 : 1;
	if (signbit == (unsigned)(abs_sum & 1)) return;
	const int start = is_last_cg ? last_nz : 15;
	int min_cost = 0x7fffffff, min_n = -1, final_change = 0;
	uint32_t min_pos = 0;
	int min_lv = 0, min_sv = 0;
#pragma unroll
	for (int n = 0; n < 16; n++) {
		if (n > start) continue;
		int cur_cost, cur_change = 0;
		if (q.lv[n] != 0) {
			if (q.du[n] > 0) { cur_cost = -q.du[n]; cur_change = 1; }
			else if (n == first_nz && habs((int)q.lv[n]) == 1) cur_cost = 0x7fffffff;
			else { cur_cost = q.du[n]; cur_change = -1; }
		} else if (n < first_nz) {
			const unsigned this_sign = q.sv[n] >= 0 ? 0 : 1;
			if (this_sign != signbit) cur_cost = 0x7fffffff;
			else { cur_cost = -q.du[n]; cur_change = 1; }
		} else { cur_cost = -q.du[n]; cur_change = 1; }
		// descending walk with "<": a position replaces the running minimum only when strictly cheaper, so among equals the largest n stays;
		// 0x7fffffff never replaces anything (the initial minimum is 0x7fffffff too)
		if (cur_cost != 0x7fffffff && cur_cost <= min_cost) { min_cost = cur_cost; final_change = cur_change; min_n = n; min_pos = q.pos[n]; min_lv = q.lv[n]; min_sv = q.sv[n]; }
	}
	if (min_n < 0) return;   // cannot happen after the parity test (the reference would index with -1); kept as a guard
	if (min_lv == 32767 || min_lv == -32768) final_change = -1;
	dst[min_pos] = (int16_t)(min_sv >= 0 ? min_lv + final_change : min_lv - final_change);
}

// the sign-hiding pass over the coefficient groups of a block (after the levels are visible to the group)
template <class G, class ScanT>
HENC_HD void sbh_pass(const G g, const int16_t *src, int16_t *dst, const int16_t *delta_u, const ScanT *scan, int total)
{
	const int ngroups = total >> 4;
	// the last group holding a level (in scan order) starts its walk at its last level
	if (G::n >= 32) {      // (a wavefront, or a half of one with a block of at most 16 x 16: a coefficient group per lane)
		// one group per lane: gather once, find the last non-empty group with a ballot, apply where there is anything to hide a sign in
		SbhGroup q;
		const int cg = g.tid;
		const bool nz = cg < ngroups && sbh_gather(q, dst, src, delta_u, scan, cg);
		const uint64_t mask = g.ballot(nz);
		const int last_cg = mask ? 63 - __builtin_clzll(mask) : -1;
		if (nz) sbh_apply(q, dst, cg == last_cg);
	} else {
		SbhGroup q;
		int last_cg = -1;
		for (int cg = g.tid; cg < ngroups; cg += g.n)
			if (sbh_gather(q, dst, src, delta_u, scan, cg)) last_cg = cg;
		for (int cg = g.tid; cg < ngroups; cg += g.n)
			if (sbh_gather(q, dst, src, delta_u, scan, cg)) sbh_apply(q, dst, cg == last_cg);
	}
	g.sync();
}

// returns ac_sum (the sum of the levels BEFORE sign hiding, as the reference reports it).  src / dst / delta_u are the worker's (fast memory); the lists and
// the scan come from F when it caches this QP remainder, else from T.
template <class G>
HENC_PRIM int quantize(const G g, const FastTables *F, const DevTables *T, const int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp,
		     int is_intra, int slice_is_intra, int sign_hiding, int n, int per, int rem)
{
	PRIM_T0();
	if constexpr (G::n == 64) {
		src = uni_ptr(src); dst = uni_ptr(dst); delta_u = uni_ptr(delta_u); per = uni(per); rem = uni(rem); comp = uni(comp);
	}
	T = uni_ptr(T); n = uni(n); depth = uni(depth); scan_mode = uni(scan_mode); is_intra = uni(is_intra); slice_is_intra = uni(slice_is_intra); sign_hiding = uni(sign_hiding);
	HENC_OP_IN_LDS(src); HENC_OP_IN_LDS(dst); HENC_OP_IN_LDS(delta_u);
	const int inv_depth = 6 - (depth + (comp != 0));
	const int list = (is_intra ? 0 : 3) + comp, rc = comp != 0;
	const bool fast = F && F->valid && F->rem[rc] == rem;
	const int qbits = 14 + per + (15 - 8 - inv_depth), qbits8 = qbits - 8;
	const int32_t add = (int32_t)((uint32_t)(slice_is_intra ? 171 : 85) << (qbits - 9));
	const int total = n * n;
	uint32_t sum = 0;
	// four coefficients per lane and step; a list value is its 8 x 8 cell (DC of the replicated sizes and every 4 x 4 entry: the flat value) - from the cells cached
	// in fast memory when F holds this QP remainder, else from the 8 x 8 lists of T themselves (the per-position lists of the larger sizes are never read)
	auto run = [&](const auto *cells, uint32_t flat) {
#pragma unroll 2
		for (int i = g.tid * 4; i < total; i += g.n * 4) {
			const S4 sv4 = ld4(src + i);
			uint32_t qv[4];
			ft_list_value4(cells, flat, i, inv_depth, qv);
			S4 lv, du;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int sv = sv4.v[k];
				const uint32_t a = (uint16_t)(sv < 0 ? -sv : sv);
				const int32_t aux = (int32_t)(a * qv[k]);
				const int32_t c = (int32_t)((uint32_t)aux + (uint32_t)add) >> qbits;
				const int32_t d = (int32_t)((uint32_t)aux - ((uint32_t)c << qbits)) >> qbits8;
				const int sgn = sv > 0 ? 1 : (sv < 0 ? -1 : 0);
				sum += (uint32_t)c;
				lv.v[k] = (int16_t)(sgn * sat16(c));
				du.v[k] = sat16(d);
			}
			st4(dst + i, lv);
			st4(delta_u + i, du);
		}
	};
	const int kind = ft_list_kind(inv_depth, list);
	if (fast) run(F->q8[rc][kind], (uint32_t)F->q_flat[rc]);
	else run(T->quant[1][kind ? 3 : 0][rem], (uint32_t)(uint16_t)T->quant[0][0][rem][0]);
	const int ac_sum = (int)g.sum(sum);
	g.sync();
	if (sign_hiding && ac_sum >= 2) {
		if (fast) sbh_pass(g, src, dst, delta_u, ft_scan(*F, scan_mode, inv_depth), total);
		else sbh_pass(g, src, dst, delta_u, T->scan[scan_mode][inv_depth], total);
	}
	{ const auto prim_ret_ = ac_sum; PRIM_END(PP_QUANT); return prim_ret_; }
}

// src == dst is allowed (element-wise)
template <class G>
HENC_PRIM void dequantize(const G g, const FastTables *F, const DevTables *T, const int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int n, int per, int rem)
{
	PRIM_T0();
	if constexpr (G::n == 64) { src = uni_ptr(src); dst = uni_ptr(dst); per = uni(per); rem = uni(rem); comp = uni(comp); }
	T = uni_ptr(T); n = uni(n); depth = uni(depth); is_intra = uni(is_intra);
	HENC_OP_IN_LDS(src); HENC_OP_IN_LDS(dst);
	const int inv_depth = 6 - (depth + (comp != 0));
	const int list = is_intra ? 0 : 3 + comp, rc = comp != 0;      // (the reference's expression: intra blocks of every component use list 0)
	const bool fast = F && F->valid && F->rem[rc] == rem;
	const int iq_shift = 20 - 14 - (15 - 8 - inv_depth) + 4, total = n * n;
	const int32_t add = iq_shift > per ? 1 << (iq_shift - per - 1) : 0;
	const int sh = iq_shift > per ? iq_shift - per : per - iq_shift;
	auto run = [&](const auto *cells, uint32_t flat) {      // (as in quantize: the 8 x 8 cells, cached or from T)
		for (int i = g.tid * 4; i < total; i += g.n * 4) {
			const S4 lv = ld4(src + i);
			uint32_t v[4];
			ft_list_value4(cells, flat, i, inv_depth, v);
			S4 o;
#pragma unroll
			for (int k = 0; k < 4; k++)
				o.v[k] = iq_shift > per ? sat16((int32_t)((uint32_t)(int32_t)lv.v[k] * v[k] + (uint32_t)add) >> sh) : sat16((int32_t)(((uint32_t)(int32_t)lv.v[k] * v[k]) << sh));
			st4(dst + i, o);
		}
	};
	const int kind = ft_list_kind(inv_depth, list);
	if (fast) run(F->iq8[rc][kind], (uint32_t)F->iq_flat[rc]);
	else run(T->dequant[1][kind ? 3 : 0][rem], (uint32_t)(uint16_t)T->dequant[0][0][rem][0]);
	g.sync();
	PRIM_END(PP_DEQUANT);
}

HENC_INLINE int chroma_qp_table(int qpi)   // chroma_scale_conversion_table, hmr_encoder_lib.c:2245
{
	static constexpr uint8_t mid[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
	qpi = hclip(qpi, 0, 57);
	return qpi < 30 ? qpi : (qpi < 44 ? mid[qpi - 30] : qpi - 6);
}

// find_scan_mode, hmr_tables.c:376
HENC_INLINE int find_scan_mode(int is_intra, int is_luma, int width, int dir_mode, int up_left_luma_dir_mode)
{
	if (!is_intra) return SCAN_DIAG;
	int ctx_idx;
	switch (width) {
	case 2: ctx_idx = 6; break;
	case 4: ctx_idx = 5; break;
	case 8: ctx_idx = 4; break;
	case 16: ctx_idx = 3; break;
	case 32: ctx_idx = 2; break;
	case 64: ctx_idx = 1; break;
	default: ctx_idx = 0; break;
	}
	int scan_idx = SCAN_DIAG;
	if (is_luma) {
		if (ctx_idx > 3 && ctx_idx < 6) scan_idx = habs(dir_mode - VER_IDX) < 5 ? SCAN_HOR : (habs(dir_mode - HOR_IDX) < 5 ? SCAN_VER : SCAN_DIAG);
	} else {
		if (dir_mode == DM_CHROMA_IDX) dir_mode = up_left_luma_dir_mode;
		if (ctx_idx > 4 && ctx_idx < 7) scan_idx = habs(dir_mode - VER_IDX) < 5 ? SCAN_HOR : (habs(dir_mode - HOR_IDX) < 5 ? SCAN_VER : SCAN_DIAG);
	}
	return scan_idx;
}

}  // namespace henc
