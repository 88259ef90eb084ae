// Entropy coding and bitstream of a frame: the serial consumer of the CTU decisions.
// Restates the reference's L1: the bit writer and NAL escaping (hmr_bitstream.c:63-145), the CABAC engine
// and its counting twin (hmr_binary_encoding.c:57-358, 553-576), context set-up (hmr_arithmetic_encoding.c:128-204), the CTU syntax
// (:358-1367, 1561-1832, 2039-2134) with the delta-QP syntax of the rate-controlled modes (:1371-1530), the SAO syntax (:1839-2036) and its rate terms
// (:2377-2429), the parameter sets and slice header (hmr_headers.c) and the access-unit assembly of encoder_engine_thread (hmr_encoder_lib.c:3287-3330) +
// HOMER_enc_write_annex_b_output (:2196).
// The CABAC engine and the CTU syntax are SPMD code like the decision core (one 64-lane group codes a CTU: the syntax walk is group-uniform, the coefficient
// scans are lane-parallel): on the device they run inside the CTU kernel as the last stage of a CTU's post-decision task (enc_post.h), one sub-stream per CTU
// row; the host writes parameter sets and slice headers and assembles the access unit.
#pragma once
#include <math.h>
#include <string.h>
#include <vector>
#include "enc_cabac_tables.h"
#include "enc_common.h"
#include "enc_sao.h"

#if defined(__HIPCC__)
#define HENC_HDX __host__ __device__
#define HENC_FI __host__ __device__ __forceinline__      // the entropy coder's state (engine registers, bit writer) must stay in registers: everything on its path is inlined
#else
#define HENC_HDX
#define HENC_FI inline __attribute__((always_inline))
#endif

namespace henc {

// ---- bit writer (MSB first) over a caller-owned buffer -----------------------------------------------------------------------
// A partially filled byte is kept in `cur`; `overflow` is set instead of writing past `cap`.
struct BitWriter {
	uint8_t *buf = nullptr;
	int cap = 0, bytecnt = 0, bitcnt = 0;
	uint32_t cur = 0;
	int overflow = 0;
	void (*grow)(BitWriter *) = nullptr;   // host owners (HostBits) enlarge the buffer here
	HENC_FI void attach(uint8_t *p, int capacity) { buf = p; cap = capacity; bytecnt = bitcnt = 0; cur = 0; overflow = 0; }
	HENC_FI void init() { bytecnt = bitcnt = 0; cur = 0; overflow = 0; }
	HENC_FI void put_byte(uint32_t b)
	{
		if (bytecnt >= cap) {
#if !defined(__HIP_DEVICE_COMPILE__)
			if (grow) grow(this);
#endif
			if (bytecnt >= cap) { overflow = 1; return; }
		}
		buf[bytecnt++] = (uint8_t)b;
	}
	HENC_FI void write(uint32_t val, int n)
	{
		if (n <= 0) return;
		if (n == 8 && bitcnt == 0) { put_byte(val & 255u); return; }
		const uint64_t acc = ((uint64_t)cur << n) | (n >= 32 ? (uint64_t)val : (uint64_t)(val & ((1u << n) - 1)));
		int total = bitcnt + n;
		while (total >= 8) {
			put_byte((uint32_t)(acc >> (total - 8)) & 255u);
			total -= 8;
		}
		cur = (uint32_t)(acc & ((1u << total) - 1));
		bitcnt = total;
	}
	HENC_FI void uvlc(uint32_t val)
	{
		uint32_t length = 1, temp = ++val;
		while (temp != 1) { temp >>= 1; length += 2; }
		write(0, length >> 1);
		write(val, (length + 1) >> 1);
	}
	HENC_FI void svlc(int val) { uvlc(val <= 0 ? (uint32_t)(-val) << 1 : ((uint32_t)val << 1) - 1); }
	HENC_FI void align0() { if (bitcnt) write(0, 8 - bitcnt); }
	HENC_FI void trailing_bits() { write(1, 1); align0(); }
	HENC_FI int bitcount() const { return (bytecnt << 3) + bitcnt; }
};

// a BitWriter that owns its (growing) buffer: host side
struct HostBits : BitWriter {
	std::vector<uint8_t> store;
	void bind() { buf = store.data(); cap = (int)store.size(); grow = &HostBits::enlarge; }
	HostBits() { store.assign(256, 0); bind(); }
	HostBits(const HostBits &o) : BitWriter(o), store(o.store) { bind(); }
	HostBits &operator=(const HostBits &o) { BitWriter::operator=(o); store = o.store; bind(); return *this; }
	static void enlarge(BitWriter *b)
	{
		HostBits *h = static_cast<HostBits *>(b);
		h->store.resize(h->store.size() * 2 + 64, 0);
		h->bind();
	}
};

// hmr_bitstream_nalu_ebsp :123 - the reference's escaping loop taken literally (it looks two bytes past the end, which are zero)
inline void nalu_ebsp(const uint8_t *data, int size, std::vector<uint8_t> &out)
{
	std::vector<uint8_t> p(data, data + size);
	p.resize(size + 8, 0);
	int i = 0;
	while (i < size) {
		while (p[i] != 0 || p[i + 1] != 0) {
			out.push_back(p[i]);
			if (i++ == size) break;
		}
		if (i++ >= size) break;
		out.push_back(0);
		out.push_back(0);
		if (p[++i] <= 3) out.push_back(3);
	}
}
inline void nalu_ebsp(const BitWriter &in, std::vector<uint8_t> &out) { nalu_ebsp(in.buf, in.bytecnt, out); }

// ---- CABAC --------------------------------------------------------------------------------------------------------------
// The engine registers are plain members (group-uniform values); the context states live where `ctx` points (the worker's fast memory on the device).
struct Cabac {
	uint32_t low = 0, range = 510, buffered_byte = 0xff;
	int num_buffered = 0, bits_left = 23;
	uint64_t frac_bits = 0;
	uint8_t *ctx = nullptr;          // [CTX_TOTAL]
	const uint8_t *t_range = &kRangeLps[0][0], *t_next = kNextStateLps;      // the LPS range and transition tables (the device keeps copies in the worker's fast memory)
	bool counter = false;
#if defined(HENC_POST_PROFILE)
	uint32_t nbins = 0;
#endif
	BitWriter bw;                    // the sub-stream being written

	HENC_FI void start() { low = 0; bits_left = 23; num_buffered = 0; buffered_byte = 0xff; range = 510; }
	HENC_FI void reset_bits() { low = 0; bits_left = 23; num_buffered = 0; buffered_byte = 0xff; frac_bits &= 32767; }
	HENC_FI static uint8_t init_state(int slice_type, int qp, int i)
	{
		const int init_value = kCtxInit[slice_type][i];
		const int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
		int st = ((slope * qp) >> 4) + offset;
		st = st < 1 ? 1 : (st > 126 ? 126 : st);
		const int mp = st >= 64;
		return (uint8_t)(((mp ? st - 64 : 63 - st) << 1) + mp);
	}
	template <class G>
	HENC_FI void init_contexts(const G g, int slice_type, int qp)
	{
		for (int i = g.tid; i < CTX_TOTAL; i += g.n) ctx[i] = init_state(slice_type, qp, i);
		g.sync();
	}
	HENC_FI void write_out()
	{
		const uint32_t lead = low >> (24 - bits_left);
		bits_left += 8;
		low &= 0xffffffffu >> bits_left;
		if (lead == 0xff) num_buffered++;
		else if (num_buffered > 0) {
			const uint32_t carry = lead >> 8;
			uint32_t byte = buffered_byte + carry;
			buffered_byte = lead & 0xff;
			bw.write(byte, 8);
			byte = (0xff + carry) & 0xff;
			while (num_buffered > 1) { bw.write(byte, 8); num_buffered--; }
		} else {
			num_buffered = 1;
			buffered_byte = lead;
		}
	}
	HENC_FI static int mps_tab(int s) { return s < 124 ? s + 2 : s; }   // g_bc_next_state_MPS: saturates at 124 / 125; 126 / 127 stay
#if defined(__HIP_DEVICE_COMPILE__)
	// On the device the coder's walk is one wavefront executing group-uniform code: the context states and the two tables live in registers, spread over the
	// lanes (context i: byte i >> 6 of lane i & 63 of ONE register; kRangeLps[s] as the four bytes of lane s; kNextStateLps[4 l .. 4 l + 3] in lane l), read with
	// v_readlane and written by the lane that holds them - a bin is arithmetic on scalar registers instead of four dependent trips to the LDS.  load_ctx /
	// store_ctx move the states between `ctx` (where the rest of the stage reads them) and the register.
	uint32_t cw = 0, trw = 0, tnw = 0, my_lane = 0;
	static_assert(CTX_TOTAL <= 192, "three context bytes per lane");
	template <class G>
	HENC_FI void load_ctx(const G g)
	{
		my_lane = (uint32_t)g.tid;
		cw = (uint32_t)ctx[g.tid] | (uint32_t)ctx[64 + g.tid] << 8 | (128 + g.tid < CTX_TOTAL ? (uint32_t)ctx[128 + g.tid] << 16 : 0u);
		trw = (uint32_t)kRangeLps[g.tid][0] | (uint32_t)kRangeLps[g.tid][1] << 8 | (uint32_t)kRangeLps[g.tid][2] << 16 | (uint32_t)kRangeLps[g.tid][3] << 24;
		const int l4 = (g.tid & 31) * 4;
		tnw = (uint32_t)kNextStateLps[l4] | (uint32_t)kNextStateLps[l4 + 1] << 8 | (uint32_t)kNextStateLps[l4 + 2] << 16 | (uint32_t)kNextStateLps[l4 + 3] << 24;
	}
	template <class G>
	HENC_FI void store_ctx(const G g)
	{
		ctx[g.tid] = (uint8_t)cw;
		ctx[64 + g.tid] = (uint8_t)(cw >> 8);
		if (128 + g.tid < CTX_TOTAL) ctx[128 + g.tid] = (uint8_t)(cw >> 16);
		g.sync();
	}
	HENC_FI void encode_bin(int ci_, uint32_t bin_)
	{
#if defined(HENC_POST_PROFILE)
		nbins++;
#endif
		const int ci = __builtin_amdgcn_readfirstlane(ci_);
		const uint32_t bin = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin_);
		const int sh = (ci >> 6) * 8;
		const uint32_t st = ((uint32_t)__builtin_amdgcn_readlane((int)cw, ci & 63) >> sh) & 255u;
		if (counter) {      // (the counting coder: see the host version below)
			frac_bits += (uint64_t)kEntropyBits[st ^ bin];
			cw = my_lane == (uint32_t)(ci & 63) ? (cw & ~(255u << sh)) : cw;
			return;
		}
		const uint32_t lps = ((uint32_t)__builtin_amdgcn_readlane((int)trw, (int)(st >> 1)) >> (((range >> 6) & 3) * 8)) & 255u;
		uint32_t nst;
		range -= lps;
		if (bin != (st & 1)) {
			const int nb = __builtin_clz(lps) - 23;
			low = (low + range) << nb;
			range = lps << nb;
			nst = ((uint32_t)__builtin_amdgcn_readlane((int)tnw, (int)(st >> 2)) >> ((st & 3) * 8)) & 255u;
			bits_left -= nb;
		} else {
			nst = (uint32_t)mps_tab((int)st);
			if (range < 256) {
				low <<= 1;
				range <<= 1;
				bits_left--;
			}
		}
		cw = my_lane == (uint32_t)(ci & 63) ? (cw & ~(255u << sh)) | (nst << sh) : cw;
		if (bits_left < 12) write_out();
	}
#else
	template <class G>
	HENC_FI void load_ctx(const G &) { }
	template <class G>
	HENC_FI void store_ctx(const G &) { }
	HENC_FI void encode_bin(int ci, uint32_t bin)
	{
#if defined(HENC_POST_PROFILE)
		nbins++;
#endif
		uint8_t st = ctx[ci];
		if (counter) {
			frac_bits += (uint64_t)kEntropyBits[st ^ bin];
			// the counting coder's transition table (g_bc_next_state, hmr_binary_encoding.c:305) is filled by bc_init_next_state_table(),
			// which nothing calls: it stays all zero, so every context the counter touches falls to state 0
			ctx[ci] = 0;
			return;
		}
		const uint32_t lps = t_range[((st >> 1) << 2) + ((range >> 6) & 3)];
		range -= lps;
		if (bin != (uint32_t)(st & 1)) {
			const int nb = __builtin_clz(lps) - 23;      // kRenorm[lps >> 3]: the shift that brings the LPS range (6 .. 240) back to nine bits
			low = (low + range) << nb;
			range = lps << nb;
			ctx[ci] = t_next[st];
			bits_left -= nb;
		} else {
			ctx[ci] = (uint8_t)mps_tab(st);
			if (range >= 256) return;
			low <<= 1;
			range <<= 1;
			bits_left--;
		}
		if (bits_left < 12) write_out();
	}
#endif
	HENC_FI void encode_ep(uint32_t bin)
	{
		bin = uni(bin);
		if (counter) { frac_bits += 32768; return; }
		low <<= 1;
		if (bin) low += range;
		bits_left--;
		if (bits_left < 12) write_out();
	}
	HENC_FI void encode_bins_ep(uint32_t bins, int n)
	{
		bins = uni(bins);
		n = uni(n);
		if (counter) { frac_bits += (uint64_t)32768 * n; return; }
		while (n > 8) {
			n -= 8;
			const uint32_t pattern = bins >> n;
			low <<= 8;
			low += range * pattern;
			bins -= pattern << n;
			bits_left -= 8;
			if (bits_left < 12) write_out();
		}
		low <<= n;
		low += range * bins;
		bits_left -= n;
		if (bits_left < 12) write_out();
	}
	HENC_FI void encode_trm(uint32_t bin)
	{
		bin = uni(bin);
		if (counter) { frac_bits += (uint64_t)kEntropyBits[126 ^ bin]; return; }
		range -= 2;
		if (bin) {
			low = (low + range) << 7;
			range = 2 << 7;
			bits_left -= 7;
		} else if (range >= 256) return;
		else {
			low <<= 1;
			range <<= 1;
			bits_left--;
		}
		if (bits_left < 12) write_out();
	}
	HENC_FI void finish()
	{
		if (low >> (32 - bits_left)) {
			bw.write(buffered_byte + 1, 8);
			while (num_buffered > 1) { bw.write(0x00, 8); num_buffered--; }
			low -= 1u << (32 - bits_left);
		} else {
			if (num_buffered > 0) bw.write(buffered_byte, 8);
			while (num_buffered > 1) { bw.write(0xff, 8); num_buffered--; }
		}
		bw.write(low >> 8, 24 - bits_left);
	}
	HENC_FI uint32_t bitcnt() const { return (uint32_t)(frac_bits >> 15); }
};

// ---- what the CTU syntax reads ---------------------------------------------------------------------------------------------------
// The CTU's own side-info record (on the device: a copy in the worker's fast memory, written back where the delta-QP rules change it), the records of the
// left and above CTUs where they exist, the CTU's levels, and the QP predictor that crosses the CTU boundary.
// A CTU's side-info as the syntax functions read it: pointers to the arrays (a CTU record's own, or - for the bit estimates of the decision stage, enc_rdo.h -
// the per-depth buffers of the CU under evaluation, which is what the reference's shadow CTU `ctu_rd` points at)
struct CtuView {
	const uint8_t *cbf[3], *intra_mode[2];
	const uint8_t *inter_mode, *tr_idx, *pred_depth, *part_size_type, *pred_mode, *skipped, *merge, *merge_idx, *mv_diff_ref_idx;
	uint8_t *qp;
	const MV *mv_diff;
	int x, y;
};
HENC_FI CtuView view_of(CtuPublic &c)
{
	CtuView v;
	for (int k = 0; k < 3; k++) v.cbf[k] = c.cbf[k];
	v.intra_mode[0] = c.intra_mode[0]; v.intra_mode[1] = c.intra_mode[1];
	v.inter_mode = c.inter_mode; v.tr_idx = c.tr_idx; v.pred_depth = c.pred_depth; v.part_size_type = c.part_size_type; v.pred_mode = c.pred_mode;
	v.skipped = c.skipped; v.merge = c.merge; v.merge_idx = c.merge_idx; v.mv_diff_ref_idx = c.mv_diff_ref_idx;
	v.qp = c.qp;
	v.mv_diff = c.mv_diff;
	v.x = c.x; v.y = c.y;
	return v;
}
struct EntView {
	const Seq *seq;
	const FrameCtx *f;
	const DevTables *T;
	GeoTable geo;
	const CtuView *c;             // this CTU
	const CtuView *left, *top;    // neighbours or nullptr
	const int16_t *coeff[3];      // the levels per component, linear per TU in z-order (a CTU's final levels: 4096 luma + 2 x 1024 chroma in one buffer)
	int n;                        // CTU index
	int prev_last_qp;             // QP of the last unit of the previous CTU of the sub-stream (get_last_coded_qp :1382), or -1: the slice QP
};
// Position (y << shift | x) of the i-th coefficient of a (1 << shift)^2 block in coding order: the coefficient groups in the order `cg` gives (their raster
// index in the grid of groups), the sixteen coefficients of a group in the 4 x 4 scan of the mode - horizontal: rows; vertical: columns; diagonal: up-right
// anti-diagonals from the bottom left (scan_pyramid, hmr_tables.c:62-160; the tables themselves are compared with this in tests/test_encoder_cpu.py).
HENC_INLINE int scan4x4_raster(int scan_mode, int k)      // raster index (y * 4 + x) of element k of the 4 x 4 scan
{
	if (scan_mode == SCAN_HOR) return k;
	if (scan_mode == SCAN_VER) return ((k & 3) << 2) | (k >> 2);
	return (int)((0xfbe7ad369c258140ull >> (4 * k)) & 15u);      // 0 4 1 8 5 2 12 9 6 3 13 10 7 14 11 15
}
HENC_INLINE uint32_t scan_position(int scan_mode, int shift, int i, const uint16_t *cg)
{
	const int r = scan4x4_raster(scan_mode, i & 15), sx = r & 3, sy = r >> 2;
	if (shift == 2) return (uint32_t)((sy << 2) | sx);
	const int cgp = cg[i >> 4], lb = shift - 2, cgx = cgp & ((1 << lb) - 1), cgy = cgp >> lb;
	return (uint32_t)((((cgy << 2) + sy) << shift) | ((cgx << 2) + sx));
}
// state of the delta-QP syntax across the CUs of a CTU (henc_thread_t write_qp_flag / curr_ref_qp / found_zero_cbf, hmr_private.h:1224-1226)
struct DqpState {
	int enabled, write_qp, ref_qp, found_coded;
};

HENC_INLINE const CtuView *ent_pu_left(const EntView &v, int ni, uint32_t *idx)
{
	const Geo &q = v.geo[ni];
	*idx = q.abs_left;
	return (q.raster_index & 15) == 0 ? v.left : v.c;
}
HENC_INLINE const CtuView *ent_pu_top(const EntView &v, int ni, uint32_t *idx, int planar)
{
	const Geo &q = v.geo[ni];
	*idx = q.abs_top;
	if (q.raster_index < 16) return planar ? nullptr : v.top;
	return v.c;
}
HENC_INLINE bool node_inside(const EntView &v, int ni)
{
	const Geo &q = v.geo[ni];
	return uni(v.c->y) + q.y + q.size <= uni(v.seq->height) && uni(v.c->x) + q.x + q.size <= uni(v.seq->width);
}
#define HENC_CBF(c, idx, comp, trd) (((uni((c)->cbf[comp][idx])) >> (trd)) & 1)

// get_intra_dir_luma_predictor :545 on the final arrays
HENC_INLINE void ent_intra_preds(const EntView &v, int ni, int *p)
{
	uint32_t idx = 0;
	const CtuView *l = ent_pu_left(v, ni, &idx);
	const int ld = l ? (uni(l->pred_mode[idx]) == PM_INTRA ? uni(l->intra_mode[0][idx]) : DC_IDX) : DC_IDX;
	const CtuView *t = ent_pu_top(v, ni, &idx, 1);
	const int td = t ? (uni(t->pred_mode[idx]) == PM_INTRA ? uni(t->intra_mode[0][idx]) : DC_IDX) : DC_IDX;
	if (ld == td) {
		if (ld > 1) { p[0] = ld; p[1] = ((ld + 29) % 32) + 2; p[2] = ((ld - 1) % 32) + 2; }
		else { p[0] = PLANAR_IDX; p[1] = DC_IDX; p[2] = VER_IDX; }
	} else {
		p[0] = ld; p[1] = td;
		if (ld && td) p[2] = PLANAR_IDX;
		else p[2] = (ld + td) < 2 ? VER_IDX : DC_IDX;
	}
}

#if defined(HENC_POST_PROFILE) && defined(__HIPCC__)
static __device__ unsigned long long g_ent_prof[6];      // ticks in encode_residual, calls, context-coded bins, ticks of its lane-parallel gather, bypass bins, spare
#endif
#if defined(HENC_POST_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define ENT_PROF_T0() const unsigned long long ent_t0_ = __builtin_amdgcn_s_memtime()
#define ENT_PROF_ADD(k, v) do { if (threadIdx.x % 64 == 0) atomicAdd(&g_ent_prof[k], (unsigned long long)(v)); } while (0)
#define ENT_PROF_NOW() (__builtin_amdgcn_s_memtime() - ent_t0_)
#else
#define ENT_PROF_T0() do { } while (0)
#define ENT_PROF_ADD(k, v) do { } while (0)
#define ENT_PROF_NOW() 0
#endif
// ---- residual coding (encode_residual :1087, encode_last_significant_XY :954, get_sig_ctx_inc :1027) ---------------------------
HENC_INLINE int sig_ctx_inc(int pattern, int scan_mode, int px, int py, int shift, int comp)
{
	if (px + py == 0) return 0;
	if (shift == 2) {
		// {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}[4 * py + px], four bits each
		return (int)((0x8877886654325410ull >> (4 * (4 * py + px))) & 15u);
	}
	const int offset = shift == 3 ? (scan_mode == SCAN_DIAG ? 9 : 15) : (comp == COMP_Y ? 21 : 12);
	const int xs = px & 3, ys = py & 3;
	int cnt;
	if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
	else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
	else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
	else cnt = 2;
	return ((comp == COMP_Y && ((px >> 2) + (py >> 2)) > 0) ? 3 : 0) + offset + cnt;
}

HENC_INLINE int last_group_idx(int v) { return v < 4 ? v : (v < 6 ? 4 : (v < 8 ? 5 : (v < 12 ? 6 : (v < 16 ? 7 : (v < 24 ? 8 : 9))))); }
HENC_INLINE int last_min_in_group(int g) { return g < 4 ? g : (g == 4 ? 4 : (g == 5 ? 6 : (g == 6 ? 8 : (g == 7 ? 12 : (g == 8 ? 16 : 24))))); }

HENC_FI void encode_last_xy(Cabac &ee, int x, int y, int shift, int comp, int scan_mode)
{
	const int size = 1 << shift;
	const int cx = CTX_LAST_X + (comp ? 15 : 0), cy = CTX_LAST_Y + (comp ? 15 : 0);
	if (scan_mode == SCAN_VER) { const int t = x; x = y; y = t; }
	const int gx = last_group_idx(x), gy = last_group_idx(y);
	const int off = comp ? 0 : ((shift - 2) * 3 + (((shift - 2) + 1) >> 2));
	const int sh = comp ? (shift - 2) : (((shift - 2) + 3) >> 2);
	int k;
	for (k = 0; k < gx; k++) ee.encode_bin(cx + off + (k >> sh), 1);
	if (gx < last_group_idx(size - 1)) ee.encode_bin(cx + off + (k >> sh), 0);
	for (k = 0; k < gy; k++) ee.encode_bin(cy + off + (k >> sh), 1);
	if (gy < last_group_idx(size - 1)) ee.encode_bin(cy + off + (k >> sh), 0);
	if (gx > 3) {
		const int count = (gx - 2) >> 1;
		x -= last_min_in_group(gx);
		for (int i = count - 1; i >= 0; i--) ee.encode_ep((x >> i) & 1);
	}
	if (gy > 3) {
		const int count = (gy - 2) >> 1;
		for (int i = count - 1; i >= 0; i--) ee.encode_ep((y >> i) & 1);   // the reference leaves y unreduced (:1010); the low bits are the same
	}
}

// The TU's levels are first gathered in scan order into the scratch (lane-parallel: on the device the levels and the scan table are in HBM, and the serial
// walk below would pay a trip to memory per coefficient), together with the last significant position and the coefficient-group flags; the syntax walk then
// reads the scratch only.
template <class G>
HENC_FI void encode_residual(const G g, Cabac &ee, const EntView &v, EntScratch &sc, int ni, int comp)
{
	const Seq &S = *v.seq;
	ENT_PROF_T0();
	ENT_PROF_ADD(1, 1);
	const int is_luma = comp == COMP_Y;
	const int abs_index = v.geo[ni].abs_index;
	const int pi = is_luma ? ni : (v.geo[ni].size_chroma > 2 ? ni : v.geo[ni].parent);
	const Geo &q = v.geo[pi];
	const int size = is_luma ? q.size : q.size_chroma;
	const int shift = is_luma ? CFG_MAX_CU_SHIFT - q.depth : CFG_MAX_CU_SHIFT - 1 - q.depth;
	const int16_t *coeff = v.coeff[comp] + (q.abs_index << (4 - (is_luma ? 0 : 2)));
	const CtuView *c = v.c;
	const int num_part_in_pred_cu = NPART >> (uni(c->pred_depth[abs_index]) * 2);
	const int scan_mode = find_scan_mode(uni(c->pred_mode[q.abs_index]) == PM_INTRA, is_luma, size, uni(c->intra_mode[is_luma ? 0 : 1][q.abs_index]),
					     uni(c->intra_mode[0][(abs_index / num_part_in_pred_cu) * num_part_in_pred_cu]));
	const int blk = size >> 2, ncoef = size * size, ncg = blk * blk;
	// the coefficient-group scan: {0, 1, 2, 3} / {0, 2, 1, 3} for 8 x 8 (:1147-1150), the plain up-right diagonal of the 8 x 8 grid of groups for 32 x 32
	// (g_sigLastScanCG32x32, hmr_tables.c:71-95), the 4 x 4 scan of the mode for 16 x 16
	for (int i = g.tid; i < ncg; i += g.n) {
		uint32_t cgp;
		if (shift == 3) cgp = (scan_mode == SCAN_VER || scan_mode == SCAN_DIAG) ? ((uint32_t)((i & 1) << 1) | (uint32_t)(i >> 1)) : (uint32_t)i;
		else if (shift == 5) {
			// i-th element of the up-right diagonal scan of an 8 x 8 grid (rows from the bottom of each anti-diagonal)
			int d = 0, k = i;
			for (;; d++) {
				const int len = d < 8 ? d + 1 : 15 - d;
				if (k < len) break;
				k -= len;
			}
			const int row = (d < 8 ? d : 7) - k, col = d - row;
			cgp = (uint32_t)(row * 8 + col);
		} else cgp = shift > 3 ? (uint32_t)scan4x4_raster(scan_mode, i) : 0u;
		sc.cg[i] = (uint16_t)cgp;
		sc.cg_flag[i] = 0;
	}
	g.sync();
#if defined(__HIP_DEVICE_COMPILE__)
	// on the device the groups' positions also sit in a register (lane i: group i of the scan; at most 64 groups) and the "group holds a level" flags in a 64-bit
	// mask indexed by the group's raster position: the serial walk below reads neither from memory (a 32 x 32 TU walks up to 64 groups, most of them empty)
	const int cgv = g.tid < ncg ? (int)sc.cg[g.tid] : 0;
	uint64_t cgmask = 0;
#endif
	// lane-parallel: the last significant coefficient in coding order and the groups that hold one
	int raster_pos_last = -1;
	for (int base = 0; base < ncoef; base += g.n) {
		const int i = base + g.tid;
		int16_t val = 0;
		if (i < ncoef) {
			const uint32_t sp = scan_position(scan_mode, shift, i, sc.cg);
			val = coeff[sp];
#if !defined(__HIP_DEVICE_COMPILE__)
			if (val != 0) sc.cg_flag[sc.cg[i >> 4]] = 1;
#endif
		}
		const uint64_t m = g.ballot(val != 0);
		if (m) raster_pos_last = base + 63 - __builtin_clzll(m);
#if defined(__HIP_DEVICE_COMPILE__)
		for (int j = 0; j < 4; j++)
			if ((m >> (16 * j)) & 0xffffu) cgmask |= 1ull << __builtin_amdgcn_readlane(cgv, (base >> 4) + j);
#endif
	}
	g.sync();
	ENT_PROF_ADD(3, ENT_PROF_NOW());
	if (raster_pos_last < 0) return;
	const int scan_pos_last = uni((int)scan_position(scan_mode, shift, raster_pos_last, sc.cg));
	const int last_y = scan_pos_last >> shift, last_x = scan_pos_last - (last_y << shift);
	const int valid = uni(S.sign_hiding);
	encode_last_xy(ee, last_x, last_y, shift, comp, scan_mode);
	const int last_scan_set = raster_pos_last >> 4;
	uint32_t c1 = 1, go_rice;
	int scan_pos_sig = raster_pos_last;
	const int base_cg = CTX_SIG_CG + (is_luma ? 0 : 2), base_sig = CTX_SIG + (is_luma ? 0 : 27);
	for (int subset = last_scan_set; subset >= 0; subset--) {
		int num_non_zero = 0;
		const int sub_pos = subset << 4;
		uint32_t coeff_signs = 0;
		int last_nz = -1, first_nz = 16;
		// the absolute levels of the subset in coding order, four bits short of what a register holds: kept as two 64-bit words of 16-bit fields + the rest
		uint64_t abs_lo = 0, abs_mid = 0, abs_hi = 0, abs_top = 0;      // abs_coeff[0..3], [4..7], [8..11], [12..15]
		auto put_abs = [&](int k, int a) {
			const uint64_t x = (uint64_t)(uint16_t)a << (16 * (k & 3));
			if (k < 4) abs_lo |= x; else if (k < 8) abs_mid |= x; else if (k < 12) abs_hi |= x; else abs_top |= x;
		};
		auto get_abs = [&](int k) -> int {
			const uint64_t w = k < 4 ? abs_lo : (k < 8 ? abs_mid : (k < 12 ? abs_hi : abs_top));
			return (int)((w >> (16 * (k & 3))) & 0xffffu);
		};
		go_rice = 0;
		if (scan_pos_sig == raster_pos_last) {
			const int lv = uni((int)coeff[scan_pos_last]);
			put_abs(0, habs(lv));
			coeff_signs = lv < 0;
			num_non_zero = 1;
			last_nz = first_nz = scan_pos_sig;
			scan_pos_sig--;
		}
#if defined(__HIP_DEVICE_COMPILE__)
		const int cg_block_pos = __builtin_amdgcn_readlane(cgv, subset);
#define HENC_CG_FLAG(pos) ((uint32_t)((cgmask >> (pos)) & 1ull))
#define HENC_CG_SET(pos) (cgmask |= 1ull << (pos))
#else
		const int cg_block_pos = sc.cg[subset];
#define HENC_CG_FLAG(pos) ((uint32_t)(sc.cg_flag[pos] != 0))
#define HENC_CG_SET(pos) (sc.cg_flag[pos] = 1)
#endif
		const int cg_y = cg_block_pos / blk, cg_x = cg_block_pos - cg_y * blk;
		if (subset == last_scan_set || subset == 0) HENC_CG_SET(cg_block_pos);
		else {
			const uint32_t sig_cg = HENC_CG_FLAG(cg_block_pos);
			int right = 0, lower = 0;
			if (cg_x < blk - 1) right = HENC_CG_FLAG(cg_y * blk + cg_x + 1);
			if (cg_y < blk - 1) lower = HENC_CG_FLAG((cg_y + 1) * blk + cg_x);
			ee.encode_bin(base_cg + (right || lower), sig_cg);
		}
		if (HENC_CG_FLAG(cg_block_pos)) {
			uint32_t right = 0, lower = 0;
			if (cg_x < blk - 1) right = HENC_CG_FLAG(cg_y * blk + cg_x + 1);
			if (cg_y < blk - 1) lower = HENC_CG_FLAG((cg_y + 1) * blk + cg_x);
			const int pattern = right + (lower << 1);
#if defined(__HIP_DEVICE_COMPILE__)
			// the group's sixteen positions and levels by sixteen lanes at once: the serial walk below reads them from the register (two trips to the LDS per
			// group instead of two per coefficient)
			const uint32_t bp_lane = scan_position(scan_mode, shift, sub_pos + (g.tid & 15), sc.cg);
			const int lv_lane = coeff[bp_lane];
#endif
			for (; scan_pos_sig >= sub_pos; scan_pos_sig--) {
#if defined(__HIP_DEVICE_COMPILE__)
				const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)bp_lane, scan_pos_sig - sub_pos);
				const int lv = __builtin_amdgcn_readlane(lv_lane, scan_pos_sig - sub_pos);
#else
				const uint32_t bp = scan_position(scan_mode, shift, scan_pos_sig, sc.cg);
				const int lv = coeff[bp];
#endif
				const uint32_t py = bp >> shift, px = bp - (py << shift);
				const uint32_t sig = lv != 0;
				if (scan_pos_sig > sub_pos || subset == 0 || num_non_zero) ee.encode_bin(base_sig + sig_ctx_inc(pattern, scan_mode, px, py, shift, comp), sig);
				if (sig) {
					put_abs(num_non_zero, habs(lv));
					coeff_signs = 2 * coeff_signs + (lv < 0);
					num_non_zero++;
					if (last_nz == -1) last_nz = scan_pos_sig;
					first_nz = scan_pos_sig;
				}
			}
		} else scan_pos_sig = sub_pos - 1;
		if (num_non_zero > 0) {
			const int sign_hidden = (last_nz - first_nz >= 4);
			uint32_t ctx_set = (subset > 0 && is_luma) ? 2 : 0;
			if (c1 == 0) ctx_set++;
			c1 = 1;
			int base = CTX_ONE + 4 * ctx_set + (is_luma ? 0 : 16);
			const int num_c1 = num_non_zero < 8 ? num_non_zero : 8;
			int first_c2 = -1;
			for (int idx = 0; idx < num_c1; idx++) {
				const uint32_t sym = get_abs(idx) > 1;
				ee.encode_bin(base + c1, sym);
				if (sym) {
					c1 = 0;
					if (first_c2 == -1) first_c2 = idx;
				} else if (c1 < 3 && c1 > 0) c1++;
			}
			if (c1 == 0) {
				base = CTX_ABS + ctx_set + (is_luma ? 0 : 4);
				if (first_c2 != -1) ee.encode_bin(base, get_abs(first_c2) > 2);
			}
			if (valid && sign_hidden) ee.encode_bins_ep(coeff_signs >> 1, num_non_zero - 1);
			else ee.encode_bins_ep(coeff_signs, num_non_zero);
			if (c1 == 0 || num_non_zero > 8) {
				int first_coeff2 = 1;
				for (int idx = 0; idx < num_non_zero; idx++) {
					const int base_level = idx < 8 ? (2 + first_coeff2) : 1;
					const int a = get_abs(idx);
					if (a >= base_level) {
						int code = a - base_level;
						const uint32_t r = go_rice;
						if (code < (3 << r)) {
							const uint32_t length = code >> r;
							ee.encode_bins_ep((1u << (length + 1)) - 2, length + 1);
							ee.encode_bins_ep(code % (1 << r), r);
						} else {
							uint32_t length = r;
							code -= 3 << r;
							while (code >= (1 << length)) code -= 1 << (length++);
							ee.encode_bins_ep((1u << (3 + length + 1 - r)) - 2, 3 + length + 1 - r);
							ee.encode_bins_ep(code, length);
						}
						if (a > 3 * (1 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
					}
					if (a >= 2) first_coeff2 = 0;
				}
			}
		}
	}
	ENT_PROF_ADD(0, ENT_PROF_NOW());
}
#undef HENC_CG_FLAG
#undef HENC_CG_SET

// ---- CU syntax -----------------------------------------------------------------------------------------------------------------
HENC_FI void encode_qt_cbf(Cabac &ee, int comp, int tr_depth, int cbf)
{
	const int ctx = comp ? tr_depth : (tr_depth == 0 ? 1 : 0);
	ee.encode_bin(CTX_QT_CBF + (comp ? 4 : 0) + ctx, cbf);
}

// encode_delta_qp :1502 (TEncSbac::codeDeltaQP) for the CU at `abs_index`; the predictor is DqpState::ref_qp: with diff_cu_qp_delta_depth = 0 (the only
// value the reference sets under rate control, hmr_encoder_lib.c:983) a quantisation group is a CTU, it has no left / above group inside the CTU
// (get_qp_min_cu_left / _top :1432 / :1459 return NULL at the CTU boundary), so get_ref_qp :1487 is the last coded QP
HENC_FI void encode_delta_qp(Cabac &ee, const EntView &v, const DqpState &dq, int abs_index)
{
	int diff_qp = (int)uni(v.c->qp[abs_index]) - dq.ref_qp;
	diff_qp = (diff_qp + 78) % 52 - 26;
	const uint32_t a = (uint32_t)habs(diff_qp), tu = a < 5 ? a : 5;      // CU_DQP_TU_CMAX 5, CU_DQP_EG_k 0
	// write_unary_max_simbol :358 (offset 1, max 5)
	ee.encode_bin(CTX_DQP, tu ? 1 : 0);
	if (tu) {
		for (uint32_t k = 1; k < tu; k++) ee.encode_bin(CTX_DQP + 1, 1);
		if (5 > tu) ee.encode_bin(CTX_DQP + 1, 0);
	}
	if (a >= 5) {
		// write_ep_ex_golomb :703
		uint32_t symbol = a - 5, count = 0, bins = 0;
		int nb = 0;
		while (symbol >= (1u << count)) { bins = 2 * bins + 1; nb++; symbol -= 1u << count; count++; }
		bins = 2 * bins;
		nb++;
		bins = (bins << count) | symbol;
		nb += (int)count;
		ee.encode_bins_ep(bins, nb);
	}
	if (a > 0) ee.encode_ep(diff_qp > 0 ? 0 : 1);
}

// transform_tree :1561
template <class G>
HENC_FI void encode_transform_tree(const G g, Cabac &ee, const EntView &v, EntScratch &sc, DqpState &dq, int top_ni)
{
	const Seq &S = *v.seq;
	const CtuView *c = v.c;
	const int depth = v.geo[top_ni].depth, top_abs = v.geo[top_ni].abs_index;
	int abs_index = top_abs;
	int is_intra = uni(c->pred_mode[abs_index]) == PM_INTRA;
	if (!is_intra) {
		const uint32_t qtroot = HENC_CBF(c, abs_index, 0, 0) || HENC_CBF(c, abs_index, 1, 0) || HENC_CBF(c, abs_index, 2, 0);
		if (!(uni(c->merge[abs_index]) && uni(c->part_size_type[abs_index]) == PART_2Nx2N)) ee.encode_bin(CTX_QT_ROOT_CBF, qtroot);
		if (!qtroot) return;
	}
	DepthState depth_state;
	int curr = top_ni, parent = top_ni, curr_depth = depth;
	while (curr_depth != depth || depth_state.get(curr_depth) != 1) {
		const Geo &q = v.geo[curr];
		curr_depth = q.depth;
		abs_index = q.abs_index;
		const int shift = CFG_MAX_CU_SHIFT - curr_depth;
		const int pred_depth = uni(c->pred_depth[abs_index]), tr_depth = curr_depth - pred_depth, first = tr_depth == 0;
		const int tr_idx = uni(c->tr_idx[abs_index]);
		is_intra = uni(c->pred_mode[abs_index]) == PM_INTRA;
		const int part = uni(c->part_size_type[abs_index]);
		const int split_flag = (tr_idx + pred_depth) > curr_depth;
		const int log2_tr = CFG_MAX_CU_SHIFT - curr_depth, log2_cu = CFG_MAX_CU_SHIFT - pred_depth;
		const int intra_split = is_intra && part == PART_NxN, inter_split = !is_intra && uni(S.max_inter_tr_depth) == 1 && part != PART_2Nx2N;
		const int max_tr = is_intra ? uni(S.max_intra_tr_depth) : uni(S.max_inter_tr_depth);
		int tu_min_in_cu;
		if (log2_cu < uni(S.min_tu_size_shift) + max_tr - 1 + inter_split + intra_split) tu_min_in_cu = uni(S.min_tu_size_shift);
		else {
			tu_min_in_cu = log2_cu - (max_tr - 1 + inter_split + intra_split);
			if (tu_min_in_cu > uni(S.max_tu_size_shift)) tu_min_in_cu = uni(S.max_tu_size_shift);
		}
		if (!(is_intra && part == PART_NxN && curr_depth == pred_depth) && !(!is_intra && part != PART_2Nx2N && curr_depth == pred_depth && uni(S.max_inter_tr_depth) == 1) &&
		    !(log2_tr > uni(S.max_tu_size_shift)) && !(log2_tr == uni(S.min_tu_size_shift)) && !(log2_tr == tu_min_in_cu))
			ee.encode_bin(CTX_TRANS_SUBDIV + 5 - shift, split_flag);
		if (first || shift > 2) {
			if (first || HENC_CBF(c, abs_index, 1, tr_depth - 1)) encode_qt_cbf(ee, 1, tr_depth, HENC_CBF(c, abs_index, 1, tr_depth));
			if (first || HENC_CBF(c, abs_index, 2, tr_depth - 1)) encode_qt_cbf(ee, 2, tr_depth, HENC_CBF(c, abs_index, 2, tr_depth));
		}
		depth_state.inc(curr_depth);
		if (split_flag) {
			parent = curr;
			curr_depth++;
		} else {
			const uint32_t cbf_y = HENC_CBF(c, abs_index, 0, tr_depth), cbf_u = HENC_CBF(c, abs_index, 1, tr_depth), cbf_v = HENC_CBF(c, abs_index, 2, tr_depth);
			if (uni(c->pred_mode[abs_index]) == PM_INTRA || tr_depth != 0 || cbf_u || cbf_v) encode_qt_cbf(ee, 0, tr_depth, cbf_y);
			if ((cbf_y || cbf_u || cbf_v) && dq.enabled && dq.write_qp) {      // the delta QP: once per quantisation group, with the first coded TU (:1660-1674)
				encode_delta_qp(ee, v, dq, top_abs);
				dq.write_qp = 0;
			}
			// luma, then chroma - of a 4 x 4 luma block's quadruple with the last of the four (:1690-1712).  (One call site: the coder is inlined.)
			const bool chroma_here = shift > 2 || q.list_index == v.geo[v.geo[q.parent].child[0]].list_index + 3;
			for (int comp = 0; comp < 3; comp++) {
				const bool coded = comp == 0 ? cbf_y != 0 : (chroma_here && (comp == 1 ? cbf_u : cbf_v) != 0);
				if (coded) encode_residual(g, ee, v, sc, curr, comp);
			}
			while (depth_state.get(curr_depth) == 4) {
				depth_state.set(curr_depth, 0);
				parent = v.geo[parent].parent;
				curr_depth--;
			}
			if (curr_depth == 0 && depth_state.get(curr_depth) == 1) break;
		}
		if (curr_depth == depth && depth_state.get(curr_depth) == 1) break;
		curr = v.geo[parent].child[depth_state.get(curr_depth)];
	}
}

HENC_FI void encode_mvd(Cabac &ee, const CtuView *c, int idx)
{
	const int h = uni(c->mv_diff[idx].x), ver = uni(c->mv_diff[idx].y);
	const int h0 = h != 0, v0 = ver != 0, ha = habs(h), va = habs(ver);
	ee.encode_bin(CTX_MVD, h0);
	ee.encode_bin(CTX_MVD, v0);
	if (h0) ee.encode_bin(CTX_MVD + 1, ha > 1);
	if (v0) ee.encode_bin(CTX_MVD + 1, va > 1);
	auto golomb = [&](uint32_t symbol, uint32_t count) {
		uint32_t bins = 0;
		int nb = 0;
		while (symbol >= (1u << count)) { bins = 2 * bins + 1; nb++; symbol -= 1u << count; count++; }
		bins = 2 * bins;
		nb++;
		bins = (bins << count) | symbol;
		nb += count;
		ee.encode_bins_ep(bins, nb);
	};
	if (h0) { if (ha > 1) golomb(ha - 2, 1); ee.encode_ep(h < 0); }
	if (v0) { if (va > 1) golomb(va - 2, 1); ee.encode_ep(ver < 0); }
}

// encode_end_of_cu :1723
HENC_FI void encode_end_of_cu(Cabac &ee, const EntView &v, int ni)
{
	const Seq &S = *v.seq;
	const Geo &q = v.geo[ni];
	const uint32_t cu_addr = (uint32_t)v.n * NPART + q.abs_index;
	const int width = uni(S.width), height = uni(S.height);
	uint32_t real_end;
	if (width % 64 || height % 64) {
		int wr = (width % 64) >> 2, hr = (height % 64) >> 2;
		if (hr == 0) hr = 15;
		else if (wr) hr -= 1;
		const int aux = hr * 16 + wr;
		real_end = (uint32_t)uni(S.nctu) * NPART - NPART + raster2abs(aux - 1) + 1;
	} else real_end = (uint32_t)uni(S.nctu) * NPART;
	const int px = uni(v.c->x) + q.x, py = uni(v.c->y) + q.y;
	const int boundary = ((px + q.size) % 64 == 0 || (px + q.size) == width) && ((py + q.size) % 64 == 0 || (py + q.size) == height);
	const int terminate = cu_addr + q.num_part == real_end;
	if (boundary && !terminate) ee.encode_trm(0);
}

// ee_encode_coding_unit :1787
template <class G>
HENC_FI void encode_coding_unit(const G g, Cabac &ee, const EntView &v, EntScratch &sc, DqpState &dq, int ni)
{
	const Seq &S = *v.seq;
	const CtuView *c = v.c;
	const Geo &q = v.geo[ni];
	const int abs_index = q.abs_index, is_intra = uni(c->pred_mode[abs_index]) == PM_INTRA, part = uni(c->part_size_type[abs_index]);
	const int p_slice = uni(v.f->slice_type) != SLICE_I;
	uint32_t idx = 0;
	if (p_slice) {
		const CtuView *l = ent_pu_left(v, ni, &idx);
		int ctx = l ? (uni(l->skipped[idx]) ? 1 : 0) : 0;
		const CtuView *t = ent_pu_top(v, ni, &idx, 0);
		ctx += t ? (uni(t->skipped[idx]) ? 1 : 0) : 0;
		ee.encode_bin(CTX_SKIP_FLAG + ctx, uni(c->skipped[abs_index]));
	}
	auto merge_index = [&](int a) {
		// encode_merge_index :613 with two candidates: one context-coded bin
		const uint32_t unary = uni(c->merge_idx[a]);
		for (int ui = 0; ui < CFG_NUM_MERGE_CAND - 1; ui++) {
			const uint32_t sym = ui == (int)unary ? 0 : 1;
			if (ui == 0) ee.encode_bin(CTX_MERGE_IDX, sym);
			else ee.encode_ep(sym);
			if (sym == 0) break;
		}
	};
	if (uni(c->skipped[abs_index])) {
		merge_index(abs_index);
		encode_end_of_cu(ee, v, ni);
		return;
	}
	if (p_slice) ee.encode_bin(CTX_PRED_MODE, uni(c->pred_mode[abs_index]));
	// encode_part_size :436
	const int min_cu_depth = uni(S.max_cu_depth) - uni(S.mincu_mintr_shift_diff);
	if (is_intra) {
		if (q.depth == min_cu_depth) ee.encode_bin(CTX_PART_SIZE, part == PART_2Nx2N ? 1 : 0);
	} else if (part == PART_2Nx2N) ee.encode_bin(CTX_PART_SIZE, 1);
	else if (part == PART_NxN) {
		if (q.depth == min_cu_depth && !(q.size == 8)) {
			ee.encode_bin(CTX_PART_SIZE, 0);
			ee.encode_bin(CTX_PART_SIZE + 1, 0);
			ee.encode_bin(CTX_PART_SIZE + 2, 0);
		}
	}
	if (is_intra) {
		// encode_intra_dir_luma_ang :838 (is_multiple)
		const int part_num = part == PART_NxN ? 4 : 1;
		int dir[4], preds[4][3], pred_idx[4] = {-1, -1, -1, -1};
		for (int j = 0; j < part_num; j++) {
			const int pn = part_num == 4 ? q.child[j] : ni;
			dir[j] = uni(c->intra_mode[0][v.geo[pn].abs_index]);
			ent_intra_preds(v, pn, preds[j]);
			for (int i = 0; i < 3; i++)
				if (dir[j] == preds[j][i]) pred_idx[j] = i;
			ee.encode_bin(CTX_INTRA_PRED, pred_idx[j] != -1 ? 1 : 0);
		}
		for (int j = 0; j < part_num; j++) {
			if (pred_idx[j] != -1) {
				ee.encode_ep(pred_idx[j] ? 1 : 0);
				if (pred_idx[j]) ee.encode_ep(pred_idx[j] - 1);
			} else {
				int *p = preds[j];
				if (p[0] > p[1]) { const int t = p[0]; p[0] = p[1]; p[1] = t; }
				if (p[0] > p[2]) { const int t = p[0]; p[0] = p[2]; p[2] = t; }
				if (p[1] > p[2]) { const int t = p[1]; p[1] = p[2]; p[2] = t; }
				for (int i = 2; i >= 0; i--) dir[j] = dir[j] > p[i] ? dir[j] - 1 : dir[j];
				ee.encode_bins_ep(dir[j], 5);
			}
		}
		// encode_intra_dir_chroma :907
		uint32_t chroma = uni(c->intra_mode[1][abs_index]);
		if (chroma == DM_CHROMA_IDX) ee.encode_bin(CTX_CHROMA_PRED, 0);
		else {
			int list[5];
			const int luma = uni(c->intra_mode[0][abs_index]);
			list[0] = PLANAR_IDX; list[1] = VER_IDX; list[2] = HOR_IDX; list[3] = DC_IDX; list[4] = DM_CHROMA_IDX;
			for (int i = 0; i < 4; i++)
				if (luma == list[i]) { list[i] = 34; break; }
			for (int i = 0; i < 4; i++)
				if ((int)chroma == list[i]) { chroma = i; break; }
			ee.encode_bin(CTX_CHROMA_PRED, 1);
			ee.encode_bins_ep(chroma, 2);
		}
	} else {
		// encode_inter_motion_info :777, P slice with one reference picture
		const int num_pu = part == PART_2Nx2N ? 1 : (part == PART_NxN ? 4 : 2);
		const uint32_t pu_off_part = part == PART_2Nx2N ? 0u : (part == 1 ? 8u : (part == 2 ? 4u : (part == 3 ? 4u : (part == 4 ? 2u : (part == 5 ? 10u : (part == 6 ? 1u : 5u))))));   // {0, 8, 4, 4, 2, 10, 1, 5}
		const uint32_t pu_offset = (pu_off_part << ((uni(S.max_cu_depth) - uni(c->pred_depth[abs_index])) << 1)) >> 4;
		for (int p = 0, sub = abs_index; p < num_pu; p++, sub += pu_offset) {
			ee.encode_bin(CTX_MERGE_FLAG, uni(c->merge[sub]));
			if (uni(c->merge[sub])) merge_index(sub);
			else if (uni(c->inter_mode[sub]) & 1) {
				encode_mvd(ee, c, sub);
				ee.encode_bin(CTX_MVP_IDX, uni(c->mv_diff_ref_idx[sub]) ? 1 : 0);
			}
		}
	}
	encode_transform_tree(g, ee, v, sc, dq, ni);
	encode_end_of_cu(ee, v, ni);
}

// ee_encode_ctu :2039.  Under rate control the walk also applies the reference's QP rule for uncoded CUs: a CU without levels that comes before the first
// coded CU of its quantisation group (the CTU) takes the predictor as its QP (:2091-2104) - written into the record, where the next CTU's predictor and the
// delta-QP of this one read it.
template <class G>
HENC_FI void encode_ctu_syntax(const G g, Cabac &ee, const EntView &v, EntScratch &sc)
{
	const Seq &S = *v.seq;
	DepthState depth_state;
	int curr = 0, curr_depth = 0;
	const int min_cu_depth = uni(S.max_cu_depth) - uni(S.mincu_mintr_shift_diff);
	DqpState dq = {uni(S.bitrate_mode) != 0, 1, 0, 0};
	while (curr_depth != 0 || depth_state.get(curr_depth) != 1) {
		const Geo &q = v.geo[curr];
		const bool inside = node_inside(v, curr);
		if (inside && q.depth != min_cu_depth) {
			// encode_split_flag :391
			uint32_t idx = 0;
			const int split = uni(v.c->pred_depth[q.abs_index]) > q.depth;
			const CtuView *l = ent_pu_left(v, curr, &idx);
			int ctx = l ? (uni(l->pred_depth[idx]) > q.depth ? 1 : 0) : 0;
			const CtuView *t = ent_pu_top(v, curr, &idx, 0);
			ctx += t ? (uni(t->pred_depth[idx]) > q.depth ? 1 : 0) : 0;
			ee.encode_bin(CTX_SPLIT_FLAG + ctx, split);
		}
		if (dq.enabled && q.depth == 0) {      // a new quantisation group (diff_cu_qp_delta_depth = 0): get_ref_qp :1487 = the last coded QP
			dq.ref_qp = v.prev_last_qp >= 0 ? v.prev_last_qp : uni(v.f->qp);
			dq.found_coded = 0;
		}
		const int pred_depth = uni(v.c->pred_depth[q.abs_index]);
		depth_state.inc(curr_depth);
		if (curr_depth < pred_depth) {
			if (dq.enabled && depth_state.get(curr_depth) == 1 && curr_depth == 0) dq.write_qp = 1;
			curr_depth++;
			curr = q.child[depth_state.get(curr_depth)];
		} else {
			if (inside) {
				if (dq.enabled && !dq.found_coded) {
					if (uni(v.c->cbf[0][q.abs_index]) || uni(v.c->cbf[1][q.abs_index]) || uni(v.c->cbf[2][q.abs_index])) {
						dq.found_coded = 1;
						if (depth_state.get(curr_depth) > 1) dq.write_qp = 1;
					} else {
						for (int i = g.tid; i < q.num_part; i += g.n) v.c->qp[q.abs_index + i] = (uint8_t)dq.ref_qp;
						g.sync();
					}
				}
				if (dq.enabled && q.depth <= 0) dq.write_qp = 1;
				encode_coding_unit(g, ee, v, sc, dq, curr);
			}
			while (depth_state.get(curr_depth) == 4) {
				depth_state.set(curr_depth, 0);
				curr_depth--;
				curr = v.geo[curr].parent;
			}
			if (v.geo[curr].parent >= 0) curr = v.geo[v.geo[curr].parent].child[depth_state.get(curr_depth)];
		}
	}
}

struct EntropyState {
	int last_idr = 0;
};

// ---- parameter sets, slice header, access unit ------------------------------------------------------------------------------------
inline void put_nal_header(std::vector<uint8_t> &out, int type)
{
	out.push_back((uint8_t)(type << 1));
	out.push_back(1);
}
inline void put_profile_tier_level(BitWriter &bs, int profile)
{
	bs.write(0, 2); bs.write(0, 1); bs.write(profile, 5);
	for (int j = 0; j < 32; j++) bs.write(j == profile || (profile == 1 && j == 2) ? 1 : 0, 1);
	bs.write(0, 1); bs.write(0, 1); bs.write(0, 1); bs.write(0, 1);
	bs.write(0, 16); bs.write(0, 16); bs.write(0, 12);
	bs.write(0, 8);   // level_idc: the reference leaves it 0
}
// hmr_put_vps_header :99, hmr_put_seq_header :204, hmr_put_pic_header :312 for one sub-layer
inline void write_parameter_sets(const Seq &S, int profile, std::vector<uint8_t> &vps, std::vector<uint8_t> &sps, std::vector<uint8_t> &pps)
{
	HostBits bs;
	bs.write(0, 4); bs.write(3, 2); bs.write(0, 6); bs.write(0, 3); bs.write(1, 1); bs.write(0xffff, 16);
	put_profile_tier_level(bs, profile);
	bs.write(1, 1);
	bs.uvlc(S.num_ref_frames + 1 - 1); bs.uvlc(0); bs.uvlc(0);
	bs.write(0, 6); bs.uvlc(0); bs.write(0, 1); bs.write(0, 1);
	bs.trailing_bits();
	put_nal_header(vps, 32);
	nalu_ebsp(bs, vps);

	bs = HostBits();
	bs.write(0, 4); bs.write(0, 3); bs.write(1, 1);
	put_profile_tier_level(bs, profile);
	bs.uvlc(0); bs.uvlc(1);
	bs.uvlc(S.width); bs.uvlc(S.height);
	bs.write(1, 1); bs.uvlc(0); bs.uvlc(0); bs.uvlc(0); bs.uvlc(0);   // conformance window: always flagged, offsets 0 (sizes are multiples of the minimum CU)
	bs.uvlc(0); bs.uvlc(0);
	bs.uvlc(0);       // log2_max_pic_order_cnt_lsb_minus4
	bs.write(1, 1);
	bs.uvlc(S.num_ref_frames + 1 - 1); bs.uvlc(0); bs.uvlc(0);
	const int min_cu_shift = 6 - (S.max_cu_depth - S.mincu_mintr_shift_diff);
	bs.uvlc(min_cu_shift - 3);
	bs.uvlc(S.max_cu_depth - S.mincu_mintr_shift_diff);
	bs.uvlc(S.min_tu_size_shift - 2);
	bs.uvlc(S.max_tu_size_shift - S.min_tu_size_shift);
	bs.uvlc(S.max_inter_tr_depth - 1);
	bs.uvlc(S.max_intra_tr_depth - 1);
	bs.write(1, 1); bs.write(0, 1);       // scaling_list_enabled_flag, no list data: the default lists
	bs.write(0, 1);                       // amp
	bs.write(S.sao, 1);
	bs.write(0, 1);                       // pcm
	const int num_rps = S.gop_size + S.num_ref_frames;
	bs.uvlc(num_rps);
	for (int i = 0; i < num_rps; i++) {
		if (i > 0) bs.write(0, 1);
		const int neg = i < num_rps - 1 ? (i == 0 ? S.num_ref_frames : i) : 0;
		bs.uvlc(neg); bs.uvlc(0);
		int prev = 0;
		for (int j = 0; j < neg; j++) { bs.uvlc(prev - (-(j + 1)) - 1); prev = -(j + 1); bs.write(1, 1); }
	}
	bs.write(0, 1);                       // long-term reference pictures
	bs.write(0, 1);                       // temporal mvp
	bs.write(1, 1);                       // strong intra smoothing
	bs.write(0, 1);                       // vui
	bs.write(0, 1);                       // extension
	bs.trailing_bits();
	put_nal_header(sps, 33);
	nalu_ebsp(bs, sps);

	bs = HostBits();
	bs.uvlc(0); bs.uvlc(0);
	bs.write(0, 1); bs.write(0, 1); bs.write(0, 3);
	bs.write(S.sign_hiding, 1);
	bs.write(0, 1);
	bs.uvlc(S.num_ref_frames - 1); bs.uvlc(S.num_ref_frames - 1);
	bs.svlc(S.qp - 26);
	bs.write(0, 1); bs.write(0, 1);
	bs.write(S.bitrate_mode == 0 ? 0 : 1, 1);
	if (S.bitrate_mode != 0) bs.uvlc(0);
	bs.svlc(S.chroma_qp_offset); bs.svlc(S.chroma_qp_offset);
	bs.write(0, 1); bs.write(0, 1); bs.write(0, 1); bs.write(0, 1); bs.write(0, 1);
	bs.write(S.wpp, 1);
	bs.write(1, 1);   // loop filter across slices
	bs.write(0, 1);   // deblocking control
	bs.write(0, 1);   // scaling list data
	bs.write(0, 1);   // lists modification
	bs.uvlc(0);       // parallel merge level
	bs.write(0, 1); bs.write(0, 1);
	bs.trailing_bits();
	put_nal_header(pps, 34);
	nalu_ebsp(bs, pps);
}

// count_needed_start_codes, hmr_headers.c:573
inline uint32_t count_escapes(const uint8_t *data, int size)
{
	uint32_t cnt = 0;
	std::vector<uint8_t> p(data, data + size);
	p.resize(size + 8, 0);
	int i = 0;
	while (i < size) {
		while (i < size) {
			if (p[i] == 0 && p[i + 1] == 0) {
				i++;
				if (i == size) break;
				if (p[++i] <= 3) break;
			} else i++;
		}
		if (i < size) cnt++;
	}
	return cnt;
}

// The access unit of a frame from its CABAC sub-streams (one per CTU row with WPP, else one): parameter sets in front of an IDR picture, slice header with the
// entry points, the sub-streams, NAL escaping, Annex-B start codes (encoder_engine_thread :3287-3330, HOMER_enc_write_annex_b_output :2196) - appended to `out`.
// init_qp: the QP the picture parameter set was written with (pic_init_qp_minus26 + 26: the configured QP, hmr_encoder_lib.c:1611).
inline void assemble_access_unit(EntropyState &es, const Seq &S, const FrameCtx &f, int profile, const uint8_t *const *row_data, const int *row_bytes, std::vector<uint8_t> &out)
{
	const int H = S.hctu, nrows = S.wpp ? H : 1;
	const bool idr = f.slice_type == SLICE_I;
	(void)es;
	std::vector<std::vector<uint8_t>> nals;
	if (idr) {
		std::vector<uint8_t> vps, sps, pps;
		write_parameter_sets(S, profile, vps, sps, pps);
		nals.push_back(vps); nals.push_back(sps); nals.push_back(pps);
	}
	HostBits sh;
	{
		// hmr_put_slice_header :375
		sh.write(1, 1);                        // first_slice_in_pic_flag
		if (idr) sh.write(0, 1);               // no_output_of_prior_pics_flag
		sh.uvlc(0);                            // pps id
		sh.uvlc(f.slice_type);
		if (!idr) {
			sh.write((f.poc - f.last_idr + 16) % 16, 4);
			sh.write(1, 1);                    // short_term_ref_pic_set_sps_flag
			int num_bits = 0;
			while ((1 << num_bits) < S.gop_size + S.num_ref_frames) num_bits++;
			if (num_bits) sh.write(0, num_bits);
		}
		if (S.sao) { sh.write(1, 1); sh.write(1, 1); }
		if (f.slice_type != SLICE_I) {
			sh.write(0, 1);                    // num_ref_idx_active_override_flag
			sh.uvlc(5 - S.num_merge_cand);
		}
		sh.svlc(f.qp - S.qp);
		sh.write(1, 1);                        // slice_loop_filter_across_slices_enabled_flag
		if (S.wpp) {
			// hmr_slice_header_code_wfpp_entry_points :617
			const int num = H - 1;
			uint32_t max_offset = 0, len_m1 = 1;
			std::vector<uint32_t> ep(num > 0 ? num : 0);
			for (int i = 0; i < num; i++) {
				ep[i] = row_bytes[i] + count_escapes(row_data[i], row_bytes[i]);
				if (ep[i] > max_offset) max_offset = ep[i];
			}
			while (max_offset >= (1u << (len_m1 + 1))) len_m1++;
			sh.uvlc(num);
			if (num > 0) sh.uvlc(len_m1);
			for (int i = 0; i < num; i++) sh.write(ep[i] - 1, len_m1 + 1);
		}
		sh.trailing_bits();
	}
	{
		size_t total = sh.bytecnt;
		for (int r = 0; r < nrows; r++) total += row_bytes[r];
		std::vector<uint8_t> all(total);
		memcpy(all.data(), sh.buf, sh.bytecnt);
		size_t o = sh.bytecnt;
		for (int r = 0; r < nrows; r++) {
			memcpy(all.data() + o, row_data[r], row_bytes[r]);
			o += row_bytes[r];
		}
		std::vector<uint8_t> nal;
		put_nal_header(nal, idr ? 19 : 1);
		nalu_ebsp(all.data(), (int)all.size(), nal);
		nals.push_back(nal);
	}
	// HOMER_enc_write_annex_b_output :2196
	for (size_t k = 0; k < nals.size(); k++) {
		const int type = (nals[k][0] >> 1) & 63;
		if (k == 0 || type == 33 || type == 34) out.push_back(0);
		out.push_back(0); out.push_back(0); out.push_back(1);
		out.insert(out.end(), nals[k].begin(), nals[k].end());
	}
}

}  // namespace henc
