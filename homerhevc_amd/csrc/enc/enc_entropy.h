// Entropy coding and bitstream of a frame: the serial consumer of the CTU decisions.
// Restates the reference's L1: the bit writer and NAL escaping (hmr_bitstream.c:63-145), the CABAC engine
// and its counting twin (hmr_binary_encoding.c:57-358, 553-576), context set-up (hmr_arithmetic_encoding.c:128-204), the CTU syntax
// (:358-1367, 1561-1832, 2039-2134), the SAO syntax (:1839-2036) and its rate terms (:2377-2429), the SAO mode decision
// (hmr_sao.c:363-955, 1295-1441), the parameter sets and slice header (hmr_headers.c) and the access-unit assembly of
// encoder_engine_thread (hmr_encoder_lib.c:3287-3330) + HOMER_enc_write_annex_b_output (:2196).
// Host code: one pass over the CTUs in raster order, the order the reference's single WPP thread codes them in.
#pragma once
#include <math.h>
#include <string.h>
#include <vector>
#include "enc_cabac_tables.h"
#include "enc_prims.h"
#include "enc_sao.h"

namespace henc {

// ---- bit writer (MSB first; hmr_bitstream_write_bits ORs into a zeroed buffer) ---------------------------------------------
struct BitWriter {
	std::vector<uint8_t> buf;
	int bytecnt = 0, bitcnt = 0;
	void init() { buf.assign(buf.size(), 0); bytecnt = bitcnt = 0; }
	void need(size_t n) { if (buf.size() < n) buf.resize(n * 2 + 64, 0); }
	void write(uint32_t val, int n)
	{
		if (n <= 0) return;
		need((size_t)bytecnt + 16);
		uint64_t v = n >= 32 ? val : (val & ((1u << n) - 1));
		v <<= (64 - n - bitcnt);
		for (int k = 0; k < 8 && (k * 8 < bitcnt + n); k++) buf[bytecnt + k] |= (uint8_t)(v >> (56 - 8 * k));
		bitcnt += n;
		bytecnt += bitcnt >> 3;
		bitcnt &= 7;
	}
	void uvlc(uint32_t val)
	{
		uint32_t length = 1, temp = ++val;
		while (temp != 1) { temp >>= 1; length += 2; }
		write(0, length >> 1);
		write(val, (length + 1) >> 1);
	}
	void svlc(int val) { uvlc(val <= 0 ? (uint32_t)(-val) << 1 : ((uint32_t)val << 1) - 1); }
	void align0() { if (bitcnt) write(0, 8 - bitcnt); }
	void trailing_bits() { write(1, 1); align0(); }
	int bitcount() const { return (bytecnt << 3) + bitcnt; }
};

// hmr_bitstream_nalu_ebsp :123 - the reference's escaping loop taken literally (it looks two bytes past the end, which are zero)
inline void nalu_ebsp(const BitWriter &in, std::vector<uint8_t> &out)
{
	const int size = in.bytecnt;
	std::vector<uint8_t> p(in.buf.begin(), in.buf.begin() + size);
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

// ---- CABAC --------------------------------------------------------------------------------------------------------------
struct Cabac {
	uint32_t low = 0, range = 510, buffered_byte = 0xff;
	int num_buffered = 0, bits_left = 23;
	uint64_t frac_bits = 0;
	uint8_t ctx[CTX_TOTAL];
	bool counter = false;
	BitWriter *bs = nullptr;

	static int next_mps(int s) { return s < 124 ? s + 2 : (s < 126 ? s : s); }
	static int next_lps(int s) { return kNextStateLps[s]; }
	static int mps_next(int s) { return s >= 124 && s < 126 ? s : (s >= 126 ? s : s + 2); }

	void start() { low = 0; bits_left = 23; num_buffered = 0; buffered_byte = 0xff; range = 510; }
	void reset_bits() { low = 0; bits_left = 23; num_buffered = 0; buffered_byte = 0xff; frac_bits &= 32767; }
	void init_contexts(int slice_type, int qp)
	{
		for (int i = 0; i < CTX_TOTAL; i++) {
			const int init_value = kCtxInit[slice_type][i];
			const int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
			int init_state = ((slope * qp) >> 4) + offset;
			init_state = init_state < 1 ? 1 : (init_state > 126 ? 126 : init_state);
			const int mp = init_state >= 64;
			ctx[i] = (uint8_t)(((mp ? init_state - 64 : 63 - init_state) << 1) + mp);
		}
	}
	void write_out()
	{
		const uint32_t lead = low >> (24 - bits_left);
		bits_left += 8;
		low &= 0xffffffffu >> bits_left;
		if (lead == 0xff) num_buffered++;
		else if (num_buffered > 0) {
			const uint32_t carry = lead >> 8;
			uint32_t byte = buffered_byte + carry;
			buffered_byte = lead & 0xff;
			bs->write(byte, 8);
			byte = (0xff + carry) & 0xff;
			while (num_buffered > 1) { bs->write(byte, 8); num_buffered--; }
		} else {
			num_buffered = 1;
			buffered_byte = lead;
		}
	}
	void encode_bin(int ci, uint32_t bin)
	{
		uint8_t &st = ctx[ci];
		if (counter) {
			frac_bits += (uint64_t)kEntropyBits[st ^ bin];
			// the counting coder's transition table (g_bc_next_state, hmr_binary_encoding.c:305) is filled by bc_init_next_state_table(),
			// which nothing calls: it stays all zero, so every context the counter touches falls to state 0
			st = 0;
			return;
		}
		const uint32_t lps = kRangeLps[st >> 1][(range >> 6) & 3];
		range -= lps;
		if (bin != (uint32_t)(st & 1)) {
			const int nb = kRenorm[lps >> 3];
			low = (low + range) << nb;
			range = lps << nb;
			st = (uint8_t)next_lps(st);
			bits_left -= nb;
		} else {
			st = (uint8_t)mps_tab(st);
			if (range >= 256) return;
			low <<= 1;
			range <<= 1;
			bits_left--;
		}
		if (bits_left < 12) write_out();
	}
	static int mps_tab(int s) { return s < 124 ? s + 2 : s; }   // g_bc_next_state_MPS: saturates at 124 / 125; 126 / 127 stay
	void encode_ep(uint32_t bin)
	{
		if (counter) { frac_bits += 32768; return; }
		low <<= 1;
		if (bin) low += range;
		bits_left--;
		if (bits_left < 12) write_out();
	}
	void encode_bins_ep(uint32_t bins, int n)
	{
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
	void encode_trm(uint32_t bin)
	{
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
	void finish()
	{
		if (low >> (32 - bits_left)) {
			bs->write(buffered_byte + 1, 8);
			while (num_buffered > 1) { bs->write(0x00, 8); num_buffered--; }
			low -= 1u << (32 - bits_left);
		} else {
			if (num_buffered > 0) bs->write(buffered_byte, 8);
			while (num_buffered > 1) { bs->write(0xff, 8); num_buffered--; }
		}
		bs->write(low >> 8, 24 - bits_left);
	}
	uint32_t bitcnt() const { return (uint32_t)(frac_bits >> 15); }
	// bm_copy_binary_model + ee_copy_entropy_model
	void load(const Cabac &src)
	{
		low = src.low; range = src.range; bits_left = src.bits_left; buffered_byte = src.buffered_byte; num_buffered = src.num_buffered; frac_bits = src.frac_bits;
		if (&src != this) memcpy(ctx, src.ctx, sizeof ctx);
	}
};

// ---- frame view the entropy stage works on ---------------------------------------------------------------------------------
struct EntropyFrame {
	const Seq *seq;
	const FrameCtx *f;
	const DevTables *T;
	const Geo *geo;
	const uint8_t *ctu_base;      // CtuPublic records, `ctu_pitch` bytes apart
	size_t ctu_pitch;
	const int16_t *coeff;         // [nctu][6144]
	const CtuPublic &ctu(int n) const { return *(const CtuPublic *)(ctu_base + (size_t)n * ctu_pitch); }
	CtuPublic &ctu_rw(int n) const { return *(CtuPublic *)(ctu_base + (size_t)n * ctu_pitch); }
};

struct CuView {
	const EntropyFrame *fr;
	int n;                        // CTU index
	const CtuPublic *c;
	const CtuPublic *left() const { return c->has_left ? &fr->ctu(n - 1) : nullptr; }
	const CtuPublic *top() const { return c->has_top ? &fr->ctu(n - fr->seq->wctu) : nullptr; }
};
inline const CtuPublic *ent_pu_left(const CuView &v, int ni, uint32_t *idx)
{
	const Geo &q = v.fr->geo[ni];
	*idx = q.abs_left;
	return (q.raster_index & 15) == 0 ? v.left() : v.c;
}
inline const CtuPublic *ent_pu_top(const CuView &v, int ni, uint32_t *idx, int planar)
{
	const Geo &q = v.fr->geo[ni];
	*idx = q.abs_top;
	if (q.raster_index < 16) return planar ? nullptr : v.top();
	return v.c;
}
inline bool node_inside(const CuView &v, int ni)
{
	const Geo &q = v.fr->geo[ni];
	return v.c->y + q.y + q.size <= v.fr->seq->height && v.c->x + q.x + q.size <= v.fr->seq->width;
}
#define HENC_CBF(c, idx, comp, trd) ((((c)->cbf[comp][idx]) >> (trd)) & 1)

// get_intra_dir_luma_predictor :545 on the final arrays
inline void ent_intra_preds(const CuView &v, int ni, int *p)
{
	uint32_t idx = 0;
	const CtuPublic *l = ent_pu_left(v, ni, &idx);
	const int ld = l ? (l->pred_mode[idx] == PM_INTRA ? l->intra_mode[0][idx] : DC_IDX) : DC_IDX;
	const CtuPublic *t = ent_pu_top(v, ni, &idx, 1);
	const int td = t ? (t->pred_mode[idx] == PM_INTRA ? t->intra_mode[0][idx] : DC_IDX) : DC_IDX;
	if (ld == td) {
		if (ld > 1) { p[0] = ld; p[1] = ((ld + 29) % 32) + 2; p[2] = ((ld - 1) % 32) + 2; }
		else { p[0] = PLANAR_IDX; p[1] = DC_IDX; p[2] = VER_IDX; }
	} else {
		p[0] = ld; p[1] = td;
		if (ld && td) p[2] = PLANAR_IDX;
		else p[2] = (ld + td) < 2 ? VER_IDX : DC_IDX;
	}
}

// ---- residual coding (encode_residual :1087, encode_last_significant_XY :954, get_sig_ctx_inc :1027) ---------------------------
inline int sig_ctx_inc(int pattern, int scan_mode, int px, int py, int shift, int comp)
{
	static const int map4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
	if (px + py == 0) return 0;
	if (shift == 2) return map4[4 * py + px];
	const int offset = shift == 3 ? (scan_mode == SCAN_DIAG ? 9 : 15) : (comp == COMP_Y ? 21 : 12);
	const int xs = px & 3, ys = py & 3;
	int cnt;
	if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
	else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
	else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
	else cnt = 2;
	return ((comp == COMP_Y && ((px >> 2) + (py >> 2)) > 0) ? 3 : 0) + offset + cnt;
}

inline void encode_last_xy(Cabac &ee, int x, int y, int shift, int comp, int scan_mode)
{
	static const int group_idx[32] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
	static const int min_in_group[10] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24};
	const int size = 1 << shift;
	const int cx = CTX_LAST_X + (comp ? 15 : 0), cy = CTX_LAST_Y + (comp ? 15 : 0);
	if (scan_mode == SCAN_VER) { const int t = x; x = y; y = t; }
	const int gx = group_idx[x], gy = group_idx[y];
	const int off = comp ? 0 : ((shift - 2) * 3 + (((shift - 2) + 1) >> 2));
	const int sh = comp ? (shift - 2) : (((shift - 2) + 3) >> 2);
	int k;
	for (k = 0; k < gx; k++) ee.encode_bin(cx + off + (k >> sh), 1);
	if (gx < group_idx[size - 1]) ee.encode_bin(cx + off + (k >> sh), 0);
	for (k = 0; k < gy; k++) ee.encode_bin(cy + off + (k >> sh), 1);
	if (gy < group_idx[size - 1]) ee.encode_bin(cy + off + (k >> sh), 0);
	if (gx > 3) {
		const int count = (gx - 2) >> 1;
		x -= min_in_group[gx];
		for (int i = count - 1; i >= 0; i--) ee.encode_ep((x >> i) & 1);
	}
	if (gy > 3) {
		const int count = (gy - 2) >> 1;
		for (int i = count - 1; i >= 0; i--) ee.encode_ep((y >> i) & 1);   // the reference leaves y unreduced (:1010); the low bits are the same
	}
}

inline void encode_residual(Cabac &ee, const CuView &v, int ni, int comp)
{
	const EntropyFrame &fr = *v.fr;
	const Seq &S = *fr.seq;
	const int is_luma = comp == COMP_Y;
	const int abs_index = fr.geo[ni].abs_index;
	const int pi = is_luma ? ni : (fr.geo[ni].size_chroma > 2 ? ni : fr.geo[ni].parent);
	const Geo &q = fr.geo[pi];
	const int size = is_luma ? q.size : q.size_chroma;
	const int shift = is_luma ? S.max_cu_size_shift - q.depth : S.max_cu_size_shift - 1 - q.depth;
	const int16_t *coeff = fr.coeff + (size_t)v.n * 6144 + (comp == 0 ? 0 : (comp == 1 ? 4096 : 5120)) + (q.abs_index << (4 - (is_luma ? 0 : 2)));
	const CtuPublic *c = v.c;
	const int num_part_in_pred_cu = NPART >> (c->pred_depth[abs_index] * 2);
	const int scan_mode = find_scan_mode(c->pred_mode[q.abs_index] == PM_INTRA, is_luma, size, c->intra_mode[is_luma ? 0 : 1][q.abs_index],
					     c->intra_mode[0][(abs_index / num_part_in_pred_cu) * num_part_in_pred_cu]);
	const uint32_t *scan = fr.T->scan[scan_mode][shift];
	uint32_t cg_tmp[64];
	const uint32_t *scan_cg;
	const int blk = size >> 2;
	if (shift == 3) {
		static const uint32_t s8[4][4] = {{0, 1, 2, 3}, {0, 1, 2, 3}, {0, 2, 1, 3}, {0, 2, 1, 3}};
		scan_cg = s8[scan_mode];
	} else if (shift == 5) {
		// g_sigLastScanCG32x32 (hmr_tables.c:71-95): the plain up-right diagonal scan of the 8 x 8 grid of coefficient groups
		int k = 0;
		for (int d = 0; d < 15; d++)
			for (int row = d < 8 ? d : 7, col = d - row; row >= 0 && col < 8; row--, col++) cg_tmp[k++] = (uint32_t)(row * 8 + col);
		scan_cg = cg_tmp;
	} else scan_cg = fr.T->scan[scan_mode][shift > 3 ? shift - 2 : 0];
	uint8_t cg_flag[64];
	memset(cg_flag, 0, sizeof cg_flag);
	int num_nz = 0, raster_pos_last = 0, scan_pos_last = 0, last_x = 0, last_y = 0;
	for (int i = 0; i < size * size; i++) {
		const int sp = scan[i];
		if (coeff[sp] != 0) {
			raster_pos_last = i;
			scan_pos_last = sp;
			num_nz++;
			last_y = sp >> shift;
			last_x = sp - (last_y << shift);
			cg_flag[blk * (last_y >> 2) + (last_x >> 2)] = 1;
		}
	}
	if (num_nz == 0) return;
	const int valid = S.sign_hiding;
	encode_last_xy(ee, last_x, last_y, shift, comp, scan_mode);
	const int last_scan_set = raster_pos_last >> 4;
	uint32_t c1 = 1, go_rice;
	int scan_pos_sig = raster_pos_last;
	const int base_cg = CTX_SIG_CG + (is_luma ? 0 : 2), base_sig = CTX_SIG + (is_luma ? 0 : 27);
	int abs_coeff[16];
	for (int subset = last_scan_set; subset >= 0; subset--) {
		int num_non_zero = 0;
		const int sub_pos = subset << 4;
		uint32_t coeff_signs = 0;
		int last_nz = -1, first_nz = 16;
		go_rice = 0;
		if (scan_pos_sig == raster_pos_last) {
			abs_coeff[0] = habs(coeff[scan_pos_last]);
			coeff_signs = coeff[scan_pos_last] < 0;
			num_non_zero = 1;
			last_nz = first_nz = scan_pos_sig;
			scan_pos_sig--;
		}
		const int cg_block_pos = scan_cg[subset];
		const int cg_y = cg_block_pos / blk, cg_x = cg_block_pos - cg_y * blk;
		if (subset == last_scan_set || subset == 0) cg_flag[cg_block_pos] = 1;
		else {
			const uint32_t sig_cg = cg_flag[cg_block_pos] != 0;
			int right = 0, lower = 0;
			if (cg_x < blk - 1) right = cg_flag[cg_y * blk + cg_x + 1] != 0;
			if (cg_y < blk - 1) lower = cg_flag[(cg_y + 1) * blk + cg_x] != 0;
			ee.encode_bin(base_cg + (right || lower), sig_cg);
		}
		if (cg_flag[cg_block_pos]) {
			uint32_t right = 0, lower = 0;
			if (cg_x < blk - 1) right = cg_flag[cg_y * blk + cg_x + 1] != 0;
			if (cg_y < blk - 1) lower = cg_flag[(cg_y + 1) * blk + cg_x] != 0;
			const int pattern = right + (lower << 1);
			for (; scan_pos_sig >= sub_pos; scan_pos_sig--) {
				const uint32_t bp = scan[scan_pos_sig];
				const uint32_t py = bp >> shift, px = bp - (py << shift);
				const uint32_t sig = coeff[bp] != 0;
				if (scan_pos_sig > sub_pos || subset == 0 || num_non_zero) ee.encode_bin(base_sig + sig_ctx_inc(pattern, scan_mode, px, py, shift, comp), sig);
				if (sig) {
					abs_coeff[num_non_zero] = habs(coeff[bp]);
					coeff_signs = 2 * coeff_signs + (coeff[bp] < 0);
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
				const uint32_t sym = abs_coeff[idx] > 1;
				ee.encode_bin(base + c1, sym);
				if (sym) {
					c1 = 0;
					if (first_c2 == -1) first_c2 = idx;
				} else if (c1 < 3 && c1 > 0) c1++;
			}
			if (c1 == 0) {
				base = CTX_ABS + ctx_set + (is_luma ? 0 : 4);
				if (first_c2 != -1) ee.encode_bin(base, abs_coeff[first_c2] > 2);
			}
			if (valid && sign_hidden) ee.encode_bins_ep(coeff_signs >> 1, num_non_zero - 1);
			else ee.encode_bins_ep(coeff_signs, num_non_zero);
			if (c1 == 0 || num_non_zero > 8) {
				int first_coeff2 = 1;
				for (int idx = 0; idx < num_non_zero; idx++) {
					const int base_level = idx < 8 ? (2 + first_coeff2) : 1;
					if (abs_coeff[idx] >= base_level) {
						int code = abs_coeff[idx] - base_level;
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
						if (abs_coeff[idx] > 3 * (1 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
					}
					if (abs_coeff[idx] >= 2) first_coeff2 = 0;
				}
			}
		}
	}
}

// ---- CU syntax -----------------------------------------------------------------------------------------------------------------
inline void encode_qt_cbf(Cabac &ee, int comp, int tr_depth, int cbf)
{
	const int ctx = comp ? tr_depth : (tr_depth == 0 ? 1 : 0);
	ee.encode_bin(CTX_QT_CBF + (comp ? 4 : 0) + ctx, cbf);
}

// transform_tree :1561 (fixed QP: no delta-QP syntax)
inline void encode_transform_tree(Cabac &ee, const CuView &v, int top_ni)
{
	const EntropyFrame &fr = *v.fr;
	const Seq &S = *fr.seq;
	const CtuPublic *c = v.c;
	const int depth = fr.geo[top_ni].depth;
	int abs_index = fr.geo[top_ni].abs_index;
	int is_intra = c->pred_mode[abs_index] == PM_INTRA;
	if (!is_intra) {
		const uint32_t qtroot = HENC_CBF(c, abs_index, 0, 0) || HENC_CBF(c, abs_index, 1, 0) || HENC_CBF(c, abs_index, 2, 0);
		if (!(c->merge[abs_index] && c->part_size_type[abs_index] == PART_2Nx2N)) ee.encode_bin(CTX_QT_ROOT_CBF, qtroot);
		if (!qtroot) return;
	}
	int depth_state[NDEPTH] = {0, 0, 0, 0, 0};
	int curr = top_ni, parent = top_ni, curr_depth = depth;
	while (curr_depth != depth || depth_state[curr_depth] != 1) {
		const Geo &q = fr.geo[curr];
		curr_depth = q.depth;
		abs_index = q.abs_index;
		const int shift = S.max_cu_size_shift - curr_depth;
		const int pred_depth = c->pred_depth[abs_index], tr_depth = curr_depth - pred_depth, first = tr_depth == 0;
		const int tr_idx = c->tr_idx[abs_index];
		is_intra = c->pred_mode[abs_index] == PM_INTRA;
		const int part = c->part_size_type[abs_index];
		const int split_flag = (tr_idx + pred_depth) > curr_depth;
		const int log2_tr = S.max_cu_size_shift - curr_depth, log2_cu = S.max_cu_size_shift - pred_depth;
		const int intra_split = is_intra && part == PART_NxN, inter_split = !is_intra && S.max_inter_tr_depth == 1 && part != PART_2Nx2N;
		const int max_tr = is_intra ? S.max_intra_tr_depth : S.max_inter_tr_depth;
		int tu_min_in_cu;
		if (log2_cu < S.min_tu_size_shift + max_tr - 1 + inter_split + intra_split) tu_min_in_cu = S.min_tu_size_shift;
		else {
			tu_min_in_cu = log2_cu - (max_tr - 1 + inter_split + intra_split);
			if (tu_min_in_cu > S.max_tu_size_shift) tu_min_in_cu = S.max_tu_size_shift;
		}
		if (!(is_intra && part == PART_NxN && curr_depth == pred_depth) && !(!is_intra && part != PART_2Nx2N && curr_depth == pred_depth && S.max_inter_tr_depth == 1) &&
		    !(log2_tr > S.max_tu_size_shift) && !(log2_tr == S.min_tu_size_shift) && !(log2_tr == tu_min_in_cu))
			ee.encode_bin(CTX_TRANS_SUBDIV + 5 - shift, split_flag);
		if (first || shift > 2) {
			if (first || HENC_CBF(c, abs_index, 1, tr_depth - 1)) encode_qt_cbf(ee, 1, tr_depth, HENC_CBF(c, abs_index, 1, tr_depth));
			if (first || HENC_CBF(c, abs_index, 2, tr_depth - 1)) encode_qt_cbf(ee, 2, tr_depth, HENC_CBF(c, abs_index, 2, tr_depth));
		}
		depth_state[curr_depth]++;
		if (split_flag) {
			parent = curr;
			curr_depth++;
		} else {
			const uint32_t cbf_y = HENC_CBF(c, abs_index, 0, tr_depth), cbf_u = HENC_CBF(c, abs_index, 1, tr_depth), cbf_v = HENC_CBF(c, abs_index, 2, tr_depth);
			if (c->pred_mode[abs_index] == PM_INTRA || tr_depth != 0 || cbf_u || cbf_v) encode_qt_cbf(ee, 0, tr_depth, cbf_y);
			if (cbf_y) encode_residual(ee, v, curr, COMP_Y);
			if (shift > 2) {
				if (cbf_u) encode_residual(ee, v, curr, COMP_U);
				if (cbf_v) encode_residual(ee, v, curr, COMP_V);
			} else if (q.list_index == fr.geo[fr.geo[q.parent].child[0]].list_index + 3) {
				if (cbf_u) encode_residual(ee, v, curr, COMP_U);
				if (cbf_v) encode_residual(ee, v, curr, COMP_V);
			}
			while (depth_state[curr_depth] == 4) {
				depth_state[curr_depth] = 0;
				parent = fr.geo[parent].parent;
				curr_depth--;
			}
			if (curr_depth == 0 && depth_state[curr_depth] == 1) break;
		}
		if (curr_depth == depth && depth_state[curr_depth] == 1) break;
		curr = fr.geo[parent].child[depth_state[curr_depth]];
	}
}

inline void encode_mvd(Cabac &ee, const CtuPublic *c, int idx)
{
	const int h = c->mv_diff[idx].x, ver = c->mv_diff[idx].y;
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
inline void encode_end_of_cu(Cabac &ee, const CuView &v, int ni)
{
	const EntropyFrame &fr = *v.fr;
	const Seq &S = *fr.seq;
	const Geo &q = fr.geo[ni];
	const uint32_t cu_addr = (uint32_t)v.n * NPART + q.abs_index;
	const int width = S.width, height = S.height;
	uint32_t real_end;
	if (width % 64 || height % 64) {
		int wr = (width % 64) >> 2, hr = (height % 64) >> 2;
		if (hr == 0) hr = 15;
		else if (wr) hr -= 1;
		const int aux = hr * 16 + wr;
		real_end = (uint32_t)S.nctu * NPART - NPART + raster2abs(aux - 1) + 1;
	} else real_end = (uint32_t)S.nctu * NPART;
	const int px = v.c->x + q.x, py = v.c->y + q.y;
	const int boundary = ((px + q.size) % 64 == 0 || (px + q.size) == width) && ((py + q.size) % 64 == 0 || (py + q.size) == height);
	const int terminate = cu_addr + q.num_part == real_end;
	if (boundary && !terminate) ee.encode_trm(0);
}

// ee_encode_coding_unit :1787
inline void encode_coding_unit(Cabac &ee, const CuView &v, int ni)
{
	const EntropyFrame &fr = *v.fr;
	const Seq &S = *fr.seq;
	const CtuPublic *c = v.c;
	const Geo &q = fr.geo[ni];
	const int abs_index = q.abs_index, is_intra = c->pred_mode[abs_index] == PM_INTRA, part = c->part_size_type[abs_index];
	const int p_slice = fr.f->slice_type != SLICE_I;
	uint32_t idx = 0;
	if (p_slice) {
		const CtuPublic *l = ent_pu_left(v, ni, &idx);
		int ctx = l ? (l->skipped[idx] ? 1 : 0) : 0;
		const CtuPublic *t = ent_pu_top(v, ni, &idx, 0);
		ctx += t ? (t->skipped[idx] ? 1 : 0) : 0;
		ee.encode_bin(CTX_SKIP_FLAG + ctx, c->skipped[abs_index]);
	}
	auto merge_index = [&](int a) {
		// encode_merge_index :613 with two candidates: one context-coded bin
		if (S.num_merge_cand > 1) {
			const uint32_t unary = c->merge_idx[a];
			for (int ui = 0; ui < S.num_merge_cand - 1; ui++) {
				const uint32_t sym = ui == (int)unary ? 0 : 1;
				if (ui == 0) ee.encode_bin(CTX_MERGE_IDX, sym);
				else ee.encode_ep(sym);
				if (sym == 0) break;
			}
		}
	};
	if (c->skipped[abs_index]) {
		merge_index(abs_index);
		encode_end_of_cu(ee, v, ni);
		return;
	}
	if (p_slice) ee.encode_bin(CTX_PRED_MODE, c->pred_mode[abs_index]);
	// encode_part_size :436
	const int min_cu_depth = S.max_cu_depth - S.mincu_mintr_shift_diff;
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
			dir[j] = c->intra_mode[0][fr.geo[pn].abs_index];
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
		uint32_t chroma = c->intra_mode[1][abs_index];
		if (chroma == DM_CHROMA_IDX) ee.encode_bin(CTX_CHROMA_PRED, 0);
		else {
			int list[5];
			const int luma = c->intra_mode[0][abs_index];
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
		static const uint32_t pu_off[8] = {0, 8, 4, 4, 2, 10, 1, 5};
		const uint32_t pu_offset = (pu_off[part] << ((S.max_cu_depth - c->pred_depth[abs_index]) << 1)) >> 4;
		for (int p = 0, sub = abs_index; p < num_pu; p++, sub += pu_offset) {
			ee.encode_bin(CTX_MERGE_FLAG, c->merge[sub]);
			if (c->merge[sub]) merge_index(sub);
			else if (c->inter_mode[sub] & 1) {
				encode_mvd(ee, c, sub);
				ee.encode_bin(CTX_MVP_IDX, c->mv_diff_ref_idx[sub] ? 1 : 0);
			}
		}
	}
	encode_transform_tree(ee, v, ni);
	encode_end_of_cu(ee, v, ni);
}

// ee_encode_ctu :2039
inline void encode_ctu_syntax(Cabac &ee, const EntropyFrame &fr, int n)
{
	const Seq &S = *fr.seq;
	CuView v{&fr, n, &fr.ctu(n)};
	int depth_state[NDEPTH] = {0, 0, 0, 0, 0};
	int curr = 0, curr_depth = 0;
	const int min_cu_depth = S.max_cu_depth - S.mincu_mintr_shift_diff;
	while (curr_depth != 0 || depth_state[curr_depth] != 1) {
		const Geo &q = fr.geo[curr];
		const bool inside = node_inside(v, curr);
		if (inside && q.depth != min_cu_depth) {
			// encode_split_flag :391
			uint32_t idx = 0;
			const int split = v.c->pred_depth[q.abs_index] > q.depth;
			const CtuPublic *l = ent_pu_left(v, curr, &idx);
			int ctx = l ? (l->pred_depth[idx] > q.depth ? 1 : 0) : 0;
			const CtuPublic *t = ent_pu_top(v, curr, &idx, 0);
			ctx += t ? (t->pred_depth[idx] > q.depth ? 1 : 0) : 0;
			ee.encode_bin(CTX_SPLIT_FLAG + ctx, split);
		}
		const int pred_depth = v.c->pred_depth[q.abs_index];
		depth_state[curr_depth]++;
		if (curr_depth < pred_depth) {
			curr_depth++;
			curr = q.child[depth_state[curr_depth]];
		} else {
			if (inside) encode_coding_unit(ee, v, curr);
			while (depth_state[curr_depth] == 4) {
				depth_state[curr_depth] = 0;
				curr_depth--;
				curr = fr.geo[curr].parent;
			}
			if (fr.geo[curr].parent >= 0) curr = fr.geo[fr.geo[curr].parent].child[depth_state[curr_depth]];
		}
	}
}

// ---- SAO decision: the shared part is enc_sao.h; here the candidate derivation from the statistics as the reference does it on the CPU ------------
// est_iter_offset :445
inline int sao_iter_offset(int type_idx, double lambda, int offset_input, int64_t count, int64_t diff, int64_t *best_dist, double *best_cost)
{
	int iter = offset_input, out = 0;
	double min_cost = lambda;
	while (iter != 0) {
		int64_t rate = type_idx == SAO_BO ? habs(iter) + 2 : habs(iter) + 1;
		if (habs(iter) == 7) rate--;
		const int64_t dist = est_sao_dist(count, iter, diff);
		const double cost = (double)dist + lambda * (double)rate;
		if (cost < min_cost) {
			min_cost = cost;
			out = iter;
			*best_dist = dist;
			*best_cost = cost;
		}
		iter = iter > 0 ? iter - 1 : iter + 1;
	}
	return out;
}

// sao_derive_offsets :480
inline void sao_derive_offsets(const double *lambdas, int comp, int type, const int32_t (*st)[32], int *q, int *aux)
{
	memset(q, 0, sizeof(int) * 32);
	const int num = type == SAO_BO ? 32 : 5;
	for (int k = 0; k < num; k++) {
		if (type != SAO_BO && k == 2) continue;
		if (st[1][k] == 0) continue;
		const double x = (double)(int64_t)st[0][k] / (double)(int64_t)st[1][k];
		q[k] = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
		q[k] = hclip(q[k], -7, 7);
	}
	if (type != SAO_BO) {
		int64_t d;
		double cst;
		for (int k = 0; k < 5; k++) {
			if (k == 0 && q[k] < 0) q[k] = 0;
			if (k == 1 && q[k] < 0) q[k] = 0;
			if (k == 3 && q[k] > 0) q[k] = 0;
			if (k == 4 && q[k] > 0) q[k] = 0;
			if (q[k] != 0) q[k] = sao_iter_offset(type, lambdas[comp], q[k], st[1][k], st[0][k], &d, &cst);
		}
		*aux = 0;
	} else {
		int64_t dist[32];
		double cost[32];
		memset(dist, 0, sizeof dist);
		for (int k = 0; k < 32; k++) {
			cost[k] = lambdas[comp];
			if (q[k] != 0) q[k] = sao_iter_offset(type, lambdas[comp], q[k], st[1][k], st[0][k], &dist[k], &cost[k]);
		}
		double min_cost = MAX_COST;
		for (int band = 0; band < 32 - 4 + 1; band++) {
			double cst = cost[band];
			cst += cost[band + 1];
			cst += cost[band + 2];
			cst += cost[band + 3];
			if (cst < min_cost) { min_cost = cst; *aux = band; }
		}
		int clear[32];
		memset(clear, 0, sizeof clear);
		for (int i = 0; i < 4; i++) { const int band = (*aux + i) % 32; clear[band] = q[band]; }
		memcpy(q, clear, sizeof clear);
	}
}
// sao_invert_quant_offsets :592 (8 bit: step 1) - also clears what the type does not use
inline void sao_invert_quant(int type, int aux, int *dst, const int *src)
{
	int coded[32];
	memcpy(coded, src, sizeof coded);
	memset(dst, 0, sizeof(int) * 32);
	if (type == SAO_BO)
		for (int i = 0; i < 4; i++) dst[(aux + i) % 32] = coded[(aux + i) % 32];
	else
		for (int i = 0; i < 5; i++) dst[i] = coded[i];
}
// the candidates of SAO_MODE_NEW from the statistics (on the device k_sao_offsets has them ready: k_saooffsets.hip)
struct SaoCandFromStats {
	const double *lambdas;
	const SaoStats *st;
	int64_t get(int comp, int type, SaoOffset &t) const
	{
		int inv[32];
		sao_derive_offsets(lambdas, comp, type, (*st)[comp][type], t.offset, &t.type_aux);
		sao_invert_quant(type, t.type_aux, inv, t.offset);
		return sao_distortion(type, t.type_aux, inv, (*st)[comp][type]);
	}
};
// sao_decide_blk_params :1295 for CTU n with the real coder `ee` standing before the CTU's SAO syntax
inline void sao_decide_ctu(const Cabac &ee, const EntropyFrame &fr, int n, const SaoStats &stats, const double *lambdas)
{
	const Seq &S = *fr.seq;
	CtuPublic &c = fr.ctu_rw(n);
	const int cx = n % S.wctu, cy = n / S.wctu;
	const SaoTables T = {kEntropyBits, kNextStateLps};
	const SaoCandFromStats cand = {lambdas, &stats};
	sao_decide(T, ee.ctx[CTX_SAO_MERGE], ee.ctx[CTX_SAO_TYPE], cand, stats, cx > 0 ? fr.ctu(n - 1).sao_recon : nullptr, cy > 0 ? fr.ctu(n - S.wctu).sao_recon : nullptr, lambdas,
		   c.sao_coded, c.sao_recon);
}

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
	BitWriter bs;
	bs.write(0, 4); bs.write(3, 2); bs.write(0, 6); bs.write(0, 3); bs.write(1, 1); bs.write(0xffff, 16);
	put_profile_tier_level(bs, profile);
	bs.write(1, 1);
	bs.uvlc(S.num_ref_frames + 1 - 1); bs.uvlc(0); bs.uvlc(0);
	bs.write(0, 6); bs.uvlc(0); bs.write(0, 1); bs.write(0, 1);
	bs.trailing_bits();
	put_nal_header(vps, 32);
	nalu_ebsp(bs, vps);

	bs = BitWriter();
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

	bs = BitWriter();
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
inline uint32_t count_escapes(const BitWriter &b)
{
	uint32_t cnt = 0;
	const int size = b.bytecnt;
	std::vector<uint8_t> p(b.buf.begin(), b.buf.begin() + size);
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

struct EntropyState {
	std::vector<BitWriter> rows;      // one sub-stream per CTU row (aux_bs)
	Cabac ee, saved;                  // the coding environment and the copy the next row starts from (ee_list pair)
	int last_idr = 0;
	bool sets_written = false;
};

// The entropy stage of one frame: SAO decision + CTU syntax per CTU in raster order, then the access unit in Annex-B form appended to `out`.
// stats: SAO statistics of every CTU: the SAO parameters are decided here and land in the CTU records (sao_coded / sao_recon); nullptr when SAO is off or when
// the records already carry the parameters (the device path: k_sao_decide).
inline void encode_frame_entropy(EntropyState &es, const EntropyFrame &fr, const SaoStats *stats, int profile, std::vector<uint8_t> &out)
{
	const Seq &S = *fr.seq;
	const FrameCtx &f = *fr.f;
	const int W = S.wctu, H = S.hctu;
	es.rows.resize(H);
	double sao_lambda[3];
	sao_lambdas(S, f, sao_lambda);
	es.ee.counter = false;
	for (int n = 0; n < S.nctu; n++) {
		const int cx = n % W, cy = n / W;
		// wfpp_encode_select_bitstream :2299
		if (n == 0) {
			es.ee.bs = &es.rows[0];
			es.rows[0].init();
			es.ee.init_contexts(f.slice_type, f.qp);
			es.ee.start();
			es.ee.reset_bits();
		} else if (S.wpp) {
			if (cy > 0 && cx == 0) memcpy(es.ee.ctx, es.saved.ctx, sizeof es.ee.ctx);
			es.ee.bs = &es.rows[cy];
			if (cx == 0) {
				es.rows[cy].init();
				es.ee.start();
				es.ee.reset_bits();
			}
		}
		if (S.sao) {
			if (stats) sao_decide_ctu(es.ee, fr, n, stats[n], sao_lambda);   // (no statistics: the records already hold the decision, made on the device)
#if defined(HENC_SAO_TRACE)
			if (henc_sao_trace_file && stats) {
				fprintf(henc_sao_trace_file, "SAO frame=%d ctu=%d", f.num_encoded_frames, n);
				for (int c3 = 0; c3 < 3; c3++) {
					const SaoOffset &o = fr.ctu(n).sao_coded[c3];
					fprintf(henc_sao_trace_file, " | %d", o.mode_idc);
					if (o.mode_idc != SAO_OFF) {
						fprintf(henc_sao_trace_file, " %d %d :", o.type_idc, o.type_aux);
						if (o.mode_idc == SAO_NEW)
							for (int k = 0; k < (o.type_idc == SAO_BO ? 32 : 5); k++) fprintf(henc_sao_trace_file, " %d", o.offset[k]);
					}
				}
				fprintf(henc_sao_trace_file, " bits=%d\n", es.ee.bs->bitcount());
				for (int c3 = 0; c3 < 3; c3++)
					for (int t = 0; t < 5; t++) {
						fprintf(henc_sao_trace_file, "  ST %d %d :", c3, t);
						for (int k = 0; k < (t == 4 ? 32 : 5); k++) fprintf(henc_sao_trace_file, " %d/%d", stats[n][c3][t][0][k], stats[n][c3][t][1][k]);
						fprintf(henc_sao_trace_file, "\n");
					}
			}
#endif
			code_sao_blk_param(es.ee, fr.ctu(n).sao_coded, cx > 0, cy > 0);
		}
		encode_ctu_syntax(es.ee, fr, n);
		if (cx == 1 && cy + 1 != H && S.wpp) memcpy(es.saved.ctx, es.ee.ctx, sizeof es.saved.ctx);
		if ((S.wpp && cx + 1 == W) || (!S.wpp && n + 1 == S.nctu)) {
			es.ee.encode_trm(1);
			es.ee.finish();
			es.ee.bs->trailing_bits();
		}
	}
	// ---- access unit (encoder_engine_thread :3287-3330)
	const bool idr = f.slice_type == SLICE_I;
	if (idr) es.last_idr = f.poc;
	std::vector<std::vector<uint8_t>> nals;
	if (idr) {
		std::vector<uint8_t> vps, sps, pps;
		write_parameter_sets(S, profile, vps, sps, pps);
		nals.push_back(vps); nals.push_back(sps); nals.push_back(pps);
	}
	BitWriter sh;
	{
		// hmr_put_slice_header :375
		sh.write(1, 1);                        // first_slice_in_pic_flag
		if (idr) sh.write(0, 1);               // no_output_of_prior_pics_flag
		sh.uvlc(0);                            // pps id
		sh.uvlc(f.slice_type);
		if (!idr) {
			sh.write((f.poc - es.last_idr + 16) % 16, 4);
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
				ep[i] = es.rows[i].bytecnt + count_escapes(es.rows[i]);
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
		BitWriter all;
		all.need((size_t)sh.bytecnt + 16);
		size_t total = sh.bytecnt;
		for (int r = 0; r < (S.wpp ? H : 1); r++) total += es.rows[r].bytecnt;
		all.buf.assign(total + 16, 0);
		memcpy(all.buf.data(), sh.buf.data(), sh.bytecnt);
		all.bytecnt = sh.bytecnt;
		for (int r = 0; r < (S.wpp ? H : 1); r++) {
			memcpy(all.buf.data() + all.bytecnt, es.rows[r].buf.data(), es.rows[r].bytecnt);
			all.bytecnt += es.rows[r].bytecnt;
		}
		std::vector<uint8_t> nal;
		put_nal_header(nal, idr ? 19 : 1);
		nalu_ebsp(all, nal);
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
