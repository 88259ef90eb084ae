/*
 * TEST INFRASTRUCTURE - CPU oracle, see hmr_oracle.h.
 *
 * Plain-C restatement of the reference's hot-path kernels.  Arithmetic follows
 * the reference's scalar sources (the readable spec) with the
 * two SSE4.2-path deviations that are part of the contract:
 *   Q1  quant rounding offset 85 on non-I slices (hmr_sse42_functions_quant.c:47)
 *   Q2  byte-widened modified_variance (hmr_sse42_functions_pixel.c:953-1103)
 * and signed-saturating packs (_mm_packs_epi32) where the SSE path packs to i16.
 * Each function cites the reference file:line it restates.
 */
#include "hmr_oracle.h"

#include <limits.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int16_t sat16(int32_t v) { return (int16_t)clip3(v, -32768, 32767); }
static inline int ilog2(int n) { int s = 0; while ((1 << s) < n) s++; return s; }

/* ------------------------------------------------------------------ tables */

/* HEVC transform basis: T32[k][n] = c[(k*(2n+1)) mod 128] with the 33 integer
 * cosine samples of the standard; smaller sizes are sub-sampled rows.  Equals
 * g_aiT4..g_aiT32 of hmr_transform.c:54-128. */
static const int16_t k_cos64[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
				    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static int16_t g_dct[4][32 * 32];
/* DST-VII 4x4, hmr_transform.c:133-152 written as a matrix (coeff[k] = sum_n M[k][n] x[n]) */
static const int16_t k_dst4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};

/* default scaling lists, hmr_tables.h:53-82 (HEVC spec Table 7-6) */
static const int16_t k_sl_intra8[64] = {16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22,
					25, 29, 16, 16, 18, 21, 24, 27, 31, 36, 17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27,
					35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115};
static const int16_t k_sl_inter8[64] = {16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24,
					25, 28, 16, 17, 18, 20, 24, 25, 28, 33, 17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25,
					28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91};

static uint32_t g_scan[4][6][32 * 32];           /* [mode][log2 size] */
static int32_t g_quant[4][6][6][32 * 32];        /* [log2 size - 2][list][rem] */
static int32_t g_dequant[4][6][6][32 * 32];
static int g_tables_ready;

static int cos_sample(int m)
{
	m &= 127;
	if (m > 64) m = 128 - m;
	return m <= 32 ? k_cos64[m] : -k_cos64[64 - m];
}

/* up-right diagonal scan of a w x w block, hmr_tables.c:67-90 */
static void diag_scan(uint32_t *out, int w)
{
	int pos = 0, line, n = w * w;
	for (line = 0; pos < n; line++) {
		int prim = line, scnd = 0;
		while (prim >= w) { scnd++; prim--; }
		while (prim >= 0 && scnd < w) { out[pos++] = (uint32_t)(prim * w + scnd); scnd++; prim--; }
	}
}

static void build_tables(void)
{
	int l, k, n, mode, list, rem;
	uint32_t cg8[64];
	if (g_tables_ready) return;
	for (l = 2; l <= 5; l++) {
		int N = 1 << l, step = 32 / N;
		for (k = 0; k < N; k++)
			for (n = 0; n < N; n++)
				g_dct[l - 2][k * N + n] = (int16_t)cos_sample(k * step * (2 * n + 1));
	}
	/* scan pyramid, hmr_tables.c:62-188 (sizes 2..32; mode 0 "zigzag" is never filled by the reference) */
	diag_scan(cg8, 8);
	for (l = 1; l <= 5; l++) {
		int w = 1 << l;
		uint32_t *H = g_scan[ORA_SCAN_HOR][l], *V = g_scan[ORA_SCAN_VER][l], *D = g_scan[ORA_SCAN_DIAG][l];
		if (w == 2 || w == 4)
			diag_scan(D, w);
		if (w > 4) {
			int side = w >> 2, blks = side * side, b;
			uint32_t inner[16];
			const uint32_t *cg = (w == 32) ? cg8 : g_scan[ORA_SCAN_DIAG][l - 2];
			diag_scan(inner, 4);
			for (b = 0; b < blks; b++) {
				int oy = (int)cg[b] / side, ox = (int)cg[b] - oy * side, i;
				for (i = 0; i < 16; i++) {
					int py = (int)inner[i] >> 2, px = (int)inner[i] & 3;
					D[16 * b + i] = (uint32_t)(py * w + px + 4 * (ox + oy * w));
				}
			}
		}
		if (w > 2) {
			int side = w >> 2, by, bx, x, y, cnt = 0;
			for (by = 0; by < side; by++)
				for (bx = 0; bx < side; bx++)
					for (y = 0; y < 4; y++)
						for (x = 0; x < 4; x++)
							H[cnt++] = (uint32_t)(y * w + x + by * 4 * w + bx * 4);
			cnt = 0;
			for (bx = 0; bx < side; bx++)
				for (by = 0; by < side; by++)
					for (x = 0; x < 4; x++)
						for (y = 0; y < 4; y++)
							V[cnt++] = (uint32_t)(y * w + x + by * 4 * w + bx * 4);
		} else {
			H[0] = 0; H[1] = 1; H[2] = 2; H[3] = 3;
			V[0] = 0; V[1] = 2; V[2] = 1; V[3] = 3;
		}
	}
	/* quant / dequant pyramids, hmr_tables.c:221-250 + hmr_encoder_lib.c:112-140 */
	{
		static const int qs[6] = {26214, 23302, 20560, 18396, 16384, 14564};
		static const int iqs[6] = {40, 45, 51, 57, 64, 72};
		for (l = 2; l <= 5; l++) {
			int size = 1 << l, ms = size < 8 ? size : 8, ratio = size / ms, nlist = (l == 5) ? 2 : 6;
			for (list = 0; list < 6; list++) {
				int src_list = list;
				const int16_t *tab;
				if (l == 5) src_list = (list == 0) ? 0 : 1; /* [3][3] aliases [3][1]; other 32x32 lists are unused */
				(void)nlist;
				if (l == 2) tab = NULL;
				else if (l == 5) tab = src_list < 1 ? k_sl_intra8 : k_sl_inter8;
				else tab = src_list < 3 ? k_sl_intra8 : k_sl_inter8;
				for (rem = 0; rem < 6; rem++) {
					int i, j;
					for (j = 0; j < size; j++)
						for (i = 0; i < size; i++) {
							int t = tab ? tab[ms * (j / ratio) + i / ratio] : 16;
							g_quant[l - 2][list][rem][j * size + i] = (qs[rem] << 4) / t;
							g_dequant[l - 2][list][rem][j * size + i] = iqs[rem] * t;
						}
					if (ratio > 1) {
						g_quant[l - 2][list][rem][0] = (qs[rem] << 4) / 16;
						g_dequant[l - 2][list][rem][0] = iqs[rem] * 16;
					}
				}
			}
		}
	}
	(void)mode; (void)n;
	g_tables_ready = 1;
}

const uint32_t *ora_scan_table(int scan_mode, int log2_size) { build_tables(); return g_scan[scan_mode][log2_size]; }
const int32_t *ora_quant_table(int log2_size, int list, int rem) { build_tables(); return g_quant[log2_size - 2][list][rem]; }
const int32_t *ora_dequant_table(int log2_size, int list, int rem) { build_tables(); return g_dequant[log2_size - 2][list][rem]; }
const int16_t *ora_dct_matrix(int log2_size) { build_tables(); return g_dct[log2_size - 2]; }
const int16_t *ora_dst_matrix(void) { return k_dst4; }

/* ------------------------------------------------------------------ K6 copies */

/* hmr_sse42_functions_pixel.c:152 - width 4, 8 or a multiple of 16 at the call sites;
 * the SSE code rounds other widths up to 16, the contract is the [0,width) region. */
void ora_copy_16_16(const int16_t *src, uint32_t src_stride, int16_t *dst, uint32_t dst_stride, int height, int width)
{
	int j;
	for (j = 0; j < height; j++)
		memcpy(dst + (size_t)j * dst_stride, src + (size_t)j * src_stride, (size_t)width * sizeof(int16_t));
}

/* hmr_sse42_functions_pixel.c:236 - u8 -> i16 zero extension */
void ora_copy_8_16(const uint8_t *src, uint32_t src_stride, int16_t *dst, uint32_t dst_stride, int height, int width)
{
	int i, j;
	for (j = 0; j < height; j++)
		for (i = 0; i < width; i++)
			dst[(size_t)j * dst_stride + i] = (int16_t)src[(size_t)j * src_stride + i];
}

/* hmr_sse42_functions_pixel.c:319 - i16 -> u8 with unsigned saturation (_mm_packus_epi16) */
void ora_copy_16_8(const int16_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, int height, int width)
{
	int i, j;
	for (j = 0; j < height; j++)
		for (i = 0; i < width; i++)
			dst[(size_t)j * dst_stride + i] = (uint8_t)clip3(src[(size_t)j * src_stride + i], 0, 255);
}

/* ------------------------------------------------------------------ K1-K5 */

/* hmr_sse42_functions_pixel.c:462 (scalar hmr_motion_intra.c:51).  The SSE lanes accumulate
 * |a-b| in u16; exact for 8-bit samples, which is the only domain the encoder feeds it
 * (SURVEY.md §8 Q10).  Any size other than 4/8/16/32 takes the 64x64 path in the reference. */
uint32_t ora_sad(const int16_t *src, uint32_t src_stride, const int16_t *pred, uint32_t pred_stride, int size)
{
	uint32_t acc = 0;
	int x, y;
	if (size != 4 && size != 8 && size != 16 && size != 32) size = 64;
	for (y = 0; y < size; y++)
		for (x = 0; x < size; x++)
			acc += (uint32_t)abs((int16_t)(src[(size_t)y * src_stride + x] - pred[(size_t)y * pred_stride + x]));
	return acc;
}

/* hmr_sse42_functions_pixel.c:728: (a-b) in i16, squared and summed in 32 bits; pred_stride may be 0 */
uint32_t ora_ssd16b(const int16_t *src, uint32_t src_stride, const int16_t *pred, uint32_t pred_stride, int size)
{
	uint32_t acc = 0;
	int x, y;
	if (size != 4 && size != 8 && size != 16 && size != 32) size = 64;
	for (y = 0; y < size; y++)
		for (x = 0; x < size; x++) {
			int32_t d = (int16_t)(src[(size_t)y * src_stride + x] - pred[(size_t)y * pred_stride + x]);
			acc += (uint32_t)(d * d);
		}
	return acc;
}

/* hmr_sse42_functions_pixel.c:817 (scalar hmr_motion_intra.c:152) */
void ora_predict(const int16_t *orig, int orig_stride, const int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int size)
{
	int x, y;
	for (y = 0; y < size; y++)
		for (x = 0; x < size; x++)
			residual[y * residual_stride + x] = (int16_t)(orig[y * orig_stride + x] - pred[y * pred_stride + x]);
}

/* hmr_sse42_functions_pixel.c:919: saturating i16 add, then clip to 0..255; residual_stride may be 0 */
void ora_reconst(const int16_t *pred, int pred_stride, const int16_t *residual, int residual_stride, int16_t *decoded, int decoded_stride, int size)
{
	int x, y;
	for (y = 0; y < size; y++)
		for (x = 0; x < size; x++)
			decoded[y * decoded_stride + x] = (int16_t)clip3(sat16(pred[y * pred_stride + x] + residual[y * residual_stride + x]), 0, 255);
}

/* hmr_sse42_functions_pixel.c:1123 (Q2).  The SSE code loads int16 rows and zero-extends their
 * BYTES: per row it consumes `size` bytes (for size >= 16: bytes [32g,32g+8) and [32g+16,32g+24)
 * of every 16-sample group g), i.e. the low and high bytes of half of the samples. */
uint32_t ora_modified_variance(const int16_t *p, int size, int stride, int modif)
{
	uint32_t sum = 0, var = 0;
	int pass, j, k, avg = 0;
	for (pass = 0; pass < 2; pass++) {
		for (j = 0; j < size; j++) {
			const uint8_t *row = (const uint8_t *)(p + (size_t)j * stride);
			for (k = 0; k < size; k++) {
				int off = size < 16 ? k : (k >> 4) * 32 + ((k >> 3) & 1) * 16 + (k & 7);
				int v = row[off];
				if (!pass) sum += (uint32_t)v;
				else {
					int16_t d = (int16_t)(1 + (int16_t)((int16_t)(v - avg) * (int16_t)modif));
					var += (uint32_t)((int32_t)d * d);
				}
			}
		}
		if (!pass) avg = (int)(sum / (uint32_t)(size * size));
	}
	return var;
}

/* ------------------------------------------------------------------ K7/K8 intra prediction */

/* hmr_motion_intra.c:408-439 (SSE: hmr_sse42_functions_prediction.c:199; full-width write) */
void ora_intra_planar(int16_t *pred, int pred_stride, const int16_t *adi, int adi_size, int cu_size)
{
	const int16_t *mid = adi + (adi_size >> 1);
	int shift = ilog2(cu_size), i, j;
	int top_row[64], bottom_row[64], right_col[64], left_col[64];
	int bl = mid[-(cu_size + 1)], tr = mid[cu_size + 1];
	for (i = 0; i < cu_size; i++) {
		int left = mid[-(i + 1)], top = mid[i + 1];
		bottom_row[i] = bl - top;
		right_col[i] = tr - left;
		top_row[i] = top << shift;
		left_col[i] = left << shift;
	}
	for (j = 0; j < cu_size; j++) {
		int hor = left_col[j] + cu_size;
		for (i = 0; i < cu_size; i++) {
			hor += right_col[j];
			top_row[i] += bottom_row[i];
			pred[j * pred_stride + i] = (int16_t)((hor + top_row[i]) >> (shift + 1));
		}
	}
}

/* hmr_motion_intra.c:482-625 (SSE: hmr_sse42_functions_prediction.c:926); ctu->top/left are always 1 */
void ora_intra_angular(int16_t *pred, int pred_stride, const int16_t *adi, int adi_size, int cu_size, int cu_mode, int is_luma)
{
	static const int ang_table[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};            /* hmr_encoder_lib.c:35 */
	static const int inv_ang_table[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256}; /* :36 */
	const int16_t *mid = adi + (adi_size >> 1);
	int is_dc = cu_mode < 2, is_hor = !is_dc && cu_mode < 18, is_ver = !is_dc && !is_hor;
	int pred_angle = is_ver ? cu_mode - 26 : is_hor ? -(cu_mode - 10) : 0;
	int abs_angle = abs(pred_angle), sign = pred_angle < 0 ? -1 : (pred_angle > 0 ? 1 : 0);
	int inv_angle = inv_ang_table[abs_angle];
	int i, j;
	abs_angle = ang_table[abs_angle];
	pred_angle = sign * abs_angle;
	if (is_dc) {
		int acc = 0, dc;
		for (i = 1; i <= cu_size; i++) acc += mid[i];
		for (i = 1; i <= cu_size; i++) acc += mid[-i];
		dc = (uint16_t)((acc + cu_size) / (2 * cu_size));
		for (j = 0; j < cu_size; j++)
			for (i = 0; i < cu_size; i++)
				pred[j * pred_stride + i] = (uint8_t)dc;
		if (cu_size <= 16 && cu_mode == 1 && is_luma) {
			pred[0] = (int16_t)((mid[-1] + mid[1] + 2 * pred[0] + 2) >> 2);
			for (i = 1; i < cu_size; i++) pred[i] = (int16_t)((mid[1 + i] + 3 * pred[i] + 2) >> 2);
			for (j = 1; j < cu_size; j++) pred[j * pred_stride] = (int16_t)((mid[-1 - j] + 3 * pred[j * pred_stride] + 2) >> 2);
		}
		return;
	}
	{
		int16_t above[2 * 64 + 1 + 64], left[2 * 64 + 1 + 64];
		int16_t *ref_main, *ref_side;
		int filter = is_luma ? (cu_size <= 16) : 0;
		int s1 = is_hor ? 1 : pred_stride, s2 = is_hor ? pred_stride : 1;
		memset(above, 0, sizeof above);
		memset(left, 0, sizeof left);
		if (pred_angle < 0) {
			int inv_sum = 128;
			for (i = 0; i < cu_size + 1; i++) { above[i + cu_size - 1] = mid[i]; left[i + cu_size - 1] = mid[-i]; }
			ref_main = (is_ver ? above : left) + (cu_size - 1);
			ref_side = (is_ver ? left : above) + (cu_size - 1);
			for (i = -1; i > ((cu_size * pred_angle) >> 5); i--) {
				inv_sum += inv_angle;
				ref_main[i] = ref_side[inv_sum >> 8];
			}
		} else {
			for (i = 0; i < 2 * cu_size + 1; i++) { above[i] = mid[i]; left[i] = mid[-i]; }
			ref_main = is_ver ? above : left;
			ref_side = is_ver ? left : above;
		}
		if (pred_angle == 0) {
			for (j = 0; j < cu_size; j++)
				for (i = 0; i < cu_size; i++)
					pred[j * s1 + i * s2] = (uint8_t)ref_main[i + 1];
			if (filter)
				for (i = 0; i < cu_size; i++)
					pred[i * s1] = (int16_t)clip3(pred[i * s1] + ((ref_side[i + 1] - ref_side[0]) >> 1), 0, 255);
		} else {
			int pos = 0;
			for (j = 0; j < cu_size; j++) {
				int delta, fract;
				pos += pred_angle;
				delta = pos >> 5;
				fract = pos & 31;
				for (i = 0; i < cu_size; i++) {
					int idx = i + delta + 1;
					pred[j * s1 + i * s2] = fract ? (uint8_t)(((32 - fract) * ref_main[idx] + fract * ref_main[idx + 1] + 16) >> 5)
								      : (uint8_t)ref_main[idx];
				}
			}
		}
	}
}

/* ------------------------------------------------------------------ K19 intra reference build */

/* hmr_motion_intra.c:246-390 with the partition node flattened: left/top/bottom_left/top_right are the
 * node's neighbour flags, bl_size / tr_size the reference's left_bottom_size / top_right_size
 * (min(n, rows/cols left inside the picture), :289,335).  `decoded` points at the top-left corner
 * sample (-1,-1).  adi[0] is the bottom-most bottom-left sample, adi[2n] the corner, adi[4n] the
 * right-most top-right sample. */
void ora_fill_reference_samples(const int16_t *decoded, int stride, int n, int left, int top, int bottom_left, int top_right,
				int bl_size, int tr_size, int16_t *adi)
{
	int adi_size = 4 * n + 1, i;
	int16_t first_sample = 0, last_sample = 0;
	int16_t *pad_left = adi, *pad_top = adi;
	int pad_left_size = 0, pad_top_size = 0;
	int16_t *ptr;
	const int16_t *ref;
	if (!left && !top) {
		for (i = 0; i < adi_size; i++) adi[i] = 128;
		return;
	}
	ref = decoded + n * stride;
	ptr = adi + n;
	if (left) {
		for (i = 0; i < n; i++) *ptr++ = ref[-i * stride];
		first_sample = ptr[-n];
		last_sample = ptr[-1];
	} else {
		pad_left = ptr;
		pad_left_size = n;
	}
	ref = decoded + (n + 1) * stride;
	ptr = adi + n - 1;
	if (bottom_left) {
		for (i = 0; i < bl_size; i++) *ptr-- = ref[i * stride];
		first_sample = ptr[1];
		if (bl_size != n) { pad_left = adi; pad_left_size = n - bl_size; }
	} else {
		pad_left = adi;
		if (left) pad_left_size = n;
		else pad_left_size += n;
	}
	ptr = adi + 2 * n + 1;
	ref = decoded + 1;
	if (top) {
		for (i = 0; i < n; i++) *ptr++ = *ref++;
		if (!left) first_sample = ptr[-n];
		last_sample = ptr[-1];
	} else {
		pad_top = ptr;
		pad_top_size = n;
	}
	if (top_right) {
		for (i = 0; i < tr_size; i++) *ptr++ = *ref++;
		last_sample = ptr[-1];
		if (tr_size != n) { pad_top = ptr; pad_top_size = n - tr_size; }
	} else {
		if (top) { pad_top = ptr; pad_top_size = n; }
		else pad_top_size += n;
	}
	if (left && top) adi[2 * n] = decoded[0];
	else if (left) { pad_top--; pad_top_size++; }
	else pad_left_size++;
	for (i = 0; i < pad_left_size; i++) *pad_left++ = first_sample;
	for (i = 0; i < pad_top_size; i++) *pad_top++ = last_sample;
}

/* hmr_motion_intra.c:189-243: [1 2 1]/4 smoothing, or the strong bilinear filter for n >= 32 when
 * both edges are nearly linear.  size_shift = log2(2n) for the luma sizes that reach the strong
 * branch (max_cu_size_shift - depth + 1, :206). */
void ora_adi_filter(const int16_t *adi, int16_t *out, int adi_size, int n, int strong_enabled)
{
	int i;
	if (strong_enabled) {
		int bl = adi[0], tl = adi[2 * n], tr = adi[adi_size - 1];
		int lin_left = abs(bl + tl - 2 * adi[n]) < 8;
		int lin_top = abs(tl + tr - 2 * adi[3 * n]) < 8;
		if (n >= 32 && lin_left && lin_top) {
			int shift = ilog2(2 * n);
			out[0] = adi[0];
			out[2 * n] = adi[2 * n];
			out[adi_size - 1] = adi[adi_size - 1];
			for (i = 1; i < 2 * n; i++) out[i] = (int16_t)(((2 * n - i) * bl + i * tl + n) >> shift);
			for (i = 1; i < 2 * n; i++) out[2 * n + i] = (int16_t)(((2 * n - i) * tl + i * tr + n) >> shift);
			return;
		}
	}
	out[0] = adi[0];
	for (i = 1; i < adi_size - 1; i++) out[i] = (int16_t)((adi[i - 1] + 2 * adi[i] + adi[i + 1] + 2) >> 2);
	out[adi_size - 1] = adi[adi_size - 1];
}

/* ------------------------------------------------------------------ K9-K11 interpolation */

static const int16_t k_luma_taps[4][8] = {   /* hmr_motion_inter.c:240-246 */
	{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int16_t k_chroma_taps[8][4] = { /* hmr_motion_inter.c:248-258 */
	{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

/* hmr_motion_inter.c:262-311 (filter_copy) */
static void filter_copy(const int16_t *src, int ss, int16_t *dst, int ds, int w, int h, int first, int last)
{
	int r, c;
	for (r = 0; r < h; r++)
		for (c = 0; c < w; c++) {
			int v = src[r * ss + c];
			if (first == last) dst[r * ds + c] = (int16_t)v;
			else if (first) dst[r * ds + c] = (int16_t)((int16_t)(v << 6) - 8192);
			else dst[r * ds + c] = (int16_t)clip3((v + 8192 + 32) >> 6, 0, 255);
		}
}

/* hmr_motion_inter.c:314-377 / :878-936: separable FIR with the HM stage rules */
static void fir(const int16_t *src, int ss, int16_t *dst, int ds, const int16_t *taps, int ntaps, int w, int h, int vert, int first, int last)
{
	int rs = vert ? ss : 1, shift = 6, offset, r, c, t;
	src -= (ntaps / 2 - 1) * rs;
	if (last) {
		shift += first ? 0 : 6;
		offset = 1 << (shift - 1);
		offset += first ? 0 : 8192 << 6;
	} else {
		shift -= first ? 6 : 0;
		offset = first ? -(8192 << shift) : 0;
	}
	for (r = 0; r < h; r++)
		for (c = 0; c < w; c++) {
			int sum = 0;
			int16_t v;
			for (t = 0; t < ntaps; t++) sum += src[r * ss + c + t * rs] * taps[t];
			v = sat16((sum + offset) >> shift);   /* _mm_packs_epi32 */
			if (last) v = (int16_t)clip3(v, 0, 255);
			dst[r * ds + c] = v;
		}
}

void ora_interpolate_luma(const int16_t *src, int ss, int16_t *dst, int ds, int fraction, int w, int h, int vert, int first, int last)
{
	if (fraction == 0) filter_copy(src, ss, dst, ds, w, h, first, last);
	else fir(src, ss, dst, ds, k_luma_taps[fraction], 8, w, h, vert, first, last);
}

/* hmr_sse42_functions_inter_prediction.c:818: fraction 0 with width < 4 is a silent no-op */
void ora_interpolate_chroma(const int16_t *src, int ss, int16_t *dst, int ds, int fraction, int w, int h, int vert, int first, int last)
{
	if (fraction == 0) {
		if (w < 4) return;
		filter_copy(src, ss, dst, ds, w, h, first, last);
	} else
		fir(src, ss, dst, ds, k_chroma_taps[fraction], 4, w, h, vert, first, last);
}

/* hmr_sse42_functions_inter_prediction.c:944: clip((a + b + 64 + 2*8192) >> 7) */
void ora_weighted_average(const int16_t *a, int as, const int16_t *b, int bs, int16_t *d, int ds, int h, int w)
{
	int r, c;
	for (r = 0; r < h; r++)
		for (c = 0; c < w; c++)
			d[r * ds + c] = (int16_t)clip3(sat16((a[r * as + c] + b[r * bs + c] + 64 + 16384) >> 7), 0, 255);
}

/* ------------------------------------------------------------------ K12/K13 transforms */

/* One 1-D stage written as a matrix product: out[k*n + j] = sat16((sum_i M[k][i] * in[j][i] + rnd) >> shift).
 * The butterflies of hmr_transform.c:154-500 are an exact factorisation of this product (no
 * intermediate rounding), and the SSE kernels pack with signed saturation (Q3). */
static void fwd_stage(const int16_t *in, int in_stride, int16_t *out, const int16_t *M, int n, int shift)
{
	int j, k, i, rnd = 1 << (shift - 1);
	for (j = 0; j < n; j++)
		for (k = 0; k < n; k++) {
			int32_t s = 0;
			for (i = 0; i < n; i++) s += M[k * n + i] * in[j * in_stride + i];
			out[k * n + j] = sat16((s + rnd) >> shift);
		}
}

/* out[j][k] = sat16((sum_i M[i][k] * in[i*n + j] + rnd) >> shift) */
static void inv_stage(const int16_t *in, int16_t *out, int out_stride, const int16_t *M, int n, int shift)
{
	int j, k, i, rnd = 1 << (shift - 1);
	for (j = 0; j < n; j++)
		for (k = 0; k < n; k++) {
			int32_t s = 0;
			for (i = 0; i < n; i++) s += M[i * n + k] * in[i * n + j];
			out[j * out_stride + k] = sat16((s + rnd) >> shift);
		}
}

/* hmr_sse42_functions_transform.c:1670: shifts {log2N-1, log2N+6} for 8-bit video; DST-VII for 4x4 intra luma */
void ora_transform(const int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst)
{
	int16_t tmp[32 * 32];
	int l = ilog2(n);
	const int16_t *M = (n == 4 && is_dst) ? k_dst4 : ora_dct_matrix(l);
	if (n != 4 && n != 8 && n != 16 && n != 32) return; /* unsupported sizes fall through silently, :1672-1694 */
	fwd_stage(block, block_stride, tmp, M, n, l - 1);
	fwd_stage(tmp, n, coeff, M, n, l + 6);
}

/* hmr_sse42_functions_transform.c:1700: shifts {7, 12} */
void ora_itransform(int16_t *block, const int16_t *coeff, int block_stride, int n, int is_dst)
{
	int16_t tmp[32 * 32];
	int l = ilog2(n);
	const int16_t *M = (n == 4 && is_dst) ? k_dst4 : ora_dct_matrix(l);
	if (n != 4 && n != 8 && n != 16 && n != 32) return;
	inv_stage(coeff, tmp, n, M, n, 7);
	inv_stage(tmp, block, block_stride, M, n, 12);
}

/* ------------------------------------------------------------------ K14/K15 quant */

/* hmr_quant.c:61-169, shared by the scalar and SSE quantisers */
void ora_sign_bit_hiding(int16_t *dst, const int16_t *src, const uint32_t *scan, const int16_t *delta_u, int n_coeffs)
{
	int last_cg = -1, subset, n;
	for (subset = (n_coeffs - 1) >> 4; subset >= 0; subset--) {
		int sub_pos = subset << 4, first_nz = 16, last_nz = -1, abs_sum = 0;
		for (n = 15; n >= 0; --n)
			if (dst[scan[n + sub_pos]]) { last_nz = n; break; }
		for (n = 0; n < 16; n++)
			if (dst[scan[n + sub_pos]]) { first_nz = n; break; }
		for (n = first_nz; n <= last_nz; n++) abs_sum += dst[scan[n + sub_pos]];
		if (last_nz >= 0 && last_cg == -1) last_cg = 1;
		if (last_nz - first_nz >= 4) {
			unsigned signbit = dst[scan[sub_pos + first_nz]] > 0 ? 0 : 1;
			if (signbit != (unsigned)(abs_sum & 1)) {
				int min_cost = INT_MAX, min_pos = -1, final_change = 0, cur_cost = INT_MAX, cur_change = 0;
				for (n = (last_cg == 1 ? last_nz : 15); n >= 0; --n) {
					unsigned pos = scan[n + sub_pos];
					if (dst[pos] != 0) {
						if (delta_u[pos] > 0) { cur_cost = -delta_u[pos]; cur_change = 1; }
						else if (n == first_nz && abs(dst[pos]) == 1) cur_cost = INT_MAX;
						else { cur_cost = delta_u[pos]; cur_change = -1; }
					} else if (n < first_nz) {
						unsigned this_sign = src[pos] >= 0 ? 0 : 1;
						if (this_sign != signbit) cur_cost = INT_MAX;
						else { cur_cost = -delta_u[pos]; cur_change = 1; }
					} else { cur_cost = -delta_u[pos]; cur_change = 1; }
					if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = (int)pos; }
				}
				if (dst[min_pos] == 32767 || dst[min_pos] == -32768) final_change = -1;
				if (src[min_pos] >= 0) dst[min_pos] = (int16_t)(dst[min_pos] + final_change);
				else dst[min_pos] = (int16_t)(dst[min_pos] - final_change);
			}
		}
		if (last_cg == 1) last_cg = 0;
	}
}

/* hmr_sse42_functions_quant.c:34-131.  32-bit wrapping arithmetic (_mm_mullo_epi32), arithmetic shifts,
 * signed-saturating packs, sign re-applied with a 16-bit multiply. */
void ora_quant(const int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra,
	       int slice_is_intra, int sign_hiding, int *ac_sum, int cu_size, int per, int rem)
{
	int inv_depth = 6 - (depth + (comp != 0));
	const uint32_t *scan = ora_scan_table(scan_mode, inv_depth);
	const int32_t *q = ora_quant_table(inv_depth, (is_intra ? 0 : 3) + comp, rem);
	int qbits = 14 + per + (15 - 8 - inv_depth), qbits8 = qbits - 8;
	int32_t add = (int32_t)((uint32_t)(slice_is_intra ? 171 : 85) << (qbits - 9));
	int16_t scratch[32 * 32];
	int n, total = cu_size * cu_size;
	uint32_t sum = 0;
	if (!delta_u) delta_u = scratch;
	for (n = 0; n < total; n++) {
		uint32_t a = (uint16_t)(src[n] < 0 ? -src[n] : src[n]);      /* _mm_abs_epi16 then zero-extend */
		int32_t aux = (int32_t)(a * (uint32_t)q[n]);
		int32_t c = (int32_t)((uint32_t)aux + (uint32_t)add) >> qbits;
		int32_t d = (int32_t)((uint32_t)aux - ((uint32_t)c << qbits)) >> qbits8;
		int sgn = src[n] > 0 ? 1 : (src[n] < 0 ? -1 : 0);
		sum += (uint32_t)c;
		dst[n] = (int16_t)(sgn * sat16(c));
		delta_u[n] = sat16(d);
	}
	*ac_sum = (int)sum;
	if (sign_hiding && *ac_sum >= 2) ora_sign_bit_hiding(dst, src, scan, delta_u, total);
}

/* hmr_sse42_functions_quant.c:135-246; list index keeps the reference's precedence slip (:138) */
void ora_inv_quant(const int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem)
{
	int inv_depth = 6 - (depth + (comp != 0));
	const int32_t *iq = ora_dequant_table(inv_depth, is_intra ? 0 : 3 + comp, rem);
	int iq_shift = 20 - 14 - (15 - 8 - inv_depth) + 4, n, total = cu_size * cu_size;
	if (iq_shift > per) {
		int32_t add = 1 << (iq_shift - per - 1);
		int sh = iq_shift - per;
		for (n = 0; n < total; n++)
			dst[n] = sat16((int32_t)((uint32_t)(int32_t)src[n] * (uint32_t)iq[n] + (uint32_t)add) >> sh);
	} else {
		int sh = per - iq_shift;
		for (n = 0; n < total; n++)
			dst[n] = sat16((int32_t)(((uint32_t)(int32_t)src[n] * (uint32_t)iq[n]) << sh));
	}
}

/* ====================================================================================================
 * Frame-level in-loop kernels (not in the reference's table): deblocking, SAO
 * statistics, SAO offset, border padding.  The reference runs them CTU by CTU in a lagged software
 * pipeline (hmr_encoder_lib.c:2386); every sample they read is final by the time it is read, so the
 * result equals two frame passes (all vertical edges, then all horizontal edges), one statistics pass
 * on the deblocked picture and one offset pass from a pre-SAO copy - which is what is restated here.
 * Side-info is a structure-of-arrays over the picture's 4x4 units in raster order.
 * ==================================================================================================== */

#define UNIT_INTRA 1
#define UNIT_CBF_Y 2
#define UNIT_EDGE_VER 4
#define UNIT_EDGE_HOR 8

static const uint8_t k_tc_table[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
				       2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static const uint8_t k_beta_table[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
					 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static int chroma_qp(int qpi)   /* chroma_scale_conversion_table, hmr_encoder_lib.c:2245 */
{
	static const uint8_t mid[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
	qpi = clip3(qpi, 0, 57);
	return qpi < 30 ? qpi : (qpi < 44 ? mid[qpi - 30] : qpi - 6);
}

/* Transform/CU edge flags from the coding tree (hmr_deblocking_filter.c:737-825): a unit's left/top edge is an
 * edge when it lies on the boundary of the leaf of size max(8, 64 >> (pred_depth + tr_idx)); picture borders are not. */
void ora_make_edge_flags(const uint8_t *pred_depth, const uint8_t *tr_idx, int width, int height, int units_stride, uint8_t *flags)
{
	int ux, uy;
	for (uy = 0; uy < height / 4; uy++)
		for (ux = 0; ux < width / 4; ux++) {
			int o = uy * units_stride + ux, total = pred_depth[o] + tr_idx[o];
			int leaf = 64 >> total;
			if (leaf < 8) leaf = 8;
			flags[o] &= (uint8_t)~(UNIT_EDGE_VER | UNIT_EDGE_HOR);
			if (ux && (ux * 4) % leaf == 0) flags[o] |= UNIT_EDGE_VER;
			if (uy && (uy * 4) % leaf == 0) flags[o] |= UNIT_EDGE_HOR;
		}
}

/* get_boundary_strength_single, hmr_deblocking_filter.c:138-268, P slices (single reference list) */
static int boundary_strength(int q, int p, const int16_t *mvx, const int16_t *mvy, const int8_t *ref_idx, const uint8_t *flags)
{
	int mqx, mqy, mpx, mpy;
	if ((flags[p] & UNIT_INTRA) || (flags[q] & UNIT_INTRA)) return 2;
	if ((flags[p] & UNIT_CBF_Y) || (flags[q] & UNIT_CBF_Y)) return 1;
	mqx = ref_idx[q] < 0 ? 0 : mvx[q]; mqy = ref_idx[q] < 0 ? 0 : mvy[q];
	mpx = ref_idx[p] < 0 ? 0 : mvx[p]; mpy = ref_idx[p] < 0 ? 0 : mvy[p];
	if ((ref_idx[p] < 0 ? -1 : ref_idx[p]) != (ref_idx[q] < 0 ? -1 : ref_idx[q])) return 1;
	return (abs(mqx - mpx) >= 4 || abs(mqy - mpy) >= 4) ? 1 : 0;
}

/* filter_luma :287-349 on one line; s = step across the edge */
static void luma_line(int16_t *src, int s, int tc, int strong, int thr_cut, int filt_p, int filt_q)
{
	int m4 = src[0], m3 = src[-s], m5 = src[s], m2 = src[-2 * s], m6 = src[2 * s], m1 = src[-3 * s], m7 = src[3 * s], m0 = src[-4 * s];
	if (strong) {
		src[-s] = (int16_t)clip3((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3, m3 - 2 * tc, m3 + 2 * tc);
		src[0] = (int16_t)clip3((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3, m4 - 2 * tc, m4 + 2 * tc);
		src[-2 * s] = (int16_t)clip3((m1 + m2 + m3 + m4 + 2) >> 2, m2 - 2 * tc, m2 + 2 * tc);
		src[s] = (int16_t)clip3((m3 + m4 + m5 + m6 + 2) >> 2, m5 - 2 * tc, m5 + 2 * tc);
		src[-3 * s] = (int16_t)clip3((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - 2 * tc, m1 + 2 * tc);
		src[2 * s] = (int16_t)clip3((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3, m6 - 2 * tc, m6 + 2 * tc);
	} else {
		int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
		if (abs(delta) < thr_cut) {
			int tc2 = tc >> 1;
			delta = clip3(delta, -tc, tc);
			src[-s] = (int16_t)clip3(m3 + delta, 0, 255);
			src[0] = (int16_t)clip3(m4 - delta, 0, 255);
			if (filt_p) src[-2 * s] = (int16_t)clip3(m2 + clip3((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1, -tc2, tc2), 0, 255);
			if (filt_q) src[s] = (int16_t)clip3(m5 + clip3((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1, -tc2, tc2), 0, 255);
		}
	}
}

static int strong_decision(const int16_t *src, int s, int d, int beta, int tc)   /* use_strong_filter :275 */
{
	int m4 = src[0], m3 = src[-s], m7 = src[3 * s], m0 = src[-4 * s];
	return (abs(m0 - m3) + abs(m7 - m4) < (beta >> 3)) && (d < (beta >> 2)) && (abs(m3 - m4) < ((tc * 5 + 1) >> 1));
}

/* One direction over the whole picture: deblock_filter_luma :353-458 and deblock_filter_chroma :504-634 */
static void deblock_pass(int dir, int16_t *y, int ys, int16_t *u, int16_t *v, int cs, int width, int height, int units_stride, const int16_t *mvx,
			 const int16_t *mvy, const int8_t *ref_idx, const uint8_t *qp, const uint8_t *flags, int cb_off, int cr_off, int beta_off,
			 int tc_off, uint8_t *bs_out)
{
	int ux, uy, i, comp;
	for (uy = 0; uy < height / 4; uy++)
		for (ux = 0; ux < width / 4; ux++) {
			int q = uy * units_stride + ux, p, bs, qpa, tc, beta;
			int on_grid = dir == 0 ? (ux & 1) == 0 : (uy & 1) == 0;   /* edges on the 8x8 luma grid only, :667,689 */
			if (bs_out) bs_out[q] = 0;
			if (!(flags[q] & (dir == 0 ? UNIT_EDGE_VER : UNIT_EDGE_HOR)) || !on_grid) continue;
			p = dir == 0 ? q - 1 : q - units_stride;
			bs = boundary_strength(q, p, mvx, mvy, ref_idx, flags);
			if (bs_out) bs_out[q] = (uint8_t)(0x80 | bs);
			if (!bs) continue;
			qpa = (qp[p] + qp[q] + 1) >> 1;
			{
				int16_t *e = y + (size_t)uy * 4 * ys + ux * 4;
				int s = dir == 0 ? 1 : ys, t = dir == 0 ? ys : 1;   /* s across the edge, t along it */
				int dp0, dq0, dp3, dq3, d0, d3, d, side;
				tc = k_tc_table[clip3(qpa + 2 * (bs - 1) + (tc_off << 1), 0, 53)];
				beta = k_beta_table[clip3(qpa + (beta_off << 1), 0, 51)];
				side = (beta + (beta >> 1)) >> 3;
				dp0 = abs(e[-3 * s] - 2 * e[-2 * s] + e[-s]);
				dq0 = abs(e[0] - 2 * e[s] + e[2 * s]);
				dp3 = abs(e[3 * t - 3 * s] - 2 * e[3 * t - 2 * s] + e[3 * t - s]);
				dq3 = abs(e[3 * t] - 2 * e[3 * t + s] + e[3 * t + 2 * s]);
				d0 = dp0 + dq0; d3 = dp3 + dq3; d = d0 + d3;
				if (d < beta) {
					int fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
					int sw = strong_decision(e, s, 2 * d0, beta, tc) && strong_decision(e + 3 * t, s, 2 * d3, beta, tc);
					for (i = 0; i < 4; i++) luma_line(e + i * t, s, tc, sw, tc * 10, fp, fq);
				}
			}
			/* chroma: only bs == 2, only edges on the 8x8 chroma grid (16 luma samples), :533,545 */
			if (bs > 1 && (dir == 0 ? (ux & 3) == 0 : (uy & 3) == 0))
				for (comp = 1; comp <= 2; comp++) {
					int16_t *pl = comp == 1 ? u : v;
					int16_t *e = pl + (size_t)uy * 2 * cs + ux * 2;
					int s = dir == 0 ? 1 : cs, t = dir == 0 ? cs : 1;
					int cq = chroma_qp(qpa + (comp == 1 ? cb_off : cr_off));
					int tcc = k_tc_table[clip3(cq + 2 * (bs - 1) + (tc_off << 1), 0, 53)];
					for (i = 0; i < 2; i++) {
						int16_t *x = e + i * t;
						int m4 = x[0], m3 = x[-s], m5 = x[s], m2 = x[-2 * s];
						int delta = clip3((((m4 - m3) << 2) + m2 - m5 + 4) >> 3, -tcc, tcc);
						x[-s] = (int16_t)clip3(m3 + delta, 0, 255);
						x[0] = (int16_t)clip3(m4 - delta, 0, 255);
					}
				}
		}
}

void ora_deblock_frame(int16_t *y, int ys, int16_t *u, int16_t *v, int cs, int width, int height, int units_stride, const int16_t *mvx,
		       const int16_t *mvy, const int8_t *ref_idx, const uint8_t *qp, const uint8_t *flags, int cb_qp_offset, int cr_qp_offset,
		       int beta_offset_div2, int tc_offset_div2, uint8_t *bs_ver, uint8_t *bs_hor)
{
	deblock_pass(0, y, ys, u, v, cs, width, height, units_stride, mvx, mvy, ref_idx, qp, flags, cb_qp_offset, cr_qp_offset, beta_offset_div2, tc_offset_div2, bs_ver);
	deblock_pass(1, y, ys, u, v, cs, width, height, units_stride, mvx, mvy, ref_idx, qp, flags, cb_qp_offset, cr_qp_offset, beta_offset_div2, tc_offset_div2, bs_hor);
}

static inline int sgn(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

/* SAO statistics per CTU and component, hmr_sse42_sao.c:35 (scalar spec hmr_sao.c:75-348), written per sample:
 * class = sign(c - a) + sign(c - b) for the two neighbours of the edge type, regions as derived from the loops
 * (right 5/3 columns and bottom 4/2 rows skipped when a neighbour CTU exists, hmr_sao.c:60-61).
 * stats layout: [ctu][comp][type][0 = diff, 1 = count][32] int32; EO classes -2..2 at index 0..4, BO bands 0..31. */
void ora_sao_stats_frame(const int16_t *oy, const int16_t *ou, const int16_t *ov, int os_y, int os_c, const int16_t *ry, const int16_t *ru,
			 const int16_t *rv, int rs_y, int rs_c, int width, int height, int32_t *stats)
{
	static const int skip_r[3] = {5, 3, 3}, skip_b[3] = {4, 2, 2};
	static const int dx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, dy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
	int ctus_x = (width + 63) / 64, ctus_y = (height + 63) / 64, cx, cy, comp, type, x, y;
	memset(stats, 0, (size_t)ctus_x * ctus_y * 3 * 5 * 2 * 32 * sizeof(int32_t));
	for (cy = 0; cy < ctus_y; cy++)
		for (cx = 0; cx < ctus_x; cx++) {
			int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
			int la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
			for (comp = 0; comp < 3; comp++) {
				int sh = comp ? 1 : 0, h = hl >> sh, w = wl >> sh, rs = comp ? rs_c : rs_y, os = comp ? os_c : os_y;
				const int16_t *rec = (comp == 0 ? ry : comp == 1 ? ru : rv) + (size_t)(cy * 64 >> sh) * rs + (cx * 64 >> sh);
				const int16_t *org = (comp == 0 ? oy : comp == 1 ? ou : ov) + (size_t)(cy * 64 >> sh) * os + (cx * 64 >> sh);
				for (type = 0; type < 5; type++) {
					int32_t *diff = stats + (((((size_t)cy * ctus_x + cx) * 3 + comp) * 5 + type) * 2) * 32, *count = diff + 32;
					int sx = (type == 1 || type == 4) ? 0 : (la ? 0 : 1);
					int ex = ra ? w - skip_r[comp] : ((type == 1 || type == 4) ? w : w - 1);
					int sy = (type == 0 || type == 4) ? 0 : (ta ? 0 : 1);
					int ey = ba ? h - skip_b[comp] : ((type == 0 || type == 4) ? h : h - 1);
					for (y = sy; y < ey; y++)
						for (x = sx; x < ex; x++) {
							int c = rec[(size_t)y * rs + x], k;
							if (type == 4) k = c >> 3;
							else {
								/* first row of the diagonal types: the above-left (135) sample of column 0 needs the
								 * above-left CTU, i.e. both neighbours (first_line_start_x, hmr_sao.c:214) */
								if (type == 2 && y == 0 && x == 0 && !(ta && la)) continue;
								k = 2 + sgn(c - rec[(size_t)(y + dy[type][0]) * rs + x + dx[type][0]]) +
								    sgn(c - rec[(size_t)(y + dy[type][1]) * rs + x + dx[type][1]]);
							}
							diff[k] += org[(size_t)y * os + x] - c;
							count[k]++;
						}
				}
			}
		}
}

/* SAO offset, sao_offset_ctu hmr_sao.c:1210 + offset_block :960, per sample.  src = pre-SAO (deblocked) picture,
 * dst = output picture, which must already hold a copy of src (samples outside the per-type regions stay as they are).
 * params[ctu][comp][34] = {modeIdc (0 = off), typeIdc, offset[32]}; EO offsets at [0..4] for classes -2..2. */
void ora_sao_apply_frame(const int16_t *sy_, const int16_t *su, const int16_t *sv, int16_t *dy_, int16_t *du, int16_t *dv, int stride_y, int stride_c,
			 int width, int height, const int32_t *params)
{
	static const int dx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, dyy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
	int ctus_x = (width + 63) / 64, ctus_y = (height + 63) / 64, cx, cy, comp, x, y;
	for (cy = 0; cy < ctus_y; cy++)
		for (cx = 0; cx < ctus_x; cx++) {
			int hl = (cy * 64 + 64 > height) ? height - cy * 64 : 64, wl = (cx * 64 + 64 > width) ? width - cx * 64 : 64;
			int la = cx > 0, ta = cy > 0, ra = cx * 64 + 64 < width, ba = cy * 64 + 64 < height;
			const int32_t *pc = params + ((size_t)cy * ctus_x + cx) * 3 * 34;
			if (!pc[0] && !pc[34] && !pc[68]) continue;
			for (comp = 0; comp < 3; comp++) {
				const int32_t *p = pc + comp * 34, *off = p + 2;
				int sh = comp ? 1 : 0, h = hl >> sh, w = wl >> sh, st = comp ? stride_c : stride_y, type = p[1];
				const int16_t *src = (comp == 0 ? sy_ : comp == 1 ? su : sv) + (size_t)(cy * 64 >> sh) * st + (cx * 64 >> sh);
				int16_t *dst = (comp == 0 ? dy_ : comp == 1 ? du : dv) + (size_t)(cy * 64 >> sh) * st + (cx * 64 >> sh);
				if (!p[0]) continue;
				for (y = 0; y < h; y++)
					for (x = 0; x < w; x++) {
						int c = src[(size_t)y * st + x], k;
						if (type == 4) k = c >> 3;
						else {
							/* availability of the two neighbours this type reads */
							int x0 = x + dx[type][0], y0 = y + dyy[type][0], x1 = x + dx[type][1], y1 = y + dyy[type][1];
							int ok0 = (x0 >= 0 || la) && (x0 < w || ra) && (y0 >= 0 || ta) && (y0 < h || ba);
							int ok1 = (x1 >= 0 || la) && (x1 < w || ra) && (y1 >= 0 || ta) && (y1 < h || ba);
							if (!ok0 || !ok1) continue;
							/* the reference also skips column 0 / w-1 rows it cannot start from, see the per-type loops */
							if (type != 1 && ((x == 0 && !la) || (x == w - 1 && !ra))) continue;
							if (type != 0 && ((y == 0 && !ta) || (y == h - 1 && !ba))) continue;
							k = 2 + sgn(c - src[(size_t)y0 * st + x0]) + sgn(c - src[(size_t)y1 * st + x1]);
						}
						dst[(size_t)y * st + x] = (int16_t)clip3(c + off[k], 0, 255);
					}
			}
		}
}

/* reference_picture_border_padding_ctu over every CTU, hmr_encoder_lib.c:1723 == replicate the picture edge
 * into the pad_x / pad_y margins (corners from the corner sample).  `pic` points at sample (0,0). */
void ora_pad_plane(int16_t *pic, int stride, int width, int height, int pad_x, int pad_y)
{
	int x, y;
	for (y = 0; y < height; y++)
		for (x = 0; x < pad_x; x++) {
			pic[(size_t)y * stride - 1 - x] = pic[(size_t)y * stride];
			pic[(size_t)y * stride + width + x] = pic[(size_t)y * stride + width - 1];
		}
	for (y = 0; y < pad_y; y++) {
		memcpy(pic + (ptrdiff_t)(-1 - y) * stride - pad_x, pic - pad_x, (size_t)(width + 2 * pad_x) * 2);
		memcpy(pic + (size_t)(height + y) * stride - pad_x, pic + (size_t)(height - 1) * stride - pad_x, (size_t)(width + 2 * pad_x) * 2);
	}
}

/* ====================================================================================================
 * Motion: compensation (a17), and the integer + sub-pel motion search driver (a15/a16).
 * ==================================================================================================== */

/* hmr_motion_compensation_luma, hmr_motion_inter.c:1779-1812.  `ref` points at the co-located block (mv = 0). */
void ora_mc_luma(const int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int width, int height, int mvx, int mvy, int is_bi)
{
	int xf = mvx & 3, yf = mvy & 3;
	const int16_t *src = ref + (mvy >> 2) * ref_stride + (mvx >> 2);
	if (xf == 0) ora_interpolate_luma(src, ref_stride, pred, pred_stride, yf, width, height, 1, 1, !is_bi);
	else if (yf == 0) ora_interpolate_luma(src, ref_stride, pred, pred_stride, xf, width, height, 0, 1, !is_bi);
	else {
		int16_t tmp[(64 + 8) * 80];
		ora_interpolate_luma(src - 3 * ref_stride, ref_stride, tmp, 80, xf, width, height + 7, 0, 1, 0);
		ora_interpolate_luma(tmp + 3 * 80, 80, pred, pred_stride, yf, width, height, 1, 0, !is_bi);
	}
}

/* hmr_motion_compensation_chroma, hmr_motion_inter.c:1860-1907 (eighth-sample vectors) */
void ora_mc_chroma(const int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int size, int mvx, int mvy, int is_bi)
{
	int xf = mvx & 7, yf = mvy & 7;
	const int16_t *src = ref + (mvy >> 3) * ref_stride + (mvx >> 3);
	if (xf == 0) ora_interpolate_chroma(src, ref_stride, pred, pred_stride, yf, size, size, 1, 1, !is_bi);
	else if (yf == 0) ora_interpolate_chroma(src, ref_stride, pred, pred_stride, xf, size, size, 0, 1, !is_bi);
	else {
		int16_t tmp[(32 + 8) * 40];
		ora_interpolate_chroma(src - ref_stride, ref_stride, tmp, 40, xf, size, size + 3, 0, 1, 0);
		ora_interpolate_chroma(tmp + 40, 40, pred, pred_stride, yf, size, size, 1, 0, !is_bi);
	}
}

/* select_mv_candidate_fast, hmr_motion_inter.c:1004-1031; `corr` = calc_mv_correction(qp, avg_dist) (hmr_common.h:53) is a
 * host-side double.  cands = (x, y) pairs in quarter samples. */
static uint32_t mv_cost(const int32_t *cands, int n, double corr, int mvx, int mvy)
{
	uint32_t best = INT_MAX;
	int i;
	for (i = 0; i < n; i++) {
		double cx = corr * ((float)abs(cands[2 * i] - mvx)), cy = corr * ((float)abs(cands[2 * i + 1] - mvy));
		uint32_t c = (uint32_t)(cx + cy + .5);
		if (best > c) best = c;
	}
	return best;
}

/* SAD of the source block against the prediction at quarter-sample vector (qx, qy): the planes the reference builds
 * with hmr_half/quarter_pixel_estimation_luma_hm (hmr_motion_inter.c:395,442) hold exactly these samples. */
static uint32_t subpel_sad(const int16_t *orig, int orig_stride, const int16_t *ref, int ref_stride, int size, int qx, int qy)
{
	int16_t pred[64 * 64];
	ora_mc_luma(ref, ref_stride, pred, 64, size, size, qx, qy, 0);
	return ora_sad(orig, (uint32_t)orig_stride, pred, 64, size);
}

/* hmr_motion_estimation, hmr_motion_inter.c:1404-1775.  out = {mv.x, mv.y, subpix.x, subpix.y}; returns best SAD.
 * action bits: 1 integer search, 2 half-sample, 4 quarter-sample refinement. */
uint32_t ora_motion_estimation(const int16_t *orig, int orig_stride, const int16_t *ref, int ref_stride, int gx, int gy, int init_x, int init_y,
			       int size, int range_x, int range_y, int frame_w, int frame_h, const int32_t *amvp, int n_amvp,
			       const int32_t *search, int n_search, double corr, int action, int32_t *out)
{
	static const int ds[4][2] = {{-1, 0}, {0, -1}, {1, 0}, {0, 1}};
	static const int db[8][2] = {{-2, 0}, {-1, -1}, {0, -2}, {1, -1}, {2, 0}, {1, 1}, {0, 2}, {-1, 1}};
	static const int ref_h[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
	static const int ref_q[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}};
	int xlow = (gx - range_x) < 0 ? -gx : -range_x, xhigh = (gx + range_x) > (frame_w - size) ? frame_w - gx - size : range_x;
	int ylow = (gy - range_y) < 0 ? -gy : -range_y, yhigh = (gy + range_y) > (frame_h - size) ? frame_h - gy - size : range_y;
	uint32_t cur_sad = 0, cur_rd = 0, best_sad = 0xffffffffu;
	int cur_x = 0, cur_y = 0, best_x = 0, best_y = 0, mvx = 0, mvy = 0, subx = 0, suby = 0, i, dist, end, next_start, search_size;
#define IN_WIN(x, y) ((x) >= xlow && (x) <= xhigh && (y) >= ylow && (y) <= yhigh)
#define SAD_AT(x, y) ora_sad(orig, (uint32_t)orig_stride, ref + (y) * ref_stride + (x), (uint32_t)ref_stride, size)
#define TRY(x, y, on_better)                                                          \
	do {                                                                          \
		if (IN_WIN(x, y)) {                                                   \
			uint32_t s_ = SAD_AT(x, y);                                   \
			uint32_t rd_ = s_ + mv_cost(amvp, n_amvp, corr, (x) << 2, (y) << 2); \
			if (rd_ < cur_rd) { on_better; cur_sad = s_; cur_rd = rd_; cur_x = (x); cur_y = (y); } \
		}                                                                     \
	} while (0)
	if (action & 1) {
		cur_x = clip3(init_x, xlow, xhigh);
		cur_y = clip3(init_y, ylow, yhigh);
		cur_sad = SAD_AT(cur_x, cur_y);
		cur_rd = cur_sad + mv_cost(amvp, n_amvp, corr, cur_x << 2, cur_y << 2);
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		if (best_sad <= 0) goto last_search;
		for (i = 0; i < n_search; i++) {
			int x = search[2 * i] >> 2, y = search[2 * i + 1] >> 2;
			if (x == 0 && y == 0) continue;
			TRY(x, y, (void)0);
		}
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		if (best_sad <= 0) goto last_search;
		for (i = 0; i < 4; i++) {
			int x = best_x + ds[i][0], y = best_y + ds[i][1];
			TRY(x, y, (void)0);
		}
		if (best_sad <= 0) goto last_search;
		dist = 2;
		end = (best_x != 0 && best_y != 0) ? 4 : 8;
		next_start = 0; search_size = 8;
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		while (dist < end) {
			for (i = next_start; i < next_start + search_size; i++) {
				int idx = i % 8, x = best_x + db[idx][0] * dist, y = best_y + db[idx][1] * dist;
				TRY(x, y, (next_start = (idx - 2 + 8) % 8, search_size = 5));
			}
			dist *= 2;
		}
	last_search:
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		next_start = 0; search_size = 4;   /* (the reference also tracks a runner-up here; it is never read) */
		for (;;) {
			for (i = next_start; i < next_start + search_size; i++) {
				int idx = i % 4, x = best_x + ds[idx][0], y = best_y + ds[idx][1];
				TRY(x, y, (next_start = (idx - 1 + 4) % 4, search_size = 3));
			}
			if (best_x == cur_x && best_y == cur_y) break;
			best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		}
		best_sad = cur_sad; best_x = cur_x; best_y = cur_y;
		mvx = best_x << 2; mvy = best_y << 2;
	} else {
		mvx = init_x << 2; mvy = init_y << 2;   /* caller-supplied integer vector */
	}
	if (action & 2) {
		int bidx = 0, bx, by;
		best_x = mvx >> 2; best_y = mvy >> 2;
		if (!(action & 1)) cur_sad = SAD_AT(best_x, best_y);
		bx = 0; by = 0;
		for (i = 0; i < 9; i++) {
			int cx = ref_h[i][0] * 2, cy = ref_h[i][1] * 2;
			uint32_t s = subpel_sad(orig, orig_stride, ref, ref_stride, size, (best_x << 2) + cx, (best_y << 2) + cy);
			if (s < cur_sad) { cur_sad = s; bx = cx; by = cy; bidx = i; }
		}
		mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
		best_sad = cur_sad;
		if (action & 4) {
			int hx = ref_h[bidx][0], hy = ref_h[bidx][1];
			bx = hx * 2; by = hy * 2;
			for (i = 0; i < 9; i++) {
				int cx = hx * 2 + ref_q[i][0], cy = hy * 2 + ref_q[i][1];
				uint32_t s = subpel_sad(orig, orig_stride, ref, ref_stride, size, (best_x << 2) + cx, (best_y << 2) + cy);
				if (s < cur_sad) { cur_sad = s; bx = cx; by = cy; }
			}
			best_sad = cur_sad;
			mvx = (best_x << 2) + bx; mvy = (best_y << 2) + by; subx = bx; suby = by;
		}
	}
	out[0] = mvx; out[1] = mvy; out[2] = subx; out[3] = suby;
	return best_sad;
#undef IN_WIN
#undef SAD_AT
#undef TRY
}

/* ====================================================================================================
 * TU chain: the per-TU call sequence of encode_intra_cu (hmr_motion_intra.c:1030-1068) / encode_inter_cu
 * (hmr_motion_inter.c:40-230) as one function - the composition the fused GPU kernel must reproduce.
 * ==================================================================================================== */
uint32_t ora_tu_chain(const int16_t *orig, int orig_stride, const int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride,
		      int size, int is_dst, int scan_mode, int comp, int is_intra, int slice_is_intra, int sign_hiding, int per, int rem, int *ac_sum)
{
	int16_t res[64 * 64], coef[32 * 32], deq[32 * 32], zero_row[64];
	int depth = 6 - ilog2(size) - (comp != 0);
	memset(zero_row, 0, sizeof zero_row);
	ora_predict(orig, orig_stride, pred, pred_stride, res, 64, size);
	ora_transform(res, coef, 64, size, is_dst);
	ora_quant(coef, levels, NULL, scan_mode, depth, comp, is_intra, slice_is_intra, sign_hiding, ac_sum, size, per, rem);
	if (*ac_sum) {
		ora_inv_quant(levels, deq, depth, comp, is_intra, size, per, rem);
		ora_itransform(res, deq, 64, size, is_dst);
		ora_reconst(pred, pred_stride, res, 64, recon, recon_stride, size);
	} else
		ora_reconst(pred, pred_stride, zero_row, 0, recon, recon_stride, size);   /* "quant buff is full of zeros", :1065 */
	return ora_ssd16b(orig, (uint32_t)orig_stride, recon, (uint32_t)recon_stride, size);
}

/* ====================================================================================================
 * Intra mode search of one PU: homer_loop1_motion_intra (hmr_motion_intra.c:1084-1179).  The reference builds the raw and
 * the smoothed neighbour arrays once, then walks four rounds of candidate modes (search_points, :1076-1080): planar/DC,
 * five coarse angles, +-4/+-2 around the best so far, +-1 around that; each candidate is predicted, its SAD against the
 * source taken, and cost = SAD + bits * sqrt_lambda compared with strict <.  The most-probable-mode list and the bit
 * counts are host-side inputs here: `preds` (3 entries, -1 = none) with `pred_bits` for a candidate equal to one of
 * them, `other_bits` for any other candidate (RD_FAST: 1 / 12, RD_FULL: CABAC estimate / 6, RD_DIST_ONLY: 0 / 0).
 * Leaves behind what the reference leaves behind: both neighbour arrays and, in `pred`, the prediction of the LAST
 * candidate evaluated (not the best one).  out[0] = best mode, out[1] = bit count of the best, *best_cost its cost.
 * ==================================================================================================== */
void ora_intra_search(const int16_t *orig, int orig_stride, const int16_t *decoded_corner, int decoded_stride, int n, int left, int top, int bottom_left,
		      int top_right, int bl_size, int tr_size, int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits,
		      double sqrt_lambda, int16_t *adi, int16_t *adi_filtered, int16_t *pred, int pred_stride, int32_t *out, double *best_cost_out)
{
	static const int points[4][5] = {{0, 1, 0, 8, 16}, {2, 10, 16, 22, 30}, {-4, -2, 2, 4, 0}, {-1, 1, 0, 0, 0}};   /* :1076 */
	static const int num_points[4] = {2, 5, 4, 2};                                                                /* :1080 */
	static const int filter_thr[5] = {10, 7, 1, 0, 10};                                                           /* intra_filter, :148 */
	const int adi_size = 4 * n + 1, l2 = ilog2(n);
	int best = 0, new_best = 0, best_bits = 0, min_mode = 0, max_mode = 1, loop, k;
	double best_cost = (double)(UINT32_MAX / 8);   /* MAX_COST, hmr_private.h:54 */
	ora_fill_reference_samples(decoded_corner, decoded_stride, n, left, top, bottom_left, top_right, bl_size, tr_size, adi);
	ora_adi_filter(adi, adi_filtered, adi_size, n, strong_enabled);
	for (loop = 0; loop < 4; loop++) {
		unsigned bits = 0;
		if (loop == 1) {
			best = 2;
			min_mode = 2;
			max_mode = 34;
		}
		for (k = 0; k < num_points[loop]; k++) {
			const int mode = best + points[loop][k];
			int diff, filtered;
			double cost;
			if (mode < min_mode || mode > max_mode) continue;
			diff = abs(mode - 10) < abs(mode - 26) ? abs(mode - 10) : abs(mode - 26);
			filtered = mode != 1 && diff > filter_thr[l2 - 2];
			if (mode == 0) ora_intra_planar(pred, pred_stride, filtered ? adi_filtered : adi, adi_size, n);
			else ora_intra_angular(pred, pred_stride, filtered ? adi_filtered : adi, adi_size, n, mode, 1);
			cost = (double)ora_sad(orig, (uint32_t)orig_stride, pred, (uint32_t)pred_stride, n);
			if (preds[0] == mode) bits = (unsigned)pred_bits[0];
			else if (preds[1] == mode) bits = (unsigned)pred_bits[1];
			else if (preds[2] == mode) bits = (unsigned)pred_bits[2];
			else bits = (unsigned)other_bits;
			cost += bits * sqrt_lambda;
			if (cost < best_cost) {
				best_cost = cost;
				new_best = mode;
				best_bits = (int)bits;
			}
		}
		best = new_best;
	}
	out[0] = best;
	out[1] = best_bits;
	*best_cost_out = best_cost;
}

/* ====================================================================================================
 * Intra TU chain: the whole of encode_intra_cu's data path (hmr_motion_intra.c:1011-1068) - neighbour array (smoothed when the
 * host's is_filtered rule :1011-1012 says so), planar / DC / angular prediction into `pred`, then the TU chain of ora_tu_chain.
 * `recon` may be the block inside the plane the neighbours were read from (the reference reconstructs in place).
 * ==================================================================================================== */
uint32_t ora_intra_tu_chain(const int16_t *orig, int orig_stride, const int16_t *decoded_corner, int decoded_stride, int left, int top, int bottom_left,
			    int top_right, int bl_size, int tr_size, int strong_enabled, int is_filtered, int mode, int is_luma, int16_t *pred, int pred_stride,
			    int16_t *levels, int16_t *recon, int recon_stride, int size, int is_dst, int scan_mode, int comp, int slice_is_intra, int sign_hiding,
			    int per, int rem, int *ac_sum)
{
	int16_t adi[4 * 64 + 1], adif[4 * 64 + 1];
	const int adi_size = 4 * size + 1;
	ora_fill_reference_samples(decoded_corner, decoded_stride, size, left, top, bottom_left, top_right, bl_size, tr_size, adi);
	if (is_filtered) ora_adi_filter(adi, adif, adi_size, size, strong_enabled);
	if (mode == 0) ora_intra_planar(pred, pred_stride, is_filtered ? adif : adi, adi_size, size);
	else ora_intra_angular(pred, pred_stride, is_filtered ? adif : adi, adi_size, size, mode, is_luma);
	return ora_tu_chain(orig, orig_stride, pred, pred_stride, levels, recon, recon_stride, size, is_dst, scan_mode, comp, 1, slice_is_intra, sign_hiding, per, rem,
			    ac_sum);
}

/* ====================================================================================================
 * Side-info layout (a3): ctu_info_t keeps its per-unit arrays in z-order inside each CTU (hmr_private.h:792-843); the in-loop
 * kernels take raster arrays over the picture.  abs2raster_table (hmr_encoder_lib.c:95-100) is the Morton de-interleave:
 * x from the even bits, y from the odd bits of the z-order index.
 * ==================================================================================================== */
int ora_zscan_to_raster(int a)
{
	int x = 0, y = 0, b;
	for (b = 0; b < 4; b++) {
		x |= ((a >> (2 * b)) & 1) << b;
		y |= ((a >> (2 * b + 1)) & 1) << b;
	}
	return y * 16 + x;
}
/* src arrays: ctus * 256 entries each (CTU-major, z-order inside); dst arrays: raster, units_stride per row */
void ora_units_from_ctus(const int16_t *mvx, const int16_t *mvy, const int8_t *ref_idx, const uint8_t *qp, const uint8_t *pred_mode, const uint8_t *cbf_y,
			 const uint8_t *pred_depth, const uint8_t *tr_idx, int ctus_x, int ctus_y, int units_stride, int16_t *o_mvx, int16_t *o_mvy,
			 int8_t *o_ref, uint8_t *o_qp, uint8_t *o_flags, uint8_t *o_pred_depth, uint8_t *o_tr_idx)
{
	int c, a;
	for (c = 0; c < ctus_x * ctus_y; c++)
		for (a = 0; a < 256; a++) {
			const int r = ora_zscan_to_raster(a), s = c * 256 + a;
			const size_t o = (size_t)((c / ctus_x) * 16 + r / 16) * units_stride + (c % ctus_x) * 16 + r % 16;
			o_mvx[o] = mvx[s]; o_mvy[o] = mvy[s]; o_ref[o] = ref_idx[s]; o_qp[o] = qp[s];
			o_flags[o] = (uint8_t)((pred_mode[s] == 1 ? ORA_UNIT_INTRA : 0) | (((cbf_y[s] >> tr_idx[s]) & 1) ? 2 : 0));   /* INTRA_MODE = 1; CBF() hmr_common.h:73 */
			o_pred_depth[o] = pred_depth[s]; o_tr_idx[o] = tr_idx[s];
		}
}

/* ====================================================================================================
 * Inter TU chain: encode_inter_cu / encode_inter_cu_chroma (hmr_motion_inter.c:40-131, 133-230).  The residual of the CU is already
 * there (predict ran on the whole CU); per TU: DCT, quantisation as non-intra, and for a coded TU the reference weighs keeping the
 * levels against dropping them: ssd_zero = SSD(residual, 0), ssd = SSD(residual, de-quantised residual), both scaled by the chroma
 * weight and truncated to uint32; the levels are zeroed when ssd_zero <= ssd + zero_thr * sum (doubles), zero_thr being the host's
 * clip(avg_dist / 2.5 - 5, 1, 20000).  Returns ssd (of the coded candidate even when it was dropped, as the reference does).
 * ==================================================================================================== */
uint32_t ora_inter_tu_chain(const int16_t *residual, int residual_stride, const int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride,
			    int size, int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem, double weight, double zero_thr, int *ac_sum)
{
	int16_t coef[32 * 32], deq[32 * 32], res_dec[32 * 32], zero_row[64];
	const int depth = 6 - ilog2(size) - (comp != 0);
	uint32_t ssd;
	memset(zero_row, 0, sizeof zero_row);
	ora_transform(residual, coef, residual_stride, size, 0);
	ora_quant(coef, levels, NULL, scan_mode, depth, comp, 0, slice_is_intra, sign_hiding, ac_sum, size, per, rem);
	if (*ac_sum > 0) {
		const uint32_t ssd_zero = (uint32_t)(weight * ora_ssd16b(residual, (uint32_t)residual_stride, zero_row, 0, size));
		ora_inv_quant(levels, deq, depth, comp, 0, size, per, rem);
		ora_itransform(res_dec, deq, size, size, 0);
		ssd = (uint32_t)(weight * ora_ssd16b(residual, (uint32_t)residual_stride, res_dec, (uint32_t)size, size));
		if ((double)ssd_zero <= (comp == 0 ? (double)(int)ssd : (double)ssd) + zero_thr * *ac_sum) {   /* luma keeps ssd in an int (:42), chroma in a uint32 (:135) */
			memset(levels, 0, (size_t)size * size * sizeof levels[0]);
			*ac_sum = 0;
			ora_reconst(pred, pred_stride, zero_row, 0, recon, recon_stride, size);
		} else
			ora_reconst(pred, pred_stride, res_dec, size, recon, recon_stride, size);
	} else {
		ssd = (uint32_t)(weight * ora_ssd16b(residual, (uint32_t)residual_stride, zero_row, 0, size));
		ora_reconst(pred, pred_stride, zero_row, 0, recon, recon_stride, size);
	}
	return ssd;
}


/* ====================================================================================================
 * Intra luma transform tree of one 2Nx2N CU, one level deep (the walk of encode_intra_luma, hmr_motion_intra.c:1441-1566, for the default
 * max_intra_tr_depth = 2 and rd_mode != RD_FULL): the CU is coded as one TU of `size` in the parent level's windows, then as four TUs of
 * size/2 in z-order in the child level's windows (each child predicted from its own level's plane, which already holds its siblings'
 * reconstruction), all with the same prediction mode; smoothing (:1011-1012), scan (find_scan_mode, hmr_tables.c:376) and DST follow from
 * mode and TU size.  Then the consolidation (:1479-1557): distortion / sum of the four children against the parent's,
 *     rule 0 (RD_DIST_ONLY)  dist < parent dist
 *     rule 1 (RD_FAST)       1.25 * (dist + 45 * sum) < parent dist + 45 * parent sum        (uint32 products, double comparison, :1496)
 * children win  -> their levels and reconstruction are copied up (synchronize_motion_buffers_luma, :866), cbf of child k = nz_k << 1 | any nz,
 *                  tr_idx 1 (:1510-1522);
 * parent wins   -> the parent's bottom row and right column go down into the child plane (synchronize_reference_buffs, :844), cbf = nz, tr_idx 0.
 * A 64x64 CU has no parent TU (its cost is preset to INT_MAX, :1402): size 64 codes the four 32x32 TUs and always consolidates them.
 * nb: 5 x {left, top, bottom_left, top_right, bl_size, tr_size} for parent, child 0..3 (cu_partition_info_t's neighbour fields + picture bounds).
 * dec_par / dec_chl: the CU's first sample in decoded_mbs_wnd[depth + 1] / [depth + 2]; lev_par / lev_chl: size*size linear each, child k at k*(size/2)^2
 * (abs_index order).  out: {split, cost, distortion, sum, cbf[0..3], tr_idx, ssd[5], sum[5]} (19 words).
 * ==================================================================================================== */
int ora_intra_is_filtered(int mode, int size)
{
	static const int filter_thr[5] = {10, 7, 1, 0, 10};                                                           /* intra_filter, :148 */
	const int diff = abs(mode - 10) < abs(mode - 26) ? abs(mode - 10) : abs(mode - 26);
	return mode != 1 && diff > filter_thr[ilog2(size) - 2];
}
int ora_intra_scan_mode(int mode, int size)     /* find_scan_mode(is_intra, is_luma, ...), hmr_tables.c:398-402: 1 horizontal, 2 vertical, 3 diagonal (hmr_private.h:91-94) */
{
	if (size != 4 && size != 8) return 3;
	return abs(mode - 26) < 5 ? 1 : abs(mode - 10) < 5 ? 2 : 3;
}
void ora_intra_cu_tree(const int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
		       int strong_enabled, int mode, int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra,
		       int sign_hiding, int per, int rem, int rule, int32_t *out)
{
	const int h = size / 2;
	uint32_t ssd[5] = {0, 0, 0, 0, 0}, dist, sum, par_cost, par_sum;
	int ac[5] = {0, 0, 0, 0, 0}, k, y, split;
	if (size <= 32)
		ssd[0] = ora_intra_tu_chain(orig, orig_stride, dec_par - dec_par_stride - 1, dec_par_stride, nb[0], nb[1], nb[2], nb[3], nb[4], nb[5], strong_enabled,
					    ora_intra_is_filtered(mode, size), mode, 1, pred, pred_stride, lev_par, dec_par, dec_par_stride, size, size == 4,
					    ora_intra_scan_mode(mode, size), 0, slice_is_intra, sign_hiding, per, rem, &ac[0]);
	for (k = 0; k < 4; k++) {
		const int x0 = (k & 1) * h, y0 = (k >> 1) * h;
		const int32_t *n = nb + 6 * (k + 1);
		int16_t *d = dec_chl + y0 * dec_chl_stride + x0;
		ssd[k + 1] = ora_intra_tu_chain(orig + y0 * orig_stride + x0, orig_stride, d - dec_chl_stride - 1, dec_chl_stride, n[0], n[1], n[2], n[3], n[4], n[5],
						strong_enabled, ora_intra_is_filtered(mode, h), mode, 1, pred + y0 * pred_stride + x0, pred_stride, lev_chl + k * h * h, d,
						dec_chl_stride, h, h == 4, ora_intra_scan_mode(mode, h), 0, slice_is_intra, sign_hiding, per, rem, &ac[k + 1]);
	}
	dist = ssd[1] + ssd[2] + ssd[3] + ssd[4];
	sum = (uint32_t)ac[1] + (uint32_t)ac[2] + (uint32_t)ac[3] + (uint32_t)ac[4];
	par_cost = size <= 32 ? ssd[0] : (uint32_t)INT32_MAX;
	par_sum = (uint32_t)ac[0];
	if (size > 32) split = 1;
	else if (rule == 1) split = 1.25 * ((double)dist + 45 * sum) < (double)(uint32_t)(par_cost + 45 * par_sum);
	else split = (double)dist < (double)par_cost;
	if (split) {
		const int any = (ac[1] || ac[2] || ac[3] || ac[4]) ? 1 : 0;
		for (y = 0; y < size; y++) memcpy(dec_par + y * dec_par_stride, dec_chl + y * dec_chl_stride, (size_t)size * sizeof dec_par[0]);
		memcpy(lev_par, lev_chl, (size_t)size * size * sizeof lev_par[0]);
		for (k = 0; k < 4; k++) out[4 + k] = ((ac[k + 1] ? 1 : 0) << 1) | any;
	} else {
		memcpy(dec_chl + (size - 1) * dec_chl_stride, dec_par + (size - 1) * dec_par_stride, (size_t)size * sizeof dec_par[0]);
		for (y = 0; y < size - 1; y++) dec_chl[y * dec_chl_stride + size - 1] = dec_par[y * dec_par_stride + size - 1];
		for (k = 0; k < 4; k++) out[4 + k] = ac[0] ? 1 : 0;
	}
	out[0] = split;
	out[1] = out[2] = (int32_t)(split ? dist : par_cost);
	out[3] = (int32_t)(split ? sum : par_sum);
	out[8] = split;
	for (k = 0; k < 5; k++) { out[9 + k] = (int32_t)ssd[k]; out[14 + k] = ac[k]; }
	if (split) { out[9] = (int32_t)dist; out[14] = (int32_t)sum; }     /* the parent node carries the consolidated figures from here on (:1499-1501) */
}

/* encode_intra_luma's data path for one 2Nx2N CU (hmr_motion_intra.c:1226-1632): the mode search on the parent level's plane (ora_intra_search), then
 * the transform tree above with the mode it found.  out as ora_intra_cu_tree, then out[19] = mode, out[20] = its bit count; *best_cost its search cost. */
void ora_intra_luma_cu(const int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
		       int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits, double sqrt_lambda, int16_t *adi, int16_t *adi_filtered,
		       int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra, int sign_hiding, int per, int rem, int rule,
		       int32_t *out, double *best_cost)
{
	int32_t found[2];
	ora_intra_search(orig, orig_stride, dec_par - dec_par_stride - 1, dec_par_stride, size, nb[0], nb[1], nb[2], nb[3], nb[4], nb[5], strong_enabled, preds, pred_bits,
			 other_bits, sqrt_lambda, adi, adi_filtered, pred, pred_stride, found, best_cost);
	ora_intra_cu_tree(orig, orig_stride, dec_par, dec_par_stride, dec_chl, dec_chl_stride, nb, strong_enabled, found[0], pred, pred_stride, lev_par, lev_chl, size,
			  slice_is_intra, sign_hiding, per, rem, rule, out);
	out[19] = found[0];
	out[20] = found[1];
}

/* ====================================================================================================
 * Chroma half of an intra CU: encode_intra_chroma (hmr_motion_intra_chroma.c:114-471, the non-HM path, rd_mode != RD_FULL).
 *  1. mode list (create_chroma_dir_list, :92): planar, vertical, horizontal, DC, DM (= the luma mode); a list entry equal to the luma mode becomes 34.
 *  2. search (:176-233): every candidate is predicted for U and V from UNFILTERED neighbours of the CU (of its first quadrant only for a 64x64 CU) (is_luma = 0: no edge filters) and compared by SAD;
 *     cost = dU + (dU + dV) (the running distortion is added once per component, :211-213) + (uint32)(bits * sqrt_lambda + .5), bits = 1 for DM, 12 otherwise;
 *     the three best are kept by homer_update_cand_list (hmr_motion_intra.c:893, strict >) and only the best is coded.
 *  3. TUs (:262-352) following the luma transform tree: split = 0 -> one TU of `size` per component, split = 1 -> four TUs of size / 2 in z-order (a 4x4
 *     chroma CU is always one TU: 2x2 TUs do not exist, :289-293); per TU and component: neighbours from the plane under reconstruction, prediction, DCT,
 *     quantisation as intra, reconstruction in place; distortion += (int)(weight * SSD).  Scan: find_scan_mode(intra, chroma): mode dependent for 4x4 only.
 * size: chroma CU size (4 ... 32).  nb: 5 x {left, top, bottom_left, top_right, bl_size, tr_size} in chroma samples: the CU, then its four quadrants.
 * lev_u / lev_v: size * size each, TU k at k * (size / 2)^2 when split.
 * out: [0] coded chroma mode (36 = DM), [1] prediction mode used, [2] bits, [3] search cost of the winner, [4] distortion, [5] sum of the TUs' ac sums,
 *      [6..9] ac sum of the U TUs, [10..13] of the V TUs (slot 0 only when not split).
 * ==================================================================================================== */
void ora_intra_chroma_cu(const int16_t *orig_u, const int16_t *orig_v, int orig_stride, int16_t *dec_u, int16_t *dec_v, int dec_stride, const int32_t *nb, int luma_mode,
			 int split, double sqrt_lambda, double weight, int16_t *pred_u, int16_t *pred_v, int pred_stride, int16_t *lev_u, int16_t *lev_v, int size,
			 int slice_is_intra, int sign_hiding, int per, int rem, int32_t *out)
{
	int list[5] = {0, 26, 10, 1, 36}, cand_mode[3] = {0, 0, 0}, i, c, k;
	double cand_cost[3] = {1.7e+308, 1.7e+308, 1.7e+308};
	unsigned cand_bits[3] = {0, 0, 0};
	const int16_t *orig[2] = {orig_u, orig_v};
	int16_t *dec[2] = {dec_u, dec_v}, *pred[2] = {pred_u, pred_v}, *lev[2] = {lev_u, lev_v};
	int16_t adi[4 * 64 + 1];
	const int do_split = split && size > 4, n = do_split ? size / 2 : size, ntu = do_split ? 4 : 1;
	/* a 64x64 CU (chroma 32) is walked as its four 32x32 children from the start (:165-169): the SEARCH sees only the first of them */
	const int ss = size == 32 ? 16 : size;
	const int32_t *snb = size == 32 ? nb + 6 : nb;
	uint32_t distortion = 0, sum = 0;
	int mode;
	for (i = 0; i < 4; i++)
		if (luma_mode == list[i]) { list[i] = 34; break; }
	for (i = 0; i < 5; i++) {
		const int m = list[i] == 36 ? luma_mode : list[i];
		uint32_t dist = 0, cost = 0;
		unsigned bits = list[i] == 36 ? 1 : 12;
		double dcost;
		int mm = list[i];
		for (c = 0; c < 2; c++) {
			ora_fill_reference_samples(dec[c] - dec_stride - 1, dec_stride, ss, snb[0], snb[1], snb[2], snb[3], snb[4], snb[5], adi);
			if (m == 0) ora_intra_planar(pred[c], pred_stride, adi, 4 * ss + 1, ss);
			else ora_intra_angular(pred[c], pred_stride, adi, 4 * ss + 1, ss, m, 0);
			dist += ora_sad(orig[c], (uint32_t)orig_stride, pred[c], (uint32_t)pred_stride, ss);
			cost += dist;
		}
		cost += (uint32_t)(bits * sqrt_lambda + .5);
		dcost = (double)cost;
		for (k = 0; k < 3; k++)                     /* homer_update_cand_list */
			if (cand_cost[k] > dcost) {
				const int am = cand_mode[k]; const double ac = cand_cost[k]; const unsigned ab = cand_bits[k];
				cand_cost[k] = dcost; cand_mode[k] = mm; cand_bits[k] = bits;
				dcost = ac; mm = am; bits = ab;
			}
	}
	mode = cand_mode[0] == 36 ? luma_mode : cand_mode[0];
	for (k = 0; k < 8; k++) out[6 + k] = 0;
	for (k = 0; k < ntu; k++) {
		const int x0 = do_split ? (k & 1) * n : 0, y0 = do_split ? (k >> 1) * n : 0;
		const int32_t *f = nb + (do_split ? 6 * (k + 1) : 0);
		const int scan = n == 4 ? (abs(mode - 26) < 5 ? 1 : abs(mode - 10) < 5 ? 2 : 3) : 3;     /* find_scan_mode, hmr_tables.c:403-411 */
		int part = 0;
		for (c = 0; c < 2; c++) {
			int16_t *d = dec[c] + y0 * dec_stride + x0;
			int ac = 0;
			const uint32_t ssd = ora_intra_tu_chain(orig[c] + y0 * orig_stride + x0, orig_stride, d - dec_stride - 1, dec_stride, f[0], f[1], f[2], f[3], f[4], f[5], 0, 0,
								mode, 0, pred[c] + y0 * pred_stride + x0, pred_stride, lev[c] + k * n * n, d, dec_stride, n, 0, scan, c + 1,
								slice_is_intra, sign_hiding, per, rem, &ac);
			sum += (uint32_t)ac;
			out[6 + 4 * c + k] = ac;
			part += (int)(weight * ssd);
		}
		distortion += (uint32_t)part;
	}
	out[0] = cand_mode[0];
	out[1] = mode;
	out[2] = (int32_t)cand_bits[0];
	out[3] = (int32_t)(uint32_t)cand_cost[0];
	out[4] = (int32_t)distortion;
	out[5] = (int32_t)sum;
}

/* ====================================================================================================
 * SAO offset derivation of one CTU (8-bit): sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion (hmr_sao.c:480-659) for the five types of
 * the three components - the part of the SAO decision (sao_derive_mode_new_rdo, :663) that is a pure function of the statistics; the rate terms and the
 * merge / off comparison need the CABAC state and stay on the host.
 *   initial offset  = round-half-away(diff / count) in double (x_round_ibdi, :432), clipped to +-7; edge classes keep only the sign their shape allows (:522-525)
 *   refinement      = est_iter_offset (:445): walk the offset towards 0, keep the one with the least dist + lambda * rate (double), dist = count*o*o - 2*diff*o,
 *                     rate = |o| + 1 (edge) / |o| + 2 (band), one less at |o| = 7; an offset that never beats lambda alone becomes 0
 *   band position   = first minimum of the sum of four consecutive band costs (:555-567), the other 28 bands are cleared
 * stats: [3][5][2][32] int32 (diff, count) as ora_sao_stats_frame leaves them; lambdas[3]; offsets [3][5][32], aux [3][5] (band position, 0 for edge types),
 * dist [3][5] int64 = sao_get_distortion of the derived offsets.
 * ==================================================================================================== */
static int sao_iter_offset(int is_bo, double lambda, int offset_input, int64_t count, int64_t diff, int64_t *best_dist, double *best_cost)
{
	int iter = offset_input, out = 0;
	double min_cost = lambda;
	while (iter != 0) {
		int64_t rate = is_bo ? abs(iter) + 2 : abs(iter) + 1, d;
		double cost;
		if (abs(iter) == 7) rate--;
		d = count * iter * iter - diff * iter * 2;
		cost = (double)d + lambda * (double)rate;
		if (cost < min_cost) {
			min_cost = cost;
			out = iter;
			*best_dist = d;
			*best_cost = cost;
		}
		iter = iter > 0 ? iter - 1 : iter + 1;
	}
	return out;
}
void ora_sao_offsets_ctu(const int32_t *stats, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist)
{
	int comp, type, c, i;
	for (comp = 0; comp < 3; comp++)
		for (type = 0; type < 5; type++) {
			const int32_t *df = stats + ((comp * 5 + type) * 2) * 32, *cn = df + 32;
			int32_t *q = offsets + (comp * 5 + type) * 32;
			const int is_bo = type == 4, n = is_bo ? 32 : 5;
			const double lambda = lambdas[comp];
			int64_t d = 0;
			memset(q, 0, 32 * sizeof q[0]);
			for (c = 0; c < n; c++) {
				double x;
				int v;
				if ((!is_bo && c == 2) || cn[c] == 0) continue;
				x = (double)(int64_t)df[c] / (double)(int64_t)cn[c];
				v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
				q[c] = v < -7 ? -7 : v > 7 ? 7 : v;
			}
			if (!is_bo) {
				for (c = 0; c < 5; c++) {
					int64_t cd;
					double cc;
					if (c < 2 && q[c] < 0) q[c] = 0;
					if (c > 2 && q[c] > 0) q[c] = 0;
					if (q[c] != 0) q[c] = sao_iter_offset(0, lambda, q[c], cn[c], df[c], &cd, &cc);
				}
				aux[comp * 5 + type] = 0;
				for (c = 0; c < 5; c++) d += (int64_t)cn[c] * q[c] * q[c] - (int64_t)df[c] * q[c] * 2;
			} else {
				double cost[32], min_cost = (double)(UINT32_MAX / 8);
				int64_t dd[32];
				int band = 0, keep[32];
				for (c = 0; c < 32; c++) {
					cost[c] = lambda;
					dd[c] = 0;
					if (q[c] != 0) q[c] = sao_iter_offset(1, lambda, q[c], cn[c], df[c], &dd[c], &cost[c]);
				}
				for (i = 0; i < 29; i++) {
					double s = cost[i];
					s += cost[i + 1]; s += cost[i + 2]; s += cost[i + 3];
					if (s < min_cost) { min_cost = s; band = i; }
				}
				memset(keep, 0, sizeof keep);
				for (i = 0; i < 4; i++) keep[(band + i) % 32] = q[(band + i) % 32];
				for (c = 0; c < 32; c++) q[c] = keep[c];
				aux[comp * 5 + type] = band;
				for (i = band; i < band + 4; i++) d += (int64_t)cn[i % 32] * q[i % 32] * q[i % 32] - (int64_t)df[i % 32] * q[i % 32] * 2;
			}
			dist[comp * 5 + type] = d;
		}
}
