// Host-side construction of the constant tables the kernels index: HEVC core transform matrices,
// coefficient scan orders and the default-scaling-list quantiser pyramids.  The reference builds the
// same tables in HOMER_enc_init (hmr_encoder_lib.c:93-140) with init_scan_pyramid (hmr_tables.c:62)
// and init_quant_pyramids (hmr_tables.c:221); tests compare them entry by entry.
#include <stdarg.h>
#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";
void hmr_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}
extern "C" const char *hmr_gpu_last_error(void) { return g_err; }

namespace {

// integer samples of 64*sqrt(2)*cos(m*pi/64) fixed by the HEVC standard, m = 0..32
const int16_t kCos[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
			  61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
// HEVC default scaling lists (spec Table 7-6; hmr_tables.h:53-82)
const int16_t kIntra8[64] = {16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29, 16, 16, 18, 21, 24, 27, 31, 36,
			     17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115};
const int16_t kInter8[64] = {16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28, 16, 17, 18, 20, 24, 25, 28, 33,
			     17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91};
const int kQuantScale[6] = {26214, 23302, 20560, 18396, 16384, 14564};
const int kInvQuantScale[6] = {40, 45, 51, 57, 64, 72};

int basis(int m)
{
	m &= 127;
	if (m > 64) m = 128 - m;
	return m <= 32 ? kCos[m] : -kCos[64 - m];
}

// anti-diagonal walk, bottom-left to top-right, of a w x w grid; entries are row*pitch + col + base
void up_right(uint32_t *out, int w, int pitch, uint32_t base)
{
	int k = 0;
	for (int d = 0; d < 2 * w - 1; d++)
		for (int row = (d < w ? d : w - 1), col = d - row; row >= 0 && col < w; row--, col++)
			out[k++] = (uint32_t)(row * pitch + col) + base;
}

DevTables *g_host = nullptr;
std::once_flag g_once;

void build()
{
	DevTables *t = new DevTables;
	memset(t, 0, sizeof *t);
	for (int l = 2; l <= 5; l++) {
		int n = 1 << l, step = 32 >> l;
		for (int k = 0; k < n; k++)
			for (int x = 0; x < n; x++) t->dct[l - 2][k * n + x] = (int16_t)basis(k * step * (2 * x + 1));
	}
	const int16_t dst[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
	memcpy(t->dst4, dst, sizeof dst);
	for (int l = 2; l <= 5; l++) {
		const int n = 1 << l;
		for (int k = 0; k < n; k++)
			for (int x = 0; x < n; x++) t->dct_t[l - 2][x * n + k] = t->dct[l - 2][k * n + x];
	}
	for (int k = 0; k < 4; k++)
		for (int x = 0; x < 4; x++) t->dst4_t[x * 4 + k] = dst[k * 4 + x];
	// MFMA operand fragments of the bases (tables_layout.h)
	{
		auto half_bits = [](int v) -> uint16_t {      // binary16 of an integer, |v| < 2048: exact
			if (v == 0) return 0;
			const uint16_t sign = v < 0 ? 0x8000 : 0;
			unsigned a = (unsigned)(v < 0 ? -v : v);
			int e = 0;
			while ((a >> (e + 1)) != 0) e++;              // a = 1.xxx * 2^e
			const unsigned mant = (a << (10 - e)) & 0x3ff;
			return (uint16_t)(sign | ((unsigned)(e + 15) << 10) | mant);
		};
		for (int dir = 0; dir < 2; dir++) {
			for (int b = 0; b < 4; b++) {
				const int n = b == 3 ? 4 : 4 << b;
				const int16_t *M = b == 3 ? (dir ? t->dst4_t : t->dst4) : (dir ? t->dct_t[b] : t->dct[b]);
				for (int lane = 0; lane < 64; lane++)
					for (int e = 0; e < 4; e++) {
						const int row = lane % 16, col = 4 * (lane / 16) + e;
						t->frag16[dir][b][lane * 4 + e] = row < n && col < n ? half_bits(M[row * n + col]) : 0;
					}
			}
			for (int b = 0; b < 2; b++) {
				const int n = 4 << b;
				const int16_t *M = dir ? t->dct_t[b] : t->dct[b];
				for (int lane = 0; lane < 64; lane++)
					for (int e = 0; e < 4; e++) {
						const int row = lane % 8, col = 4 * (lane / 16 % 2) + e;
						t->fragp[dir][b][lane * 4 + e] = (lane % 16 / 8 == lane / 32 && row < n && col < n) ? half_bits(M[row * n + col]) : 0;
					}
			}
			{
				const int16_t *M = dir ? t->dct_t[0] : t->dct[0];
				for (int lane = 0; lane < 64; lane++)
					for (int e = 0; e < 4; e++)
						t->fragq[dir][lane * 4 + e] = (lane % 16 / 4 == lane / 16) ? half_bits(M[(lane % 4) * 4 + e]) : 0;
			}
			for (int R = 0; R < 2; R++)
				for (int K = 0; K < 2; K++)
					for (int lane = 0; lane < 64; lane++)
						for (int e = 0; e < 4; e++)
							t->frag32t[dir][R][K][lane * 4 + e] = half_bits((dir ? t->dct_t[3] : t->dct[3])[(16 * R + lane % 16) * 32 + 16 * K + 4 * (lane / 16) + e]);
		}
	}
	// even / odd split of the inverse DCT: out[k] = E[k] + O[k], out[N-1-k] = E[k] - O[k] with E over the even, O over the odd input indices
	for (int l = 3; l <= 5; l++) {
		const int n = 1 << l;
		for (int k = 0; k < n / 2; k++)
			for (int i = 0; i < n / 4; i++) {
				int16_t *row = t->dct_eo[l - 2] + k * n;
				row[2 * i] = t->dct[l - 2][(4 * i) * n + k];
				row[2 * i + 1] = t->dct[l - 2][(4 * i + 2) * n + k];
				row[n / 2 + 2 * i] = t->dct[l - 2][(4 * i + 1) * n + k];
				row[n / 2 + 2 * i + 1] = t->dct[l - 2][(4 * i + 3) * n + k];
			}
	}

	// scans: sizes 2..32.  Diagonal: 4x4 coefficient groups visited up-right, each scanned up-right.
	uint32_t cg_order[64];
	for (int l = 1; l <= 5; l++) {
		int w = 1 << l;
		uint32_t *H = t->scan[1][l], *V = t->scan[2][l], *D = t->scan[3][l];
		if (w <= 4) {
			up_right(D, w, w, 0);
		} else {
			int side = w >> 2;
			up_right(cg_order, side, side, 0);
			for (int b = 0; b < side * side; b++) {
				int gy = cg_order[b] / side, gx = cg_order[b] % side;
				up_right(D + 16 * b, 4, w, (uint32_t)(4 * (gy * w + gx)));
			}
		}
		if (w == 2) {
			for (int i = 0; i < 4; i++) H[i] = i;
			V[0] = 0; V[1] = 2; V[2] = 1; V[3] = 3;
		} else {
			int side = w >> 2, k = 0;
			for (int gy = 0; gy < side; gy++)
				for (int gx = 0; gx < side; gx++)
					for (int y = 0; y < 4; y++)
						for (int x = 0; x < 4; x++) H[k++] = (gy * 4 + y) * w + gx * 4 + x;
			k = 0;
			for (int gx = 0; gx < side; gx++)
				for (int gy = 0; gy < side; gy++)
					for (int x = 0; x < 4; x++)
						for (int y = 0; y < 4; y++) V[k++] = (gy * 4 + y) * w + gx * 4 + x;
		}
	}
	// inverse of the scans at coefficient-group granularity (every scan visits whole 4x4 groups)
	for (int mode = 1; mode <= 3; mode++)
		for (int l = 2; l <= 5; l++) {
			const int w = 1 << l, side = w >> 2;
			for (int cg = 0; cg < side * side; cg++) {
				const uint32_t pos = t->scan[mode][l][cg * 16];
				t->blk2cg[mode][l][((pos / w) >> 2) * side + ((pos % w) >> 2)] = (uint8_t)cg;
			}
		}
	// quantiser pyramids: 8x8 lists are up-sampled for 16/32, DC forced to 16 when up-sampled
	for (int l = 2; l <= 5; l++) {
		int n = 1 << l, ms = n < 8 ? n : 8, ratio = n / ms;
		for (int list = 0; list < 6; list++) {
			const int16_t *sl = nullptr;
			if (l == 5) sl = (list == 0) ? kIntra8 : kInter8;   // 32x32 has lists {0: intra, 1/3: inter}
			else if (l > 2) sl = list < 3 ? kIntra8 : kInter8;
			for (int rem = 0; rem < 6; rem++) {
				int32_t *q = t->quant[l - 2][list][rem], *iq = t->dequant[l - 2][list][rem];
				for (int y = 0; y < n; y++)
					for (int x = 0; x < n; x++) {
						int m = sl ? sl[ms * (y / ratio) + x / ratio] : 16;
						q[y * n + x] = (kQuantScale[rem] << 4) / m;
						iq[y * n + x] = kInvQuantScale[rem] * m;
					}
				if (ratio > 1) {
					q[0] = (kQuantScale[rem] << 4) / 16;
					iq[0] = kInvQuantScale[rem] * 16;
				}
			}
		}
	}
	g_host = t;
}

}  // namespace

const DevTables *hmr_host_tables()
{
	std::call_once(g_once, build);
	return g_host;
}

extern "C" int hmr_gpu_get_scan_table(int scan_mode, int log2_size, uint32_t *out)
{
	if (scan_mode < 1 || scan_mode > 3 || log2_size < 1 || log2_size > 5 || !out) return HMR_GPU_ERR_ARG;
	memcpy(out, hmr_host_tables()->scan[scan_mode][log2_size], sizeof(uint32_t) << (2 * log2_size));
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_get_quant_table(int log2_size, int list, int rem, int32_t *quant, int32_t *dequant)
{
	if (log2_size < 2 || log2_size > 5 || list < 0 || list > 5 || rem < 0 || rem > 5) return HMR_GPU_ERR_ARG;
	size_t bytes = sizeof(int32_t) << (2 * log2_size);
	if (quant) memcpy(quant, hmr_host_tables()->quant[log2_size - 2][list][rem], bytes);
	if (dequant) memcpy(dequant, hmr_host_tables()->dequant[log2_size - 2][list][rem], bytes);
	return HMR_GPU_OK;
}
