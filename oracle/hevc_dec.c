/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * A small HEVC decoder written from ITU-T H.265 (04/2013), for the decoder-side check of SURVEY.md 8-f.4: the image holds no decoder
 * (no ffmpeg, libde265 or HM) and the reference ships none, so byte identity with the reference's streams was the only argument for
 * conformance.  This file decodes the subset of the standard the reference encoder emits - Main profile, 8 bit 4:2:0, one slice per
 * picture, I and P slices with one reference picture, CTB 64, CABAC, wavefront entry points, sign data hiding, default scaling lists,
 * cu_qp_delta, merge / AMVP without temporal candidates, deblocking, SAO - and REFUSES everything else (exit code 2) rather than guess.
 * It shares no code with homerhevc_amd/ or with the other files of oracle/: the parsing process, the tables and the reconstruction
 * are restated from the clauses cited below.
 *
 * What it checks while decoding (exit code 3 with a message when violated):
 *   - every CABAC sub-stream ends on a terminating bin followed by the stop bit and zero bits up to the byte boundary (9.3.2.5), and
 *     the next sub-stream starts exactly where the slice header's entry point says (7.4.7.1);
 *   - end_of_slice_segment_flag is 1 exactly after the picture's last CTU and the slice data end with it;
 *   - syntax element ranges (intra modes, merge index, cu_qp_delta, coefficient positions, QP range).
 * The decoded pictures are written as 8-bit planar YUV; tests/test_decoder_check.py compares them with the reconstruction the
 * compiled reference encoder itself dumped (ref_lockstep recon=...) - for the checker build in tests/test_stream_cpu.py, for the device in tests/test_gpu_stream.py.
 *
 * What it found in the reference encoder: two cases in which the encoder's own reconstruction is NOT what a decoder
 * reconstructs from its stream - (R1) under rate control the encoder deblocks coding units whose QP the stream does not carry (those of a CTU
 * that precede its first coded cu_qp_delta, or all of a CTU without one) with its rate-control QP or with the predicted one, depending on
 * how far its lagged filter / entropy pipeline has got; a decoder (8.6.1) always uses the predicted QP; (R2) quirk Q12, a merge candidate far
 * outside the picture predicted from a stale window.  `--ref-deblock-qp` deblocks every coding unit of a CTU that has a coded delta with
 * that CTU's QP: where the stream lets a decoder know the encoder's QP this reproduces the reference (tests/test_decoder_check.py).
 *
 * usage: hevcdec in.265 out.yuv|- [-v] [--ref-deblock-qp]          prints "DECODED pictures=N ..." on success
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FAIL(code, ...) do { fprintf(stderr, "hevcdec: " __VA_ARGS__); fprintf(stderr, "\n"); exit(code); } while (0)
#define UNSUPPORTED(...) FAIL(2, "unsupported: " __VA_ARGS__)
#define VIOLATION(...) FAIL(3, "violation: " __VA_ARGS__)

static int verbose, ref_deblock_qp;
static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int sgn(int a) { return (a > 0) - (a < 0); }

/* ---- bit reader over an RBSP (7.2) ---------------------------------------------------------------------------------- */
typedef struct {
	const uint8_t *d;
	int n;      /* bytes */
	long pos;   /* bit position */
} BR;
static int br_bit(BR *b)
{
	if (b->pos >= (long)b->n * 8) VIOLATION("read past the end of the NAL unit payload");
	const int v = (b->d[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
	b->pos++;
	return v;
}
static unsigned br_u(BR *b, int n) { unsigned v = 0; while (n-- > 0) v = (v << 1) | (unsigned)br_bit(b); return v; }
static unsigned br_ue(BR *b)      /* 9.2 */
{
	int z = 0;
	while (!br_bit(b)) if (++z > 31) VIOLATION("ue(v) prefix longer than 31 bits");
	return (1u << z) - 1 + br_u(b, z);
}
static int br_se(BR *b) { const unsigned k = br_ue(b); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }

/* ---- parameter sets (7.3.2) ------------------------------------------------------------------------------------------ */
typedef struct {
	int valid, w, h, ctb_log2, min_cb_log2, min_tb_log2, max_tb_log2, max_th_inter, max_th_intra;
	int scaling_list, amp, sao, strong_intra, log2_max_poc, tmvp, num_rps;
	struct { int nneg, dpoc[16], used[16]; } rps[16];
	int wctb, hctb, w4, h4;
} SPS;
typedef struct {
	int valid, sign_hiding, cabac_init_present, num_ref_l0, init_qp, cu_qp_delta, diff_cu_qp_delta_depth, cb_off, cr_off;
	int wpp, lf_across_slices, par_mrg_level, dependent_slices, output_flag_present, num_extra_bits, lists_mod, sh_ext;
} PPS;
static SPS sps;
static PPS pps;

static void parse_ptl(BR *b, int max_sub_layers_minus1)
{
	br_u(b, 2); br_u(b, 1); br_u(b, 5);
	br_u(b, 32);
	br_u(b, 4);
	br_u(b, 32); br_u(b, 11);      /* 43 reserved bits */
	br_u(b, 1);
	br_u(b, 8);                    /* general_level_idc */
	if (max_sub_layers_minus1) UNSUPPORTED("sub-layers");
}
static void parse_sps(BR *b)
{
	memset(&sps, 0, sizeof sps);
	br_u(b, 4);
	const int msl = (int)br_u(b, 3);
	br_u(b, 1);
	parse_ptl(b, msl);
	if (br_ue(b) != 0) UNSUPPORTED("sps id != 0");
	if (br_ue(b) != 1) UNSUPPORTED("chroma_format_idc != 1");
	sps.w = (int)br_ue(b); sps.h = (int)br_ue(b);
	if (br_u(b, 1)) { if (br_ue(b) | br_ue(b) | br_ue(b) | br_ue(b)) UNSUPPORTED("conformance window offsets"); }
	if (br_ue(b) != 0 || br_ue(b) != 0) UNSUPPORTED("bit depth != 8");
	sps.log2_max_poc = (int)br_ue(b) + 4;
	const int sub_info = (int)br_u(b, 1);
	for (int i = sub_info ? 0 : msl; i <= msl; i++) { br_ue(b); br_ue(b); br_ue(b); }
	sps.min_cb_log2 = (int)br_ue(b) + 3;
	sps.ctb_log2 = sps.min_cb_log2 + (int)br_ue(b);
	sps.min_tb_log2 = (int)br_ue(b) + 2;
	sps.max_tb_log2 = sps.min_tb_log2 + (int)br_ue(b);
	sps.max_th_inter = (int)br_ue(b);
	sps.max_th_intra = (int)br_ue(b);
	sps.scaling_list = (int)br_u(b, 1);
	if (sps.scaling_list && br_u(b, 1)) UNSUPPORTED("scaling list data in the SPS");
	sps.amp = (int)br_u(b, 1);
	sps.sao = (int)br_u(b, 1);
	if (br_u(b, 1)) UNSUPPORTED("pcm");
	sps.num_rps = (int)br_ue(b);
	if (sps.num_rps > 16) UNSUPPORTED("more than 16 short-term reference picture sets");
	for (int i = 0; i < sps.num_rps; i++) {      /* 7.3.7 */
		if (i > 0 && br_u(b, 1)) UNSUPPORTED("inter RPS prediction");
		sps.rps[i].nneg = (int)br_ue(b);
		if (br_ue(b)) UNSUPPORTED("positive pictures in an RPS");
		if (sps.rps[i].nneg > 16) VIOLATION("num_negative_pics");
		int poc = 0;
		for (int j = 0; j < sps.rps[i].nneg; j++) {
			poc -= (int)br_ue(b) + 1;
			sps.rps[i].dpoc[j] = poc;
			sps.rps[i].used[j] = (int)br_u(b, 1);
		}
	}
	if (br_u(b, 1)) UNSUPPORTED("long-term reference pictures");
	sps.tmvp = (int)br_u(b, 1);
	if (sps.tmvp) UNSUPPORTED("temporal motion vector prediction");
	sps.strong_intra = (int)br_u(b, 1);
	if (br_u(b, 1)) UNSUPPORTED("VUI");
	if (br_u(b, 1)) UNSUPPORTED("SPS extension");
	if (sps.ctb_log2 < 4 || sps.ctb_log2 > 6 || sps.min_cb_log2 != 3 || sps.min_tb_log2 != 2 || sps.max_tb_log2 > 5 || sps.max_tb_log2 > sps.ctb_log2)
		UNSUPPORTED("block size limits (CTB %d, min CB %d, TB %d..%d)", sps.ctb_log2, sps.min_cb_log2, sps.min_tb_log2, sps.max_tb_log2);
	if (sps.w % 8 || sps.h % 8) VIOLATION("picture size not a multiple of the minimum coding block");
	sps.wctb = (sps.w + (1 << sps.ctb_log2) - 1) >> sps.ctb_log2;
	sps.hctb = (sps.h + (1 << sps.ctb_log2) - 1) >> sps.ctb_log2;
	sps.w4 = sps.w / 4; sps.h4 = sps.h / 4;
	sps.valid = 1;
}
static void parse_pps(BR *b)
{
	memset(&pps, 0, sizeof pps);
	if (br_ue(b) != 0 || br_ue(b) != 0) UNSUPPORTED("pps / sps id != 0");
	pps.dependent_slices = (int)br_u(b, 1);
	pps.output_flag_present = (int)br_u(b, 1);
	pps.num_extra_bits = (int)br_u(b, 3);
	pps.sign_hiding = (int)br_u(b, 1);
	pps.cabac_init_present = (int)br_u(b, 1);
	pps.num_ref_l0 = (int)br_ue(b) + 1; br_ue(b);
	pps.init_qp = 26 + br_se(b);
	if (br_u(b, 1)) UNSUPPORTED("constrained intra prediction");
	if (br_u(b, 1)) UNSUPPORTED("transform skip");
	pps.cu_qp_delta = (int)br_u(b, 1);
	if (pps.cu_qp_delta) pps.diff_cu_qp_delta_depth = (int)br_ue(b);
	pps.cb_off = br_se(b); pps.cr_off = br_se(b);
	if (br_u(b, 1)) UNSUPPORTED("slice chroma qp offsets");
	if (br_u(b, 1) | br_u(b, 1)) UNSUPPORTED("weighted prediction");
	if (br_u(b, 1)) UNSUPPORTED("transquant bypass");
	if (br_u(b, 1)) UNSUPPORTED("tiles");
	pps.wpp = (int)br_u(b, 1);
	pps.lf_across_slices = (int)br_u(b, 1);
	if (br_u(b, 1)) UNSUPPORTED("deblocking filter control");
	if (br_u(b, 1)) UNSUPPORTED("scaling list data in the PPS");
	pps.lists_mod = (int)br_u(b, 1);
	pps.par_mrg_level = (int)br_ue(b) + 2;
	pps.sh_ext = (int)br_u(b, 1);
	if (br_u(b, 1)) UNSUPPORTED("PPS extension");
	if (pps.dependent_slices || pps.output_flag_present || pps.num_extra_bits || pps.lists_mod || pps.sh_ext || pps.num_ref_l0 != 1 || pps.par_mrg_level != 2 || pps.diff_cu_qp_delta_depth != 0)
		UNSUPPORTED("a PPS tool the reference encoder does not use");
	pps.valid = 1;
}

/* ---- pictures and per-unit side information -------------------------------------------------------------------------- */
typedef struct {
	uint8_t *pl[3];
	int w[3], h[3];
} Pic;
static Pic cur, dbk_out, ref;
static int have_ref;
enum { PM_NONE = 0, PM_INTER = 1, PM_INTRA = 2 };
/* per 4x4 luma unit */
static uint8_t *u_pm, *u_skip, *u_depth, *u_imode, *u_tuedge_v, *u_tuedge_h, *u_puedge_v, *u_puedge_h, *u_nz;
static int8_t *u_qp;
static int16_t *u_mvx, *u_mvy;
static int pic_alloc_w, pic_alloc_h;

static void pic_alloc(Pic *p)
{
	for (int c = 0; c < 3; c++) {
		p->w[c] = c ? sps.w / 2 : sps.w; p->h[c] = c ? sps.h / 2 : sps.h;
		free(p->pl[c]);
		p->pl[c] = (uint8_t *)calloc((size_t)p->w[c] * p->h[c], 1);
	}
}
static void alloc_all(void)
{
	if (pic_alloc_w == sps.w && pic_alloc_h == sps.h) return;
	pic_alloc(&cur); pic_alloc(&dbk_out); pic_alloc(&ref);
	const size_t n = (size_t)sps.w4 * sps.h4;
#define RE(p, T) do { free(p); p = (T *)calloc(n, sizeof(T)); } while (0)
	RE(u_pm, uint8_t); RE(u_skip, uint8_t); RE(u_depth, uint8_t); RE(u_imode, uint8_t); RE(u_tuedge_v, uint8_t); RE(u_tuedge_h, uint8_t);
	RE(u_puedge_v, uint8_t); RE(u_puedge_h, uint8_t); RE(u_nz, uint8_t); RE(u_qp, int8_t); RE(u_mvx, int16_t); RE(u_mvy, int16_t);
	pic_alloc_w = sps.w; pic_alloc_h = sps.h; have_ref = 0;
}
#define U(x, y) ((size_t)((y) >> 2) * sps.w4 + ((x) >> 2))

/* z-scan order address of the 4x4 unit holding luma sample (x, y) (6.5.2): CTBs in raster order, units inside a CTB in z order */
static unsigned zs_addr(int x, int y)
{
	const int cl = sps.ctb_log2;
	const unsigned ctb = (unsigned)((y >> cl) * sps.wctb + (x >> cl));
	const unsigned ux = (unsigned)((x & ((1 << cl) - 1)) >> 2), uy = (unsigned)((y & ((1 << cl) - 1)) >> 2);
	unsigned z = 0;
	for (int b = 0; b < 4; b++) z |= ((ux >> b) & 1u) << (2 * b) | ((uy >> b) & 1u) << (2 * b + 1);
	return (ctb << 8) | z;
}
/* 6.4.1: the unit at (xn, yn) is available to the block at (xc, yc) when it lies in the picture and precedes it in decoding order (one slice, no tiles) */
static int avail_z(int xc, int yc, int xn, int yn)
{
	if (xn < 0 || yn < 0 || xn >= sps.w || yn >= sps.h) return 0;
	return zs_addr(xn, yn) <= zs_addr(xc, yc);
}

/* ---- CABAC (9.3) ------------------------------------------------------------------------------------------------------ */
static const uint8_t range_lps[64][4] = {      /* Table 9-46 */
	{128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166},
	{95, 116, 137, 158}, {90, 110, 130, 150}, {85, 104, 123, 142}, {81, 99, 117, 135}, {77, 94, 111, 128}, {73, 89, 105, 122}, {69, 85, 100, 116}, {66, 80, 95, 110},
	{62, 76, 90, 104}, {59, 72, 86, 99}, {56, 69, 81, 94}, {53, 65, 77, 89}, {51, 62, 73, 85}, {48, 59, 69, 80}, {46, 56, 66, 76}, {43, 53, 63, 72},
	{41, 50, 59, 69}, {39, 48, 56, 65}, {37, 45, 54, 62}, {35, 43, 51, 59}, {33, 41, 48, 56}, {32, 39, 46, 53}, {30, 37, 43, 50}, {29, 35, 41, 48},
	{27, 33, 39, 45}, {26, 31, 37, 43}, {24, 30, 35, 41}, {23, 28, 33, 39}, {22, 27, 32, 37}, {21, 26, 30, 35}, {20, 24, 29, 33}, {19, 23, 27, 31},
	{18, 22, 26, 30}, {17, 21, 25, 28}, {16, 20, 23, 27}, {15, 19, 22, 25}, {14, 18, 21, 24}, {14, 17, 20, 23}, {13, 16, 19, 22}, {12, 15, 18, 21},
	{12, 14, 17, 20}, {11, 14, 16, 19}, {11, 13, 15, 18}, {10, 12, 15, 17}, {10, 12, 14, 16}, {9, 11, 13, 15}, {9, 11, 12, 14}, {8, 10, 12, 14},
	{8, 9, 11, 13}, {7, 9, 11, 12}, {7, 9, 10, 12}, {7, 8, 10, 11}, {6, 8, 9, 11}, {6, 7, 9, 10}, {6, 7, 8, 9}, {2, 2, 2, 2}};
static const uint8_t trans_lps[64] = {      /* Table 9-47 */
	0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
	24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

enum {      /* context variables, one block per syntax element (Tables 9-5 ... 9-37) */
	C_SAO_MERGE = 0, C_SAO_TYPE = 1, C_SPLIT_CU = 2, C_SKIP = 5, C_MERGE_FLAG = 8, C_MERGE_IDX = 9, C_PART_MODE = 10, C_PRED_MODE = 14, C_PREV_INTRA = 15,
	C_CHROMA_PRED = 16, C_RQT_ROOT = 17, C_MVD_G0 = 18, C_MVD_G1 = 19, C_REF_IDX = 20, C_MVP = 22, C_SPLIT_TR = 23, C_CBF_LUMA = 26, C_CBF_C = 28, C_DQP = 32,
	C_LAST_X = 34, C_LAST_Y = 52, C_CSBF = 70, C_SIG = 74, C_G1 = 116, C_G2 = 140, C_TOTAL = 146
};
/* initValue per initType: 0 = I slices, 1 = P slices (cabac_init_flag 0), 2 = B slices.  255 = the element does not occur in that slice type */
typedef struct { int first, count; uint8_t v[3][42]; } CtxInit;
static const CtxInit ctx_init[] = {
	{C_SAO_MERGE, 1, {{153}, {153}, {153}}},
	{C_SAO_TYPE, 1, {{200}, {185}, {160}}},
	{C_SPLIT_CU, 3, {{139, 141, 157}, {107, 139, 126}, {107, 139, 126}}},
	{C_SKIP, 3, {{255, 255, 255}, {197, 185, 201}, {197, 185, 201}}},
	{C_MERGE_FLAG, 1, {{255}, {110}, {154}}},
	{C_MERGE_IDX, 1, {{255}, {122}, {137}}},
	{C_PART_MODE, 4, {{184, 255, 255, 255}, {154, 139, 154, 154}, {154, 139, 154, 154}}},
	{C_PRED_MODE, 1, {{255}, {149}, {134}}},
	{C_PREV_INTRA, 1, {{184}, {154}, {183}}},
	{C_CHROMA_PRED, 1, {{63}, {152}, {152}}},
	{C_RQT_ROOT, 1, {{255}, {79}, {79}}},
	{C_MVD_G0, 1, {{255}, {140}, {169}}},
	{C_MVD_G1, 1, {{255}, {198}, {198}}},
	{C_REF_IDX, 2, {{255, 255}, {153, 153}, {153, 153}}},
	{C_MVP, 1, {{255}, {168}, {168}}},
	{C_SPLIT_TR, 3, {{153, 138, 138}, {124, 138, 94}, {224, 167, 122}}},
	{C_CBF_LUMA, 2, {{111, 141}, {153, 111}, {153, 111}}},
	{C_CBF_C, 4, {{94, 138, 182, 154}, {149, 107, 167, 154}, {149, 92, 167, 154}}},
	{C_DQP, 2, {{154, 154}, {154, 154}, {154, 154}}},
	{C_LAST_X, 18, {{110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63},
			{125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108},
			{125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93}}},
	{C_LAST_Y, 18, {{110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63},
			{125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108},
			{125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93}}},
	{C_CSBF, 4, {{91, 171, 134, 141}, {121, 140, 61, 154}, {121, 140, 61, 154}}},
	{C_SIG, 42, {{111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
		      140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111},
		     {155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
		      170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140},
		     {170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
		      170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140}}},
	{C_G1, 24, {{140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197},
		    {154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182},
		    {154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182}}},
	{C_G2, 6, {{138, 153, 136, 167, 152, 152}, {107, 167, 91, 122, 107, 167}, {107, 167, 91, 107, 107, 167}}},
};

typedef struct {
	BR *b;
	unsigned range, offset;
	uint8_t st[C_TOTAL];      /* pStateIdx << 1 | valMps */
} Cabac;
static Cabac cab;
static uint8_t wpp_saved[C_TOTAL];
static int wpp_saved_valid;

static void ctx_init_all(int init_type, int slice_qp)      /* 9.3.2.2 */
{
	const int q = clip3(0, 51, slice_qp);
	for (size_t e = 0; e < sizeof ctx_init / sizeof ctx_init[0]; e++)
		for (int i = 0; i < ctx_init[e].count; i++) {
			int v = ctx_init[e].v[init_type][i];
			if (v == 255) v = 154;      /* never read in this slice type */
			const int slope = v >> 4, off = v & 15, m = slope * 5 - 45, n = (off << 3) - 16;
			const int pre = clip3(1, 126, ((m * q) >> 4) + n);
			const int mps = pre <= 63 ? 0 : 1;
			cab.st[ctx_init[e].first + i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
		}
}
static void cabac_start(void)      /* 9.3.2.5 */
{
	if (cab.b->pos & 7) VIOLATION("a CABAC sub-stream that does not start on a byte boundary");
	cab.range = 510;
	cab.offset = br_u(cab.b, 9);
	if (cab.offset >= 510) VIOLATION("ivlOffset 510 or 511 at initialisation");
}
static int ae_ctx(int ctx)      /* 9.3.4.3.2 */
{
	const int p = cab.st[ctx] >> 1;
	int mps = cab.st[ctx] & 1, bin, np;
	const unsigned lps = range_lps[p][(cab.range >> 6) & 3];
	cab.range -= lps;
	if (cab.offset >= cab.range) {
		bin = !mps;
		cab.offset -= cab.range;
		cab.range = lps;
		if (p == 0) mps = !mps;
		np = trans_lps[p];
	} else {
		bin = mps;
		np = p < 62 ? p + 1 : p;
	}
	cab.st[ctx] = (uint8_t)((np << 1) | mps);
	while (cab.range < 256) { cab.range <<= 1; cab.offset = (cab.offset << 1) | (unsigned)br_bit(cab.b); }
	return bin;
}
static int ae_bypass(void)      /* 9.3.4.3.4 */
{
	cab.offset = (cab.offset << 1) | (unsigned)br_bit(cab.b);
	if (cab.offset >= cab.range) { cab.offset -= cab.range; return 1; }
	return 0;
}
static unsigned ae_bypass_bits(int n) { unsigned v = 0; while (n-- > 0) v = (v << 1) | (unsigned)ae_bypass(); return v; }
static int ae_terminate(void)      /* 9.3.4.3.5 */
{
	cab.range -= 2;
	if (cab.offset >= cab.range) return 1;
	while (cab.range < 256) { cab.range <<= 1; cab.offset = (cab.offset << 1) | (unsigned)br_bit(cab.b); }
	return 0;
}
/* After a terminating bin equal to 1 the arithmetic decoder has consumed the encoder's whole flush (9.3.4.3.5 with the encoder's 9.3.5.x flush: seven renormalisation
 * bits, then put(low >> 9), then two bits of which the last is 1).  That last bit is the stop / alignment bit: check it, and that zero bits follow up to the byte
 * boundary. */
static void check_substream_end(const char *what)
{
	BR *b = cab.b;
	if (b->pos < 1 || !((b->d[(b->pos - 1) >> 3] >> (7 - ((b->pos - 1) & 7))) & 1)) VIOLATION("%s: the bit that ends the CABAC data is not 1", what);
	while (b->pos & 7) if (br_bit(b)) VIOLATION("%s: non-zero alignment bit", what);
}

/* ---- slice state -------------------------------------------------------------------------------------------------------- */
typedef struct {
	int type;      /* 0 B, 1 P, 2 I (slice_type) */
	int poc, qp, sao_luma, sao_chroma, max_merge, idr;
	int n_entry;
	long entry[256];      /* byte offsets (escaped) */
} Slice;
static Slice sl;
static int cu_qp_delta_val, is_cu_qp_delta_coded, qp_pred, last_cu_qp;

/* SAO parameters per CTB */
typedef struct { uint8_t type[3], band[3], eo[3]; int8_t off[3][4]; } Sao;
static Sao *sao_ctb;

/* scans (6.5.3 - 6.5.5): [log2 size 1..3][scanIdx 0 diag, 1 horizontal, 2 vertical][pos][x, y] */
static uint8_t scan_tab[4][3][64][2];
static void build_scans(void)
{
	for (int l = 1; l <= 3; l++) {
		const int n = 1 << l;
		int i = 0, x = 0, y = 0, stop = 0;
		while (!stop) {
			while (y >= 0) {
				if (x < n && y < n) { scan_tab[l][0][i][0] = (uint8_t)x; scan_tab[l][0][i][1] = (uint8_t)y; i++; }
				y--; x++;
			}
			y = x; x = 0;
			if (i >= n * n) stop = 1;
		}
		for (i = 0; i < n * n; i++) {
			scan_tab[l][1][i][0] = (uint8_t)(i % n); scan_tab[l][1][i][1] = (uint8_t)(i / n);
			scan_tab[l][2][i][0] = (uint8_t)(i / n); scan_tab[l][2][i][1] = (uint8_t)(i % n);
		}
	}
}

/* ---- transform and scaling (8.6) ---------------------------------------------------------------------------------------- */
static const int8_t dct_col0[32] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4};
static int dct_mat[32][32];      /* transMatrix (8-xxx): row k, column n of the 32-point transform; smaller sizes take every (32 / N)-th row */
static const int dst_mat[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
static void build_dct(void)
{
	for (int k = 0; k < 32; k++)
		for (int n = 0; n < 32; n++) {
			int a = (k * (2 * n + 1)) % 128;      /* angle in units of pi / 64 */
			if (a > 64) a = 128 - a;
			dct_mat[k][n] = k == 0 ? 64 : (a > 32 ? -dct_col0[64 - a] : dct_col0[a]);
		}
}
static const uint8_t sl_intra8[64] = {16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29, 16, 16, 18, 21, 24, 27, 31, 36,
				      17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115};
static const uint8_t sl_inter8[64] = {16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28, 16, 17, 18, 20, 24, 25, 28, 33,
				      17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91};
/* Table 7-5 / 7-6 with the default lists: flat 16 for 4x4, the 8x8 lists upsampled for 16 and 32 with DC 16 (7.4.5) */
static int scaling_factor(int log2n, int intra, int x, int y)
{
	if (!sps.scaling_list || log2n == 2) return 16;
	const int sh = log2n - 3;
	if (x == 0 && y == 0 && log2n > 3) return 16;
	return (intra ? sl_intra8 : sl_inter8)[(y >> sh) * 8 + (x >> sh)];
}
static const int level_scale[6] = {40, 45, 51, 57, 64, 72};
static const uint8_t qpc_tab[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
static int chroma_qp(int qpy, int off)      /* 8.6.1, ChromaArrayType 1 */
{
	const int qpi = clip3(0, 57, qpy + off);
	return qpi < 30 ? qpi : (qpi >= 44 ? qpi - 6 : qpc_tab[qpi - 30]);
}
/* residual of one transform block from its levels: scaling (8.6.4.2), then the two one-dimensional transforms (8.6.4.2 - 8.6.4.3) */
static void residual_from_levels(const int16_t *lev, int log2n, int qp, int intra, int dst, int *res)
{
	const int n = 1 << log2n, bd_shift = log2n + 3;      /* BitDepth + Log2(nTbS) - 5 */
	static int d[32 * 32], e[32 * 32];
	for (int y = 0; y < n; y++)
		for (int x = 0; x < n; x++) {
			const int l = lev[y * n + x];
			if (!l) { d[y * n + x] = 0; continue; }
			const long long v = ((long long)l * scaling_factor(log2n, intra, x, y) * level_scale[qp % 6] << (qp / 6)) + (1ll << (bd_shift - 1));
			d[y * n + x] = clip3(-32768, 32767, (int)(v >> bd_shift));
		}
	/* first stage: columns */
	for (int x = 0; x < n; x++)
		for (int y = 0; y < n; y++) {
			long long s = 0;
			for (int k = 0; k < n; k++) s += (long long)(dst ? dst_mat[k][y] : dct_mat[k * (32 >> log2n)][y]) * d[k * n + x];
			e[y * n + x] = clip3(-32768, 32767, (int)((s + 64) >> 7));
		}
	/* second stage: rows, bdShift = 20 - BitDepth */
	for (int y = 0; y < n; y++)
		for (int x = 0; x < n; x++) {
			long long s = 0;
			for (int k = 0; k < n; k++) s += (long long)(dst ? dst_mat[k][x] : dct_mat[k * (32 >> log2n)][x]) * e[y * n + k];
			res[y * n + x] = (int)((s + 2048) >> 12);
		}
}

/* ---- intra prediction (8.4.4.2) ------------------------------------------------------------------------------------------ */
static const int8_t intra_angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
static const int16_t inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};      /* modes 11 .. 25 */
static void intra_predict(int c, int x0, int y0, int log2n, int mode)      /* (x0, y0) in samples of component c */
{
	const int n = 1 << log2n, W = cur.w[c];
	uint8_t *pl = cur.pl[c];
	const int sc = c ? 1 : 0;      /* chroma sample -> luma sample: shift */
	int buf_l[2 * 64 + 1], buf_t[2 * 64 + 1];      /* left[i] = p[-1][i - 1], top[i] = p[i - 1][-1]; index 0 is the corner */
	int av_l[2 * 64 + 1], av_t[2 * 64 + 1];
	const int xl = x0 << sc, yl = y0 << sc;
	int any = 0;
	for (int i = 0; i <= 2 * n; i++) {
		const int yy = y0 - 1 + i, xx = x0 - 1;
		av_l[i] = avail_z(xl, yl, xx << sc, yy << sc) && u_pm[U(xx << sc, yy << sc)] != PM_NONE;
		if (av_l[i]) { buf_l[i] = pl[yy * W + xx]; any = 1; }
	}
	for (int i = 1; i <= 2 * n; i++) {
		const int xx = x0 - 1 + i, yy = y0 - 1;
		av_t[i] = avail_z(xl, yl, xx << sc, yy << sc) && u_pm[U(xx << sc, yy << sc)] != PM_NONE;
		if (av_t[i]) { buf_t[i] = pl[yy * W + xx]; any = 1; }
	}
	av_t[0] = av_l[0]; buf_t[0] = buf_l[0];
	if (!any) {
		for (int i = 0; i <= 2 * n; i++) buf_l[i] = buf_t[i] = 128;
	} else {      /* 8.4.4.2.2 */
		if (!av_l[2 * n]) {
			int found = 0, v = 0;
			for (int i = 2 * n - 1; i >= 0 && !found; i--) if (av_l[i]) { v = buf_l[i]; found = 1; }
			for (int i = 1; i <= 2 * n && !found; i++) if (av_t[i]) { v = buf_t[i]; found = 1; }
			buf_l[2 * n] = v;
		}
		for (int i = 2 * n - 1; i >= 0; i--) if (!av_l[i]) buf_l[i] = buf_l[i + 1];
		buf_t[0] = buf_l[0];
		for (int i = 1; i <= 2 * n; i++) if (!av_t[i]) buf_t[i] = buf_t[i - 1];
	}
	/* 8.4.4.2.3: smoothing of the neighbouring samples (luma only for 4:2:0) */
	if (c == 0 && mode != 1 && n != 4) {
		const int min_dist = imin(iabs(mode - 26), iabs(mode - 10));
		const int thres = n == 8 ? 7 : (n == 16 ? 1 : 0);
		if (min_dist > thres) {
			int fl[2 * 64 + 1], ft[2 * 64 + 1];
			const int corner = buf_l[0];
			if (sps.strong_intra && n == 32 && iabs(corner + buf_t[2 * n] - 2 * buf_t[n]) < 8 && iabs(corner + buf_l[2 * n] - 2 * buf_l[n]) < 8) {
				fl[0] = ft[0] = corner;
				for (int i = 1; i < 2 * n; i++) {      /* p[-1][y], y = i - 1 = 0 .. 62 */
					fl[i] = ((64 - i) * corner + i * buf_l[2 * n] + 32) >> 6;
					ft[i] = ((64 - i) * corner + i * buf_t[2 * n] + 32) >> 6;
				}
				fl[2 * n] = buf_l[2 * n]; ft[2 * n] = buf_t[2 * n];
			} else {
				fl[0] = ft[0] = (buf_l[1] + 2 * corner + buf_t[1] + 2) >> 2;
				for (int i = 1; i < 2 * n; i++) {
					fl[i] = (buf_l[i + 1] + 2 * buf_l[i] + buf_l[i - 1] + 2) >> 2;
					ft[i] = (buf_t[i + 1] + 2 * buf_t[i] + buf_t[i - 1] + 2) >> 2;
				}
				fl[2 * n] = buf_l[2 * n]; ft[2 * n] = buf_t[2 * n];
			}
			memcpy(buf_l, fl, sizeof(int) * (2 * n + 1));
			memcpy(buf_t, ft, sizeof(int) * (2 * n + 1));
		}
	}
#define PL(y) buf_l[(y) + 1]      /* p[-1][y] */
#define PT(x) buf_t[(x) + 1]      /* p[x][-1] */
	uint8_t *dstp = pl + y0 * W + x0;
	if (mode == 0) {      /* planar 8.4.4.2.4 */
		for (int y = 0; y < n; y++)
			for (int x = 0; x < n; x++)
				dstp[y * W + x] = (uint8_t)(((n - 1 - x) * PL(y) + (x + 1) * PT(n) + (n - 1 - y) * PT(x) + (y + 1) * PL(n) + n) >> (log2n + 1));
	} else if (mode == 1) {      /* DC 8.4.4.2.5 */
		int s = n;
		for (int i = 0; i < n; i++) s += PT(i) + PL(i);
		const int dc = s >> (log2n + 1);
		for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dstp[y * W + x] = (uint8_t)dc;
		if (c == 0 && n < 32) {
			dstp[0] = (uint8_t)((PL(0) + 2 * dc + PT(0) + 2) >> 2);
			for (int x = 1; x < n; x++) dstp[x] = (uint8_t)((PT(x) + 3 * dc + 2) >> 2);
			for (int y = 1; y < n; y++) dstp[y * W] = (uint8_t)((PL(y) + 3 * dc + 2) >> 2);
		}
	} else {      /* angular 8.4.4.2.6 */
		const int ang = intra_angle[mode];
		int refb[3 * 64 + 2], *rf = refb + 64;
		if (mode >= 18) {
			for (int x = 0; x <= n; x++) rf[x] = PT(x - 1);
			if (ang < 0) {
				const int last = (n * ang) >> 5;
				if (last < -1) for (int x = last; x <= -1; x++) rf[x] = PL(-1 + ((x * inv_angle[mode - 11] + 128) >> 8));
			} else for (int x = n + 1; x <= 2 * n; x++) rf[x] = PT(x - 1);
			for (int y = 0; y < n; y++) {
				const int idx = ((y + 1) * ang) >> 5, fact = ((y + 1) * ang) & 31;
				for (int x = 0; x < n; x++)
					dstp[y * W + x] = (uint8_t)(fact ? ((32 - fact) * rf[x + idx + 1] + fact * rf[x + idx + 2] + 16) >> 5 : rf[x + idx + 1]);
			}
			if (mode == 26 && c == 0 && n < 32)
				for (int y = 0; y < n; y++) dstp[y * W] = (uint8_t)clip8(PT(0) + ((PL(y) - PL(-1)) >> 1));
		} else {
			for (int x = 0; x <= n; x++) rf[x] = PL(x - 1);
			if (ang < 0) {
				const int last = (n * ang) >> 5;
				if (last < -1) for (int x = last; x <= -1; x++) rf[x] = PT(-1 + ((x * inv_angle[mode - 11] + 128) >> 8));
			} else for (int x = n + 1; x <= 2 * n; x++) rf[x] = PL(x - 1);
			for (int x = 0; x < n; x++) {
				const int idx = ((x + 1) * ang) >> 5, fact = ((x + 1) * ang) & 31;
				for (int y = 0; y < n; y++)
					dstp[y * W + x] = (uint8_t)(fact ? ((32 - fact) * rf[y + idx + 1] + fact * rf[y + idx + 2] + 16) >> 5 : rf[y + idx + 1]);
			}
			if (mode == 10 && c == 0 && n < 32)
				for (int x = 0; x < n; x++) dstp[x] = (uint8_t)clip8(PL(0) + ((PT(x) - PT(-1)) >> 1));
		}
	}
#undef PL
#undef PT
}

/* ---- inter prediction (8.5.3.3) ------------------------------------------------------------------------------------------ */
static const int8_t luma_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t chroma_taps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
static inline int ref_sample(int c, int x, int y) { return ref.pl[c][clip3(0, ref.h[c] - 1, y) * ref.w[c] + clip3(0, ref.w[c] - 1, x)]; }
static void inter_predict(int c, int x0, int y0, int w, int h, int mvx, int mvy)      /* block in samples of component c; mv in quarter luma samples */
{
	const int taps = c ? 4 : 8, fx = c ? (mvx & 7) : (mvx & 3), fy = c ? (mvy & 7) : (mvy & 3);
	const int ix = x0 + (c ? mvx >> 3 : mvx >> 2), iy = y0 + (c ? mvy >> 3 : mvy >> 2), before = taps / 2 - 1;
	static int tmp[(64 + 8) * 64];
	uint8_t *dst = cur.pl[c] + y0 * cur.w[c] + x0;
	for (int y = 0; y < h + taps - 1; y++)      /* horizontal stage on rows -before .. h + taps/2 - 1 (shift1 = BitDepth - 8 = 0) */
		for (int x = 0; x < w; x++) {
			int s = 0;
			if (fx == 0) s = ref_sample(c, ix + x, iy + y - before) << 6;
			else for (int k = 0; k < taps; k++) s += (c ? chroma_taps[fx][k] : luma_taps[fx][k]) * ref_sample(c, ix + x + k - before, iy + y - before);
			tmp[y * w + x] = s;
		}
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			int s;
			if (fy == 0) s = fx == 0 ? tmp[(y + before) * w + x] : tmp[(y + before) * w + x];      /* horizontal only: already 14-bit intermediate */
			else if (fx == 0) {
				s = 0;
				for (int k = 0; k < taps; k++) s += (c ? chroma_taps[fy][k] : luma_taps[fy][k]) * (tmp[(y + k) * w + x] >> 6);      /* full samples again, shift1 = 0 */
			} else {
				s = 0;
				for (int k = 0; k < taps; k++) s += (c ? chroma_taps[fy][k] : luma_taps[fy][k]) * tmp[(y + k) * w + x];
				s >>= 6;      /* shift2 */
			}
			dst[y * cur.w[c] + x] = (uint8_t)clip8((s + 32) >> 6);      /* 8.5.3.3.4.2, shift1 = 14 - BitDepth */
		}
}

/* ---- CTU syntax (7.3.8) ---------------------------------------------------------------------------------------------------- */
typedef struct { int log2cb, x, y, pred_mode, part_nxn, skip, intra_y[4], intra_c, max_trafo_depth, intra_split; } CU;
static int16_t lev_buf[3][32 * 32];

static void parse_sao(int rx, int ry)      /* 7.3.8.3 */
{
	Sao *s = &sao_ctb[ry * sps.wctb + rx];
	memset(s, 0, sizeof *s);
	int merge_left = 0, merge_up = 0;
	if (rx > 0) merge_left = ae_ctx(C_SAO_MERGE);
	if (ry > 0 && !merge_left) merge_up = ae_ctx(C_SAO_MERGE);
	if (merge_left) { *s = sao_ctb[ry * sps.wctb + rx - 1]; return; }
	if (merge_up) { *s = sao_ctb[(ry - 1) * sps.wctb + rx]; return; }
	for (int c = 0; c < 3; c++) {
		if (!(c == 0 ? sl.sao_luma : sl.sao_chroma)) continue;
		if (c < 2) {
			int t = 0;
			if (ae_ctx(C_SAO_TYPE)) t = ae_bypass() ? 2 : 1;
			s->type[c] = (uint8_t)t;
		} else s->type[2] = s->type[1];
		if (!s->type[c]) continue;
		int ab[4];
		for (int i = 0; i < 4; i++) { int v = 0; while (v < 7 && ae_bypass()) v++; ab[i] = v; }
		if (s->type[c] == 1) {
			for (int i = 0; i < 4; i++) if (ab[i] && ae_bypass()) ab[i] = -ab[i];
			s->band[c] = (uint8_t)ae_bypass_bits(5);
		} else {
			if (c == 0) s->eo[0] = (uint8_t)ae_bypass_bits(2);
			else if (c == 1) s->eo[1] = (uint8_t)ae_bypass_bits(2);
			else s->eo[2] = s->eo[1];
			ab[2] = -ab[2]; ab[3] = -ab[3];
		}
		for (int i = 0; i < 4; i++) s->off[c][i] = (int8_t)ab[i];
	}
}

/* residual_coding 7.3.8.11, levels into lev (raster, n x n) */
static void parse_residual(int log2n, int c, int scan_idx, int16_t *lev)
{
	const int n = 1 << log2n;
	memset(lev, 0, sizeof(int16_t) * n * n);
	/* last significant coefficient position */
	int off_x, shift_x;
	if (c == 0) { off_x = 3 * (log2n - 2) + ((log2n - 1) >> 2); shift_x = (log2n + 1) >> 2; }
	else { off_x = 15; shift_x = log2n - 2; }
	const int cmax = (log2n << 1) - 1;
	int px = 0, py = 0;
	while (px < cmax && ae_ctx(C_LAST_X + off_x + (px >> shift_x))) px++;
	while (py < cmax && ae_ctx(C_LAST_Y + off_x + (py >> shift_x))) py++;
	int lx = px, ly = py;
	if (px > 3) { const int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + (int)ae_bypass_bits(nb); }
	if (py > 3) { const int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + (int)ae_bypass_bits(nb); }
	if (scan_idx == 2) { const int t = lx; lx = ly; ly = t; }
	if (lx >= n || ly >= n) VIOLATION("last significant coefficient outside the transform block");
	const int l2sb = log2n - 2;      /* sub-block grid */
	const uint8_t (*sb_scan)[2] = scan_tab[l2sb ? l2sb : 1][scan_idx];
	const uint8_t (*pos_scan)[2] = scan_tab[2][scan_idx];
	int last_sb = (1 << (2 * l2sb)) - 1, last_pos = 16;
	for (;;) {
		if (last_pos == 0) { last_pos = 16; last_sb--; if (last_sb < 0) VIOLATION("last position not found in the scan"); }
		last_pos--;
		const int xs = l2sb ? sb_scan[last_sb][0] : 0, ys = l2sb ? sb_scan[last_sb][1] : 0;
		if ((xs << 2) + pos_scan[last_pos][0] == lx && (ys << 2) + pos_scan[last_pos][1] == ly) break;
	}
	uint8_t csbf[8][8];
	memset(csbf, 0, sizeof csbf);
	int c1_carry = 1, first_sb_done = 0;
	for (int i = last_sb; i >= 0; i--) {
		const int xs = l2sb ? sb_scan[i][0] : 0, ys = l2sb ? sb_scan[i][1] : 0;
		const int right = xs + 1 < (1 << l2sb) ? csbf[ys][xs + 1] : 0, below = ys + 1 < (1 << l2sb) ? csbf[ys + 1][xs] : 0;
		int infer_dc = 0, coded;
		if (i < last_sb && i > 0) {
			coded = ae_ctx(C_CSBF + imin(right + below, 1) + (c ? 2 : 0));
			infer_dc = 1;
		} else coded = 1;
		csbf[ys][xs] = (uint8_t)coded;
		uint8_t sig[16];
		memset(sig, 0, sizeof sig);
		const int prev_csbf = right | (below << 1);
		const int n_start = i == last_sb ? last_pos - 1 : 15;
		if (i == last_sb) sig[last_pos] = 1;
		for (int k = n_start; k >= 0; k--) {
			const int xp = pos_scan[k][0], yp = pos_scan[k][1], xc = (xs << 2) + xp, yc = (ys << 2) + yp;
			if (coded && (k > 0 || !infer_dc)) {
				int sc;      /* 9.3.4.2.5 */
				if (log2n == 2) { static const uint8_t map[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; sc = map[(yc << 2) + xc]; }
				else if (xc + yc == 0) sc = 0;
				else {
					if (prev_csbf == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
					else if (prev_csbf == 1) sc = yp == 0 ? 2 : (yp == 1 ? 1 : 0);
					else if (prev_csbf == 2) sc = xp == 0 ? 2 : (xp == 1 ? 1 : 0);
					else sc = 2;
					if (c == 0) { if (xs | ys) sc += 3; sc += log2n == 3 ? (scan_idx == 0 ? 9 : 15) : 21; }
					else sc += log2n == 3 ? 9 : 12;
				}
				sig[k] = (uint8_t)ae_ctx(C_SIG + (c == 0 ? sc : 27 + sc));
				if (sig[k]) infer_dc = 0;
			} else if (coded && k == 0 && infer_dc) sig[0] = 1;
		}
		int nsig = 0;
		for (int k = 0; k < 16; k++) nsig += sig[k];
		if (!nsig) continue;
		/* greater-than-1 / greater-than-2 flags (9.3.4.2.6, 9.3.4.2.7) */
		int ctx_set = (i == 0 || c > 0) ? 0 : 2;
		if (first_sb_done && c1_carry == 0) ctx_set++;
		first_sb_done = 1;
		int g1ctx = 1, num_g1 = 0, last_g1_pos = -1, first_sig = 16, last_sig = -1;
		uint8_t g1[16], g2[16];
		memset(g1, 0, sizeof g1); memset(g2, 0, sizeof g2);
		for (int k = 15; k >= 0; k--) {
			if (!sig[k]) continue;
			if (num_g1 < 8) {
				g1[k] = (uint8_t)ae_ctx(C_G1 + ctx_set * 4 + imin(3, g1ctx) + (c ? 16 : 0));
				num_g1++;
				if (g1[k]) { g1ctx = 0; if (last_g1_pos < 0) last_g1_pos = k; }
				else if (g1ctx > 0) g1ctx++;
			}
			if (last_sig < 0) last_sig = k;
			first_sig = k;
		}
		c1_carry = g1ctx;
		const int sign_hidden = last_sig - first_sig > 3;
		if (last_g1_pos >= 0) g2[last_g1_pos] = (uint8_t)ae_ctx(C_G2 + ctx_set + (c ? 4 : 0));
		uint8_t sign[16];
		memset(sign, 0, sizeof sign);
		for (int k = 15; k >= 0; k--)
			if (sig[k] && (!pps.sign_hiding || !sign_hidden || k != first_sig)) sign[k] = (uint8_t)ae_bypass();
		int num_sig = 0, sum_abs = 0, rice = 0;
		for (int k = 15; k >= 0; k--) {
			if (!sig[k]) continue;
			int base = 1 + g1[k] + g2[k];
			const int thr = num_sig < 8 ? (k == last_g1_pos ? 3 : 2) : 1;
			int absl = base;
			if (base == thr) {      /* coeff_abs_level_remaining 9.3.3.11 */
				int prefix = 0;
				while (prefix < 32 && ae_bypass()) prefix++;
				if (prefix >= 32) VIOLATION("coeff_abs_level_remaining prefix");
				int rem;
				if (prefix <= 3) rem = (prefix << rice) + (int)ae_bypass_bits(rice);
				else rem = (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)ae_bypass_bits(prefix - 3 + rice);
				absl = base + rem;
				if (absl > 3 * (1 << rice)) rice = imin(rice + 1, 4);
			}
			int v = sign[k] ? -absl : absl;
			if (pps.sign_hiding && sign_hidden) {
				sum_abs += absl;
				if (k == first_sig && (sum_abs & 1)) v = -v;
			}
			if (v < -32768 || v > 32767) VIOLATION("coefficient level out of range");
			lev[((ys << 2) + pos_scan[k][1]) * n + (xs << 2) + pos_scan[k][0]] = (int16_t)v;
			num_sig++;
		}
	}
}

static void add_residual(int c, int x0, int y0, int log2n, const int *res)
{
	const int n = 1 << log2n, W = cur.w[c];
	uint8_t *p = cur.pl[c] + y0 * W + x0;
	for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) p[y * W + x] = (uint8_t)clip8(p[y * W + x] + res[y * n + x]);
}

static int cur_cu_qp(void)
{
	const int q = qp_pred + cu_qp_delta_val;
	return ((q + 52) % 52);
}

/* transform_unit 7.3.8.10 with the reconstruction of 8.6 done block by block */
static void transform_unit(const CU *cu, int x0, int y0, int xb, int yb, int log2n, int depth, int blk, int cbf_l, int cbf_cb, int cbf_cr, int pcbf_cb, int pcbf_cr)
{
	const int chroma_here = log2n > 2, chroma_parent = log2n == 2 && blk == 3;
	const int ccb = chroma_here ? cbf_cb : (chroma_parent ? pcbf_cb : 0), ccr = chroma_here ? cbf_cr : (chroma_parent ? pcbf_cr : 0);
	const int any_c = log2n > 2 ? (cbf_cb | cbf_cr) : (pcbf_cb | pcbf_cr);      /* cbfChroma of the syntax: the parent's flags for 4x4 luma blocks */
	(void)depth;
	if (cbf_l || any_c) {
		if (pps.cu_qp_delta && !is_cu_qp_delta_coded) {      /* cu_qp_delta_abs 9.3.3.10 */
			int v = 0;
			if (ae_ctx(C_DQP)) { v = 1; while (v < 5 && ae_ctx(C_DQP + 1)) v++; }
			if (v == 5) { int k = 0; while (ae_bypass()) { v += 1 << k; k++; if (k > 16) VIOLATION("cu_qp_delta_abs suffix"); } v += (int)ae_bypass_bits(k); }
			if (v && ae_bypass()) v = -v;
			if (v < -26 || v > 25) VIOLATION("CuQpDeltaVal %d out of range", v);
			is_cu_qp_delta_coded = 1;
			cu_qp_delta_val = v;
		}
	}
	const int intra = cu->pred_mode == PM_INTRA, qp = cur_cu_qp();
	static int res[32 * 32];
	/* luma */
	{
		int imode = 0, scan_idx = 0;
		if (intra) {
			imode = u_imode[U(x0, y0)];
			if (log2n == 2 || log2n == 3) scan_idx = (imode >= 6 && imode <= 14) ? 2 : ((imode >= 22 && imode <= 30) ? 1 : 0);
		}
		if (cbf_l) parse_residual(log2n, 0, scan_idx, lev_buf[0]);
		if (ccb || ccr) {
			const int l2c = chroma_here ? log2n - 1 : 2;
			int cscan = 0;
			if (intra && l2c == 2) { const int m = cu->intra_c; cscan = (m >= 6 && m <= 14) ? 2 : ((m >= 22 && m <= 30) ? 1 : 0); }
			if (ccb) parse_residual(l2c, 1, cscan, lev_buf[1]);
			if (ccr) parse_residual(l2c, 2, cscan, lev_buf[2]);
		}
		if (intra) intra_predict(0, x0, y0, log2n, imode);
		if (cbf_l) {
			residual_from_levels(lev_buf[0], log2n, qp, intra, intra && log2n == 2, res);
			add_residual(0, x0, y0, log2n, res);
		}
		for (int y = y0; y < y0 + (1 << log2n); y += 4)
			for (int x = x0; x < x0 + (1 << log2n); x += 4) {
				u_nz[U(x, y)] = (uint8_t)cbf_l;
				u_pm[U(x, y)] = (uint8_t)cu->pred_mode;      /* (reconstructed: available to later intra blocks) */
			}
	}
	/* chroma: with the luma block when it is larger than 4x4, else after the fourth 4x4 luma block for the 8x8 parent */
	if (chroma_here || chroma_parent) {
		const int xc = (chroma_here ? x0 : xb) >> 1, yc = (chroma_here ? y0 : yb) >> 1, l2c = chroma_here ? log2n - 1 : 2;
		for (int c = 1; c < 3; c++) {
			if (intra) intra_predict(c, xc, yc, l2c, cu->intra_c);
			if (c == 1 ? ccb : ccr) {
				residual_from_levels(lev_buf[c], l2c, chroma_qp(qp, c == 1 ? pps.cb_off : pps.cr_off), intra, 0, res);
				add_residual(c, xc, yc, l2c, res);
			}
		}
	}
	/* transform block edges for the deblocking filter (8.7.2.3) */
	for (int k = 0; k < (1 << log2n); k += 4) { u_tuedge_v[U(x0, y0 + k)] = 1; u_tuedge_h[U(x0 + k, y0)] = 1; }
}

static void transform_tree(const CU *cu, int x0, int y0, int xb, int yb, int log2n, int depth, int blk, int pcbf_cb, int pcbf_cr)
{
	int split;
	if (log2n <= sps.max_tb_log2 && log2n > sps.min_tb_log2 && depth < cu->max_trafo_depth && !(cu->intra_split && depth == 0))
		split = ae_ctx(C_SPLIT_TR + 5 - log2n);
	else split = log2n > sps.max_tb_log2 || (cu->intra_split && depth == 0);      /* (interSplitFlag: only with max_transform_hierarchy_depth_inter 0 and a partitioned CU - not parsed below 2Nx2N) */
	int cbf_cb = 0, cbf_cr = 0;
	if (log2n > 2) {
		if (depth == 0 || pcbf_cb) cbf_cb = ae_ctx(C_CBF_C + depth);
		if (depth == 0 || pcbf_cr) cbf_cr = ae_ctx(C_CBF_C + depth);
	} else { cbf_cb = pcbf_cb; cbf_cr = pcbf_cr; }      /* 7.4.9.8: inferred from the parent for 4x4 luma blocks */
	if (split) {
		const int h = 1 << (log2n - 1);
		transform_tree(cu, x0, y0, x0, y0, log2n - 1, depth + 1, 0, cbf_cb, cbf_cr);
		transform_tree(cu, x0 + h, y0, x0, y0, log2n - 1, depth + 1, 1, cbf_cb, cbf_cr);
		transform_tree(cu, x0, y0 + h, x0, y0, log2n - 1, depth + 1, 2, cbf_cb, cbf_cr);
		transform_tree(cu, x0 + h, y0 + h, x0, y0, log2n - 1, depth + 1, 3, cbf_cb, cbf_cr);
	} else {
		int cbf_l = 1;
		if (cu->pred_mode == PM_INTRA || depth != 0 || cbf_cb || cbf_cr) cbf_l = ae_ctx(C_CBF_LUMA + (depth == 0 ? 1 : 0));
		transform_unit(cu, x0, y0, xb, yb, log2n, depth, blk, cbf_l, log2n > 2 ? cbf_cb : 0, log2n > 2 ? cbf_cr : 0, pcbf_cb, pcbf_cr);
	}
}

/* 6.4.2 for merge / AMVP neighbours of a 2Nx2N prediction block at (xp, yp): available, decoded, and inter */
static int nb_inter(int xp, int yp, int xn, int yn)
{
	return avail_z(xp, yp, xn, yn) && u_pm[U(xn, yn)] == PM_INTER;
}

static int prediction_unit(CU *cu, int x0, int y0, int w, int h)      /* 7.3.8.6, 8.5.3.2; returns merge_flag */
{
	int merge = cu->skip, merge_idx = 0, mvx = 0, mvy = 0;
	if (!cu->skip) merge = ae_ctx(C_MERGE_FLAG);
	if (merge) {
		if (sl.max_merge > 1) { if (ae_ctx(C_MERGE_IDX)) { merge_idx = 1; while (merge_idx < sl.max_merge - 1 && ae_bypass()) merge_idx++; } }
		/* 8.5.3.2.2 / 8.5.3.2.3: spatial candidates A1, B1, B0, A0, B2 with their comparisons, then zero candidates (P slice, one reference picture) */
		int cx[6], cy[6], nc = 0;
		const int xa1 = x0 - 1, ya1 = y0 + h - 1, xb1 = x0 + w - 1, yb1 = y0 - 1, xb0 = x0 + w, yb0 = y0 - 1, xa0 = x0 - 1, ya0 = y0 + h, xb2 = x0 - 1, yb2 = y0 - 1;
		const int a1 = nb_inter(x0, y0, xa1, ya1);
		if (a1) { cx[nc] = u_mvx[U(xa1, ya1)]; cy[nc] = u_mvy[U(xa1, ya1)]; nc++; }
		const int b1 = nb_inter(x0, y0, xb1, yb1) && !(a1 && u_mvx[U(xa1, ya1)] == u_mvx[U(xb1, yb1)] && u_mvy[U(xa1, ya1)] == u_mvy[U(xb1, yb1)]);
		if (b1) { cx[nc] = u_mvx[U(xb1, yb1)]; cy[nc] = u_mvy[U(xb1, yb1)]; nc++; }
		const int b1_av = nb_inter(x0, y0, xb1, yb1);
		const int b0 = nb_inter(x0, y0, xb0, yb0) && !(b1_av && u_mvx[U(xb1, yb1)] == u_mvx[U(xb0, yb0)] && u_mvy[U(xb1, yb1)] == u_mvy[U(xb0, yb0)]);
		if (b0) { cx[nc] = u_mvx[U(xb0, yb0)]; cy[nc] = u_mvy[U(xb0, yb0)]; nc++; }
		const int a0 = nb_inter(x0, y0, xa0, ya0) && !(a1 && u_mvx[U(xa1, ya1)] == u_mvx[U(xa0, ya0)] && u_mvy[U(xa1, ya1)] == u_mvy[U(xa0, ya0)]);
		if (a0) { cx[nc] = u_mvx[U(xa0, ya0)]; cy[nc] = u_mvy[U(xa0, ya0)]; nc++; }
		if (nc < 4) {
			const int b2 = nb_inter(x0, y0, xb2, yb2) && !(a1 && u_mvx[U(xa1, ya1)] == u_mvx[U(xb2, yb2)] && u_mvy[U(xa1, ya1)] == u_mvy[U(xb2, yb2)]) &&
				       !(b1_av && u_mvx[U(xb1, yb1)] == u_mvx[U(xb2, yb2)] && u_mvy[U(xb1, yb1)] == u_mvy[U(xb2, yb2)]);
			if (b2) { cx[nc] = u_mvx[U(xb2, yb2)]; cy[nc] = u_mvy[U(xb2, yb2)]; nc++; }
		}
		while (nc < 6) { cx[nc] = 0; cy[nc] = 0; nc++; }
		if (merge_idx >= sl.max_merge) VIOLATION("merge_idx");
		mvx = cx[merge_idx]; mvy = cy[merge_idx];
	} else {
		/* (one reference picture: no ref_idx) mvd_coding 7.3.8.9 */
		const int g0x = ae_ctx(C_MVD_G0), g0y = ae_ctx(C_MVD_G0);
		int g1x = 0, g1y = 0, dx = 0, dy = 0;
		if (g0x) g1x = ae_ctx(C_MVD_G1);
		if (g0y) g1y = ae_ctx(C_MVD_G1);
		for (int comp = 0; comp < 2; comp++) {
			const int g0 = comp ? g0y : g0x, g1 = comp ? g1y : g1x;
			int v = 0;
			if (g0) {
				v = 1;
				if (g1) {      /* abs_mvd_minus2: EG1 */
					int k = 1, a = 0;
					while (ae_bypass()) { a += 1 << k; k++; if (k > 20) VIOLATION("abs_mvd_minus2"); }
					a += (int)ae_bypass_bits(k);
					v = a + 2;
				}
				if (ae_bypass()) v = -v;
			}
			if (comp) dy = v; else dx = v;
		}
		const int mvp_flag = ae_ctx(C_MVP);
		/* 8.5.3.2.6 / 8.5.3.2.7 with one reference picture: every inter neighbour refers to it, no scaling */
		int px[3], py[3], np = 0;
		const int xa0 = x0 - 1, ya0 = y0 + h, xa1 = x0 - 1, ya1 = y0 + h - 1;
		const int av_a0 = nb_inter(x0, y0, xa0, ya0), av_a1 = nb_inter(x0, y0, xa1, ya1);
		int have_a = 0, ax = 0, ay = 0;
		if (av_a0) { have_a = 1; ax = u_mvx[U(xa0, ya0)]; ay = u_mvy[U(xa0, ya0)]; }
		else if (av_a1) { have_a = 1; ax = u_mvx[U(xa1, ya1)]; ay = u_mvy[U(xa1, ya1)]; }
		const int bxs[3] = {x0 + w, x0 + w - 1, x0 - 1}, bys[3] = {y0 - 1, y0 - 1, y0 - 1};
		int have_b = 0, bx = 0, by = 0;
		for (int k = 0; k < 3 && !have_b; k++)
			if (nb_inter(x0, y0, bxs[k], bys[k])) { have_b = 1; bx = u_mvx[U(bxs[k], bys[k])]; by = u_mvy[U(bxs[k], bys[k])]; }
		if (!(av_a0 || av_a1) && have_b) { have_a = 1; ax = bx; ay = by; }      /* isScaledFlag 0: B moves to A, and the re-derived B is the same vector */
		if (have_a) { px[np] = ax; py[np] = ay; np++; }
		if (have_b && !(have_a && ax == bx && ay == by)) { px[np] = bx; py[np] = by; np++; }
		while (np < 2) { px[np] = 0; py[np] = 0; np++; }
		mvx = (int16_t)(px[mvp_flag] + dx); mvy = (int16_t)(py[mvp_flag] + dy);
	}
	if (!have_ref) VIOLATION("a P block without a reference picture");
	for (int y = y0; y < y0 + h; y += 4)
		for (int x = x0; x < x0 + w; x += 4) { u_mvx[U(x, y)] = (int16_t)mvx; u_mvy[U(x, y)] = (int16_t)mvy; }
	inter_predict(0, x0, y0, w, h, mvx, mvy);
	inter_predict(1, x0 / 2, y0 / 2, w / 2, h / 2, mvx, mvy);
	inter_predict(2, x0 / 2, y0 / 2, w / 2, h / 2, mvx, mvy);
	return merge;      /* (a merged 2Nx2N block has no rqt_root_cbf) */
}

static void coding_unit(int x0, int y0, int log2cb)      /* 7.3.8.5 */
{
	CU cu;
	memset(&cu, 0, sizeof cu);
	cu.log2cb = log2cb; cu.x = x0; cu.y = y0;
	const int n = 1 << log2cb;
	if (sl.type != 2) {
		const int cl = x0 > 0 && u_skip[U(x0 - 1, y0)], ca = y0 > 0 && u_skip[U(x0, y0 - 1)];
		cu.skip = ae_ctx(C_SKIP + cl + ca);
	}
	cu.pred_mode = sl.type == 2 ? PM_INTRA : PM_INTER;
	int rqt_root = 1, merge2n = 0;
	for (int y = y0; y < y0 + n; y += 4)
		for (int x = x0; x < x0 + n; x += 4) {
			u_skip[U(x, y)] = (uint8_t)cu.skip; u_depth[U(x, y)] = (uint8_t)(sps.ctb_log2 - log2cb);
			u_puedge_v[U(x, y)] = x == x0; u_puedge_h[U(x, y)] = y == y0;
			u_nz[U(x, y)] = 0;
		}
	if (cu.skip) {
		for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) u_pm[U(x, y)] = PM_NONE;      /* (not yet available to itself) */
		(void)prediction_unit(&cu, x0, y0, n, n);
		rqt_root = 0;
	} else {
		if (sl.type != 2) cu.pred_mode = ae_ctx(C_PRED_MODE) ? PM_INTRA : PM_INTER;
		if (cu.pred_mode != PM_INTRA || log2cb == sps.min_cb_log2) {
			if (!ae_ctx(C_PART_MODE)) {
				if (cu.pred_mode == PM_INTRA) cu.part_nxn = 1;
				else UNSUPPORTED("inter partitions other than 2Nx2N");
			}
		}
		if (cu.pred_mode == PM_INTRA) {
			cu.intra_split = cu.part_nxn;
			const int np = cu.part_nxn ? 4 : 1, pb = cu.part_nxn ? n / 2 : n;
			int prev[4], mpm_idx[4], rem[4];
			for (int i = 0; i < np; i++) prev[i] = ae_ctx(C_PREV_INTRA);
			for (int i = 0; i < np; i++) {
				if (prev[i]) { mpm_idx[i] = ae_bypass() ? (ae_bypass() ? 2 : 1) : 0; rem[i] = 0; }
				else { rem[i] = (int)ae_bypass_bits(5); mpm_idx[i] = 0; }
			}
			for (int i = 0; i < np; i++) {      /* 8.4.2 */
				const int xp = x0 + (i & 1) * pb, yp = y0 + (i >> 1) * pb;
				int ca = 1, cb = 1;
				if (avail_z(xp, yp, xp - 1, yp + 0) && u_pm[U(xp - 1, yp)] == PM_INTRA) ca = u_imode[U(xp - 1, yp)];
				if (avail_z(xp, yp, xp, yp - 1) && u_pm[U(xp, yp - 1)] == PM_INTRA && yp - 1 >= ((yp >> sps.ctb_log2) << sps.ctb_log2)) cb = u_imode[U(xp, yp - 1)];
				int cand[3];
				if (ca == cb) {
					if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
					else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
				} else {
					cand[0] = ca; cand[1] = cb;
					cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26);
				}
				int mode;
				if (prev[i]) mode = cand[mpm_idx[i]];
				else {
					if (cand[0] > cand[1]) { const int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
					if (cand[0] > cand[2]) { const int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
					if (cand[1] > cand[2]) { const int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
					mode = rem[i];
					for (int k = 0; k < 3; k++) if (mode >= cand[k]) mode++;
				}
				if (mode < 0 || mode > 34) VIOLATION("intra prediction mode %d", mode);
				cu.intra_y[i] = mode;
				for (int y = yp; y < yp + pb; y += 4)
					for (int x = xp; x < xp + pb; x += 4) { u_imode[U(x, y)] = (uint8_t)mode; u_pm[U(x, y)] = PM_NONE; }
				/* (the neighbour derivation of the next partition reads u_pm: mark this one intra for that purpose) */
				for (int y = yp; y < yp + pb; y += 4) for (int x = xp; x < xp + pb; x += 4) u_pm[U(x, y)] = PM_INTRA;
			}
			/* intra_chroma_pred_mode 9.3.3.8, 8.4.3 */
			int icp = 4;
			if (ae_ctx(C_CHROMA_PRED)) icp = (int)ae_bypass_bits(2);
			static const uint8_t cmode[4] = {0, 26, 10, 1};
			cu.intra_c = icp == 4 ? cu.intra_y[0] : (cmode[icp] == cu.intra_y[0] ? 34 : cmode[icp]);
			/* the CU's samples are not reconstructed yet: intra prediction inside the CU must not see them as available before their block is done */
			for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) u_pm[U(x, y)] = PM_NONE;
			cu.max_trafo_depth = sps.max_th_intra + cu.intra_split;
		} else {
			for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) u_pm[U(x, y)] = PM_NONE;
			merge2n = prediction_unit(&cu, x0, y0, n, n);
			if (!merge2n) rqt_root = ae_ctx(C_RQT_ROOT);
			cu.max_trafo_depth = sps.max_th_inter;
		}
	}
	if (cu.pred_mode == PM_INTER)
		for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) u_pm[U(x, y)] = PM_INTER;
	if (rqt_root) transform_tree(&cu, x0, y0, x0, y0, log2cb, 0, 0, 0, 0);
	else for (int k = 0; k < n; k += 4) { u_tuedge_v[U(x0, y0 + k)] = 1; u_tuedge_h[U(x0 + k, y0)] = 1; }
	const int qp = cur_cu_qp();
	if (qp < 0 || qp > 51) VIOLATION("QpY %d", qp);
	for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) { u_qp[U(x, y)] = (int8_t)qp; u_pm[U(x, y)] = (uint8_t)cu.pred_mode; }
	last_cu_qp = qp;
}

static void coding_quadtree(int x0, int y0, int log2cb, int depth)      /* 7.3.8.4 */
{
	const int n = 1 << log2cb;
	int split;
	if (x0 + n <= sps.w && y0 + n <= sps.h && log2cb > sps.min_cb_log2) {
		const int cl = x0 > 0 && avail_z(x0, y0, x0 - 1, y0) && u_depth[U(x0 - 1, y0)] > depth;
		const int ca = y0 > 0 && avail_z(x0, y0, x0, y0 - 1) && u_depth[U(x0, y0 - 1)] > depth;
		split = ae_ctx(C_SPLIT_CU + cl + ca);
	} else split = log2cb > sps.min_cb_log2;
	if (pps.cu_qp_delta && log2cb >= sps.ctb_log2 - pps.diff_cu_qp_delta_depth) { is_cu_qp_delta_coded = 0; cu_qp_delta_val = 0; }
	if (split) {
		const int h = n >> 1;
		coding_quadtree(x0, y0, log2cb - 1, depth + 1);
		if (x0 + h < sps.w) coding_quadtree(x0 + h, y0, log2cb - 1, depth + 1);
		if (y0 + h < sps.h) coding_quadtree(x0, y0 + h, log2cb - 1, depth + 1);
		if (x0 + h < sps.w && y0 + h < sps.h) coding_quadtree(x0 + h, y0 + h, log2cb - 1, depth + 1);
	} else coding_unit(x0, y0, log2cb);
}

/* ---- deblocking (8.7.2) --------------------------------------------------------------------------------------------------- */
static const uint8_t tc_tab[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static int beta_of(int q) { return q < 16 ? 0 : (q <= 28 ? q - 10 : 2 * q - 38); }
static int edge_bs(int xp, int yp, int xq, int yq, int is_tu_edge)      /* 8.7.2.4 */
{
	const size_t p = U(xp, yp), q = U(xq, yq);
	if (u_pm[p] == PM_INTRA || u_pm[q] == PM_INTRA) return 2;
	if (is_tu_edge && (u_nz[p] || u_nz[q])) return 1;
	if (iabs(u_mvx[p] - u_mvx[q]) >= 4 || iabs(u_mvy[p] - u_mvy[q]) >= 4) return 1;
	return 0;
}
static void deblock_luma_edge(uint8_t *s, int xs, int ys, int bs, int qp)      /* four lines; (xs, ys) = step across / along the edge in samples */
{
	const int beta = beta_of(clip3(0, 51, qp)), tc = tc_tab[clip3(0, 53, qp + 2 * (bs - 1))];
#define P(i, k) s[-((i) + 1) * xs + (k) * ys]
#define Q(i, k) s[(i) * xs + (k) * ys]
	const int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
	const int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
	const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
	if (d >= beta) return;
	const int ds0 = 2 * dpq0 < (beta >> 2) && iabs(P(3, 0) - P(0, 0)) + iabs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && iabs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
	const int ds3 = 2 * dpq3 < (beta >> 2) && iabs(P(3, 3) - P(0, 3)) + iabs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && iabs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
	const int strong = ds0 && ds3, dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
	for (int k = 0; k < 4; k++) {
		const int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0 = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
		if (strong) {
			P(0, k) = (uint8_t)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
			P(1, k) = (uint8_t)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
			P(2, k) = (uint8_t)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
			Q(0, k) = (uint8_t)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
			Q(1, k) = (uint8_t)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
			Q(2, k) = (uint8_t)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
		} else {
			int dl = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
			if (iabs(dl) >= tc * 10) continue;
			dl = clip3(-tc, tc, dl);
			P(0, k) = (uint8_t)clip8(p0 + dl);
			Q(0, k) = (uint8_t)clip8(q0 - dl);
			if (dep) P(1, k) = (uint8_t)clip8(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + dl) >> 1));
			if (deq) Q(1, k) = (uint8_t)clip8(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - dl) >> 1));
		}
	}
#undef P
#undef Q
}
static void deblock_chroma_edge(uint8_t *s, int xs, int ys, int qp_avg, int off)      /* the two chroma lines that belong to a four-line luma segment */
{
	const int qpc = chroma_qp(qp_avg, off), tc = tc_tab[clip3(0, 53, qpc + 2)];
	for (int k = 0; k < 2; k++) {
		const int p0 = s[-xs + k * ys], p1 = s[-2 * xs + k * ys], q0 = s[k * ys], q1 = s[xs + k * ys];
		const int dl = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
		s[-xs + k * ys] = (uint8_t)clip8(p0 + dl);
		s[k * ys] = (uint8_t)clip8(q0 - dl);
	}
}
static void deblock_picture(void)
{
	const int W = sps.w, H = sps.h;
	/* vertical edges of the whole picture, then horizontal edges on their output (8.7.2) */
	for (int dir = 0; dir < 2; dir++) {
		/* boundary strengths first: the decisions of one direction read unfiltered samples of that direction only, and edges are 8 apart */
		for (int y = 0; y < H; y += 4)
			for (int x = 0; x < W; x += 4) {
				int bs = 0;
				if (dir == 0) {
					if (x == 0 || (x & 7)) continue;
					const int tu = u_tuedge_v[U(x, y)], pu = u_puedge_v[U(x, y)];
					if (!tu && !pu) continue;
					bs = edge_bs(x - 1, y, x, y, tu);
					if (!bs) continue;
					const int qp = (u_qp[U(x - 1, y)] + u_qp[U(x, y)] + 1) >> 1;
					deblock_luma_edge(cur.pl[0] + y * W + x, 1, W, bs, qp);
				} else {
					if (y == 0 || (y & 7)) continue;
					const int tu = u_tuedge_h[U(x, y)], pu = u_puedge_h[U(x, y)];
					if (!tu && !pu) continue;
					bs = edge_bs(x, y - 1, x, y, tu);
					if (!bs) continue;
					const int qp = (u_qp[U(x, y - 1)] + u_qp[U(x, y)] + 1) >> 1;
					deblock_luma_edge(cur.pl[0] + y * W + x, W, 1, bs, qp);
				}
			}
		/* chroma: edges on the 8-sample chroma grid (16 luma samples), strength 2 only */
		for (int y = 0; y < H; y += 4)
			for (int x = 0; x < W; x += 4) {
				if (dir == 0) {
					if (x == 0 || (x & 15)) continue;
					if (!u_tuedge_v[U(x, y)] && !u_puedge_v[U(x, y)]) continue;
					if (edge_bs(x - 1, y, x, y, 1) != 2) continue;
					const int qp = (u_qp[U(x - 1, y)] + u_qp[U(x, y)] + 1) >> 1;
					for (int c = 1; c < 3; c++) deblock_chroma_edge(cur.pl[c] + (y / 2) * (W / 2) + x / 2, 1, W / 2, qp, c == 1 ? pps.cb_off : pps.cr_off);
				} else {
					if (y == 0 || (y & 15)) continue;
					if (!u_tuedge_h[U(x, y)] && !u_puedge_h[U(x, y)]) continue;
					if (edge_bs(x, y - 1, x, y, 1) != 2) continue;
					const int qp = (u_qp[U(x, y - 1)] + u_qp[U(x, y)] + 1) >> 1;
					for (int c = 1; c < 3; c++) deblock_chroma_edge(cur.pl[c] + (y / 2) * (W / 2) + x / 2, W / 2, 1, qp, c == 1 ? pps.cb_off : pps.cr_off);
				}
			}
	}
}

/* ---- sample adaptive offset (8.7.3) --------------------------------------------------------------------------------------- */
static void sao_picture(void)
{
	for (int c = 0; c < 3; c++) memcpy(dbk_out.pl[c], cur.pl[c], (size_t)cur.w[c] * cur.h[c]);
	if (!sps.sao || !(sl.sao_luma || sl.sao_chroma)) return;
	for (int ry = 0; ry < sps.hctb; ry++)
		for (int rx = 0; rx < sps.wctb; rx++) {
			const Sao *s = &sao_ctb[ry * sps.wctb + rx];
			for (int c = 0; c < 3; c++) {
				if (!s->type[c]) continue;
				const int cs = (1 << sps.ctb_log2) >> (c ? 1 : 0), W = cur.w[c], H = cur.h[c];
				const int x0 = rx * cs, y0 = ry * cs, x1 = imin(x0 + cs, W), y1 = imin(y0 + cs, H);
				const uint8_t *in = cur.pl[c];
				uint8_t *out = dbk_out.pl[c];
				if (s->type[c] == 1) {
					int band_tab[32];
					memset(band_tab, 0, sizeof band_tab);
					for (int k = 0; k < 4; k++) band_tab[(k + s->band[c]) & 31] = k + 1;
					for (int y = y0; y < y1; y++)
						for (int x = x0; x < x1; x++) {
							const int b = band_tab[in[y * W + x] >> 3];
							if (b) out[y * W + x] = (uint8_t)clip8(in[y * W + x] + s->off[c][b - 1]);
						}
				} else {
					static const int8_t hp[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vp[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
					const int e = s->eo[c];
					for (int y = y0; y < y1; y++)
						for (int x = x0; x < x1; x++) {
							const int xa = x + hp[e][0], ya = y + vp[e][0], xb = x + hp[e][1], yb = y + vp[e][1];
							if (xa < 0 || ya < 0 || xa >= W || ya >= H || xb < 0 || yb < 0 || xb >= W || yb >= H) continue;
							const int v = in[y * W + x];
							int idx = 2 + sgn(v - in[ya * W + xa]) + sgn(v - in[yb * W + xb]);
							if (idx == 0 || idx == 1 || idx == 2) idx = idx == 2 ? 0 : idx + 1;
							if (idx) out[y * W + x] = (uint8_t)clip8(v + s->off[c][idx - 1]);
						}
				}
			}
		}
}

/* ---- slice (7.3.6, 7.3.8.1) ----------------------------------------------------------------------------------------------- */
static int prev_poc_tid0, pictures_out;
static long substreams_checked, entry_points_checked;

static void decode_slice(int nal_type, const uint8_t *rbsp, int n, const int *epb_pos, int n_epb)
{
	BR b = {rbsp, n, 0};
	if (!sps.valid || !pps.valid) VIOLATION("slice before its parameter sets");
	alloc_all();
	if (!br_u(&b, 1)) UNSUPPORTED("more than one slice segment per picture");
	const int irap = nal_type >= 16 && nal_type <= 23;
	if (irap) br_u(&b, 1);
	if (br_ue(&b) != 0) UNSUPPORTED("pps id");
	sl.type = (int)br_ue(&b);
	if (sl.type > 2) VIOLATION("slice_type");
	if (sl.type == 0) UNSUPPORTED("B slices");
	sl.idr = nal_type == 19 || nal_type == 20;
	int poc_lsb = 0, rps_idx = -1;
	if (!sl.idr) {
		poc_lsb = (int)br_u(&b, sps.log2_max_poc);
		if (!br_u(&b, 1)) UNSUPPORTED("a reference picture set in the slice header");
		int nb = 0;
		while ((1 << nb) < sps.num_rps) nb++;
		rps_idx = nb ? (int)br_u(&b, nb) : 0;
		if (rps_idx >= sps.num_rps) VIOLATION("short_term_ref_pic_set_idx");
	}
	if (sl.idr) { sl.poc = 0; }
	else {      /* 8.3.1 */
		const int max_lsb = 1 << sps.log2_max_poc, prev_lsb = prev_poc_tid0 & (max_lsb - 1), prev_msb = prev_poc_tid0 - prev_lsb;
		int msb = prev_msb;
		if (poc_lsb < prev_lsb && prev_lsb - poc_lsb >= max_lsb / 2) msb += max_lsb;
		else if (poc_lsb > prev_lsb && poc_lsb - prev_lsb > max_lsb / 2) msb -= max_lsb;
		sl.poc = msb + poc_lsb;
	}
	sl.sao_luma = sl.sao_chroma = 0;
	if (sps.sao) { sl.sao_luma = (int)br_u(&b, 1); sl.sao_chroma = (int)br_u(&b, 1); }
	sl.max_merge = 5;
	if (sl.type != 2) {
		if (br_u(&b, 1)) UNSUPPORTED("num_ref_idx_active_override");
		if (pps.cabac_init_present) if (br_u(&b, 1)) UNSUPPORTED("cabac_init_flag");
		sl.max_merge = 5 - (int)br_ue(&b);
		if (sl.max_merge < 1 || sl.max_merge > 5) VIOLATION("five_minus_max_num_merge_cand");
	}
	sl.qp = pps.init_qp + br_se(&b);
	if (sl.qp < 0 || sl.qp > 51) VIOLATION("SliceQpY %d", sl.qp);
	if (pps.lf_across_slices) br_u(&b, 1);      /* present because SAO or deblocking is on */
	sl.n_entry = 0;
	if (pps.wpp) {
		sl.n_entry = (int)br_ue(&b);
		if (sl.n_entry > 255) UNSUPPORTED("more than 255 entry points");
		if (sl.n_entry != 0 && sl.n_entry != sps.hctb - 1) VIOLATION("num_entry_point_offsets %d for %d CTB rows", sl.n_entry, sps.hctb);
		if (sl.n_entry) {
			const int len = (int)br_ue(&b) + 1;
			if (len > 32) VIOLATION("offset_len_minus1");
			for (int i = 0; i < sl.n_entry; i++) sl.entry[i] = (long)br_u(&b, len) + 1;
		}
	}
	if (!br_u(&b, 1)) VIOLATION("byte_alignment of the slice header");
	while (b.pos & 7) if (br_bit(&b)) VIOLATION("byte_alignment of the slice header");
	/* reference picture: the set names exactly the previous picture (8.3.2) */
	if (sl.type == 1) {
		if (rps_idx < 0 || sps.rps[rps_idx].nneg != 1 || sps.rps[rps_idx].dpoc[0] != -1 || !sps.rps[rps_idx].used[0]) UNSUPPORTED("a reference picture set other than {POC - 1}");
		if (!have_ref) VIOLATION("P slice without a decoded reference picture");
	}
	/* entry points: byte positions in the escaped slice data -> positions in the unescaped payload (7.4.7.1) */
	const long data_start = b.pos >> 3;
	long entry_unesc[256];
	{
		/* escaped offset of the slice data start = data_start + emulation prevention bytes before it */
		int before = 0;
		for (int i = 0; i < n_epb; i++) if (epb_pos[i] < data_start) before++;
		long esc = data_start + before;
		for (int i = 0; i < sl.n_entry; i++) {
			esc += sl.entry[i];
			/* unescaped = escaped - number of emulation prevention bytes located before that escaped position */
			int cnt = 0;
			for (int k = 0; k < n_epb; k++) if ((long)epb_pos[k] + k < esc) cnt++;      /* epb k sits at escaped position epb_pos[k] + k */
			entry_unesc[i] = esc - cnt;
		}
	}
	/* slice data */
	const int init_type = sl.type == 2 ? 0 : 1;
	memset(u_pm, 0, (size_t)sps.w4 * sps.h4); memset(u_skip, 0, (size_t)sps.w4 * sps.h4); memset(u_tuedge_v, 0, (size_t)sps.w4 * sps.h4);
	memset(u_tuedge_h, 0, (size_t)sps.w4 * sps.h4); memset(u_puedge_v, 0, (size_t)sps.w4 * sps.h4); memset(u_puedge_h, 0, (size_t)sps.w4 * sps.h4);
	memset(u_nz, 0, (size_t)sps.w4 * sps.h4); memset(u_depth, 0, (size_t)sps.w4 * sps.h4);
	free(sao_ctb);
	sao_ctb = (Sao *)calloc((size_t)sps.wctb * sps.hctb, sizeof(Sao));
	cab.b = &b;
	ctx_init_all(init_type, sl.qp);
	cabac_start();
	wpp_saved_valid = 0;
	last_cu_qp = sl.qp;
	const int nctb = sps.wctb * sps.hctb;
	for (int a = 0; a < nctb; a++) {
		const int rx = a % sps.wctb, ry = a / sps.wctb;
		if (pps.wpp && rx == 0 && a > 0) {      /* 9.3.1: synchronisation with the CTB above-right, or initialisation when there is none */
			if (sps.wctb > 1) {
				if (!wpp_saved_valid) VIOLATION("no stored context variables for the row start");
				memcpy(cab.st, wpp_saved, sizeof wpp_saved);
			} else ctx_init_all(init_type, sl.qp);
		}
		/* 8.6.1: the first quantization group of a slice, and of a CTB row with wavefronts, predicts from SliceQpY; others from the last CU of the previous group */
		qp_pred = (a == 0 || (pps.wpp && rx == 0)) ? sl.qp : last_cu_qp;
		cu_qp_delta_val = 0; is_cu_qp_delta_coded = 0;
		if (sl.sao_luma || sl.sao_chroma) parse_sao(rx, ry);
		coding_quadtree(rx << sps.ctb_log2, ry << sps.ctb_log2, sps.ctb_log2, 0);
		if (ref_deblock_qp && is_cu_qp_delta_coded) {      /* NOT the standard: see the header (R1) */
			const int cs = 1 << sps.ctb_log2;
			for (int y = ry * cs; y < imin(sps.h, (ry + 1) * cs); y += 4)
				for (int x = rx * cs; x < imin(sps.w, (rx + 1) * cs); x += 4) u_qp[U(x, y)] = (int8_t)last_cu_qp;
		}
		if (pps.wpp && rx == 1) { memcpy(wpp_saved, cab.st, sizeof wpp_saved); wpp_saved_valid = 1; }
		const int end = ae_terminate();
		if (end != (a == nctb - 1)) VIOLATION("end_of_slice_segment_flag %d after CTU %d of %d", end, a, nctb);
		if (end) { check_substream_end("slice data"); substreams_checked++; break; }
		if (pps.wpp && rx == sps.wctb - 1) {
			if (!ae_terminate()) VIOLATION("end_of_subset_one_bit is 0 after CTB row %d", ry);
			check_substream_end("CTB row");
			substreams_checked++;
			if (sl.n_entry) {
				if ((b.pos >> 3) != entry_unesc[ry]) VIOLATION("CTB row %d ends at byte %ld of the payload, its entry point says %ld", ry, b.pos >> 3, entry_unesc[ry]);
				entry_points_checked++;
			}
			cabac_start();
		}
	}
	if ((b.pos >> 3) != n) VIOLATION("%ld bytes of slice data decoded, the NAL unit payload has %d", b.pos >> 3, n);
	deblock_picture();
	sao_picture();
	for (int c = 0; c < 3; c++) memcpy(ref.pl[c], dbk_out.pl[c], (size_t)ref.w[c] * ref.h[c]);
	have_ref = 1;
	prev_poc_tid0 = sl.poc;
	if (verbose) fprintf(stderr, "picture %d: %s POC %d QP %d, %d payload bytes, %d entry points\n", pictures_out, sl.type == 2 ? "I" : "P", sl.poc, sl.qp, n, sl.n_entry);
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s in.265 out.yuv|- [-v] [--ref-deblock-qp]\n", argv[0]); return 1; }
	for (int i = 3; i < argc; i++) {
		if (!strcmp(argv[i], "-v")) verbose = 1;
		else if (!strcmp(argv[i], "--ref-deblock-qp")) ref_deblock_qp = 1;
		else FAIL(1, "unknown option %s", argv[i]);
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) FAIL(1, "cannot open %s", argv[1]);
	fseek(f, 0, SEEK_END);
	const long size = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *data = (uint8_t *)malloc((size_t)size + 4);
	if (fread(data, 1, (size_t)size, f) != (size_t)size) FAIL(1, "short read");
	fclose(f);
	FILE *fo = strcmp(argv[2], "-") ? fopen(argv[2], "wb") : NULL;
	build_scans();
	build_dct();
	uint8_t *rbsp = (uint8_t *)malloc((size_t)size + 4);
	int *epb = (int *)malloc(sizeof(int) * ((size_t)size / 3 + 4));
	long pos = 0;
	int nals = 0;
	/* Annex B: start code prefixes, NAL unit header (7.3.1.2), emulation prevention removal (7.3.1.1) */
	while (pos + 3 <= size) {
		if (!(data[pos] == 0 && data[pos + 1] == 0 && data[pos + 2] == 1)) { if (data[pos] != 0) VIOLATION("bytes between NAL units at %ld", pos); pos++; continue; }
		pos += 3;
		long end = pos;
		while (end + 3 <= size && !(data[end] == 0 && data[end + 1] == 0 && (data[end + 2] == 1 || data[end + 2] == 0))) end++;
		if (end + 3 > size) end = size;
		if (end - pos < 2) VIOLATION("NAL unit shorter than its header");
		if (data[pos] & 0x80) VIOLATION("forbidden_zero_bit");
		const int type = (data[pos] >> 1) & 63, tid = (data[pos + 1] & 7) - 1;
		if (tid != 0) UNSUPPORTED("temporal sub-layers");
		int n = 0, ne = 0, zeros = 0;
		for (long i = pos + 2; i < end; i++) {
			if (zeros >= 2 && data[i] == 3) { epb[ne++] = n; zeros = 0; continue; }      /* position in the unescaped payload at which a byte was removed */
			if (zeros >= 2 && data[i] < 3) VIOLATION("start code emulation inside a NAL unit at %ld", i);
			rbsp[n++] = data[i];
			zeros = data[i] == 0 ? zeros + 1 : 0;
		}
		nals++;
		BR b = {rbsp, n, 0};
		if (type == 32) { /* VPS: nothing the decoding process needs */ }
		else if (type == 33) parse_sps(&b);
		else if (type == 34) parse_pps(&b);
		else if (type == 19 || type == 20 || type == 1 || type == 0) {
			/* cabac_zero_words / trailing zero bytes are not part of the slice data */
			decode_slice(type, rbsp, n, epb, ne);
			if (fo) for (int c = 0; c < 3; c++) fwrite(dbk_out.pl[c], 1, (size_t)dbk_out.w[c] * dbk_out.h[c], fo);
			pictures_out++;
		} else if (type >= 35 && type <= 40) { /* AUD, EOS, EOB, filler, SEI: ignored */ }
		else UNSUPPORTED("NAL unit type %d", type);
		pos = end;
	}
	if (fo) fclose(fo);
	printf("DECODED pictures=%d nal_units=%d substreams=%ld entry_points=%ld width=%d height=%d\n", pictures_out, nals, substreams_checked, entry_points_checked, sps.w, sps.h);
	return 0;
}
