/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * Flat C entry points over the *compiled reference* (oracle/_ref/libhomer_ref.so)
 * so that tests/ and tests/golden/make_golden.py can call the reference's SSE4.2
 * kernels - the parity target, SURVEY.md §0-2 - with plain pointers, without
 * knowing henc_thread_t.  Compiled against the reference's headers where they
 * lie (-I/root/reference/src/homer_lib); nothing of the reference is copied here.
 *
 * A real encoder instance (HOMER_enc_init + HOMER_SETCFG on a small picture,
 * hmr_encoder_lib.c:66,704) supplies the henc_thread_t / tables that five of
 * the table entries dereference (SURVEY.md §8-b "Struct coupling").
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "hmr_private.h"
#include "hmr_common.h"
#include "hmr_sse42_functions.h"

static void *g_handle;
static hvenc_enc_t *g_enc;
static hvenc_engine_t *g_eng;
static henc_thread_t *g_et;
static ctu_info_t g_ctu;

int refh_open(int width, int height)
{
	HVENC_Cfg c;
	int fd_out, fd_null;
	if (g_handle)
		return 0;
	/* the library prints banners on stdout; silence them while we initialise */
	fflush(stdout);
	fd_out = dup(1);
	fd_null = open("/dev/null", 1);
	dup2(fd_null, 1);
	memset(&c, 0, sizeof c);
	c.size = sizeof c;
	c.width = width; c.height = height; c.profile = PROFILE_MAIN;
	c.gop_size = 1; c.num_b = 0; c.intra_period = 100; c.qp = 32;
	c.bitrate_mode = BR_FIXED_QP; c.bitrate = 1000; c.vbv_size = 1000; c.vbv_init = 350;
	c.wfpp_num_threads = 1; c.wfpp_enable = 1; c.num_enc_engines = 1;
	c.sample_adaptive_offset = 1; c.performance_mode = 2; c.rd_mode = 2;
	c.max_intra_tr_depth = 2; c.max_inter_tr_depth = 1;
	c.motion_estimation_precision = QUARTER_PEL; c.frame_rate = 25;
	c.num_ref_frames = 1; c.cu_size = 64; c.max_pred_partition_depth = 4;
	c.sign_hiding = 1; c.chroma_qp_offset = 2; c.reinit_gop_on_scene_change = 1;
	g_handle = HOMER_enc_init();
	g_enc = (hvenc_enc_t *)g_handle;
	int ok = HOMER_enc_control(g_handle, HOMER_SETCFG, &c);
	fflush(stdout);
	dup2(fd_out, 1);
	close(fd_out);
	close(fd_null);
	if (!ok)
		return -1;
	g_eng = g_enc->encoder_engines[0];
	g_et = g_eng->thread[0];
	memset(&g_ctu, 0, sizeof g_ctu);
	g_ctu.top = 1;   /* fill_reference_samples always sets both, hmr_motion_intra.c:257-258 */
	g_ctu.left = 1;
	return 0;
}

/* ---- K6 copies (hmr_sse42_functions_pixel.c:152,236,319) ---- */
void refh_copy_16_16(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { sse_copy_16_16(s, ss, d, ds, h, w); }
void refh_copy_8_16(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { sse_copy_8_16(s, ss, d, ds, h, w); }
void refh_copy_16_8(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { sse_copy_16_8(s, ss, d, ds, h, w); }

/* ---- K1-K5 (hmr_sse42_functions_pixel.c:462,728,817,919,1123) ---- */
uint32_t refh_sad(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n) { return sse_aligned_sad(s, ss, p, ps, n); }
uint32_t refh_ssd16b(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n) { return sse_aligned_ssd16b(s, ss, p, ps, n); }
void refh_predict(int16_t *o, int os, int16_t *p, int ps, int16_t *r, int rs, int n) { sse_aligned_predict(o, os, p, ps, r, rs, n); }
void refh_reconst(int16_t *p, int ps, int16_t *r, int rs, int16_t *d, int ds, int n) { sse_aligned_reconst(p, ps, r, rs, d, ds, n); }
uint32_t refh_modified_variance(int16_t *p, int size, int stride, int modif) { return sse_modified_variance(p, size, stride, modif); }
/* scalar twins (hmr_motion_intra.c:51,125,152,167) - K21 calls scalar sad directly */
uint32_t refh_sad_scalar(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n) { return sad(s, ss, p, ps, n); }

/* ---- K12/K13 (hmr_sse42_functions_transform.c:1670,1700) ---- */
void refh_transform(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	int sh = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
	sse_transform(8, block, coeff, stride, n, n, sh, sh, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
}
void refh_itransform(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	sse_itransform(8, block, coeff, stride, n, n, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
}
void refh_transform_scalar(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	int sh = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
	transform(8, block, coeff, stride, n, n, sh, sh, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
}
void refh_itransform_scalar(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	itransform(8, block, coeff, stride, n, n, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
}

/* ---- K14/K15 (hmr_sse42_functions_quant.c:34,135) ---- */
void refh_quant(int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra,
		int slice_is_intra, int sign_hiding, int *ac_sum, int cu_size, int per, int rem)
{
	g_eng->current_pict.slice.slice_type = slice_is_intra ? I_SLICE : P_SLICE;
	g_et->pps->sign_data_hiding_flag = sign_hiding;
	sse_aligned_quant(g_et, src, dst, scan_mode, depth, comp, 0, is_intra, ac_sum, cu_size, per, rem);
	if (delta_u)
		memcpy(delta_u, g_et->aux_buff, (size_t)cu_size * cu_size * sizeof(int16_t));
}
void refh_inv_quant(int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem)
{
	sse_aligned_inv_quant(g_et, src, dst, depth, comp, is_intra, cu_size, per, rem);
}
/* tables built by HOMER_enc_init (hmr_encoder_lib.c:93-140) */
void refh_get_scan(int scan_mode, int idx, uint32_t *out, int n) { memcpy(out, g_enc->scan_pyramid[scan_mode][idx], (size_t)n * 4); }
void refh_get_quant(int size_idx, int list, int rem, int32_t *q, int32_t *iq, int n)
{
	memcpy(q, g_enc->quant_pyramid[size_idx][list][rem], (size_t)n * 4);
	memcpy(iq, g_enc->dequant_pyramid[size_idx][list][rem], (size_t)n * 4);
}

/* ---- K7/K8 (hmr_sse42_functions_prediction.c:199,926) ---- */
void refh_intra_planar(int16_t *pred, int pred_stride, int16_t *adi, int adi_size, int n)
{
	int sh = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : n == 32 ? 5 : 6;
	sse_create_intra_planar_prediction(g_et, pred, pred_stride, adi, adi_size, n, sh);
}
void refh_intra_angular(int16_t *pred, int pred_stride, int16_t *adi, int adi_size, int n, int mode, int is_luma)
{
	sse_create_intra_angular_prediction(g_et, &g_ctu, pred, pred_stride, adi, adi_size, n, mode, is_luma);
}
/* K19 smoothing (hmr_motion_intra.c:189); depth chosen so that max_cu_size_shift-depth+1 == log2(2n) */
void refh_adi_filter(int16_t *ptr, int16_t *out, int adi_size, int n, int strong)
{
	int l = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : n == 32 ? 5 : 6;
	adi_filter(ptr, out, 6 - l, adi_size, n, 6, strong, 8);
}
/* K19 gather (hmr_motion_intra.c:246): flat flags instead of the partition node; the picture size is
 * chosen so that the reference derives the requested left_bottom_size / top_right_size (:289,335) */
void refh_fill_reference_samples(int16_t *decoded, int stride, int n, int left, int top, int bottom_left, int top_right,
				 int bl_size, int tr_size, int16_t *adi_out)
{
	cu_partition_info_t pi;
	ctu_info_t ctu;
	int adi_size = 4 * n + 1;
	int save_w = g_et->pict_width[0], save_h = g_et->pict_height[0];
	memset(&pi, 0, sizeof pi);
	memset(&ctu, 0, sizeof ctu);
	pi.left_neighbour = left; pi.top_neighbour = top;
	pi.left_bottom_neighbour = bottom_left; pi.top_right_neighbour = top_right;
	g_et->pict_width[0] = n + tr_size; g_et->pict_height[0] = n + bl_size;
	fill_reference_samples(g_et, &ctu, &pi, adi_size, decoded, stride, n, Y_COMP, 0);
	memcpy(adi_out, g_et->adi_pred_buff, (size_t)adi_size * 2);
	g_et->pict_width[0] = save_w; g_et->pict_height[0] = save_h;
}

/* ---- K9-K11 (hmr_sse42_functions_inter_prediction.c:796,818,944) ---- */
void refh_interpolate_luma(int16_t *src, int ss, int16_t *dst, int ds, int frac, int w, int h, int vert, int first, int last)
{
	sse_interpolate_luma(src, ss, dst, ds, frac, w, h, vert, first, last);
}
void refh_interpolate_chroma(int16_t *src, int ss, int16_t *dst, int ds, int frac, int w, int h, int vert, int first, int last)
{
	sse_interpolate_chroma(src, ss, dst, ds, frac, w, h, vert, first, last);
}
void refh_weighted_average(int16_t *a, int as, int16_t *b, int bs, int16_t *d, int ds, int h, int w)
{
	sse_weighted_average_motion(a, as, b, bs, d, ds, h, w, 8);
}
