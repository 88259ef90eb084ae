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

/* =====================================================================================================
 * In-loop filters at frame level, driven by REAL side-info: a clip is encoded with the reference
 * encoder (lockstep), after which engine->ctu_info[] holds every decision of the last frame.  The
 * functions below export that side-info as flat per-4x4-unit raster arrays and run the reference's own
 * deblock / SAO / padding code (hmr_deblocking_filter.c:737, hmr_sse42_sao.c:35, hmr_sao.c:1210,
 * hmr_encoder_lib.c:1723) over caller-supplied sample planes, CTU by CTU in the order of the unused
 * frame-level driver hmr_deblock_filter (hmr_deblocking_filter.c:829).
 * ===================================================================================================== */

static int g_w, g_h;

/* (re)open with an explicit configuration; several encoders may be created in one process (the old one leaks) */
int refh_open_cfg(int width, int height, int qp, int sao, int perf, int rd)
{
	HVENC_Cfg c;
	int fd_out, fd_null, ok;
	fflush(stdout);
	fd_out = dup(1);
	fd_null = open("/dev/null", 1);
	dup2(fd_null, 1);
	memset(&c, 0, sizeof c);
	c.size = sizeof c;
	c.width = width; c.height = height; c.profile = PROFILE_MAIN;
	c.gop_size = 1; c.num_b = 0; c.intra_period = 100; c.qp = qp;
	c.bitrate_mode = BR_FIXED_QP; c.bitrate = 1000; c.vbv_size = 1000; c.vbv_init = 350;
	c.wfpp_num_threads = 1; c.wfpp_enable = 1; c.num_enc_engines = 1;
	c.sample_adaptive_offset = sao; c.performance_mode = perf; c.rd_mode = rd;
	c.max_intra_tr_depth = 2; c.max_inter_tr_depth = 1;
	c.motion_estimation_precision = QUARTER_PEL; c.frame_rate = 25;
	c.num_ref_frames = 1; c.cu_size = 64; c.max_pred_partition_depth = 4;
	c.sign_hiding = 1; c.chroma_qp_offset = 2; c.reinit_gop_on_scene_change = 1;
	g_handle = HOMER_enc_init();
	g_enc = (hvenc_enc_t *)g_handle;
	ok = HOMER_enc_control(g_handle, HOMER_SETCFG, &c);
	fflush(stdout);
	dup2(fd_out, 1);
	close(fd_out);
	close(fd_null);
	if (!ok) return -1;
	g_eng = g_enc->encoder_engines[0];
	g_et = g_eng->thread[0];
	g_w = width; g_h = height;
	memset(&g_ctu, 0, sizeof g_ctu);
	g_ctu.top = g_ctu.left = 1;
	return 0;
}

/* feed one frame, wait for its NALs; returns the Annex-B byte count (stream copied to out if it fits) */
int refh_encode_frame(uint8_t *y, uint8_t *u, uint8_t *v, int force_intra, uint8_t *out, int out_cap, int *frame_type)
{
	static uint8_t *buf;
	encoder_in_out_t inf, os, rec;
	nalu_t *nal[8];
	unsigned nn = 0;
	int fd_out, fd_null;
	if (!buf) buf = malloc(0x2000000);
	memset(&inf, 0, sizeof inf); memset(&os, 0, sizeof os); memset(&rec, 0, sizeof rec);
	os.stream.streams[0] = buf;
	inf.stream.streams[0] = y; inf.stream.streams[1] = u; inf.stream.streams[2] = v;
	inf.stream.data_stride[0] = g_w; inf.stream.data_stride[1] = inf.stream.data_stride[2] = g_w / 2;
	inf.image_type = force_intra ? IMAGE_I : IMAGE_AUTO;
	fflush(stdout);
	fd_out = dup(1); fd_null = open("/dev/null", 1); dup2(fd_null, 1);
	HOMER_enc_encode(g_handle, &inf);
	for (;;) {
		HOMER_enc_get_coded_frame(g_handle, &rec, nal, &nn);
		if (nn) break;
		usleep(100);
	}
	HOMER_enc_write_annex_b_output(nal, nn, &os);
	fflush(stdout);
	dup2(fd_out, 1); close(fd_out); close(fd_null);
	if (frame_type) *frame_type = rec.image_type;
	if (out && os.stream.data_size[0] <= out_cap) memcpy(out, buf, os.stream.data_size[0]);
	return os.stream.data_size[0];
}

void refh_pict_geometry(int *geo)
{
	geo[0] = g_eng->pict_width_in_ctu; geo[1] = g_eng->pict_height_in_ctu;
	geo[2] = g_eng->curr_reference_frame->img.window_size_x[0]; geo[3] = g_eng->curr_reference_frame->img.window_size_x[1];
	geo[4] = g_eng->curr_reference_frame->img.data_padding_x[0]; geo[5] = g_eng->curr_reference_frame->img.data_padding_y[0];
	geo[6] = g_eng->curr_reference_frame->img.data_padding_x[1]; geo[7] = g_eng->curr_reference_frame->img.data_padding_y[1];
	geo[8] = g_eng->current_pict.slice.slice_type == I_SLICE;
	geo[9] = g_eng->current_pict.slice.pps->cb_qp_offset; geo[10] = g_eng->current_pict.slice.pps->cr_qp_offset;
	geo[11] = g_eng->current_pict.slice.slice_beta_offset_div2; geo[12] = g_eng->current_pict.slice.slice_tc_offset_div2;
	geo[13] = g_eng->current_pict.slice.deblocking_filter_disabled_flag;
}

/* per-4x4-unit side info of the last encoded frame, raster order over (ctus_x*16) x (ctus_y*16) units */
void refh_get_unit_info(int16_t *mvx, int16_t *mvy, int8_t *ref_idx, uint8_t *qp, uint8_t *pred_mode, uint8_t *cbf_y, uint8_t *pred_depth,
			uint8_t *tr_idx, uint8_t *part_size, uint8_t *skipped)
{
	int W4 = g_eng->pict_width_in_ctu * 16, c, a;
	for (c = 0; c < g_eng->pict_total_ctu; c++) {
		ctu_info_t *ctu = &g_eng->ctu_info[c];
		int cx = c % g_eng->pict_width_in_ctu, cy = c / g_eng->pict_width_in_ctu;
		for (a = 0; a < 256; a++) {
			int r = g_enc->abs2raster_table[a];
			size_t o = (size_t)(cy * 16 + r / 16) * W4 + cx * 16 + r % 16;
			mvx[o] = (int16_t)ctu->mv_ref[0][a].hor_vector;
			mvy[o] = (int16_t)ctu->mv_ref[0][a].ver_vector;
			ref_idx[o] = ctu->mv_ref_idx[0][a];
			qp[o] = ctu->qp[a];
			pred_mode[o] = ctu->pred_mode[a];
			cbf_y[o] = CBF(ctu, a, Y_COMP, ctu->tr_idx[a]);
			pred_depth[o] = ctu->pred_depth[a];
			tr_idx[o] = ctu->tr_idx[a];
			part_size[o] = ctu->part_size_type[a];
			skipped[o] = ctu->skipped[a];
		}
	}
}

static void planes_in(wnd_t *w, int16_t *y, int16_t *u, int16_t *v)
{
	int16_t *src[3] = {y, u, v};
	int comp, j;
	for (comp = 0; comp < 3; comp++) {
		int pw = comp ? g_w / 2 : g_w, ph = comp ? g_h / 2 : g_h;
		for (j = 0; j < ph; j++) memcpy((int16_t *)w->pwnd[comp] + (size_t)j * w->window_size_x[comp], src[comp] + (size_t)j * pw, (size_t)pw * 2);
	}
}
static void planes_out(wnd_t *w, int16_t *y, int16_t *u, int16_t *v)
{
	int16_t *dst[3] = {y, u, v};
	int comp, j;
	for (comp = 0; comp < 3; comp++) {
		int pw = comp ? g_w / 2 : g_w, ph = comp ? g_h / 2 : g_h;
		for (j = 0; j < ph; j++) memcpy(dst[comp] + (size_t)j * pw, (int16_t *)w->pwnd[comp] + (size_t)j * w->window_size_x[comp], (size_t)pw * 2);
	}
}

/* planes are dense picture-size int16 arrays (in/out); bs_ver/bs_hor receive the boundary strengths the reference derived */
void refh_deblock_frame(int16_t *y, int16_t *u, int16_t *v, uint8_t *bs_ver, uint8_t *bs_hor)
{
	slice_t *slice = &g_eng->current_pict.slice;
	wnd_t *img = &g_eng->curr_reference_frame->img;
	int W4 = g_eng->pict_width_in_ctu * 16, dir, c, a;
	planes_in(img, y, u, v);
	for (dir = EDGE_VER; dir <= EDGE_HOR; dir++)
		for (c = 0; c < g_eng->pict_total_ctu; c++) {
			ctu_info_t *ctu = &g_eng->ctu_info[c];
			uint8_t *bs = dir == EDGE_VER ? bs_ver : bs_hor;
			int cx = c % g_eng->pict_width_in_ctu, cy = c / g_eng->pict_width_in_ctu;
			create_partition_ctu_neighbours(g_et, ctu, ctu->partition_list);
			hmr_deblock_filter_cu(g_et, slice, ctu, dir);
			if (bs)
				for (a = 0; a < 256; a++) {
					int r = g_enc->abs2raster_table[a];
					bs[(size_t)(cy * 16 + r / 16) * W4 + cx * 16 + r % 16] =
						g_et->deblock_edge_filter[dir][a] ? (uint8_t)(0x80 | g_et->deblock_filter_strength_bs[dir][a]) : 0;
				}
		}
	planes_out(img, y, u, v);
}

/* stats[ctu][comp][type][0=diff,1=count][32] as int64 */
void refh_sao_stats_frame(int16_t *oy, int16_t *ou, int16_t *ov, int16_t *ry, int16_t *ru, int16_t *rv, int64_t *stats)
{
	slice_t *slice = &g_eng->current_pict.slice;
	int c, comp, t;
	planes_in(&g_eng->current_pict.img2encode->img, oy, ou, ov);
	planes_in(&g_eng->curr_reference_frame->img, ry, ru, rv);
	for (c = 0; c < g_eng->pict_total_ctu; c++) {
		ctu_info_t *ctu = &g_eng->ctu_info[c];
		memset(&ctu->stat_data[0][0], 0, sizeof(ctu->stat_data));
		sse_sao_get_ctu_stats(g_et, slice, ctu, ctu->stat_data);
		for (comp = 0; comp < 3; comp++)
			for (t = 0; t < NUM_SAO_NEW_TYPES; t++) {
				int64_t *o = stats + ((((size_t)c * 3 + comp) * NUM_SAO_NEW_TYPES + t) * 2) * 32;
				memcpy(o, ctu->stat_data[comp][t].diff, 32 * 8);
				memcpy(o + 32, ctu->stat_data[comp][t].count, 32 * 8);
			}
	}
}

/* params[ctu][comp][34] = {modeIdc, typeIdc, offset[32]}; planes in/out.  Source = pre-SAO copy (sao_aux_wnd), hmr_sao.c:1435 */
void refh_sao_apply_frame(int16_t *y, int16_t *u, int16_t *v, int32_t *params)
{
	int c, comp, k;
	planes_in(&g_eng->curr_reference_frame->img, y, u, v);
	for (c = 0; c < g_eng->pict_total_ctu; c++) {
		ctu_info_t *ctu = &g_eng->ctu_info[c];
		wnd_copy_ctu(sse_copy_16_16, &g_eng->curr_reference_frame->img, &g_eng->sao_aux_wnd, ctu);
		reference_picture_border_padding_ctu(&g_eng->sao_aux_wnd, ctu);
	}
	for (c = 0; c < g_eng->pict_total_ctu; c++) {
		ctu_info_t *ctu = &g_eng->ctu_info[c];
		sao_blk_param_t p;
		memset(&p, 0, sizeof p);
		for (comp = 0; comp < 3; comp++) {
			int32_t *src = params + ((size_t)c * 3 + comp) * 34;
			p.offsetParam[comp].modeIdc = src[0];
			p.offsetParam[comp].typeIdc = src[1];
			for (k = 0; k < 32; k++) p.offsetParam[comp].offset[k] = src[2 + k];
		}
		sao_offset_ctu(g_et, ctu, &p);
	}
	planes_out(&g_eng->curr_reference_frame->img, y, u, v);
}

/* the SAO parameters the encoder decided for the last frame, same layout as above */
void refh_get_sao_params(int32_t *params)
{
	int c, comp, k;
	for (c = 0; c < g_eng->pict_total_ctu; c++)
		for (comp = 0; comp < 3; comp++) {
			sao_offset_t *o = &g_eng->ctu_info[c].recon_params.offsetParam[comp];
			int32_t *dst = params + ((size_t)c * 3 + comp) * 34;
			dst[0] = o->modeIdc; dst[1] = o->typeIdc;
			for (k = 0; k < 32; k++) dst[2 + k] = o->offset[k];
		}
}

/* picture planes in; padded planes out (full allocated window incl. margins: stride x (h + 2*pad_y) per component) */
void refh_pad_frame(int16_t *y, int16_t *u, int16_t *v, int16_t *py, int16_t *pu, int16_t *pv)
{
	wnd_t *img = &g_eng->curr_reference_frame->img;
	int16_t *out[3] = {py, pu, pv};
	int c, comp, j;
	for (comp = 0; comp < 3; comp++) {   /* poison the margins so that missing writes show */
		int16_t *base = (int16_t *)img->palloc[comp];
		for (j = 0; j < img->window_size_y[comp] * img->window_size_x[comp]; j++) base[j] = 0x1234;
	}
	planes_in(img, y, u, v);
	for (c = 0; c < g_eng->pict_total_ctu; c++) reference_picture_border_padding_ctu(img, &g_eng->ctu_info[c]);
	for (comp = 0; comp < 3; comp++) {
		memcpy(out[comp], img->palloc[comp], (size_t)img->window_size_y[comp] * img->window_size_x[comp] * 2);
	}
}
/* reconstruction of the last encoded frame as the encoder left it (deblocked, SAO'd), dense int16 planes */
void refh_get_recon(int16_t *y, int16_t *u, int16_t *v) { planes_out(&g_eng->curr_reference_frame->img, y, u, v); }

/* =====================================================================================================
 * Motion: hmr_motion_compensation_luma/chroma (hmr_motion_inter.c:1779,1860) and the search driver
 * hmr_motion_estimation (:1404) with its sub-pel plane builders (:395,442), called with a synthetic
 * partition node at window position (0,0).
 * ===================================================================================================== */
uint32_t hmr_motion_estimation(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *curr_cu_info, int16_t *orig_buff, int orig_buff_stride,
			       int16_t *reference_buff, int reference_buff_stride, int curr_part_global_x, int curr_part_global_y, int init_x, int init_y,
			       int curr_part_size, int curr_part_size_shift, int search_range_x, int search_range_y, int frame_size_x, int frame_size_y,
			       motion_vector_t *mv, motion_vector_t *subpix_mv, mv_candiate_list_t *amvp_candidate_list, uint32_t threshold, unsigned int action);
void hmr_motion_compensation_luma(henc_thread_t *et, cu_partition_info_t *curr_cu_info, int16_t *reference_buff, int reference_buff_stride, int16_t *pred_buff,
				  int pred_buff_stride, int width, int height, int curr_part_size_shift, motion_vector_t *mv, int is_bi_predict);
void hmr_motion_compensation_chroma(henc_thread_t *et, int16_t *reference_buff, int reference_buff_stride, int16_t *pred_buff, int pred_buff_stride,
				    int curr_part_size, int curr_part_size_shift, motion_vector_t *mv, int is_bi_predict);

static int log2i(int n) { int s = 0; while ((1 << s) < n) s++; return s; }

void refh_mc_luma(int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int width, int height, int mvx, int mvy, int is_bi)
{
	cu_partition_info_t cu;
	motion_vector_t mv = {mvx, mvy};
	memset(&cu, 0, sizeof cu);
	hmr_motion_compensation_luma(g_et, &cu, ref, ref_stride, pred, pred_stride, width, height, log2i(width), &mv, is_bi);
}
void refh_mc_chroma(int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int size, int mvx, int mvy, int is_bi)
{
	motion_vector_t mv = {mvx, mvy};
	hmr_motion_compensation_chroma(g_et, ref, ref_stride, pred, pred_stride, size, log2i(size), &mv, is_bi);
}
uint32_t refh_motion_estimation(int16_t *orig, int orig_stride, int16_t *ref, int ref_stride, int gx, int gy, int init_x, int init_y, int size, int range_x,
				int range_y, int frame_w, int frame_h, int32_t *amvp, int n_amvp, int32_t *search, int n_search, int qp, double avg_dist,
				int action, int32_t *out)
{
	cu_partition_info_t cu;
	ctu_info_t ctu;
	mv_candiate_list_t al;
	motion_vector_t mv = {init_x << 2, init_y << 2}, sub = {0, 0};
	uint32_t r;
	int i;
	memset(&cu, 0, sizeof cu);
	memset(&ctu, 0, sizeof ctu);
	memset(&al, 0, sizeof al);
	cu.size = size; cu.qp = qp;
	al.num_mv_candidates = n_amvp;
	for (i = 0; i < n_amvp; i++) { al.mv_candidates[i].mv.hor_vector = amvp[2 * i]; al.mv_candidates[i].mv.ver_vector = amvp[2 * i + 1]; }
	g_et->mv_search_candidates.num_mv_candidates = n_search;
	for (i = 0; i < n_search; i++) {
		g_et->mv_search_candidates.mv_candidates[i].mv.hor_vector = search[2 * i];
		g_et->mv_search_candidates.mv_candidates[i].mv.ver_vector = search[2 * i + 1];
	}
	g_eng->avg_dist = avg_dist;
	r = hmr_motion_estimation(g_et, &ctu, &cu, orig, orig_stride, ref, ref_stride, gx, gy, init_x, init_y, size, log2i(size), range_x, range_y, frame_w,
				  frame_h, &mv, &sub, &al, 0, (unsigned)action);
	out[0] = mv.hor_vector; out[1] = mv.ver_vector; out[2] = sub.hor_vector; out[3] = sub.ver_vector;
	return r;
}

/* the per-TU call sequence of encode_intra_cu (hmr_motion_intra.c:1030-1068) issued through the reference's own table */
uint32_t refh_tu_chain(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size, int is_dst,
		       int scan_mode, int comp, int is_intra, int slice_is_intra, int sign_hiding, int per, int rem, int *ac_sum)
{
	static int16_t *res, *coef, *deq, *zeros;
	low_level_funcs_t *f = &g_enc->funcs;
	int sh = log2i(size), depth = 6 - sh - (comp != 0);
	if (!res) {
		res = hmr_aligned_alloc(64 * 64, 2); coef = hmr_aligned_alloc(32 * 32, 2); deq = hmr_aligned_alloc(32 * 32, 2); zeros = hmr_aligned_alloc(64 * 64, 2);
		memset(zeros, 0, 64 * 64 * 2);
	}
	g_eng->current_pict.slice.slice_type = slice_is_intra ? I_SLICE : P_SLICE;
	g_et->pps->sign_data_hiding_flag = sign_hiding;
	f->predict(orig, orig_stride, pred, pred_stride, res, 64, size);
	f->transform(8, res, coef, 64, size, size, sh, sh, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
	f->quant(g_et, coef, levels, scan_mode, depth, comp, 0, is_intra, ac_sum, size, per, rem);
	if (*ac_sum) {
		f->inv_quant(g_et, levels, deq, depth, comp, is_intra, size, per, rem);
		f->itransform(8, res, deq, 64, size, size, is_dst ? 0 : REG_DCT, g_et->pred_aux_buff);
		f->reconst(pred, pred_stride, res, 64, recon, recon_stride, size);
	} else
		f->reconst(pred, pred_stride, zeros, 0, recon, recon_stride, size);
	return f->ssd16b(orig, orig_stride, recon, recon_stride, size);
}

/* homer_loop1_motion_intra (hmr_motion_intra.c:1084) on flat arguments.  The most-probable-mode list is derived by the reference
 * itself from the neighbour modes given here (left_mode / top_mode, -1 = neighbour not intra): the PU is placed at 4x4-unit
 * (1,1) of the CTU so that both neighbours are read from ctu_rd (hmr_arithmetic_encoding.c:229,282).  rd_mode 0 (RD_DIST_ONLY)
 * or 2 (RD_FAST).  out = {best mode, bit cost, num_preds, preds[3]}. */
int homer_loop1_motion_intra(henc_thread_t *et, ctu_info_t *ctu, ctu_info_t *ctu_rd, cu_partition_info_t *curr_partition_info, int16_t *pred_buff,
			     int pred_buff_stride, int16_t *orig_buff, int orig_buff_stride, int16_t *decoded_buff, int decoded_buff_stride, int depth,
			     int curr_depth, int curr_part_size, int curr_part_size_shift, int part_size_type, int curr_adi_size, int best_pred_modes[3],
			     double best_pred_cost[3]);
int get_intra_dir_luma_predictor(ctu_info_t *ctu, cu_partition_info_t *curr_partition_info, int *arr_intra_dir, int *piMode);
void refh_intra_search(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int n, int left, int top, int bottom_left, int top_right,
		       int bl_size, int tr_size, int strong_enabled, int left_mode, int top_mode, int rd_mode, double sqrt_lambda, int16_t *adi_out,
		       int16_t *adi_filtered_out, int16_t *pred_out, int pred_stride, int32_t *out, double *best_cost)
{
	static uint8_t pred_mode[MAX_NUM_PARTITIONS];
	cu_partition_info_t pi;
	ctu_info_t ctu, ctu_rd;
	const int sh = log2i(n), curr_depth = 6 - sh, adi_size = 4 * n + 1;
	int save_w = g_et->pict_width[0], save_h = g_et->pict_height[0], save_rd = g_et->rd_mode, save_strong = g_et->sps->strong_intra_smooth_enabled_flag;
	double save_lambda = g_et->rd.sqrt_lambda;
	int modes[3], preds[3] = {-1, -1, -1}, np;
	double costs[3];
	memset(&pi, 0, sizeof pi);
	memset(&ctu, 0, sizeof ctu);
	memset(&ctu_rd, 0, sizeof ctu_rd);
	pi.left_neighbour = left; pi.top_neighbour = top;
	pi.left_bottom_neighbour = bottom_left; pi.top_right_neighbour = top_right;
	pi.depth = curr_depth;
	pi.size = n;
	pi.raster_index = 17;                   /* unit (1,1): left and top neighbours lie inside this CTU */
	pi.abs_index_left_partition = 1;
	pi.abs_index_top_partition = 2;
	ctu.size = ctu_rd.size = 64;
	ctu_rd.pred_mode = pred_mode;
	pred_mode[1] = left_mode >= 0 ? INTRA_MODE : INTER_MODE;
	pred_mode[2] = top_mode >= 0 ? INTRA_MODE : INTER_MODE;
	g_et->intra_mode_buffs[Y_COMP][curr_depth][1] = (uint8_t)(left_mode >= 0 ? left_mode : 0);
	g_et->intra_mode_buffs[Y_COMP][curr_depth][2] = (uint8_t)(top_mode >= 0 ? top_mode : 0);
	g_et->pict_width[0] = n + tr_size; g_et->pict_height[0] = n + bl_size;
	g_et->rd_mode = rd_mode;
	g_et->rd.sqrt_lambda = sqrt_lambda;
	g_et->sps->strong_intra_smooth_enabled_flag = strong_enabled;
	out[1] = homer_loop1_motion_intra(g_et, &ctu, &ctu_rd, &pi, pred_out, pred_stride, orig, orig_stride, decoded_corner + decoded_stride + 1, decoded_stride,
					  curr_depth, curr_depth, n, sh, SIZE_2Nx2N, adi_size, modes, costs);
	out[0] = modes[0];
	*best_cost = costs[0];
	np = get_intra_dir_luma_predictor(&ctu_rd, &pi, preds, NULL);
	out[2] = np; out[3] = preds[0]; out[4] = preds[1]; out[5] = preds[2];
	memcpy(adi_out, g_et->adi_pred_buff, (size_t)adi_size * 2);
	memcpy(adi_filtered_out, g_et->adi_filtered_pred_buff, (size_t)adi_size * 2);
	g_et->pict_width[0] = save_w; g_et->pict_height[0] = save_h;
	g_et->rd_mode = save_rd; g_et->rd.sqrt_lambda = save_lambda;
	g_et->sps->strong_intra_smooth_enabled_flag = save_strong;
}

/* encode_intra_cu's data path (hmr_motion_intra.c:1011-1068) issued through the reference's own fill_reference_samples and table */
uint32_t refh_intra_tu_chain(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int left, int top, int bottom_left, int top_right,
			     int bl_size, int tr_size, int strong_enabled, int is_filtered, int mode, int is_luma, int16_t *pred, int pred_stride, int16_t *levels,
			     int16_t *recon, int recon_stride, int size, int is_dst, int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem,
			     int *ac_sum)
{
	cu_partition_info_t pi;
	ctu_info_t ctu;
	const int sh = log2i(size), adi_size = 4 * size + 1;
	int save_w = g_et->pict_width[0], save_h = g_et->pict_height[0], save_strong = g_et->sps->strong_intra_smooth_enabled_flag;
	int16_t *adi;
	memset(&pi, 0, sizeof pi);
	memset(&ctu, 0, sizeof ctu);
	pi.left_neighbour = left; pi.top_neighbour = top;
	pi.left_bottom_neighbour = bottom_left; pi.top_right_neighbour = top_right;
	pi.depth = 6 - sh;
	g_et->pict_width[0] = size + tr_size; g_et->pict_height[0] = size + bl_size;
	g_et->sps->strong_intra_smooth_enabled_flag = strong_enabled;
	fill_reference_samples(g_et, &ctu, &pi, adi_size, decoded_corner, decoded_stride, size, Y_COMP, is_filtered);
	g_et->pict_width[0] = save_w; g_et->pict_height[0] = save_h;
	g_et->sps->strong_intra_smooth_enabled_flag = save_strong;
	adi = is_filtered ? g_et->adi_filtered_pred_buff : g_et->adi_pred_buff;
	if (mode == PLANAR_IDX) g_enc->funcs.create_intra_planar_prediction(g_et, pred, pred_stride, adi, adi_size, size, sh);
	else g_enc->funcs.create_intra_angular_prediction(g_et, &ctu, pred, pred_stride, adi, adi_size, size, mode, is_luma);
	return refh_tu_chain(orig, orig_stride, pred, pred_stride, levels, recon, recon_stride, size, is_dst, scan_mode, comp, 1, slice_is_intra, sign_hiding, per, rem, ac_sum);
}

/* abs2raster_table (hmr_encoder_lib.c:95-100, g_auiZscanToRaster): z-order index of a 4x4 unit inside a 64x64 CTU -> raster index (16 per row) */
void refh_abs2raster(int32_t *out256)
{
	int a;
	for (a = 0; a < 256; a++) out256[a] = g_enc->abs2raster_table[a];
}

/* encode_inter_cu / encode_inter_cu_chroma's per-TU sequence (hmr_motion_inter.c:82-128, 185-227) issued through the reference's own table, with the
 * reference's own expressions for the keep-or-drop decision */
uint32_t refh_inter_tu_chain(int16_t *residual, int residual_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size,
			     int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem, double weight, double zero_thr, int *ac_sum)
{
	static int16_t *coef, *deq, *res_dec, *zeros;
	low_level_funcs_t *f = &g_enc->funcs;
	int sh = log2i(size), depth = 6 - sh - (comp != 0);
	if (!coef) {
		coef = hmr_aligned_alloc(32 * 32, 2); deq = hmr_aligned_alloc(32 * 32, 2); res_dec = hmr_aligned_alloc(64 * 64, 2); zeros = hmr_aligned_alloc(64 * 64, 2);
		memset(zeros, 0, 64 * 64 * 2);
	}
	g_eng->current_pict.slice.slice_type = slice_is_intra ? I_SLICE : P_SLICE;
	g_et->pps->sign_data_hiding_flag = sign_hiding;
	f->transform(8, residual, coef, residual_stride, size, size, sh, sh, REG_DCT, g_et->pred_aux_buff);
	f->quant(g_et, coef, levels, scan_mode, depth, comp, REG_DCT, 0, ac_sum, size, per, rem);
	if (comp == 0) {                    /* :93-127, `int ssd_` */
		int ssd_;
		if (*ac_sum > 0) {
			uint32_t ssd_zero = f->ssd16b(residual, residual_stride, zeros, 0, size);
			f->inv_quant(g_et, levels, deq, depth, comp, 0, size, per, rem);
			f->itransform(8, res_dec, deq, residual_stride, size, size, REG_DCT, g_et->pred_aux_buff);
			ssd_ = f->ssd16b(residual, residual_stride, res_dec, residual_stride, size);
			if (ssd_zero <= ssd_ + zero_thr * (*ac_sum)) {
				memset(levels, 0, size * size * sizeof(levels[0]));
				*ac_sum = 0;
				f->reconst(pred, pred_stride, levels, 0, recon, recon_stride, size);
			} else
				f->reconst(pred, pred_stride, res_dec, residual_stride, recon, recon_stride, size);
		} else {
			ssd_ = f->ssd16b(residual, residual_stride, levels, 0, size);
			f->reconst(pred, pred_stride, levels, 0, recon, recon_stride, size);
		}
		return (uint32_t)ssd_;
	} else {                            /* :195-226, `uint32_t ssd_`, weighted */
		uint32_t ssd_;
		if (*ac_sum > 0) {
			uint32_t ssd_zero = (uint32_t)(weight * f->ssd16b(residual, residual_stride, zeros, 0, size));
			f->inv_quant(g_et, levels, deq, depth, comp, 0, size, per, rem);
			f->itransform(8, res_dec, deq, residual_stride, size, size, REG_DCT, g_et->pred_aux_buff);
			ssd_ = (uint32_t)(weight * f->ssd16b(residual, residual_stride, res_dec, residual_stride, size));
			if (ssd_zero <= ssd_ + zero_thr * (*ac_sum)) {
				memset(levels, 0, size * size * sizeof(levels[0]));
				*ac_sum = 0;
				f->reconst(pred, pred_stride, levels, 0, recon, recon_stride, size);
			} else
				f->reconst(pred, pred_stride, res_dec, residual_stride, recon, recon_stride, size);
		} else {
			ssd_ = (uint32_t)(weight * f->ssd16b(residual, residual_stride, levels, 0, size));
			f->reconst(pred, pred_stride, levels, 0, recon, recon_stride, size);
		}
		return ssd_;
	}
}

/* encode_intra_luma (hmr_motion_intra.c:1226-1632) itself, on the encoder instance's own thread context: the luma of one 2Nx2N CU at the origin of CTU 0 -
 * mode search, transform tree (one level with the harness configuration max_intra_tr_depth = 2) and consolidation.  Inputs are written where the
 * encoder keeps them (curr_mbs_wnd, the neighbour L-shape in decoded_mbs_wnd[depth+1] and [depth+2], the neighbour flags on the five partition nodes);
 * outputs are read back from where it leaves them.  nbflags: 5 x {left, top, bottom_left, top_right}; pict_w / pict_h: picture extent measured from
 * the CU's origin (bounds the below-left / above-right runs, :289,335).
 * out: {return value, cost, distortion, sum, cbf of the 4 quadrants (first unit), tr_idx (first unit), ssd[5], sum[5], mode, 0, preds[3]} */
uint32_t encode_intra_luma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize part_size_type);
int refh_intra_luma_cu(int16_t *orig, int16_t *top, int16_t *left, const int32_t *nbflags, int pict_w, int pict_h, int size, int qp, double sqrt_lambda, int rd_mode,
		       int slice_is_intra, int sign_hiding, int strong_enabled, int32_t *out, int16_t *dec_par, int16_t *dec_chl, int16_t *lev_par, int16_t *lev_chl)
{
	henc_thread_t *et = g_et;
	ctu_info_t *ctu = &g_eng->ctu_info[0];
	const int depth = 6 - log2i(size), h = size / 2;
	cu_partition_info_t *pi = &ctu->partition_list[et->partition_depth_start[depth]], *node[5];
	const int save_w = et->pict_width[0], save_h = et->pict_height[0], save_rd = et->rd_mode, save_strong = et->sps->strong_intra_smooth_enabled_flag;
	const double save_lambda = et->rd.sqrt_lambda;
	wnd_t *dp = et->decoded_mbs_wnd[depth + 1], *dc = et->decoded_mbs_wnd[depth + 2], *qp_ = et->transform_quant_wnd[depth + 1], *qc = et->transform_quant_wnd[depth + 2];
	int16_t *o = WND_POSITION_2D(int16_t *, et->curr_mbs_wnd, Y_COMP, 0, 0, 0, et->ctu_width);
	const int os = WND_STRIDE_2D(et->curr_mbs_wnd, Y_COMP);
	int k, x, y, w;
	uint32_t ret;
	node[0] = pi;
	for (k = 0; k < 4; k++) node[k + 1] = pi->children[k];
	for (k = 0; k < 5; k++) {
		node[k]->left_neighbour = (uint16_t)nbflags[4 * k]; node[k]->top_neighbour = (uint16_t)nbflags[4 * k + 1];
		node[k]->left_bottom_neighbour = (uint16_t)nbflags[4 * k + 2]; node[k]->top_right_neighbour = (uint16_t)nbflags[4 * k + 3];
		node[k]->cost = node[k]->distortion = node[k]->sum = 0;
	}
	pi->qp = (uint32_t)qp;
	ctu->x[Y_COMP] = ctu->y[Y_COMP] = 0;
	ctu->ctu_left = ctu->ctu_top = NULL;
	et->ctu_rd->ctu_left = et->ctu_rd->ctu_top = NULL;
	et->pict_width[0] = pict_w; et->pict_height[0] = pict_h;
	et->rd_mode = rd_mode; et->rd.sqrt_lambda = sqrt_lambda;
	et->sps->strong_intra_smooth_enabled_flag = strong_enabled;
	g_eng->current_pict.slice.slice_type = slice_is_intra ? I_SLICE : P_SLICE;
	et->pps->sign_data_hiding_flag = sign_hiding;
	for (y = 0; y < size; y++) memcpy(o + y * os, orig + y * size, (size_t)size * 2);
	for (w = 0; w < 2; w++) {
		wnd_t *d = w ? dc : dp;
		int16_t *p = WND_POSITION_2D(int16_t *, *d, Y_COMP, 0, 0, 0, et->ctu_width);
		const int s = WND_STRIDE_2D(*d, Y_COMP);
		for (x = 0; x < 2 * size + 1; x++) p[-s - 1 + x] = top[x];
		for (y = 0; y < 2 * size; y++) p[y * s - 1] = left[y];
		for (y = 0; y < size; y++) for (x = 0; x < size; x++) p[y * s + x] = 0x0101;
	}
	ret = encode_intra_luma(et, ctu, 0, depth, 0, SIZE_2Nx2N);
	{
		int16_t *p = WND_POSITION_2D(int16_t *, *dp, Y_COMP, 0, 0, 0, et->ctu_width), *c = WND_POSITION_2D(int16_t *, *dc, Y_COMP, 0, 0, 0, et->ctu_width);
		const int sp = WND_STRIDE_2D(*dp, Y_COMP), sc = WND_STRIDE_2D(*dc, Y_COMP);
		for (y = 0; y < size; y++) { memcpy(dec_par + y * size, p + y * sp, (size_t)size * 2); memcpy(dec_chl + y * size, c + y * sc, (size_t)size * 2); }
		memcpy(lev_par, WND_POSITION_1D(int16_t *, *qp_, Y_COMP, 0, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift)), (size_t)size * size * 2);
		memcpy(lev_chl, WND_POSITION_1D(int16_t *, *qc, Y_COMP, 0, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift)), (size_t)size * size * 2);
	}
	out[0] = (int32_t)ret; out[1] = (int32_t)pi->cost; out[2] = (int32_t)pi->distortion; out[3] = (int32_t)pi->sum;
	for (k = 0; k < 4; k++) out[4 + k] = et->cbf_buffs[Y_COMP][depth][node[k + 1]->abs_index];
	out[8] = et->tr_idx_buffs[depth][pi->abs_index];
	for (k = 0; k < 5; k++) { out[9 + k] = (int32_t)node[k]->distortion; out[14 + k] = (int32_t)node[k]->sum; }
	out[19] = et->intra_mode_buffs[Y_COMP][depth][pi->abs_index];
	out[20] = h;
	et->pict_width[0] = save_w; et->pict_height[0] = save_h;
	et->rd_mode = save_rd; et->rd.sqrt_lambda = save_lambda;
	et->sps->strong_intra_smooth_enabled_flag = save_strong;
	return 0;
}

/* encode_intra_chroma (hmr_motion_intra_chroma.c:114) itself on the encoder instance's thread context: the chroma of one 2Nx2N CU at the origin of CTU 0.
 * The luma decisions it reads are planted where it looks for them (intra_mode_buffs[Y], tr_idx_buffs), the neighbour L-shapes go into the chroma planes of
 * the last decoded window, the neighbour flags onto the partition nodes.  size = chroma CU size; pict_w / pict_h = chroma picture extent from the CU origin.
 * out: {coded mode, -, bits, -, distortion, sum}; dec_u / dec_v: size x size from the work window (and checked equal to the consolidated one). */
uint32_t encode_intra_chroma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, int part_size_type);
int refh_intra_chroma_cu(int16_t *orig_u, int16_t *orig_v, int16_t *top_u, int16_t *left_u, int16_t *top_v, int16_t *left_v, const int32_t *nbflags, int pict_w,
			 int pict_h, int size, int luma_mode, int split, int qp, int slice_qp, double sqrt_lambda, int rd_mode, int slice_is_intra, int sign_hiding,
			 int32_t *out, int16_t *dec_u, int16_t *dec_v, int16_t *lev_u, int16_t *lev_v)
{
	henc_thread_t *et = g_et;
	ctu_info_t *ctu = &g_eng->ctu_info[0];
	const int depth = 6 - log2i(2 * size);
	cu_partition_info_t *pi = &ctu->partition_list[et->partition_depth_start[depth]], *node[5];
	const int save_w = et->pict_width[1], save_h = et->pict_height[1], save_rd = et->rd_mode, save_qp = g_eng->current_pict.slice.qp;
	const double save_lambda = et->rd.sqrt_lambda, save_avg = g_eng->avg_dist;
	wnd_t *dw = et->decoded_mbs_wnd[NUM_DECODED_WNDS - 1], *dd = et->decoded_mbs_wnd[depth + 1], *qw = et->transform_quant_wnd[NUM_QUANT_WNDS - 1];
	int16_t *orig[2] = {orig_u, orig_v}, *top[2] = {top_u, top_v}, *left[2] = {left_u, left_v}, *dec[2] = {dec_u, dec_v}, *lev[2] = {lev_u, lev_v};
	int k, x, y, c, bits;
	uint32_t ret;
	node[0] = pi;
	for (k = 0; k < 4; k++) node[k + 1] = pi->children[k];
	for (k = 0; k < 5; k++) {
		node[k]->left_neighbour = (uint16_t)nbflags[4 * k]; node[k]->top_neighbour = (uint16_t)nbflags[4 * k + 1];
		node[k]->left_bottom_neighbour = (uint16_t)nbflags[4 * k + 2]; node[k]->top_right_neighbour = (uint16_t)nbflags[4 * k + 3];
		node[k]->sum = 0;
	}
	pi->qp = (uint32_t)qp;
	for (k = 0; k < 3; k++) ctu->x[k] = ctu->y[k] = 0;
	et->pict_width[1] = et->pict_width[2] = pict_w; et->pict_height[1] = et->pict_height[2] = pict_h;
	et->rd_mode = rd_mode; et->rd.sqrt_lambda = sqrt_lambda;
	g_eng->avg_dist = 0;                                    /* calc_mv_correction = qp * .15 */
	g_eng->current_pict.slice.qp = slice_qp;
	g_eng->current_pict.slice.slice_type = slice_is_intra ? I_SLICE : P_SLICE;
	et->pps->sign_data_hiding_flag = sign_hiding;
	memset(&et->intra_mode_buffs[Y_COMP][depth][pi->abs_index], luma_mode, pi->num_part_in_cu);
	memset(&et->tr_idx_buffs[depth][pi->abs_index], split, pi->num_part_in_cu);
	for (c = 0; c < 2; c++) {
		const int comp = U_COMP + c;
		int16_t *o = WND_POSITION_2D(int16_t *, et->curr_mbs_wnd, comp, 0, 0, 0, et->ctu_width);
		int16_t *p = WND_POSITION_2D(int16_t *, *dw, comp, 0, 0, 0, et->ctu_width);
		const int os = WND_STRIDE_2D(et->curr_mbs_wnd, comp), s = WND_STRIDE_2D(*dw, comp);
		for (y = 0; y < size; y++) memcpy(o + y * os, orig[c] + y * size, (size_t)size * 2);
		for (x = 0; x < 2 * size + 1; x++) p[-s - 1 + x] = top[c][x];
		for (y = 0; y < 2 * size; y++) p[y * s - 1] = left[c][y];
		for (y = 0; y < size; y++) for (x = 0; x < size; x++) p[y * s + x] = 0x0101;
	}
	ret = encode_intra_chroma(et, ctu, 0, depth, 0, SIZE_2Nx2N);
	out[0] = et->intra_mode_buffs[CHR_COMP][depth][pi->abs_index];
	bits = out[0] == DM_CHROMA_IDX ? 1 : 12;
	out[2] = bits;
	out[4] = (int32_t)(ret - (uint32_t)(bits * calc_mv_correction(pi->qp, g_eng->avg_dist) + .5));
	out[5] = (int32_t)((split && size > 4) ? pi->sum : pi->sum / 2);
	for (c = 0; c < 2; c++) {
		const int comp = U_COMP + c;
		int16_t *p = WND_POSITION_2D(int16_t *, *dw, comp, 0, 0, 0, et->ctu_width), *q = WND_POSITION_2D(int16_t *, *dd, comp, 0, 0, 0, et->ctu_width);
		const int s = WND_STRIDE_2D(*dw, comp), s2 = WND_STRIDE_2D(*dd, comp);
		for (y = 0; y < size; y++) {
			memcpy(dec[c] + y * size, p + y * s, (size_t)size * 2);
			if (memcmp(p + y * s, q + y * s2, (size_t)size * 2)) return -1;            /* synchronize_motion_buffers_chroma, :453 */
		}
		memcpy(lev[c], WND_POSITION_1D(int16_t *, *qw, comp, 0, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift) >> 2), (size_t)size * size * 2);
	}
	et->pict_width[1] = et->pict_width[2] = save_w; et->pict_height[1] = et->pict_height[2] = save_h;
	et->rd_mode = save_rd; et->rd.sqrt_lambda = save_lambda;
	g_eng->avg_dist = save_avg; g_eng->current_pict.slice.qp = save_qp;
	return 0;
}

/* sao_derive_offsets / sao_invert_quant_offsets / sao_get_distortion (hmr_sao.c:480,592,620) themselves, for the 15 (component, type) pairs of one CTU;
 * stats in the frame array layout [3][5][2][32] int32 */
void sao_derive_offsets(henc_thread_t *wpp_thread, int component, int type_idc, sao_stat_data_t *stats, int *quant_offsets, int *type_aux_info);
void sao_invert_quant_offsets(int component, int type_idc, int typeAuxInfo, int *dstOffsets, int *srcOffsets);
int64_t sao_get_distortion(int typeIdc, int typeAuxInfo, int *invQuantOffset, sao_stat_data_t *stats, int bit_depth);
void refh_sao_offsets_ctu(const int32_t *stats, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist)
{
	double save[3];
	int comp, type, c;
	for (c = 0; c < 3; c++) { save[c] = g_eng->sao_lambdas[c]; g_eng->sao_lambdas[c] = lambdas[c]; }
	for (comp = 0; comp < 3; comp++)
		for (type = 0; type < 5; type++) {
			sao_stat_data_t st;
			int q[MAX_NUM_SAO_CLASSES], inv[MAX_NUM_SAO_CLASSES], a = 0;
			const int32_t *df = stats + ((comp * 5 + type) * 2) * 32;
			for (c = 0; c < 32; c++) { st.diff[c] = df[c]; st.count[c] = df[32 + c]; }
			sao_derive_offsets(g_et, comp, type, &st, q, &a);
			sao_invert_quant_offsets(comp, type, a, inv, q);
			dist[comp * 5 + type] = sao_get_distortion(type, a, inv, &st, g_et->bit_depth);
			aux[comp * 5 + type] = a;
			for (c = 0; c < 32; c++) offsets[(comp * 5 + type) * 32 + c] = inv[c];
		}
	for (c = 0; c < 3; c++) g_eng->sao_lambdas[c] = save[c];
}
