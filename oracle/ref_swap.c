/*
 * TEST INFRASTRUCTURE - the "swap test" (SURVEY.md §8-V): the reference's own L3 code (quadtree walk, mode
 * decisions, CABAC, all unmodified, from oracle/_ref/libhomer_ref.so) drives the GPU kernels through the
 * drop-in C ABI, and the emitted .265 must be byte-identical to the unswapped run.
 *
 * It is the binding of INTEGRATION.md §1 compiled for real: low_level_funcs_t entries are overwritten after
 * HOMER_enc_init with the hmr_gpu_* entries (adapters for the henc_thread_t* members), and the kernels the
 * reference calls directly are interposed through the PLT (sad, fill_reference_samples).
 * HOMER_SWAP selects what is swapped: "all" (default), "none", or a comma list of member names.
 *
 * Built by oracle/Makefile into oracle/_ref/ref_swap (needs hmr_private.h -> build container only; runs on the
 * GPU box).  Same command line as ref_lockstep.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "hmr_private.h"
#include "hmr_common.h"
#include "homer_gpu.h"

static FILE *g_trace;
static int want(const char *name)
{
	const char *s = getenv("HOMER_SWAP");
	if (g_trace) return 0;             /* trace mode: everything that is not traced runs the reference's own code */
	if (!s || !strcmp(s, "all")) return 1;
	if (!strcmp(s, "none")) return 0;
	if (!strncmp(s, "all,", 4)) {      /* "all,-name,-name": everything except the names listed */
		size_t n = strlen(name);
		const char *p = s;
		while ((p = strstr(p, name))) {
			if (p > s + 1 && p[-1] == '-' && p[-2] == ',' && (p[n] == 0 || p[n] == ',')) return 0;
			p += n;
		}
		return 1;
	}
	{
		size_t n = strlen(name);
		const char *p = s;
		while ((p = strstr(p, name))) {
			if ((p == s || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return 1;
			p += n;
		}
	}
	return 0;
}


/* ---- trace mode (HOMER_TRACE=file [HOMER_TRACE_STRIDE=k]): the interposers of the block drivers run the REFERENCE's own function and log the
 * flat inputs and the outputs it produced, every k-th call - real-encode vectors for the batched GPU entries (tests/golden/make_trace.py).
 * record = int32 kind, nh, h[nh], nd, double d[nd], nb, then nb x {int32 count, int16 data[count]} ---- */
static int g_trace_stride = 1;
static unsigned long g_trace_calls;
typedef struct { int32_t kind, nh, h[40], nd; double d[4]; int32_t nb, cnt[8]; int16_t *data[8]; } trec;
static int tr_pick(void) { return g_trace && (g_trace_calls++ % g_trace_stride) == 0; }
static void tr_blob(trec *t, const int16_t *p, int stride, int w, int h)
{
	int y;
	int16_t *d = malloc((size_t)w * h * 2);
	for (y = 0; y < h; y++) memcpy(d + (size_t)y * w, p + (size_t)y * stride, (size_t)w * 2);
	t->cnt[t->nb] = w * h;
	t->data[t->nb++] = d;
}
/* the neighbours an intra block may read: row 0 (corner + 2n) and column 0 (2n below the corner) of the (2n+1)^2 square, what is not available as 0 */
static void tr_lshape(trec *t, const int16_t *corner, int stride, int n, int left, int top, int bl, int tr, int bl_size, int tr_size)
{
	int16_t *d = calloc((size_t)(4 * n + 1), 2);
	int rows = left ? n + (bl ? bl_size : 0) : 0, cols = top ? n + (tr ? tr_size : 0) : 0, i;
	if (left || top) d[0] = corner[0];
	for (i = 1; i <= cols; i++) d[i] = corner[i];
	for (i = 1; i <= rows; i++) d[2 * n + i] = corner[(size_t)i * stride];
	t->cnt[t->nb] = 4 * n + 1;
	t->data[t->nb++] = d;
}
static void tr_write(trec *t)
{
	int i;
	fwrite(&t->kind, 4, 1, g_trace); fwrite(&t->nh, 4, 1, g_trace); fwrite(t->h, 4, t->nh, g_trace);
	fwrite(&t->nd, 4, 1, g_trace); fwrite(t->d, 8, t->nd, g_trace); fwrite(&t->nb, 4, 1, g_trace);
	for (i = 0; i < t->nb; i++) { fwrite(&t->cnt[i], 4, 1, g_trace); fwrite(t->data[i], 2, t->cnt[i], g_trace); free(t->data[i]); }
	fflush(g_trace);
}
#define TR_H(t, ...) do { int32_t v_[] = {__VA_ARGS__}; memcpy((t)->h + (t)->nh, v_, sizeof v_); (t)->nh += sizeof v_ / 4; } while (0)
enum { TR_INTER_TU = 1, TR_INTRA_TU = 2, TR_INTRA_SEARCH = 3, TR_MC = 4, TR_REF_PLANE = 5, TR_ME = 6 };

#include "../integration/homer_gpu_install.c"   /* the adapters and hmr_gpu_install / hmr_gpu_uninstall: the product-tree file a maintainer compiles */

void lockstep_post_init(void *handle)
{
	int n = 0;
	if (getenv("HOMER_TRACE")) {
		g_trace = fopen(getenv("HOMER_TRACE"), "wb");
		if (getenv("HOMER_TRACE_STRIDE")) g_trace_stride = atoi(getenv("HOMER_TRACE_STRIDE"));
		fprintf(stderr, "ref_swap: trace mode, table untouched\n");
		return;
	}
	n = hmr_gpu_install(handle, want);
	fprintf(stderr, "ref_swap: %d table entries routed to libhomer_gpu.so\n", n);
}

/* ---- off-table kernels, interposed through the PLT ---- */
#define REAL(name) ({ static void *p_; if (!p_) p_ = dlsym(RTLD_NEXT, #name); p_; })

uint32_t sad(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n)   /* hmr_motion_inter.c:1712,1757 call this directly */
{
	if (want("sad_direct")) return hmr_gpu_sad(s, ss, p, ps, n);
	return ((uint32_t(*)(int16_t *, uint32_t, int16_t *, uint32_t, int))REAL(sad))(s, ss, p, ps, n);
}

void fill_reference_samples(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, int adi_size, int16_t *decoded, int stride, int n, int comp, int is_filtered)
{
	if (!want("fill_reference_samples")) {
		((void (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int16_t *, int, int, int, int))REAL(fill_reference_samples))(et, ctu, pi, adi_size, decoded, stride, n, comp, is_filtered);
		return;
	}
	/* the sizes the reference derives at hmr_motion_intra.c:289,335 */
	int bl = min(n, et->pict_height[comp] - (ctu->y[comp] + (comp == Y_COMP ? pi->y_position : pi->y_position_chroma) + n));
	int tr = min(n, et->pict_width[comp] - (ctu->x[comp] + (comp == Y_COMP ? pi->x_position : pi->x_position_chroma) + n));
	ctu->top = 1;
	ctu->left = 1;
	hmr_gpu_fill_reference_samples(decoded, stride, n, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour, pi->top_right_neighbour, bl, tr, et->adi_pred_buff);
	if (is_filtered) hmr_gpu_adi_filter(et->adi_pred_buff, et->adi_filtered_pred_buff, adi_size, n, et->sps->strong_intra_smooth_enabled_flag);
}

/* ---- L3 helpers that sit directly on the kernels: motion search driver and motion compensation ---- */
uint32_t hmr_motion_estimation(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *cu, int16_t *orig, int orig_stride, int16_t *ref, int ref_stride,
			       int gx, int gy, int init_x, int init_y, int size, int size_shift, int range_x, int range_y, int frame_w, int frame_h,
			       motion_vector_t *mv, motion_vector_t *subpix_mv, mv_candiate_list_t *amvp, uint32_t threshold, unsigned int action)
{
	if (tr_pick()) {
		/* the padded reference picture is logged once per picture, a search record then names it and the PU position */
		static const int16_t *last_plane;
		static int plane_id = -1;
		const int16_t *plane = ref - (ptrdiff_t)gy * ref_stride - gx;
		trec t = {TR_ME};
		int i, na = amvp->num_mv_candidates, ns = et->mv_search_candidates.num_mv_candidates;
		int ix = init_x, iy = init_y;
		uint32_t r_;
		static uint32_t last_poc = 0xffffffffu;
		if (plane != last_plane || et->enc_engine->current_pict.slice.poc != last_poc) {
			trec p = {TR_REF_PLANE};
			plane_id++;
			last_plane = plane; last_poc = et->enc_engine->current_pict.slice.poc;
			TR_H(&p, plane_id, frame_w, frame_h, 80);
			tr_blob(&p, plane - (ptrdiff_t)80 * ref_stride - 80, ref_stride, frame_w + 160, frame_h + 160);
			tr_write(&p);
		}
		if (!(action & MOTION_PEL_MASK)) { ix = mv->hor_vector >> 2; iy = mv->ver_vector >> 2; }
		TR_H(&t, cu->size, plane_id, gx, gy, ix, iy, range_x, range_y, frame_w, frame_h, (int)action, na, ns);
		for (i = 0; i < 2; i++) TR_H(&t, i < na ? amvp->mv_candidates[i].mv.hor_vector : 0, i < na ? amvp->mv_candidates[i].mv.ver_vector : 0);
		for (i = 0; i < 5; i++)
			TR_H(&t, i < ns ? et->mv_search_candidates.mv_candidates[i].mv.hor_vector : 0, i < ns ? et->mv_search_candidates.mv_candidates[i].mv.ver_vector : 0);
		t.nd = 1; t.d[0] = calc_mv_correction(cu->qp, et->enc_engine->avg_dist);
		tr_blob(&t, orig, orig_stride, cu->size, cu->size);
		r_ = ((uint32_t(*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int, int, int, int, int, int, int, int, int, int,
				   motion_vector_t *, motion_vector_t *, mv_candiate_list_t *, uint32_t, unsigned int))REAL(hmr_motion_estimation))(
			et, ctu, cu, orig, orig_stride, ref, ref_stride, gx, gy, init_x, init_y, size, size_shift, range_x, range_y, frame_w, frame_h, mv, subpix_mv, amvp,
			threshold, action);
		TR_H(&t, mv->hor_vector, mv->ver_vector, subpix_mv->hor_vector, subpix_mv->ver_vector, (int32_t)r_);
		tr_write(&t);
		return r_;
	}
	if (!want("motion_estimation"))
		return ((uint32_t(*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int, int, int, int, int, int, int, int, int, int,
				     motion_vector_t *, motion_vector_t *, mv_candiate_list_t *, uint32_t, unsigned int))REAL(hmr_motion_estimation))(
			et, ctu, cu, orig, orig_stride, ref, ref_stride, gx, gy, init_x, init_y, size, size_shift, range_x, range_y, frame_w, frame_h, mv, subpix_mv, amvp,
			threshold, action);
	int32_t a[4] = {0, 0, 0, 0}, s[10], out[4];
	int i, na = amvp->num_mv_candidates, ns = et->mv_search_candidates.num_mv_candidates;
	for (i = 0; i < na && i < 2; i++) { a[2 * i] = amvp->mv_candidates[i].mv.hor_vector; a[2 * i + 1] = amvp->mv_candidates[i].mv.ver_vector; }
	for (i = 0; i < ns && i < 5; i++) { s[2 * i] = et->mv_search_candidates.mv_candidates[i].mv.hor_vector; s[2 * i + 1] = et->mv_search_candidates.mv_candidates[i].mv.ver_vector; }
	/* without the integer stage the reference starts from *mv (hmr_motion_inter.c:1668) */
	if (!(action & MOTION_PEL_MASK)) { init_x = mv->hor_vector >> 2; init_y = mv->ver_vector >> 2; }
	double corr = calc_mv_correction(cu->qp, et->enc_engine->avg_dist);
	uint32_t r = hmr_gpu_motion_estimation(orig, orig_stride, ref, ref_stride, gx, gy, init_x, init_y, cu->size, range_x, range_y, frame_w, frame_h, a, na, s, ns, corr,
					       (int)action, out);
	mv->hor_vector = out[0]; mv->ver_vector = out[1];
	subpix_mv->hor_vector = out[2]; subpix_mv->ver_vector = out[3];
	return r;
}

void hmr_motion_compensation_luma(henc_thread_t *et, cu_partition_info_t *cu, int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int width, int height,
				  int size_shift, motion_vector_t *mv, int is_bi)
{
	if (tr_pick()) {
		trec t = {TR_MC};
		TR_H(&t, 1, width, height, mv->hor_vector & 3, mv->ver_vector & 3, is_bi);
		tr_blob(&t, ref + (ptrdiff_t)((mv->ver_vector >> 2) - 4) * ref_stride + (mv->hor_vector >> 2) - 4, ref_stride, width + 8, height + 8);
		((void (*)(henc_thread_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_luma))(
			et, cu, ref, ref_stride, pred, pred_stride, width, height, size_shift, mv, is_bi);
		tr_blob(&t, pred, pred_stride, width, height);
		tr_write(&t);
		return;
	}
	if (!want("motion_compensation")) {
		((void (*)(henc_thread_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_luma))(
			et, cu, ref, ref_stride, pred, pred_stride, width, height, size_shift, mv, is_bi);
		return;
	}
	hmr_gpu_mc_luma(ref, ref_stride, pred, pred_stride, width, height, mv->hor_vector, mv->ver_vector, is_bi);
}

void hmr_motion_compensation_chroma(henc_thread_t *et, int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int size, int size_shift, motion_vector_t *mv, int is_bi)
{
	if (tr_pick()) {
		trec t = {TR_MC};
		TR_H(&t, 0, size, size, mv->hor_vector & 7, mv->ver_vector & 7, is_bi);
		tr_blob(&t, ref + (ptrdiff_t)((mv->ver_vector >> 3) - 4) * ref_stride + (mv->hor_vector >> 3) - 4, ref_stride, size + 8, size + 8);
		((void (*)(henc_thread_t *, int16_t *, int, int16_t *, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_chroma))(et, ref, ref_stride, pred,
																		   pred_stride, size, size_shift, mv, is_bi);
		tr_blob(&t, pred, pred_stride, size, size);
		tr_write(&t);
		return;
	}
	if (!want("motion_compensation")) {
		((void (*)(henc_thread_t *, int16_t *, int, int16_t *, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_chroma))(et, ref, ref_stride, pred,
																		   pred_stride, size, size_shift, mv, is_bi);
		return;
	}
	hmr_gpu_mc_chroma(ref, ref_stride, pred, pred_stride, size, mv->hor_vector, mv->ver_vector, is_bi);
}

/* ---- intra mode search driver (hmr_motion_intra.c:1084): neighbour arrays, the candidate schedule and its cost comparison in one GPU call.
 * Host side keeps what is control plane: the most-probable-mode list and (RD_FULL) the CABAC bit estimate of its three entries. ---- */
int get_intra_dir_luma_predictor(ctu_info_t *ctu, cu_partition_info_t *curr_partition_info, int *arr_intra_dir, int *piMode);
uint fast_rd_estimate_bits_intra_luma_mode(henc_thread_t *et, cu_partition_info_t *partition_info, uint pred_depth, int dir, int *preds, int num_preds);
static unsigned long g_intra_searches;
int homer_loop1_motion_intra(henc_thread_t *et, ctu_info_t *ctu, ctu_info_t *ctu_rd, cu_partition_info_t *pi, int16_t *pred_buff, int pred_buff_stride,
			     int16_t *orig_buff, int orig_buff_stride, int16_t *decoded_buff, int decoded_buff_stride, int depth, int curr_depth, int size,
			     int size_shift, int part_size_type, int adi_size, int best_pred_modes[3], double best_pred_cost[3])
{
	if (et->rd_mode == RD_FAST && tr_pick()) {
		trec t = {TR_INTRA_SEARCH};
		int32_t tp[3] = {-1, -1, -1};
		int tbl = min(size, et->pict_height[Y_COMP] - (ctu->y[Y_COMP] + pi->y_position + size)), ttr = min(size, et->pict_width[Y_COMP] - (ctu->x[Y_COMP] + pi->x_position + size));
		int r_;
		ctu_rd->intra_mode[Y_COMP] = et->intra_mode_buffs[Y_COMP][curr_depth];
		get_intra_dir_luma_predictor(ctu_rd, pi, (int *)tp, NULL);
		TR_H(&t, size, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour, pi->top_right_neighbour, tbl, ttr, et->sps->strong_intra_smooth_enabled_flag,
		     tp[0], tp[1], tp[2], 1, 1, 1, 12);
		t.nd = 1; t.d[0] = et->rd.sqrt_lambda;
		tr_blob(&t, orig_buff, orig_buff_stride, size, size);
		tr_lshape(&t, decoded_buff - decoded_buff_stride - 1, decoded_buff_stride, size, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour,
			  pi->top_right_neighbour, tbl, ttr);
		r_ = ((int (*)(henc_thread_t *, ctu_info_t *, ctu_info_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int16_t *, int, int, int, int, int, int, int, int *,
			       double *))REAL(homer_loop1_motion_intra))(et, ctu, ctu_rd, pi, pred_buff, pred_buff_stride, orig_buff, orig_buff_stride, decoded_buff,
									   decoded_buff_stride, depth, curr_depth, size, size_shift, part_size_type, adi_size, best_pred_modes,
									   best_pred_cost);
		TR_H(&t, best_pred_modes[0], r_);
		t.nd = 2; t.d[1] = best_pred_cost[0];
		tr_blob(&t, et->adi_pred_buff, 0, 4 * size + 1, 1);
		tr_blob(&t, et->adi_filtered_pred_buff, 0, 4 * size + 1, 1);
		tr_blob(&t, pred_buff, pred_buff_stride, size, size);
		tr_write(&t);
		return r_;
	}
	if (!want("intra_search"))
		return ((int (*)(henc_thread_t *, ctu_info_t *, ctu_info_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int16_t *, int, int, int, int, int, int,
				 int, int *, double *))REAL(homer_loop1_motion_intra))(et, ctu, ctu_rd, pi, pred_buff, pred_buff_stride, orig_buff, orig_buff_stride,
										      decoded_buff, decoded_buff_stride, depth, curr_depth, size, size_shift,
										      part_size_type, adi_size, best_pred_modes, best_pred_cost);
	int32_t preds[3] = {-1, -1, -1}, bits[3] = {0, 0, 0}, out[2];
	int other = 0, np, i;
	double cost;
	int bl = min(size, et->pict_height[Y_COMP] - (ctu->y[Y_COMP] + pi->y_position + size));
	int tr = min(size, et->pict_width[Y_COMP] - (ctu->x[Y_COMP] + pi->x_position + size));
	ctu->top = 1;                                                            /* fill_reference_samples, :256-257 */
	ctu->left = 1;
	ctu_rd->intra_mode[Y_COMP] = et->intra_mode_buffs[Y_COMP][curr_depth];   /* :1102 */
	np = get_intra_dir_luma_predictor(ctu_rd, pi, (int *)preds, NULL);
	if (et->rd_mode == RD_FULL) {
		for (i = 0; i < np; i++) {
			int aux[3] = {preds[0], preds[1], preds[2]};
			bits[i] = (int32_t)fast_rd_estimate_bits_intra_luma_mode(et, pi, depth - (part_size_type == SIZE_NxN), preds[i], aux, np);
		}
		other = 6;
	} else if (et->rd_mode == RD_FAST) {
		bits[0] = bits[1] = bits[2] = 1;
		other = 12;
	}
	hmr_gpu_intra_search(orig_buff, orig_buff_stride, decoded_buff - decoded_buff_stride - 1, decoded_buff_stride, size, pi->left_neighbour, pi->top_neighbour,
			     pi->left_bottom_neighbour, pi->top_right_neighbour, bl, tr, et->sps->strong_intra_smooth_enabled_flag, preds, bits, other,
			     et->rd.sqrt_lambda, et->adi_pred_buff, et->adi_filtered_pred_buff, pred_buff, pred_buff_stride, out, &cost);
	best_pred_modes[0] = out[0];
	best_pred_modes[1] = best_pred_modes[2] = 100;                          /* PRED_MODE_INVALID, :1082 */
	best_pred_cost[0] = cost;
	(void)adi_size; (void)size_shift;
	if (!g_intra_searches++) fprintf(stderr, "ref_swap: intra mode search routed to libhomer_gpu.so\n");
	return out[1];
}

/* ---- in-loop filters at the reference's own call granularity (hmr_deblock_sao_pad_sync_ctu, hmr_encoder_lib.c:2386, drives them CTU by CTU) ---- */
static int16_t *g_mvx, *g_mvy;
static int8_t *g_ref;
static uint8_t *g_qp, *g_fl, *g_pd, *g_tr;
static int g_us;
/* ctu_info_t keeps its side-info in z-order per CTU (hmr_private.h:792-843); the GPU side takes raster arrays over the picture */
static void export_units(hvenc_engine_t *eng, int n)
{
	ctu_info_t *ctu = &eng->ctu_info[n];
	int cx = n % eng->pict_width_in_ctu, cy = n / eng->pict_width_in_ctu, a;
	for (a = 0; a < 256; a++) {
		int r = eng->abs2raster_table[a];
		size_t o = (size_t)(cy * 16 + r / 16) * g_us + cx * 16 + r % 16;
		g_mvx[o] = (int16_t)ctu->mv_ref[0][a].hor_vector;
		g_mvy[o] = (int16_t)ctu->mv_ref[0][a].ver_vector;
		g_ref[o] = ctu->mv_ref_idx[0][a];
		g_qp[o] = ctu->qp[a];
		g_fl[o] = (uint8_t)((ctu->pred_mode[a] == INTRA_MODE ? HMR_GPU_UNIT_INTRA : 0) | (CBF(ctu, a, Y_COMP, ctu->tr_idx[a]) ? HMR_GPU_UNIT_CBF_Y : 0));
		g_pd[o] = ctu->pred_depth[a];
		g_tr[o] = ctu->tr_idx[a];
	}
}
static pthread_mutex_t g_units_lock = PTHREAD_MUTEX_INITIALIZER;   /* WPP threads filter different CTU rows at the same time; the raster copy is shared */
void hmr_deblock_filter_cu(henc_thread_t *et, slice_t *slice, ctu_info_t *ctu, int dir)
{
	hvenc_engine_t *eng = et->enc_engine;
	if (!want("deblock") || slice->deblocking_filter_disabled_flag) {
		((void (*)(henc_thread_t *, slice_t *, ctu_info_t *, int))REAL(hmr_deblock_filter_cu))(et, slice, ctu, dir);
		return;
	}
	pthread_mutex_lock(&g_units_lock);
	if (!g_mvx) {
		size_t n = (size_t)eng->pict_width_in_ctu * 16 * eng->pict_height_in_ctu * 16;
		g_us = eng->pict_width_in_ctu * 16;
		g_mvx = calloc(n, 2); g_mvy = calloc(n, 2); g_ref = calloc(n, 1); g_qp = calloc(n, 1); g_fl = calloc(n, 1); g_pd = calloc(n, 1); g_tr = calloc(n, 1);
		fprintf(stderr, "ref_swap: deblocking routed to libhomer_gpu.so\n");
	}
	{
		int n = ctu->ctu_number, w = eng->pict_width_in_ctu;
		wnd_t *img = &eng->curr_reference_frame->img;
		int16_t *planes[3] = {(int16_t *)img->pwnd[0], (int16_t *)img->pwnd[1], (int16_t *)img->pwnd[2]};
		int strides[3] = {img->window_size_x[0], img->window_size_x[1], img->window_size_x[2]};
		export_units(eng, n);
		if (n % w) export_units(eng, n - 1);
		if (n >= w) export_units(eng, n - w);
		hmr_gpu_deblock_filter_ctu(planes, strides, et->pict_width[Y_COMP], et->pict_height[Y_COMP], g_us, g_mvx, g_mvy, g_ref, g_qp, g_fl, g_pd, g_tr, ctu->x[Y_COMP],
					   ctu->y[Y_COMP], ctu->size, dir, slice->pps->cb_qp_offset, slice->pps->cr_qp_offset, slice->slice_beta_offset_div2,
					   slice->slice_tc_offset_div2);
	}
	pthread_mutex_unlock(&g_units_lock);
}
/* sao_derive_offsets (hmr_sao.c:480): the offsets of one (component, type) from its statistics on the GPU; the rate terms around it (sao_derive_mode_new_rdo)
 * read the CABAC state and stay the reference's */
void sao_derive_offsets(henc_thread_t *et, int component, int type_idc, sao_stat_data_t *stats, int *quant_offsets, int *type_aux_info)
{
	static int said;
	if (!want("sao_offsets") || et->bit_depth != 8) {
		((void (*)(henc_thread_t *, int, int, sao_stat_data_t *, int *, int *))REAL(sao_derive_offsets))(et, component, type_idc, stats, quant_offsets, type_aux_info);
		return;
	}
	int32_t st[3][5][2][32], off[3][5][32], aux[3][5];
	int64_t dist[3][5];
	int c;
	memset(st, 0, sizeof st);
	for (c = 0; c < 32; c++) {
		st[component][type_idc][0][c] = (int32_t)stats->diff[c];
		st[component][type_idc][1][c] = (int32_t)stats->count[c];
	}
	hmr_gpu_sao_offsets_ctu(&st[0][0][0][0], et->enc_engine->sao_lambdas, &off[0][0][0], &aux[0][0], &dist[0][0]);
	for (c = 0; c < MAX_NUM_SAO_CLASSES; c++) quant_offsets[c] = off[component][type_idc][c];
	*type_aux_info = aux[component][type_idc];
	if (!said++) fprintf(stderr, "ref_swap: SAO offset derivation routed to libhomer_gpu.so\n");
}
void sao_offset_ctu(henc_thread_t *et, ctu_info_t *ctu, sao_blk_param_t *p)
{
	static int said;
	if (!want("sao_offset")) {
		((void (*)(henc_thread_t *, ctu_info_t *, sao_blk_param_t *))REAL(sao_offset_ctu))(et, ctu, p);
		return;
	}
	wnd_t *sw = &et->enc_engine->sao_aux_wnd, *dw = &et->enc_engine->curr_reference_frame->img;
	const int16_t *src[3] = {(int16_t *)sw->pwnd[0], (int16_t *)sw->pwnd[1], (int16_t *)sw->pwnd[2]};
	int16_t *dst[3] = {(int16_t *)dw->pwnd[0], (int16_t *)dw->pwnd[1], (int16_t *)dw->pwnd[2]};
	int ss[3] = {sw->window_size_x[0], sw->window_size_x[1], sw->window_size_x[2]}, ds[3] = {dw->window_size_x[0], dw->window_size_x[1], dw->window_size_x[2]};
	int32_t params[3][34];
	int c, k;
	for (c = 0; c < 3; c++) {
		params[c][0] = p->offsetParam[c].modeIdc;
		params[c][1] = p->offsetParam[c].typeIdc;
		for (k = 0; k < 32; k++) params[c][2 + k] = p->offsetParam[c].offset[k];
	}
	hmr_gpu_sao_offset_ctu(src, ss, dst, ds, et->pict_width[Y_COMP], et->pict_height[Y_COMP], ctu->x[Y_COMP], ctu->y[Y_COMP], &params[0][0]);
	if (!said++) fprintf(stderr, "ref_swap: SAO offset routed to libhomer_gpu.so\n");
}
void reference_picture_border_padding_ctu(wnd_t *wnd, ctu_info_t *ctu)
{
	static int said;
	if (!want("pad")) {
		((void (*)(wnd_t *, ctu_info_t *))REAL(reference_picture_border_padding_ctu))(wnd, ctu);
		return;
	}
	int16_t *planes[3] = {(int16_t *)wnd->pwnd[0], (int16_t *)wnd->pwnd[1], (int16_t *)wnd->pwnd[2]};
	int strides[3] = {wnd->window_size_x[0], wnd->window_size_x[1], wnd->window_size_x[2]};
	hmr_gpu_pad_ctu(planes, strides, wnd->data_width[Y_COMP], wnd->data_height[Y_COMP], wnd->data_padding_x[Y_COMP], wnd->data_padding_y[Y_COMP], ctu->x[Y_COMP],
			ctu->y[Y_COMP], ctu->size);
	if (!said++) fprintf(stderr, "ref_swap: border padding routed to libhomer_gpu.so\n");
}

/* ---- encode_intra_cu (hmr_motion_intra.c:970): the luma intra TU - neighbour array, prediction and the whole TU chain in one GPU call; the
 * bookkeeping on the partition node (:1040-1049) stays on the host ---- */
int find_scan_mode(int is_intra, int is_luma, int width, int dir_mode, int up_left_luma_dir_mode);
uint encode_intra_cu(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, int depth, int cu_mode, PartSize part_size_type, int *curr_sum, int gcnt)
{
	static int said;
	const int traced = tr_pick();
	if (!traced && !want("intra_tu_chain"))
		return ((uint (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_intra_cu))(et, ctu, pi, depth, cu_mode,
															       part_size_type, curr_sum, gcnt);
	static const uint8_t filter_thr[5] = {10, 7, 1, 0, 10};                 /* intra_filter, :148 */
	int curr_depth = pi->depth, x = pi->x_position, y = pi->y_position, size = pi->size;
	int scan_mode = find_scan_mode(TRUE, TRUE, size, cu_mode, 0);
	int per = pi->qp / 6, rem = pi->qp % 6;
	wnd_t *quant_wnd = et->transform_quant_wnd[curr_depth + 1], *decoded_wnd = et->decoded_mbs_wnd[curr_depth + 1];
	int pred_stride = WND_STRIDE_2D(et->prediction_wnd[0], Y_COMP), orig_stride = WND_STRIDE_2D(et->curr_mbs_wnd, Y_COMP), dec_stride = WND_STRIDE_2D(*decoded_wnd, Y_COMP);
	int16_t *pred = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], Y_COMP, x, y, gcnt, et->ctu_width);
	int16_t *orig = WND_POSITION_2D(int16_t *, et->curr_mbs_wnd, Y_COMP, x, y, gcnt, et->ctu_width);
	int16_t *dec = WND_POSITION_2D(int16_t *, *decoded_wnd, Y_COMP, x, y, gcnt, et->ctu_width);
	int16_t *quant = WND_POSITION_1D(int16_t *, *quant_wnd, Y_COMP, gcnt, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift));
	int inv_depth = et->max_cu_size_shift - curr_depth;
	int diff = min(abs(cu_mode - HOR_IDX), abs(cu_mode - VER_IDX));
	int is_filtered = (cu_mode != DC_IDX) && (diff > filter_thr[inv_depth - 2]);
	int bl = min(size, et->pict_height[Y_COMP] - (ctu->y[Y_COMP] + y + size)), tr = min(size, et->pict_width[Y_COMP] - (ctu->x[Y_COMP] + x + size));
	int shift = curr_depth - depth + (part_size_type == SIZE_NxN);
	uint32_t ssd;
	if (traced) {
		trec t = {TR_INTRA_TU};
		TR_H(&t, size, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour, pi->top_right_neighbour, bl, tr, et->sps->strong_intra_smooth_enabled_flag,
		     is_filtered, cu_mode, scan_mode, et->enc_engine->current_pict.slice.slice_type == I_SLICE, et->pps->sign_data_hiding_flag, per, rem);
		tr_blob(&t, orig, orig_stride, size, size);
		tr_lshape(&t, dec - dec_stride - 1, dec_stride, size, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour, pi->top_right_neighbour, bl, tr);
		ssd = ((uint (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_intra_cu))(et, ctu, pi, depth, cu_mode,
															      part_size_type, curr_sum, gcnt);
		TR_H(&t, *curr_sum, (int32_t)ssd);
		tr_blob(&t, pred, pred_stride, size, size);
		tr_blob(&t, quant, size, size, size);
		tr_blob(&t, dec, dec_stride, size, size);
		tr_write(&t);
		return ssd;
	}
	ctu->top = 1;
	ctu->left = 1;
	ssd = hmr_gpu_intra_tu_chain(orig, orig_stride, dec - dec_stride - 1, dec_stride, pi->left_neighbour, pi->top_neighbour, pi->left_bottom_neighbour,
				     pi->top_right_neighbour, bl, tr, et->sps->strong_intra_smooth_enabled_flag, is_filtered, cu_mode, 1, pred, pred_stride, quant, dec,
				     dec_stride, size, size == 4, scan_mode, Y_COMP, et->enc_engine->current_pict.slice.slice_type == I_SLICE,
				     et->pps->sign_data_hiding_flag, per, rem, curr_sum);
	pi->sum = *curr_sum;
	pi->intra_cbf[Y_COMP] = ((*curr_sum ? 1 : 0) << shift);
	pi->intra_tr_idx = shift;
	pi->intra_mode[Y_COMP] = cu_mode;
	if (et->rd_mode == RD_FULL) {
		memset(&et->cbf_buffs[Y_COMP][curr_depth][pi->abs_index], ((pi->sum ? 1 : 0) << shift), pi->num_part_in_cu * sizeof(et->cbf_buffs[0][0][0]));
		memset(&et->tr_idx_buffs[curr_depth][pi->abs_index], shift, pi->num_part_in_cu * sizeof(et->tr_idx_buffs[0][0]));
		memset(&et->intra_mode_buffs[Y_COMP][curr_depth][pi->abs_index], cu_mode, pi->num_part_in_cu * sizeof(et->intra_mode_buffs[Y_COMP][0][0]));
	}
	if (!said++) fprintf(stderr, "ref_swap: intra TU chain routed to libhomer_gpu.so\n");
	return ssd;
}

/* ---- encode_intra_luma (hmr_motion_intra.c:1226): the luma decision of one 2Nx2N intra CU - mode search, parent TU, four child TUs and the tree
 * consolidation - as ONE GPU submission when the transform tree is one level deep and no CABAC bit estimate enters the comparison (rd_mode != RD_FULL;
 * the default configuration).  The host keeps the control plane: most-probable modes, neighbour flags, the bookkeeping on the partition nodes and the
 * side-info buffers (:1040-1049, :1499-1545).  Every other shape of call runs the reference's own function (whose inner calls are still interposed). ---- */
uint32_t encode_intra_luma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize part_size_type)
{
	static int said;
	int log2cu = et->max_cu_size_shift - depth, cu_min, mtp, routed = 0;
	if (log2cu < et->min_tu_size_shift + et->max_intra_tr_depth - 1)                        /* :1418-1432, part_size_type == SIZE_2Nx2N */
		cu_min = et->min_tu_size_shift;
	else {
		cu_min = log2cu - (et->max_intra_tr_depth - 1);
		if (cu_min > MAX_TU_SIZE_SHIFT) cu_min = MAX_TU_SIZE_SHIFT;
	}
	mtp = et->max_cu_size_shift - cu_min;
	if (et->performance_mode >= PERF_FAST_COMPUTATION) mtp = (depth + 2 <= mtp) ? depth + 2 : ((depth + 1 <= mtp) ? depth + 1 : mtp);
	if (want("intra_luma_cu") && part_size_type == SIZE_2Nx2N && et->rd_mode != RD_FULL && mtp == depth + 1 && (depth > 0 || et->max_cu_size == MAX_CU_SIZE))
		routed = 1;
	if (!routed)
		return ((uint32_t (*)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize))REAL(encode_intra_luma))(et, ctu, gcnt, depth, part_position, part_size_type);
	{
		cu_partition_info_t *pi = &ctu->partition_list[et->partition_depth_start[depth]] + part_position, *node[5];
		ctu_info_t *ctu_rd = et->ctu_rd;
		const int size = pi->size, x = pi->x_position, y = pi->y_position, qp = (int)pi->qp, has_parent = depth > 0;
		wnd_t *dp = et->decoded_mbs_wnd[depth + 1], *dc = et->decoded_mbs_wnd[depth + 2], *qpw = et->transform_quant_wnd[depth + 1], *qcw = et->transform_quant_wnd[depth + 2];
		int16_t *pred = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], Y_COMP, x, y, gcnt, et->ctu_width);
		int16_t *orig = WND_POSITION_2D(int16_t *, et->curr_mbs_wnd, Y_COMP, x, y, gcnt, et->ctu_width);
		int16_t *dec_par = WND_POSITION_2D(int16_t *, *dp, Y_COMP, x, y, gcnt, et->ctu_width), *dec_chl = WND_POSITION_2D(int16_t *, *dc, Y_COMP, x, y, gcnt, et->ctu_width);
		int16_t *lev_par = WND_POSITION_1D(int16_t *, *qpw, Y_COMP, gcnt, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift));
		int16_t *lev_chl = WND_POSITION_1D(int16_t *, *qcw, Y_COMP, gcnt, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift));
		int32_t nb[30], preds[3] = {-1, -1, -1}, bits[3] = {0, 0, 0}, out[24];
		int other = 0, k, mode, split;
		double cost;
		node[0] = pi;
		for (k = 0; k < 4; k++) node[k + 1] = pi->children[k];
		for (k = 0; k < 5; k++) {
			cu_partition_info_t *q = node[k];
			nb[6 * k] = q->left_neighbour; nb[6 * k + 1] = q->top_neighbour; nb[6 * k + 2] = q->left_bottom_neighbour; nb[6 * k + 3] = q->top_right_neighbour;
			nb[6 * k + 4] = min(q->size, et->pict_height[Y_COMP] - (ctu->y[Y_COMP] + q->y_position + q->size));
			nb[6 * k + 5] = min(q->size, et->pict_width[Y_COMP] - (ctu->x[Y_COMP] + q->x_position + q->size));
		}
		ctu->top = 1;                                                            /* fill_reference_samples, :256-257 */
		ctu->left = 1;
		ctu_rd->intra_mode[Y_COMP] = et->intra_mode_buffs[Y_COMP][depth];        /* homer_loop1_motion_intra, :1102 */
		get_intra_dir_luma_predictor(ctu_rd, pi, (int *)preds, NULL);
		if (et->rd_mode == RD_FAST) {
			bits[0] = bits[1] = bits[2] = 1;
			other = 12;
		}
		hmr_gpu_intra_luma_cu(orig, WND_STRIDE_2D(et->curr_mbs_wnd, Y_COMP), dec_par, WND_STRIDE_2D(*dp, Y_COMP), dec_chl, WND_STRIDE_2D(*dc, Y_COMP), nb,
				      et->sps->strong_intra_smooth_enabled_flag, preds, bits, other, et->rd.sqrt_lambda, et->adi_pred_buff, et->adi_filtered_pred_buff, pred,
				      WND_STRIDE_2D(et->prediction_wnd[0], Y_COMP), lev_par, lev_chl, size, et->enc_engine->current_pict.slice.slice_type == I_SLICE,
				      et->pps->sign_data_hiding_flag, qp / 6, qp % 6, et->rd_mode == RD_FAST, out, &cost);
		mode = out[19];
		split = out[0];
		/* what encode_intra_cu leaves on the nodes (:1040-1049) */
		for (k = has_parent ? 0 : 1; k < 5; k++) {
			cu_partition_info_t *q = node[k];
			const int shift = k ? 1 : 0;
			q->qp = (uint32_t)qp;
			q->distortion = q->cost = (uint32_t)out[9 + k];
			q->sum = (uint32_t)out[14 + k];
			q->intra_cbf[Y_COMP] = (out[14 + k] ? 1 : 0) << shift;
			q->intra_tr_idx = shift;
			q->intra_mode[Y_COMP] = mode;
		}
		if (split) {                                                             /* :1497-1522 */
			pi->cost = pi->distortion = (uint32_t)out[1];
			pi->sum = (uint32_t)out[3];
			for (k = 1; k < 5; k++) {
				cu_partition_info_t *q = node[k];
				q->intra_cbf[Y_COMP] = out[3 + k];
				memset(&et->cbf_buffs[Y_COMP][depth][q->abs_index], q->intra_cbf[Y_COMP], q->num_part_in_cu * sizeof(et->cbf_buffs[0][0][0]));
				memset(&et->tr_idx_buffs[depth][q->abs_index], q->intra_tr_idx, q->num_part_in_cu * sizeof(et->tr_idx_buffs[0][0]));
				memset(&et->intra_mode_buffs[Y_COMP][depth][q->abs_index], q->intra_mode[Y_COMP], q->num_part_in_cu * sizeof(et->intra_mode_buffs[0][0][0]));
			}
		} else {                                                                 /* :1549-1553 */
			memset(&et->cbf_buffs[Y_COMP][depth][pi->abs_index], pi->intra_cbf[Y_COMP], pi->num_part_in_cu * sizeof(et->cbf_buffs[0][0][0]));
			memset(&et->tr_idx_buffs[depth][pi->abs_index], pi->intra_tr_idx, pi->num_part_in_cu * sizeof(et->tr_idx_buffs[0][0]));
			memset(&et->intra_mode_buffs[Y_COMP][depth][pi->abs_index], pi->intra_mode[Y_COMP], pi->num_part_in_cu * sizeof(et->intra_mode_buffs[0][0][0]));
		}
		if (!said++) fprintf(stderr, "ref_swap: luma intra CU driver routed to libhomer_gpu.so\n");
		return (uint32_t)(pi->cost + out[20] * calc_mv_correction(pi->qp, et->enc_engine->avg_dist) + .5);   /* :1621-1624 */
	}
}

/* ---- encode_intra_chroma (hmr_motion_intra_chroma.c:114): the chroma half of a 2Nx2N intra CU - five-candidate search on U and V, then the TUs of the winner
 * along the luma transform tree - as ONE GPU submission when the luma tree is at most one level deep and rd_mode != RD_FULL.  Host side: the neighbour flags /
 * run lengths in chroma samples, the chroma QP and weight, and afterwards the cbf buffers, the node sums, the mode buffer and the reference's own
 * synchronize_motion_buffers_chroma (:426-455). ---- */
extern const uint8_t chroma_scale_conversion_table[];
void synchronize_motion_buffers_chroma(henc_thread_t *et, cu_partition_info_t *curr_cu_info, wnd_t *quant_src, wnd_t *quant_dst, wnd_t *decoded_src, wnd_t *decoded_dst, int gcnt);
uint32_t encode_intra_chroma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, int part_size_type)
{
	static int said;
	cu_partition_info_t *pi = &ctu->partition_list[et->partition_depth_start[depth]] + part_position, *node[5];
	int routed = want("intra_chroma_cu") && part_size_type == SIZE_2Nx2N && et->rd_mode != RD_FULL && pi->size_chroma >= 4 && (depth > 0 || et->max_cu_size == MAX_CU_SIZE);
	int split = 0, k;
	if (routed) {
		split = et->tr_idx_buffs[depth][pi->abs_index];
		for (k = 0; k < pi->num_part_in_cu; k++)
			if (et->tr_idx_buffs[depth][pi->abs_index + k] != split) routed = 0;
		if (split > 1 || (depth == 0 && split != 1)) routed = 0;
	}
	if (!routed)
		return ((uint32_t (*)(henc_thread_t *, ctu_info_t *, int, int, int, int))REAL(encode_intra_chroma))(et, ctu, gcnt, depth, part_position, part_size_type);
	{
		slice_t *currslice = &et->enc_engine->current_pict.slice;
		const int off = et->enc_engine->chroma_qp_offset, sc = pi->size_chroma, x = pi->x_position_chroma, y = pi->y_position_chroma;
		const double weight = pow(2.0, (currslice->qp - chroma_scale_conversion_table[clip(currslice->qp + off, 0, 57)]) / 3.0);
		const int qp_chroma = chroma_scale_conversion_table[clip(pi->qp + off, 0, 57)];
		const int luma_mode = et->intra_mode_buffs[Y_COMP][depth][pi->abs_index];
		wnd_t *dw = et->decoded_mbs_wnd[NUM_DECODED_WNDS - 1], *qw = et->transform_quant_wnd[NUM_QUANT_WNDS - 1];
		const int do_split = split && sc > 4;
		int32_t nb[30], out[16];
		int16_t *orig[2], *dec[2], *pred[2], *lev[2];
		uint8_t *cbf_buff[2] = {et->cbf_buffs_chroma[U_COMP], et->cbf_buffs_chroma[V_COMP]};
		int c, bits, any[2] = {0, 0};
		uint32_t cost, running = 0;
		node[0] = pi;
		for (k = 0; k < 4; k++) node[k + 1] = pi->children[k];
		for (k = 0; k < 5; k++) {
			cu_partition_info_t *q = (k && !do_split) ? pi : node[k];        /* the quadrants only matter when the TUs are split */
			nb[6 * k] = q->left_neighbour; nb[6 * k + 1] = q->top_neighbour; nb[6 * k + 2] = q->left_bottom_neighbour; nb[6 * k + 3] = q->top_right_neighbour;
			nb[6 * k + 4] = min(q->size_chroma, et->pict_height[CHR_COMP] - (ctu->y[CHR_COMP] + q->y_position_chroma + q->size_chroma));
			nb[6 * k + 5] = min(q->size_chroma, et->pict_width[CHR_COMP] - (ctu->x[CHR_COMP] + q->x_position_chroma + q->size_chroma));
		}
		for (c = 0; c < 2; c++) {
			const int comp = U_COMP + c;
			orig[c] = WND_POSITION_2D(int16_t *, et->curr_mbs_wnd, comp, x, y, gcnt, et->ctu_width);
			dec[c] = WND_POSITION_2D(int16_t *, *dw, comp, x, y, gcnt, et->ctu_width);
			pred[c] = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], comp, x, y, gcnt, et->ctu_width);
			lev[c] = WND_POSITION_1D(int16_t *, *qw, comp, gcnt, et->ctu_width, (pi->abs_index << et->num_partitions_in_cu_shift) >> 2);
		}
		ctu->top = 1;
		ctu->left = 1;
		hmr_gpu_intra_chroma_cu(orig[0], orig[1], WND_STRIDE_2D(et->curr_mbs_wnd, U_COMP), dec[0], dec[1], WND_STRIDE_2D(*dw, U_COMP), nb, luma_mode, split,
					et->rd.sqrt_lambda, weight, pred[0], pred[1], WND_STRIDE_2D(et->prediction_wnd[0], U_COMP), lev[0], lev[1], sc,
					currslice->slice_type == I_SLICE, et->pps->sign_data_hiding_flag, qp_chroma / 6, qp_chroma % 6, out);
		bits = out[2];
		/* cbf bytes and node sums as the TU loop leaves them (:318-320, :334, :346-361) */
		if (do_split) {
			for (k = 0; k < 4; k++) {
				cu_partition_info_t *q = node[k + 1];
				for (c = 0; c < 2; c++) {
					const int nz = out[6 + 4 * c + k] ? 1 : 0;
					memset(&cbf_buff[c][q->abs_index], (nz << 1) | (nz << 1), q->num_part_in_cu);
					any[c] |= nz;
					running += (uint32_t)out[6 + 4 * c + k];
				}
				q->sum += running;
			}
			for (k = pi->abs_index; k < pi->abs_index + pi->num_part_in_cu; k++) {
				cbf_buff[0][k] |= any[0];
				cbf_buff[1][k] |= any[1];
			}
		} else {
			const int sh = split ? 1 : 0;       /* an 8x8 CU whose luma is split: one 4x4 chroma TU that stands for four 2x2 ones (:289-293, :319) */
			for (c = 0; c < 2; c++) {
				const int nz = out[6 + 4 * c] ? 1 : 0;
				memset(&cbf_buff[c][pi->abs_index], (nz << sh) | nz, pi->num_part_in_cu);
				running += (uint32_t)out[6 + 4 * c];
			}
			pi->sum += running;
		}
		cost = (uint32_t)out[4] + (uint32_t)(bits * calc_mv_correction(pi->qp, et->enc_engine->avg_dist) + .5);       /* :406-411 */
		synchronize_motion_buffers_chroma(et, pi, qw, et->transform_quant_wnd[depth + 1], dw, et->decoded_mbs_wnd[depth + 1], gcnt);
		memcpy(&et->cbf_buffs[U_COMP][depth][pi->abs_index], &cbf_buff[0][pi->abs_index], pi->num_part_in_cu);
		memcpy(&et->cbf_buffs[V_COMP][depth][pi->abs_index], &cbf_buff[1][pi->abs_index], pi->num_part_in_cu);
		memset(&et->intra_mode_buffs[CHR_COMP][depth][pi->abs_index], out[0], pi->num_part_in_cu);
		pi->sum += (uint32_t)out[5];
		if (!said++) fprintf(stderr, "ref_swap: chroma intra CU driver routed to libhomer_gpu.so\n");
		return cost;
	}
}

/* ---- encode_inter_cu / encode_inter_cu_chroma (hmr_motion_inter.c:40,133): the inter TU - DCT, quantisation, keep-or-drop decision, reconstruction in
 * one GPU call; window addressing and the bookkeeping on the partition node stay on the host ---- */
extern const uint8_t chroma_scale_conversion_table[];
/* ---- encode_inter (hmr_motion_inter.c:3069): the transform tree of an inter CU.  Its TUs (encode_inter_cu / _chroma per node, luma and both chroma planes, up
 * to two levels) only read the CU's residual and prediction, so they are all computed in ONE GPU submission before the reference's own walk runs; the walk -
 * comparison cost < parent cost (:3211), cbf consolidation, window copies - stays the reference's code and finds every TU result already in place (the two
 * interposers below answer from this cache; a node the prediction missed is computed call by call as before). ---- */
static __thread struct { cu_partition_info_t *cu; int comp; uint32_t ssd; int sum; } g_pre[40];
static __thread int g_pre_n;
static int pre_lookup(cu_partition_info_t *cu, int comp, uint32_t *ssd, int *sum)
{
	int i;
	for (i = 0; i < g_pre_n; i++)
		if (g_pre[i].cu == cu && g_pre[i].comp == comp) { *ssd = g_pre[i].ssd; *sum = g_pre[i].sum; return 1; }
	return 0;
}
static void inter_tu_entry(henc_thread_t *et, cu_partition_info_t *cu, int comp, PartSize pst, int gcnt, hmr_gpu_inter_tu_host *t)
{
	slice_t *currslice = &et->enc_engine->current_pict.slice;
	cu_partition_info_t *pp = (comp == Y_COMP || cu->size_chroma != 2) ? cu : cu->parent;
	const int x = comp == Y_COMP ? cu->x_position : pp->x_position_chroma, y = comp == Y_COMP ? cu->y_position : pp->y_position_chroma;
	const int off = et->enc_engine->chroma_qp_offset;
	const int qp = comp == Y_COMP ? (int)cu->qp : chroma_scale_conversion_table[clip(cu->qp + off, 0, 57)];
	wnd_t *qw = et->transform_quant_wnd[cu->depth + 1 + (pst != SIZE_2Nx2N)], *dw = et->decoded_mbs_wnd[cu->depth + 1 + (pst != SIZE_2Nx2N)];
	memset(t, 0, sizeof *t);
	t->size = comp == Y_COMP ? cu->size : pp->size_chroma;
	t->residual = WND_POSITION_2D(int16_t *, et->residual_wnd, comp, x, y, gcnt, et->ctu_width); t->residual_stride = WND_STRIDE_2D(et->residual_wnd, comp);
	t->pred = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], comp, x, y, gcnt, et->ctu_width); t->pred_stride = WND_STRIDE_2D(et->prediction_wnd[0], comp);
	t->levels = WND_POSITION_1D(int16_t *, *qw, comp, gcnt, et->ctu_width,
				    comp == Y_COMP ? (cu->abs_index << et->num_partitions_in_cu_shift) : ((pp->abs_index << et->num_partitions_in_cu_shift) >> 2));
	t->recon = WND_POSITION_2D(int16_t *, *dw, comp, x, y, gcnt, et->ctu_width); t->recon_stride = WND_STRIDE_2D(*dw, comp);
	t->scan_mode = find_scan_mode(TRUE, TRUE, t->size, REG_DCT, 0);
	t->comp = comp;
	t->slice_is_intra = currslice->slice_type == I_SLICE;
	t->sign_hiding = et->pps->sign_data_hiding_flag;
	t->per = qp / 6; t->rem = qp % 6;
	t->weight = comp == Y_COMP ? 1.0 : pow(2.0, (currslice->qp - chroma_scale_conversion_table[clip(currslice->qp + off, 0, 57)]) / 3.0);
	t->zero_thr = clip(et->enc_engine->avg_dist / 2.5 - 5., 1., 20000.);
}
int encode_inter(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize part_size_type)
{
	static int said;
	int r;
	if (want("inter_cu") && want("inter_tu_chain") && !g_trace) {
		/* which nodes the walk will code (:3094-3148) */
		cu_partition_info_t *first[4], *nodes[20], *curr, *qnode;
		hmr_gpu_inter_tu_host tus[40];
		cu_partition_info_t *owner[40];
		int nfirst = 0, nn = 0, n = 0, i, k, mtp, cu_min, log2cu = et->max_cu_size_shift - depth, qp;
		if (log2cu < et->min_tu_size_shift + et->max_inter_tr_depth - 1 + (et->max_inter_tr_depth == 1 && part_size_type != SIZE_2Nx2N)) cu_min = et->min_tu_size_shift;
		else {
			cu_min = log2cu - (et->max_inter_tr_depth - 1 + (et->max_inter_tr_depth == 1 && part_size_type != SIZE_2Nx2N));
			if (cu_min > et->max_tu_size_shift) cu_min = et->max_tu_size_shift;
		}
		mtp = et->max_cu_size_shift - cu_min;
		if (et->performance_mode >= PERF_FAST_COMPUTATION) mtp = depth == 0 ? 1 : (depth + ((part_size_type != SIZE_2Nx2N) ? 1 : 0));
		if (depth == 0 && et->max_cu_size == MAX_CU_SIZE) {
			qnode = &ctu->partition_list[et->partition_depth_start[depth]];
			for (k = 0; k < 4; k++) first[nfirst++] = qnode->children[k];
		} else {
			qnode = curr = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
			if (et->max_inter_tr_depth == 1 && part_size_type != SIZE_2Nx2N && log2cu > mtp)
				for (k = 0; k < 4; k++) first[nfirst++] = curr->children[k];
			else
				first[nfirst++] = curr;
		}
		qp = (int)qnode->qp;                       /* the walk stamps this QP on every node it codes (:3162) */
		for (i = 0; i < nfirst; i++) {
			nodes[nn++] = first[i];
			if (first[i]->depth < mtp && nfirst == 1 && first[i]->size > 8)
				for (k = 0; k < 4; k++) nodes[nn++] = first[i]->children[k];
		}
		for (i = 0; i < nn; i++) {
			cu_partition_info_t *q = nodes[i];
			const int is_first_child = q->parent && q->parent->children[0] == q;
			q->qp = (uint32_t)qp;
			owner[n] = q; inter_tu_entry(et, q, Y_COMP, part_size_type, gcnt, &tus[n++]);
			if (q->size_chroma != 2 || is_first_child) {
				owner[n] = q; inter_tu_entry(et, q, U_COMP, part_size_type, gcnt, &tus[n++]);
				owner[n] = q; inter_tu_entry(et, q, V_COMP, part_size_type, gcnt, &tus[n++]);
			}
		}
		if (n <= 32) {
			hmr_gpu_inter_tu_chain_n(tus, n);
			for (i = 0; i < n; i++) { g_pre[i].cu = owner[i]; g_pre[i].comp = tus[i].comp; g_pre[i].ssd = tus[i].ssd; g_pre[i].sum = tus[i].ac_sum; }
			g_pre_n = n;
			if (!said++) fprintf(stderr, "ref_swap: inter CU transform tree routed to libhomer_gpu.so\n");
		}
	}
	r = ((int (*)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize))REAL(encode_inter))(et, ctu, gcnt, depth, part_position, part_size_type);
	g_pre_n = 0;
	return r;
}
int encode_inter_cu(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *cu, int depth, PartSize part_size_type, int *curr_sum, int gcnt)
{
	static int said;
	const int traced = tr_pick();
	if (!traced && !want("inter_tu_chain"))
		return ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, PartSize, int *, int))REAL(encode_inter_cu))(et, ctu, cu, depth, part_size_type, curr_sum, gcnt);
	int curr_depth = cu->depth, x = cu->x_position, y = cu->y_position, size = cu->size;
	int scan_mode = find_scan_mode(TRUE, TRUE, size, REG_DCT, 0);
	wnd_t *quant_wnd = et->transform_quant_wnd[curr_depth + 1 + (part_size_type != SIZE_2Nx2N)], *decoded_wnd = et->decoded_mbs_wnd[curr_depth + 1 + (part_size_type != SIZE_2Nx2N)];
	int16_t *pred = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], Y_COMP, x, y, gcnt, et->ctu_width);
	int16_t *res = WND_POSITION_2D(int16_t *, et->residual_wnd, Y_COMP, x, y, gcnt, et->ctu_width);
	int16_t *quant = WND_POSITION_1D(int16_t *, *quant_wnd, Y_COMP, gcnt, et->ctu_width, (cu->abs_index << et->num_partitions_in_cu_shift));
	int16_t *dec = WND_POSITION_2D(int16_t *, *decoded_wnd, Y_COMP, x, y, gcnt, et->ctu_width);
	double thr = clip(et->enc_engine->avg_dist / 2.5 - 5., 1., 20000.);
	if (traced) {
		trec t = {TR_INTER_TU};
		int r_;
		TR_H(&t, size, Y_COMP, scan_mode, et->enc_engine->current_pict.slice.slice_type == I_SLICE, et->pps->sign_data_hiding_flag, cu->qp / 6, cu->qp % 6);
		t.nd = 2; t.d[0] = 1.0; t.d[1] = thr;
		tr_blob(&t, res, WND_STRIDE_2D(et->residual_wnd, Y_COMP), size, size);
		tr_blob(&t, pred, WND_STRIDE_2D(et->prediction_wnd[0], Y_COMP), size, size);
		r_ = ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, PartSize, int *, int))REAL(encode_inter_cu))(et, ctu, cu, depth, part_size_type, curr_sum, gcnt);
		TR_H(&t, *curr_sum, r_);
		tr_blob(&t, quant, size, size, size);
		tr_blob(&t, dec, WND_STRIDE_2D(*decoded_wnd, Y_COMP), size, size);
		tr_write(&t);
		return r_;
	}
	uint32_t pre_ssd;
	int ssd;
	if (pre_lookup(cu, Y_COMP, &pre_ssd, curr_sum)) ssd = (int)pre_ssd;      /* computed ahead with the rest of the CU's tree (encode_inter above) */
	else
		ssd = (int)hmr_gpu_inter_tu_chain(res, WND_STRIDE_2D(et->residual_wnd, Y_COMP), pred, WND_STRIDE_2D(et->prediction_wnd[0], Y_COMP), quant, dec,
						  WND_STRIDE_2D(*decoded_wnd, Y_COMP), size, scan_mode, Y_COMP, et->enc_engine->current_pict.slice.slice_type == I_SLICE,
						  et->pps->sign_data_hiding_flag, cu->qp / 6, cu->qp % 6, 1.0, thr, curr_sum);
	cu->inter_cbf[Y_COMP] = ((*curr_sum ? 1 : 0) << (curr_depth - depth));
	cu->inter_tr_idx = (curr_depth - depth);
	cu->sum = *curr_sum;
	(void)ctu;
	if (!said++) fprintf(stderr, "ref_swap: inter TU chain routed to libhomer_gpu.so\n");
	return ssd;
}
int encode_inter_cu_chroma(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *cu, int component, int depth, PartSize part_size_type, int *curr_sum, int gcnt)
{
	const int traced = tr_pick();
	if (!traced && !want("inter_tu_chain"))
		return ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_inter_cu_chroma))(et, ctu, cu, component, depth,
																	 part_size_type, curr_sum, gcnt);
	slice_t *currslice = &et->enc_engine->current_pict.slice;
	int original_depth = cu->depth;
	cu_partition_info_t *pp = (cu->size_chroma != 2) ? cu : cu->parent;
	int x = pp->x_position_chroma, y = pp->y_position_chroma, size = pp->size_chroma;
	int scan_mode = find_scan_mode(TRUE, TRUE, size, REG_DCT, 0);
	int chr_qp_offset = et->enc_engine->chroma_qp_offset;
	int qp_chroma = chroma_scale_conversion_table[clip(cu->qp + chr_qp_offset, 0, 57)];
	double weight = pow(2.0, (currslice->qp - chroma_scale_conversion_table[clip(currslice->qp + chr_qp_offset, 0, 57)]) / 3.0);
	wnd_t *quant_wnd = et->transform_quant_wnd[original_depth + 1 + (part_size_type != SIZE_2Nx2N)], *decoded_wnd = et->decoded_mbs_wnd[original_depth + 1 + (part_size_type != SIZE_2Nx2N)];
	int16_t *pred = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], component, x, y, gcnt, et->ctu_width);
	int16_t *res = WND_POSITION_2D(int16_t *, et->residual_wnd, component, x, y, gcnt, et->ctu_width);
	int16_t *quant = WND_POSITION_1D(int16_t *, *quant_wnd, component, gcnt, et->ctu_width, (pp->abs_index << et->num_partitions_in_cu_shift) >> 2);
	int16_t *dec = WND_POSITION_2D(int16_t *, *decoded_wnd, component, x, y, gcnt, et->ctu_width);
	double thr = clip(et->enc_engine->avg_dist / 2.5 - 5., 1., 20000.);
	if (traced) {
		trec t = {TR_INTER_TU};
		int r_;
		TR_H(&t, size, component, scan_mode, currslice->slice_type == I_SLICE, et->pps->sign_data_hiding_flag, qp_chroma / 6, qp_chroma % 6);
		t.nd = 2; t.d[0] = weight; t.d[1] = thr;
		tr_blob(&t, res, WND_STRIDE_2D(et->residual_wnd, component), size, size);
		tr_blob(&t, pred, WND_STRIDE_2D(et->prediction_wnd[0], component), size, size);
		r_ = ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_inter_cu_chroma))(et, ctu, cu, component, depth,
																       part_size_type, curr_sum, gcnt);
		TR_H(&t, *curr_sum, r_);
		tr_blob(&t, quant, size, size, size);
		tr_blob(&t, dec, WND_STRIDE_2D(*decoded_wnd, component), size, size);
		tr_write(&t);
		return r_;
	}
	uint32_t ssd;
	if (!pre_lookup(cu, component, &ssd, curr_sum))
		ssd = hmr_gpu_inter_tu_chain(res, WND_STRIDE_2D(et->residual_wnd, component), pred, WND_STRIDE_2D(et->prediction_wnd[0], component), quant, dec,
					     WND_STRIDE_2D(*decoded_wnd, component), size, scan_mode, component, currslice->slice_type == I_SLICE,
					     et->pps->sign_data_hiding_flag, qp_chroma / 6, qp_chroma % 6, weight, thr, curr_sum);
	cu->inter_cbf[component] = ((*curr_sum ? 1 : 0) << (original_depth - depth));
	cu->sum += *curr_sum;
	(void)ctu;
	return (int)ssd;
}
