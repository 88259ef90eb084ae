/*
 * homer_gpu_install.c - what a HomerHEVC maintainer adds to route the encoder's low-level function table to libhomer_gpu.so.
 *
 * Compile this file inside the HomerHEVC tree (it needs the private header hmr_private.h for struct low_level_funcs_t, hmr_private.h:1063-1092, and
 * henc_thread_t) and link with -lhomer_gpu.  Eleven members of the table bind directly to the drop-in entries of include/homer_gpu.h; the other eight take a
 * henc_thread_t* (or carry arguments the kernels do not use) and go through the adapters below, which read the scalars the kernel needs.
 *
 *   void *h = HOMER_enc_init();
 *   hmr_gpu_install(h, NULL);                 // after init, before HOMER_enc_control(HOMER_SETCFG): the WPP threads copy &hvenc->funcs at SETCFG (hmr_encoder_lib.c:1262)
 *   ... encode ...
 *   hmr_gpu_uninstall(h);                     // puts back the table HOMER_enc_init had filled
 *   HOMER_enc_close(h);
 *
 * `want` (optional) selects members by name, e.g. to bring entries up one at a time.  This repo compiles the file into oracle/_ref/ref_swap (oracle/ref_swap.c,
 * test infrastructure), whose .265 output tests/test_gpu_swap.py requires to be byte-identical to the unmodified reference's.
 */
#include <string.h>
#include "hmr_private.h"
#include "hmr_common.h"
#include "homer_gpu.h"

static void gpu_planar(henc_thread_t *et, int16_t *pred, int ps, int16_t *adi, int adi_size, int n, int shift)
{ (void)et; (void)shift; hmr_gpu_intra_planar(pred, ps, adi, adi_size, n); }
static void gpu_angular(henc_thread_t *et, ctu_info_t *ctu, int16_t *pred, int ps, int16_t *adi, int adi_size, int n, int mode, int luma)
{ (void)et; (void)ctu; hmr_gpu_intra_angular(pred, ps, adi, adi_size, n, mode, luma); }
static void gpu_quant(henc_thread_t *et, int16_t *src, int16_t *dst, int scan, int depth, int comp, int cu_mode, int is_intra, int *ac_sum, int cu_size, int per, int rem)
{
	(void)cu_mode;
	hmr_gpu_quant(src, dst, et->aux_buff, scan, depth, comp, is_intra, et->enc_engine->current_pict.slice.slice_type == I_SLICE,
		      et->pps->sign_data_hiding_flag, ac_sum, cu_size, per, rem);
}
static void gpu_iquant(henc_thread_t *et, short *src, short *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem)
{ (void)et; hmr_gpu_inv_quant(src, dst, depth, comp, is_intra, cu_size, per, rem); }
static void gpu_transform(int bd, int16_t *block, int16_t *coeff, int stride, int w, int h, int ws, int hs, uint16_t mode, int16_t *aux)
{ (void)bd; (void)h; (void)ws; (void)hs; (void)aux; hmr_gpu_transform(block, coeff, stride, w, mode != REG_DCT); }
static void gpu_itransform(int bd, int16_t *block, int16_t *coeff, int stride, int w, int h, unsigned mode, int16_t *aux)
{ (void)bd; (void)h; (void)aux; hmr_gpu_itransform(block, coeff, stride, w, mode != REG_DCT); }
static void gpu_wavg(int16_t *a, int as, int16_t *b, int bs, int16_t *d, int ds, int h, int w, int bit_depth)
{ (void)bit_depth; hmr_gpu_weighted_average(a, as, b, bs, d, ds, h, w); }

static void gpu_sao_stats(henc_thread_t *et, slice_t *slice, ctu_info_t *ctu, sao_stat_data_t stats[][NUM_SAO_NEW_TYPES])
{
	wnd_t *ow = &et->enc_engine->current_pict.img2encode->img, *rw = &et->enc_engine->curr_reference_frame->img;
	const int16_t *o[3] = {ow->pwnd[0], ow->pwnd[1], ow->pwnd[2]}, *r[3] = {rw->pwnd[0], rw->pwnd[1], rw->pwnd[2]};
	int os[3] = {ow->window_size_x[0], ow->window_size_x[1], ow->window_size_x[2]}, rs[3] = {rw->window_size_x[0], rw->window_size_x[1], rw->window_size_x[2]};
	int64_t flat[3][NUM_SAO_NEW_TYPES][2][32];
	int c, t;
	(void)slice;
	hmr_gpu_get_sao_stats(o, os, r, rs, et->pict_width[0], et->pict_height[0], ctu->x[0], ctu->y[0], &flat[0][0][0][0]);
	for (c = 0; c < 3; c++)
		for (t = 0; t < NUM_SAO_NEW_TYPES; t++) {
			memcpy(stats[c][t].diff, flat[c][t][0], sizeof flat[c][t][0]);
			memcpy(stats[c][t].count, flat[c][t][1], sizeof flat[c][t][1]);
		}
}

static low_level_funcs_t g_saved_funcs;
static void *g_saved_for;

/* returns the number of members routed to the GPU */
int hmr_gpu_install(void *handle, int (*want)(const char *member))
{
	low_level_funcs_t *f = &((hvenc_enc_t *)handle)->funcs;
	int n = 0;
	g_saved_funcs = *f;
	g_saved_for = handle;
#define SWAP(member, fn) if (!want || want(#member)) { f->member = fn; n++; }
	SWAP(sse_copy_16_16, hmr_gpu_copy_16_16) SWAP(sse_copy_16_8, hmr_gpu_copy_16_8) SWAP(sse_copy_8_16, hmr_gpu_copy_8_16)
	SWAP(sad, hmr_gpu_sad) SWAP(ssd16b, hmr_gpu_ssd16b) SWAP(predict, hmr_gpu_predict) SWAP(reconst, hmr_gpu_reconst)
	SWAP(modified_variance, hmr_gpu_modified_variance)
	SWAP(create_intra_planar_prediction, gpu_planar) SWAP(create_intra_angular_prediction, gpu_angular)
	SWAP(interpolate_luma_m_compensation, hmr_gpu_interpolate_luma) SWAP(interpolate_luma_m_estimation, hmr_gpu_interpolate_luma)
	SWAP(interpolate_chroma_m_compensation, hmr_gpu_interpolate_chroma) SWAP(weighted_average_motion, gpu_wavg)
	SWAP(quant, gpu_quant) SWAP(inv_quant, gpu_iquant) SWAP(transform, gpu_transform) SWAP(itransform, gpu_itransform)
	SWAP(get_sao_stats, gpu_sao_stats)
#undef SWAP
	return n;
}

void hmr_gpu_uninstall(void *handle)
{
	if (handle && handle == g_saved_for) {
		((hvenc_enc_t *)handle)->funcs = g_saved_funcs;
		g_saved_for = NULL;
	}
}
