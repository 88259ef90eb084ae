/*
 * TEST INFRASTRUCTURE - CPU oracle for the HomerHEVC per-block encode hot path.
 *
 * A plain-C restatement of the reference's low_level_funcs_t kernels
 * (hmr_private.h:1063-1092) and of the in-loop kernels that live outside the
 * table (deblock, SAO offset, intra reference build, border padding), with the
 * henc_thread_t* arguments flattened to scalars.  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (homerhevc_amd/, libhomer_gpu.so) never does.
 *
 * Pinning: every function here is diffed against the compiled reference's
 * SSE4.2 symbols (oracle/_ref/libhomer_ref.so, "oracle B" flags) by
 * tests/test_oracle_vs_ref.py in the build container, and against the golden
 * vectors minted from that build (tests/golden/) everywhere else.
 *
 * All samples are int16_t, strides are in elements.
 */
#ifndef HMR_ORACLE_H
#define HMR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_SCAN_ZIGZAG 0
#define ORA_SCAN_HOR 1
#define ORA_SCAN_VER 2
#define ORA_SCAN_DIAG 3

/* tables (hmr_tables.c:62,221; hmr_encoder_lib.c:93-140) */
const uint32_t *ora_scan_table(int scan_mode, int log2_size);          /* log2_size 1..5 */
const int32_t *ora_quant_table(int log2_size, int list, int rem);      /* log2_size 2..5 */
const int32_t *ora_dequant_table(int log2_size, int list, int rem);
const int16_t *ora_dct_matrix(int log2_size);                          /* N x N row-major, log2 2..5 */
const int16_t *ora_dst_matrix(void);                                   /* 4 x 4 */

/* K6 copies (hmr_sse42_functions_pixel.c:152,236,319): arg order (src,sstride,dst,dstride,height,width) */
void ora_copy_16_16(const int16_t *src, uint32_t src_stride, int16_t *dst, uint32_t dst_stride, int height, int width);
void ora_copy_8_16(const uint8_t *src, uint32_t src_stride, int16_t *dst, uint32_t dst_stride, int height, int width);
void ora_copy_16_8(const int16_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, int height, int width);

/* K1-K5 (hmr_sse42_functions_pixel.c:462,728,817,919,1123) */
uint32_t ora_sad(const int16_t *src, uint32_t src_stride, const int16_t *pred, uint32_t pred_stride, int size);
uint32_t ora_ssd16b(const int16_t *src, uint32_t src_stride, const int16_t *pred, uint32_t pred_stride, int size);
void ora_predict(const int16_t *orig, int orig_stride, const int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int size);
void ora_reconst(const int16_t *pred, int pred_stride, const int16_t *residual, int residual_stride, int16_t *decoded, int decoded_stride, int size);
uint32_t ora_modified_variance(const int16_t *p, int size, int stride, int modif);

/* K7/K8 intra prediction (hmr_sse42_functions_prediction.c:199,926; scalar spec hmr_motion_intra.c:408,482) */
void ora_intra_planar(int16_t *pred, int pred_stride, const int16_t *adi, int adi_size, int cu_size);
void ora_intra_angular(int16_t *pred, int pred_stride, const int16_t *adi, int adi_size, int cu_size, int cu_mode, int is_luma);

/* K19 intra reference build (hmr_motion_intra.c:246,189) */
void ora_fill_reference_samples(const int16_t *decoded, int stride, int n, int left, int top, int bottom_left, int top_right,
				int bl_size, int tr_size, int16_t *adi);
void ora_adi_filter(const int16_t *adi, int16_t *out, int adi_size, int n, int strong_enabled);

/* K9-K11 (hmr_sse42_functions_inter_prediction.c:796,818,944; scalar spec hmr_motion_inter.c:262-391,878) */
void ora_interpolate_luma(const int16_t *src, int src_stride, int16_t *dst, int dst_stride, int fraction, int width, int height,
			  int is_vertical, int is_first, int is_last);
void ora_interpolate_chroma(const int16_t *src, int src_stride, int16_t *dst, int dst_stride, int fraction, int width, int height,
			    int is_vertical, int is_first, int is_last);
void ora_weighted_average(const int16_t *src0, int s0_stride, const int16_t *src1, int s1_stride, int16_t *dst, int dst_stride,
			  int height, int width);

/* K12/K13 (hmr_sse42_functions_transform.c:1670,1700; scalar spec hmr_transform.c:514,553) */
void ora_transform(const int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst);
void ora_itransform(int16_t *block, const int16_t *coeff, int block_stride, int n, int is_dst);

/* K14/K15 (hmr_sse42_functions_quant.c:34,135; hmr_quant.c:61) */
void ora_quant(const int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra,
	       int slice_is_intra, int sign_hiding, int *ac_sum, int cu_size, int per, int rem);
void ora_inv_quant(const int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem);
void ora_sign_bit_hiding(int16_t *dst, const int16_t *src, const uint32_t *scan, const int16_t *delta_u, int n_coeffs);


/* ---- frame-level in-loop kernels (K16-K18, K20); side-info = SoA over 4x4 units, raster order ---- */
#define ORA_UNIT_INTRA 1
#define ORA_UNIT_CBF_Y 2
#define ORA_UNIT_EDGE_VER 4
#define ORA_UNIT_EDGE_HOR 8
void ora_make_edge_flags(const uint8_t *pred_depth, const uint8_t *tr_idx, int width, int height, int units_stride, uint8_t *flags);
/* hmr_deblocking_filter.c:737 over the picture: all vertical edges, then all horizontal edges (in place) */
void ora_deblock_frame(int16_t *y, int ys, int16_t *u, int16_t *v, int cs, int width, int height, int units_stride, const int16_t *mvx,
		       const int16_t *mvy, const int8_t *ref_idx, const uint8_t *qp, const uint8_t *flags, int cb_qp_offset, int cr_qp_offset,
		       int beta_offset_div2, int tc_offset_div2, uint8_t *bs_ver, uint8_t *bs_hor);
/* hmr_sse42_sao.c:35; stats[ctu][comp][type][diff|count][32] int32 */
void ora_sao_stats_frame(const int16_t *oy, const int16_t *ou, const int16_t *ov, int os_y, int os_c, const int16_t *ry, const int16_t *ru,
			 const int16_t *rv, int rs_y, int rs_c, int width, int height, int32_t *stats);
/* hmr_sao.c:1210,960; params[ctu][comp][34] = {modeIdc, typeIdc, offset[32]} */
void ora_sao_apply_frame(const int16_t *sy, const int16_t *su, const int16_t *sv, int16_t *dy, int16_t *du, int16_t *dv, int stride_y, int stride_c,
			 int width, int height, const int32_t *params);
/* hmr_encoder_lib.c:1723 over every CTU */
void ora_pad_plane(int16_t *pic, int stride, int width, int height, int pad_x, int pad_y);


/* ---- motion (a15-a17) ---- */
void ora_mc_luma(const int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int width, int height, int mvx, int mvy, int is_bi);
void ora_mc_chroma(const int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int size, int mvx, int mvy, int is_bi);
uint32_t ora_motion_estimation(const int16_t *orig, int orig_stride, const int16_t *ref, int ref_stride, int gx, int gy, int init_x, int init_y,
			       int size, int range_x, int range_y, int frame_w, int frame_h, const int32_t *amvp, int n_amvp,
			       const int32_t *search, int n_search, double corr, int action, int32_t *out);


/* ---- per-TU call sequence (encode_intra_cu hmr_motion_intra.c:1030-1068), returns the SSD ---- */
uint32_t ora_tu_chain(const int16_t *orig, int orig_stride, const int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride,
		      int size, int is_dst, int scan_mode, int comp, int is_intra, int slice_is_intra, int sign_hiding, int per, int rem, int *ac_sum);

/* ---- intra mode search of one PU (homer_loop1_motion_intra, hmr_motion_intra.c:1084) ---- */
void ora_intra_search(const int16_t *orig, int orig_stride, const int16_t *decoded_corner, int decoded_stride, int n, int left, int top, int bottom_left,
		      int top_right, int bl_size, int tr_size, int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits,
		      double sqrt_lambda, int16_t *adi, int16_t *adi_filtered, int16_t *pred, int pred_stride, int32_t *out, double *best_cost);

/* ---- side-info layout: per-CTU z-order -> picture raster (abs2raster_table, hmr_encoder_lib.c:95-100) ---- */
int ora_zscan_to_raster(int a);
void ora_units_from_ctus(const int16_t *mvx, const int16_t *mvy, const int8_t *ref_idx, const uint8_t *qp, const uint8_t *pred_mode, const uint8_t *cbf_y,
			 const uint8_t *pred_depth, const uint8_t *tr_idx, int ctus_x, int ctus_y, int units_stride, int16_t *o_mvx, int16_t *o_mvy,
			 int8_t *o_ref, uint8_t *o_qp, uint8_t *o_flags, uint8_t *o_pred_depth, uint8_t *o_tr_idx);

/* ---- intra TU: neighbour array + prediction + TU chain (encode_intra_cu, hmr_motion_intra.c:1011-1068), returns the SSD ---- */
uint32_t ora_intra_tu_chain(const int16_t *orig, int orig_stride, const int16_t *decoded_corner, int decoded_stride, int left, int top, int bottom_left,
			    int top_right, int bl_size, int tr_size, int strong_enabled, int is_filtered, int mode, int is_luma, int16_t *pred, int pred_stride,
			    int16_t *levels, int16_t *recon, int recon_stride, int size, int is_dst, int scan_mode, int comp, int slice_is_intra, int sign_hiding,
			    int per, int rem, int *ac_sum);

/* ---- inter TU: DCT + quant + keep-or-drop decision + reconstruction (encode_inter_cu / _chroma, hmr_motion_inter.c:40,133) ---- */
uint32_t ora_inter_tu_chain(const int16_t *residual, int residual_stride, const int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride,
			    int size, int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem, double weight, double zero_thr, int *ac_sum);

/* ---- intra luma transform tree of one CU and the whole luma CU driver (encode_intra_luma, hmr_motion_intra.c:1226-1632) ---- */
int ora_intra_is_filtered(int mode, int size);
int ora_intra_scan_mode(int mode, int size);
void ora_intra_cu_tree(const int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
		       int strong_enabled, int mode, int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra,
		       int sign_hiding, int per, int rem, int rule, int32_t *out);
void ora_intra_luma_cu(const int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
		       int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits, double sqrt_lambda, int16_t *adi, int16_t *adi_filtered,
		       int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra, int sign_hiding, int per, int rem, int rule,
		       int32_t *out, double *best_cost);

/* ---- chroma half of an intra CU (encode_intra_chroma, hmr_motion_intra_chroma.c:114-471): five-candidate mode search on U and V, then the TUs of the winner ---- */
void ora_intra_chroma_cu(const int16_t *orig_u, const int16_t *orig_v, int orig_stride, int16_t *dec_u, int16_t *dec_v, int dec_stride, const int32_t *nb, int luma_mode,
			 int split, double sqrt_lambda, double weight, int16_t *pred_u, int16_t *pred_v, int pred_stride, int16_t *lev_u, int16_t *lev_v, int size,
			 int slice_is_intra, int sign_hiding, int per, int rem, int32_t *out);

/* ---- SAO offset derivation of one CTU from its statistics (sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion, hmr_sao.c:480-659) ---- */
void ora_sao_offsets_ctu(const int32_t *stats, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist);

#ifdef __cplusplus
}
#endif
#endif
