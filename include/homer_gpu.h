/*
 * homer_gpu.h - C ABI of the MI355X (gfx950) backend for HomerHEVC's per-block encode hot path.
 *
 * The reference's plug-in surface is `struct low_level_funcs_t` (hmr_private.h:1063-1092), a table of
 * 19 function pointers filled once in HOMER_enc_init (hmr_encoder_lib.c:155-214).  This library
 * replaces what is behind that table, plus the in-loop kernels the reference keeps outside it
 * (deblock, SAO offset, intra reference build, border padding), with hand-written HIP
 * kernels.  Three layers, all `extern "C"`, plain pointers and sizes:
 *
 *   1. context / device memory          hmr_gpu_create ... hmr_gpu_timer_stop
 *   2. drop-in entries                   hmr_gpu_sad(...) etc.: HOST pointers, synchronous, the exact
 *                                        low_level_funcs_t signatures (henc_thread_t* arguments flattened
 *                                        to the scalars they carry); one launch per call - correct, not fast
 *   3. batched entries                   hmr_gpu_*_batch(ctx, jobs, n, ...): DEVICE-resident planes addressed by
 *                                        job descriptors; one launch per batch - the performance path
 *      frame-level in-loop passes        hmr_gpu_deblock_frame / sao_* / pad_frame
 *
 * All samples are int16_t and strides are in elements, as in the reference.
 * Every entry returns 0 on success or a negative hmr_gpu_status; nothing falls back to the CPU.
 */
#ifndef HOMER_GPU_H
#define HOMER_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmr_gpu_ctx hmr_gpu_ctx;

enum hmr_gpu_status {
	HMR_GPU_OK = 0,
	HMR_GPU_ERR_NO_DEVICE = -1,   /* no HIP device / wrong architecture */
	HMR_GPU_ERR_HIP = -2,         /* a HIP call failed, see hmr_gpu_last_error */
	HMR_GPU_ERR_ARG = -3,         /* unsupported size or malformed argument */
	HMR_GPU_ERR_NOMEM = -4
};

/* ------------------------------------------------------------------------------------------------
 * 1. context, device memory, timing
 * ------------------------------------------------------------------------------------------------ */

/* Create a context on `device` (HIP ordinal).  `stream` may be an existing hipStream_t (e.g. the
 * torch current stream, passed as void*) or NULL to let the context own one. */
int hmr_gpu_create(hmr_gpu_ctx **out, int device, void *stream);
void hmr_gpu_destroy(hmr_gpu_ctx *ctx);
const char *hmr_gpu_last_error(void);
int hmr_gpu_sync(hmr_gpu_ctx *ctx);
void *hmr_gpu_stream(hmr_gpu_ctx *ctx);   /* the hipStream_t every launch of this context goes to */

int hmr_gpu_malloc(hmr_gpu_ctx *ctx, void **dev_ptr, size_t bytes);
int hmr_gpu_free(hmr_gpu_ctx *ctx, void *dev_ptr);
int hmr_gpu_upload(hmr_gpu_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);     /* sync */
int hmr_gpu_download(hmr_gpu_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);   /* sync */
int hmr_gpu_memset(hmr_gpu_ctx *ctx, void *dev_dst, int value, size_t bytes);

/* An empty launch on the context's stream (timing calibration: what an event pair around a launch costs besides the kernel). */
int hmr_gpu_nop(hmr_gpu_ctx *ctx);
/* VALU issue probe: blocks * 4 wavefronts, each issuing iters * 8 independent packed dot products and nothing else (out: any device word, never written in
 * practice).  bench.py times it to state the integer issue rate the kernels are priced against as measured on the box, next to the nominal one. */
int hmr_gpu_valu_probe(hmr_gpu_ctx *ctx, int kind, int blocks, int iters, uint32_t *out);   /* kind 0: v_dot2_i32_i16, 1: v_sad_u16 */
/* Cap on the workgroups of one batched launch (process-wide, default 4096).  Batched kernels grid-stride over their jobs, so results do not
 * depend on it; lower it to leave compute units to concurrent streams. */
int hmr_gpu_set_max_grid(int blocks);

/* HIP-event timer on the context's stream (bench.py brackets the timed region with it). */
int hmr_gpu_timer_start(hmr_gpu_ctx *ctx);
int hmr_gpu_timer_stop(hmr_gpu_ctx *ctx, float *elapsed_ms);   /* records, synchronises, returns ms */
/* explicit event pairs on the same stream, for per-kernel durations (elapsed() is valid once ev1 has completed) */
int hmr_gpu_event_create(hmr_gpu_ctx *ctx, void **ev);
int hmr_gpu_event_record(hmr_gpu_ctx *ctx, void *ev);
int hmr_gpu_event_elapsed(void *ev0, void *ev1, float *elapsed_ms);
int hmr_gpu_event_destroy(void *ev);

/* Constant tables the kernels use, built at context creation (defaults of hmr_tables.c:62,221 and
 * hmr_encoder_lib.c:93-140); exposed so the host can check them against its own. */
int hmr_gpu_get_scan_table(int scan_mode, int log2_size, uint32_t *out);          /* (1<<log2)^2 entries */
int hmr_gpu_get_quant_table(int log2_size, int list, int rem, int32_t *quant, int32_t *dequant);

/* ------------------------------------------------------------------------------------------------
 * 2. drop-in entries: host pointers, synchronous.  Signatures = low_level_funcs_t members
 *    (hmr_private.h:1066-1091); they run on the process-wide default context (device 0, created on
 *    first use, or the one set with hmr_gpu_set_default).
 * ------------------------------------------------------------------------------------------------ */
int hmr_gpu_set_default(hmr_gpu_ctx *ctx);

/* hmr_private.h:1066-1068 (sse_copy_16_16 / sse_copy_16_8 / sse_copy_8_16), SSE impl hmr_sse42_functions_pixel.c:152,319,236 */
void hmr_gpu_copy_16_16(void *src, uint32_t src_stride, void *dst, uint32_t dst_stride, int height, int width);
void hmr_gpu_copy_16_8(void *src, uint32_t src_stride, void *dst, uint32_t dst_stride, int height, int width);
void hmr_gpu_copy_8_16(void *src, uint32_t src_stride, void *dst, uint32_t dst_stride, int height, int width);
/* :1069 sad, :1071 ssd16b, :1072 predict, :1073 reconst, :1074 modified_variance */
uint32_t hmr_gpu_sad(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size);
uint32_t hmr_gpu_ssd16b(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size);
void hmr_gpu_predict(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int size);
void hmr_gpu_reconst(int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int16_t *decoded, int decoded_stride, int size);
uint32_t hmr_gpu_modified_variance(int16_t *p, int size, int stride, int modif);
/* :1076 create_intra_planar_prediction, :1077 create_intra_angular_prediction (henc_thread_t*, ctu_info_t* dropped:
 * they only carry scratch buffers, the angle tables and ctu->top/left which are always 1, hmr_motion_intra.c:257) */
void hmr_gpu_intra_planar(int16_t *prediction, int pred_stride, int16_t *adi_pred_buff, int adi_size, int cu_size);
void hmr_gpu_intra_angular(int16_t *prediction, int pred_stride, int16_t *adi_pred_buff, int adi_size, int cu_size, int cu_mode, int is_luma);
/* not in the table: fill_reference_samples hmr_motion_intra.c:246 and adi_filter :189, partition node flattened to flags */
void hmr_gpu_fill_reference_samples(int16_t *decoded_corner, int stride, int n, int left, int top, int bottom_left, int top_right,
				    int bl_size, int tr_size, int16_t *adi_out);
void hmr_gpu_adi_filter(int16_t *adi, int16_t *out, int adi_size, int n, int strong_enabled);
/* :1079-1081 interpolate_luma_m_compensation/_m_estimation, interpolate_chroma_m_compensation, :1083 weighted_average_motion */
void hmr_gpu_interpolate_luma(int16_t *reference_buff, int reference_buff_stride, int16_t *pred_buff, int pred_buff_stride, int fraction,
			      int width, int height, int is_vertical, int is_first, int is_last);
void hmr_gpu_interpolate_chroma(int16_t *reference_buff, int reference_buff_stride, int16_t *pred_buff, int pred_buff_stride, int fraction,
				int width, int height, int is_vertical, int is_first, int is_last);
void hmr_gpu_weighted_average(int16_t *src0, int src0_stride, int16_t *src1, int src1_stride, int16_t *dst, int dst_stride, int height, int width);
/* :1088 transform, :1089 itransform (bit depth fixed at 8, uiMode reduced to "DST-VII for 4x4 intra luma") */
void hmr_gpu_transform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst);
void hmr_gpu_itransform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst);
/* :1085 quant, :1086 inv_quant.  henc_thread_t* replaced by what it is read for: slice type (hmr_sse42_functions_quant.c:47),
 * pps->sign_data_hiding_flag (:121) and the aux_buff scratch that receives deltaU (:48; NULL = do not return it). */
void hmr_gpu_quant(int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra, int slice_is_intra,
		   int sign_hiding, int *ac_sum, int cu_size, int per, int rem);
void hmr_gpu_inv_quant(int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem);
/* :1091 get_sao_stats.  The henc_thread_t, slice_t and ctu_info_t pointers are replaced by what they are read for: the picture planes (HOST pointers to
 * sample (0,0), strides in elements), the picture size and the CTU's luma position; stats[comp][type][diff|count][32] as int64 like
 * sao_stat_data_t (hmr_private.h:455).  Only the CTU and its one-sample ring are transferred. */
void hmr_gpu_get_sao_stats(const int16_t *const orig[3], const int orig_stride[3], const int16_t *const recon[3], const int recon_stride[3], int pict_width,
			   int pict_height, int ctu_x, int ctu_y, int64_t *stats);

/* In-loop filters at the reference's own call granularity, HOST pointers (planes address sample (0,0), strides in elements; unit arrays are raster over
 * the picture's 4x4 units like hmr_gpu_units; unit_flags = HMR_GPU_UNIT_INTRA | HMR_GPU_UNIT_CBF_Y).  Only the CTU's neighbourhood is transferred.
 * hmr_deblock_filter_cu (hmr_deblocking_filter.c:737), sao_offset_ctu (hmr_sao.c:1210), reference_picture_border_padding_ctu (hmr_encoder_lib.c:1723). */
void hmr_gpu_deblock_filter_ctu(int16_t *const planes[3], const int strides[3], int width, int height, int units_stride, const int16_t *mvx, const int16_t *mvy,
				const int8_t *ref_idx, const uint8_t *qp, const uint8_t *unit_flags, const uint8_t *pred_depth, const uint8_t *tr_idx, int ctu_x,
				int ctu_y, int ctu_size, int dir, int cb_qp_offset, int cr_qp_offset, int beta_offset_div2, int tc_offset_div2);
void hmr_gpu_sao_offset_ctu(const int16_t *const src[3], const int src_stride[3], int16_t *const dst[3], const int dst_stride[3], int width, int height, int ctu_x,
			    int ctu_y, const int32_t *params /* [3][34] = {modeIdc, typeIdc, offset[32]} per component */);
void hmr_gpu_pad_ctu(int16_t *const planes[3], const int strides[3], int width, int height, int pad_x, int pad_y, int ctu_x, int ctu_y, int ctu_size);

/* ------------------------------------------------------------------------------------------------
 * 3. batched entries: device-resident operands, one launch per call, asynchronous on the context's
 *    stream.  A job addresses up to three operands by ELEMENT offset from the base pointer passed to
 *    the call; all jobs of one call share `size` (the host groups work by block size, the way it
 *    groups CUs by depth).  `jobs` is a DEVICE pointer (upload with hmr_gpu_upload).
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_job {
	uint32_t a_off, a_stride;   /* first operand  (src / orig / pred / reference / adi / coeff-in)  */
	uint32_t b_off, b_stride;   /* second operand (pred / residual / ...)                            */
	uint32_t c_off, c_stride;   /* output                                                            */
	uint16_t w, h;              /* extent where the kernel is not square (copies, interpolation)    */
	uint32_t p0, p1;            /* kernel-specific parameters, see each entry                        */
} hmr_gpu_job;

/* out[i] = SAD / SSD of job i.  a = src, b = pred (b_stride may be 0 for ssd) */
int hmr_gpu_sad_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, const int16_t *b_base, uint32_t *out);
int hmr_gpu_ssd16b_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, const int16_t *b_base, uint32_t *out);
/* c = a - b */
int hmr_gpu_predict_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, const int16_t *b_base, int16_t *c_base);
/* c = clip(a (pred) + b (residual, stride may be 0)) */
int hmr_gpu_reconst_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, const int16_t *b_base, int16_t *c_base);
/* c[h x w] = a[h x w]; kind 0: i16->i16, 1: u8->i16, 2: i16->u8 (offsets/strides in elements of each side's type).
 * kind | N << 8 promises that every job is an N x N int16 square (N = 4, 8, 16, 32, 64): vectorised fast path. */
int hmr_gpu_copy_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int kind, const void *a_base, void *c_base);
/* out[i] = modified variance of the size x size block at a; p0 = modif */
int hmr_gpu_modified_variance_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, uint32_t *out);
/* a = adi array (4*size+1 entries at a_off), c = prediction; p0 = cu_mode (0 planar, 1 DC, 2..34), p1 = is_luma */
int hmr_gpu_intra_pred_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base);
/* a = reconstructed plane, a_off addresses the corner sample (-1,-1), c = adi out (4*size+1), b_off = filtered adi out;
 * p0 bits: 0 left, 1 top, 2 bottom_left, 3 top_right, 4 write filtered copy, 5 strong filter enabled; p1 = bl_size | tr_size << 16 */
int hmr_gpu_intra_refs_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base);
/* a = reference/intermediate, c = out, w/h = extent; p0 = fraction, p1 bits: 0 vertical, 1 first, 2 last.
 * flags bit 0: luma (8 taps) / chroma (4 taps); bits 8..15: lanes-per-job hint 4 / 8 / 16 / 32 / 64 (0 = 64).  A job is
 * ceil(w/4)*ceil(h/4) work items when vertical (4x4 outputs from a register window of rows) or ceil(w/8)*h when
 * horizontal (8 outputs per item); that many lanes share it, so batches of small blocks should pass a small hint. */
int hmr_gpu_interpolate_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int flags, const int16_t *a_base, int16_t *c_base);
int hmr_gpu_weighted_average_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, const int16_t *a_base, const int16_t *b_base, int16_t *c_base);
/* a = residual block (strided), c = coefficients (linear size*size at c_off); p0 = is_dst */
int hmr_gpu_transform_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base);
/* a = coefficients (linear), c = residual block (strided); p0 = is_dst */
int hmr_gpu_itransform_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base);
/* a = coefficients in, c = levels out (both linear), b_off = deltaU out when delta_u_base != NULL;
 * p0 bits 0-1 scan_mode, 2-3 comp, 4 is_intra, 5 slice_is_intra, 6 sign_hiding; p1 = per | rem << 8; ac_sum[i] out */
int hmr_gpu_quant_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base,
			int16_t *delta_u_base, int32_t *ac_sum);
/* a = levels, c = coefficients; p0 bits 2-3 comp, 4 is_intra; p1 = per | rem << 8 */
int hmr_gpu_inv_quant_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a_base, int16_t *c_base);


/* ------------------------------------------------------------------------------------------------
 * 4. frame-level in-loop passes (the reference keeps these outside the table and runs them CTU by CTU in a
 *    lagged pipeline, hmr_encoder_lib.c:2386 hmr_deblock_sao_pad_sync_ctu; whole-picture passes are
 *    equivalent).  All pointers are DEVICE pointers; launches are asynchronous on the context's stream.
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_frame {
	int width, height;          /* luma size, multiples of 8 (hmr_encoder_lib.c:1001) */
	int16_t *y, *u, *v;         /* sample (0,0) of each plane; margins, if any, lie before/after */
	int stride_y, stride_c;     /* elements */
} hmr_gpu_frame;

/* side-info the encoder decided, one entry per 4x4 luma unit in raster order (ctu_info_t arrays of
 * hmr_private.h:792-843 are the same data in z-order per CTU) */
#define HMR_GPU_UNIT_INTRA 1     /* pred_mode == INTRA_MODE */
#define HMR_GPU_UNIT_CBF_Y 2     /* CBF(ctu, idx, Y_COMP, tr_idx) */
#define HMR_GPU_UNIT_EDGE_VER 4  /* left edge of the unit is a transform/CU edge (set by hmr_gpu_edge_flags_frame) */
#define HMR_GPU_UNIT_EDGE_HOR 8  /* top edge */
typedef struct hmr_gpu_units {
	int units_stride;           /* units per row of the arrays below */
	const int16_t *mvx, *mvy;   /* mv_ref[REF_PIC_LIST_0], quarter-sample units */
	const int8_t *ref_idx;      /* mv_ref_idx[REF_PIC_LIST_0], < 0 = none */
	const uint8_t *qp;
	uint8_t *flags;
} hmr_gpu_units;

/* The same side-info as the encoder keeps it: per CTU, 256 entries in z-order (ctu_info_t: mv_ref[0], mv_ref_idx[0], qp, pred_mode, cbf[Y_COMP],
 * pred_depth, tr_idx), CTU-major.  hmr_gpu_units_from_ctus re-orders it on the device into the raster arrays above (abs2raster_table,
 * hmr_encoder_lib.c:95-100; flags = INTRA | CBF_Y at the unit's tr_idx) plus raster pred_depth / tr_idx for hmr_gpu_edge_flags_frame. */
typedef struct hmr_gpu_ctu_units {
	const int16_t *mvx, *mvy;
	const int8_t *ref_idx;
	const uint8_t *qp, *pred_mode, *cbf_y, *pred_depth, *tr_idx;
} hmr_gpu_ctu_units;
int hmr_gpu_units_from_ctus(hmr_gpu_ctx *ctx, const hmr_gpu_ctu_units *src, int ctus_x, int ctus_y, const hmr_gpu_units *dst, uint8_t *pred_depth_raster,
			    uint8_t *tr_idx_raster);

/* derive the EDGE bits of flags[] from the coding tree: pred_depth + tr_idx per unit (hmr_deblocking_filter.c:737-825) */
int hmr_gpu_edge_flags_frame(hmr_gpu_ctx *ctx, const uint8_t *pred_depth, const uint8_t *tr_idx, int width, int height, int units_stride, uint8_t *flags);
/* hmr_deblock_filter_cu over the picture (hmr_deblocking_filter.c:737): boundary strength, luma strong/weak, chroma; in place.
 * bs_ver / bs_hor (optional, units_stride x height/4) receive 0x80|bs for every evaluated edge segment. P/I slices. */
int hmr_gpu_deblock_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *frame, const hmr_gpu_units *info, int cb_qp_offset, int cr_qp_offset,
			  int beta_offset_div2, int tc_offset_div2, uint8_t *bs_ver, uint8_t *bs_hor);
/* hmr_deblock_filter_cu's own granularity (hmr_deblocking_filter.c:737): the edges of one direction (0 = EDGE_VER, 1 = EDGE_HOR, hmr_private.h:122)
 * inside the CTU at luma position (ctu_x, ctu_y).  pred_depth / tr_idx (optional, DEVICE, raster like the unit arrays): when given, the EDGE bits of
 * the CTU's units are derived first. */
int hmr_gpu_deblock_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *frame, const hmr_gpu_units *info, const uint8_t *pred_depth, const uint8_t *tr_idx,
			int cb_qp_offset, int cr_qp_offset, int beta_offset_div2, int tc_offset_div2, int ctu_x, int ctu_y, int ctu_size, int dir);
/* low_level_funcs_t.get_sao_stats (hmr_private.h:1091, hmr_sse42_sao.c:35) for every CTU:
 * stats[ctu][comp][type EO0,EO90,EO135,EO45,BO][0 = diff, 1 = count][32] as int32 */
int hmr_gpu_sao_stats_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *orig, const hmr_gpu_frame *recon, int32_t *stats);
/* the same for one CTU (the table's call granularity); stats[comp][type][diff|count][32] */
int hmr_gpu_sao_stats_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *orig, const hmr_gpu_frame *recon, int ctu_index, int32_t *stats);
/* sao_offset_ctu (hmr_sao.c:1210) for every CTU: src = pre-SAO picture, dst = output (must hold a copy of src);
 * params[ctu][comp][34] = {modeIdc, typeIdc, offset[32]} */
int hmr_gpu_sao_apply_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *src, const hmr_gpu_frame *dst, const int32_t *params);
/* the same for one CTU (sao_offset_ctu's own granularity); params = that CTU's [3][34] */
int hmr_gpu_sao_apply_ctu(hmr_gpu_ctx *ctx, const hmr_gpu_frame *src, const hmr_gpu_frame *dst, int ctu_index, const int32_t *params);
/* reference_picture_border_padding_ctu (hmr_encoder_lib.c:1723) for every CTU: replicate edges into pad_x/pad_y (luma; chroma half) */
int hmr_gpu_pad_frame(hmr_gpu_ctx *ctx, const hmr_gpu_frame *frame, int pad_x, int pad_y);


/* ------------------------------------------------------------------------------------------------
 * 5. motion: search driver and compensation (the reference's L3 helpers that sit directly on the kernels)
 * ------------------------------------------------------------------------------------------------ */
/* One PU of hmr_motion_estimation (hmr_motion_inter.c:1404).  Vectors are in quarter samples. */
typedef struct hmr_gpu_me_job {
	double corr;                 /* calc_mv_correction(qp, avg_dist) = qp * clip(avg_dist / 2000., .15, 1.4), hmr_common.h:53 (host-side double) */
	uint32_t orig_off, orig_stride;   /* source block */
	uint32_t ref_off, ref_stride;     /* co-located block (mv = 0) in the padded reference picture */
	int16_t gx, gy;              /* curr_part_global_x / _y */
	int16_t init_x, init_y;      /* start vector, integer samples */
	int16_t n_amvp, n_search;    /* candidate counts: AMVP list (vector cost, 1..2), search list et->mv_search_candidates (0..5) */
	int16_t amvp[2][2];
	int16_t search[5][2];
	uint32_t action;             /* MOTION_PEL_MASK 1 | MOTION_HALF_PEL_MASK 2 | MOTION_QUARTER_PEL_MASK 4, hmr_common.h:77-79 */
	uint32_t reserved;
} hmr_gpu_me_job;
typedef struct hmr_gpu_me_result {
	int32_t mvx, mvy;            /* *mv */
	int32_t subx, suby;          /* *subpix_mv */
	uint32_t sad;                /* return value */
} hmr_gpu_me_result;
/* jobs / out are DEVICE arrays; all PUs of a call share `size` (8, 16, 32, 64); range = MOTION_SEARCH_RANGE_X/Y (128/64, hmr_private.h:76-77).
 * The 16 sub-pel planes of hmr_half/quarter_pixel_estimation_luma_hm (hmr_motion_inter.c:395,442) are produced and consumed on chip. */
int hmr_gpu_motion_estimation_batch(hmr_gpu_ctx *ctx, const hmr_gpu_me_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *ref_base,
				    int range_x, int range_y, int frame_w, int frame_h, hmr_gpu_me_result *out);
/* hmr_motion_compensation_luma / _chroma (hmr_motion_inter.c:1779,1860): a = co-located block, c = prediction, w/h extent,
 * p0 = mv.x, p1 = mv.y (as int32; quarter samples for luma, eighth samples for chroma).
 * flags bit 0: luma / chroma; bits 8..15: lanes-per-job hint 4 / 8 / 16 / 32 / 64 (0 = 64): that many lanes share a block of w*h samples (a work item is
 * four adjacent outputs; a two-stage block whose (h + taps - 1) x w first stage does not fit lanes/64 of the wave's LDS tile falls back to a slower form) */
int hmr_gpu_mc_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int flags, int is_bi_predict, const int16_t *a_base, int16_t *c_base);
/* host-pointer (drop-in) forms */
uint32_t hmr_gpu_motion_estimation(int16_t *orig, int orig_stride, int16_t *ref, int ref_stride, int gx, int gy, int init_x, int init_y, int size,
				   int range_x, int range_y, int frame_w, int frame_h, const int32_t *amvp, int n_amvp, const int32_t *search, int n_search,
				   double corr, int action, int32_t *out_mv4);
void hmr_gpu_mc_luma(int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int width, int height, int mvx, int mvy, int is_bi);
void hmr_gpu_mc_chroma(int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int size, int mvx, int mvy, int is_bi);


/* ------------------------------------------------------------------------------------------------
 * 6. command lists: one frame's launches described once, replayed from C (optionally as a hipGraph)
 * ------------------------------------------------------------------------------------------------ */
enum hmr_gpu_op {
	HMR_GPU_OP_SAD = 1, HMR_GPU_OP_SSD16B, HMR_GPU_OP_PREDICT, HMR_GPU_OP_RECONST, HMR_GPU_OP_COPY, HMR_GPU_OP_VARIANCE, HMR_GPU_OP_INTRA_PRED,
	HMR_GPU_OP_INTRA_REFS, HMR_GPU_OP_INTERPOLATE, HMR_GPU_OP_WAVG, HMR_GPU_OP_TRANSFORM, HMR_GPU_OP_ITRANSFORM, HMR_GPU_OP_QUANT,
	HMR_GPU_OP_INV_QUANT, HMR_GPU_OP_MC, HMR_GPU_OP_ME, HMR_GPU_OP_EDGE_FLAGS, HMR_GPU_OP_DEBLOCK, HMR_GPU_OP_SAO_STATS, HMR_GPU_OP_SAO_APPLY, HMR_GPU_OP_PAD,
	HMR_GPU_OP_TU_CHAIN,  /* jobs = hmr_gpu_tu_job*, a = orig base, b = pred base, c = level base, out = ssd; the reconstruction base and ac_sum follow in p64[0..1] */
	HMR_GPU_OP_INTRA_SEARCH,  /* jobs = hmr_gpu_intra_job*, a = orig base, b = decoded base, c = output base, out = hmr_gpu_intra_result* */
	HMR_GPU_OP_INTER_TU_CHAIN = 25, /* jobs = hmr_gpu_inter_tu_job*, a = residual base, b = pred base, c = level base, out = ssd; p64 = {recon base, ac_sum} */
	HMR_GPU_OP_SAO_OFFSETS = 30,    /* a = stats, njobs = CTUs, b = lambdas, c = offsets, out = aux, p64[0] = dist */
	HMR_GPU_OP_CHROMA_SEARCH = 29,  /* jobs = hmr_gpu_chroma_job*, a = orig base, b = decoded base, p64[0] = luma search results or NULL, out = hmr_gpu_intra_result* */
	HMR_GPU_OP_TU_MULTI = 28,       /* jobs = hmr_gpu_tu_segment* (host), njobs = segments, a = orig base, b = decoded base, c = level base, p64[0] = recon / prediction base */
	HMR_GPU_OP_PIXEL_MULTI = 27,    /* jobs = hmr_gpu_segment* (host), njobs = segments, size = HMR_GPU_OP_SAD / SSD16B / PREDICT / RECONST / COPY */
	HMR_GPU_OP_TREE_DECIDE = 26,    /* jobs = hmr_gpu_tree_job*, a = ssd, b = ac_sum, c = recon base, out = hmr_gpu_tree_result*; p64[0] = level base */
	HMR_GPU_OP_INTRA_TU_CHAIN = 24 /* jobs = hmr_gpu_itu_job*, a = orig base, b = decoded base, c = level base, out = ssd; p64 = {recon base, ac_sum, hmr_gpu_intra_result* or NULL};
	                           * the prediction plane shares the recon base; p[0] = rounds (0 / 1 = one set of jobs) */
};
/* One call of the batched / frame-level API: `size` is that entry's size/kind/flags argument, a/b/c/out its pointer arguments in
 * declaration order (frame-level ops take host pointers to hmr_gpu_frame / hmr_gpu_units that must outlive the list), p[] its
 * scalar arguments (deblock: cb, cr, beta, tc offsets; pad: pad_x, pad_y; edge flags: width, height, units_stride; ME: range_x,
 * range_y, frame_w, frame_h; MC: size = flags, p[0] = is_bi_predict).  QUANT: b = deltaU base (may be NULL), out = ac_sum. */
typedef struct hmr_gpu_cmd {
	int op, njobs, size;
	int p[4];
	const void *jobs, *a, *b;
	void *c, *out;
	void *p64[3];               /* extra pointer arguments (TU_CHAIN: recon base, ac_sum; INTRA_TU_CHAIN: + the search results when jobs take their mode from them) */
	int branch;                 /* 0 = the context's stream.  Commands with the same branch run in list order; different branches are
	                             * declared independent of each other and may overlap when the list is replayed as a graph (fork at the
	                             * start of the list, join at its end).  hmr_gpu_cmdlist_run ignores it (one stream, list order). */
} hmr_gpu_cmd;
#define HMR_GPU_MAX_BRANCHES 64
/* A context and its command lists belong to one host thread at a time (capture temporarily redirects the context's stream to the branch
 * streams); use one context per thread / per encoder engine. */
/* Several batches of ONE pixel kernel that differ only in the block size, in one launch: a frame's SAD (or SSD, residual, reconstruction, square int16
 * copy) batches are short launches that cannot fill the GPU one at a time.  op = HMR_GPU_OP_SAD / SSD16B / PREDICT / RECONST / COPY; every segment is what
 * the corresponding hmr_gpu_*_batch call takes (jobs = device array, size = block size, out = that batch's result array for SAD / SSD, ignored otherwise);
 * a, b, c as in the single-batch entries (COPY: a -> c).  Command lists: op HMR_GPU_OP_PIXEL_MULTI, size = the pixel op, jobs = host array of segments
 * (must outlive the list), njobs = number of segments. */
#define HMR_GPU_MAX_SEGMENTS 8
typedef struct hmr_gpu_segment {
	const hmr_gpu_job *jobs;
	void *out;
	int njobs, size;
} hmr_gpu_segment;
int hmr_gpu_pixel_multi(hmr_gpu_ctx *ctx, int op, const hmr_gpu_segment *segs, int nseg, const int16_t *a, const int16_t *b, int16_t *c);
/* The same for the fused TU chains of section 7: any mix of TU size (4, 8, 16, 32) and kind - 0 given prediction (hmr_gpu_tu_job), 1 intra (hmr_gpu_itu_job),
 * 2 inter (hmr_gpu_inter_tu_job) - as segments of one launch; all segments address the same five bases.  rounds / modes as in
 * hmr_gpu_intra_tu_chain_rounds_batch (0 / NULL when unused).  A launch that contains a 32x32 segment reserves that body's LDS image (52 KB per workgroup)
 * for every segment.  Command lists: op HMR_GPU_OP_TU_MULTI, jobs = host array of segments, njobs = number of segments, a = orig, b = decoded, c = level base,
 * p64[0] = recon / prediction base. */
struct hmr_gpu_intra_result;
typedef struct hmr_gpu_tu_segment {
	const void *jobs;
	uint32_t *ssd;
	int32_t *ac_sum;
	const struct hmr_gpu_intra_result *modes;
	int njobs, size, kind, rounds;
} hmr_gpu_tu_segment;
int hmr_gpu_tu_chain_multi(hmr_gpu_ctx *ctx, const hmr_gpu_tu_segment *segs, int nseg, const int16_t *orig_base, const int16_t *decoded_base, int16_t *pred_base,
			   int16_t *level_base, int16_t *recon_base);
typedef struct hmr_gpu_cmdlist hmr_gpu_cmdlist;
int hmr_gpu_cmdlist_create(hmr_gpu_ctx *ctx, const hmr_gpu_cmd *cmds, int n, hmr_gpu_cmdlist **out);
/* eager replay; event_pairs (optional, 2*n events from hmr_gpu_event_create) brackets every command for per-kernel timing */
int hmr_gpu_cmdlist_run(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *list, void **event_pairs);
/* capture the list into a hipGraph once, then launch the graph (a frame is a fixed launch sequence) */
int hmr_gpu_cmdlist_capture(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *list);
int hmr_gpu_cmdlist_replay(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *list);
void hmr_gpu_cmdlist_destroy(hmr_gpu_cmdlist *list);


/* ------------------------------------------------------------------------------------------------
 * 7. fused transform-unit chain: predict -> transform -> quant(+SBH) -> [inv_quant -> itransform] -> reconst -> ssd16b
 *    for a batch of TUs in one launch (the sequence of encode_intra_cu hmr_motion_intra.c:1014-1069 and
 *    encode_inter_cu hmr_motion_inter.c:40-230); bit-identical to the seven calls issued one after the other.
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_tu_job {
	uint32_t orig_off, orig_stride;   /* source block */
	uint32_t pred_off, pred_stride;   /* prediction block */
	uint32_t rec_off, rec_stride;     /* reconstruction out */
	uint32_t lev_off;                 /* quantised levels out, linear size*size */
	uint32_t p0;                      /* bits 0-1 scan_mode, 2-3 comp, 4 is_intra, 5 slice_is_intra, 6 sign_hiding, 7 is_dst (4x4 intra luma) */
	uint32_t p1;                      /* per | rem << 8 */
} hmr_gpu_tu_job;
int hmr_gpu_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_tu_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *pred_base,
			   int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum);
/* host-pointer (drop-in) form; returns the SSD between source and reconstruction */
uint32_t hmr_gpu_tu_chain(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size,
			  int is_dst, int scan_mode, int comp, int is_intra, int slice_is_intra, int sign_hiding, int per, int rem, int *ac_sum);

/* The same chain for an intra TU, prediction included: encode_intra_cu's data path (hmr_motion_intra.c:1011-1068) - neighbour array from the plane
 * under reconstruction (smoothed when the host's is_filtered rule :1011-1012 says so), planar / DC / angular prediction, then the chain above.
 * The reconstruction may go back into the plane the neighbours are read from (the reference works in place); jobs of one call must not depend
 * on each other's reconstruction. */
typedef struct hmr_gpu_itu_job {
	uint32_t orig_off, orig_stride;   /* source block */
	uint32_t pred_off, pred_stride;   /* prediction out (prediction_wnd) */
	uint32_t rec_off, rec_stride;     /* reconstruction out */
	uint32_t lev_off;                 /* quantised levels out, linear size*size */
	uint32_t p0, p1;                  /* as hmr_gpu_tu_job (is_intra is implied) */
	uint32_t dec_off, dec_stride;     /* corner sample (-1,-1) of the block in the plane under reconstruction */
	uint32_t flags;                   /* bits 0 left, 1 top, 2 bottom_left, 3 top_right, 5 strong_intra_smooth_enabled, 6 is_filtered, 7 is_luma */
	uint32_t sizes;                   /* bl_size | tr_size << 16 */
	uint32_t mode;                    /* 0 planar, 1 DC, 2..34 angular */
} hmr_gpu_itu_job;
int hmr_gpu_intra_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *decoded_base,
				 int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum);
/* host-pointer (drop-in) form; returns the SSD between source and reconstruction.  `recon` may point into the plane `decoded_corner` belongs to. */
uint32_t hmr_gpu_intra_tu_chain(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int left, int top, int bottom_left, int top_right,
				int bl_size, int tr_size, int strong_enabled, int is_filtered, int mode, int is_luma, int16_t *pred, int pred_stride, int16_t *levels,
				int16_t *recon, int recon_stride, int size, int is_dst, int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem,
				int *ac_sum);
/* As above with the prediction mode handed over ON THE DEVICE: a job whose flags carry bit 8 takes its mode from modes[job.mode].best_mode (the result
 * array of an intra search launched before it) and derives the smoothing rule (hmr_motion_intra.c:1011-1012) and the scan (find_scan_mode,
 * hmr_tables.c:398-402) from mode and TU size itself; is_filtered / scan_mode / mode of such a job are ignored.  A job with is_luma = 0 (chroma TU: its mode
 * comes from a chroma search, section 10) is never smoothed and takes the chroma scan rule (mode dependent for 4x4 only, hmr_tables.c:403-411); only the
 * low 8 bits of best_mode are the prediction mode. */
#define HMR_GPU_ITU_MODE_FROM_SEARCH 0x100u
struct hmr_gpu_intra_result;   /* section 8 */
int hmr_gpu_intra_tu_chain_modes_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *decoded_base,
				       int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum,
				       const struct hmr_gpu_intra_result *modes);
/* `rounds` sets of njobs jobs in one launch (jobs[r * njobs + j], ssd / ac_sum indexed the same way): job j of set r + 1 may read what job j of set r
 * reconstructed - the four children of a CU in z-order (hmr_motion_intra.c:1441-1477) run back to back on the same lanes instead of as four launches.
 * modes may be NULL when no job carries HMR_GPU_ITU_MODE_FROM_SEARCH. */
int hmr_gpu_intra_tu_chain_rounds_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int rounds, int size, const int16_t *orig_base,
					const int16_t *decoded_base, int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum,
					const struct hmr_gpu_intra_result *modes);
/* The inter TU: encode_inter_cu / encode_inter_cu_chroma (hmr_motion_inter.c:40-230).  The residual of the motion-compensated CU is given; per TU: DCT,
 * quantisation as non-intra, and for a coded TU the keep-or-drop decision ssd_zero <= ssd + zero_thr * sum on SSDs in the residual domain, scaled by
 * `weight` and truncated to uint32 like the reference (luma: weight 1.0).  zero_thr = clip(avg_dist / 2.5 - 5, 1, 20000) (:59-60,108; host double).
 * Outputs: levels (zeros when dropped), reconstruction, ac_sum (0 when dropped), ssd = the residual-domain SSD of the coded candidate. */
typedef struct hmr_gpu_inter_tu_job {
	uint32_t orig_off, orig_stride;   /* RESIDUAL block (residual_wnd) */
	uint32_t pred_off, pred_stride;   /* prediction block */
	uint32_t rec_off, rec_stride;     /* reconstruction out */
	uint32_t lev_off;                 /* quantised levels out, linear size*size */
	uint32_t p0, p1;                  /* as hmr_gpu_tu_job (is_intra = 0, is_dst = 0) */
	uint32_t reserved;                /* bit 0 (HMR_GPU_INTER_TU_FROM_SOURCE): orig_off addresses the SOURCE block and the residual = source - prediction (16-bit wrap, the
	                                   * `predict` call the reference issues per CU ahead of encode_inter, hmr_motion_inter.c:3054-3056) is formed in the kernel */
	double weight, zero_thr;
} hmr_gpu_inter_tu_job;
#define HMR_GPU_INTER_TU_FROM_SOURCE 1u
int hmr_gpu_inter_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_inter_tu_job *jobs, int njobs, int size, const int16_t *residual_base, const int16_t *pred_base,
				 int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum);
uint32_t hmr_gpu_inter_tu_chain(int16_t *residual, int residual_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size,
				int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem, double weight, double zero_thr, int *ac_sum);

/* All inter TUs of one CU's transform tree (encode_inter, hmr_motion_inter.c:3069-3290: encode_inter_cu + encode_inter_cu_chroma per node, luma and both chroma
 * planes, parent and child level) in one submission, host pointers: inter TUs read only the CU's residual and prediction, so the tree can be computed ahead of
 * the walk that compares (cost < parent cost, :3211) and consolidates it.  In: the pointer / parameter fields; out: levels, recon, ssd, ac_sum of every entry. */
typedef struct hmr_gpu_inter_tu_host {
	int16_t *residual; int residual_stride;
	int16_t *pred; int pred_stride;
	int16_t *levels;
	int16_t *recon; int recon_stride;
	int size, scan_mode, comp, slice_is_intra, sign_hiding, per, rem;
	double weight, zero_thr;
	uint32_t ssd;
	int ac_sum;
} hmr_gpu_inter_tu_host;
void hmr_gpu_inter_tu_chain_n(hmr_gpu_inter_tu_host *tus, int n);

/* ------------------------------------------------------------------------------------------------
 * 8. intra mode search of one PU: homer_loop1_motion_intra (hmr_motion_intra.c:1084-1179) - reference build, smoothing,
 *    up to 13 {prediction, SAD} rounds over the search_points schedule (:1076) and the strict-< cost comparison
 *    SAD + bits * sqrt_lambda (double) in one launch.  The most-probable-mode list (get_intra_dir_luma_predictor,
 *    hmr_arithmetic_encoding.c:545) and the bit counts are host inputs: RD_FAST 1 / 12 (:1153-1157), RD_FULL the CABAC
 *    estimate of each list entry / 6 (:1141-1150), RD_DIST_ONLY 0 / 0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_intra_job {
	double sqrt_lambda;               /* et->rd.sqrt_lambda */
	uint32_t orig_off, orig_stride;   /* source block */
	uint32_t dec_off, dec_stride;     /* corner sample (-1,-1) of the block in the plane under reconstruction */
	uint32_t adi_off, adif_off;       /* out: raw and smoothed neighbour arrays, 4*size+1 each (et->adi_pred_buff / adi_filtered_pred_buff) */
	uint32_t pred_off, pred_stride;   /* out: prediction of the LAST candidate evaluated (what the reference leaves in prediction_wnd) */
	uint32_t flags;                   /* bits 0 left, 1 top, 2 bottom_left, 3 top_right, 5 strong_intra_smooth_enabled */
	uint32_t sizes;                   /* bl_size | tr_size << 16 (hmr_motion_intra.c:289,335) */
	int32_t preds[3];                 /* most probable modes, -1 = none */
	uint32_t pred_bits[3];            /* bit count when the candidate equals preds[i] */
	uint32_t other_bits;              /* bit count of any other candidate */
	uint32_t reserved;
} hmr_gpu_intra_job;
typedef struct hmr_gpu_intra_result {
	int32_t best_mode;                /* best_pred_modes[0] */
	int32_t bits;                     /* return value: bit count of the best mode */
	double cost;                      /* best_pred_cost[0] */
} hmr_gpu_intra_result;
/* jobs / out are DEVICE arrays; all PUs of a call share `size` (4, 8, 16, 32, 64) */
int hmr_gpu_intra_search_batch(hmr_gpu_ctx *ctx, const hmr_gpu_intra_job *jobs, int njobs, int size, const int16_t *orig_base,
			       const int16_t *decoded_base, int16_t *out_base, hmr_gpu_intra_result *out);
/* host-pointer (drop-in) form; out = {best mode, bit count} */
void hmr_gpu_intra_search(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int n, int left, int top, int bottom_left,
			  int top_right, int bl_size, int tr_size, int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits,
			  double sqrt_lambda, int16_t *adi, int16_t *adi_filtered, int16_t *pred, int pred_stride, int32_t *out, double *best_cost);


/* ------------------------------------------------------------------------------------------------
 * 9. transform-tree consolidation and the luma intra CU driver: the walk of encode_intra_luma (hmr_motion_intra.c:1441-1566) for a
 *    one-level tree (max_intra_tr_depth = 2, rd_mode != RD_FULL).  The TUs themselves are launches of section 7 - the parent TUs of
 *    all CUs in one launch on the parent level's plane, then child 0, 1, 2, 3 of all CUs in four launches on the child level's plane
 *    (a child reads its siblings' reconstruction) - and this kernel is the comparison and the buffer consolidation (:1479-1557):
 *       rule 0 (RD_DIST_ONLY)  dist < parent dist            rule 1 (RD_FAST)  1.25 * (dist + 45 * sum) < parent dist + 45 * parent sum
 *    children win -> levels and reconstruction copied up (synchronize_motion_buffers_luma, :866), cbf of quadrant k = nz_k << 1 | any nz;
 *    parent wins  -> its bottom row and right column copied down (synchronize_reference_buffs, :844), cbf = nz.
 *    No host round trip between search, TUs and decision: a CU's whole luma decision is one ordered chain of launches.
 * ------------------------------------------------------------------------------------------------ */
#define HMR_GPU_TREE_NO_PARENT 0xffffffffu      /* a 64x64 CU has no parent TU (cost preset to INT_MAX, :1402): always consolidated */
typedef struct hmr_gpu_tree_job {
	uint32_t parent;                          /* index of the parent TU in ssd[] / ac_sum[] */
	uint32_t child[4];                        /* indices of the four child TUs, z-order */
	uint32_t par_rec_off, par_rec_stride;     /* the CU in the parent level's plane (decoded_mbs_wnd[depth + 1]) */
	uint32_t chl_rec_off, chl_rec_stride;     /* the CU in the child level's plane (decoded_mbs_wnd[depth + 2]) */
	uint32_t par_lev_off, chl_lev_off;        /* size * size levels each, linear; the children's blocks are consecutive (abs_index order) */
	uint32_t size;                            /* CU size: 8, 16, 32, 64 */
	uint32_t rule;                            /* 0 / 1 above */
} hmr_gpu_tree_job;
typedef struct hmr_gpu_tree_result {
	uint32_t split;                           /* 1 = the four children were kept */
	uint32_t cost, sum;                       /* what the partition node carries afterwards (cost = distortion) */
	uint8_t cbf[4];                           /* cbf byte of the four quadrants (tr_idx = split) */
} hmr_gpu_tree_result;
int hmr_gpu_tree_decide_batch(hmr_gpu_ctx *ctx, const hmr_gpu_tree_job *jobs, int njobs, const uint32_t *ssd, const int32_t *ac_sum, int16_t *recon_base,
			      int16_t *level_base, hmr_gpu_tree_result *out);
/* host-pointer (drop-in) form of the whole luma CU: mode search, parent TU, four child TUs and the consolidation in one submission.
 * nb: 5 x {left, top, bottom_left, top_right, bl_size, tr_size} for the CU and its four quadrants; dec_par / dec_chl: the CU's first sample in the
 * two planes (both hold the neighbours); lev_par / lev_chl: size * size each.
 * out[0] split, [1] = [2] cost, [3] sum, [4..7] cbf, [8] tr_idx, [9..13] ssd of parent + children, [14..18] their sums (slot 0 carries the consolidated
 * figures after a split), [19] mode, [20] its bit count. */
void hmr_gpu_intra_luma_cu(int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
			   int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits, double sqrt_lambda, int16_t *adi, int16_t *adi_filtered,
			   int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra, int sign_hiding, int per, int rem, int rule,
			   int32_t *out, double *best_cost);


/* ------------------------------------------------------------------------------------------------
 * 10. chroma half of an intra CU: encode_intra_chroma (hmr_motion_intra_chroma.c:114-471, rd_mode != RD_FULL).
 *     Search: five candidates (planar, vertical, horizontal, DC, DM = the luma mode; an entry equal to the luma mode becomes 34) predicted for U and V
 *     from unfiltered neighbours, cost = dU + (dU + dV) + (uint32)(bits * sqrt_lambda + .5) with bits 1 for DM and 12 otherwise, first minimum wins.
 *     The TUs of the winner are hmr_gpu_itu_job launches of section 7 with is_luma = 0 whose mode comes from this search's result array.
 *     Result: best_mode = prediction mode | coded chroma mode << 8 (36 = DM), bits, cost.
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_chroma_job {
	double sqrt_lambda;                       /* et->rd.sqrt_lambda */
	uint32_t orig_u_off, orig_v_off, orig_stride;
	uint32_t dec_u_off, dec_v_off, dec_stride;   /* corner sample (-1,-1) of the CU in the two planes under reconstruction */
	uint32_t flags;                           /* bits 0 left, 1 top, 2 bottom_left, 3 top_right; bit 8: luma_mode is an index into the luma search results */
	uint32_t sizes;                           /* bl_size | tr_size << 16, in chroma samples */
	uint32_t luma_mode;
	uint32_t reserved;
} hmr_gpu_chroma_job;
/* size = chroma size of the searched block: 4, 8, 16 (a 64x64 CU is searched on its first 32x32 quadrant only, hmr_motion_intra_chroma.c:165-169) */
int hmr_gpu_chroma_search_batch(hmr_gpu_ctx *ctx, const hmr_gpu_chroma_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *decoded_base,
				const hmr_gpu_intra_result *luma_modes, hmr_gpu_intra_result *out);
/* host-pointer (drop-in) form of the whole function: search + the U / V TUs of the winner along the luma tree (split = 0: one TU of `size` per component,
 * split = 1: four of size / 2 in z-order; a 4x4 chroma CU is always one TU) in one submission.  size: chroma CU size 4 ... 32; nb: 5 x {left, top,
 * bottom_left, top_right, bl_size, tr_size} in chroma samples for the CU and its quadrants; weight: the chroma distortion weight (:149).
 * out: [0] coded mode, [1] prediction mode, [2] bits, [3] search cost, [4] distortion = sum of (int)(weight * SSD), [5] sum, [6..9] / [10..13] ac sums of
 * the U / V TUs. */
void hmr_gpu_intra_chroma_cu(int16_t *orig_u, int16_t *orig_v, int orig_stride, int16_t *dec_u, int16_t *dec_v, int dec_stride, const int32_t *nb, int luma_mode,
			     int split, double sqrt_lambda, double weight, int16_t *pred_u, int16_t *pred_v, int pred_stride, int16_t *lev_u, int16_t *lev_v, int size,
			     int slice_is_intra, int sign_hiding, int per, int rem, int32_t *out);


/* ------------------------------------------------------------------------------------------------
 * 11. SAO offset derivation (8-bit): sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion (hmr_sao.c:480-659) for the five types of the
 *     three components of every CTU, straight from the statistics array of hmr_gpu_sao_stats_frame - the part of the SAO decision (sao_derive_mode_new_rdo,
 *     :663) that is a pure function of the statistics.  The rate terms and the off / new / merge comparison read the CABAC state and stay on the host.
 *     stats [ctu][3][5][2][32] int32, lambdas [ctu][3] double (hmr_wpp_sao_ctu, :1415), offsets [ctu][3][5][32], aux [ctu][3][5] (band position; 0 for the
 *     edge types), dist [ctu][3][5] int64; all device pointers.
 * ------------------------------------------------------------------------------------------------ */
int hmr_gpu_sao_offsets_frame(hmr_gpu_ctx *ctx, const int32_t *stats, int n_ctu, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist);
/* host-pointer (drop-in) form for one CTU; lambdas[3] */
void hmr_gpu_sao_offsets_ctu(const int32_t *stats, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist);


/* ------------------------------------------------------------------------------------------------
 * 12. frame encoder: the reference's public API (homer_hevc_enc_api.h:169-174) with the CTU loop of its WPP thread
 *     (wfpp_encoder_thread, hmr_encoder_lib.c:2849-2975: stage the CTU, motion_inter / motion_intra, store, in-loop filters,
 *     entropy coding) running on the device.  One persistent launch walks the picture as the WPP wavefront: a wavefront
 *     per CTU row, row r+1 two CTUs behind row r (:2885-2898); planes, side-info and levels stay resident in HBM.
 *     hmr_gpu_enc_cfg has the layout and field names of HVENC_Cfg (homer_hevc_enc_api.h:138-167), so a caller passes
 *     the struct it already fills for HOMER_enc_control(HOMER_SETCFG).
 *     Built rows: 8-bit 4:2:0, 64x64 CTUs, I / P slices with one reference picture, performance_mode 0-3, max_intra_tr_depth / max_inter_tr_depth up to 4,
 *     rd_mode 0 / 2 and - at fixed QP - rd_mode 1 (RD_FULL: the CABAC bit counter prices the intra decisions); fixed QP and bitrate_mode 1 / 2 (CBR / VBR:
 *     bitrate, vbv_size, vbv_init, frame_rate) - BASELINE configs 0 .. 4.  Rate control and RD_FULL run with one WPP thread (the reference's deterministic
 *     serial mode: the picture one CTU at a time in raster order), with a thread per CTU row, and - rate control - with num_enc_engines > 1 (round 6); RD_FULL with
 *     several engines and RD_FULL under rate control are refused.  Anything else, pictures of more than 128 CTU rows and pictures of more than 192 wavefront steps (CTU columns + 2 x (CTU rows - 1))
 *     make hmr_gpu_enc_create return HMR_GPU_ERR_ARG (hmr_gpu_last_error says why).
 *     wfpp_num_threads = 1: the stream of the reference's single worker thread.  wfpp_num_threads = CTU rows: the stream of its
 *     multi-thread mode with the threads advancing as a synchronous wavefront (pinned by oracle/ref_ctudump.c, HOMER_TURNSTILE);
 *     fewer threads than rows are accepted when 2 x threads >= CTU columns (2160p: 32 threads, the reference's maximum, for 34 rows), others are refused.
 *     Picture grids the compiled reference cannot run are refused too (no stream exists to compare with): two CTU columns x several rows (it crashes),
 *     num_enc_engines > 1 on fewer than nine CTU columns with more than four CTU rows (its engines deadlock), and the two small-grid corners of its
 *     lagged filter pipeline that the frame passes here do not reproduce (enc/enc_host.h make_seq: one CTU column; three columns with four or more rows;
 *     SAO on at most five columns with at least as many rows).
 *     Deblocking, SAO (statistics, decision, syntax, offsets), the CABAC coding of the CTU rows' sub-streams and border padding are tasks of the same launch
 *     (enc/enc_post.h); the host writes parameter sets, slice header, entry points and the NAL escaping.
 * ------------------------------------------------------------------------------------------------ */
typedef struct hmr_gpu_enc_cfg {
	int32_t size, profile, width, height;
	float frame_rate;
	int32_t cu_size, max_pred_partition_depth, max_intra_tr_depth, max_inter_tr_depth, intra_period, gop_size, num_b, num_ref_frames;
	int32_t motion_estimation_precision, qp, chroma_qp_offset, num_enc_engines, wfpp_enable, wfpp_num_threads, sign_hiding, sample_adaptive_offset;
	int32_t bitrate_mode, bitrate, vbv_size, vbv_init, reinit_gop_on_scene_change, rd_mode, performance_mode;
} hmr_gpu_enc_cfg;
typedef struct hmr_gpu_enc hmr_gpu_enc;

/* HOMER_enc_init + HOMER_enc_control(HOMER_SETCFG) */
int hmr_gpu_enc_create(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, hmr_gpu_enc **out);
/* The same for BATCHES of sequences in the reference's deterministic single-thread order (wfpp_num_threads = 1, num_enc_engines = 1: BASELINE.md's parity mode, the
 * stream of the plain reference without any pinned interleaving): the object's pictures run CTU by CTU in raster order as tasks of the batch launch (section 12b) - one
 * decision in flight per picture, the post-decision tasks beside it - so it takes many pictures per launch to fill the device.  An object made by hmr_gpu_enc_create with
 * one thread, fixed QP and RD_FAST runs the guess / verify / re-encode schedule instead: faster for ONE sequence, not a batch schedule (the batch calls refuse it). */
int hmr_gpu_enc_create_serial_pool(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, hmr_gpu_enc **out);
/* HOMER_enc_close */
void hmr_gpu_enc_destroy(hmr_gpu_enc *enc);
/* bytes of one per-CTU record of hmr_gpu_enc_frame_ctus (layout of oracle/ref_ctudump.c) */
int hmr_gpu_enc_record_bytes(void);
/* The CTU decisions of one frame (the part of HOMER_enc_encode between slice set-up and the in-loop filters): y / u / v are host
 * 8-bit planes (width x height, width/2 x height/2); image_type as encoder_in_out_t.image_type (0 auto, 3 forced intra);
 * ref_y / ref_u / ref_v, when not NULL, replace the encoder's reference picture (8-bit, unpadded) - used by the parity tests to
 * compare frame by frame against the reference's own pictures; avg_dist < 0 keeps the encoder's own running value.
 * records (host, may be NULL): nctu x hmr_gpu_enc_record_bytes() bytes, one record per CTU. Returns the slice type (1 P, 2 I) or a negative status. */
int hmr_gpu_enc_frame_ctus(hmr_gpu_enc *enc, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
			   const uint8_t *ref_v, double avg_dist, uint8_t *records);
/* milliseconds the CTU passes of the last frame took on the device (HIP events on the context's stream) */
float hmr_gpu_enc_last_ctu_ms(hmr_gpu_enc *enc);

/* HOMER_enc_encode (homer_hevc_enc_api.h:173, hmr_encoder_lib.c:1655 + encoder_engine_thread :3043-3260): one picture in, one access unit out.
 * CTU decisions, rate control, deblocking, SAO statistics / parameter decision (hmr_sao.c:663-1410) / offsets, the CABAC coding of the CTU rows' sub-streams
 * (hmr_arithmetic_encoding.c, hmr_binary_encoding.c) and border padding run on the device; the host assembles the access unit (hmr_headers.c, hmr_bitstream.c)
 * from the sub-streams.  y / u / v: host 8-bit planes; image_type as encoder_in_out_t.image_type (0 auto, 3 forced intra).  stream receives the
 * Annex-B bytes of the access unit (VPS / SPS / PPS in front of an IDR), *stream_bytes their count; recon (optional) the final picture,
 * 8-bit planar.  Returns the slice type (1 P, 2 I) or a negative status. */
int hmr_gpu_enc_encode(hmr_gpu_enc *enc, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, uint8_t *stream, long cap, long *stream_bytes,
		       uint8_t *recon);
/* the same with the source picture already resident in HBM: load_source converts and keeps a picture in slot `slot`, encode_source encodes it */
int hmr_gpu_enc_load_source(hmr_gpu_enc *enc, int slot, const uint8_t *y, const uint8_t *u, const uint8_t *v);
int hmr_gpu_enc_encode_source(hmr_gpu_enc *enc, int slot, int image_type, uint8_t *stream, long cap, long *stream_bytes, uint8_t *recon);
/* one frame of each of n sequences with ONE launch for all their CTU stages: encs[i] encodes its resident picture slots[i] (image_types may be NULL: automatic)
 * into streams[i] (capacity caps[i], size stream_bytes[i]).  The encoders use the row-per-thread schedule (wfpp_num_threads > 1) and the same device; the access
 * units are those hmr_gpu_enc_encode_source gives one by one.  The launch is a pool of persistent row workers (four per CU; as many as the pictures can keep busy)
 * that claim CTUs of any of the n pictures whose wavefront step is open (k_encode_pool); n is at most 512, a few hundred pictures' worth of CTU rows saturate the pool
 * (120 at 1080p; a launch costs about 110 ms of ramps on top of 1.7 ms per 1080p picture, so more pictures per launch amortise it: 256 -> 512 is +10 %).  A worker never waits for a CTU that is not already running, so the launch does not depend on all its
 * workgroups being resident; a watchdog (HENC_WATCHDOG_S seconds, fractions allowed, 120 by default) makes a launch in which a worker has found nothing to do - no CTU
 * to decide, no post-decision task - for that long SINCE IT LAST DID return HMR_GPU_ERR_HIP instead of hanging. */
int hmr_gpu_enc_encode_batch(hmr_gpu_enc **encs, int n, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes);
/* the same, pipelined: call k launches the frames slots[] and delivers the access units of call k - 1's frames (stream_bytes[i] = 0 on the first call), whose download
 * and entropy coding run while the device is busy with call k's CTU stage - a frame's successor needs its reconstruction and its distortion statistic, not its bytes
 * (the reference's interface is asynchronous in the same way: HOMER_enc_encode queues a picture, HOMER_enc_get_coded_frame, hmr_encoder_lib.c:2997, takes coded frames
 * from the output queue later).  slots == NULL: deliver the outstanding
 * access units only (flush).  encs / n stay the same from call to call until the flush; an encoder with an access unit outstanding is refused by the other encode calls. */
int hmr_gpu_enc_encode_batch_pipelined(hmr_gpu_enc **encs, int n, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes);
/* the last frame: passes of the CTU schedule, CTU encodes (>= the number of CTUs), device milliseconds of the CTU passes and of the whole frame */
int hmr_gpu_enc_last_stats(hmr_gpu_enc *enc, int *passes, int *ctu_encodes, float *ctu_ms, float *frame_ms);
/* The one place where byte identity with the reference is not guaranteed, counted: a merge candidate whose vector points outside the padded reference picture
 * (more than 80 samples beyond the frame) is evaluated by the reference on whatever its thread's prediction window holds (check_rd_cost_merge_2nx2n leaves out
 * the motion compensation and nothing else, hmr_motion_inter.c:3651).  The window travels with the thread here too (the stream is the
 * reference's when its content came from block writes), but the reference's SSE predictors also leave samples outside the blocks they predict (e.g. 255s right of
 * a 4 x 4 angular chroma block), which are not reproduced.  *last_picture: such evaluations in the last picture encoded through this object (-1: none encoded
 * yet; counted in both thread orders), *all_pictures: since the object was created.  0 = the quirk did not occur (every clip of bench.py; 12 of about 3000 random fuzz cases
 * had it, one of them differs: profiles/r04_encoder_fuzz.md).  Either pointer may be NULL. */
int hmr_gpu_enc_stale_predictions(hmr_gpu_enc *enc, long *last_picture, long *all_pictures);
/* profiling build (-DHENC_PROFILE): per-row phase timers, [ctu rows][12] */
int hmr_gpu_enc_profile(hmr_gpu_enc *enc, unsigned long long *out, int reset);
/* profiling build, row-per-thread schedule: per CTU four 100 MHz timestamps {wait start, encode start, first use of the intra share (0: none), end}, [ctus][4] */
int hmr_gpu_enc_timeline(hmr_gpu_enc *enc, unsigned long long *out);

/* ------------------------------------------------------------------------------------------------
 * 12b. Engines: hvenc_enc_t.num_encoder_engines / encoder_engine_thread (hmr_encoder_lib.c:3043-3330; hmr_private.h:1232)
 *     hmr_gpu_enc_cfg.num_enc_engines = E > 1 (row-per-thread schedule only) gives the stream of the reference's frame pipeline in the
 *     interleaving oracle/ref_ctudump.c's engine turnstile pins on it: frame n is encoded on the complete reconstruction of frame
 *     n - 1, starts from the avg_dist frame n - E left behind and works on the persistent state of engine n mod E.
 *     hmr_gpu_enc_create keeps all E engines in one object.  hmr_gpu_enc_create_engine makes ONE engine (index k of E): it is given
 *     only the frames k, k + E, ... and, before each of them except frame 0, the hand-over of the engine before it -
 *     hmr_gpu_enc_export_reference on that engine after its frame, hmr_gpu_enc_import_reference here: the reconstructed picture
 *     (three padded int16 planes in DEVICE memory, hmr_gpu_enc_reference_elems(enc, comp) elements each, e.g. buffers an RCCL
 *     send / recv moves between GPUs) and hmr_gpu_enc_state_bytes() bytes of frame-to-frame scalars (host memory).
 * ------------------------------------------------------------------------------------------------ */
int hmr_gpu_enc_create_engine(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, int engine_index, hmr_gpu_enc **out);
int hmr_gpu_enc_state_bytes(void);
long hmr_gpu_enc_reference_elems(hmr_gpu_enc *enc, int comp);
int hmr_gpu_enc_export_reference(hmr_gpu_enc *enc, int16_t *dev_y, int16_t *dev_u, int16_t *dev_v, void *state);
int hmr_gpu_enc_import_reference(hmr_gpu_enc *enc, const int16_t *dev_y, const int16_t *dev_u, const int16_t *dev_v, const void *state);
/* The hand-over of n engines at once with the picture as it travels between GPUs: 8-bit samples without margins (hmr_gpu_enc_reference_bytes: width x height
 * luma, then the two chroma planes; 3.1 MB at 1080p against 6.6 MB of padded int16 planes), picture i at dev_rows + i * pitch in DEVICE memory; the importer
 * widens the samples and rebuilds the margins (reference_picture_border_padding_ctu, hmr_encoder_lib.c:1723).  states: n x hmr_gpu_enc_state_bytes() bytes (host). */
long hmr_gpu_enc_reference_bytes(hmr_gpu_enc *enc);
/* 12c. Overlapping engines on one GPU.  The reference's engine k + 1 follows engine k's frame at the distance of the search window and the filter lag, CTU row by CTU row
 * (hmr_encoder_lib.c:2393-2403 the lag arithmetic, :2440-2445 the per-CTU SEM_POST of synchro_signal[1], :3154-3211 the frame hand-out).  hmr_gpu_enc_encode_chain encodes
 * the frames slots[0 .. n - 1] (n <= num_enc_engines; more with twins, below) of ONE sequence on its engine objects encs[0 .. n - 1] (in coding order; prev = the object that encoded the frame
 * before, NULL at the sequence start) in one launch of the CTU kernel: frame j predicts from the final picture of frame j - 1 where it lies (no copy) and from its phase
 * planes, which the same launch produces CTU by CTU; a wavefront step of frame j starts when the part of that picture its vectors can reach is ready.  Same streams as frame
 * by frame (the engine turnstile's interleaving, oracle/ref_ctudump.c).  Returns HMR_GPU_ERR_ARG when two frames of the chain both detect a scene change (sequentially the
 * first switches the detection off for the second): repeat those frames one by one.  A call that is refused while its frames are being set up (wrong engine for a frame, a P
 * frame without the object that holds the picture before it, an I frame too early in the chain) leaves every object as it found it.  Chains of more than one frame are refused
 * under rate control and with rd_mode RD_FULL: both read entropy-coder state of the frame before (hmr_rate_control.c:266-282, the coder objects' context states), which a
 * chain's frames, started from a predicted state, do not have - encode those sequences frame by frame.
 * More frames than engines (n up to 32): the reference's engine k takes frame t + num_enc_engines when its frame t is finished.  hmr_gpu_enc_create_engine_twin makes
 * another object for the same engine - it shares the engine's persistent state (CTU records, the WPP threads' mode buffers) with `of` and has its own pictures, filter state
 * and sub-stream buffers; encs[j] for j >= num_enc_engines has to be a twin of encs[j - num_enc_engines] (or the other way round).  The launch then starts an engine's next
 * frame when its previous one is finished and hands on the average distortion it leaves (hmr_encoder_lib.c:3217-3262) inside the launch; the engines never idle between the
 * frames of a call.  An I frame inside the sequence must be among the last num_enc_engines frames of its call (it hands on the value of the frame before it).  Destroy the
 * twins before the object they were made from. */
int hmr_gpu_enc_create_engine_twin(hmr_gpu_ctx *ctx, hmr_gpu_enc *of, hmr_gpu_enc **out);
int hmr_gpu_enc_encode_chain(hmr_gpu_enc **encs, int n, hmr_gpu_enc *prev, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes);
int hmr_gpu_enc_export_references8(hmr_gpu_enc **encs, int n, uint8_t *dev_rows, long pitch, void *states);
int hmr_gpu_enc_import_references8(hmr_gpu_enc **encs, int n, const uint8_t *dev_rows, long pitch, const void *states);

/* ------------------------------------------------------------------------------------------------
 * 13. Phase planes of a reference picture
 *     Replaces the per-block interpolation calls of the motion search and of motion compensation - the sixteen planes of
 *     hmr_half_pixel_estimation_luma_hm / hmr_quarter_pixel_estimation_luma_hm (hmr_motion_inter.c:395,442) and
 *     hmr_motion_compensation_luma / _chroma (:1779,1860), all through low_level_funcs_t.interpolate_luma / interpolate_chroma
 *     (hmr_private.h:1077-1078, stage rules hmr_motion_inter.c:240-391) - by ONE bandwidth-bound pass per reference picture.
 *     All pointers are device memory.  pic_y / pic_u / pic_v: the padded int16 planes from the first element of their allocation,
 *     stride x rows elements each (strides multiples of 4).  out_y: 16 x stride_y x rows_y bytes, plane fy * 4 + fx = the picture
 *     displaced by (fx, fy) quarter samples, clipped to 8 bits like the final stage of the interpolation; out_u / out_v: 64 planes
 *     each, plane fy * 8 + fx in eighth samples (pic_u / pic_v / out_u / out_v may be NULL: luma only).  The planes are
 *     row-interleaved - row y of plane f starts at byte (y * 16 + f) * stride_y (chroma: (y * 64 + f) * stride_c) - so that the
 *     window a CTU reads of all of them is one stretch of memory.  Taps run over row ends
 *     linearly, as the reference's pointer arithmetic does; taps outside the allocation read zero.
 * ------------------------------------------------------------------------------------------------ */
int hmr_gpu_subpel_planes(hmr_gpu_ctx *ctx, const int16_t *pic_y, const int16_t *pic_u, const int16_t *pic_v, int stride_y, int rows_y, int stride_c, int rows_c,
			  uint8_t *out_y, uint8_t *out_u, uint8_t *out_v);

/* ------------------------------------------------------------------------------------------------
 * 14. Measurement aid (no counterpart in the reference): the vector-instruction issue rate of the device
 *     k_encode_pool is bound by the instruction issue of a few wavefronts per CU and by the latency of its dependent chains, not by HBM bandwidth; bench.py prices it
 *     against wave-instructions per second.  This measures that ceiling on the device at hand: every CU runs waves_per_simd (1 .. 4) wavefronts per SIMD, each a
 *     loop of 64 integer multiply-adds without memory accesses - on eight independent accumulators, or (dependent != 0) as ONE chain, the rate a single dependent
 *     instruction stream reaches.  *wave_instr_per_s: vector instructions per second of all wavefronts together; *ms (may be NULL): the launch's duration.
 * ------------------------------------------------------------------------------------------------ */
int hmr_gpu_probe_valu_issue(hmr_gpu_ctx *ctx, int waves_per_simd, int dependent, double *wave_instr_per_s, double *ms);
/* the same for one instruction kind: op 0 v_mad_u32_u24, 1 v_add_u32, 2 v_mov_b32, 3 v_perm_b32, 4 s_add_u32 (scalar unit) */
int hmr_gpu_probe_issue(hmr_gpu_ctx *ctx, int op, int waves_per_simd, int dependent, double *wave_instr_per_s, double *ms);

/* ------------------------------------------------------------------------------------------------
 * 15. Test aid: the block primitives the frame encoder's CTU walk runs (homerhevc_amd/csrc/enc/enc_prims.h), one call at a time
 *     k_encode_pool does not go through the table kernels of sections 1 - 7: its worker wavefront runs SPMD primitives of its own on data in LDS.  These entries run exactly
 *     those primitives - one wavefront, as in the walk - behind the flat signatures of the table functions they restate (hmr_private.h:1063-1092; the same argument meaning as
 *     hmr_gpu_sad ... hmr_gpu_inv_quant above), so that the parity sweep of the table kernels also holds the walk's primitives to the oracle (tests/test_gpu_prims.py).
 *     hmr_gpu_prim_bytes(1): sample operands the worker keeps as bytes (source block, prediction window) are narrowed first, i.e. the byte instantiations run - for
 *     calls whose samples are 0 .. 255; hmr_gpu_prim_sad then is the motion search's multi-candidate byte SAD.  Host pointers, synchronous, default context.
 * ------------------------------------------------------------------------------------------------ */
void hmr_gpu_prim_bytes(int on);
uint32_t hmr_gpu_prim_sad(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size);
uint32_t hmr_gpu_prim_ssd16b(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size);
void hmr_gpu_prim_predict(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int size);
void hmr_gpu_prim_reconst(int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int16_t *decoded, int decoded_stride, int size);
uint32_t hmr_gpu_prim_modified_variance(int16_t *p, int size, int stride, int modif);
void hmr_gpu_prim_intra_planar(int16_t *prediction, int pred_stride, int16_t *adi_pred_buff, int adi_size, int cu_size);
void hmr_gpu_prim_intra_angular(int16_t *prediction, int pred_stride, int16_t *adi_pred_buff, int adi_size, int cu_size, int cu_mode, int is_luma);
void hmr_gpu_prim_fill_reference_samples(int16_t *decoded_corner, int stride, int n, int left, int top, int bottom_left, int top_right, int bl_size, int tr_size, int16_t *adi);
void hmr_gpu_prim_adi_filter(int16_t *adi, int16_t *out, int adi_size, int n, int strong_enabled);
void hmr_gpu_prim_transform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst);
void hmr_gpu_prim_itransform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst);
void hmr_gpu_prim_quant(int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra, int slice_is_intra, int sign_hiding, int *ac_sum,
			int cu_size, int per, int rem);
void hmr_gpu_prim_inv_quant(int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem);

#ifdef __cplusplus
}
#endif
#endif /* HOMER_GPU_H */
