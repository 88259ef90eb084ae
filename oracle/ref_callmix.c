/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * Records WHICH hot-path calls the reference makes per frame (function, block size, stage flags - no
 * sample data): the "call mix" that bench.py replays as batched GPU launches and that DESIGN.md uses to
 * turn per-kernel bytes into per-frame bytes.  It overwrites hvenc_enc_t.funcs (hmr_private.h:1443) with
 * counting wrappers after HOMER_enc_init - the registration route verified in SURVEY.md §8-b - and wraps
 * the kernels that bypass the table - the scalar sad() of the sub-pel refinement (hmr_motion_inter.c:1712,1757),
 * fill_reference_samples, the sub-pel plane builders, deblock, SAO offset, padding - by ELF symbol interposition
 * (it links the shared oracle/_ref/libhomer_ref.so with -rdynamic).  Built by oracle/Makefile into oracle/_ref/ref_callmix; includes ref_lockstep.c.
 *
 * usage: ref_callmix in.yuv out.json W H frames [key=value ...]     (same keys as ref_lockstep)
 */
#define _GNU_SOURCE
#define main lockstep_main_unused
#include "ref_lockstep.c"
#undef main
#include "hmr_private.h"
#include "hmr_common.h"
#include "hmr_sse42_functions.h"

#define MAXKEYS 4096
static struct { char key[64]; long n; } g_cnt[MAXKEYS];
static int g_nkeys;
static low_level_funcs_t g_orig;

/* calls made from inside a driver that the GPU side replaces as a whole are keyed name@driver:... */
static __thread const char *g_sfx = "";
static void bump(const char *fmt, int a, int b, int c, int d)
{
	char key[64], raw[64];
	int i;
	snprintf(raw, sizeof raw, fmt, a, b, c, d);
	if (*g_sfx && !strchr(raw, '@')) {
		const char *colon = strchr(raw, ':');
		size_t n = colon ? (size_t)(colon - raw) : strlen(raw);
		snprintf(key, sizeof key, "%.*s%s%s", (int)n, raw, g_sfx, colon ? colon : "");
	} else
		snprintf(key, sizeof key, "%s", raw);
	for (i = 0; i < g_nkeys; i++)
		if (!strcmp(g_cnt[i].key, key)) { g_cnt[i].n++; return; }
	if (g_nkeys < MAXKEYS) { strcpy(g_cnt[g_nkeys].key, key); g_cnt[g_nkeys++].n = 1; }
}

static void w_c1616(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { bump("copy_16_16:%d:%d", h, w, 0, 0); g_orig.sse_copy_16_16(s, ss, d, ds, h, w); }
static void w_c168(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { bump("copy_16_8:%d:%d", h, w, 0, 0); g_orig.sse_copy_16_8(s, ss, d, ds, h, w); }
static void w_c816(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { bump("copy_8_16:%d:%d", h, w, 0, 0); g_orig.sse_copy_8_16(s, ss, d, ds, h, w); }
static uint32_t w_sad(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n) { bump("sad:%d", n, 0, 0, 0); return g_orig.sad(s, ss, p, ps, n); }
static uint32_t w_ssd(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n) { bump("ssd16b:%d:%d", n, ps == 0, 0, 0); return g_orig.ssd16b(s, ss, p, ps, n); }
static void w_predict(int16_t *o, int os, int16_t *p, int ps, int16_t *r, int rs, int n) { bump("predict:%d", n, 0, 0, 0); g_orig.predict(o, os, p, ps, r, rs, n); }
static void w_reconst(int16_t *p, int ps, int16_t *r, int rs, int16_t *d, int ds, int n) { bump("reconst:%d:%d", n, rs == 0, 0, 0); g_orig.reconst(p, ps, r, rs, d, ds, n); }
static uint32_t w_var(int16_t *p, int size, int stride, int modif) { bump("modified_variance:%d:%d", size, modif, 0, 0); return g_orig.modified_variance(p, size, stride, modif); }
static void w_planar(henc_thread_t *et, int16_t *pr, int ps, int16_t *adi, int as, int n, int sh) { bump("intra_planar:%d", n, 0, 0, 0); g_orig.create_intra_planar_prediction(et, pr, ps, adi, as, n, sh); }
static void w_ang(henc_thread_t *et, ctu_info_t *ctu, int16_t *pr, int ps, int16_t *adi, int as, int n, int mode, int luma)
{
	bump("intra_angular:%d:%d:%d", n, mode, luma, 0);
	g_orig.create_intra_angular_prediction(et, ctu, pr, ps, adi, as, n, mode, luma);
}
/* where an interpolation call comes from: 0 = direct, 1 = sub-pel plane builders, 2 = motion compensation */
static __thread int g_origin;
static const char *k_il[3] = {"interp_luma:%d:%d:%d:%d", "interp_luma@planes:%d:%d:%d:%d", "interp_luma@mc:%d:%d:%d:%d"};
static const char *k_ic[3] = {"interp_chroma:%d:%d:%d:%d", "interp_chroma@planes:%d:%d:%d:%d", "interp_chroma@mc:%d:%d:%d:%d"};
static void w_il(int16_t *s, int ss, int16_t *d, int ds, int fr, int w, int h, int v, int f, int l)
{
	bump(k_il[g_origin], w, h, (fr != 0) | (v << 1) | (f << 2) | (l << 3), 0);
	g_orig.interpolate_luma_m_compensation(s, ss, d, ds, fr, w, h, v, f, l);
}
static void w_ic(int16_t *s, int ss, int16_t *d, int ds, int fr, int w, int h, int v, int f, int l)
{
	bump(k_ic[g_origin], w, h, (fr != 0) | (v << 1) | (f << 2) | (l << 3), 0);
	g_orig.interpolate_chroma_m_compensation(s, ss, d, ds, fr, w, h, v, f, l);
}
static void w_wavg(int16_t *a, int as, int16_t *b, int bs, int16_t *d, int ds, int h, int w, int bd) { bump("weighted_average:%d:%d", w, h, 0, 0); g_orig.weighted_average_motion(a, as, b, bs, d, ds, h, w, bd); }
static void w_quant(henc_thread_t *et, int16_t *s, int16_t *d, int scan, int depth, int comp, int mode, int intra, int *ac, int n, int per, int rem)
{
	bump("quant:%d:%d:%d", n, comp, intra, 0);
	g_orig.quant(et, s, d, scan, depth, comp, mode, intra, ac, n, per, rem);
}
static void w_iquant(henc_thread_t *et, short *s, short *d, int depth, int comp, int intra, int n, int per, int rem)
{
	bump("inv_quant:%d:%d:%d", n, comp, intra, 0);
	g_orig.inv_quant(et, s, d, depth, comp, intra, n, per, rem);
}
static void w_tr(int bd, int16_t *b, int16_t *c, int bs, int w, int h, int ws, int hs, uint16_t mode, int16_t *aux)
{
	bump("transform:%d:%d", w, w == 4 && mode != REG_DCT, 0, 0);
	g_orig.transform(bd, b, c, bs, w, h, ws, hs, mode, aux);
}
static void w_itr(int bd, int16_t *b, int16_t *c, int bs, int w, int h, unsigned mode, int16_t *aux)
{
	bump("itransform:%d:%d", w, w == 4 && mode != REG_DCT, 0, 0);
	g_orig.itransform(bd, b, c, bs, w, h, mode, aux);
}
static void w_sao(henc_thread_t *t, slice_t *s, ctu_info_t *c, sao_stat_data_t st[][NUM_SAO_NEW_TYPES]) { bump("sao_stats_ctu", 0, 0, 0, 0); g_orig.get_sao_stats(t, s, c, st); }

/* Kernels outside the table are reached through the shared library's PLT, so defining the symbol in this
 * executable (linked -rdynamic) interposes it - the same route INTEGRATION.md uses to swap them. */
#include <dlfcn.h>
#define REAL(name) ({ static void *p_; if (!p_) p_ = dlsym(RTLD_NEXT, #name); p_; })
uint32_t sad(int16_t *s, uint32_t ss, int16_t *p, uint32_t ps, int n)
{
	bump("sad_direct:%d", n, 0, 0, 0);
	return ((uint32_t(*)(int16_t *, uint32_t, int16_t *, uint32_t, int))REAL(sad))(s, ss, p, ps, n);
}
void fill_reference_samples(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, int adi_size, int16_t *dec, int stride, int n, int comp, int filt)
{
	bump("fill_reference_samples:%d:%d:%d", n, comp != 0, filt, 0);
	((void (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int16_t *, int, int, int, int))REAL(fill_reference_samples))(et, ctu, pi, adi_size, dec, stride, n, comp, filt);
}
int homer_loop1_motion_intra(henc_thread_t *et, ctu_info_t *ctu, ctu_info_t *ctu_rd, cu_partition_info_t *pi, int16_t *pred_buff, int pred_buff_stride,
			     int16_t *orig_buff, int orig_buff_stride, int16_t *decoded_buff, int decoded_buff_stride, int depth, int curr_depth, int size,
			     int size_shift, int part_size_type, int adi_size, int best_pred_modes[3], double best_pred_cost[3])
{
	int r;
	bump("intra_search:%d", size, 0, 0, 0);
	g_sfx = "@search";
	r = ((int (*)(henc_thread_t *, ctu_info_t *, ctu_info_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int16_t *, int, int, int, int, int, int, int,
		      int *, double *))REAL(homer_loop1_motion_intra))(et, ctu, ctu_rd, pi, pred_buff, pred_buff_stride, orig_buff, orig_buff_stride, decoded_buff,
								       decoded_buff_stride, depth, curr_depth, size, size_shift, part_size_type, adi_size,
								       best_pred_modes, best_pred_cost);
	g_sfx = "";
	return r;
}
/* encode_intra_luma calls whose whole luma decision (search + one-level transform tree + consolidation) is one device-side chain on the GPU side:
 * 2Nx2N, no CABAC bit estimate in the comparison, tree exactly one level deep (hmr_motion_intra.c:1418-1438) - the condition oracle/ref_swap.c routes on.
 * Counted on top of the intra_search / intra_tu calls it contains (those keys keep counting every call). */
uint32_t encode_intra_luma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize part_size_type)
{
	int log2cu = et->max_cu_size_shift - depth, cu_min, mtp;
	if (log2cu < et->min_tu_size_shift + et->max_intra_tr_depth - 1) cu_min = et->min_tu_size_shift;
	else {
		cu_min = log2cu - (et->max_intra_tr_depth - 1);
		if (cu_min > MAX_TU_SIZE_SHIFT) cu_min = MAX_TU_SIZE_SHIFT;
	}
	mtp = et->max_cu_size_shift - cu_min;
	if (et->performance_mode >= PERF_FAST_COMPUTATION) mtp = (depth + 2 <= mtp) ? depth + 2 : ((depth + 1 <= mtp) ? depth + 1 : mtp);
	if (part_size_type == SIZE_2Nx2N && et->rd_mode != RD_FULL && mtp == depth + 1 && (depth > 0 || et->max_cu_size == MAX_CU_SIZE))
		bump("intra_cu:%d", et->max_cu_size >> depth, 0, 0, 0);
	return ((uint32_t (*)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize))REAL(encode_intra_luma))(et, ctu, gcnt, depth, part_position, part_size_type);
}
/* encode_intra_chroma calls that the GPU side takes as one chroma CU driver (search + the U / V TUs along a luma tree of at most one level; the condition
 * oracle/ref_swap.c routes on): keyed intra_chroma_cu:<chroma size>:<TUs split>, and every table call made inside carries @chroma. */
uint32_t encode_intra_chroma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, int part_size_type)
{
	cu_partition_info_t *pi = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
	int routed = part_size_type == SIZE_2Nx2N && et->rd_mode != RD_FULL && pi->size_chroma >= 4 && (depth > 0 || et->max_cu_size == MAX_CU_SIZE), split = 0, k;
	uint32_t r;
	if (routed) {
		split = et->tr_idx_buffs[depth][pi->abs_index];
		for (k = 0; k < pi->num_part_in_cu; k++)
			if (et->tr_idx_buffs[depth][pi->abs_index + k] != split) routed = 0;
		if (split > 1 || (depth == 0 && split != 1)) routed = 0;
	}
	if (routed) {
		bump("intra_chroma_cu:%d:%d", pi->size_chroma, split && pi->size_chroma > 4, 0, 0);
		g_sfx = "@chroma";
	}
	r = ((uint32_t (*)(henc_thread_t *, ctu_info_t *, int, int, int, int))REAL(encode_intra_chroma))(et, ctu, gcnt, depth, part_position, part_size_type);
	g_sfx = "";
	return r;
}
uint encode_intra_cu(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, int depth, int cu_mode, PartSize part_size_type, int *curr_sum, int gcnt)
{
	uint r;
	bump("intra_tu:%d", pi->size, 0, 0, 0);
	g_sfx = "@itu";
	r = ((uint (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_intra_cu))(et, ctu, pi, depth, cu_mode, part_size_type,
														    curr_sum, gcnt);
	g_sfx = "";
	return r;
}
int encode_inter_cu(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *cu, int depth, PartSize part_size_type, int *curr_sum, int gcnt)
{
	int r;
	bump("inter_tu:%d:%d", cu->size, 0, 0, 0);
	g_sfx = "@etu";
	r = ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, PartSize, int *, int))REAL(encode_inter_cu))(et, ctu, cu, depth, part_size_type, curr_sum, gcnt);
	g_sfx = "";
	return r;
}
int encode_inter_cu_chroma(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *cu, int component, int depth, PartSize part_size_type, int *curr_sum, int gcnt)
{
	int r;
	bump("inter_tu:%d:%d", ((cu->size_chroma != 2) ? cu : cu->parent)->size_chroma, component, 0, 0);
	g_sfx = "@etu";
	r = ((int (*)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, int, int, PartSize, int *, int))REAL(encode_inter_cu_chroma))(et, ctu, cu, component, depth, part_size_type,
															   curr_sum, gcnt);
	g_sfx = "";
	return r;
}
void hmr_half_pixel_estimation_luma_hm(henc_thread_t *et, int16_t *r, int rs, cu_partition_info_t *cu, int w, int h, int sh, motion_vector_t *mv)
{
	bump("half_pel_planes:%d", w, 0, 0, 0);
	g_origin = 1;
	((void (*)(henc_thread_t *, int16_t *, int, cu_partition_info_t *, int, int, int, motion_vector_t *))REAL(hmr_half_pixel_estimation_luma_hm))(et, r, rs, cu, w, h, sh, mv);
	g_origin = 0;
}
void hmr_quarter_pixel_estimation_luma_hm(henc_thread_t *et, int16_t *r, int rs, cu_partition_info_t *cu, int w, int h, int sh, motion_vector_t *mv)
{
	bump("quarter_pel_planes:%d", w, 0, 0, 0);
	g_origin = 1;
	((void (*)(henc_thread_t *, int16_t *, int, cu_partition_info_t *, int, int, int, motion_vector_t *))REAL(hmr_quarter_pixel_estimation_luma_hm))(et, r, rs, cu, w, h, sh, mv);
	g_origin = 0;
}
void hmr_motion_compensation_luma(henc_thread_t *et, cu_partition_info_t *cu, int16_t *ref, int rs, int16_t *pred, int ps, int w, int h, int sh, motion_vector_t *mv, int bi)
{
	bump("mc_luma:%d:%d:%d:%d", w, h, (mv->hor_vector & 3) != 0, (mv->ver_vector & 3) != 0);
	g_origin = 2;
	((void (*)(henc_thread_t *, cu_partition_info_t *, int16_t *, int, int16_t *, int, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_luma))(et, cu, ref, rs, pred, ps, w, h, sh, mv, bi);
	g_origin = 0;
}
void hmr_motion_compensation_chroma(henc_thread_t *et, int16_t *ref, int rs, int16_t *pred, int ps, int size, int sh, motion_vector_t *mv, int bi)
{
	bump("mc_chroma:%d:%d:%d", size, (mv->hor_vector & 7) != 0, (mv->ver_vector & 7) != 0, 0);
	g_origin = 2;
	((void (*)(henc_thread_t *, int16_t *, int, int16_t *, int, int, int, motion_vector_t *, int))REAL(hmr_motion_compensation_chroma))(et, ref, rs, pred, ps, size, sh, mv, bi);
	g_origin = 0;
}
void hmr_deblock_filter_cu(henc_thread_t *et, slice_t *s, ctu_info_t *ctu, int dir)
{
	bump("deblock_ctu:%d", dir, 0, 0, 0);
	((void (*)(henc_thread_t *, slice_t *, ctu_info_t *, int))REAL(hmr_deblock_filter_cu))(et, s, ctu, dir);
}
void sao_offset_ctu(henc_thread_t *et, ctu_info_t *ctu, sao_blk_param_t *p)
{
	bump("sao_offset_ctu", 0, 0, 0, 0);
	((void (*)(henc_thread_t *, ctu_info_t *, sao_blk_param_t *))REAL(sao_offset_ctu))(et, ctu, p);
}
void reference_picture_border_padding_ctu(wnd_t *w, ctu_info_t *ctu)
{
	bump("pad_ctu", 0, 0, 0, 0);
	((void (*)(wnd_t *, ctu_info_t *))REAL(reference_picture_border_padding_ctu))(w, ctu);
}

static void dump_frame(FILE *fo, int frame, int first)
{
	int i;
	fprintf(fo, "%s\n {\"frame\": %d, \"calls\": {", first ? "" : ",", frame);
	for (i = 0; i < g_nkeys; i++) fprintf(fo, "%s\"%s\": %ld", i ? ", " : "", g_cnt[i].key, g_cnt[i].n);
	fprintf(fo, "}}");
	g_nkeys = 0;
}

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: %s in.yuv out.json W H frames [key=value ...]\n", argv[0]); return 2; }
	const char *in = argv[1], *out = argv[2];
	int W = atoi(argv[3]), H = atoi(argv[4]), N = atoi(argv[5]), i, force_intra = 0;
	HVENC_Cfg c;
	memset(&c, 0, sizeof c);
	c.size = sizeof c; c.width = W; c.height = H; c.profile = PROFILE_MAIN;
	c.gop_size = 1; c.num_b = 0; c.intra_period = 100; c.qp = 32; c.bitrate_mode = BR_FIXED_QP; c.bitrate = 20000;
	c.wfpp_num_threads = 1; c.wfpp_enable = 1; c.num_enc_engines = 1; c.sample_adaptive_offset = 1; c.performance_mode = 2; c.rd_mode = 2;
	c.max_intra_tr_depth = 2; c.max_inter_tr_depth = 1; c.motion_estimation_precision = QUARTER_PEL; c.frame_rate = 25;
	c.num_ref_frames = 1; c.cu_size = 64; c.max_pred_partition_depth = 4; c.sign_hiding = 1; c.chroma_qp_offset = 2; c.reinit_gop_on_scene_change = 1;
	for (i = 6; i < argc; i++) {
		char *eq = strchr(argv[i], '=');
		if (!eq) continue;
		*eq = 0;
		const char *k = argv[i], *v = eq + 1;
		if (!strcmp(k, "qp")) c.qp = atoi(v);
		else if (!strcmp(k, "perf")) c.performance_mode = atoi(v);
		else if (!strcmp(k, "rd")) c.rd_mode = atoi(v);
		else if (!strcmp(k, "sao")) c.sample_adaptive_offset = atoi(v);
		else if (!strcmp(k, "force_intra")) force_intra = atoi(v);
		else if (!strcmp(k, "intra_tr")) c.max_intra_tr_depth = atoi(v);
		else { fprintf(stderr, "unknown key %s\n", k); return 2; }
	}
	c.vbv_size = c.bitrate; c.vbv_init = (int)(c.bitrate * 0.35);
	void *h = HOMER_enc_init();
	hvenc_enc_t *hv = (hvenc_enc_t *)h;
	g_orig = hv->funcs;
	hv->funcs.sse_copy_16_16 = w_c1616; hv->funcs.sse_copy_16_8 = w_c168; hv->funcs.sse_copy_8_16 = w_c816;
	hv->funcs.sad = w_sad; hv->funcs.ssd16b = w_ssd; hv->funcs.predict = w_predict; hv->funcs.reconst = w_reconst;
	hv->funcs.modified_variance = w_var; hv->funcs.create_intra_planar_prediction = w_planar; hv->funcs.create_intra_angular_prediction = w_ang;
	hv->funcs.interpolate_luma_m_compensation = w_il; hv->funcs.interpolate_luma_m_estimation = w_il; hv->funcs.interpolate_chroma_m_compensation = w_ic;
	hv->funcs.weighted_average_motion = w_wavg; hv->funcs.quant = w_quant; hv->funcs.inv_quant = w_iquant;
	hv->funcs.transform = w_tr; hv->funcs.itransform = w_itr; hv->funcs.get_sao_stats = w_sao;
	FILE *fi = fopen(in, "rb"), *fo = fopen(out, "w");
	if (!fi || !fo) { fprintf(stderr, "cannot open files\n"); return 1; }
	if (!HOMER_enc_control(h, HOMER_SETCFG, &c)) { fprintf(stderr, "SETCFG failed\n"); return 1; }
	unsigned char *y = malloc((size_t)W * H), *u = malloc((size_t)W * H / 4), *v = malloc((size_t)W * H / 4);
	encoder_in_out_t inf, os, rec;
	memset(&inf, 0, sizeof inf); memset(&os, 0, sizeof os); memset(&rec, 0, sizeof rec);
	os.stream.streams[0] = malloc(0x4000000);
	long bytes = 0;
	int fed = 0;
	fprintf(fo, "{\"width\": %d, \"height\": %d, \"qp\": %d, \"perf\": %d, \"rd\": %d, \"force_intra\": %d, \"frames\": [", W, H, c.qp, c.performance_mode, c.rd_mode, force_intra);
	while (fed < N && fread(y, 1, (size_t)W * H, fi) == (size_t)W * H && fread(u, 1, (size_t)W * H / 4, fi) == (size_t)W * H / 4 &&
	       fread(v, 1, (size_t)W * H / 4, fi) == (size_t)W * H / 4) {
		inf.stream.streams[0] = y; inf.stream.streams[1] = u; inf.stream.streams[2] = v;
		inf.stream.data_stride[0] = W; inf.stream.data_stride[1] = inf.stream.data_stride[2] = W / 2;
		inf.pts = fed;
		inf.image_type = force_intra ? IMAGE_I : IMAGE_AUTO;
		HOMER_enc_encode(h, &inf);
		while (!drain_one(h, &rec, &os, NULL, NULL, W, H, &bytes)) usleep(100);
		dump_frame(fo, fed, fed == 0);
		fed++;
	}
	fprintf(fo, "\n]}\n");
	fclose(fo);
	_exit(0);
}
