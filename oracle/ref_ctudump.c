/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * The reference encoder (oracle/_ref/libhomer_ref.so, compiled from /root/reference where it lies) run in lockstep with
 * an interposer on hmr_deblock_sao_pad_sync_ctu (hmr_encoder_lib.c:2386), the first call after a CTU's decisions are
 * final (wfpp_encoder_thread :2942-2948): the CTU's side-info arrays, its levels, its reconstruction before the loop
 * filters and the worker thread's mode buffers are appended to $HOMER_CTUDUMP.  tests/golden/make_ctu_golden.py turns
 * that into the fixtures the device-resident CTU encoder is compared with; tools/ctu_diff.py uses it for drill-down.
 * With HOMER_CUTRACE=file the block drivers are interposed as well and log one text line per call.
 *
 * Same command line as ref_lockstep.  Built by oracle/Makefile (needs hmr_private.h -> build container only).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "hmr_private.h"
#include "hmr_common.h"

static FILE *g_dump, *g_trace;
static void *next(const char *name)
{
	void *p = dlsym(RTLD_NEXT, name);
	if (!p) { fprintf(stderr, "ref_ctudump: %s not found\n", name); abort(); }
	return p;
}

static void put(const void *p, size_t n) { fwrite(p, 1, n, g_dump); }

/* ---- HOMER_TURNSTILE=1: a deterministic schedule for wfpp_num_threads > 1 -------------------------------------------------------------------------
 * With several WPP threads the reference reads counters that other threads update (hmr_motion_inter.c:3769-3776), so its output depends on timing.
 * The turnstile forces ONE legal interleaving - the synchronous wavefront: CTU (r, c) belongs to step t = c + 2r; the CTUs of a step start when every
 * CTU of the steps before has finished completely, none of them leaves its decision function before all of the step have (so that all read the counters
 * as of the end of step t - 1), and their post-decision sections (counter update is theirs alone; lagged filters / SAO / entropy coding in
 * hmr_deblock_sao_pad_sync_ctu) run one after the other in row order.  This is the schedule a row-parallel device executes. */
#define TS_MAX_STEPS 4096
static pthread_mutex_t ts_m = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t ts_c = PTHREAD_COND_INITIALIZER;
static int ts_on = -1, ts_frame = -1, ts_W, ts_H, ts_steps_done;
static int ts_returned[TS_MAX_STEPS], ts_completed[TS_MAX_STEPS], ts_post_done[TS_MAX_STEPS], ts_row0_returned[TS_MAX_STEPS];
static int ts_enabled(void) { if (ts_on < 0) ts_on = getenv("HOMER_TURNSTILE") ? 1 : 0; return ts_on; }
static int ts_count(int t)
{
	int r, k = 0;
	for (r = 0; r < ts_H; r++) { const int c = t - 2 * r; if (c >= 0 && c < ts_W) k++; }
	return k;
}
static void ts_enter(henc_thread_t *et, int n)
{
	int t;
	pthread_mutex_lock(&ts_m);
	if (ts_frame != (int)et->enc_engine->num_encoded_frames) {
		ts_frame = (int)et->enc_engine->num_encoded_frames;
		ts_W = et->pict_width_in_ctu; ts_H = et->pict_height_in_ctu;
		memset(ts_returned, 0, sizeof ts_returned); memset(ts_completed, 0, sizeof ts_completed); memset(ts_post_done, 0, sizeof ts_post_done); memset(ts_row0_returned, 0, sizeof ts_row0_returned);
		ts_steps_done = 0;
	}
	t = n % ts_W + 2 * (n / ts_W);
	while (ts_steps_done < t) pthread_cond_wait(&ts_c, &ts_m);
	/* thread 0 is the one that may detect a scene change (et->index == 0, hmr_motion_inter.c:3791): its CTU of the step decides first, so that what the
	 * other threads read in is_scene_change (:2916) does not depend on timing */
	if (n >= ts_W && t < ts_W)
		while (!ts_row0_returned[t]) pthread_cond_wait(&ts_c, &ts_m);
	pthread_mutex_unlock(&ts_m);
}
static void ts_leave_decision(int n)
{
	const int t = n % ts_W + 2 * (n / ts_W);
	pthread_mutex_lock(&ts_m);
	ts_returned[t]++;
	if (n < ts_W) ts_row0_returned[t] = 1;
	pthread_cond_broadcast(&ts_c);
	while (ts_returned[t] < ts_count(t)) pthread_cond_wait(&ts_c, &ts_m);
	pthread_mutex_unlock(&ts_m);
}
/* RD_FULL: the bit estimates of a decision run on the counter object et->ec = ec_list[row of the CTU the thread coded last mod threads] (wfpp_encode_select_bitstream,
 * hmr_encoder_lib.c:2311, :2332), which the thread two rows up owns too - two decisions of one step that estimate at the same time add into each other's bit
 * count and the stream changes from run to run (seen from eight CTU columns on).  Under the turnstile the decisions of a step therefore run one after the other,
 * rows top down, when rd_mode is RD_FULL: every estimate starts from its own copy of the contexts, so one at a time they do not see each other. */
static void ts_decide_in_turn(int n)
{
	const int r = n / ts_W, t = n % ts_W + 2 * r;
	int pos = 0, rr;
	for (rr = 0; rr < r; rr++) { const int c = t - 2 * rr; if (c >= 0 && c < ts_W) pos++; }
	pthread_mutex_lock(&ts_m);
	while (ts_returned[t] < pos) pthread_cond_wait(&ts_c, &ts_m);
	pthread_mutex_unlock(&ts_m);
}
static void ts_post_begin(int n)
{
	const int r = n / ts_W, t = n % ts_W + 2 * r;
	int pos = 0, rr;
	for (rr = 0; rr < r; rr++) { const int c = t - 2 * rr; if (c >= 0 && c < ts_W) pos++; }
	pthread_mutex_lock(&ts_m);
	while (ts_post_done[t] < pos) pthread_cond_wait(&ts_c, &ts_m);
	pthread_mutex_unlock(&ts_m);
}
static void ts_post_end(int n)
{
	const int t = n % ts_W + 2 * (n / ts_W);
	pthread_mutex_lock(&ts_m);
	ts_post_done[t]++;
	ts_completed[t]++;
	while (ts_steps_done < TS_MAX_STEPS && ts_completed[ts_steps_done] == ts_count(ts_steps_done) && ts_steps_done < ts_W + 2 * (ts_H - 1)) ts_steps_done++;
	pthread_cond_broadcast(&ts_c);
	pthread_mutex_unlock(&ts_m);
}
/* ---- HOMER_TURNSTILE=1 with num_enc_engines = E > 1: a deterministic schedule for the frame pipeline ---------------------------------------------------
 * The reference deals frames to E engine threads (encoder_engine_thread, hmr_encoder_lib.c:3043) that overlap: engine k+1 starts frame n+1 as soon as engine k
 * has left the input section of frame n, its WPP threads follow frame n's rows at a distance (synchro_sem[1], :2393-2445), and when a frame ends its engine
 * pushes avg_dist into the engines that are ahead (:3268-3279) - whenever that happens to be.  Its output therefore depends on timing.  The turnstile forces
 * ONE legal interleaving:
 *   * the CTU work of frame n+1 starts when the CTU work of frame n is complete (its WPP threads have returned: reconstruction filtered and padded);
 *   * the end-of-frame section of frame n (avg_dist, the push, output) runs when the CTU work of frame n+E-1 is complete, i.e. as late as the pipeline allows -
 *     the frames it pushes into have made their decisions by then, so the push has no effect on them, and frame n+E (same engine, next) starts from it.
 * What a frame then sees: the complete reconstruction of the frame before it, the frame typing state as in the single-engine run (ref_lockstep feeds frame f+1
 * after frame f+1-E came out, i.e. after frame f's CTU work), avg_dist as of the end of frame n-E (zero for the first E frames), and the persistent per-engine
 * state (ctu_info arrays, WPP thread contexts) of engine n mod E.  A device can overlap the frames as far as the reference rows it reads allow without changing
 * any of this.  Inside a frame the synchronous wavefront above applies as before. */
static pthread_mutex_t eg_m = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t eg_c = PTHREAD_COND_INITIALIZER;
static int eg_complete, eg_fed = -1, eg_count[1 << 16];
void lockstep_all_fed(int frames)
{
	pthread_mutex_lock(&eg_m);
	eg_fed = frames;
	pthread_cond_broadcast(&eg_c);
	pthread_mutex_unlock(&eg_m);
}
THREAD_RETURN_TYPE wfpp_encoder_thread(void *h)
{
	static THREAD_RETURN_TYPE (*real)(void *);
	henc_thread_t *et = (henc_thread_t *)h;
	const int engines = et->enc_engine->hvenc->num_encoder_engines, n = (int)et->enc_engine->num_encoded_frames, T = et->enc_engine->wfpp_num_threads;
	if (!real) real = next("wfpp_encoder_thread");
	if (!ts_enabled() || engines < 2) return real(h);
	if (getenv("HOMER_TS_DEBUG")) fprintf(stderr, "EG frame %d thread %d: at start gate (complete %d)\n", n, et->index, eg_complete);
	pthread_mutex_lock(&eg_m);
	while (eg_complete < n) pthread_cond_wait(&eg_c, &eg_m);
	pthread_mutex_unlock(&eg_m);
	if (getenv("HOMER_TS_DEBUG")) fprintf(stderr, "EG frame %d thread %d: runs\n", n, et->index);
	real(h);
	if (getenv("HOMER_TS_DEBUG")) fprintf(stderr, "EG frame %d thread %d: CTUs done\n", n, et->index);
	pthread_mutex_lock(&eg_m);
	if (++eg_count[n & 0xffff] == T) {
		eg_count[n & 0xffff] = 0;
		eg_complete = n + 1;
		pthread_cond_broadcast(&eg_c);
	}
	while (eg_complete < n + engines && !(eg_fed >= 0 && eg_complete >= eg_fed)) pthread_cond_wait(&eg_c, &eg_m);
	pthread_mutex_unlock(&eg_m);
	return THREAD_RETURN;
}

/* the step starts at init_ctu (hmr_encoder_lib.c:2900), before the thread looks at is_scene_change (:2916) */
ctu_info_t *init_ctu(henc_thread_t *et)
{
	static ctu_info_t *(*real)(henc_thread_t *);
	if (!real) real = next("init_ctu");
	if (ts_enabled() && et->wfpp_num_threads > 1) ts_enter(et, et->cu_current);
	return real(et);
}
uint32_t motion_inter(henc_thread_t *et, ctu_info_t *ctu)
{
	static uint32_t (*real)(henc_thread_t *, ctu_info_t *);
	uint32_t r;
	if (!real) real = next("motion_inter");
	if (!ts_enabled() || et->wfpp_num_threads < 2) return real(et, ctu);
	if (et->rd_mode == RD_FULL) ts_decide_in_turn(ctu->ctu_number);
	r = real(et, ctu);
	ts_leave_decision(ctu->ctu_number);
	return r;
}
uint32_t motion_intra(henc_thread_t *et, ctu_info_t *ctu, int gcnt)
{
	static uint32_t (*real)(henc_thread_t *, ctu_info_t *, int);
	uint32_t r;
	if (!real) real = next("motion_intra");
	if (!ts_enabled() || et->wfpp_num_threads < 2) return real(et, ctu, gcnt);
	if (et->rd_mode == RD_FULL) ts_decide_in_turn(ctu->ctu_number);
	r = real(et, ctu, gcnt);
	ts_leave_decision(ctu->ctu_number);
	return r;
}


void hmr_deblock_sao_pad_sync_ctu(henc_thread_t *et, slice_t *currslice, ctu_info_t *ctu)
{
	static void (*real)(henc_thread_t *, slice_t *, ctu_info_t *);
	if (!real) {
		real = next("hmr_deblock_sao_pad_sync_ctu");
		if (getenv("HOMER_CTUDUMP")) g_dump = fopen(getenv("HOMER_CTUDUMP"), "wb");
	}
	if (ts_enabled() && et->wfpp_num_threads > 1) ts_post_begin(ctu->ctu_number);
	if (g_dump) {
		int32_t hdr[8] = {0x43545544, et->enc_engine->num_encoded_frames, ctu->ctu_number, (int32_t)currslice->slice_type,
				  (int32_t)ctu->partition_list[0].cost, (int32_t)ctu->partition_list[0].distortion, (int32_t)ctu->partition_list[0].sum,
				  et->enc_engine->is_scene_change};
		int c, y, d;
		put(hdr, sizeof hdr);
		for (c = 0; c < 3; c++) put(ctu->cbf[c], 256);
		put(ctu->intra_mode[0], 256); put(ctu->intra_mode[1], 256);
		put(ctu->inter_mode, 256); put(ctu->tr_idx, 256); put(ctu->pred_depth, 256); put(ctu->part_size_type, 256); put(ctu->pred_mode, 256);
		put(ctu->skipped, 256); put(ctu->merge, 256); put(ctu->merge_idx, 256); put(ctu->qp, 256);
		put(ctu->mv_ref_idx[0], 256); put(ctu->mv_diff_ref_idx[0], 256);
		put(ctu->mv_ref[0], 256 * sizeof(motion_vector_t)); put(ctu->mv_diff[0], 256 * sizeof(motion_vector_t));
		for (c = 0; c < 3; c++) put(et->transform_quant_wnd[0]->pwnd[c], (c ? 1024 : 4096) * 2);
		for (c = 0; c < 3; c++) {
			int n = c ? 32 : 64, s = et->decoded_mbs_wnd[0]->window_size_x[c];
			for (y = 0; y < n; y++) put((int16_t *)et->decoded_mbs_wnd[0]->pwnd[c] + y * s, n * 2);
		}
		for (c = 0; c < 2; c++)
			for (d = 0; d < 5; d++) put(et->intra_mode_buffs[c][d], 256);
		fflush(g_dump);
	}
	real(et, currslice, ctu);
	if (ts_enabled() && et->wfpp_num_threads > 1) ts_post_end(ctu->ctu_number);
}

/* the SAO decision of every CTU, logged when it is entropy coded (wfpp_encode_ctu, hmr_encoder_lib.c:2347) */
void wfpp_encode_ctu(henc_thread_t *et, ctu_info_t *ctu)
{
	static void (*real)(henc_thread_t *, ctu_info_t *);
	static FILE *f;
	static int init;
	if (!real) real = next("wfpp_encode_ctu");
	if (!init) { init = 1; if (getenv("HOMER_SAODUMP")) f = fopen(getenv("HOMER_SAODUMP"), "w"); }
	if (f) {
		int c, k;
		fprintf(f, "SAO frame=%d ctu=%d", et->enc_engine->num_encoded_frames, ctu->ctu_number);
		for (c = 0; c < 3; c++) {
			sao_offset_t *o = &ctu->coded_params.offsetParam[c];
			fprintf(f, " | %d", o->modeIdc);
			if (o->modeIdc != SAO_MODE_OFF) {
				fprintf(f, " %d %d :", o->typeIdc, o->typeAuxInfo);
				if (o->modeIdc == SAO_MODE_NEW)
					for (k = 0; k < (o->typeIdc == SAO_TYPE_BO ? 32 : 5); k++) fprintf(f, " %d", o->offset[k]);
			}
		}
		fprintf(f, " bits=%d\n", hmr_bitstream_bitcount(et->ee->bs));
		if (getenv("HOMER_SAODUMP_STATS")) {
			int t;
			for (c = 0; c < 3; c++)
				for (t = 0; t < 5; t++) {
					fprintf(f, "  ST %d %d :", c, t);
					for (k = 0; k < (t == 4 ? 32 : 5); k++) fprintf(f, " %d/%d", (int)ctu->stat_data[c][t].diff[k], (int)ctu->stat_data[c][t].count[k]);
					fprintf(f, "\n");
				}
		}
		fflush(f);
	}
	real(et, ctu);
}


/* costs of the two SAO mode searches (hmr_sao.c:663, :854) */
void sao_derive_mode_new_rdo(henc_thread_t *wt, sao_blk_param_t **ml, int mls, sao_stat_data_t stats[][NUM_SAO_NEW_TYPES], sao_blk_param_t *mp, double *mode_cost, int se[])
{
	static void (*real)(henc_thread_t *, sao_blk_param_t **, int, sao_stat_data_t [][NUM_SAO_NEW_TYPES], sao_blk_param_t *, double *, int []);
	if (!real) real = next("sao_derive_mode_new_rdo");
	real(wt, ml, mls, stats, mp, mode_cost, se);
	if (getenv("HOMER_SAOCOST")) fprintf(stderr, "NEWCOST %.6f lambdas %.6f %.6f\n", *mode_cost, wt->enc_engine->sao_lambdas[0], wt->enc_engine->sao_lambdas[1]);
}
void sao_derive_mode_merge_rdo(henc_thread_t *wt, sao_blk_param_t **ml, int mls, int *se, sao_stat_data_t stats[][NUM_SAO_NEW_TYPES], sao_blk_param_t *mp, double *mode_cost)
{
	static void (*real)(henc_thread_t *, sao_blk_param_t **, int, int *, sao_stat_data_t [][NUM_SAO_NEW_TYPES], sao_blk_param_t *, double *);
	if (!real) real = next("sao_derive_mode_merge_rdo");
	real(wt, ml, mls, se, stats, mp, mode_cost);
	if (getenv("HOMER_SAOCOST")) fprintf(stderr, "MERGECOST %.6f type %d\n", *mode_cost, mp->offsetParam[0].typeIdc);
}

/* ---- optional per-call text trace of the block drivers ---- */
static FILE *trace(void)
{
	static int init;
	if (!init) { init = 1; if (getenv("HOMER_CUTRACE")) g_trace = fopen(getenv("HOMER_CUTRACE"), "w"); }
	return g_trace;
}

uint32_t check_rd_cost_merge_2nx2n(henc_thread_t *et, ctu_info_t *ctu, int depth, int position)
{
	static uint32_t (*real)(henc_thread_t *, ctu_info_t *, int, int);
	if (!real) real = next("check_rd_cost_merge_2nx2n");
	uint32_t r = real(et, ctu, depth, position);
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + position;
		fprintf(g_trace, "MERGE ctu=%d d=%d abs=%d dist=%u idx=%d skip=%d mv=(%d,%d) sum=%u cands=(%d,%d)(%d,%d)\n", ctu->ctu_number, depth, cu->abs_index, r, cu->merge_idx,
			cu->skipped, cu->inter_mv[0].hor_vector, cu->inter_mv[0].ver_vector, cu->sum, et->merge_mvp_candidates[0].mv_candidates[0].mv.hor_vector,
			et->merge_mvp_candidates[0].mv_candidates[0].mv.ver_vector, et->merge_mvp_candidates[0].mv_candidates[1].mv.hor_vector,
			et->merge_mvp_candidates[0].mv_candidates[1].mv.ver_vector);
	}
	return r;
}

int hmr_cu_motion_estimation(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize pst, uint threshold, unsigned int action)
{
	static int (*real)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize, uint, unsigned int);
	if (!real) real = next("hmr_cu_motion_estimation");
	int r = real(et, ctu, gcnt, depth, part_position, pst, threshold, action);
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		fprintf(g_trace, "ME ctu=%d d=%d abs=%d ret=%d mv=(%d,%d) amvp=(%d,%d)(%d,%d)\n", ctu->ctu_number, depth, cu->abs_index, r, cu->inter_mv[0].hor_vector,
			cu->inter_mv[0].ver_vector, et->amvp_candidates[0].mv_candidates[0].mv.hor_vector, et->amvp_candidates[0].mv_candidates[0].mv.ver_vector,
			et->amvp_candidates[0].mv_candidates[1].mv.hor_vector, et->amvp_candidates[0].mv_candidates[1].mv.ver_vector);
	}
	return r;
}

static void trace_pw(henc_thread_t *et, ctu_info_t *ctu, int gcnt, const char *tag)
{
	unsigned a[3] = {0, 0, 0};
	int c, x, y;
	for (c = 0; c < 3; c++) {
		int n = c ? 32 : 64, st = WND_STRIDE_2D(et->prediction_wnd[0], c);
		int16_t *p = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], c, 0, 0, gcnt, et->ctu_width);
		for (y = 0; y < n; y++) for (x = 0; x < n; x++) a[c] += (unsigned)(p[y * st + x] & 0xffff) * (unsigned)(1 + ((x + 3 * y) & 7));
	}
	fprintf(g_trace, "PW %s ctu=%d pred=%u,%u,%u\n", tag, ctu->ctu_number, a[0], a[1], a[2]);
}

int predict_inter(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize pst)
{
	static int (*real)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize);
	if (!real) real = next("predict_inter");
	int r = real(et, ctu, gcnt, depth, part_position, pst);
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		fprintf(g_trace, "PRED ctu=%d d=%d abs=%d mvcost=%d\n", ctu->ctu_number, depth, cu->abs_index, r);
		trace_pw(et, ctu, gcnt, "pred");
	}
	return r;
}

int encode_inter(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize pst)
{
	static int (*real)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize);
	if (!real) real = next("encode_inter");
	if (trace()) {      /* what the prediction window holds for this CU when the evaluation starts (quirk Q12: it may be stale) */
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		unsigned sy = 0, su = 0, sv = 0;
		int x, y, c;
		for (c = 0; c < 3; c++) {
			int n = c ? cu->size_chroma : cu->size, px = c ? cu->x_position_chroma : cu->x_position, py = c ? cu->y_position_chroma : cu->y_position;
			int st = WND_STRIDE_2D(et->prediction_wnd[0], c);
			int16_t *p = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], c, px, py, gcnt, et->ctu_width);
			unsigned a = 0;
			for (y = 0; y < n; y++) for (x = 0; x < n; x++) a += (unsigned)(p[y * st + x] & 0xffff) * (unsigned)(1 + ((x + 3 * y) & 7));
			if (c == 0) sy = a; else if (c == 1) su = a; else sv = a;
		}
		fprintf(g_trace, "EIN ctu=%d d=%d abs=%d pred=%u,%u,%u\n", ctu->ctu_number, depth, cu->abs_index, sy, su, sv);
	}
	int r = real(et, ctu, gcnt, depth, part_position, pst);
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		fprintf(g_trace, "EINTER ctu=%d d=%d abs=%d ret=%d sum=%u cbf=%d,%d,%d\n", ctu->ctu_number, depth, cu->abs_index, r, cu->sum, cu->inter_cbf[0], cu->inter_cbf[1],
			cu->inter_cbf[2]);
		trace_pw(et, ctu, gcnt, "einter");
	}
	return r;
}

uint32_t encode_intra_luma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, PartSize pst)
{
	static uint32_t (*real)(henc_thread_t *, ctu_info_t *, int, int, int, PartSize);
	if (!real) real = next("encode_intra_luma");
	uint32_t r = real(et, ctu, gcnt, depth, part_position, pst);
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		fprintf(g_trace, "ILUMA ctu=%d d=%d abs=%d ret=%u mode=%d sum=%u cost=%u\n", ctu->ctu_number, depth, cu->abs_index, r, cu->intra_mode[0], cu->sum, cu->cost);
		trace_pw(et, ctu, gcnt, "iluma");
	}
	return r;
}

/* (the pieces of an estimate, with HOMER_RDTRACE_CTX: what the counter has accumulated after each) */
void encode_residual(henc_thread_t *et, enc_env_t *ee, ctu_info_t *ctu, cu_partition_info_t *pi, int component, int gcnt)
{
	static void (*real)(henc_thread_t *, enc_env_t *, ctu_info_t *, cu_partition_info_t *, int, int);
	if (!real) real = next("encode_residual");
	real(et, ee, ctu, pi, component, gcnt);
	if (trace() && ee->type == EE_COUNTER && getenv("HOMER_RDTRACE_CTX")) fprintf(g_trace, "  RES d=%d abs=%d comp=%d frac=%llu ec=%p\n", pi->depth, pi->abs_index, component, (unsigned long long)ee->b_ctx->m_fracBits, (void *)ee);
}
void encode_qt_cbf(enc_env_t *ee, cu_partition_info_t *pi, int component, int tr_depth, int cbf)
{
	static void (*real)(enc_env_t *, cu_partition_info_t *, int, int, int);
	if (!real) real = next("encode_qt_cbf");
	real(ee, pi, component, tr_depth, cbf);
	if (trace() && ee->type == EE_COUNTER && getenv("HOMER_RDTRACE_CTX")) fprintf(g_trace, "  CBF d=%d abs=%d comp=%d trd=%d cbf=%d frac=%llu ec=%p\n", pi->depth, pi->abs_index, component, tr_depth, cbf, (unsigned long long)ee->b_ctx->m_fracBits, (void *)ee);
}
void encode_intra_dir_luma_ang(enc_env_t *ee, ctu_info_t *ctu, cu_partition_info_t *pi, int is_multiple)
{
	static void (*real)(enc_env_t *, ctu_info_t *, cu_partition_info_t *, int);
	if (!real) real = next("encode_intra_dir_luma_ang");
	real(ee, ctu, pi, is_multiple);
	if (trace() && ee->type == EE_COUNTER && getenv("HOMER_RDTRACE_CTX")) fprintf(g_trace, "  DIR d=%d abs=%d dir=%d frac=%llu ec=%p\n", pi->depth, pi->abs_index, ctu->intra_mode[0][pi->abs_index], (unsigned long long)ee->b_ctx->m_fracBits, (void *)ee);
}

/* RD_FULL: the counter's estimates (hmr_arithmetic_encoding.c:2186, :2198, :2362), for tools/ctu_diff.py --trace */
uint rd_get_intra_bits_qt(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, uint pred_depth, int is_luma, int gcnt)
{
	static uint (*real)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, uint, int, int);
	if (!real) real = next("rd_get_intra_bits_qt");
	uint r = real(et, ctu, pi, pred_depth, is_luma, gcnt);
	if (trace()) {
		char line[1024];      /* (one write per line: the threads of a step trace side by side) */
		int o = snprintf(line, sizeof line, "RDQT ctu=%d d=%d abs=%d pd=%u luma=%d bits=%u", ctu->ctu_number, pi->depth, pi->abs_index, pred_depth, is_luma, r);
		if (getenv("HOMER_RDTRACE_CTX")) {
			int i;
			o += snprintf(line + o, sizeof line - o, " ctx ");
			for (i = 0; i < NUM_CTXs; i++) o += snprintf(line + o, sizeof line - o, "%02x", et->ee->contexts[i].state);
		}
		fprintf(g_trace, "%s ec=%p\n", line, (void *)et->ec);
	}
	return r;
}
uint fast_rd_estimate_bits_intra_luma_mode(henc_thread_t *et, cu_partition_info_t *pi, uint pred_depth, int dir, int *preds, int num_preds)
{
	static uint (*real)(henc_thread_t *, cu_partition_info_t *, uint, int, int *, int);
	if (!real) real = next("fast_rd_estimate_bits_intra_luma_mode");
	uint r = real(et, pi, pred_depth, dir, preds, num_preds);
	if (trace()) fprintf(g_trace, "RDLM d=%d abs=%d dir=%d bits=%u\n", pi->depth, pi->abs_index, dir, r);
	return r;
}
uint rd_estimate_bits_intra_mode(henc_thread_t *et, ctu_info_t *ctu, cu_partition_info_t *pi, uint pred_depth, int is_luma)
{
	static uint (*real)(henc_thread_t *, ctu_info_t *, cu_partition_info_t *, uint, int);
	if (!real) real = next("rd_estimate_bits_intra_mode");
	uint r = real(et, ctu, pi, pred_depth, is_luma);
	if (trace()) fprintf(g_trace, "RDCM ctu=%d d=%d abs=%d luma=%d bits=%u\n", ctu->ctu_number, pi->depth, pi->abs_index, is_luma, r);
	return r;
}

uint32_t encode_intra_chroma(henc_thread_t *et, ctu_info_t *ctu, int gcnt, int depth, int part_position, int pst)
{
	static uint32_t (*real)(henc_thread_t *, ctu_info_t *, int, int, int, int);
	if (!real) real = next("encode_intra_chroma");
	int16_t dbg_before[32 * 32];
	int dbg = trace() && getenv("HOMER_DBG_PW") && ctu->ctu_number == 0 && depth == 3 && part_position == 0 && et->enc_engine->num_encoded_frames == 0;
	if (dbg) {
		int st = WND_STRIDE_2D(et->prediction_wnd[0], 1), x, y;
		int16_t *p = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], 1, 0, 0, gcnt, et->ctu_width);
		for (y = 0; y < 32; y++) for (x = 0; x < 32; x++) dbg_before[y * 32 + x] = p[y * st + x];
	}
	uint32_t r = real(et, ctu, gcnt, depth, part_position, pst);
	if (dbg) {
		int st = WND_STRIDE_2D(et->prediction_wnd[0], 1), x, y;
		int16_t *p = WND_POSITION_2D(int16_t *, et->prediction_wnd[0], 1, 0, 0, gcnt, et->ctu_width);
		for (y = 0; y < 32; y++) for (x = 0; x < 32; x++)
			if (dbg_before[y * 32 + x] != p[y * st + x]) fprintf(g_trace, "DBGPW U (%d,%d) %d -> %d\n", x, y, dbg_before[y * 32 + x], p[y * st + x]);
	}
	if (trace()) {
		cu_partition_info_t *cu = &ctu->partition_list[et->partition_depth_start[depth]] + part_position;
		fprintf(g_trace, "ICHROMA ctu=%d d=%d abs=%d ret=%u mode=%d\n", ctu->ctu_number, depth, cu->abs_index, r,
			et->intra_mode_buffs[1][depth][depth == 0 ? 0 : (pst == SIZE_NxN ? cu->parent->abs_index : cu->abs_index)]);
		trace_pw(et, ctu, gcnt, "ichroma");
	}
	return r;
}

void consolidate_prediction_info(henc_thread_t *et, ctu_info_t *ctu, ctu_info_t *ctu_rd, cu_partition_info_t *parent, uint32_t parent_cost, uint32_t children_cost,
				 int is_max_depth, uint32_t *cost_sum)
{
	static void (*real)(henc_thread_t *, ctu_info_t *, ctu_info_t *, cu_partition_info_t *, uint32_t, uint32_t, int, uint32_t *);
	if (!real) real = next("consolidate_prediction_info");
	if (trace())
		fprintf(g_trace, "CONS ctu=%d d=%d abs=%d parent=%u children=%u max=%d mode=%d\n", ctu->ctu_number, parent->depth, parent->abs_index, parent_cost, children_cost,
			is_max_depth, parent->prediction_mode);
	real(et, ctu, ctu_rd, parent, parent_cost, children_cost, is_max_depth, cost_sum);
}
