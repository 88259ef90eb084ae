// Frame encoder on the device (include/homer_gpu.h section 12).  The decision code is enc/enc_ctu.h, instantiated for the 64-lane group.
//
// Row-per-thread schedule (wfpp_num_threads > 1, the product's mode): the reference's WPP threads (wfpp_encoder_thread, hmr_encoder_lib.c:2849-2975)
// pinned to the synchronous wavefront - CTU (row, c) belongs to step c + 2 row, and a step may start when the step before it is complete.
// k_encode_pool runs that as a task pool: persistent workgroups (a worker wavefront + NHELP helper wavefronts, two workgroups per CU) claim the next
// CTU of ANY picture of the launch whose step is open, load the owning WPP thread's state (mode buffers, "seen intra"), encode, store the state and
// close the step with a release store when they were its last CTU.  No workgroup waits for work that is not already running.
//
// Single-thread order (wfpp_num_threads = 1): the output has to be what the reference produces with ONE thread in raster order, and two inputs
// of a CTU depend on every CTU before it in that order; enc/enc_sched.h explains the guess / verify / re-encode scheme.  On the device it is:
// k_encode_ctus (workgroup r = the worker of CTU row r, waits for row r - 1 to be two CTUs ahead - a cooperative launch, so that it fails at launch
// time when the rows cannot all be resident; pass 0: all CTUs; later passes: only the CTUs marked wrong or whose neighbours changed), k_sched_scan
// (the true chains, one thread per 4x4 unit column), k_sched_check (one wavefront per CTU replays its logs against the truth), repeated until
// nothing is wrong, then k_sched_finish.
//
// Before the CTU stage of a P frame k_subpel.hip writes the reference picture at every sub-sample phase (planes borrowed from g_plane_pool; overlapped
// frames of a sequence - hmr_gpu_enc_encode_chain - get them from S tasks of the launch instead).  Behind a CTU's decisions everything else of the frame is
// post-decision TASKS of the same launch (enc/enc_post.h: deblocking, SAO, CABAC of the CTU rows' sub-streams, padding); the single-thread order, whose
// decisions are final only after its verification passes, runs them as one more launch (k_post_frame).  The host assembles the access unit.
#include <stddef.h>
#include <stdlib.h>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "enc/enc_sched.h"
#include "enc/enc_host.h"
#include "enc/enc_post.h"

using namespace henc;

// k_subpel.hip: the phase planes of a reference picture, queued on `stream` (all of them / one component's)
int hmr_subpel_plane_on(hipStream_t stream, int comp, const int16_t *pic, int stride, int rows, uint8_t *out);
int hmr_subpel_planes_on(hipStream_t stream, const int16_t *pic_y, const int16_t *pic_u, const int16_t *pic_v, int stride_y, int rows_y, int stride_c, int rows_c, uint8_t *out_y,
			 uint8_t *out_u, uint8_t *out_v);

struct EncDev {
	const Seq *seq;
	const FrameCtx *frame;
	const DevTables *tables;
	const Geo *geo;
	CtuInfo *ctus, *ctus_start;   // the CTUs (persistent across frames) and their state when the frame started
	WorkSlow *work_slow;          // one per CTU row: the transform / decoded windows (the rest of a worker's state is in LDS)
	int16_t *coeff;
	int *progress;                // [hctu] CTUs passed per row in the running pass
	uint32_t *prefix;             // [hctu][wctu + 1] running count of intra partitions along each row (first pass)
	unsigned long long *prof;     // [hctu][PF_COUNT] phase timers (profiling build)
	uint8_t *guess, *truth, *outtok;   // [nctu][MODE_STATE_BYTES]: mode state a CTU was given / should have been given / left behind (tokens)
	uint8_t *chain_start, *chain_end;  // [MODE_STATE_BYTES] the single thread's mode buffers before the first / after the last CTU
	int *valid, *dirty;           // [nctu]
	unsigned long long *hash;     // [nctu] digest of what other CTUs can see of a CTU
	uint32_t *intra_before, *used_intra, *used_parts;   // [nctu] true intra count before the CTU; the counters it was given
	uint8_t *rowstate;            // [hctu][ROW_STATE_BYTES] row-per-thread schedule: what a WPP thread carries from CTU to CTU and from frame to frame - its mode buffers and its prediction window
	int *thread_seen;             // [threads] lockstep schedule: has the WPP thread ever taken the intra walk (Work::thread_seen_intra)
	int *row0_checked;            // lockstep schedule: steps for which row 0 has made its scene-change check
	int *counters;                // [0] CTUs found wrong by the last check, [1] CTU encodes of the frame, [2] the CTU at which a scene change is detected (-1: none)
	int threads;                  // lockstep schedule: wfpp_num_threads (row r is encoded by thread r % threads)
	int dep, dep_full;            // overlapping frames of a sequence (hmr_gpu_enc_encode_chain): the picture of this launch whose final picture this one predicts from, or -1
	int after, pad_after_;        // ... the picture of this launch that the same engine encodes before this one (it has to be finished: the engine's persistent state), or -1
	FrameCtx *next_frame;         // ... the frame parameters of the picture the same engine encodes next in this launch (it starts from this picture's average distortion), or nullptr
	double *fin;                  // [0] the picture's distortion total (frame_acc_dist, enc_host.h), written by the worker that completes the picture's last task
	RcFrame *rc_dyn;              // rate control: the frame's parameters after a scene change moved them (hmr_rc_change_pic_mode), [0]; valid once counters[2] >= 0
	// RD_FULL (enc_rdo.h): what every CTU's bit estimates copy - rd_src[n] = kind << 28 | slot << 24 | index; kind 0: all-zero states (rd_init), 1: the initial
	// states of the slice of the frame in ring slot `slot` (rd_init + (1 + slot) x RD_CTX_BYTES), 3: the states after coded CTU `index` of that frame (rd_ring)
	const int *rd_src;
	const uint8_t *rd_init, *rd_ring;
	PostPic post;                 // the post-decision stage of the picture (enc_post.h): deblocking, SAO, entropy coding, padding as tasks of the CTU kernel
};

__device__ __forceinline__ void wave_copy_words(void *dst, const void *src, int bytes, int tid)
{
	uint32_t *d = (uint32_t *)dst;
	const uint32_t *s = (const uint32_t *)src;
	for (int i = tid; i < bytes / 4; i += 64) d[i] = s[i];
}
// (16 bytes per lane and step; both sides 16-byte aligned, bytes a multiple of 1024)
__device__ __forceinline__ void wave_copy_quads(void *dst, const void *src, int bytes, int tid)
{
	uint4 *d = (uint4 *)dst;
	const uint4 *s = (const uint4 *)src;
#pragma unroll 4
	for (int i = tid; i < bytes / 16; i += 64) d[i] = s[i];
}
// The thread's prediction window (Work::pred_y, pred_c: 64 x 64 + 2 x 32 x 32 samples).  A merge candidate whose vector points outside the padded reference
// picture is evaluated on whatever the window holds (check_rd_cost_merge_2nx2n leaves out the motion compensation and nothing else,
// hmr_motion_inter.c:3651) - for the first CUs of a CTU that is what the thread's CTU before left there.  So the window travels with the thread like the mode
// buffers do (found by tools/encoder_fuzz.py --gpu: 400x104, clip 657909, QP 22, a 122-sample vector next to the right picture edge).
constexpr int PRED_STATE_BYTES = (64 * 64 + 2 * 32 * 32) * (int)sizeof(pred_t), ROW_STATE_BYTES = MODE_STATE_BYTES + PRED_STATE_BYTES;
static_assert(offsetof(Work, pred_c) == offsetof(Work, pred_y) + 64 * 64 * sizeof(pred_t) && sizeof(((Work *)nullptr)->pred_c) == 2 * 32 * 32 * sizeof(pred_t), "the prediction windows are one block of Work");
static_assert(MODE_STATE_BYTES % 16 == 0 && offsetof(Work, pred_y) % 16 == 0, "wave_copy_quads alignment");

// LDS of a row worker: its Work, a copy of the CTU's partition nodes and of the partition geometry
constexpr size_t LDS_WORK = (sizeof(Work) + 15) & ~(size_t)15, LDS_NODES = (sizeof(Node) * NODE_SLOTS + 15) & ~(size_t)15,
		 LDS_GEO = NHELP * HSCRATCH_ELEMS * 2;   // the helpers' scratch (per helper: 2 x 1024 coefficients - a 32 x 32 chroma TU - and a chroma neighbour array): behind Work and the nodes, so
		                                         // that the post-decision stage, which runs between two CTUs when the helpers have no job, can use the three as one scratch area.
		                                         // (The partition geometry, whose place this was, is in constant memory: enc_common.h GeoTable.)
namespace henc { __constant__ Geo henc_geo_table[NNODES]; }
// Two things that used to sit in LDS do not any more, so that TWO row workers fit a CU (80 KB each): the CTU's side-info record (the worker reads and writes
// it in HBM: measured in round 2 to make no difference) and the TU tables (FastTables: transform bases, scans, quantiser cells - from DevTables through L2
// instead: 1-2 % per worker, against twice the workers).  Set to true to get them back (one worker per CU).
constexpr bool LDS_KEEPS_CTU_RECORD = false, LDS_KEEPS_TU_TABLES = false;
static_assert(LDS_KEEPS_TU_TABLES == (HENC_TU_TABLES_IN_LDS != 0), "enc_platform.h: HENC_FT");
constexpr size_t LDS_SEQ = (sizeof(Seq) + sizeof(FrameCtx) + 31) & ~(size_t)15, LDS_CTU = LDS_KEEPS_CTU_RECORD ? (sizeof(CtuPublic) + 15) & ~(size_t)15 : 0,
		 LDS_FT = LDS_KEEPS_TU_TABLES ? (sizeof(FastTables) + 15) & ~(size_t)15 : 0;
#if defined(HENC_PROFILE)
constexpr size_t LDS_BOX = LDS_BOX_BYTES + (1 + NHELP) * LDS_ENC_BYTES, LDS_HSCRATCH = 0, LDS_RD = (sizeof(WorkRd) + 15) & ~(size_t)15;
constexpr size_t LDS_BYTES = HENC_LDS_PROF_OFFSET + 2 * PP_COUNT * 8;   // the primitive timers sit at HENC_LDS_PROF_OFFSET
constexpr size_t LDS_FT_OFFSET = LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX + LDS_HSCRATCH;
static_assert(LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX + LDS_HSCRATCH + LDS_FT + LDS_RD <= HENC_LDS_PROF_OFFSET, "profile table overlaps the worker state");
#else
constexpr size_t LDS_BOX = LDS_BOX_BYTES + (1 + NHELP) * LDS_ENC_BYTES, LDS_HSCRATCH = 0, LDS_RD = (sizeof(WorkRd) + 15) & ~(size_t)15;
constexpr size_t LDS_BYTES = LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX + LDS_HSCRATCH + LDS_FT + LDS_RD;
constexpr size_t LDS_FT_OFFSET = LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX + LDS_HSCRATCH;
#endif
static_assert(sizeof(SubpelScratch) <= LDS_WORK + LDS_NODES + LDS_GEO, "task S works where the post stage does");
static_assert(sizeof(PostScratch) <= LDS_WORK + LDS_NODES + LDS_GEO, "the post stage works in what is idle between two CTUs: the worker's Work area, the CTU's partition nodes, the helpers' scratch");
static_assert(LDS_OFF_RD == (int)(LDS_BYTES - LDS_RD) && LDS_OFF_ENC == (int)(LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX_BYTES) || LDS_FT != 0 || LDS_CTU != 0, "the RD_FULL arrays are the tail of a worker's LDS: launches without RD_FULL pictures leave them out");
static_assert(LDS_OFF_NODES == (int)LDS_WORK && LDS_OFF_SEQ == (int)(LDS_WORK + LDS_NODES + LDS_GEO) && LDS_OFF_BOX == (int)(LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU),
	      "enc_common.h: the fixed places of Enc's LDS members");
#if !defined(HENC_WAVES_PER_EU)
#define HENC_WAVES_PER_EU 2      // wavefronts per SIMD the pool kernel's register budget allows (512 / this registers per lane)
#endif
constexpr int ENC_THREADS = 64 * (1 + NHELP);   // the row worker + its helper wavefronts: one wavefront per SIMD of the CU
static_assert(LDS_BYTES <= 160 * 1024, "a workgroup has 160 KiB of LDS on gfx950");
// workers a CU holds: by LDS (160 KiB in 1280-byte granules on gfx950) and by wavefront slots; a launch without RD_FULL pictures asks for LDS_BYTES - LDS_RD
constexpr int workers_per_cu(size_t lds_bytes)
{
	const int by_lds = (int)((160 * 1024) / ((lds_bytes + 1279) / 1280 * 1280)), by_waves = 4 * HENC_WAVES_PER_EU / (1 + NHELP);
	return by_lds < by_waves ? by_lds : by_waves;
}
constexpr int WORKERS_PER_CU = workers_per_cu(LDS_BYTES - LDS_RD);
#if defined(HENC_PRINT_LDS)
static_assert(LDS_BYTES - LDS_RD == 0 && LDS_WORK == 0 && LDS_NODES == 0 && LDS_BOX == 0 && sizeof(PostScratch) == 0 && WORKERS_PER_CU == 0, "sizes");
#endif

// a helper wavefront: run the jobs the worker posts (HelperBox, enc_common.h) until it says quit
__device__ void helper_loop(HelperBox *box, int h, int16_t *scratch)
{
	WaveGrp g{(int)(threadIdx.x & 63)};
	extern __shared__ __align__(16) uint8_t lds[];
	Enc &e = *(Enc *)(lds + LDS_OFF_ENC + (1 + h) * LDS_ENC_BYTES);      // (this helper's own context; LDS starts zeroed)
	HENC_ENC_IN_LDS(e);
	const Enc &worker = *(const Enc *)(lds + LDS_OFF_ENC);
	for (int seq = 1;; seq++) {
		while (__hip_atomic_load(&box->cmd[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);   // (two workgroups share a CU now: a helper that spins takes issue cycles from the other workgroup's worker on its SIMD)
		const int job = box->job[h];
		if (job == HJOB_QUIT) return;
		if (job == HJOB_NEW_CTU) {
			wave_copy_words(&e, &worker, (int)sizeof(Enc), g.tid);
			g.sync();
			e.box = nullptr;
			e.on_helper = 1;
			e.scratch_a = scratch;
			e.scratch_b = scratch + 1024;
			e.adi_c = scratch + 2048;
			e.prof = nullptr;
		}
		const int *a = box->a[h];
		uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0;      // (a job for both chroma planes reports U in r0 .. r2, V in r3 .. r5 - the SSD job: r0, r1)
		switch (job) {
		case HJOB_INTER_TU: {
			int sum = 0;
			uint32_t raw = 0;
			r0 = encode_inter_tu(g, e, a[0], a[1] == COMP_UV ? COMP_U : a[1], a[2], a[3], &sum, &raw);
			r1 = (uint32_t)sum;
			r2 = raw;
			if (a[1] == COMP_UV) {
				r3 = encode_inter_tu(g, e, a[0], COMP_V, a[2], a[3], &sum, &raw);
				r4 = (uint32_t)sum;
				r5 = raw;
			}
			break;
		}
		case HJOB_SYNC_CU:
			if (a[1] == COMP_UV) sync_cu_chroma_both(g, e, a[0], a[2], a[3], a[4], a[5]);
			else sync_cu_comp(g, e, a[0], a[2], a[3], a[4], a[5], a[1]);
			break;
		case HJOB_SSD: {
			const Geo &q = e.geo[a[0]];
			if (a[1] == COMP_UV) {
				// both planes in one pass (curr_c[1] / pred_c[1] lie 32 x 32 samples behind curr_c[0] / pred_c[0])
				const int n = q.size_chroma, l = ilog2i(n);
				uint32_t acc[2] = {0, 0};
				for (int i = g.tid * 4; i < n * n; i += 64 * 4) {
					const int o = q.yc * 32 + q.xc + (i >> l) * 32 + (i & (n - 1));
#pragma unroll
					for (int c = 0; c < 2; c++) {
						const S4 va = ld4(e.w->curr_c[c] + o), vb = ld4(e.w->pred_c[c] + o);
#pragma unroll
						for (int k = 0; k < 4; k++) { const int32_t dd = (int16_t)(va.v[k] - vb.v[k]); acc[c] += (uint32_t)(dd * dd); }
					}
				}
				r0 = g.sum(acc[0]);
				r1 = g.sum(acc[1]);
			} else {
				const int c = a[1] - 1;
				r0 = blk_ssd(g, e.w->curr_c[c] + q.yc * 32 + q.xc, 32, e.w->pred_c[c] + q.yc * 32 + q.xc, 32, q.size_chroma);
			}
			break;
		}
		case HJOB_CHROMA_SEARCH: {
			const int cand[5] = {a[2] & 255, (a[2] >> 8) & 255, (a[2] >> 16) & 255, (a[2] >> 24) & 255, a[3]};
			uint32_t sads[5];
			chroma_search_comp(g, e, a[0], a[1], cand, sads);
			g.sync();
			if (g.tid == 0)
				for (int k = 0; k < 5; k++) box->r[h][k] = sads[k];
			break;
		}
		case HJOB_CHROMA_TU: {
			int cs = 0;
			r0 = (uint32_t)chroma_tu_comp(g, e, a[0], a[1], a[2], a[3], a[4], a[5] & 255, a[5] >> 8, &cs);
			r1 = (uint32_t)cs;
			break;
		}
		case HJOB_INTRA_SAD: {
			const Geo &q = e.geo[a[0]];
			r0 = intra_predict_sad(g, (pred_t *)nullptr, 0, e.w->curr_y + q.y * CTU_STRIDE_Y + q.x, CTU_STRIDE_Y, a[3] ? e.w->adi_f : e.w->adi, a[1], a[2], 1);
			break;
		}
		default: break;
		}
		g.sync();
		if (g.tid == 0) {
			if (job != HJOB_CHROMA_SEARCH) { box->r[h][0] = r0; box->r[h][1] = r1; box->r[h][2] = r2; box->r[h][3] = r3; box->r[h][4] = r4; box->r[h][5] = r5; }
			__hip_atomic_store(&box->done[h], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
}

// A workgroup is a row worker (wavefront 0) and its two helpers.  rows_enter sets the mailbox up and sends the helper wavefronts into their service loop; it
// returns true on the worker only.  release_helpers lets them go (without it the workgroup never ends).
__device__ __forceinline__ bool rows_enter(unsigned lds_bytes)
{
	extern __shared__ __align__(16) uint8_t lds[];
	HelperBox *box = (HelperBox *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU);
	const int wave = (int)(threadIdx.x >> 6);
	// LDS keeps what the last workgroup on this CU left in it: everything starts from zero (the helpers' scratch included), whatever ran here before
	for (int i = (int)threadIdx.x; i < (int)(lds_bytes / 4); i += ENC_THREADS) ((uint32_t *)lds)[i] = 0;
	__syncthreads();
	if (wave > 0) {
		helper_loop(box, wave - 1, (int16_t *)(lds + LDS_WORK + LDS_NODES) + (wave - 1) * HSCRATCH_ELEMS);
		return false;
	}
	return true;
}
__device__ __forceinline__ void release_helpers(int *hseq)
{
	extern __shared__ __align__(16) uint8_t lds[];
	HelperBox *box = (HelperBox *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU);
	if ((threadIdx.x & 63) == 0)
		for (int h = 0; h < NHELP; h++) {
			box->job[h] = HJOB_QUIT;
			__hip_atomic_store(&box->cmd[h], ++hseq[h], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
}
// the work of one row worker on one picture: `row` of the picture described by d (hseq: the helpers' job counters, which live as long as the workgroup)
__device__ __forceinline__ void encode_row(const EncDev &d, int pass, int row, int *hseq)
{
	extern __shared__ __align__(16) uint8_t lds[];
	const Seq &S = *d.seq;
	const int W = S.wctu, H = S.hctu;
	WaveGrp g{(int)(threadIdx.x & 63)};
	Work *lw = (Work *)lds;
	HelperBox *box = (HelperBox *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU);
	for (int i = g.tid; i < (int)(LDS_WORK / 4); i += 64) ((uint32_t *)lds)[i] = 0;
	g.sync();
	// the sequence and frame parameters are read all through the control code: a copy next to the worker
	Seq *lseq = (Seq *)(lds + LDS_WORK + LDS_NODES + LDS_GEO);
	FrameCtx *lframe = (FrameCtx *)((uint8_t *)lseq + ((sizeof(Seq) + 15) & ~(size_t)15));
	for (int i = g.tid; i < (int)(sizeof(Seq) / 4); i += 64) ((uint32_t *)lseq)[i] = ((const uint32_t *)d.seq)[i];
	for (int i = g.tid; i < (int)(sizeof(FrameCtx) / 4); i += 64) ((uint32_t *)lframe)[i] = ((const uint32_t *)d.frame)[i];
	if (g.tid == 0) lw->slow = (HENC_GLOBAL_PTR(WorkSlow))(d.work_slow + row);
	g.sync();
	if (g.tid == 0) lframe->scene_cut_ctu = d.counters[2];
	// the transform bases, scans and this frame's quantiser lists next to the worker (enc_prims.h: FastTables)
	FastTables *lft = LDS_KEEPS_TU_TABLES ? (FastTables *)(lds + LDS_FT_OFFSET) : nullptr;
	if (lft) fast_tables_fill(g, *lft, d.tables, lframe->qp % 6, chroma_qp_table(lframe->qp + S.chroma_qp_offset) % 6);
#if defined(HENC_PROFILE)
	if (g.tid < 2 * PP_COUNT) ((unsigned long long *)(lds + HENC_LDS_PROF_OFFSET))[g.tid] = 0;
#endif
	g.sync();
	Enc &e = *(Enc *)(lds + LDS_OFF_ENC);      // (the context lives in LDS: enc_platform.h HENC_ENC_IN_LDS)
	HENC_ENC_IN_LDS(e);
	e.seq = lseq;
	e.f = lframe;
	e.T = d.tables;
	e.ft = lft;
	e.geo.p = nullptr;
	e.ctus = d.ctus;
	e.ctu = nullptr;
	e.w = lw;
	e.nodes = nullptr;
	e.nodes_fast = (Node *)(lds + LDS_WORK);
	e.node_quad = -1;
	e.on_helper = 0;
	e.ctu_g = nullptr;
	e.ctu_fast = LDS_KEEPS_CTU_RECORD ? (CtuPublic *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ) : nullptr;
	e.box = box;
	for (int h = 0; h < NHELP; h++) e.hseq[h] = hseq[h];
	e.prof = d.prof ? d.prof + (size_t)row * PF_COUNT : nullptr;
	e.timeline = nullptr;
	uint32_t *my_prefix = d.prefix + (size_t)row * (W + 1);
	if (pass <= 0 && g.tid == 0) my_prefix[0] = 0;
	int encodes = 0;
	if (g.tid == 0) lw->thread_seen_intra = 1;   // the single thread has been through the first (intra) frame's CTUs before anything else looks (enc_types.h)
	g.sync();
	for (int c = 0; c < W; c++) {
		if (row > 0) {
			HENC_PROF_T0();
			const int need = c + 2 < W ? c + 2 : W;
			while (__hip_atomic_load(&d.progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(32);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			HENC_PROF_ADD(e, PF_WAIT);
		}
		const int n = row * W + c;
		const int redo = pass == 0 || !__hip_atomic_load(&d.valid[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
				 __hip_atomic_load(&d.dirty[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (redo) {
			HENC_PROF_T0();
			uint8_t *gs = d.guess + (size_t)n * MODE_STATE_BYTES;
			uint32_t ui, up;
			if (pass > 0) {
				// start again from what the CTU looked like when the frame started, with the inputs the last check derived
				wave_copy_words(&d.ctus[n], &d.ctus_start[n], (int)offsetof(CtuInfo, n_spec_reads), g.tid);
				wave_copy_words(d.ctus[n].nodes, d.ctus_start[n].nodes, (int)sizeof(d.ctus[n].nodes), g.tid);
				wave_copy_words(gs, d.truth + (size_t)n * MODE_STATE_BYTES, MODE_STATE_BYTES, g.tid);
				ui = d.intra_before[n];
				up = (uint32_t)n * NPART;
			} else {
				if (n == 0) wave_copy_words(gs, d.chain_start, MODE_STATE_BYTES, g.tid);
				sched_known_intra(d.prefix, W, row, c, &ui, &up);
			}
			g.sync();
			if (g.tid == 0) e.w->mode_in = (const uint8_t (*)[NDEPTH][NPART])gs;      // (the guess is not written while the CTU is encoded)
			if (g.tid == 0) { d.used_intra[n] = ui; d.used_parts[n] = up; }
			const unsigned long long old_hash = d.hash[n];
			e.total_intra_partitions = ui;
			e.total_partitions = up;
			e.coeff = d.coeff + (size_t)n * 6144;
			e.ctu_qp = lframe->qp;
			g.sync();
			encode_ctu(g, e, n);
			encodes++;
			wave_copy_words(d.outtok + (size_t)n * MODE_STATE_BYTES, e.w->intra_mode_buffs, MODE_STATE_BYTES, g.tid);
			const unsigned long long h = sched_output_hash(g, S, *lframe, d.ctus[n]);
			if (g.tid == 0) { d.hash[n] = h; d.dirty[n] = 0; }
			if (pass == 0) {
				// the guesses further on: what this worker's buffers would hold if its own guesses were right
				const int snap = row + 1 < H && c == (W > 1 ? 1 : 0);
				if (c + 1 < W || snap) {
					const uint8_t *tok = &e.w->intra_mode_buffs[0][0][0];
					uint8_t *nx = gs + MODE_STATE_BYTES, *rs = d.guess + (size_t)(row + 1) * W * MODE_STATE_BYTES;
					for (int i = g.tid; i < MODE_STATE_BYTES; i += 64) {
						const uint8_t v = tok[i];
						const uint8_t r = (v & MODE_TOKEN) ? gs[(i / (NDEPTH * NPART)) * NDEPTH * NPART + (v & 7) * NPART + i % NPART] : v;
						if (c + 1 < W) nx[i] = r;
						if (snap) rs[i] = r;
					}
				}
				if (g.tid == 0) my_prefix[c + 1] = my_prefix[c] + d.ctus[n].intra_parts;
			} else if (h != old_hash && g.tid == 0) {
				if (c + 1 < W) d.dirty[n + 1] = 1;
				if (row + 1 < H) {
					if (c > 0) d.dirty[n + W - 1] = 1;
					d.dirty[n + W] = 1;
					if (c + 1 < W) d.dirty[n + W + 1] = 1;
				}
			}
			HENC_PROF_ADD(e, PF_TOTAL);
		}
		g.sync();
		if (g.tid == 0) __hip_atomic_store(&d.progress[row], c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
	for (int h = 0; h < NHELP; h++) hseq[h] = e.hseq[h];
	if (g.tid == 0 && encodes) atomicAdd(&d.counters[1], encodes);
#if defined(HENC_PROFILE)
	if (g.tid == 0 && e.prof) {
		const unsigned long long *pp = (const unsigned long long *)(lds + HENC_LDS_PROF_OFFSET);
		for (int k = 0; k < 2 * PP_COUNT; k++) e.prof[PF_PRIM0 + k] += pp[k];
	}
#endif
}

// ---- The row-per-thread schedule as a pool of CTU tasks ---------------------------------------------------------------------------------------------------
// In the synchronous wavefront (enc_sched.h) the CTUs of step t = c + 2 row of a picture may start when all its CTUs of step t - 1 are done, and nothing else
// orders them.  With one workgroup nailed to each CTU row, a step lasts as long as its slowest CTU and the other rows' CUs wait (46 % of the row workers' time
// at 1080p, profiles/r03_history.md).  Here the CTUs are tasks instead: a persistent workgroup (row worker + helpers, as before) claims the next CTU of ANY
// picture of the launch whose step is open, encodes it and closes the step when it was the step's last.  What a CTU needs from "its thread" - the mode buffers
// the reference's WPP thread carries along its rows, and its prediction window (Q12 above) - travels through EncDev::rowstate; everything else in Work is scratch
// (checked by wiping it between CTUs, oracle/enc_cpu.cpp HENC_WIPE_WORK).  No workgroup ever waits for a CTU that is not already running (the one exception - a step's CTUs wait for thread 0's
// scene-change check - is claimed first in its step), so the launch needs no co-residency: any number of workgroups makes progress.
struct PoolSeq {
	int *cur_step;      // the open step of the picture
	int *ticket;        // [steps] CTUs of the step handed out
	int *done;          // [steps] CTUs of the step finished
};
constexpr int POOL_MAX_STEPS = HENC_MAX_STEPS;  // W + 2 (H - 1): enc_host.h refuses pictures with more wavefront steps
constexpr int POOL_STRIDE = 1 + 2 * POOL_MAX_STEPS;

// rows of step t: r_lo .. r_hi (empty when r_lo > r_hi)
__device__ __forceinline__ void pool_step_rows(int t, int W, int H, int *r_lo, int *r_hi)
{
	const int lo = t - W + 1;
	*r_lo = lo <= 0 ? 0 : (lo + 1) >> 1;
	*r_hi = (t >> 1) < H - 1 ? (t >> 1) : H - 1;
}

__device__ void pool_encode_ctu(const EncDev &d, Enc &__restrict__ e, const WaveGrp &g, Seq *lseq, FrameCtx *lframe, FastTables *lft, int t, int row, int *cached_rem, const int *abort_flag)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *lseq;
	const int W = S.wctu, H = S.hctu;
	const int T = d.threads, me = row % T, c = t - 2 * row, n = row * W + c;
	// counters as of the end of step t - 1
	uint32_t ti = 0, tc = 0;
	for (int r2 = g.tid; r2 < H; r2 += 64) {
		const int have = t - 2 * r2 < 0 ? 0 : (t - 2 * r2 < W ? t - 2 * r2 : W);
		ti += d.prefix[(size_t)r2 * (W + 1) + have];
		tc += (uint32_t)have;
	}
	ti = g.sum(ti);
	tc = g.sum(tc);
	// the mode buffers of the thread that owns this row, as the CTU before left them (this row's, or the last one of row - T)
	wave_copy_words(&e.w->intra_mode_buffs[0][0][0], d.rowstate + (size_t)me * ROW_STATE_BYTES, MODE_STATE_BYTES, g.tid);
	wave_copy_quads(e.w->pred_y, d.rowstate + (size_t)me * ROW_STATE_BYTES + MODE_STATE_BYTES, PRED_STATE_BYTES, g.tid);
	if (g.tid == 0) e.w->thread_seen_intra = d.thread_seen[me];
	if (S.rd_mode == RDM_FULL) {
		// RD_FULL: the thread's shadow CTU keeps its prediction modes from CTU to CTU - all INTRA once the thread has taken the intra walk (motion_intra :2003),
		// the zeroes it was created with before (the worker's fast memory is not the thread's: enc_rdo.h)
		const int seen = d.thread_seen[me];
		for (int i = g.tid; i < NPART; i += 64) e.wrd->rd_pred_mode[i] = seen ? PM_INTRA : PM_INTER;
	}
	const int rem_y = lframe->qp % 6, rem_c = chroma_qp_table(lframe->qp + S.chroma_qp_offset) % 6;
	if (lft && (cached_rem[0] != rem_y || cached_rem[1] != rem_c)) {
		fast_tables_fill(g, *lft, d.tables, rem_y, rem_c);
		cached_rem[0] = rem_y; cached_rem[1] = rem_c;
	}
	g.sync();
	const int hrow = t / (2 * T) * T;   // thread 0's row that is inside the picture at this step, if any
	// rate control: the bits and the number of the CTUs the reference has entropy coded when this step starts
	uint32_t rc_bits = 0;
	int rc_ctus = 0;
	if (S.rd_mode == RDM_FULL) {
		const int code = d.rd_src[n], kind = code >> 28, slot = (code >> 24) & 15;
		const size_t off = (size_t)(code & 0x00ffffff) * RD_CTX_BYTES;
		e.rd_ctx = kind == 3 ? d.rd_ring + ((size_t)slot * S.nctu) * RD_CTX_BYTES + off : d.rd_init + (size_t)(kind ? 1 + slot : 0) * RD_CTX_BYTES;
	}
	if (lframe->rc.on) rc_consumed(g, d.post, W, H, t, &rc_bits, &rc_ctus);
	if (row == hrow) {
		if (g.tid == 0) {
			if (d.counters[2] < 0 && lframe->slice_type == SLICE_P && scene_cut_fires(S, *lframe, ti, tc * NPART)) {
				if (lframe->rc.on) {
					RcFrame nrc = lframe->rc;
					rc_change_pic_mode(nrc, S.reinit_gop, S.intra_period, S.nctu, nrc.sqrt_clipped_intra_period, rc_bits, rc_ctus);
					*d.rc_dyn = nrc;
				}
				d.counters[2] = n;
			}
			__hip_atomic_store(d.row0_checked, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
	} else if (hrow < H && t - 2 * hrow < W) {
		while (__hip_atomic_load(d.row0_checked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t + 1 && !__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(8);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	g.sync();
	if (g.tid == 0) lframe->scene_cut_ctu = __hip_atomic_load(&d.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	g.sync();
	e.ctu_qp = lframe->qp;
	if (lframe->rc.on) {
		// hmr_rc_calc_cu_qp at the CTU's root.  is_scene_change: set by the detecting CTU before its own walk (hmr_motion_inter.c:3795-3796) and seen by everything
		// from its step on; the picture target it moved comes from d.rc_dyn
		const int cut = lframe->scene_cut_ctu;
		const int is_sc = cut >= 0 && t >= cut % W + 2 * (cut / W);
		if (is_sc) {
			wave_copy_words(&lframe->rc, d.rc_dyn, (int)sizeof(RcFrame), g.tid);
			g.sync();
		}
		e.ctu_qp = rc_calc_cu_qp(lframe->rc, (double)rc_bits, rc_ctus, lframe->slice_type, is_sc, S.reinit_gop, S.intra_period, lframe->avg_dist, lframe->num_encoded_frames);
	}
	if (g.tid == 0) e.w->mode_in = (const uint8_t (*)[NDEPTH][NPART])(d.rowstate + (size_t)me * ROW_STATE_BYTES);      // (rewritten below, after the tokens are resolved)
	e.total_intra_partitions = ti;
	e.total_partitions = tc * NPART;
	e.coeff = d.coeff + (size_t)n * 6144;
	g.sync();
#if defined(HENC_PROFILE)
	e.prof = d.prof ? d.prof + (size_t)row * PF_COUNT : nullptr;     // (a picture has one CTU per row in flight)
	{
		HENC_PROF_T0();
		encode_ctu(g, e, n);
		HENC_PROF_ADD(e, PF_TOTAL);
	}
	if (g.tid == 0 && e.prof) {
		unsigned long long *pp = (unsigned long long *)(henc_lds + HENC_LDS_PROF_OFFSET);
		for (int k = 0; k < 2 * PP_COUNT; k++) { e.prof[PF_PRIM0 + k] += pp[k]; pp[k] = 0; }
	}
#else
	encode_ctu(g, e, n);
#endif
	resolve_mode_tokens(g, *e.w, d.ctus[n]);
	wave_copy_words(d.outtok + (size_t)n * MODE_STATE_BYTES, e.w->intra_mode_buffs, MODE_STATE_BYTES, g.tid);
	wave_copy_words(d.rowstate + (size_t)me * ROW_STATE_BYTES, &e.w->intra_mode_buffs[0][0][0], MODE_STATE_BYTES, g.tid);
	wave_copy_quads(d.rowstate + (size_t)me * ROW_STATE_BYTES + MODE_STATE_BYTES, e.w->pred_y, PRED_STATE_BYTES, g.tid);
	if (g.tid == 0) {
		d.thread_seen[me] = e.w->thread_seen_intra;
		uint32_t *my_prefix = d.prefix + (size_t)row * (W + 1);
		my_prefix[c + 1] = my_prefix[c] + d.ctus[n].intra_parts;
		atomicAdd(&d.counters[1], 1);
	}
	g.sync();
	// the CTU is decided: its post-decision tasks may run (enc_post.h)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	if (g.tid == 0) post_st_release(&d.post.rows[row].dec, c + 1);
}

// post-decision tasks of picture q that are ready, on this worker (its Work area is scratch between two CTUs); *finished counts the pictures whose last task is done
__device__ __attribute__((noinline)) int pool_post_drain(const EncDev &d, const WaveGrp &g, const Seq *lseq, const FrameCtx *lframe, Work *lw, WorkSlow *my_slow, int *finished)
{
	if (post_finished(*lseq, d.post)) return 0;
	PostCtx x;
	x.seq = lseq; x.f = lframe; x.T = d.tables; x.geo.p = nullptr; x.ctus = d.ctus; x.coeff = d.coeff; x.pic = &d.post;
	PostScratch &sc = *(PostScratch *)in_fast_memory((uint8_t *)lw);
	const int ran = post_drain(g, x, sc);
	if (ran) {
		if (g.tid == 0) lw->slow = (HENC_GLOBAL_PTR(WorkSlow))my_slow;      // (the scratch overlays the worker's Work)
		g.sync();
		// whoever completes the picture's last task says so - exactly one worker wins the claim (errors[1]: 0 -> 2) - after the picture's last duty: its distortion
		// total (hmr_encoder_lib.c:3217-3228: every WPP thread adds up its rows' CTUs in a uint32 that may wrap, the engine adds the threads in a double), from
		// which the engine's next picture in this launch takes its average distortion (end_frame, enc_host.h - the same arithmetic)
		if (post_finished(*lseq, d.post)) {
			int won = 0;
			if (g.tid == 0) won = atomicCAS(&d.post.errors[1], 0, 2) == 0;
			if (__builtin_amdgcn_readfirstlane(won)) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				const int W = lseq->wctu, H = lseq->hctu, T = d.threads < 1 ? 1 : d.threads;
				int64_t part = 0;
				for (int t = g.tid; t < T && t < H; t += 64) {
					uint32_t acc = 0;
					for (int r = t; r < H; r += T)
						for (int c = 0; c < W; c++) acc += d.ctus[r * W + c].distortion;
					part += acc;
				}
				const double total = (double)g.sum64(part);
				if (g.tid == 0) {
					d.fin[0] = total;
					if (d.next_frame) {
						double a = total;
						a /= lseq->nctu * NPART;
						a = a < .1 ? .1 : a;
						if (lframe->slice_type == SLICE_I) a *= 1.5;
						else if (__hip_atomic_load(&d.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) a *= 1.375;
						d.next_frame->avg_dist = a;      // (an I frame inside the sequence would hand on the value before it: such chains are refused by the host)
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
					post_st_release(&d.post.errors[1], 1);
					atomicAdd(finished, 1);
				}
			}
		}
	}
	return ran;
}

__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(HENC_WAVES_PER_EU))) void k_encode_pool(const EncDev *devs, int nseq, int *state, int *finished, WorkSlow *slow, unsigned long long watchdog_ticks, unsigned lds_bytes)
{
	// finished[0]: pictures whose last task is done; finished[1]: abort - a worker has waited longer than the watchdog allows (a faulted or starved peer): everybody
	// leaves and the host reports an error instead of the launch hanging
	unsigned long long t_start = wall_clock64();      // when this worker last had something to do (the watchdog's clock)
	if (!rows_enter(lds_bytes)) return;
	extern __shared__ __align__(16) uint8_t lds[];
	WaveGrp g{(int)(threadIdx.x & 63)};
	Work *lw = (Work *)lds;
	HelperBox *box = (HelperBox *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU);
	Seq *lseq = (Seq *)(lds + LDS_WORK + LDS_NODES + LDS_GEO);
	FrameCtx *lframe = (FrameCtx *)((uint8_t *)lseq + ((sizeof(Seq) + 15) & ~(size_t)15));
	FastTables *lft = LDS_KEEPS_TU_TABLES ? (FastTables *)(lds + LDS_FT_OFFSET) : nullptr;
	if (g.tid == 0) lw->slow = (HENC_GLOBAL_PTR(WorkSlow))(slow + blockIdx.x);
#if defined(HENC_PROFILE)
	if (g.tid < 2 * PP_COUNT) ((unsigned long long *)(lds + HENC_LDS_PROF_OFFSET))[g.tid] = 0;
#endif
	g.sync();
	Enc &e = *(Enc *)(lds + LDS_OFF_ENC);      // (the context lives in LDS: enc_platform.h HENC_ENC_IN_LDS)
	HENC_ENC_IN_LDS(e);
	e.seq = lseq;
	e.f = lframe;
	e.ft = lft;
	e.geo.p = nullptr;
	e.ctu = nullptr;
	e.w = lw;
	e.nodes = nullptr;
	e.nodes_fast = (Node *)(lds + LDS_WORK);
	e.node_quad = -1;
	e.on_helper = 0;
	e.ctu_g = nullptr;
	e.ctu_fast = LDS_KEEPS_CTU_RECORD ? (CtuPublic *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ) : nullptr;
	e.box = box;
	for (int h = 0; h < NHELP; h++) e.hseq[h] = 0;
	e.prof = nullptr;
	e.timeline = nullptr;
	int cached_rem[2] = {-1, -1}, cached_q = -1, idle_rounds = 0;
	int start = (int)(blockIdx.x % (unsigned)nseq);
	for (;;) {
		// look for a picture with an open step that still has CTUs to hand out: 64 pictures at a time, one per lane
		int q = -1, t = 0, k = 0;
		bool all_finished = true;
		for (int base = 0; base < nseq && q < 0; base += 64) {
			const int i = base + g.tid, cand = i < nseq ? (start + i) % nseq : -1;
			int ct = -1;
			bool open = false;
			if (cand >= 0) {
				const int *st = state + (size_t)cand * POOL_STRIDE;
				const EncDev &dd = devs[cand];
				const int W = dd.seq->wctu, H = dd.seq->hctu, steps = W + 2 * (H - 1);
				ct = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (ct < steps) {
					all_finished = false;
					int r_lo, r_hi;
					pool_step_rows(ct, W, H, &r_lo, &r_hi);
					open = __hip_atomic_load(&st[1 + ct], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r_hi - r_lo + 1;
				}
			}
			if (g.any(!all_finished)) all_finished = false;
			uint64_t m = g.ballot(open);
			// rate control: a step opens when the CTUs its decisions read the bits of have been entropy coded (enc_rc.h)
			while (m) {
				const int lane0 = __builtin_ctzll(m);
				const int q0 = __builtin_amdgcn_readlane(cand, lane0), t0 = __builtin_amdgcn_readlane(ct, lane0);
				const EncDev &d0 = devs[q0];
				bool ready = !d0.post.rc_need || rc_ready(g, d0.post, d0.seq->hctu, t0);
				// overlapping frames: the phase planes of the part of the reference picture this step's CTUs can reach (rows r - 2 .. r + 2, columns c - 3 .. c + 3:
				// +-128 x +-64 samples of search, a sample of refinement, four of filter) have been written by the S tasks of the picture it predicts from
				if (ready && d0.dep >= 0) {
					const EncDev &dd = devs[d0.dep];
					const int H0 = dd.seq->hctu, W0 = dd.seq->wctu;
					for (int base2 = 0; base2 < H0 && ready; base2 += 64) {
						const int r2 = base2 + g.tid;
						bool miss = false;
						if (r2 < H0) {
							int need = t0 - 2 * r2 + 8;
							need = need < 0 ? 0 : (need > W0 ? W0 : need);
							if (d0.dep_full) need = W0;
							miss = post_ld(&dd.post.rows[r2].s_done) < need;
						}
						if (g.any(miss)) ready = false;
					}
				}
				// the engine's previous picture of this launch is finished (the picture continues in the engine's persistent records)
				if (ready && d0.after >= 0) ready = post_ld(&devs[d0.after].post.errors[1]) == 1;
				if (ready) break;
				m &= m - 1;
			}
			if (m) {
				const int lane = __builtin_ctzll(m);
				const int qq = __builtin_amdgcn_readlane(cand, lane), tt = __builtin_amdgcn_readlane(ct, lane);
				int kk = 0;
				if (g.tid == 0) kk = atomicAdd(&state[(size_t)qq * POOL_STRIDE + 1 + tt], 1);
				kk = __builtin_amdgcn_readfirstlane(kk);
				const EncDev &dd = devs[qq];
				int r_lo, r_hi;
				pool_step_rows(tt, dd.seq->wctu, dd.seq->hctu, &r_lo, &r_hi);
				if (kk < r_hi - r_lo + 1) { q = qq; t = tt; k = kk; }
				else base -= 64;                     // somebody else took the last one: look again from the same place
			}
		}
		if (q < 0) {
			if (__hip_atomic_load(finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nseq) break;
			if (__hip_atomic_load(finished + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
			if (wall_clock64() - t_start > watchdog_ticks) { if (g.tid == 0) __hip_atomic_store(finished + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
			// no CTU to decide: post-decision tasks of any picture (deblocking, SAO, entropy coding of CTUs whose neighbourhood is decided)
			int ran = 0;
			for (int i = 0; i < nseq; i++) {
				const int cand = (start + i) % nseq;
				const EncDev &dd = devs[cand];      // (by reference: an idle scan does not copy every picture's descriptor into private memory)
				ran += pool_post_drain(dd, g, dd.seq, dd.frame, lw, slow + blockIdx.x, finished);
				if (ran) { start = cand; break; }
			}
			if (!ran) {
				// nothing to decide, nothing to filter or code: back off - a worker that rescans at once keeps 64 x nseq atomic loads and the rows' counters of every
				// picture in flight on the L2 of an XCD whose other workers are deciding CTUs (rate control gates steps on the coder's progress: most of a
				// 2160p CBR batch's 1024 workers are idle at any time, and with a fixed 1 us pause the batch ran at 30 frames/s on 1024 workers, 44 on 512)
				idle_rounds = idle_rounds < 16 ? idle_rounds + 1 : 16;
				for (int i = 0; i < idle_rounds; i++) __builtin_amdgcn_s_sleep(127);
			} else {
				idle_rounds = 0;
				t_start = wall_clock64();
			}
			continue;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the step was opened with a release store after its predecessors' results)
		const EncDev d = devs[q];
		if (q != cached_q) {
			for (int i = g.tid; i < (int)(sizeof(Seq) / 4); i += 64) ((uint32_t *)lseq)[i] = ((const uint32_t *)d.seq)[i];
			for (int i = g.tid; i < (int)(sizeof(FrameCtx) / 4); i += 64) ((uint32_t *)lframe)[i] = ((const uint32_t *)d.frame)[i];
			cached_q = q;
			g.sync();
		}
		e.T = d.tables;
		e.ctus = d.ctus;
		const int W = lseq->wctu, H = lseq->hctu;
		// the k-th CTU of the step: the row of thread 0 first (it makes the scene-change check the others of the step wait for), then the rest top down
		int r_lo, r_hi;
		pool_step_rows(t, W, H, &r_lo, &r_hi);
		const int hrow = t / (2 * d.threads) * d.threads;
		const bool hvalid = hrow >= r_lo && hrow <= r_hi;
		int row;
		if (hvalid) row = k == 0 ? hrow : (r_lo + k - 1 < hrow ? r_lo + k - 1 : r_lo + k);
		else row = r_lo + k;
		pool_encode_ctu(d, e, g, lseq, lframe, lft, t, row, cached_rem, finished + 1);
		// close the step when this was its last CTU
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (g.tid == 0) {
			int *st = state + (size_t)q * POOL_STRIDE;
			const int dn = atomicAdd(&st[1 + POOL_MAX_STEPS + t], 1) + 1;
			if (dn == r_hi - r_lo + 1) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the other finishers' results happen before the step is declared closed)
				__hip_atomic_store(&st[0], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		pool_post_drain(d, g, lseq, lframe, lw, slow + blockIdx.x, finished);
		idle_rounds = 0;
		t_start = wall_clock64();
		start = (q + 1) % nseq;
	}
	int hseq[NHELP];
	for (int h = 0; h < NHELP; h++) hseq[h] = e.hseq[h];
	release_helpers(hseq);
}

// the true chains in raster order: threads 0..255 one unit column of the mode buffers each, thread 256 the intra counter
__global__ __launch_bounds__(320) void k_sched_scan(EncDev d)
{
	const int nctu = d.seq->nctu, k = threadIdx.x;
	if (k < NPART) {
		uint8_t st[2][NDEPTH];
		for (int comp = 0; comp < 2; comp++)
			for (int dd = 0; dd < NDEPTH; dd++) st[comp][dd] = d.chain_start[(comp * NDEPTH + dd) * NPART + k];
		for (int n = 0; n < nctu; n++) {
			uint8_t *t = d.truth + (size_t)n * MODE_STATE_BYTES;
			for (int comp = 0; comp < 2; comp++)
				for (int dd = 0; dd < NDEPTH; dd++) t[(comp * NDEPTH + dd) * NPART + k] = st[comp][dd];
			sched_chain_step(st, d.outtok + (size_t)n * MODE_STATE_BYTES, k);
		}
		for (int comp = 0; comp < 2; comp++)
			for (int dd = 0; dd < NDEPTH; dd++) d.chain_end[(comp * NDEPTH + dd) * NPART + k] = st[comp][dd];
	} else if (k == NPART) {
		uint32_t ib = 0;
		int cut = -1;
		const FrameCtx &f = *d.frame;
		for (int n = 0; n < nctu; n++) {
			d.intra_before[n] = ib;
			if (cut < 0 && f.slice_type == SLICE_P && scene_cut_fires(*d.seq, f, ib, (uint32_t)n * NPART)) cut = n;
			ib += d.ctus[n].intra_parts;
		}
		d.counters[0] = 0;
		d.counters[2] = cut;
	}
}

__global__ __launch_bounds__(64) void k_sched_check(EncDev d)
{
	const int n = blockIdx.x;
	WaveGrp g{(int)threadIdx.x};
	FrameCtx f = *d.frame;
	f.scene_cut_ctu = d.counters[2];
	const int uses_ratio = f.slice_type != SLICE_I;
	const int ok = sched_guesses_hold(g, d.ctus[n], f, d.truth + (size_t)n * MODE_STATE_BYTES, d.guess + (size_t)n * MODE_STATE_BYTES, d.intra_before[n], (uint32_t)n * NPART,
					  d.used_intra[n], d.used_parts[n], uses_ratio);
	if (g.tid == 0) {
		d.valid[n] = ok;
		if (!ok) atomicAdd(&d.counters[0], 1);
	}
}

// the frame has converged: tokens left in the CTUs' mode arrays become the values they stand for, the chain moves on
__global__ __launch_bounds__(64) void k_sched_finish(EncDev d)
{
	const int n = blockIdx.x, tid = threadIdx.x;
	CtuInfo &c = d.ctus[n];
	const uint8_t *t = d.truth + (size_t)n * MODE_STATE_BYTES;
	for (int i = tid; i < 2 * NPART; i += 64) {
		uint8_t &v = (&c.intra_mode[0][0])[i];
		if (v & MODE_TOKEN) v = t[((i / NPART) * NDEPTH + (v & 7)) * NPART + i % NPART];
	}
}

// host 8-bit planes -> int16 device planes (sse_copy_8_16 at frame entry, hmr_encoder_lib.c:295-305)
__global__ void k_widen_plane(const uint8_t *src, int w, int h, int16_t *dst, int stride)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x < w && y < h) dst[(size_t)y * stride + x] = src[(size_t)y * w + x];
}
__global__ void k_narrow_plane(const int16_t *src, int stride, int w, int h, uint8_t *dst)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x < w && y < h) dst[(size_t)y * w + x] = (uint8_t)src[(size_t)y * stride + x];
}

struct SrcSlot {
	int16_t *p[3];
};

__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void k_encode_ctus(EncDev d, int pass)
{
	if (!rows_enter((unsigned)LDS_BYTES)) return;
	int hseq[NHELP_MAX] = {0, 0, 0};
	encode_row(d, pass, (int)blockIdx.x, hseq);
	release_helpers(hseq);
}
// The post-decision stage of a picture whose CTU decisions are all final (the single-thread order: CTUs are re-encoded until the schedule's verification
// passes, so the stage cannot run along with them): one wavefront per workgroup, each runs whatever task is ready until the picture's last task is done.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_post_frame(EncDev d)
{
	extern __shared__ __align__(16) uint8_t lds[];
	WaveGrp g{(int)threadIdx.x};
	for (int r = (int)blockIdx.x * 64 + g.tid; r < d.seq->hctu; r += (int)gridDim.x * 64) d.post.rows[r].dec = d.seq->wctu;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	PostCtx x;
	x.seq = d.seq; x.f = d.frame; x.T = d.tables; x.geo.p = nullptr; x.ctus = d.ctus; x.coeff = d.coeff; x.pic = &d.post;
	PostScratch &sc = *(PostScratch *)in_fast_memory((uint8_t *)lds);
	const unsigned long long t_start = wall_clock64();
	while (!post_finished(*d.seq, d.post)) {
		if (!post_drain(g, x, sc)) {
			if (wall_clock64() - t_start > 3000000000ull) { if (g.tid == 0) d.post.errors[2] = 1; break; }      // 30 s: a peer is gone - report instead of hanging
			__builtin_amdgcn_s_sleep(16);
		}
	}
}

// page-locked host memory for what comes back from the device every frame (records and levels: 11 MB per 1080p frame; a pageable target is copied through a
// staging buffer at a fraction of the link's rate)
template <class T>
struct PinnedAlloc {
	typedef T value_type;
	PinnedAlloc() = default;
	template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
	T *allocate(size_t n)
	{
		void *p = nullptr;
		if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
		return (T *)p;
	}
	void deallocate(T *p, size_t) { (void)hipHostFree(p); }
	template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
	template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};

// The phase planes (124 MB for a 1080p picture) are needed from the start of a P frame's CTU stage to its end.  An encoder object borrows a set for that
// time from a per-process pool instead of owning one: hundreds of sequences can be resident on a GPU (an engine ring keeps an engine object of EVERY
// sequence on every GPU) while only the pictures of the running step need planes.
struct PlaneSet { uint8_t *y = nullptr, *c[2] = {nullptr, nullptr}; size_t bytes_y = 0, bytes_c = 0; int device = 0; };
struct PlanePool {
	std::mutex m;
	std::vector<PlaneSet> free_sets;
	int acquire(int device, size_t by, size_t bc, PlaneSet *out)
	{
		{
			std::lock_guard<std::mutex> lk(m);
			for (size_t i = 0; i < free_sets.size(); i++)
				if (free_sets[i].device == device && free_sets[i].bytes_y == by && free_sets[i].bytes_c == bc) {
					*out = free_sets[i];
					free_sets.erase(free_sets.begin() + (long)i);
					return HMR_GPU_OK;
				}
		}
		PlaneSet p;
		p.device = device; p.bytes_y = by; p.bytes_c = bc;
		if (hipMalloc((void **)&p.y, by) != hipSuccess || hipMalloc((void **)&p.c[0], bc) != hipSuccess || hipMalloc((void **)&p.c[1], bc) != hipSuccess) {
			(void)hipGetLastError();
			if (p.y) (void)hipFree(p.y);
			if (p.c[0]) (void)hipFree(p.c[0]);
			if (p.c[1]) (void)hipFree(p.c[1]);
			hmr_set_error("phase planes: out of device memory (%zu bytes per picture)", by + 2 * bc);
			return HMR_GPU_ERR_HIP;
		}
		*out = p;
		return HMR_GPU_OK;
	}
	void release(const PlaneSet &p)
	{
		std::lock_guard<std::mutex> lk(m);
		free_sets.push_back(p);
	}
	// the pool keeps the sets of destroyed encoders for the next ones (a batch of 180 sequences allocates 38 GB of them once); when the last encoder is gone they
	// go back to the device
	int live = 0;
	void encoder_created() { std::lock_guard<std::mutex> lk(m); live++; }
	void encoder_destroyed()
	{
		std::lock_guard<std::mutex> lk(m);
		if (--live > 0) return;
		for (PlaneSet &p : free_sets) {
			(void)hipSetDevice(p.device);
			(void)hipFree(p.y); (void)hipFree(p.c[0]); (void)hipFree(p.c[1]);
		}
		free_sets.clear();
	}
};
PlanePool g_plane_pool;
struct hmr_gpu_enc {
	hmr_gpu_ctx *ctx;
	HostCfg cfg;
	// evaluations on a stale prediction window (quirk Q12, hmr_gpu_enc_stale_predictions): of the last picture (-1: not counted - single-thread order), of all pictures
	long stale_last = -1, stale_total = 0;
	void note_stale_predictions(uint32_t n) { stale_last = (long)n; stale_total += (long)n; }
	Seq seq;
	HostState st;
	FrameCtx f;
	EncDev d;
	Seq *d_seq;
	FrameCtx *d_frame;
	Geo *d_geo;
	std::vector<Geo> geo;
	std::vector<SrcSlot> src;
	int16_t *d_pic[2][3], *d_pre[3], *d_rec[3];   // final pictures (current / reference), the deblocked picture, the reconstruction before the loop filters
	// post-decision stage (enc_post.h)
	PostRow *d_rows = nullptr;
	RowEnt *d_ent = nullptr;
	uint8_t *d_bs = nullptr;
	uint32_t *d_cumbits = nullptr;
	double *d_sao_tab = nullptr;         // [2 slice types: P, I][52][2]
	int *d_post_err = nullptr;
	// RD_FULL: the replay of the reference's coder objects (enc_rc.h), the states after every coded CTU of this frame and the one before, the table of the frame
	RdCtxSim rdsim;
	uint8_t *d_ctx_ring = nullptr, *d_rd_init = nullptr;      // (ring: frame f in slot f mod RD_RING; d_rd_init: [all-zero | the slices' initial states per slot] x RD_CTX_BYTES)
	int *d_rd_src = nullptr;
	uint8_t h_rd_init[(1 + RD_RING) * RD_CTX_BYTES] = {0};
	uint16_t *d_rc_need = nullptr;       // rate control: the CTUs of each row that are coded when a wavefront step starts (enc_rc.h rc_need_table)
	RcFrame *d_rc_dyn = nullptr;
	int row_cap = 0;
	std::vector<RowEnt> h_ent;
	std::vector<uint8_t, PinnedAlloc<uint8_t>> h_bs;
	PlaneSet chain_planes2;          // (the set of its frame before: the object that ends a chain holds the planes the chain's first frame predicts from while its own S tasks write the next)
	PlaneSet chain_planes;           // overlapping frames (hmr_gpu_enc_encode_chain): the phase planes of THIS object's final picture, written by the S tasks of its CTU launch
	PlaneSet planes;                 // phase planes of the reference picture (k_subpel.hip: 16 luma, 2 x 64 chroma), borrowed from g_plane_pool for the CTU stage of a P frame
	size_t src_elems[3], pic_elems[3];
	uint8_t *d_bytes;              // staging for 8-bit planes (one 4:2:0 picture)
	// raster unit arrays of the filters, SAO statistics and parameters
	int units_stride, units_rows;
	int16_t *d_mvx, *d_mvy;
	int8_t *d_ref;
	uint8_t *d_qp, *d_flags;
	uint8_t *d_public;                                 // the CTUs' side-info records, packed for the download
	// host side of the entropy stage
	std::vector<uint8_t, PinnedAlloc<uint8_t>> h_public;
	std::vector<int16_t, PinnedAlloc<int16_t>> h_coeff;
	std::vector<int32_t> h_stats, h_params;
	std::vector<double> h_lambdas;
	hipEvent_t ev_frame = nullptr, ev_ready = nullptr, ev_batch0 = nullptr, ev_batch1 = nullptr;   // (the batch launch has events of its own: frame_finish re-records the context's)   // start of the frame on the encoder's stream; its CTU stage may be launched
	int n_cus = 0;
	uint8_t *d_stage = nullptr, *h_stage = nullptr;      // batch: the side-info records and levels of all sequences, on the device and page-locked on the host
	size_t stage_bytes = 0;
	void *d_batch = nullptr;                             // hmr_gpu_enc_encode_batch (lead encoder): the sequences' EncDev records and first rows
	// pipelined batch (lead encoder): the step whose access units are still to be delivered
	bool pending = false, download_queued = false;
	std::vector<hmr_gpu_enc *> pend_encs;
	std::vector<size_t> pend_off;                        // where each sequence's sub-streams lie in the staging buffer
	std::vector<std::vector<uint32_t>> pend_rows;        // and the bytes of each of its rows
	size_t *h_offs = nullptr;                            // (page-locked: k_pack_streams reads it)
	size_t pend_total = 0;
	hipStream_t copy_stream = nullptr;
	uint32_t *d_gather = nullptr, *h_gather = nullptr;   // per picture of a launch: the frame's counters and the CTUs' distortions
	size_t gather_words = 0;
	// (every encoder of the batch)
	bool awaiting_delivery = false;                      // its last frame's access unit has not been coded yet
	FrameCtx f_pending;                                  // that frame's parameters (e->f moves on with the next set_frame)
	double acc_pending = 0;
	hipEvent_t ev_packed = nullptr;                      // its records and levels are in the staging buffer
	hipEvent_t ev_decided = nullptr;                     // (lead) the batch's SAO decisions are made
	FrameCtx *d_frames = nullptr, *h_frames = nullptr;   // (lead) the frame parameters of a batch's pictures, on the device and page-locked on the host
	EncDev *h_devs = nullptr;                            // (lead) their EncDev records, page-locked
	hipStream_t plane_stream[2] = {nullptr, nullptr};    // (lead) side streams for the chroma phase planes of a batch
	hipEvent_t ev_plane[3] = {nullptr, nullptr, nullptr};
	int *d_pool_state = nullptr;                         // k_encode_pool: per picture of the launch the open step and the steps' ticket / done counters, then the finished-pictures counter
	WorkSlow *d_pool_slow = nullptr;                     // the pool workers' transform / decoded windows
	int pool_workers = 0;
	EntropyState es;
	// engines (enc_host.h): the persistent state of each engine this object runs - engine k = frames k, k + E, ... - swapped into d at set_frame
	CtuInfo *d_ctus_eng[MAX_ENGINES] = {nullptr};
	uint8_t *d_rowstate_eng[MAX_ENGINES] = {nullptr};
	int *d_seen_eng[MAX_ENGINES] = {nullptr};
	hmr_gpu_enc *twin_of = nullptr;   // hmr_gpu_enc_create_engine_twin: the object whose persistent engine state this one shares
	int local_engines = 1, engine_index = -1;     // engine_index >= 0: this object is ONE engine of st.engines (the others live elsewhere, hmr_gpu_enc_create_engine)
	int cur, lockstep;
	float last_ms, last_total_ms;
	int last_passes, last_encodes;
};

namespace {
void release_planes(hmr_gpu_enc *e);
constexpr int REC_BYTES = 32 + 3 * 256 + 2 * 256 + 9 * 256 + 256 + 256 + 2048 + 2048 + 6144 * 2 + 6144 * 2 + 2 * 5 * 256;

int16_t *plane0(hmr_gpu_enc *e, int which, int comp)
{
	const Seq &s = e->seq;
	const int st = comp ? s.stride_c : s.stride_y, m = comp ? s.margin_c : s.margin_y;
	return e->d_pic[which][comp] + (size_t)m * st + m;
}

// device memory, cleared ON THE ENCODER'S STREAM: the stream is non-blocking, so a hipMemset (null stream) is not ordered against the work that follows on it -
// a picture slot allocated by hmr_gpu_enc_load_source could be cleared after the picture had been written into it (seen as an occasional different stream)
template <class T>
int dev_alloc(T **p, size_t n, hipStream_t st)
{
	HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
	HIP_TRY(hipMemsetAsync(*p, 0, n * sizeof(T), st));
	HIP_TRY(hipStreamSynchronize(st));   // (and a synchronous copy into the buffer, which runs on the null stream, must not overtake the clearing either)
	return HMR_GPU_OK;
}
#define DEV_ALLOC(p, n)                           \
	do {                                      \
		const int rc_ = dev_alloc(&(p), (n), e->ctx->stream); \
		if (rc_) return rc_;              \
	} while (0)

int load_planes(hmr_gpu_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int16_t *const dst[3], int stride_y, int stride_c)
{
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	const uint8_t *in[3] = {y, u, v};
	for (int c = 0; c < 3; c++) {
		const int w = c ? s.width / 2 : s.width, h = c ? s.height / 2 : s.height;
		HIP_TRY(hipMemcpyAsync(e->d_bytes, in[c], (size_t)w * h, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_widen_plane, dim3((w + 255) / 256, h), dim3(256), 0, st, e->d_bytes, w, h, dst[c], c ? stride_c : stride_y);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));   // d_bytes is reused by the next plane
	}
	return HMR_GPU_OK;
}


// what a frame's CTU stage needs on the stream before its first launch
// the reference picture, interpolated once at every sub-sample phase: what motion search and compensation read (k_subpel.hip)
int reference_planes(hmr_gpu_enc *e, hipStream_t st)
{
	const Seq &s = e->seq;
	if (e->f.slice_type == SLICE_I) return HMR_GPU_OK;
	return hmr_subpel_planes_on(st, e->d_pic[e->cur ^ 1][0], e->d_pic[e->cur ^ 1][1], e->d_pic[e->cur ^ 1][2], s.stride_y, s.height + 2 * s.margin_y, s.stride_c,
				    s.height / 2 + 2 * s.margin_c, e->planes.y, e->planes.c[0], e->planes.c[1]);
}
int ctu_stage_prepare(hmr_gpu_enc *e, bool planes_elsewhere = false)
{
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	if (!e->lockstep)   // (the frame-start state CTUs are re-encoded from in the single-thread order)
		HIP_TRY(hipMemcpyAsync(e->d.ctus_start, e->d.ctus, sizeof(CtuInfo) * s.nctu, hipMemcpyDeviceToDevice, st));
	if (!planes_elsewhere) {
		const int rc = reference_planes(e, st);
		if (rc) return rc;
	}
	{
		static const int zero_counters[3] = {0, 0, -1};
		HIP_TRY(hipMemcpyAsync(e->d.counters, zero_counters, sizeof zero_counters, hipMemcpyHostToDevice, st));
	}
	if (e->lockstep) {
		HIP_TRY(hipMemsetAsync(e->d.progress, 0, sizeof(int) * s.hctu, st));
		HIP_TRY(hipMemsetAsync(e->d.row0_checked, 0, sizeof(int), st));
		HIP_TRY(hipMemsetAsync(e->d.prefix, 0, sizeof(uint32_t) * s.hctu * (s.wctu + 1), st));
	}
	HIP_TRY(hipMemsetAsync(e->d_rows, 0, sizeof(PostRow) * s.hctu, st));
	HIP_TRY(hipMemsetAsync(e->d_post_err, 0, sizeof(int) * 4, st));
	return HMR_GPU_OK;
}
// row-per-thread schedule, after the launch: what the frame found
int lockstep_collect(hmr_gpu_enc *e)
{
	hipStream_t st = e->ctx->stream;
	int counters[3], aborted = 0;
	HIP_TRY(hipMemcpyAsync(counters, e->d.counters, sizeof counters, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&aborted, e->d_pool_state + 256 * POOL_STRIDE + 1, sizeof(int), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	e->last_encodes = counters[1];
	e->f.scene_cut_ctu = counters[2];
	e->last_passes = 1;
	release_planes(e);
	if (aborted) {
		hmr_set_error("k_encode_pool: the launch was abandoned by its watchdog (a worker found nothing to do for too long: HENC_WATCHDOG_S)");
		return HMR_GPU_ERR_HIP;
	}
	return HMR_GPU_OK;
}

// CTUs of a picture that can be decided at the same time: a wavefront step holds one CTU of every second column, so at most min(rows, (columns + 1) / 2); under rate
// control a step also waits for the entropy coder's progress (enc_rc.h), which about halves it.  Workers beyond what a launch can keep busy are not merely idle: every
// active worker runs slower the more of them share the chip's caches (a 2160p CBR batch of 32: 45 frames/s on 512 workers, 43 on 640, 30 on 1024)
static int pool_inflight(const Seq &s)
{
	const int by_step = s.hctu < (s.wctu + 1) / 2 ? s.hctu : (s.wctu + 1) / 2;
	return s.bitrate_mode != 0 ? (by_step + 1) / 2 : by_step;
}
// the row-per-thread schedule of n pictures (their EncDev records already at lead->d_batch) as ONE pool launch on `st`; rows_total: the sum of their pool_inflight()
int launch_pool(hmr_gpu_enc *lead, int n, int rows_total, bool needs_rd, hipStream_t st)
{
	if (!lead->n_cus) HIP_TRY(hipDeviceGetAttribute(&lead->n_cus, hipDeviceAttributeMultiprocessorCount, lead->ctx->device));
	// a worker's LDS: without RD_FULL pictures in the launch the RD arrays at its end are left out (a fourth worker then fits a CU)
	const size_t lds_default = needs_rd ? LDS_BYTES : LDS_BYTES - LDS_RD;
	const size_t lds_bytes = getenv("HENC_LDS_BYTES") && (size_t)atoi(getenv("HENC_LDS_BYTES")) > lds_default ? (size_t)atoi(getenv("HENC_LDS_BYTES")) : lds_default;   // (experiment: a larger request keeps a CU to fewer workers)
	const int cap = lead->n_cus * workers_per_cu(lds_bytes);                  // what the LDS lets be resident; more would only queue behind
	int workers = rows_total < cap ? rows_total : cap;                        // (a picture never has more CTUs in flight than rows)
	if (getenv("HENC_POOL_WORKERS") && atoi(getenv("HENC_POOL_WORKERS")) > 0 && atoi(getenv("HENC_POOL_WORKERS")) < workers) workers = atoi(getenv("HENC_POOL_WORKERS"));      // (experiment)
	if (!lead->d_pool_state) HIP_TRY(hipMalloc((void **)&lead->d_pool_state, sizeof(int) * (256 * POOL_STRIDE + 4)));
	if (lead->pool_workers < workers) {
		if (lead->d_pool_slow) (void)hipFree(lead->d_pool_slow);
		lead->d_pool_slow = nullptr;
		lead->pool_workers = 0;
		HIP_TRY(hipMalloc((void **)&lead->d_pool_slow, sizeof(WorkSlow) * workers));
		HIP_TRY(hipMemsetAsync(lead->d_pool_slow, 0, sizeof(WorkSlow) * workers, st));
		lead->pool_workers = workers;
	}
	HIP_TRY(hipMemsetAsync(lead->d_pool_state, 0, sizeof(int) * (256 * POOL_STRIDE + 4), st));
	// the watchdog (100 MHz ticks): a launch is a second or two of work; a worker that finds nothing to do for this long gives up for everybody
	double watchdog_s = getenv("HENC_WATCHDOG_S") ? atof(getenv("HENC_WATCHDOG_S")) : 120.0;
	if (!(watchdog_s > 0)) watchdog_s = 120.0;
	const unsigned long long watchdog = (unsigned long long)(watchdog_s * 1e8);
	hipLaunchKernelGGL(k_encode_pool, dim3(workers), dim3(ENC_THREADS), lds_bytes, st, (const EncDev *)lead->d_batch, n, lead->d_pool_state, lead->d_pool_state + 256 * POOL_STRIDE,
			   lead->d_pool_slow, watchdog, (unsigned)lds_default);
	const hipError_t launched = hipGetLastError();
	if (launched != hipSuccess) {
		hmr_set_error("k_encode_pool: %s", hipGetErrorString(launched));
		return HMR_GPU_ERR_HIP;
	}
	return HMR_GPU_OK;
}

// the CTU decisions of the frame set up in e->f / e->d_frame: passes until the check finds nothing wrong
int run_ctu_passes(hmr_gpu_enc *e)
{
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	int rc = ctu_stage_prepare(e);
	if (rc) return rc;
	HIP_TRY(hipEventRecord(e->ctx->ev0, st));
	if (e->lockstep) {
		// wfpp_num_threads > 1: the synchronous wavefront, one launch, nothing to verify - the picture's CTUs as a pool of tasks (k_encode_pool)
		if (!e->d_batch) HIP_TRY(hipMalloc((void **)&e->d_batch, 256 * sizeof(EncDev)));
		HIP_TRY(hipMemcpyAsync(e->d_batch, &e->d, sizeof(EncDev), hipMemcpyHostToDevice, st));
		if ((rc = launch_pool(e, 1, s.hctu, s.rd_mode == RDM_FULL, st))) return rc;      // (one picture: a worker per row, each on a CU of its own)
		HIP_TRY(hipEventRecord(e->ctx->ev1, st));
		if ((rc = lockstep_collect(e))) return rc;      // (waits for the launch)
		HIP_TRY(hipEventElapsedTime(&e->last_ms, e->ctx->ev0, e->ctx->ev1));
		return HMR_GPU_OK;
	}
	// wfpp_num_threads = 1: the single thread's order.  Row workers wait for the row above (progress[]), so every workgroup of the launch has to be resident
	// or the waiting ones spin for ever: a cooperative launch makes the runtime guarantee that - it fails at launch time when the grid does not fit (another
	// process on the GPU, a partition mode) instead of hanging.
	int pass = 0;
	for (;; pass++) {
		HIP_TRY(hipMemsetAsync(e->d.progress, 0, sizeof(int) * s.hctu, st));
		{
			EncDev dd = e->d;
			int pp = pass;
			void *args[] = {&dd, &pp};
			const hipError_t launched = hipLaunchCooperativeKernel((const void *)k_encode_ctus, dim3(s.hctu), dim3(ENC_THREADS), args, (unsigned)LDS_BYTES, st);
			if (launched != hipSuccess) {
				hmr_set_error("k_encode_ctus (%d row workers, cooperative): %s", s.hctu, hipGetErrorString(launched));
				return HMR_GPU_ERR_HIP;
			}
		}
		hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(320), 0, st, e->d);
		hipLaunchKernelGGL(k_sched_check, dim3(s.nctu), dim3(64), 0, st, e->d);
		HIP_TRY(hipGetLastError());
		int counters[3];
		HIP_TRY(hipMemcpyAsync(counters, e->d.counters, sizeof counters, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		e->last_encodes = counters[1];
		e->f.scene_cut_ctu = counters[2];
		if (counters[0] == 0) break;
		if (pass > s.nctu + 2) {
			hmr_set_error("hmr_gpu_enc: the CTU schedule did not converge");
			return HMR_GPU_ERR_HIP;
		}
	}
	hipLaunchKernelGGL(k_sched_finish, dim3(s.nctu), dim3(64), 0, st, e->d);
	HIP_TRY(hipMemcpyAsync(e->d.chain_start, e->d.chain_end, MODE_STATE_BYTES, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipMemcpyAsync(e->d_frame, &e->f, sizeof(FrameCtx), hipMemcpyHostToDevice, st));      // (scene_cut_ctu as the passes found it)
	HIP_TRY(hipFuncSetAttribute((const void *)k_post_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PostScratch)));
	hipLaunchKernelGGL(k_post_frame, dim3(s.hctu < 32 ? s.hctu : 32), dim3(64), sizeof(PostScratch), st, e->d);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e->ctx->ev1, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipEventElapsedTime(&e->last_ms, e->ctx->ev0, e->ctx->ev1));
	e->last_passes = pass + 1;
	release_planes(e);
	return HMR_GPU_OK;
}

int set_frame(hmr_gpu_enc *e, int slot, int image_type, double avg_dist, bool upload = true, bool chain = false)
{
	const Seq &s = e->seq;
	e->cur ^= 1;
	{
		const int k = e->local_engines > 1 ? e->st.num_encoded_frames % e->local_engines : 0;
		e->d.ctus = e->d_ctus_eng[k];
		e->d.rowstate = e->d_rowstate_eng[k];
		e->d.thread_seen = e->d_seen_eng[k];
	}
	begin_frame(s, e->st, image_type, e->f);
	if (avg_dist >= 0) e->f.avg_dist = avg_dist;
	e->f.lockstep = e->lockstep;
	for (int c = 0; c < 3; c++) {
		e->f.src[c] = e->src[slot].p[c];
		e->f.ref[c] = plane0(e, e->cur ^ 1, c);
		e->f.rec[c] = e->d_rec[c] + (size_t)(c ? s.margin_c : s.margin_y) * (c ? s.stride_c : s.stride_y) + (c ? s.margin_c : s.margin_y);
		e->d.post.fin[c] = plane0(e, e->cur, c);
	}
	e->d.post.sao_lambda = e->d_sao_tab + (e->f.slice_type == SLICE_I ? 104 : 0);
	e->d.post.ctx_after = nullptr;
	if (s.rd_mode == RDM_FULL) {
		// which coder states each CTU's estimates copy this frame (enc_rc.h RdCtxSim), as a table for the CTU kernel
		std::vector<RdCtxVersion> src;
		e->rdsim.frame(e->f.num_encoded_frames, src);
		std::vector<int> codes(s.nctu);
		for (int n = 0; n < s.nctu; n++) {
			const RdCtxVersion v = src[n];
			if (v.frame > e->f.num_encoded_frames || (v.frame >= 0 && v.frame <= e->f.num_encoded_frames - RD_RING)) {
				hmr_set_error("RD_FULL: CTU %d copies coder states of frame %d in frame %d (the states of %d frames are kept)", n, v.frame, e->f.num_encoded_frames, RD_RING);
				return HMR_GPU_ERR_ARG;
			}
			const int slot = v.frame < 0 ? 0 : v.frame % RD_RING;
			codes[n] = v.frame < 0 ? 0 : (v.k == 0 ? (1 << 28) | (slot << 24) : (3 << 28) | (slot << 24) | (v.row * s.wctu + v.k - 1));
		}
		const int slot_now = e->f.num_encoded_frames % RD_RING;
		for (int i = 0; i < CTX_TOTAL; i++) e->h_rd_init[(1 + slot_now) * RD_CTX_BYTES + i] = Cabac::init_state(e->f.slice_type, e->f.qp, i);
		HIP_TRY(hipMemcpyAsync(e->d_rd_src, codes.data(), sizeof(int) * s.nctu, hipMemcpyHostToDevice, e->ctx->stream));      // (pageable sources: copied before the call returns)
		HIP_TRY(hipMemcpyAsync(e->d_rd_init, e->h_rd_init, sizeof e->h_rd_init, hipMemcpyHostToDevice, e->ctx->stream));
		e->d.post.ctx_after = e->d_ctx_ring + (size_t)slot_now * s.nctu * RD_CTX_BYTES;
		e->d.rd_src = e->d_rd_src;
		e->d.rd_init = e->d_rd_init;
		e->d.rd_ring = e->d_ctx_ring;
	}
	e->d.dep = -1;
	e->d.dep_full = 0;
	e->d.after = -1;
	e->d.next_frame = nullptr;
	e->d.post.planes[0] = e->d.post.planes[1] = e->d.post.planes[2] = nullptr;
	if (e->f.slice_type != SLICE_I && !chain) {
		if (!e->planes.y) {
			const int rc = g_plane_pool.acquire(e->ctx->device, (size_t)16 * s.plane_elems_y, (size_t)64 * s.plane_elems_c, &e->planes);
			if (rc) return rc;
		}
		e->f.sub_y = e->planes.y + (size_t)s.margin_y * 16 * s.stride_y + s.margin_y;       // (row-interleaved: a row of the picture is 16 / 64 rows of phases)
		e->f.sub_c[0] = e->planes.c[0] + (size_t)s.margin_c * 64 * s.stride_c + s.margin_c;
		e->f.sub_c[1] = e->planes.c[1] + (size_t)s.margin_c * 64 * s.stride_c + s.margin_c;
	}
	if (upload) HIP_TRY(hipMemcpyAsync(e->d_frame, &e->f, sizeof(FrameCtx), hipMemcpyHostToDevice, e->ctx->stream));      // (a batch uploads its frames' parameters in one copy)
	return HMR_GPU_OK;
}

// the CTU stage of the frame is over (its launch has been waited for): the planes go back to the pool
void release_planes(hmr_gpu_enc *e)
{
	if (e->planes.y) g_plane_pool.release(e->planes);
	e->planes = PlaneSet();
}

// the three planes of a picture to their copies in one launch (three device-to-device copies are three copy kernels on the stream)
__global__ __launch_bounds__(256) void k_copy_planes(const int16_t *s0, const int16_t *s1, const int16_t *s2, int16_t *d0, int16_t *d1, int16_t *d2, size_t n0, size_t n12)
{
	const int c = (int)blockIdx.y;
	const uint4 *s = (const uint4 *)(c == 0 ? s0 : (c == 1 ? s1 : s2));
	uint4 *d = (uint4 *)(c == 0 ? d0 : (c == 1 ? d1 : d2));
	const size_t n = (c == 0 ? n0 : n12) / 8;      // eight samples per thread and step (the planes' sizes are multiples of eight)
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// the side-info records lie 68 KB apart inside the CTU records: packed on the device, then one linear copy to the host
__global__ void k_pack_public(const CtuInfo *ctus, uint32_t *out)
{
	const uint32_t *src = (const uint32_t *)(const CtuPublic *)(ctus + blockIdx.x);
	uint32_t *dst = out + (size_t)blockIdx.x * (sizeof(CtuPublic) / 4);
	for (int i = threadIdx.x; i < (int)(sizeof(CtuPublic) / 4); i += blockDim.x) dst[i] = src[i];
}
int download_public(hmr_gpu_enc *e)
{
	const Seq &s = e->seq;
	hipLaunchKernelGGL(k_pack_public, dim3(s.nctu), dim3(256), 0, e->ctx->stream, e->d.ctus, (uint32_t *)e->d_public);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_public.data(), e->d_public, sizeof(CtuPublic) * s.nctu, hipMemcpyDeviceToHost, e->ctx->stream));
	return HMR_GPU_OK;
}
}  // namespace

extern "C" int hmr_gpu_enc_record_bytes(void) { return REC_BYTES; }

static int enc_create(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, int engine_index, hmr_gpu_enc **out)
{
	if (!ctx || !cfg || !out) return HMR_GPU_ERR_ARG;
	static_assert(sizeof(hmr_gpu_enc_cfg) == sizeof(HostCfg), "configuration layouts must match");
	static_assert(offsetof(CtuInfo, n_spec_reads) % 4 == 0 && sizeof(Node) % 4 == 0 && sizeof(Geo) % 2 == 0 && sizeof(Seq) % 4 == 0 && sizeof(FrameCtx) % 4 == 0 && sizeof(CtuPublic) % 4 == 0 && MODE_STATE_BYTES % 4 == 0, "word copies");
	hmr_gpu_enc *e = new hmr_gpu_enc();
	e->ctx = ctx;
	memcpy(&e->cfg, cfg, sizeof(HostCfg));
	const char *why = "";
	if (!make_seq(e->cfg, e->seq, &why)) {
		hmr_set_error("hmr_gpu_enc_create: configuration outside the built rows: %s", why);
		delete e;
		return HMR_GPU_ERR_ARG;
	}
	e->st.engines = clampi(e->cfg.num_enc_engines, 1, MAX_ENGINES);
	e->engine_index = engine_index;
	e->local_engines = engine_index >= 0 ? 1 : e->st.engines;
	if (e->st.engines > 1 && e->cfg.wfpp_num_threads < 2) {
		hmr_set_error("hmr_gpu_enc_create: configuration outside the built rows: num_enc_engines > 1 needs the row-per-thread schedule (wfpp_num_threads > 1)");
		delete e;
		return HMR_GPU_ERR_ARG;
	}
	if (engine_index >= e->st.engines) { delete e; return HMR_GPU_ERR_ARG; }
	if (e->cfg.bitrate_mode != 0 && e->cfg.wfpp_num_threads < 2) {
		// (the single-thread order re-encodes CTUs until its verification passes; the bits the rate control reads come from entropy coding behind FINAL decisions)
		hmr_set_error("hmr_gpu_enc_create: configuration outside the built rows: rate control needs the row-per-thread schedule (wfpp_num_threads > 1)");
		delete e;
		return HMR_GPU_ERR_ARG;
	}
	if (e->cfg.bitrate_mode != 0) host_rc_init(e->cfg, e->seq, e->st);
	g_plane_pool.encoder_created();      // (hmr_gpu_enc_destroy - also the guard's - takes it back)
	struct Guard {               // a failure further down (HIP_TRY / DEV_ALLOC return) frees what has been allocated so far
		hmr_gpu_enc *e;
		bool ok = false;
		~Guard() { if (!ok) hmr_gpu_enc_destroy(e); }
	} guard{e};
	e->seq.wide_min_n = 0;
	const Seq &s = e->seq;
	HIP_TRY(hipSetDevice(ctx->device));
	e->geo.resize(NNODES);
	make_geo(e->geo.data());
	DEV_ALLOC(e->d_seq, 1);
	HIP_TRY(hipMemcpy(e->d_seq, &s, sizeof(Seq), hipMemcpyHostToDevice));
	DEV_ALLOC(e->d_frame, 1);
	DEV_ALLOC(e->d_geo, NNODES);
	HIP_TRY(hipMemcpy(e->d_geo, e->geo.data(), sizeof(Geo) * NNODES, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(henc_geo_table), e->geo.data(), sizeof(Geo) * NNODES));   // (the same tree for every encoder: 64 x 64 CTUs, prediction depth 4)
	DEV_ALLOC(e->d.ctus, s.nctu);
	if (e->cfg.wfpp_num_threads <= 1) DEV_ALLOC(e->d.ctus_start, s.nctu);   // (the frame-start state CTUs are re-encoded from: single-thread order only)
	{
		std::vector<CtuInfo> init(s.nctu);
		memset((void *)init.data(), 0, sizeof(CtuInfo) * s.nctu);
		for (auto &c : init) memset(c.mv_ref_idx, -1, sizeof c.mv_ref_idx);
		HIP_TRY(hipMemcpy(e->d.ctus, init.data(), sizeof(CtuInfo) * s.nctu, hipMemcpyHostToDevice));
		e->d_ctus_eng[0] = e->d.ctus;
		for (int k = 1; k < e->local_engines; k++) {
			DEV_ALLOC(e->d_ctus_eng[k], s.nctu);
			HIP_TRY(hipMemcpy(e->d_ctus_eng[k], init.data(), sizeof(CtuInfo) * s.nctu, hipMemcpyHostToDevice));
		}
	}
	if (e->cfg.wfpp_num_threads <= 1) DEV_ALLOC(e->d.work_slow, s.hctu);      // (the pool has its workers' windows: lead->d_pool_slow)
	HIP_TRY(hipFuncSetAttribute((const void *)k_encode_ctus, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
	HIP_TRY(hipFuncSetAttribute((const void *)k_encode_pool, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipEventCreate(&e->ev_frame));
	HIP_TRY(hipEventCreate(&e->ev_ready));
	HIP_TRY(hipEventCreate(&e->ev_batch0));
	HIP_TRY(hipEventCreate(&e->ev_batch1));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_packed, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_decided, hipEventDisableTiming));
	DEV_ALLOC(e->d.coeff, (size_t)6144 * s.nctu);
	DEV_ALLOC(e->d.progress, s.hctu);
	DEV_ALLOC(e->d.prefix, (size_t)s.hctu * (s.wctu + 1));
	DEV_ALLOC(e->d.prof, (size_t)s.hctu * PF_COUNT + (size_t)s.nctu * 4);   // + per CTU: 100 MHz timestamps of wait start, encode start, first use of the intra share, end
	DEV_ALLOC(e->d.guess, (size_t)s.nctu * MODE_STATE_BYTES);
	DEV_ALLOC(e->d.truth, (size_t)s.nctu * MODE_STATE_BYTES);
	DEV_ALLOC(e->d.outtok, (size_t)s.nctu * MODE_STATE_BYTES);
	DEV_ALLOC(e->d.chain_start, MODE_STATE_BYTES);
	DEV_ALLOC(e->d.chain_end, MODE_STATE_BYTES);
	DEV_ALLOC(e->d.valid, s.nctu);
	DEV_ALLOC(e->d.dirty, s.nctu);
	DEV_ALLOC(e->d.hash, s.nctu);
	DEV_ALLOC(e->d.intra_before, s.nctu);
	DEV_ALLOC(e->d.used_intra, s.nctu);
	DEV_ALLOC(e->d.used_parts, s.nctu);
	DEV_ALLOC(e->d.counters, 4);
	DEV_ALLOC(e->d.rowstate, (size_t)s.hctu * ROW_STATE_BYTES);
	DEV_ALLOC(e->d.thread_seen, 64);
	e->d_rowstate_eng[0] = e->d.rowstate;
	e->d_seen_eng[0] = e->d.thread_seen;
	for (int k = 1; k < e->local_engines; k++) {
		DEV_ALLOC(e->d_rowstate_eng[k], (size_t)s.hctu * ROW_STATE_BYTES);
		DEV_ALLOC(e->d_seen_eng[k], 64);
	}
	DEV_ALLOC(e->d.row0_checked, 1);
	for (int c = 0; c < 3; c++) {
		e->src_elems[c] = (size_t)(c ? s.src_stride_c : s.src_stride_y) * (c ? s.height / 2 : s.height);
		e->pic_elems[c] = (size_t)(c ? s.stride_c : s.stride_y) * ((c ? s.height / 2 : s.height) + 2 * (c ? s.margin_c : s.margin_y));
		for (int k = 0; k < 2; k++) DEV_ALLOC(e->d_pic[k][c], e->pic_elems[c]);
		DEV_ALLOC(e->d_pre[c], e->pic_elems[c]);
		DEV_ALLOC(e->d_rec[c], e->pic_elems[c]);
	}
	{
		// the post-decision stage: row progress, the rows' CABAC coders and sub-streams (8 KB per CTU: forty times what QP 32 needs; a sub-stream that runs
		// out of room is an error return, not a truncated stream), the SAO Lagrange multipliers of both slice types by QP
		e->row_cap = s.wctu * 8192;
		DEV_ALLOC(e->d_rows, s.hctu);
		DEV_ALLOC(e->d_ent, s.hctu);
		DEV_ALLOC(e->d_bs, (size_t)e->row_cap * s.hctu);
		DEV_ALLOC(e->d_cumbits, s.nctu);
		DEV_ALLOC(e->d_sao_tab, 2 * 52 * 2);
		DEV_ALLOC(e->d_post_err, 4 + 2 * 16 + 4);
		double tab[2][104];
		sao_lambda_table(s, SLICE_P, tab[0]);
		sao_lambda_table(s, SLICE_I, tab[1]);
		HIP_TRY(hipMemcpy(e->d_sao_tab, tab, sizeof tab, hipMemcpyHostToDevice));
		e->h_ent.resize(s.hctu);
		e->h_bs.resize((size_t)e->row_cap * s.hctu);
		DEV_ALLOC(e->d_rc_dyn, 1);
		if (s.rd_mode == RDM_FULL) {
			e->rdsim.init(e->cfg.wfpp_num_threads, s.wctu, s.hctu, s.sao);
			DEV_ALLOC(e->d_ctx_ring, (size_t)RD_RING * s.nctu * RD_CTX_BYTES);
			HIP_TRY(hipMemset(e->d_ctx_ring, 0, (size_t)RD_RING * s.nctu * RD_CTX_BYTES));
			DEV_ALLOC(e->d_rd_init, (1 + RD_RING) * RD_CTX_BYTES);
			DEV_ALLOC(e->d_rd_src, s.nctu);
		}
		if (e->cfg.bitrate_mode != 0 || s.rd_mode == RDM_FULL) {      // (RD_FULL: a step's estimates copy coder states the steps before must have produced)
			std::vector<uint16_t> need;
			if (!rc_need_table(s.wctu, s.hctu, s.sao, true, need)) {
				hmr_set_error("hmr_gpu_enc_create: rate control: the reference's entropy-coding lag is not row-monotone on a %d x %d CTU grid", s.wctu, s.hctu);
				return HMR_GPU_ERR_ARG;
			}
			DEV_ALLOC(e->d_rc_need, need.size());
			HIP_TRY(hipMemcpy(e->d_rc_need, need.data(), need.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		}
	}
	DEV_ALLOC(e->d_bytes, (size_t)s.width * s.height * 3 / 2);
	e->units_stride = s.wctu * 16;
	e->units_rows = s.hctu * 16;
	const size_t nu = (size_t)e->units_stride * e->units_rows;
	DEV_ALLOC(e->d_mvx, nu); DEV_ALLOC(e->d_mvy, nu); DEV_ALLOC(e->d_ref, nu); DEV_ALLOC(e->d_qp, nu);
	DEV_ALLOC(e->d_flags, nu);
	DEV_ALLOC(e->d_public, sizeof(CtuPublic) * s.nctu);
	e->h_public.resize(sizeof(CtuPublic) * s.nctu);
	e->h_coeff.resize((size_t)6144 * s.nctu);
	e->h_stats.resize((size_t)s.nctu * 3 * 5 * 2 * 32);
	e->h_params.resize((size_t)s.nctu * 3 * 34);
	e->d.seq = e->d_seq;
	e->d.frame = e->d_frame;
	e->d.tables = ctx->tables;
	e->d.geo = e->d_geo;
	e->d.rc_dyn = e->d_rc_dyn;
	{
		const size_t oy = (size_t)s.margin_y * s.stride_y + s.margin_y, oc = (size_t)s.margin_c * s.stride_c + s.margin_c;
		PostPic &P = e->d.post;
		memset(&P, 0, sizeof P);
		P.dbk[0] = e->d_pre[0] + oy; P.dbk[1] = e->d_pre[1] + oc; P.dbk[2] = e->d_pre[2] + oc;
		P.units_stride = e->units_stride;
		P.mvx = e->d_mvx; P.mvy = e->d_mvy; P.ref = e->d_ref; P.uqp = e->d_qp; P.flags = e->d_flags;
		P.rows = e->d_rows; P.ent = e->d_ent; P.bs = e->d_bs; P.row_cap = e->row_cap; P.cumbits = e->d_cumbits;
		P.sao_lambda = e->d_sao_tab; P.errors = e->d_post_err; P.rc_need = e->d_rc_need;
		P.prof = (unsigned long long *)(e->d_post_err + 4);      // (profiling build)
		e->d.fin = (double *)(e->d_post_err + 4 + 2 * 16);
	}
	e->cur = 0;
	e->lockstep = e->cfg.wfpp_num_threads > 1;
	e->d.threads = e->cfg.wfpp_num_threads > 1 ? e->cfg.wfpp_num_threads : 1;
	e->last_ms = e->last_total_ms = 0;
	e->last_passes = e->last_encodes = 0;
	HIP_TRY(hipStreamSynchronize(ctx->stream));   // (the buffers are cleared by now)
	guard.ok = true;
	*out = e;
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_enc_create(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, hmr_gpu_enc **out) { return enc_create(ctx, cfg, -1, out); }
extern "C" int hmr_gpu_enc_create_engine(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, int engine_index, hmr_gpu_enc **out)
{
	if (engine_index < 0) return HMR_GPU_ERR_ARG;
	return enc_create(ctx, cfg, engine_index, out);
}
// A second object for the SAME engine: it shares the engine's persistent state (the CTU records and the WPP threads' mode buffers, which a frame continues from the
// engine's previous frame) with `of` and has pictures, filter state and sub-stream buffers of its own - so that hmr_gpu_enc_encode_chain can hold the engine's next frame
// in the same launch (it starts when the one before it is finished, as the reference's engine does).  Destroy the twins before the object they were made from.
extern "C" int hmr_gpu_enc_create_engine_twin(hmr_gpu_ctx *ctx, hmr_gpu_enc *of, hmr_gpu_enc **out)
{
	if (!ctx || !of || !out || of->engine_index < 0 || of->twin_of) return HMR_GPU_ERR_ARG;
	hmr_gpu_enc *t = nullptr;
	const int rc = enc_create(ctx, (const hmr_gpu_enc_cfg *)&of->cfg, of->engine_index, &t);
	if (rc) return rc;
	(void)hipFree(t->d_ctus_eng[0]);
	(void)hipFree(t->d_rowstate_eng[0]);
	(void)hipFree(t->d_seen_eng[0]);
	t->d_ctus_eng[0] = of->d_ctus_eng[0];
	t->d_rowstate_eng[0] = of->d_rowstate_eng[0];
	t->d_seen_eng[0] = of->d_seen_eng[0];
	t->d.ctus = t->d_ctus_eng[0]; t->d.rowstate = t->d_rowstate_eng[0]; t->d.thread_seen = t->d_seen_eng[0];
	t->twin_of = of;
	*out = t;
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_enc_state_bytes(void) { return (int)sizeof(HostState); }
extern "C" long hmr_gpu_enc_reference_elems(hmr_gpu_enc *e, int comp) { return e && comp >= 0 && comp < 3 ? (long)e->pic_elems[comp] : -1; }

// The picture the next frame predicts from (padded planes, from the first element of their allocation) and the frame-to-frame scalars: what one engine hands
// to the next (encoder_engine_thread keeps both in the shared hvenc_enc_t; with an engine per GPU they travel).  Device buffers of hmr_gpu_enc_reference_elems.
extern "C" int hmr_gpu_enc_export_reference(hmr_gpu_enc *e, int16_t *dy, int16_t *du, int16_t *dv, void *state)
{
	if (!e || !dy || !du || !dv || !state) return HMR_GPU_ERR_ARG;
	hipStream_t st = e->ctx->stream;
	HIP_TRY(hipSetDevice(e->ctx->device));
	int16_t *dst[3] = {dy, du, dv};
	for (int c = 0; c < 3; c++) HIP_TRY(hipMemcpyAsync(dst[c], e->d_pic[e->cur][c], e->pic_elems[c] * 2, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	memcpy(state, &e->st, sizeof(HostState));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_enc_import_reference(hmr_gpu_enc *e, const int16_t *dy, const int16_t *du, const int16_t *dv, const void *state)
{
	if (!e || !dy || !du || !dv || !state) return HMR_GPU_ERR_ARG;
	HostState in;
	memcpy(&in, state, sizeof in);
	if (in.engines != e->st.engines || (e->engine_index >= 0 && in.num_encoded_frames % e->st.engines != e->engine_index)) {
		hmr_set_error("hmr_gpu_enc_import_reference: state of frame %d does not precede a frame of engine %d of %d", in.num_encoded_frames - 1, e->engine_index, e->st.engines);
		return HMR_GPU_ERR_ARG;
	}
	hipStream_t st = e->ctx->stream;
	HIP_TRY(hipSetDevice(e->ctx->device));
	const int16_t *src[3] = {dy, du, dv};
	for (int c = 0; c < 3; c++) HIP_TRY(hipMemcpyAsync(e->d_pic[e->cur][c], src[c], e->pic_elems[c] * 2, hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	e->st = in;
	return HMR_GPU_OK;
}

// The same hand-over for the engines of a step at once, with the picture as it travels between GPUs: 8-bit samples without margins (width x height luma, then the
// two width / 2 x height / 2 chroma planes: hmr_gpu_enc_reference_bytes), picture i at dev_rows + i * pitch; the importer widens it and pads the margins
// (reference_picture_border_padding_ctu: the margins are a function of the picture).  states: n x hmr_gpu_enc_state_bytes() bytes of host memory.
extern "C" long hmr_gpu_enc_reference_bytes(hmr_gpu_enc *e) { return e ? (long)e->seq.width * e->seq.height * 3 / 2 : -1; }
extern "C" int hmr_gpu_enc_export_references8(hmr_gpu_enc **encs, int n, uint8_t *dev_rows, long pitch, void *states)
{
	if (!encs || n <= 0 || !dev_rows || !states) return HMR_GPU_ERR_ARG;
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		if (!e || pitch < hmr_gpu_enc_reference_bytes(e)) return HMR_GPU_ERR_ARG;
		const Seq &s = e->seq;
		hipStream_t st = e->ctx->stream;
		HIP_TRY(hipSetDevice(e->ctx->device));
		uint8_t *o = dev_rows + (size_t)i * pitch;
		for (int c = 0; c < 3; c++) {
			const int w = c ? s.width / 2 : s.width, h = c ? s.height / 2 : s.height;
			hipLaunchKernelGGL(k_narrow_plane, dim3((w + 255) / 256, h), dim3(256), 0, st, plane0(e, e->cur, c), c ? s.stride_c : s.stride_y, w, h, o);
			o += (size_t)w * h;
		}
		HIP_TRY(hipGetLastError());
		memcpy((uint8_t *)states + (size_t)i * sizeof(HostState), &e->st, sizeof(HostState));
	}
	for (int i = 0; i < n; i++) HIP_TRY(hipStreamSynchronize(encs[i]->ctx->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_enc_import_references8(hmr_gpu_enc **encs, int n, const uint8_t *dev_rows, long pitch, const void *states)
{
	if (!encs || n <= 0 || !dev_rows || !states) return HMR_GPU_ERR_ARG;
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		if (!e || pitch < hmr_gpu_enc_reference_bytes(e)) return HMR_GPU_ERR_ARG;
		HostState in;
		memcpy(&in, (const uint8_t *)states + (size_t)i * sizeof(HostState), sizeof in);
		if (in.engines != e->st.engines || (e->engine_index >= 0 && in.num_encoded_frames % e->st.engines != e->engine_index)) {
			hmr_set_error("hmr_gpu_enc_import_references8: engine %d: state of frame %d does not precede a frame of engine %d of %d", i, in.num_encoded_frames - 1, e->engine_index, e->st.engines);
			return HMR_GPU_ERR_ARG;
		}
		const Seq &s = e->seq;
		hipStream_t st = e->ctx->stream;
		HIP_TRY(hipSetDevice(e->ctx->device));
		const uint8_t *o = dev_rows + (size_t)i * pitch;
		for (int c = 0; c < 3; c++) {
			const int w = c ? s.width / 2 : s.width, h = c ? s.height / 2 : s.height;
			hipLaunchKernelGGL(k_widen_plane, dim3((w + 255) / 256, h), dim3(256), 0, st, o, w, h, plane0(e, e->cur, c), c ? s.stride_c : s.stride_y);
			o += (size_t)w * h;
		}
		HIP_TRY(hipGetLastError());
		hmr_gpu_frame fr = {s.width, s.height, plane0(e, e->cur, 0), plane0(e, e->cur, 1), plane0(e, e->cur, 2), s.stride_y, s.stride_c};
		const int rc = hmr_gpu_pad_frame(e->ctx, &fr, s.margin_y, s.margin_y);
		if (rc) return rc;
		e->st = in;
	}
	for (int i = 0; i < n; i++) HIP_TRY(hipStreamSynchronize(encs[i]->ctx->stream));
	return HMR_GPU_OK;
}

extern "C" void hmr_gpu_enc_destroy(hmr_gpu_enc *e)
{
	if (!e) return;
	(void)hipSetDevice(e->ctx->device);
	(void)hipStreamSynchronize(e->ctx->stream);
	release_planes(e);
	for (PlaneSet *ps : {&e->chain_planes, &e->chain_planes2})
		if (ps->y) { (void)hipFree(ps->y); (void)hipFree(ps->c[0]); (void)hipFree(ps->c[1]); }
	e->d.ctus = e->d_ctus_eng[0]; e->d.rowstate = e->d_rowstate_eng[0]; e->d.thread_seen = e->d_seen_eng[0];
	if (e->twin_of) e->d.ctus = nullptr, e->d.rowstate = nullptr, e->d.thread_seen = nullptr;      // (they belong to the object the twin was made from)
	for (int k = 1; k < MAX_ENGINES; k++) {
		if (e->d_ctus_eng[k]) (void)hipFree(e->d_ctus_eng[k]);
		if (e->d_rowstate_eng[k]) (void)hipFree(e->d_rowstate_eng[k]);
		if (e->d_seen_eng[k]) (void)hipFree(e->d_seen_eng[k]);
	}
	if (e->ev_frame) (void)hipEventDestroy(e->ev_frame);
	if (e->ev_ready) (void)hipEventDestroy(e->ev_ready);
	if (e->ev_batch0) (void)hipEventDestroy(e->ev_batch0);
	if (e->ev_batch1) (void)hipEventDestroy(e->ev_batch1);
	if (e->ev_packed) (void)hipEventDestroy(e->ev_packed);
	if (e->ev_decided) (void)hipEventDestroy(e->ev_decided);
	if (e->d_frames) (void)hipFree(e->d_frames);
	if (e->h_frames) (void)hipHostFree(e->h_frames);
	if (e->h_devs) (void)hipHostFree(e->h_devs);
	for (int k = 0; k < 2; k++) if (e->plane_stream[k]) { (void)hipStreamSynchronize(e->plane_stream[k]); (void)hipStreamDestroy(e->plane_stream[k]); }
	for (int k = 0; k < 3; k++) if (e->ev_plane[k]) (void)hipEventDestroy(e->ev_plane[k]);
	if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
	if (e->d_gather) (void)hipFree(e->d_gather);
	if (e->h_gather) (void)hipHostFree(e->h_gather);
	if (e->h_offs) (void)hipHostFree(e->h_offs);
	if (e->d_batch) (void)hipFree(e->d_batch);
	if (e->d_pool_state) (void)hipFree(e->d_pool_state);
	if (e->d_pool_slow) (void)hipFree(e->d_pool_slow);
	if (e->d_stage) (void)hipFree(e->d_stage);
	if (e->h_stage) (void)hipHostFree(e->h_stage);
	void *p[] = {e->d_seq, e->d_frame, e->d_geo, e->d.ctus, e->d.ctus_start, e->d.work_slow, e->d.coeff, e->d.progress, e->d.prefix, e->d.prof, e->d.guess, e->d.truth, e->d.outtok,
		     e->d.chain_start, e->d.chain_end, e->d.valid, e->d.dirty, e->d.hash, e->d.intra_before, e->d.used_intra, e->d.used_parts, e->d.counters, e->d.rowstate, e->d.thread_seen, e->d.row0_checked, e->d_bytes, e->d_mvx,
		     e->d_mvy, e->d_ref, e->d_qp, e->d_flags, e->d_public};
	for (void *q : p) (void)hipFree(q);
	for (int c = 0; c < 3; c++) {
		(void)hipFree(e->d_pic[0][c]);
		(void)hipFree(e->d_pic[1][c]);
		(void)hipFree(e->d_pre[c]);
		(void)hipFree(e->d_rec[c]);
	}
	{
		void *pp[] = {e->d_rows, e->d_ent, e->d_bs, e->d_cumbits, e->d_sao_tab, e->d_post_err, e->d_rc_need, e->d_rc_dyn, e->d_ctx_ring, e->d_rd_init, e->d_rd_src};
		for (void *q : pp) (void)hipFree(q);
	}
	for (auto &sl : e->src)
		for (int c = 0; c < 3; c++) (void)hipFree(sl.p[c]);
	delete e;
	g_plane_pool.encoder_destroyed();
}

// profiling build (-DHENC_POST_PROFILE): s_memtime ticks per part of the post-decision stage since the encoder was created (enc_post.h PostProf), 16 entries
extern "C" int hmr_gpu_enc_post_profile(hmr_gpu_enc *e, unsigned long long *out)
{
	if (!e || !out) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipMemcpy(out, e->d_post_err + 4, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#if defined(HENC_POST_PROFILE)
	HIP_TRY(hipMemcpyFromSymbol(out + 10, HIP_SYMBOL(g_ent_prof), 6 * sizeof(unsigned long long)));
#endif
	return HMR_GPU_OK;
}

extern "C" float hmr_gpu_enc_last_ctu_ms(hmr_gpu_enc *e) { return e ? e->last_ms : 0.f; }

extern "C" int hmr_gpu_enc_stale_predictions(hmr_gpu_enc *e, long *last_picture, long *all_pictures)
{
	if (!e) return HMR_GPU_ERR_ARG;
	if (last_picture) *last_picture = e->stale_last;
	if (all_pictures) *all_pictures = e->stale_total;
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_enc_last_stats(hmr_gpu_enc *e, int *passes, int *ctu_encodes, float *ctu_ms, float *frame_ms)
{
	if (!e) return HMR_GPU_ERR_ARG;
	if (passes) *passes = e->last_passes;
	if (ctu_encodes) *ctu_encodes = e->last_encodes;
	if (ctu_ms) *ctu_ms = e->last_ms;
	if (frame_ms) *frame_ms = e->last_total_ms;
	return HMR_GPU_OK;
}

// profiling build: per CTU, 100 MHz timestamps of {wait start, encode start, first use of the intra share (0: none), end} in the row-per-thread schedule
extern "C" int hmr_gpu_enc_timeline(hmr_gpu_enc *e, unsigned long long *out)
{
	if (!e || !out) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipMemcpy(out, e->d.prof + (size_t)e->seq.hctu * PF_COUNT, sizeof(unsigned long long) * e->seq.nctu * 4, hipMemcpyDeviceToHost));
	return HMR_GPU_OK;
}

// profiling build (-DHENC_PROFILE): per-row phase timers in s_memtime ticks, [hctu][PF_COUNT]; all zero otherwise
extern "C" int hmr_gpu_enc_profile(hmr_gpu_enc *e, unsigned long long *out, int reset)
{
	if (!e || !out) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipMemcpy(out, e->d.prof, sizeof(unsigned long long) * e->seq.hctu * PF_COUNT, hipMemcpyDeviceToHost));
	if (reset) {
		HIP_TRY(hipMemsetAsync(e->d.prof, 0, sizeof(unsigned long long) * e->seq.hctu * PF_COUNT, e->ctx->stream));
		HIP_TRY(hipStreamSynchronize(e->ctx->stream));
	}
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_enc_load_source(hmr_gpu_enc *e, int slot, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
	if (!e || slot < 0 || slot > 4096 || !y || !u || !v) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipSetDevice(e->ctx->device));
	while ((int)e->src.size() <= slot) {
		SrcSlot sl;
		for (int c = 0; c < 3; c++) DEV_ALLOC(sl.p[c], e->src_elems[c]);
		e->src.push_back(sl);
	}
	return load_planes(e, y, u, v, e->src[slot].p, e->seq.src_stride_y, e->seq.src_stride_c);
}

extern "C" int hmr_gpu_enc_frame_ctus(hmr_gpu_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
				      const uint8_t *ref_v, double avg_dist, uint8_t *records)
{
	if (!e || !y || !u || !v) return HMR_GPU_ERR_ARG;
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	HIP_TRY(hipSetDevice(e->ctx->device));
	int rc = hmr_gpu_enc_load_source(e, 0, y, u, v);
	if (rc) return rc;
	rc = set_frame(e, 0, image_type, avg_dist);
	if (rc) return rc;
	if (ref_y && ref_u && ref_v) {
		int16_t *dst[3] = {plane0(e, e->cur ^ 1, 0), plane0(e, e->cur ^ 1, 1), plane0(e, e->cur ^ 1, 2)};
		rc = load_planes(e, ref_y, ref_u, ref_v, dst, s.stride_y, s.stride_c);
		if (rc) return rc;
		hmr_gpu_frame fr = {s.width, s.height, dst[0], dst[1], dst[2], s.stride_y, s.stride_c};
		rc = hmr_gpu_pad_frame(e->ctx, &fr, s.margin_y, s.margin_y);
		if (rc) return rc;
	}
	rc = run_ctu_passes(e);
	if (rc) return rc;
	// frame statistics (encoder_engine_thread :3217-3238)
	rc = download_public(e);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(st));
	end_frame(s, e->st, e->f, frame_acc_dist(s, e->cfg.wfpp_num_threads, [&](int n) { return ((const CtuPublic *)(e->h_public.data() + sizeof(CtuPublic) * n))->distortion; }));
	if (records) {
		std::vector<int16_t> rec[3];
		std::vector<uint8_t> truth((size_t)s.nctu * MODE_STATE_BYTES), chain_end(MODE_STATE_BYTES);
		std::vector<uint32_t> node0(3 * s.nctu);
		HIP_TRY(hipMemcpy(e->h_coeff.data(), e->d.coeff, e->h_coeff.size() * 2, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(truth.data(), e->d.truth, truth.size(), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(chain_end.data(), e->d.chain_end, MODE_STATE_BYTES, hipMemcpyDeviceToHost));
		for (int n = 0; n < s.nctu; n++) {
			Node nd;
			HIP_TRY(hipMemcpy(&nd, &e->d.ctus[n].nodes[0], sizeof(Node), hipMemcpyDeviceToHost));
			node0[3 * n] = nd.cost; node0[3 * n + 1] = nd.distortion; node0[3 * n + 2] = nd.sum;
		}
		for (int c = 0; c < 3; c++) {
			rec[c].resize(e->pic_elems[c]);
			HIP_TRY(hipMemcpy(rec[c].data(), e->d_rec[c], e->pic_elems[c] * 2, hipMemcpyDeviceToHost));
		}
		memset(records, 0, (size_t)REC_BYTES * s.nctu);
		for (int n = 0; n < s.nctu; n++) {
			uint8_t *o = records + (size_t)n * REC_BYTES;
			const CtuPublic &ci = *(const CtuPublic *)(e->h_public.data() + sizeof(CtuPublic) * n);
			int32_t hdr[8] = {0x43545544, e->f.num_encoded_frames, n, e->f.slice_type, (int32_t)node0[3 * n], (int32_t)node0[3 * n + 1], (int32_t)node0[3 * n + 2],
					  e->f.scene_cut_ctu >= 0 && n >= e->f.scene_cut_ctu};
			memcpy(o, hdr, 32); o += 32;
			for (int k = 0; k < 3; k++) { memcpy(o, ci.cbf[k], 256); o += 256; }
			memcpy(o, ci.intra_mode[0], 256); o += 256;
			memcpy(o, ci.intra_mode[1], 256); o += 256;
			const uint8_t *arrs[9] = {ci.inter_mode, ci.tr_idx, ci.pred_depth, ci.part_size_type, ci.pred_mode, ci.skipped, ci.merge, ci.merge_idx, ci.qp};
			for (int k = 0; k < 9; k++) { memcpy(o, arrs[k], 256); o += 256; }
			memcpy(o, ci.mv_ref_idx, 256); o += 256;
			memcpy(o, ci.mv_diff_ref_idx, 256); o += 256;
			memcpy(o, ci.mv_ref, 2048); o += 2048;
			memcpy(o, ci.mv_diff, 2048); o += 2048;
			memcpy(o, e->h_coeff.data() + (size_t)n * 6144, 12288); o += 12288;
			// reconstruction before the loop filters: the part of the CTU inside the picture (the rest stays zero)
			for (int c = 0; c < 3; c++) {
				const int nn = c ? 32 : 64, px = (ci.x >> (c ? 1 : 0)), py = (ci.y >> (c ? 1 : 0));
				const int pw = c ? s.width / 2 : s.width, ph = c ? s.height / 2 : s.height, rs = c ? s.stride_c : s.stride_y, m = c ? s.margin_c : s.margin_y;
				const int16_t *p = rec[c].data() + (size_t)m * rs + m;
				for (int yy = 0; yy < nn; yy++) {
					if (py + yy < ph) {
						const int ww = px + nn <= pw ? nn : pw - px;
						memcpy(o, p + (size_t)(py + yy) * rs + px, ww * 2);
					}
					o += nn * 2;
				}
			}
			// the single thread's mode buffers after the CTU
			memcpy(o, n + 1 < s.nctu ? truth.data() + (size_t)(n + 1) * MODE_STATE_BYTES : chain_end.data(), MODE_STATE_BYTES);
		}
	}
	return e->f.slice_type;
}

namespace {
constexpr int GATHER_HEAD = 12;         // words in front of the CTUs' distortions in a picture's gather record
constexpr int CHAIN_MAX_FRAMES = 32;    // frames of one hmr_gpu_enc_encode_chain call
__global__ void k_gather_results(const EncDev *devs, uint32_t *out, int pitch, const int *pool_flags)
{
	const EncDev &d = devs[blockIdx.x];
	uint32_t *o = out + (size_t)blockIdx.x * pitch;
	const int nctu = d.seq->nctu;
	if (threadIdx.x < 3) o[threadIdx.x] = (uint32_t)d.counters[threadIdx.x];
	// rate control: the sum of the CTUs' QPs (the root nodes': acc_qp, hmr_encoder_lib.c:2938), the bits of all CTUs, the picture target as the frame left it
	__shared__ uint32_t s_qp, s_bits, s_stale;
	if (threadIdx.x == 0) { s_qp = 0; s_bits = 0; s_stale = 0; }
	__syncthreads();
	uint32_t q = 0, b = 0, stale = 0;
	for (int c = threadIdx.x; c < nctu; c += blockDim.x) { q += d.ctus[c].nodes[0].qp; stale += (uint32_t)d.ctus[c].n_stale_pred; }
	for (int r = threadIdx.x; r < d.seq->hctu; r += blockDim.x) b += d.post.cumbits[r * d.seq->wctu + d.seq->wctu - 1];
	atomicAdd(&s_qp, q);
	atomicAdd(&s_bits, b);
	atomicAdd(&s_stale, stale);
	__syncthreads();
	// word 3: bit 0 a row's sub-stream outgrew its buffer, bit 1 the launch was abandoned, bits 8 ...: the picture's evaluations on a stale prediction window (Q12)
	if (threadIdx.x == 0) o[3] = (uint32_t)(d.post.errors[0] != 0) | ((uint32_t)(pool_flags && pool_flags[1] != 0) << 1) | ((s_stale > 0xffffffu ? 0xffffffu : s_stale) << 8);
	if (threadIdx.x == 0) {
		o[4] = s_qp;
		o[5] = s_bits;
		const double tp = d.counters[2] >= 0 ? d.rc_dyn->target_pict_size : d.frame->rc.target_pict_size;
		memcpy(&o[6], &tp, 8);
		// chains: the distortion total as the launch computed it, and the average distortion the picture started from (the launch sets it for an engine's later pictures)
		const double fin = d.fin[0], avg = d.frame->avg_dist;
		memcpy(&o[8], &fin, 8);
		memcpy(&o[10], &avg, 8);
	}
	for (int c = threadIdx.x; c < nctu; c += blockDim.x) o[GATHER_HEAD + c] = d.ctus[c].distortion;
	// bytes of the rows' sub-streams, behind the distortions
	for (int r = threadIdx.x; r < d.seq->hctu; r += blockDim.x) o[pitch - POST_MAX_ROWS + r] = (uint32_t)d.post.ent[r].bytecnt;
}

}  // namespace

// HOMER_enc_encode for one picture already on the device (hmr_gpu_enc_load_source): CTU decisions, deblocking, SAO, entropy coding of the CTU rows' sub-streams and
// border padding on the device (the CTU kernel and its post-decision tasks, enc_post.h); the host writes the headers and assembles the access unit.
namespace {
// the access unit of frame `f` from the rows' sub-streams (`bytes`: the sub-streams one after the other) into `stream`
int frame_assemble(hmr_gpu_enc *e, const FrameCtx &f, const uint8_t *bytes, const uint32_t *row_bytes, uint8_t *stream, long cap, long *stream_bytes)
{
	const Seq &s = e->seq;
	const int rows = s.wpp ? s.hctu : 1;
	std::vector<const uint8_t *> data(rows);
	std::vector<int> nb(rows);
	size_t o = 0;
	for (int r = 0; r < rows; r++) { data[r] = bytes + o; nb[r] = (int)row_bytes[r]; o += row_bytes[r]; }
	std::vector<uint8_t> out;
	assemble_access_unit(e->es, s, f, e->cfg.profile, data.data(), nb.data(), out);
	*stream_bytes = (long)out.size();
	if ((long)out.size() > cap) {
		hmr_set_error("hmr_gpu_enc_encode: the access unit needs %ld bytes, the buffer holds %ld", (long)out.size(), cap);
		return HMR_GPU_ERR_ARG;
	}
	memcpy(stream, out.data(), out.size());
	return f.slice_type;
}
// one sequence, behind its CTU stage (which has been waited for): the sub-streams and the CTUs' distortions to the host, the access unit, the frame bookkeeping
int frame_finish(hmr_gpu_enc *e, int slot, uint8_t *stream, long cap, long *stream_bytes, uint8_t *recon)
{
	(void)slot;
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	int rc, err[4];
	HIP_TRY(hipMemcpyAsync(e->h_ent.data(), e->d_ent, sizeof(RowEnt) * s.hctu, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(err, e->d_post_err, sizeof err, hipMemcpyDeviceToHost, st));
	const int pitch = GATHER_HEAD + s.nctu + POST_MAX_ROWS;
	if (e->lockstep) {
		// the frame's counters, distortions and rate-control sums in one small record (the picture's EncDev is at d_batch: run_ctu_passes)
		if ((size_t)pitch > e->gather_words) {
			if (e->d_gather) (void)hipFree(e->d_gather);
			if (e->h_gather) (void)hipHostFree(e->h_gather);
			e->d_gather = e->h_gather = nullptr;
			e->gather_words = 0;
			HIP_TRY(hipMalloc((void **)&e->d_gather, (size_t)pitch * 4));
			HIP_TRY(hipHostMalloc((void **)&e->h_gather, (size_t)pitch * 4, hipHostMallocDefault));
			e->gather_words = (size_t)pitch;
		}
		hipLaunchKernelGGL(k_gather_results, dim3(1), dim3(256), 0, st, (const EncDev *)e->d_batch, e->d_gather, pitch, (const int *)nullptr);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(e->h_gather, e->d_gather, (size_t)pitch * 4, hipMemcpyDeviceToHost, st));
	} else if ((rc = download_public(e))) return rc;
	HIP_TRY(hipStreamSynchronize(st));
	if (getenv("HENC_DEBUG_POST")) {
		std::vector<PostRow> rows(s.hctu);
		HIP_TRY(hipMemcpy(rows.data(), e->d_rows, sizeof(PostRow) * s.hctu, hipMemcpyDeviceToHost));
		for (int r = 0; r < s.hctu; r++)
			fprintf(stderr, "post row %d: dec %d D %d/%d P %d/%d F %d/%d\n", r, rows[r].dec, rows[r].d_claim, rows[r].d_done, rows[r].p_claim, rows[r].p_done, rows[r].f_claim, rows[r].f_done);
		fprintf(stderr, "post errors %d %d %d %d, sao %d\n", err[0], err[1], err[2], err[3], s.sao);
	}
	if (err[0]) {
		hmr_set_error("hmr_gpu_enc_encode: a CTU row's sub-stream outgrew its buffer (%d bytes)", e->row_cap);
		return HMR_GPU_ERR_HIP;
	}
	if (err[2]) {
		hmr_set_error("hmr_gpu_enc_encode: the post-decision stage was abandoned by its watchdog");
		return HMR_GPU_ERR_HIP;
	}
	const int rows = s.wpp ? s.hctu : 1;
	std::vector<uint32_t> row_bytes(rows);
	size_t total = 0;
	for (int r = 0; r < rows; r++) {
		row_bytes[r] = (uint32_t)e->h_ent[r].bytecnt;
		HIP_TRY(hipMemcpyAsync(e->h_bs.data() + total, e->d_bs + (size_t)r * e->row_cap, row_bytes[r], hipMemcpyDeviceToHost, st));
		total += row_bytes[r];
	}
	if (recon) {
		uint8_t *o = recon;
		for (int c = 0; c < 3; c++) {
			const int w = c ? s.width / 2 : s.width, h = c ? s.height / 2 : s.height;
			hipLaunchKernelGGL(k_narrow_plane, dim3((w + 255) / 256, h), dim3(256), 0, st, plane0(e, e->cur, c), c ? s.stride_c : s.stride_y, w, h, e->d_bytes);
			HIP_TRY(hipMemcpyAsync(o, e->d_bytes, (size_t)w * h, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			o += (size_t)w * h;
		}
	}
	HIP_TRY(hipStreamSynchronize(st));
	const uint32_t *gr = e->h_gather;
	if (e->lockstep) e->note_stale_predictions(gr[3] >> 8);
	const double acc = e->lockstep ? frame_acc_dist(s, e->cfg.wfpp_num_threads, [&](int n) { return gr[GATHER_HEAD + n]; })
				       : frame_acc_dist(s, e->cfg.wfpp_num_threads, [&](int n) { return ((const CtuPublic *)(e->h_public.data() + sizeof(CtuPublic) * n))->distortion; });
	// (a buffer that is too small loses the access unit; the sequence state has not moved on, but the device pictures have: the caller has to start over)
	rc = frame_assemble(e, e->f, e->h_bs.data(), row_bytes.data(), stream, cap, stream_bytes);
	if (rc < 0) return rc;
	FrameRcOut ro = {0, 0.0, 0.0};
	if (e->lockstep) {
		ro.sum_qp = (int)gr[4]; ro.consumed_bits = (double)gr[5];
		memcpy(&ro.target_pict_size, &gr[6], 8);
	}
	end_frame(s, e->st, e->f, acc, &ro);
	HIP_TRY(hipEventRecord(e->ctx->ev1, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipEventElapsedTime(&e->last_total_ms, e->ev_frame, e->ctx->ev1));
	return e->f.slice_type;
}
}  // namespace

extern "C" int hmr_gpu_enc_encode_source(hmr_gpu_enc *e, int slot, int image_type, uint8_t *stream, long cap, long *stream_bytes, uint8_t *recon)
{
	if (!e || slot < 0 || slot >= (int)e->src.size() || !stream || !stream_bytes) return HMR_GPU_ERR_ARG;
	if (e->awaiting_delivery) {
		hmr_set_error("hmr_gpu_enc_encode_source: the encoder has an access unit outstanding from a pipelined batch call: flush first");
		return HMR_GPU_ERR_ARG;
	}
	hipStream_t st = e->ctx->stream;
	HIP_TRY(hipSetDevice(e->ctx->device));
	HIP_TRY(hipEventRecord(e->ev_frame, st));
	int rc = set_frame(e, slot, image_type, -1.0);
	if (rc) return rc;
	rc = run_ctu_passes(e);
	if (rc) return rc;
	return frame_finish(e, slot, stream, cap, stream_bytes, recon);
}

// Several sequences, one frame each, with ONE launch for all their CTU stages (k_encode_pool): encs[i] encodes its picture slots[i] into streams[i].
// All encoders must use the row-per-thread schedule and live on the same device; each finishes its frame (filters, SAO, records and levels into the staging buffer)
// on its own stream.  The streams are those hmr_gpu_enc_encode_source would have produced one by one.
//
// A step has three parts: LAUNCH (frame set-up, phase planes, the pool launch), FINISH (after the launch: the frames' counters and distortions in one small
// download, frame bookkeeping, a host thread per sequence queues its filter chain and packs its records and levels into the staging buffer, whose download to the
// host is queued behind them on a copy stream) and DELIVER (wait for that download, a host thread per sequence codes its access unit).  The plain call runs
// LAUNCH, FINISH, DELIVER; the pipelined call runs LAUNCH(k), DELIVER(k - 1), FINISH(k), so that the download and the entropy coding of a step run while the
// device is busy with the next step's CTU stage (nothing of step k reads what DELIVER(k - 1) reads: the staging buffers are written again only in FINISH(k)).
namespace {
// the sub-streams of the pictures of a launch, row after row, into the staging buffer (picture i at out + offs[i]): what the host downloads to assemble the access units
__global__ __launch_bounds__(256) void k_pack_streams(const EncDev *devs, uint8_t *out, const size_t *offs)
{
	__shared__ uint32_t start[POST_MAX_ROWS + 1];
	const EncDev &d = devs[blockIdx.x];
	const int rows = d.seq->wpp ? d.seq->hctu : 1;
	if (threadIdx.x == 0) {
		uint32_t o = 0;
		for (int r = 0; r < rows; r++) { start[r] = o; o += (uint32_t)d.post.ent[r].bytecnt; }
		start[rows] = o;
	}
	__syncthreads();
	uint8_t *dst = out + offs[blockIdx.x];
	for (int r = 0; r < rows; r++) {
		const uint8_t *src = d.post.bs + (size_t)r * d.post.row_cap;
		const uint32_t n = start[r + 1] - start[r];
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[start[r] + i] = src[i];
	}
}

// The batch's pictures: their EncDev records and frame parameters come from page-locked host memory (read by the kernel itself: a host-to-device copy queued here
// waited behind the previous step's 2 GB download on the copy engines - rocprofv3 trace, 35 ms), and what ctu_stage_prepare does for one picture is done for all.
__global__ void k_batch_stage(const EncDev *h_devs, const FrameCtx *h_frames, EncDev *d_devs, FrameCtx *d_frames)
{
	const int i = blockIdx.x, t = threadIdx.x;
	for (int k = t; k < (int)(sizeof(EncDev) / 4); k += blockDim.x) ((uint32_t *)(d_devs + i))[k] = ((const uint32_t *)(h_devs + i))[k];
	for (int k = t; k < (int)(sizeof(FrameCtx) / 4); k += blockDim.x) ((uint32_t *)(d_frames + i))[k] = ((const uint32_t *)(h_frames + i))[k];
	const EncDev d = h_devs[i];
	const int H = d.seq->hctu, W = d.seq->wctu;
	if (t < 3) d.counters[t] = t == 2 ? -1 : 0;
	if (t == 3) *d.row0_checked = 0;
	if (t >= 4 && t < 8) d.post.errors[t - 4] = 0;
	for (int k = t; k < H * (int)(sizeof(PostRow) / 4); k += blockDim.x) ((int *)d.post.rows)[k] = 0;
	for (int k = t; k < H; k += blockDim.x) d.progress[k] = 0;
	for (int k = t; k < H * (W + 1); k += blockDim.x) d.prefix[k] = 0;
}

struct BatchTimes {
	std::chrono::steady_clock::time_point t[8];
	double ms(int a, int b) const { return std::chrono::duration<double, std::milli>(t[b] - t[a]).count(); }
};

// fn(i) for i in [0, n) on up to `threads` host threads (sequence i goes to thread i mod threads).  A thread per sequence cost more in thread start-up than the
// few dozen stream calls a sequence needs; a dozen threads keep the runtime's submission path busy just as well.
template <class F>
void parallel_for(int n, int threads, F fn)
{
	const int T = n < threads ? n : threads;
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++)
		th.emplace_back([=]() {
			for (int i = t; i < n; i += T) fn(i);
		});
	for (auto &x : th) x.join();
}
constexpr int QUEUE_THREADS = 16;       // for queueing device work
constexpr int CODING_THREADS = 32;      // for the access units (headers, entry points, escaping: a few microseconds per kilobyte)

int batch_check(hmr_gpu_enc **encs, int n, const int *slots, uint8_t **streams, const long *caps, long *stream_bytes)
{
	if (!encs || n <= 0 || n > 256 || !streams || !caps || !stream_bytes) return HMR_GPU_ERR_ARG;
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		if (!e || !e->lockstep || e->ctx->device != encs[0]->ctx->device || (slots && (slots[i] < 0 || slots[i] >= (int)e->src.size())) || !streams[i]) {
			hmr_set_error("hmr_gpu_enc_encode_batch: encoder %d: needs the row-per-thread schedule (wfpp_num_threads > 1), the batch's device and a loaded picture slot", i);
			return HMR_GPU_ERR_ARG;
		}
		for (int j = 0; j < i; j++)
			if (encs[j] == e) return HMR_GPU_ERR_ARG;
	}
	return HMR_GPU_OK;
}

// LAUNCH: the frames' CTU stages as one pool launch on the lead encoder's stream, their counters and distortions gathered behind it
int batch_launch(hmr_gpu_enc **encs, int n, const int *slots, const int *image_types, int *pitch_out)
{
	hmr_gpu_enc *lead = encs[0];
	hipStream_t bst = lead->ctx->stream;
	int rc, rows_total = 0, max_ctus = 0;
	bool needs_rd = false;
	// frame set-up on the host; nothing is queued on the sequences' streams: the launch's stream waits for what each of them still has in flight (the filter
	// chain and packing of its previous frame) and takes the rest - the frames' parameters in one upload, the phase planes, the per-frame state in one kernel
	if (!lead->d_frames) {
		HIP_TRY(hipMalloc((void **)&lead->d_frames, 256 * sizeof(FrameCtx)));
		HIP_TRY(hipHostMalloc((void **)&lead->h_frames, 256 * sizeof(FrameCtx), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&lead->h_devs, 256 * sizeof(EncDev), hipHostMallocDefault));
	}
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		if ((rc = set_frame(e, slots[i], image_types ? image_types[i] : 0, -1.0, false))) return rc;
		HIP_TRY(hipEventRecord(e->ev_ready, e->ctx->stream));
		if (i) HIP_TRY(hipStreamWaitEvent(bst, e->ev_ready, 0));
		lead->h_frames[i] = e->f;
		lead->h_devs[i] = e->d;
		lead->h_devs[i].frame = lead->d_frames + i;
		rows_total += pool_inflight(e->seq);
		needs_rd = needs_rd || e->seq.rd_mode == RDM_FULL;
		if (e->seq.nctu > max_ctus) max_ctus = e->seq.nctu;
	}
	// The phase planes of all the reference pictures one after the other on the launch's stream: a picture's three kernels fill the GPU (2500 workgroups, 1.9 TB/s);
	// run side by side on the sequences' streams, sixteen at a time, they reached a quarter of that between them (rocprofv3 trace: 52 ms for 180 pictures).
	// ... luma on the launch's stream, U and V on two side streams: a picture's kernels are 33 + 2 x 20 us of a GPU they do not fill at their ends, three in
	// flight overlap those ends (sixteen thrash, one leaves them exposed)
	if (!lead->plane_stream[0]) {
		for (int k = 0; k < 2; k++) {
			HIP_TRY(hipStreamCreateWithFlags(&lead->plane_stream[k], hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&lead->ev_plane[k], hipEventDisableTiming));
		}
		HIP_TRY(hipEventCreateWithFlags(&lead->ev_plane[2], hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(lead->ev_plane[2], bst));
	for (int k = 0; k < 2; k++) HIP_TRY(hipStreamWaitEvent(lead->plane_stream[k], lead->ev_plane[2], 0));
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		const Seq &s = e->seq;
		if (e->f.slice_type == SLICE_I) continue;
		if ((rc = hmr_subpel_plane_on(bst, 0, e->d_pic[e->cur ^ 1][0], s.stride_y, s.height + 2 * s.margin_y, e->planes.y))) return rc;
		for (int k = 0; k < 2; k++)
			if ((rc = hmr_subpel_plane_on(lead->plane_stream[k], 1 + k, e->d_pic[e->cur ^ 1][1 + k], s.stride_c, s.height / 2 + 2 * s.margin_c, e->planes.c[k]))) return rc;
	}
	for (int k = 0; k < 2; k++) {
		HIP_TRY(hipEventRecord(lead->ev_plane[k], lead->plane_stream[k]));
		HIP_TRY(hipStreamWaitEvent(bst, lead->ev_plane[k], 0));
	}
	if (!lead->d_batch) {
		HIP_TRY(hipMalloc((void **)&lead->d_batch, 256 * sizeof(EncDev)));
		HIP_TRY(hipDeviceGetAttribute(&lead->n_cus, hipDeviceAttributeMultiprocessorCount, lead->ctx->device));
	}
	const int pitch = GATHER_HEAD + max_ctus + POST_MAX_ROWS;
	if ((size_t)pitch * n > lead->gather_words) {
		if (lead->d_gather) (void)hipFree(lead->d_gather);
		if (lead->h_gather) (void)hipHostFree(lead->h_gather);
		lead->d_gather = lead->h_gather = nullptr;
		lead->gather_words = 0;
		HIP_TRY(hipMalloc((void **)&lead->d_gather, (size_t)pitch * 256 * 4));
		HIP_TRY(hipHostMalloc((void **)&lead->h_gather, (size_t)pitch * 256 * 4, hipHostMallocDefault));
		lead->gather_words = (size_t)pitch * 256;
	}
	hipLaunchKernelGGL(k_batch_stage, dim3(n), dim3(256), 0, bst, (const EncDev *)lead->h_devs, (const FrameCtx *)lead->h_frames, (EncDev *)lead->d_batch, lead->d_frames);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(lead->ev_batch0, bst));
	if ((rc = launch_pool(lead, n, rows_total, needs_rd, bst))) return rc;
	(void)hipEventRecord(lead->ev_batch1, bst);
	hipLaunchKernelGGL(k_gather_results, dim3(n), dim3(256), 0, bst, (const EncDev *)lead->d_batch, lead->d_gather, pitch, (const int *)(lead->d_pool_state + 256 * POOL_STRIDE));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(lead->h_gather, lead->d_gather, (size_t)pitch * n * 4, hipMemcpyDeviceToHost, bst));
	*pitch_out = pitch;
	return HMR_GPU_OK;
}

// FINISH: wait for the launch (CTU decisions, filters and entropy coding of every picture are done when it ends); frame bookkeeping from the gathered counters,
// distortions and sub-stream sizes; the sub-streams of all pictures packed into one staging buffer by one kernel and their download queued on the copy stream
int batch_finish(hmr_gpu_enc **encs, int n, const int *slots, int pitch, BatchTimes &bt)
{
	(void)slots;
	hmr_gpu_enc *lead = encs[0];
	hipStream_t bst = lead->ctx->stream;
	const hipError_t waited = hipStreamSynchronize(bst);
	if (waited != hipSuccess) {
		hmr_set_error("k_encode_pool: %s", hipGetErrorString(waited));
		return HMR_GPU_ERR_HIP;
	}
	bt.t[2] = std::chrono::steady_clock::now();
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, lead->ev_batch0, lead->ev_batch1));
	if (!lead->h_offs) HIP_TRY(hipHostMalloc((void **)&lead->h_offs, 256 * sizeof(size_t), hipHostMallocDefault));
	lead->pend_off.resize(n);
	lead->pend_rows.assign(n, std::vector<uint32_t>());
	size_t total = 0;
	for (int i = 0; i < n; i++) {
		hmr_gpu_enc *e = encs[i];
		const uint32_t *g = lead->h_gather + (size_t)i * pitch;
		if (g[3] & 2) {
			hmr_set_error("k_encode_pool: the launch was abandoned by its watchdog (a worker found nothing to do for too long: HENC_WATCHDOG_S)");
			return HMR_GPU_ERR_HIP;
		}
		if (g[3] & 1) {
			hmr_set_error("hmr_gpu_enc_encode_batch: sequence %d: a CTU row's sub-stream outgrew its buffer (%d bytes)", i, e->row_cap);
			return HMR_GPU_ERR_HIP;
		}
		e->note_stale_predictions(g[3] >> 8);
		const int rows = e->seq.wpp ? e->seq.hctu : 1;
		lead->pend_off[i] = total;
		lead->h_offs[i] = total;
		lead->pend_rows[i].assign(g + pitch - POST_MAX_ROWS, g + pitch - POST_MAX_ROWS + rows);
		for (int r = 0; r < rows; r++) total += lead->pend_rows[i][r];
		total = (total + 255) & ~(size_t)255;
		e->last_ms = e->last_total_ms = ms;
		e->last_encodes = (int)g[1];
		e->f.scene_cut_ctu = (int)g[2];
		e->last_passes = 1;
		release_planes(e);
		// the frame's statistics (encoder_engine_thread :3217-3238) need the CTUs' distortions only: the sequence can start its next frame
		e->f_pending = e->f;
		e->awaiting_delivery = true;
		FrameRcOut ro;
		ro.sum_qp = (int)g[4]; ro.consumed_bits = (double)g[5];
		memcpy(&ro.target_pict_size, &g[6], 8);
		end_frame(e->seq, e->st, e->f, frame_acc_dist(e->seq, e->cfg.wfpp_num_threads, [&](int c) { return g[GATHER_HEAD + c]; }), &ro);
	}
	if (total > lead->stage_bytes) {
		if (lead->d_stage) (void)hipFree(lead->d_stage);
		if (lead->h_stage) (void)hipHostFree(lead->h_stage);
		lead->d_stage = lead->h_stage = nullptr;
		lead->stage_bytes = 0;
		const size_t want = total * 2 + (1 << 20);
		HIP_TRY(hipMalloc((void **)&lead->d_stage, want));
		HIP_TRY(hipHostMalloc((void **)&lead->h_stage, want, hipHostMallocDefault));
		lead->stage_bytes = want;
	}
	if (!lead->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&lead->copy_stream, hipStreamNonBlocking));
	hipLaunchKernelGGL(k_pack_streams, dim3(n), dim3(256), 0, bst, (const EncDev *)lead->d_batch, lead->d_stage, (const size_t *)lead->h_offs);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(lead->ev_decided, bst));
	HIP_TRY(hipStreamWaitEvent(lead->copy_stream, lead->ev_decided, 0));
	if (total) HIP_TRY(hipMemcpyAsync(lead->h_stage, lead->d_stage, total, hipMemcpyDeviceToHost, lead->copy_stream));
	bt.t[3] = std::chrono::steady_clock::now();
	lead->pend_encs.assign(encs, encs + n);
	lead->pend_total = total;
	lead->pending = true;
	return HMR_GPU_OK;
}

// DELIVER: the outstanding step's access units
int batch_deliver(hmr_gpu_enc *lead, uint8_t **streams, const long *caps, long *stream_bytes, BatchTimes &bt)
{
	const int n = (int)lead->pend_encs.size();
	bt.t[4] = std::chrono::steady_clock::now();
	lead->pending = false;
	for (int i = 0; i < n; i++) lead->pend_encs[i]->awaiting_delivery = false;
	HIP_TRY(hipStreamSynchronize(lead->copy_stream));
	bt.t[5] = std::chrono::steady_clock::now();
	std::vector<int> rcs(n, 0);
	std::vector<std::string> errs(n);
	parallel_for(n, CODING_THREADS, [&](int i) {
		hmr_gpu_enc *e = lead->pend_encs[i];
		rcs[i] = frame_assemble(e, e->f_pending, lead->h_stage + lead->pend_off[i], lead->pend_rows[i].data(), streams[i], caps[i], &stream_bytes[i]);
		if (rcs[i] < 0) errs[i] = hmr_gpu_last_error();
	});
	bt.t[6] = std::chrono::steady_clock::now();
	for (int i = 0; i < n; i++)
		if (rcs[i] < 0) { hmr_set_error("hmr_gpu_enc_encode_batch: sequence %d: %s", i, errs[i].c_str()); return rcs[i]; }
	return HMR_GPU_OK;
}

void batch_report(const BatchTimes &bt, hmr_gpu_enc *lead, bool pipelined)
{
	if (!getenv("HENC_BATCH_TIMING")) return;
	static std::chrono::steady_clock::time_point last_end;
	const auto now = std::chrono::steady_clock::now();
	fprintf(stderr, "batch step%s: launch %.1f ms, wait for the pool %.1f, queue device parts %.1f, download wait %.1f (%.0f MB), entropy coding %.1f; the call %.1f, since the last call %.1f\n",
		pipelined ? " (pipelined)" : "", bt.ms(0, 1), pipelined ? bt.ms(6, 2) : bt.ms(1, 2), bt.ms(2, 3), bt.ms(4, 5), lead->pend_total / 1e6, bt.ms(5, 6),
		std::chrono::duration<double, std::milli>(now - bt.t[0]).count(), std::chrono::duration<double, std::milli>(bt.t[0] - last_end).count());
	last_end = now;
}
}  // namespace

extern "C" int hmr_gpu_enc_encode_batch(hmr_gpu_enc **encs, int n, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes)
{
	int rc = batch_check(encs, n, slots, streams, caps, stream_bytes);
	if (rc) return rc;
	if (!slots) return HMR_GPU_ERR_ARG;
	for (int i = 0; i < n; i++)
		if (encs[i]->awaiting_delivery || encs[i]->pending) {
			hmr_set_error("hmr_gpu_enc_encode_batch: encoder %d has an access unit outstanding from a pipelined call: flush first", i);
			return HMR_GPU_ERR_ARG;
		}
	hmr_gpu_enc *lead = encs[0];
	HIP_TRY(hipSetDevice(lead->ctx->device));
	BatchTimes bt;
	int pitch = 0;
	bt.t[0] = std::chrono::steady_clock::now();
	if ((rc = batch_launch(encs, n, slots, image_types, &pitch))) return rc;
	bt.t[1] = std::chrono::steady_clock::now();
	if ((rc = batch_finish(encs, n, slots, pitch, bt))) return rc;
	if ((rc = batch_deliver(lead, streams, caps, stream_bytes, bt))) return rc;
	batch_report(bt, lead, false);
	return HMR_GPU_OK;
}

// The pipelined form: call k launches the frames slots[] and delivers the access units of call k - 1's frames (stream_bytes[i] = 0 on the first call).
// slots == NULL: deliver the outstanding access units only.  The encoder list stays the same from call to call until the flush.
extern "C" int hmr_gpu_enc_encode_batch_pipelined(hmr_gpu_enc **encs, int n, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes)
{
	int rc = batch_check(encs, n, slots, streams, caps, stream_bytes);
	if (rc) return rc;
	hmr_gpu_enc *lead = encs[0];
	HIP_TRY(hipSetDevice(lead->ctx->device));
	if (lead->pending) {
		if ((int)lead->pend_encs.size() != n || memcmp(lead->pend_encs.data(), encs, n * sizeof(hmr_gpu_enc *))) {
			hmr_set_error("hmr_gpu_enc_encode_batch_pipelined: the encoder list changed while access units are outstanding: flush (slots = NULL) with the previous list first");
			return HMR_GPU_ERR_ARG;
		}
	} else {
		for (int i = 0; i < n; i++)
			if (encs[i]->awaiting_delivery) {
				hmr_set_error("hmr_gpu_enc_encode_batch_pipelined: encoder %d has an access unit outstanding in another batch", i);
				return HMR_GPU_ERR_ARG;
			}
	}
	BatchTimes bt;
	for (auto &t : bt.t) t = std::chrono::steady_clock::now();
	int pitch = 0;
	if (slots && (rc = batch_launch(encs, n, slots, image_types, &pitch))) return rc;
	bt.t[1] = std::chrono::steady_clock::now();
	if (lead->pending) {
		if ((rc = batch_deliver(lead, streams, caps, stream_bytes, bt))) return rc;
	} else {
		for (int i = 0; i < n; i++) stream_bytes[i] = 0;
		bt.t[6] = bt.t[1];
	}
	if (slots && (rc = batch_finish(encs, n, slots, pitch, bt))) return rc;
	if (slots) batch_report(bt, lead, true);
	return HMR_GPU_OK;
}

// Consecutive frames of ONE sequence in one CTU launch, overlapping as far as the reference samples allow (the engines' overlap of the reference, encoder_engine_thread
// hmr_encoder_lib.c:3154-3211 with the row semaphores of :2393-2445, in the interleaving the engine turnstile pins: oracle/ref_ctudump.c:88-108).
// encs[0 .. n - 1]: the engine objects (hmr_gpu_enc_create_engine) of the frames slots[0 .. n - 1] in coding order, n <= num_enc_engines; prev: the object that encoded
// the frame before slots[0] (NULL for the sequence's first frame).  Frame j predicts from the final picture of frame j - 1 where it lies - in encs[j - 1], or prev - and
// from the phase planes the S tasks of that picture's launch have written (enc_post.h); its CTUs of a wavefront step start when the S tasks of the part of the
// reference they can reach are done.  Frame typing and the frame scalars a frame starts from are those of the sequential order: they depend on frames at least n before
// it, except when a frame of the chain detects a scene change - if a later frame of the same chain detects one too, the call fails (HMR_GPU_ERR_ARG) and the chain has
// to be repeated frame by frame.
extern "C" int hmr_gpu_enc_encode_chain(hmr_gpu_enc **encs, int n, hmr_gpu_enc *prev, const int *slots, const int *image_types, uint8_t **streams, const long *caps, long *stream_bytes)
{
	if (!encs || n <= 0 || n > CHAIN_MAX_FRAMES || !slots || !streams || !caps || !stream_bytes) return HMR_GPU_ERR_ARG;
	for (int j = 0; j < n; j++) {
		hmr_gpu_enc *e = encs[j];
		if (!e || !e->lockstep || e->ctx->device != encs[0]->ctx->device || slots[j] < 0 || slots[j] >= (int)e->src.size() || !streams[j] || e->awaiting_delivery || e->engine_index < 0 ||
		    e->seq.width != encs[0]->seq.width || e->seq.height != encs[0]->seq.height) {
			hmr_set_error("hmr_gpu_enc_encode_chain: frame %d: needs an engine object (hmr_gpu_enc_create_engine) of the chain's sequence on the chain's device with a loaded picture slot", j);
			return HMR_GPU_ERR_ARG;
		}
		// rate control and RD_FULL read entropy-coder state of the frame before (rc_end_pic's VBV / QP carry, the context ring of the coder objects): a chain's frames
		// start from a PREDICTED host state, in which neither exists, and the replay check behind the launch would not see a wrong QP - refused, as make_seq refuses
		// them for num_enc_engines > 1 (hmr_rate_control.c:266-282, hmr_encoder_lib.c:3268-3279 are per-frame, in order)
		if (n > 1 && (e->seq.bitrate_mode != 0 || e->seq.rd_mode == RDM_FULL)) {
			hmr_set_error("hmr_gpu_enc_encode_chain: frame %d: rate control and RD_FULL need the frames one at a time (a chain of %d frames starts them from a predicted state)", j, n);
			return HMR_GPU_ERR_ARG;
		}
		// more frames than engines: an engine's next frame is encoded by a twin of its object (hmr_gpu_enc_create_engine_twin: the same persistent engine state)
		if (j >= e->st.engines && e->d_ctus_eng[0] != encs[j - e->st.engines]->d_ctus_eng[0]) {
			hmr_set_error("hmr_gpu_enc_encode_chain: frame %d: the object has to be a twin (hmr_gpu_enc_create_engine_twin) of the one that encodes frame %d, the engine's frame before it", j, j - e->st.engines);
			return HMR_GPU_ERR_ARG;
		}
		for (int i = 0; i < j; i++)
			if (encs[i] == e) return HMR_GPU_ERR_ARG;
	}
	hmr_gpu_enc *lead = encs[0];
	hipStream_t bst = lead->ctx->stream;
	HIP_TRY(hipSetDevice(lead->ctx->device));
	int rc, rows_total = 0;
	bool needs_rd = false;
	if (!lead->d_frames) {
		HIP_TRY(hipMalloc((void **)&lead->d_frames, 256 * sizeof(FrameCtx)));
		HIP_TRY(hipHostMalloc((void **)&lead->h_frames, 256 * sizeof(FrameCtx), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&lead->h_devs, 256 * sizeof(EncDev), hipHostMallocDefault));
	}
	if (!lead->d_batch) HIP_TRY(hipMalloc((void **)&lead->d_batch, 256 * sizeof(EncDev)));
	// the state every frame starts from, as the sequential order would hand it over: what begin_frame reads of it is older than the chain, a scene change excepted
	HostState st = prev ? prev->st : lead->st;
	const HostState start_state = st;
	for (int j = 0; j < n; j++) {
		hmr_gpu_enc *e = encs[j];
		const Seq &s = e->seq;
		if (st.num_encoded_frames % e->st.engines != e->engine_index) {
			hmr_set_error("hmr_gpu_enc_encode_chain: frame %d of the sequence belongs to engine %d, not %d", st.num_encoded_frames, st.num_encoded_frames % e->st.engines, e->engine_index);
			return HMR_GPU_ERR_ARG;
		}
		for (int k = 0; k < 2; k++) {
			PlaneSet &ps = k ? e->chain_planes2 : e->chain_planes;
			if (ps.y) continue;
			PlaneSet p;
			p.device = e->ctx->device; p.bytes_y = (size_t)16 * s.plane_elems_y; p.bytes_c = (size_t)64 * s.plane_elems_c;
			HIP_TRY(hipMalloc((void **)&p.y, p.bytes_y));
			HIP_TRY(hipMalloc((void **)&p.c[0], p.bytes_c));
			HIP_TRY(hipMalloc((void **)&p.c[1], p.bytes_c));
			ps = p;
		}
		e->st = st;
		// what the frame predicts from, taken before the object that holds it (prev may be the chain's last object) moves on to its own next frame
		const hmr_gpu_enc *r = j ? encs[j - 1] : prev;
		const int16_t *ref_planes[3] = {nullptr, nullptr, nullptr};
		PlaneSet ref_set;
		if (r) {
			for (int c = 0; c < 3; c++) ref_planes[c] = plane0(const_cast<hmr_gpu_enc *>(r), r->cur, c);
			ref_set = r->chain_planes;
		}
		std::swap(e->chain_planes, e->chain_planes2);      // (an object writes its two sets in turn: prev - often the chain's last object - keeps the set the chain's first frame reads)
		if (j == 0 && e == prev) ref_set = e->chain_planes2;       // (a one-engine sequence: the object predicts from its own last picture)
		if ((rc = set_frame(e, slots[j], image_types ? image_types[j] : 0, -1.0, false, true))) return rc;
		st = e->st;
		st.num_encoded_frames++;                       // (what end_frame will do; the distortion average it will store is not read inside the chain)
		if (e->f.slice_type != SLICE_I) {
			if (!r || !ref_set.y) {
				hmr_set_error("hmr_gpu_enc_encode_chain: frame %d is a P frame and there is no object that holds the picture before it with its phase planes", j);
				return HMR_GPU_ERR_ARG;
			}
			for (int c = 0; c < 3; c++) e->f.ref[c] = ref_planes[c];
			e->f.sub_y = ref_set.y + (size_t)s.margin_y * 16 * s.stride_y + s.margin_y;
			e->f.sub_c[0] = ref_set.c[0] + (size_t)s.margin_c * 64 * s.stride_c + s.margin_c;
			e->f.sub_c[1] = ref_set.c[1] + (size_t)s.margin_c * 64 * s.stride_c + s.margin_c;
			e->d.dep = j ? j - 1 : -1;
			e->d.dep_full = getenv("HENC_CHAIN_SERIAL") ? 1 : 0;
		}
		if (j >= e->st.engines) {
			// the engine's second (third ...) frame of the launch: it starts when the one before it is finished, from the average distortion that one leaves
			// (the launch writes it into this frame's parameters); an I frame inside the sequence hands on the value of the frame BEFORE it (end_frame), which
			// the launch does not have
			const hmr_gpu_enc *b = encs[j - e->st.engines];
			if (b->f.slice_type == SLICE_I && b->f.num_encoded_frames != 0 && s.intra_period != 1) {
				hmr_set_error("hmr_gpu_enc_encode_chain: frame %d is an I frame inside the sequence and the same engine's next frame is in the chain: end the chain before frame %d", j - e->st.engines, j);
				return HMR_GPU_ERR_ARG;
			}
			e->d.after = j - e->st.engines;
			e->f.avg_dist = 0.0;
			lead->h_devs[j - e->st.engines].next_frame = lead->d_frames + j;
		}
		e->d.post.planes[0] = e->chain_planes.y; e->d.post.planes[1] = e->chain_planes.c[0]; e->d.post.planes[2] = e->chain_planes.c[1];
		HIP_TRY(hipEventRecord(e->ev_ready, e->ctx->stream));
		if (j) HIP_TRY(hipStreamWaitEvent(bst, e->ev_ready, 0));
		lead->h_frames[j] = e->f;
		lead->h_devs[j] = e->d;
		lead->h_devs[j].frame = lead->d_frames + j;
		rows_total += s.hctu;
		needs_rd = needs_rd || s.rd_mode == RDM_FULL;
	}
	const int nctu = lead->seq.nctu, pitch = GATHER_HEAD + nctu + POST_MAX_ROWS;
	if ((size_t)pitch * n > lead->gather_words) {
		if (lead->d_gather) (void)hipFree(lead->d_gather);
		if (lead->h_gather) (void)hipHostFree(lead->h_gather);
		lead->d_gather = lead->h_gather = nullptr;
		lead->gather_words = 0;
		HIP_TRY(hipMalloc((void **)&lead->d_gather, (size_t)pitch * 256 * 4));
		HIP_TRY(hipHostMalloc((void **)&lead->h_gather, (size_t)pitch * 256 * 4, hipHostMallocDefault));
		lead->gather_words = (size_t)pitch * 256;
	}
	hipLaunchKernelGGL(k_batch_stage, dim3(n), dim3(256), 0, bst, (const EncDev *)lead->h_devs, (const FrameCtx *)lead->h_frames, (EncDev *)lead->d_batch, lead->d_frames);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(lead->ev_batch0, bst));
	if ((rc = launch_pool(lead, n, rows_total, needs_rd, bst))) return rc;
	(void)hipEventRecord(lead->ev_batch1, bst);
	hipLaunchKernelGGL(k_gather_results, dim3(n), dim3(256), 0, bst, (const EncDev *)lead->d_batch, lead->d_gather, pitch, (const int *)(lead->d_pool_state + 256 * POOL_STRIDE));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(lead->h_gather, lead->d_gather, (size_t)pitch * n * 4, hipMemcpyDeviceToHost, bst));
	{
		const hipError_t waited = hipStreamSynchronize(bst);
		if (waited != hipSuccess) { hmr_set_error("k_encode_pool: %s", hipGetErrorString(waited)); return HMR_GPU_ERR_HIP; }
	}
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, lead->ev_batch0, lead->ev_batch1));
	// the frames' bookkeeping in coding order, on the state the sequential order hands from frame to frame
	HostState seq_state = start_state;
	bool cut_before = false;
	for (int j = 0; j < n; j++) {
		hmr_gpu_enc *e = encs[j];
		const Seq &s = e->seq;
		const uint32_t *g = lead->h_gather + (size_t)j * pitch;
		if (g[3] & 2) { hmr_set_error("k_encode_pool: the launch was abandoned by its watchdog (HENC_WATCHDOG_S)"); return HMR_GPU_ERR_HIP; }
		if (g[3] & 1) { hmr_set_error("hmr_gpu_enc_encode_chain: frame %d: a CTU row's sub-stream outgrew its buffer (%d bytes)", j, e->row_cap); return HMR_GPU_ERR_HIP; }
		e->note_stale_predictions(g[3] >> 8);
		e->last_ms = e->last_total_ms = ms;
		e->last_encodes = (int)g[1];
		e->f.scene_cut_ctu = (int)g[2];
		e->last_passes = 1;
		const bool fired = (int)g[2] >= 0;
		if (fired && cut_before) {
			hmr_set_error("hmr_gpu_enc_encode_chain: two frames of the chain detected a scene change: in the sequential order the first one switches the detection off for the second; repeat the chain frame by frame");
			return HMR_GPU_ERR_ARG;
		}
		cut_before = cut_before || fired;
		// the frame's true starting state: what begin_frame made of the predicted one (picture order count, frame typing) on top of what the frames before really left
		FrameCtx replay;
		e->st = seq_state;
		begin_frame(s, e->st, image_types ? image_types[j] : 0, replay);
		double acc_dist;
		memcpy(&acc_dist, &g[8], 8);
		memcpy(&e->f.avg_dist, &g[10], 8);      // (what the frame's CTUs read: for an engine's later frames of the launch the launch itself set it)
		if (replay.slice_type != e->f.slice_type || replay.poc != e->f.poc || replay.avg_dist != e->f.avg_dist) {
			hmr_set_error("hmr_gpu_enc_encode_chain: frame %d started from a state the frames before it changed", j);
			return HMR_GPU_ERR_ARG;
		}
		FrameRcOut ro;
		ro.sum_qp = (int)g[4]; ro.consumed_bits = (double)g[5];
		memcpy(&ro.target_pict_size, &g[6], 8);
		end_frame(s, e->st, e->f, acc_dist, &ro);      // (the total the launch formed when the picture finished: by now a twin's frame may have overwritten the records)
		seq_state = e->st;
		// the access unit
		const int rows = s.wpp ? s.hctu : 1;
		std::vector<uint32_t> row_bytes(g + pitch - POST_MAX_ROWS, g + pitch - POST_MAX_ROWS + rows);
		size_t total = 0;
		for (int r = 0; r < rows; r++) {
			HIP_TRY(hipMemcpyAsync(e->h_bs.data() + total, e->d_bs + (size_t)r * e->row_cap, row_bytes[r], hipMemcpyDeviceToHost, bst));
			total += row_bytes[r];
		}
		HIP_TRY(hipStreamSynchronize(bst));
		if ((rc = frame_assemble(e, e->f, e->h_bs.data(), row_bytes.data(), streams[j], caps[j], &stream_bytes[j])) < 0) return rc;
	}
	return HMR_GPU_OK;
}

// HOMER_enc_encode (homer_hevc_enc_api.h:173): host planes in, access unit out
extern "C" int hmr_gpu_enc_encode(hmr_gpu_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, uint8_t *stream, long cap, long *stream_bytes,
				  uint8_t *recon)
{
	const int rc = hmr_gpu_enc_load_source(e, 0, y, u, v);
	if (rc) return rc;
	return hmr_gpu_enc_encode_source(e, 0, image_type, stream, cap, stream_bytes, recon);
}
