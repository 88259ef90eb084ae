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
#define HENC_TU_OPERANDS_IN_LDS 1      // (enc_platform.h: the TU primitives' operands are in this kernel's LDS)
#include <stddef.h>
#include <stdlib.h>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <type_traits>
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
	int after, raster;            // raster = 1: ONE thread in raster order (wfpp_num_threads = 1 under rate control / RD_FULL: a step is one CTU, t = its number) instead of the synchronous wavefront;
	                              // ... the picture of this launch that the same engine encodes before this one (it has to be finished: the engine's persistent state), or -1
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
#if !defined(HENC_PROFILE)
static_assert(LDS_OFF_RD == (int)(LDS_BYTES - LDS_RD), "the RD_FULL arrays are the tail of a worker's LDS");
#endif
static_assert(LDS_OFF_ENC == (int)(LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU + LDS_BOX_BYTES) || LDS_FT != 0 || LDS_CTU != 0, "the RD_FULL arrays are the tail of a worker's LDS: launches without RD_FULL pictures leave them out");
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
#if !defined(HENC_PROFILE) && HENC_WAVES_PER_EU == 2 && HENC_NHELP == 1
static_assert(WORKERS_PER_CU == 4, "a worker's LDS (without the RD_FULL arrays) has to stay within a quarter of the CU's: 32 granules of 1280 bytes");
#endif
#if defined(HENC_PRINT_LDS)
static_assert(LDS_BYTES - LDS_RD == 0 && LDS_WORK == 0 && LDS_NODES == 0 && LDS_BOX == 0 && sizeof(PostScratch) == 0 && WORKERS_PER_CU == 0, "sizes");
#endif

// a helper wavefront: run the jobs the worker posts (HelperBox, enc_common.h) until it says quit
// one ordinary job of helper h (the one with sequence number seq, which has been posted); false: it was HJOB_QUIT
__device__ __forceinline__ bool helper_serve(HelperBox *box, int h, int16_t *scratch, const WaveGrp g, int seq)
{
	extern __shared__ __align__(16) uint8_t lds[];
	Enc &e = *(Enc *)(lds + LDS_OFF_ENC + (1 + h) * LDS_ENC_BYTES);      // (this helper's own context; LDS starts zeroed)
	HENC_ENC_IN_LDS(e);
	const Enc &worker = *(const Enc *)(lds + LDS_OFF_ENC);
	{
		const int job = box->job[h];
		if (job == HJOB_QUIT) return false;
		if (job == HJOB_NEW_CTU) {
			wave_copy_words(&e, &worker, (int)sizeof(Enc), g.tid);
			g.sync();
			e.box = nullptr;
			e.on_helper = 1 + h;      // (its own slot of the chroma level buffer: enc_types.h iq_slot)
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
			if (NHELP == 1 && a[1] == COMP_UV && e.geo[a[0]].size_chroma <= 8) {
				// both planes of a small TU side by side, a half of the wavefront each (enc_platform.h PairGrp): one after the other they made the helper the
				// slower side of every 8 x 8 and 16 x 16 CU (two chains against the worker's one)
				const PairGrp pg{g.tid & 31, g.tid >> 5};
				const uint32_t d = encode_inter_tu(pg, e, a[0], COMP_U + pg.half, a[2], a[3], &sum, &raw, pg.half * 512);
				r0 = (uint32_t)__builtin_amdgcn_readlane((int)d, 0); r1 = (uint32_t)__builtin_amdgcn_readlane(sum, 0); r2 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 0);
				r3 = (uint32_t)__builtin_amdgcn_readlane((int)d, 32); r4 = (uint32_t)__builtin_amdgcn_readlane(sum, 32); r5 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 32);
				break;
			}
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
#if defined(HENC_QUAD)
		case HJOB_QUAD_C:      // the chroma blocks of all merge candidates of an 8 x 8 or 16 x 16 CU in one pass (enc_quad.h)
			quad_chroma_job(g, e, a[0], a[1]);
			break;
#endif
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
	return true;
}

// (the latency kernel's copy, out of line: its loop and its background search both call it)
__device__ __attribute__((noinline)) bool helper_serve_out(HelperBox *box, int h, int16_t *scratch, const WaveGrp g, int seq) { return helper_serve(box, h, scratch, g, seq); }

// The background slot (enc_common.h bg_post): the intra mode search of node bg_ni - homer_loop1_motion_intra, as intra_mode_search runs it, without the search log (the
// pool's schedules do not read it) - every candidate on this wavefront, an ordinary job served before each of them.  false: HJOB_QUIT was among those.
__device__ __attribute__((noinline)) bool helper_bg_search(HelperBox *box, int h, int16_t *scratch, const WaveGrp g, int &seq, int bgseq)
{
	extern __shared__ __align__(16) uint8_t lds[];
	Enc &e = *(Enc *)(lds + LDS_OFF_ENC + (1 + h) * LDS_ENC_BYTES);
	HENC_ENC_IN_LDS(e);
	const int ni = uni(box->bg_ni), depth = uni(box->bg_depth);
	const Geo &q = e.geo[ni];
	const int n = q.size, curr_depth = q.depth, inv_depth = CFG_MAX_CU_SHIFT - curr_depth;
	bool quit = false;
	int best_mode = 0, bits = -1;
	double best_cost = 0;
	if (__hip_atomic_load(&box->bg_cancel, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != bgseq) {
		node_fill_refs(g, e, ni, depth + 1, COMP_Y, 1);
		int preds[3], dirs[2];
		uint16_t src[2];
		intra_neighbour_dirs(e, ni, curr_depth, dirs, src);
		mpm_from_dirs(dirs[0], dirs[1], preds);
		const int rd_fast = e.seq->rd_mode == RDM_FAST ? 1 : 0;
		const src_t *orig = e.w->curr_y + q.y * CTU_STRIDE_Y + q.x;
		bits = intra_search_walk(preds, rd_fast, e.f->sqrt_lambda, [&](int mode) -> int64_t {
			if (__hip_atomic_load(&box->cmd[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == seq) {
				if (!helper_serve_out(box, h, scratch, g, seq)) { quit = true; return -1; }
				seq++;
			}
			if (__hip_atomic_load(&box->bg_cancel, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == bgseq) return -1;
			return (int64_t)intra_predict_sad(g, (pred_t *)nullptr, 0, orig, CTU_STRIDE_Y, intra_is_filtered(mode, inv_depth) ? e.w->adi_f : e.w->adi, n, mode, 1);
		}, &best_mode, &best_cost);
	}
	g.sync();
	if (g.tid == 0) {
		box->bg_mode = best_mode;
		box->bg_bits = bits;
		__hip_atomic_store(&box->bg_done, bgseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	return !quit;
}

// LAT: the kernel of launches with at most one worker per CU (k_encode_pool_lat), whose helpers also run the background search; the throughput kernel keeps the plain
// loop (the out-of-line dispatch and the second poll cost a batch 8 % when both kernels shared one loop - the code the helper wavefronts run counts: two CUs share an
// instruction cache)
template <bool LAT>
__device__ __attribute__((noinline)) void helper_loop(HelperBox *box, int h, int16_t *scratch)      // (out of line, as it was before it became a template: inlined it costs the kernel 38 spilled registers)
{
	WaveGrp g{(int)(threadIdx.x & 63)};
	if constexpr (!LAT) {
		for (int seq = 1;; seq++) {
			while (__hip_atomic_load(&box->cmd[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);   // (two workgroups share a CU now: a helper that spins takes issue cycles from the other workgroup's worker on its SIMD)
			if (!helper_serve(box, h, scratch, g, seq)) return;
		}
	}
	int seq = 1, bgseq = 1;
	for (;;) {
		if (__hip_atomic_load(&box->cmd[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == seq) {
			if (!helper_serve_out(box, h, scratch, g, seq)) return;
			seq++;
			continue;
		}
		if (NHELP == 1 && __hip_atomic_load(&box->bg_cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == bgseq) {
			if (!helper_bg_search(box, h, scratch, g, seq, bgseq)) return;
			bgseq++;
			continue;
		}
		__builtin_amdgcn_s_sleep(1);   // (two workgroups share a CU now: a helper that spins takes issue cycles from the other workgroup's worker on its SIMD)
	}
}

// A workgroup is a row worker (wavefront 0) and its two helpers.  rows_enter sets the mailbox up and sends the helper wavefronts into their service loop; it
// returns true on the worker only.  release_helpers lets them go (without it the workgroup never ends).
template <bool LAT = false>
__device__ __forceinline__ bool rows_enter(unsigned lds_bytes)
{
	extern __shared__ __align__(16) uint8_t lds[];
	HelperBox *box = (HelperBox *)(lds + LDS_WORK + LDS_NODES + LDS_GEO + LDS_SEQ + LDS_CTU);
	const int wave = (int)(threadIdx.x >> 6);
	// LDS keeps what the last workgroup on this CU left in it: everything starts from zero (the helpers' scratch included), whatever ran here before
	for (int i = (int)threadIdx.x; i < (int)(lds_bytes / 4); i += ENC_THREADS) ((uint32_t *)lds)[i] = 0;
	__syncthreads();
	if (wave > 0) {
		helper_loop<LAT>(box, wave - 1, (int16_t *)(lds + LDS_WORK + LDS_NODES) + (wave - 1) * HSCRATCH_ELEMS);
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
	e.bgseq = 0;
	e.bg_node = -1;
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
constexpr int BATCH_MAX = 512;      // pictures of one launch (hmr_gpu_enc_encode_batch: sequences per call); the pool's state rows and the batch's staging arrays are sized for it

// rows of step t: r_lo .. r_hi (empty when r_lo > r_hi)
__device__ __forceinline__ void pool_step_rows(int t, int W, int H, int *r_lo, int *r_hi)
{
	const int lo = t - W + 1;
	*r_lo = lo <= 0 ? 0 : (lo + 1) >> 1;
	*r_hi = (t >> 1) < H - 1 ? (t >> 1) : H - 1;
}

template <class G>
__device__ void pool_encode_ctu(const EncDev &d, Enc &__restrict__ e, const G g, Seq *lseq, FrameCtx *lframe, FastTables *lft, int t, int row, int *cached_rem, const int *abort_flag)
{
	HENC_ENC_IN_LDS(e);
	const Seq &S = *lseq;
	const int W = S.wctu, H = S.hctu;
	const int raster = d.raster;
	const int T = d.threads, me = row % T, c = raster ? t - row * W : t - 2 * row, n = row * W + c;
	// counters as of the end of step t - 1
	uint32_t ti = 0, tc = 0;
	for (int r2 = g.tid; r2 < H; r2 += 64) {
		const int have = raster ? (r2 < row ? W : (r2 == row ? c : 0)) : (t - 2 * r2 < 0 ? 0 : (t - 2 * r2 < W ? t - 2 * r2 : W));      // (raster order: as of CTU n - 1)
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
	const int hrow = raster ? row : t / (2 * T) * T;   // thread 0's row that is inside the picture at this step, if any
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
		const int is_sc = cut >= 0 && (raster ? n >= cut : t >= cut % W + 2 * (cut / W));
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
	if (g.tid == 0) post_st_fenced(&d.post.rows[row].dec, c + 1);
}

// post-decision tasks of picture q that are ready, on this worker (its Work area is scratch between two CTUs); *finished counts the pictures whose last task is done
__device__ __attribute__((noinline)) int pool_post_drain_run(const EncDev &d, const WaveGrp g, const Seq *lseq, const FrameCtx *lframe, Work *lw, WorkSlow *my_slow, int *finished);
// ... the look first, in line: the function that runs the tasks saves forty registers to private memory when it is entered, and an idle worker asks every picture of the
// launch - mostly to learn that nothing is ready
__device__ __forceinline__ int pool_post_drain(const EncDev &d, const WaveGrp g, const Seq *lseq, const FrameCtx *lframe, Work *lw, WorkSlow *my_slow, int *finished)
{
	if (post_finished(*lseq, d.post)) return 0;
	PostCtx x;
	x.seq = lseq; x.f = lframe; x.T = d.tables; x.geo.p = nullptr; x.ctus = d.ctus; x.coeff = d.coeff; x.pic = &d.post;
	int r, c, k;
	if (!post_scan(g, x, r, c, k)) return 0;
	return pool_post_drain_run(d, g, lseq, lframe, lw, my_slow, finished);
}
__device__ __attribute__((noinline)) int pool_post_drain_run(const EncDev &d, const WaveGrp g, const Seq *lseq, const FrameCtx *lframe, Work *lw, WorkSlow *my_slow, int *finished)
{
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

template <bool LAT>
__device__ __forceinline__ void encode_pool_body(const EncDev *devs, int nseq, int *state, int *finished, WorkSlow *slow, unsigned long long watchdog_ticks, unsigned lds_bytes)
{
	// finished[0]: pictures whose last task is done; finished[1]: abort - a worker has waited longer than the watchdog allows (a faulted or starved peer): everybody
	// leaves and the host reports an error instead of the launch hanging
	unsigned long long t_start = wall_clock64();      // when this worker last had something to do (the watchdog's clock)
	if (!rows_enter<LAT>(lds_bytes)) return;
	extern __shared__ __align__(16) uint8_t lds[];
	typename std::conditional<LAT, WaveGrpLat, WaveGrp>::type g{(int)(threadIdx.x & 63)};      // (the latency kernel's walk is an instantiation of its own: enc_platform.h)
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
	e.bgseq = 0;
	e.bg_node = -1;
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
				const int W = dd.seq->wctu, H = dd.seq->hctu, steps = dd.raster ? W * H : W + 2 * (H - 1);
				ct = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (ct < steps) {
					all_finished = false;
					if (dd.raster) open = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ct;      // (a step is one CTU: st[1] = CTUs handed out)
					else {
						int r_lo, r_hi;
						pool_step_rows(ct, W, H, &r_lo, &r_hi);
						open = __hip_atomic_load(&st[1 + ct], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r_hi - r_lo + 1;
					}
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
				const EncDev &dd = devs[qq];
				if (dd.raster) {
					if (g.tid == 0) kk = atomicCAS(&state[(size_t)qq * POOL_STRIDE + 1], tt, tt + 1) == tt ? 0 : 1;
					kk = __builtin_amdgcn_readfirstlane(kk);
					if (kk == 0) { q = qq; t = tt; k = 0; }
					else base -= 64;
					continue;
				}
				if (g.tid == 0) kk = atomicAdd(&state[(size_t)qq * POOL_STRIDE + 1 + tt], 1);
				kk = __builtin_amdgcn_readfirstlane(kk);
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
		const EncDev &d = devs[q];      // (by reference: a copy of the 480-byte descriptor is thirty stores to private memory per CTU, and every d.member a load from there)
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
		if (d.raster) row = t / W;
		else if (hvalid) row = k == 0 ? hrow : (r_lo + k - 1 < hrow ? r_lo + k - 1 : r_lo + k);
		else row = r_lo + k;
		pool_encode_ctu(d, e, g, lseq, lframe, lft, t, row, cached_rem, finished + 1);
		// close the step when this was its last CTU (the CTU's results were published by the release fence at the end of pool_encode_ctu)
		if (g.tid == 0) {
			int *st = state + (size_t)q * POOL_STRIDE;
			const int dn = d.raster ? 1 : atomicAdd(&st[1 + POOL_MAX_STEPS + t], 1) + 1;
			if (dn == (d.raster ? 1 : r_hi - r_lo + 1)) {
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
// the throughput kernel (many pictures, four workers per CU) and the latency kernel (at most one worker per CU: the helpers also run the background intra search)
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(HENC_WAVES_PER_EU))) void k_encode_pool(const EncDev *devs, int nseq, int *state, int *finished, WorkSlow *slow, unsigned long long watchdog_ticks, unsigned lds_bytes)
{
	encode_pool_body<false>(devs, nseq, state, finished, slow, watchdog_ticks, lds_bytes);
}
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(HENC_WAVES_PER_EU))) void k_encode_pool_lat(const EncDev *devs, int nseq, int *state, int *finished, WorkSlow *slow, unsigned long long watchdog_ticks, unsigned lds_bytes)
{
	encode_pool_body<true>(devs, nseq, state, finished, slow, watchdog_ticks, lds_bytes);
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

// the host side: the encoder object and the per-frame entry points, then the calls that put several pictures into one launch
#include "k_encode_object.inc"
#include "k_encode_batch.inc"
