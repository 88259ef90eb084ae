// Context of one CTU worker and the helpers shared by the intra and inter decision code:
// neighbour lookup over the partition tree and the window consolidation copies.
#pragma once
#include "enc_prims.h"
#if !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#endif

#if !defined(__HIPCC__) && defined(HENC_TRACE_ENABLE)
#include <stdio.h>
extern "C" FILE *henc_trace_file;
#define HENC_TRACE(...) do { if (henc_trace_file) fprintf(henc_trace_file, __VA_ARGS__); } while (0)
// the whole prediction window after a call that writes it (oracle/ref_ctudump.c prints the same sums: what a later evaluation on a stale window - quirk Q12 - would see)
#define HENC_TRACE_PW(e, tag)                                                                                                                      \
	do {                                                                                                                                       \
		if (henc_trace_file) {                                                                                                             \
			unsigned a_[3] = {0, 0, 0};                                                                                                \
			for (int c_ = 0; c_ < 3; c_++) {                                                                                           \
				const int n_ = c_ ? 32 : 64;                                                                                       \
				const pred_t *p_ = c_ ? (e).w->pred_c[c_ - 1] : (e).w->pred_y;                                                     \
				for (int y_ = 0; y_ < n_; y_++)                                                                                    \
					for (int x_ = 0; x_ < n_; x_++) a_[c_] += (unsigned)(p_[y_ * n_ + x_] & 0xffff) * (unsigned)(1 + ((x_ + 3 * y_) & 7)); \
			}                                                                                                                          \
			fprintf(henc_trace_file, "PW %s ctu=%d pred=%u,%u,%u\n", tag, (e).ctu->ctu_number, a_[0], a_[1], a_[2]);                   \
		}                                                                                                                                  \
	} while (0)
#else
#define HENC_TRACE(...) do { } while (0)
#define HENC_TRACE_PW(e, tag) do { } while (0)
#endif

// Device-side phase timers (profiling build only, -DHENC_PROFILE): lane 0 accumulates s_memtime ticks per phase into Enc::prof.
enum { PF_SETUP = 0, PF_MERGE, PF_ME_INT, PF_ME_SUB, PF_PRED_INTER, PF_ENC_INTER, PF_INTRA_SEARCH, PF_INTRA_TU, PF_INTRA_CHROMA, PF_CONSOLIDATE, PF_WAIT, PF_TOTAL, PF_PRIM0, PF_COUNT = PF_PRIM0 + 2 * PP_COUNT + 2 };   // PF_PRIM0...: ticks, then calls, per primitive class (enc_prims.h)
#if defined(__HIPCC__) && defined(HENC_PROFILE)
#define HENC_PROF_T0() const unsigned long long prof_t0_ = __builtin_amdgcn_s_memtime()
#if defined(HENC_QPROF)
// (an experiment build: the phase slots 1 .. 10 are lent to the marks of one code path - HENC_QPROF_MARK adds the time since the last mark to slot `cat` - and the
// regular phase timers only keep the total)
#define HENC_PROF_ADD(e, cat) do { if ((cat) == PF_TOTAL && (e).prof && threadIdx.x == 0) (e).prof[cat] += __builtin_amdgcn_s_memtime() - prof_t0_; } while (0)
#define HENC_QPROF_T0() unsigned long long qprof_t_ = __builtin_amdgcn_s_memtime()
#define HENC_QPROF_MARK(e, cat) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((e).prof && threadIdx.x == 0) (e).prof[cat] += now_ - qprof_t_; qprof_t_ = now_; } while (0)
#define HENC_QPROF_ARG , unsigned long long &qprof_t_
#define HENC_QPROF_PASS , qprof_t_
#else
#define HENC_PROF_ADD(e, cat) do { if ((e).prof && threadIdx.x == 0) (e).prof[cat] += __builtin_amdgcn_s_memtime() - prof_t0_; } while (0)
#endif
#else
#define HENC_PROF_T0() do { } while (0)
#define HENC_PROF_ADD(e, cat) do { } while (0)
#endif
#if !defined(HENC_QPROF_MARK)
#define HENC_QPROF_T0() do { } while (0)
#define HENC_QPROF_MARK(e, cat) do { } while (0)
#define HENC_QPROF_ARG
#define HENC_QPROF_PASS
#endif

namespace henc {

// The partition geometry is the same table for every CTU of every encoder.  Device: constant memory, indexed with a value the compiler is told is wave-uniform
// (it is: the walk is group-uniform), so the fields arrive in scalar registers through the scalar cache and block addresses, sizes and loop bounds are scalar
// arithmetic.  Checker build: a plain array.
#if defined(__HIPCC__)
extern __constant__ Geo henc_geo_table[NNODES];      // defined in k_encode.hip, filled by hmr_gpu_enc_create
#endif
#if defined(__HIP_DEVICE_COMPILE__)
struct GeoTable {
	const Geo *p;
	__device__ __forceinline__ const Geo &operator[](int i) const { return henc_geo_table[__builtin_amdgcn_readfirstlane(i)]; }
	__device__ __forceinline__ const Geo &lane(int i) const { return henc_geo_table[i]; }      // an index of the lane's own (loops over the nodes of a level)
};
#else
struct GeoTable {
	const Geo *p;
	HENC_INLINE const Geo &operator[](int i) const { return p[i]; }
	HENC_INLINE const Geo &lane(int i) const { return p[i]; }
};
#endif

constexpr int NHELP_MAX = 3;
// the fixed part of a worker's LDS (k_encode.hip checks that its layout agrees)
// Of the CTU's 341 partition nodes a worker keeps the 21 of depths 0 .. 2 and the depth-3 and depth-4 nodes (16 + 64) of ONE quadrant (the 32 x 32 area of a
// depth-1 node) in its fast memory: the walks are depth first, so while a depth-1 node and what hangs below it is evaluated no deeper node of another quadrant is
// touched (the checker build verifies exactly that at every access; the one exception, the corner units of the 64 x 64 CU, is handled where it occurs); the other
// three quadrants' nodes wait in the CTU's record in HBM (nodes_select_quad).
constexpr int NODES_RESIDENT = 21, NODES_D3 = 85, NODE_QUAD_D3 = 16, NODE_QUAD_D4 = 64, NODE_SLOTS = NODES_RESIDENT + NODE_QUAD_D3 + NODE_QUAD_D4;
HENC_INLINE int node_quadrant(int idx) { return idx < NODES_D3 ? (idx - NODES_RESIDENT) >> 4 : (idx - NODES_D3) >> 6; }      // of a node of depth 3 or 4
HENC_INLINE int node_slot(int idx)
{
	return idx < NODES_RESIDENT ? idx : (idx < NODES_D3 ? NODES_RESIDENT + ((idx - NODES_RESIDENT) & (NODE_QUAD_D3 - 1)) : NODES_RESIDENT + NODE_QUAD_D3 + ((idx - NODES_D3) & (NODE_QUAD_D4 - 1)));
}
constexpr int NHELP_ = 1
#if defined(HENC_NHELP)
	+ (HENC_NHELP) - 1
#endif
	;
constexpr int HSCRATCH_ELEMS = 2048 + 144;   // int16 per helper wavefront (its scratch sits between the nodes and the sequence parameters: k_encode.hip)
constexpr int LDS_OFF_WORK = 0, LDS_OFF_NODES = (int)((sizeof(Work) + 15) & ~(size_t)15), LDS_OFF_SEQ = LDS_OFF_NODES + (int)((sizeof(Node) * NODE_SLOTS + 15) & ~(size_t)15) + NHELP_ * HSCRATCH_ELEMS * 2,
	      LDS_OFF_FRAME = LDS_OFF_SEQ + (int)((sizeof(Seq) + 15) & ~(size_t)15), LDS_OFF_BOX = LDS_OFF_SEQ + (int)((sizeof(Seq) + sizeof(FrameCtx) + 31) & ~(size_t)15);
// (LDS_OFF_RD, the RD_FULL arrays behind the mailbox: below HelperBox)
#if defined(__HIP_DEVICE_COMPILE__)
#define HENC_AT(T, OFFSET) LdsAt<T, OFFSET>
#else
#define HENC_AT(T, OFFSET) FastPtr<T>
#endif

constexpr int LDS_BOX_BYTES = 160, LDS_ENC_BYTES = 272;      // what the mailbox and a context may take (checked behind HelperBox / Enc)
constexpr int LDS_OFF_ENC = LDS_OFF_BOX + LDS_BOX_BYTES;      // the worker's context, then one per helper
constexpr int LDS_OFF_RD = LDS_OFF_ENC + (1 + NHELP_) * LDS_ENC_BYTES;
struct Enc {
	unsigned long long *prof;   // PF_COUNT accumulators of this worker (profiling build), else unused
	HENC_AT(const Seq, LDS_OFF_SEQ) seq;
	HENC_AT(const FrameCtx, LDS_OFF_FRAME) f;
	const DevTables *T;
	FastPtr<const FastTables> ft;   // the tables a TU reads, in the worker's fast memory (enc_prims.h)
	GeoTable geo;
	CtuInfo *ctus;         // all CTUs of the picture (persistent across frames)
	CtuPublic *ctu;           // the side-info record of the CTU being encoded: ctu_g's (HBM), or the worker's fast copy of it (ctu_fast != nullptr) while the CTU is encoded -
	                          // a plain pointer: FastPtr promises the compiler LDS, which is only true for the copy
	CtuInfo *ctu_g;        // its home in the picture array (logs, nodes; neighbours are ctu_g - 1, ctu_g - wctu ...)
	CtuPublic *ctu_fast;
	HENC_AT(Work, LDS_OFF_WORK) w;
	HENC_AT(WorkRd, LDS_OFF_RD) wrd;      // RD_FULL only
	int amvp_node;            // the node whose vector predictor list w.amvp holds, straight from motion estimation (-1: none)
	int on_helper;            // 0: the worker's context; 1 + h: helper wavefront h's copy (which slot of the level buffer a TU takes: enc_types.h iq_slot)
	int16_t *coeff;        // the CTU's coefficient output: 4096 luma + 2 x 1024 chroma, linear per TU in z-order
	// speculative inputs of a P-frame CTU (enc_ctu.h)
	uint32_t total_intra_partitions, total_partitions;
	int ctu_qp;               // the QP of the CTU being decided: the frame's, or what the rate control gives it (hmr_rc_get_cu_qp with qp_depth 0: computed at the CTU's root, inherited below)
	uint32_t inter_ssq[3];    // encode_inter: the squared residual of the CU per component (the no-residual distortion), when inter_ssq_valid
	int inter_ssq_valid;
	int ctu_x, ctu_y;         // CtuPublic::x / y of the CTU being encoded (here for the same reason as nb_ctus: a read of the record is a trip to HBM, and motion compensation,
	                          // the motion search and the intra reference fill started with one before they could form their first address)
	uint32_t nb_ctus;         // which neighbour CTUs exist (bit 0 left, 1 top, 2 top right, 3 top left): CtuPublic::has_*, kept here because the record lives in HBM
	unsigned long long *timeline;   // profiling build: the CTU's timestamps
	int n_spec_reads, n_ratio_cmp, last_slog;
	int n_stale_pred;         // (CtuInfo::n_stale_pred of the CTU being encoded)
	// RD_FULL (enc_rdo.h): the context states the bit estimates of this CTU copy (what the reference's et->ee holds when the CTU is decided), where the last luma
	// estimate left the shadow CTU's luma-direction pointer (-1: at the CTU's own array), the counter's chroma direction context
	const uint8_t *rd_ctx;
	int rd_luma_depth;
	uint32_t rd_chroma_state;
	HENC_AT(Node, LDS_OFF_NODES) nodes;      // the CTU's partition nodes while the CTU is encoded: the worker's fast copy (NODE_SLOTS of them, see node_of)
	Node *nodes_fast;
	int node_quad;            // the quadrant whose depth-4 nodes are in the fast copy (-1: none yet)
	// helper wavefronts of the worker (device only; nullptr = everything runs on the group itself)
	HENC_AT(struct HelperBox, LDS_OFF_BOX) box;
	FastPtr<int16_t> adi_c;              // neighbour array of a chroma block: Work::adi, or a helper's own
	FastPtr<int16_t> mc_tmp_y;           // first-stage buffer of a two-stage luma interpolation (and its row pitch): Work::sub_tmp / 72, or a helper's own
	int mc_tmp_y_stride;
	FastPtr<int16_t> mc_tmp_c;           // first-stage buffer of a two-stage chroma interpolation: Work::sub_tmp, or a helper's own
	FastPtr<int16_t> scratch_a, scratch_b;   // transform coefficients / rounding remainders of the TU in flight: Work::pred_aux / delta_u, or a helper's own
	int hseq[NHELP_MAX];
	int bgseq, bg_node;       // the helper's background intra search (bg_post ... bg_take / bg_quiesce below): jobs posted so far, the node of the one outstanding (-1: none)
};

// Two helper wavefronts per row worker take the chroma components of a step whose three components are independent (motion compensation, the
// transform chain of a TU) and single candidates of the intra mode search, while the worker itself does luma / the first candidate.  The worker
// posts a job in LDS and goes on; a helper runs the same SPMD code on its own 64 lanes with its own scratch and reports back.
enum { HJOB_NONE = 0, HJOB_NEW_CTU, HJOB_INTER_TU, HJOB_INTRA_SAD, HJOB_SYNC_CU, HJOB_SSD, HJOB_CHROMA_SEARCH, HJOB_CHROMA_TU, HJOB_QUAD_C, HJOB_QUIT };
// One helper per worker: it takes BOTH chroma planes of a step, one after the other, while the worker does luma (a workgroup of two wavefronts, so that more
// workers fit a CU: k_encode.hip).  HENC_NHELP=2 builds the round-4 arrangement, a helper per chroma plane (a third was measured then: 3 % slower).
#if !defined(HENC_NHELP)
#define HENC_NHELP 1
#endif
constexpr int NHELP = HENC_NHELP;
static_assert(NHELP == NHELP_ && (NHELP == 1 || NHELP == 2), "one helper for both chroma planes, or one per plane");
constexpr int COMP_UV = 3;   // a helper job's component argument: U, then V
struct HelperBox {
	int cmd[NHELP], done[NHELP];   // sequence numbers: helper h runs its next job when cmd[h] moves on, and sets done[h] = cmd[h] when finished
	int job[NHELP];
	int a[NHELP][8];
	uint32_t r[NHELP][8];
	// a second, background slot (one helper): the intra mode search of the CU the worker is evaluating, run between the helper's ordinary jobs
	int bg_cmd, bg_done, bg_cancel, bg_ni, bg_depth, bg_mode, bg_bits;
};
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(HelperBox) <= LDS_BOX_BYTES && sizeof(Enc) <= LDS_ENC_BYTES, "the places of the mailbox and of the contexts in a worker's LDS");
#endif

template <class G>
HENC_HD void helper_post(const G g, Enc &__restrict__ e, int h, int job, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0, int a4 = 0, int a5 = 0)
{
	HENC_ENC_IN_LDS(e);
#if defined(__HIP_DEVICE_COMPILE__)
	HelperBox *b = e.box;
	e.hseq[h]++;               // (the release store below orders everything this wavefront has written before it)
	if (g.tid == 0) {
		b->job[h] = job;
		b->a[h][0] = a0; b->a[h][1] = a1; b->a[h][2] = a2; b->a[h][3] = a3; b->a[h][4] = a4; b->a[h][5] = a5;
		__hip_atomic_store(&b->cmd[h], e.hseq[h], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
#else
	(void)g; (void)e; (void)h; (void)job; (void)a0; (void)a1; (void)a2; (void)a3; (void)a4; (void)a5;
#endif
}
template <class G>
HENC_HD void helper_wait(const G g, Enc &__restrict__ e, int h)
{
	HENC_ENC_IN_LDS(e);
#if defined(__HIP_DEVICE_COMPILE__)
	PRIM_T0();
	HelperBox *b = e.box;
	while (__hip_atomic_load(&b->done[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e.hseq[h]) __builtin_amdgcn_s_sleep(1);
	g.sync();
	PRIM_END(PP_HWAIT);
#else
	(void)g; (void)e; (void)h;
#endif
}

// The background intra search.  In the P-slice walk the intra evaluation of a CU comes after its merge evaluation, motion search and inter transform tree, and its first
// part - the mode search: neighbour arrays, thirteen predictions and SADs - reads nothing those change (the neighbours lie outside the CU; checked on every fixture with
// the one-lane build).  The worker posts it when the merge evaluation has not skipped the CU; the helper runs it between the chroma jobs of the inter evaluation (it
// looks at its ordinary slot before every candidate); the worker takes mode and bit cost when it gets to the intra evaluation, or cancels.  Device only, one helper.
// Measured: one sequence alone +5.5 % (a CTU's chain is shorter), a batch of 256 sequences -2 % (the helper runs all thirteen candidates itself, searches are started for CUs
// that never reach their intra evaluation, and with four workers per CU the helper's instructions are not free) - so a launch uses it only when it has at most one worker
// per CU: the latency kernel k_encode_pool_lat, whose walk is compiled with these hooks (WaveGrpLat, enc_platform.h).
template <class G>
HENC_HD void bg_post(const G g, Enc &__restrict__ e, int ni, int depth)
{
	HENC_ENC_IN_LDS(e);
#if defined(__HIP_DEVICE_COMPILE__)
	HelperBox *b = e.box;
	e.bgseq++;
	e.bg_node = ni;
	if (g.tid == 0) {
		b->bg_ni = ni; b->bg_depth = depth;
		__hip_atomic_store(&b->bg_cmd, e.bgseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
#else
	(void)g; (void)e; (void)ni; (void)depth;
#endif
}
// the outstanding search, if any, is told to stop and waited for (its arrays - Work::adi / adi_f - are the worker's again)
template <class G>
HENC_HD void bg_quiesce(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
#if defined(__HIP_DEVICE_COMPILE__)
	if (e.bg_node < 0) return;
	HelperBox *b = e.box;
	if (g.tid == 0) __hip_atomic_store(&b->bg_cancel, e.bgseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	while (__hip_atomic_load(&b->bg_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e.bgseq) __builtin_amdgcn_s_sleep(1);
	g.sync();
	e.bg_node = -1;
#else
	(void)g; (void)e;
#endif
}
// the result of the search posted for node ni, if there is one: true, *mode and *bits (the winner's direction and bit cost)
template <class G>
HENC_HD bool bg_take(const G g, Enc &__restrict__ e, int ni, int *mode, int *bits)
{
	HENC_ENC_IN_LDS(e);
#if defined(__HIP_DEVICE_COMPILE__)
	if (e.bg_node != ni) { bg_quiesce(g, e); return false; }
	HelperBox *b = e.box;
	while (__hip_atomic_load(&b->bg_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e.bgseq) __builtin_amdgcn_s_sleep(1);
	g.sync();
	e.bg_node = -1;
	*mode = uni(b->bg_mode);
	*bits = uni(b->bg_bits);
	return true;
#else
	(void)g; (void)e; (void)ni; (void)mode; (void)bits;
	return false;
#endif
}

// Does the group that runs the decision walk have helper wavefronts?  On the device always (every worker of k_encode_pool / k_encode_ctus has its two helpers;
// the helpers themselves only run leaf jobs: a TU chain, a SAD, a copy), on the CPU never - a constant, so that each build carries one of the two paths (as a
// run-time test of Enc::box both were compiled into the worker's code, and the decision code is several times the instruction cache).
#if defined(__HIP_DEVICE_COMPILE__)
#define HENC_HELPERS(e) true
#else
#define HENC_HELPERS(e) false
#endif

// The quadtree walks (enc_ctu.h, enc_inter.h, enc_intra.h) keep a state (0 .. 4: which child is next) and a running cost per depth and address both with the current depth.  Registers cannot be
// indexed with a run-time value: as arrays they lived in private memory, a trip to scratch per access (seen in the ISA).  DepthState packs the five states into
// one word; DepthCosts keeps five scalars and selects with compares.  Both are plain values (nothing takes their address).
struct DepthState {
	uint32_t bits = 0;
	HENC_INLINE int get(int d) const { return (int)((bits >> (4 * d)) & 15u); }
	HENC_INLINE void set(int d, int v) { bits = (bits & ~(15u << (4 * d))) | ((uint32_t)v << (4 * d)); }
	HENC_INLINE void inc(int d) { bits += 1u << (4 * d); }
};
struct DepthCosts {
	uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
	HENC_INLINE uint32_t get(int d) const { return d == 0 ? v0 : (d == 1 ? v1 : (d == 2 ? v2 : (d == 3 ? v3 : v4))); }
	HENC_INLINE void set(int d, uint32_t x) { v0 = d == 0 ? x : v0; v1 = d == 1 ? x : v1; v2 = d == 2 ? x : v2; v3 = d == 3 ? x : v3; v4 = d == 4 ? x : v4; }
	HENC_INLINE void add(int d, uint32_t x) { set(d, get(d) + x); }
};

struct DepthInts4 {      // four ints addressed by a run-time index (a chroma CU's per-partition costs)
	int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
	HENC_INLINE int get(int i) const { return i == 0 ? v0 : (i == 1 ? v1 : (i == 2 ? v2 : v3)); }
	HENC_INLINE void set(int i, int x) { v0 = i == 0 ? x : v0; v1 = i == 1 ? x : v1; v2 = i == 2 ? x : v2; v3 = i == 3 ? x : v3; }
	HENC_INLINE void add(int i, int x) { set(i, get(i) + x); }
};

HENC_INLINE int raster2abs(int r)   // raster2abs_table for the 16 x 16 unit grid (hmr_encoder_lib.c:95-100)
{
	const int x = r & 15, y = r >> 4;
	int a = 0;
	for (int b = 0; b < 4; b++) a |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
	return a;
}
HENC_INLINE int abs2raster(int a)   // abs2raster_table: the Morton de-interleave of the unit index
{
	int x = 0, y = 0;
	for (int b = 0; b < 4; b++) {
		x |= ((a >> (2 * b)) & 1) << b;
		y |= ((a >> (2 * b + 1)) & 1) << b;
	}
	return y * 16 + x;
}
HENC_INLINE Node &node_of(Enc &__restrict__ e, int idx)
{
	HENC_ENC_IN_LDS(e);
#if !defined(__HIPCC__)
	if (idx >= NODES_RESIDENT && node_quadrant(idx) != e.node_quad) {
		fprintf(stderr, "node_of: node %d of quadrant %d while quadrant %d is resident\n", idx, node_quadrant(idx), e.node_quad);
		abort();
	}
#endif
	return e.nodes[node_slot(idx)];
}
// The candidate derivations copy a CU's left-bottom / top-right flag into the depth-4 node of its corner unit (get_amvp_candidates hmr_motion_inter.c:2354-2355,
// get_merge_mvp_candidates :1990,:2020) - a lasting change of that node.  For the 64 x 64 CU the corners lie in quadrants 1 and 2, which need not be the resident one:
// the flag then goes to the CTU's record, where the quadrant is loaded from when the walk gets there.
HENC_INLINE bool node_is_resident(const Enc &__restrict__ e, int idx) { return idx < NODES_RESIDENT || node_quadrant(idx) == e.node_quad; }
HENC_INLINE void corner_set_left_bottom(Enc &__restrict__ e, int idx, uint8_t v)
{
	HENC_ENC_IN_LDS(e);
	if (node_is_resident(e, idx)) node_of(e, idx).left_bottom_nb = v;
	else e.ctu_g->nodes[idx].left_bottom_nb = v;
}
HENC_INLINE void corner_set_top_right(Enc &__restrict__ e, int idx, uint8_t v)
{
	HENC_ENC_IN_LDS(e);
	if (node_is_resident(e, idx)) node_of(e, idx).top_right_nb = v;
	else e.ctu_g->nodes[idx].top_right_nb = v;
}
HENC_INLINE int node_at(const Enc &__restrict__ e, int depth, int position) { return cfg_depth_start(depth) + position; }

// ---- neighbour partitions (hmr_arithmetic_encoding.c:229-355).  Return the CTU that holds the neighbour (nullptr when not
// available) and its z-order unit index. -------------------------------------------------------------------------------
HENC_INLINE CtuPublic *ctu_left_of(Enc &__restrict__ e) { return (e.nb_ctus & 1) ? e.ctu_g - 1 : nullptr; }
HENC_INLINE CtuPublic *ctu_top_of(Enc &__restrict__ e) { return (e.nb_ctus & 2) ? e.ctu_g - e.seq->wctu : nullptr; }
HENC_INLINE CtuPublic *ctu_top_right_of(Enc &__restrict__ e) { return (e.nb_ctus & 4) ? e.ctu_g - e.seq->wctu + 1 : nullptr; }
HENC_INLINE CtuPublic *ctu_top_left_of(Enc &__restrict__ e) { return (e.nb_ctus & 8) ? e.ctu_g - e.seq->wctu - 1 : nullptr; }

HENC_INLINE CtuPublic *pu_left(Enc &__restrict__ e, int ni, uint32_t *idx)
{
	HENC_ENC_IN_LDS(e);
	const Geo &gq = e.geo[ni];
	*idx = gq.abs_left;
	return (gq.raster_index & 15) == 0 ? ctu_left_of(e) : e.ctu;
}
// (ni: the 4 x 4 corner unit the candidate derivations ask about; has_nb: its left_bottom_nb / top_right_nb flag, which the caller has just given it)
HENC_INLINE CtuPublic *pu_left_bottom(Enc &__restrict__ e, int ni, int has_nb, uint32_t *idx)
{
	HENC_ENC_IN_LDS(e);
	const Geo &gq = e.geo[ni];
	if (!has_nb) return nullptr;
	*idx = gq.abs_left_bottom;
	if (gq.raster_index == NPART - 16) return nullptr;                     // ctu_left_bottom never exists in raster / wavefront order
	if ((gq.raster_index & 15) == 0) return ctu_left_of(e);
	if (gq.raster_index >= NPART - 16) return nullptr;
	if (gq.abs_index > gq.abs_left_bottom) return e.ctu;
	return nullptr;
}
HENC_INLINE CtuPublic *pu_top(Enc &__restrict__ e, int ni, uint32_t *idx, int planar_at_ctu_boundary)
{
	HENC_ENC_IN_LDS(e);
	const Geo &gq = e.geo[ni];
	*idx = gq.abs_top;
	if (gq.raster_index < 16) return planar_at_ctu_boundary ? nullptr : ctu_top_of(e);
	return e.ctu;
}
HENC_INLINE CtuPublic *pu_top_right(Enc &__restrict__ e, int ni, int has_nb, uint32_t *idx)
{
	HENC_ENC_IN_LDS(e);
	const Geo &gq = e.geo[ni];
	if (!has_nb) return nullptr;
	*idx = gq.abs_top_right;
	if (gq.raster_index == 15) return ctu_top_right_of(e);
	if (gq.raster_index < 16) return ctu_top_of(e);
	if ((gq.raster_index & 15) == 15) return nullptr;
	if (gq.abs_index > gq.abs_top_right) return e.ctu;
	return nullptr;
}
HENC_INLINE CtuPublic *pu_top_left(Enc &__restrict__ e, int ni, uint32_t *idx)
{
	HENC_ENC_IN_LDS(e);
	const Geo &gq = e.geo[ni];
	*idx = gq.abs_top_left;
	if (gq.raster_index == 0) return ctu_top_left_of(e);
	if (gq.raster_index < 16) return ctu_top_of(e);
	if ((gq.raster_index & 15) == 0) return ctu_left_of(e);
	return e.ctu;
}

// ---- window consolidation (hmr_motion_intra.c:844-890, hmr_motion_intra_chroma.c:29-90, hmr_mem_transfer.c:125-176) -------
// bottom row and right column of a CU, luma: what later blocks of a deeper window need as neighbours
template <class G>
HENC_HD void sync_reference_buffs(const G g, Enc &__restrict__ e, int ni, int src_wnd, int dst_wnd)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const int16_t *s = dec_ptr(*e.w, src_wnd, COMP_Y) + q.y * DEC_STRIDE_Y + q.x;
	int16_t *d = dec_ptr(*e.w, dst_wnd, COMP_Y) + q.y * DEC_STRIDE_Y + q.x;
	const int n = q.size;
	for (int i = g.tid; i < 2 * n - 1; i += g.n) {
		const int off = i < n ? (n - 1) * DEC_STRIDE_Y + i : (i - n) * DEC_STRIDE_Y + n - 1;
		d[off] = s[off];
	}
	g.sync();
	PRIM_END(PP_SYNC);
}
// the same samples into the windows first_dst .. last_dst: read once (the windows live in HBM: every copy is a trip there and back), written to each
template <class G>
HENC_HD void sync_reference_buffs_range(const G g, Enc &__restrict__ e, int ni, int src_wnd, int first_dst, int last_dst)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const int16_t *s = dec_ptr(*e.w, src_wnd, COMP_Y) + q.y * DEC_STRIDE_Y + q.x;
	const int n = q.size, base = q.y * DEC_STRIDE_Y + q.x;
	for (int i0 = g.tid; i0 < 2 * n - 1; i0 += 2 * g.n) {      // (2 n - 1 <= 127: two samples per lane of a wavefront)
		const int i1 = i0 + g.n;
		const int off0 = i0 < n ? (n - 1) * DEC_STRIDE_Y + i0 : (i0 - n) * DEC_STRIDE_Y + n - 1;
		const int off1 = i1 < n ? (n - 1) * DEC_STRIDE_Y + i1 : (i1 - n) * DEC_STRIDE_Y + n - 1;
		const bool two = i1 < 2 * n - 1;
		const int16_t v0 = s[off0], v1 = two ? s[off1] : (int16_t)0;
		for (int wnd = first_dst; wnd <= last_dst; wnd++) {
			int16_t *d = dec_ptr(*e.w, wnd, COMP_Y) + base;
			d[off0] = v0;
			if (two) d[off1] = v1;
		}
	}
	g.sync();
	PRIM_END(PP_SYNC);
}
template <class G>
HENC_HD void sync_reference_buffs_chroma(const G g, Enc &__restrict__ e, int ni, int src_wnd, int dst_wnd)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const int n = q.size_chroma;
	// (both components' samples are read before either is written: one trip to the windows, not two)
	const int16_t *su = dec_ptr(*e.w, src_wnd, COMP_U) + q.yc * DEC_STRIDE_C + q.xc, *sv = dec_ptr(*e.w, src_wnd, COMP_V) + q.yc * DEC_STRIDE_C + q.xc;
	int16_t *du = dec_ptr(*e.w, dst_wnd, COMP_U) + q.yc * DEC_STRIDE_C + q.xc, *dv = dec_ptr(*e.w, dst_wnd, COMP_V) + q.yc * DEC_STRIDE_C + q.xc;
	for (int i = g.tid; i < 2 * n - 1; i += g.n) {
		const int off = i < n ? (n - 1) * DEC_STRIDE_C + i : (i - n) * DEC_STRIDE_C + n - 1;
		const int16_t a = su[off], b = sv[off];
		du[off] = a;
		dv[off] = b;
	}
	g.sync();
	PRIM_END(PP_SYNC);
}
// whole CU: reconstruction (2-D) and levels (linear), one component
template <class G>
HENC_HD void sync_cu_comp(const G g, Enc &__restrict__ e, int ni, int q_src, int q_dst, int d_src, int d_dst, int comp)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const int n = comp == COMP_Y ? q.size : q.size_chroma, x = comp == COMP_Y ? q.x : q.xc, y = comp == COMP_Y ? q.y : q.yc;
	const int st = dec_stride(comp), off = comp == COMP_Y ? (q.abs_index << 4) : ((q.abs_index << 4) >> 2);
	const int16_t *ds = dec_ptr(*e.w, d_src, comp) + y * st + x;
	int16_t *dd = dec_ptr(*e.w, d_dst, comp) + y * st + x;
	const int16_t *qs = tq_ptr(*e.w, q_src, comp) + off;
	int16_t *qd = tq_ptr(*e.w, q_dst, comp) + off;
	const int ln = ilog2i(n);
	// four samples per lane and step (enc_prims.h), four steps in flight: the windows are in HBM and the compiler cannot move a load above a store it cannot tell
	// apart, so the loads of a batch are issued together - a 64 x 64 CU is four trips to memory instead of thirty-two
	constexpr int BATCH = 4;
	for (int i0 = g.tid * 4; i0 < n * n; i0 += g.n * 4 * BATCH) {
		S4 a[BATCH], b[BATCH];
#pragma unroll
		for (int u = 0; u < BATCH; u++) {
			const int i = i0 + u * g.n * 4;
			if (i < n * n) {
				a[u] = ld4(ds + (i >> ln) * st + (i & (n - 1)));
				b[u] = ld4(qs + i);
			}
		}
#pragma unroll
		for (int u = 0; u < BATCH; u++) {
			const int i = i0 + u * g.n * 4;
			if (i < n * n) {
				st4(dd + (i >> ln) * st + (i & (n - 1)), a[u]);
				st4(qd + i, b[u]);
			}
		}
	}
	g.sync();
	PRIM_END(PP_SYNC);
}
// both chroma components of a CU in one pass: the reads of U and V are issued together (one trip to the windows instead of two; what a helper that takes both planes does)
template <class G>
HENC_HD void sync_cu_chroma_both(const G g, Enc &__restrict__ e, int ni, int q_src, int q_dst, int d_src, int d_dst)
{
	HENC_ENC_IN_LDS(e);
	PRIM_T0();
	const Geo &q = e.geo[ni];
	const int n = q.size_chroma, st = DEC_STRIDE_C, off = (q.abs_index << 4) >> 2, ln = ilog2i(n);
	const int16_t *ds[2], *qs[2];
	int16_t *dd[2], *qd[2];
	for (int c = 0; c < 2; c++) {
		ds[c] = dec_ptr(*e.w, d_src, COMP_U + c) + q.yc * st + q.xc;
		dd[c] = dec_ptr(*e.w, d_dst, COMP_U + c) + q.yc * st + q.xc;
		qs[c] = tq_ptr(*e.w, q_src, COMP_U + c) + off;
		qd[c] = tq_ptr(*e.w, q_dst, COMP_U + c) + off;
	}
	constexpr int BATCH = 2;
	for (int i0 = g.tid * 4; i0 < n * n; i0 += g.n * 4 * BATCH) {
		S4 a[2][BATCH], b[2][BATCH];
#pragma unroll
		for (int u = 0; u < BATCH; u++) {
			const int i = i0 + u * g.n * 4;
			if (i < n * n)
				for (int c = 0; c < 2; c++) {
					a[c][u] = ld4(ds[c] + (i >> ln) * st + (i & (n - 1)));
					b[c][u] = ld4(qs[c] + i);
				}
		}
#pragma unroll
		for (int u = 0; u < BATCH; u++) {
			const int i = i0 + u * g.n * 4;
			if (i < n * n)
				for (int c = 0; c < 2; c++) {
					st4(dd[c] + (i >> ln) * st + (i & (n - 1)), a[c][u]);
					st4(qd[c] + i, b[c][u]);
				}
		}
	}
	g.sync();
	PRIM_END(PP_SYNC);
}
template <class G>
HENC_HD void sync_motion_buffers_luma(const G g, Enc &__restrict__ e, int ni, int q_src, int q_dst, int d_src, int d_dst)
{
	HENC_ENC_IN_LDS(e);
	sync_cu_comp(g, e, ni, q_src, q_dst, d_src, d_dst, COMP_Y);
}
template <class G>
HENC_HD void sync_motion_buffers_chroma(const G g, Enc &__restrict__ e, int ni, int q_src, int q_dst, int d_src, int d_dst)
{
	HENC_ENC_IN_LDS(e);
	sync_cu_comp(g, e, ni, q_src, q_dst, d_src, d_dst, COMP_U);
	sync_cu_comp(g, e, ni, q_src, q_dst, d_src, d_dst, COMP_V);
}

// both: with helper wavefronts the chroma planes are copied while the worker copies luma
template <class G>
HENC_HD void sync_motion_buffers(const G g, Enc &__restrict__ e, int ni, int q_src, int q_dst, int d_src, int d_dst)
{
	HENC_ENC_IN_LDS(e);
	if (HENC_HELPERS(e)) {
		if (NHELP >= 2) {
			helper_post(g, e, 0, HJOB_SYNC_CU, ni, COMP_U, q_src, q_dst, d_src, d_dst);
			helper_post(g, e, NHELP - 1, HJOB_SYNC_CU, ni, COMP_V, q_src, q_dst, d_src, d_dst);
		} else helper_post(g, e, 0, HJOB_SYNC_CU, ni, COMP_UV, q_src, q_dst, d_src, d_dst);
		sync_cu_comp(g, e, ni, q_src, q_dst, d_src, d_dst, COMP_Y);
		for (int h = 0; h < NHELP; h++) helper_wait(g, e, h);
		return;
	}
	sync_motion_buffers_luma(g, e, ni, q_src, q_dst, d_src, d_dst);
	sync_motion_buffers_chroma(g, e, ni, q_src, q_dst, d_src, d_dst);
}

// the depth-3 and depth-4 nodes of quadrant `quad` into the worker's fast copy (the ones that were there go back to the CTU's record first)
template <class G>
HENC_HD void nodes_quad_move(const G g, Enc &__restrict__ e, int quad, int to_record)
{
	HENC_ENC_IN_LDS(e);
	constexpr int W3 = (int)(sizeof(Node) * NODE_QUAD_D3 / 4), W4 = (int)(sizeof(Node) * NODE_QUAD_D4 / 4);
	uint32_t *f3 = (uint32_t *)&e.nodes[NODES_RESIDENT], *f4 = (uint32_t *)&e.nodes[NODES_RESIDENT + NODE_QUAD_D3];
	uint32_t *r3 = (uint32_t *)(e.ctu_g->nodes + NODES_RESIDENT + NODE_QUAD_D3 * quad), *r4 = (uint32_t *)(e.ctu_g->nodes + NODES_D3 + NODE_QUAD_D4 * quad);
	// (both pieces are read before either is written: one trip to the record)
	for (int i = g.tid; i < W3 + W4; i += g.n) {
		if (to_record) { if (i < W3) r3[i] = f3[i]; else r4[i - W3] = f4[i - W3]; }
		else { if (i < W3) f3[i] = r3[i]; else f4[i - W3] = r4[i - W3]; }
	}
	g.sync();
}
template <class G>
HENC_HD void nodes_select_quad(const G g, Enc &__restrict__ e, int quad)
{	quad = uni(quad);      // (an argument of a function of its own arrives in a vector register: uniform again, what depends on it is scalar)

	HENC_ENC_IN_LDS(e);
	if (quad == e.node_quad) return;
	g.sync();
	if (e.node_quad >= 0) nodes_quad_move(g, e, e.node_quad, 1);
	nodes_quad_move(g, e, quad, 0);
	e.node_quad = quad;
}
// the fast copy back into the CTU's record
template <class G>
HENC_HD void nodes_write_back(const G g, Enc &__restrict__ e)
{
	HENC_ENC_IN_LDS(e);
	g.sync();
	lin_copy_words(g, (const uint32_t *)&e.nodes[0], (uint32_t *)e.ctu_g->nodes, (int)(sizeof(Node) * NODES_RESIDENT / 4));
	if (e.node_quad >= 0) nodes_quad_move(g, e, e.node_quad, 1);
}

// cost helpers (hmr_common.h:53-59): the reference's macros with their operand types
HENC_INLINE double calc_mv_correction(uint32_t qp, double avg_dist) { return qp * hclip(avg_dist / 2000., .15, 1.4); }
HENC_INLINE double depth_term(double avg_dist, int depth) { return (hclip(avg_dist - 400, 40., avg_dist) / 1.75) * depth; }
HENC_INLINE double cost_rd(double avg_dist, uint32_t sum) { return hclip(avg_dist / 1.75, 0., 20000.) * sum; }

}  // namespace henc
