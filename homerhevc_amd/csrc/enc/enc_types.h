// Data model of the device-resident CTU encoder.
//
// The layout follows what the reference keeps per encoder instance, per WPP thread and per CTU, because the decision
// code reads and writes that state in ways that are part of the bitstream (which buffer a candidate's result is left in,
// what a later candidate finds there):
//   Seq       <- hvenc_engine_t configuration (hmr_private.h:1238-1367, set in HOMER_enc_control, hmr_encoder_lib.c:704-1650)
//   FrameCtx  <- slice_t / picture_t / rate_distortion_t of the frame being encoded (hmr_private.h:1012-1060, 983-990)
//   CtuInfo   <- ctu_info_t + its cu_partition_info_t list (hmr_private.h:792-843, 738-790); persistent across frames
//   Work      <- the windows and scratch buffers of henc_thread_t (hmr_private.h:1102-1228, allocated hmr_encoder_lib.c:1334-1441)
#pragma once
#include "enc_platform.h"

namespace henc {

enum { COMP_Y = 0, COMP_U = 1, COMP_V = 2, COMP_CHR = 1 };
enum { PM_INTER = 0, PM_INTRA = 1 };
enum { PART_2Nx2N = 0, PART_NxN = 3 };
enum { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2 };
enum { RDM_DIST_ONLY = 0, RDM_FULL = 1, RDM_FAST = 2 };
enum { ME_PEL = 1, ME_HALF = 2, ME_QUARTER = 4 };
enum { PLANAR_IDX = 0, DC_IDX = 1, HOR_IDX = 10, VER_IDX = 26, DM_CHROMA_IDX = 36, REG_DCT = 65535 };
enum { SCAN_ZIGZAG = 0, SCAN_HOR = 1, SCAN_VER = 2, SCAN_DIAG = 3 };

constexpr int NPART = 256;                 // 4x4 units of a 64x64 CTU (z-order)
constexpr int NNODES = 341;                // 1 + 4 + 16 + 64 + 256 partition nodes
constexpr int NDEPTH = 5;                  // MAX_PARTITION_DEPTH
constexpr int NWND = 7;                    // NUM_QUANT_WNDS / NUM_DECODED_WNDS
constexpr uint32_t MAX_COST = 0xffffffffu / 8;   // hmr_private.h:55

constexpr int MAX_SEARCH_LOGS = 192, MAX_RATIO_CMP = 96;
constexpr int POST_MAX_ROWS = 128;            // CTU rows of a picture (8192 lines)
constexpr int HENC_MAX_STEPS = 192;            // wavefront steps of a picture in the row-per-thread schedule: CTU columns + 2 x (CTU rows - 1)
constexpr int MODE_TOKEN = 0x80;           // worker mode buffers: MODE_TOKEN | depth = "whatever this unit held at depth `depth` when the CTU started"

constexpr int SEARCH_RANGE_X = 128, SEARCH_RANGE_Y = 64;   // hmr_private.h:76-77

// windows of a CTU worker.  Decoded windows hold a 128 x 128 (64 x 64 chroma) area with a one-sample frame around it
// (hmr_encoder_lib.c:1363): the CTU sits at the origin, row -1 / column -1 carry the neighbours.
constexpr int DEC_STRIDE_Y = 144, DEC_ROWS_Y = 130, DEC_ORG_Y = 1 * DEC_STRIDE_Y + 8;
constexpr int DEC_STRIDE_C = 80, DEC_ROWS_C = 66, DEC_ORG_C = 1 * DEC_STRIDE_C + 8;
constexpr int CTU_STRIDE_Y = 64, CTU_STRIDE_C = 32;

struct MV {
	int32_t x, y;
};

// static geometry of the partition tree (init_partition_info, hmr_motion_intra.c:758; get_partition_neigbours :710).  Whole words: on the device the table
// lives in constant memory and is read with scalar loads (GeoTable, enc_common.h) - everything computed from it then stays on the scalar unit.
struct Geo {
	int32_t list_index, depth, abs_index, size, size_chroma, x, y, xc, yc, num_part, raster_index;
	int32_t abs_left, abs_left_bottom, abs_top, abs_top_right, abs_top_left;
	int32_t parent;
	int32_t child[4];
};

// a motion vector as a partition node keeps it: vectors come out of a search of +-128 x +-64 samples (or are copied from a neighbour), so 16 bits hold them
struct MV16 {
	int16_t x, y;
	HENC_INLINE operator MV() const { MV m = {x, y}; return m; }
	HENC_INLINE MV16 &operator=(const MV &m) { x = (int16_t)m.x; y = (int16_t)m.y; return *this; }
};

// dynamic part of cu_partition_info_t (hmr_private.h:738-790), packed: 341 of these per CTU live in the row worker's LDS and travel to HBM and back with every CTU
struct Node {
	uint8_t left_nb, top_nb, left_bottom_nb, top_right_nb;
	uint8_t tl_inside, b_inside, r_inside, qp;
	uint32_t sum, distortion, cost;
	uint8_t prediction_mode, inter_mode, merge_flag, merge_idx, skipped;
	uint8_t intra_cbf[3], intra_tr_idx, intra_mode[3];
	uint8_t inter_cbf[3], inter_tr_idx;
	int8_t best_candidate_idx, inter_ref_index;
	MV16 inter_mv, subpix_mv, best_dif_mv;     // list 0 (P slices; B slices are outside the built configurations)
};
static_assert(sizeof(Node) == 52, "Node layout");

struct SaoOffset {                         // sao_offset_t, hmr_private.h:463-476
	int32_t mode_idc, type_idc, type_aux;
	int32_t offset[32];
};

// one logged mode search (enc_intra.h): where the left / top directions came from (0x8000 | depth << 8 | unit = inherited, else the
// direction itself), the values used, and the SAD of every direction tried
struct SearchLog {
	uint16_t src[2];
	uint8_t used[2];
	uint8_t n, has_cmp;                    // has_cmp: the evaluation ends in one logged intra / inter comparison (P-slice walk)
	uint8_t mode[14];
	uint32_t sad[14];
	uint32_t tu_cost;                      // the luma cost before the mode bits are added (encode_intra_luma)
};

// what the stages after the CTU decisions read (in-loop filters, SAO decision, entropy coding): the head of every CtuInfo
struct CtuPublic {
	uint8_t cbf[3][NPART];
	uint8_t intra_mode[2][NPART];
	uint8_t inter_mode[NPART], tr_idx[NPART], pred_depth[NPART], part_size_type[NPART], pred_mode[NPART];
	uint8_t skipped[NPART], merge[NPART], merge_idx[NPART], qp[NPART];
	int8_t mv_ref_idx[NPART];
	uint8_t mv_diff_ref_idx[NPART];
	MV mv_ref[NPART], mv_diff[NPART];
	int32_t ctu_number, x, y, last_valid_partition;
	uint32_t distortion;
	uint8_t has_left, has_top, has_top_right, has_top_left;
	uint32_t intra_parts;                  // partitions with pred_mode == INTRA after the CTU (hmr_encoder_lib.c:2924-2928)
	SaoOffset sao_recon[3], sao_coded[3];
};

struct CtuInfo : CtuPublic {
	// speculation record of the CTU (enc_sched.h): the mode searches whose candidate list used an inherited mode, and the intra / inter
	// comparisons that used the running intra ratio (intra_dist, depth term, rate term, inter cost -> outcome taken)
	int32_t n_spec_reads, n_ratio_cmp;
	SearchLog slog[MAX_SEARCH_LOGS];
	double ratio_cmp[4 * MAX_RATIO_CMP];
	uint8_t ratio_out[MAX_RATIO_CMP];
	int16_t ratio_slog[MAX_RATIO_CMP];     // the logged search behind the comparison's intra cost, or -1
	int32_t walk_intra;                    // the CTU took the intra walk (I slice, or after a scene cut)
	int32_t n_stale_pred;                  // merge candidates of this CTU that pointed outside the padded reference picture and were evaluated on what the prediction window held (Q12)
	Node nodes[NNODES];
};

// What make_seq (enc_host.h) fixes for every configuration it accepts: the decision code uses the constants (a read of Seq is a trip to LDS and keeps the
// compiler from folding the arithmetic around it); Seq keeps the fields for the host stages and as the record of what was checked.
constexpr int CFG_MAX_CU_SIZE = 64, CFG_MAX_CU_SHIFT = 6, CFG_MAX_PRED_DEPTH = 4, CFG_MAX_CU_DEPTH = 4, CFG_NUM_MERGE_CAND = 2;
HENC_INLINE constexpr int cfg_depth_start(int depth) { return ((1 << (2 * depth)) - 1) / 3; }      // 0, 1, 5, 21, 85: the first node of a depth

struct Seq {
	int32_t width, height;                 // luma picture size (multiple of the minimum CU)
	int32_t wctu, hctu, nctu;
	int32_t max_cu_size, max_cu_size_shift;
	int32_t max_pred_depth, max_intra_tr_depth, max_inter_tr_depth, max_cu_depth, mincu_mintr_shift_diff;
	int32_t min_tu_size_shift, max_tu_size_shift;
	int32_t perf_mode, perf_min_depth, perf_fast_skip, rd_mode;
	int32_t me_precision, num_merge_cand;
	int32_t sign_hiding, strong_intra, chroma_qp_offset, sao, wpp, bitrate_mode, qp;
	int32_t intra_period, gop_size, num_ref_frames, reinit_gop;
	int32_t depth_start[NDEPTH];
	// padded picture planes (reference / reconstruction): element strides and margins (hmr_encoder_lib.c:1514)
	int32_t stride_y, stride_c, margin_y, margin_c;
	// source planes (no margin)
	int32_t src_stride_y, src_stride_c;
	// elements of one padded plane (stride x (rows + 2 margins)): the phase planes of the reference (FrameCtx::sub_y / sub_c) take 16 / 64 of them
	int32_t plane_elems_y, plane_elems_c;
	int32_t wide_min_n;                    // (unused: round 3's experiment with 192 lanes on one TU chain, profiles/r03_history.md)
	int32_t pad_;
};

// what the CTU walk reads of the rate control of its frame (enc_rc.h; rate_control_t, hmr_private.h:977-990)
struct RcFrame {
	double vbv_size, vbv_fullness, target_bits_per_ctu, target_pict_size, average_pict_size;
	double sqrt_clipped_intra_period;      // hmr_rc_change_pic_mode :65 (sqrt() stays on the host)
	int32_t extra_bits, on;                // on: bitrate_mode != BR_FIXED_QP
	int32_t qp_min, is_vbr;
};

struct FrameCtx {
	int32_t slice_type, poc, qp, num_encoded_frames, is_scene_change, ref_poc;
	// scene-change detection inside a P frame (hmr_motion_inter.c:3791-3806): when the running intra share passes 70 % the remaining CTUs of the frame
	// take the intra walk.  scene_cut_allowed: the frame-level conditions hold; scene_cut_ctu: the CTU whose inter walk fired the detection (-1 none) -
	// the CTUs AFTER it are intra.  In raster order it is found on the way, under the row-parallel schedule by the verification (enc_sched.h).
	int32_t scene_cut_allowed, scene_cut_ctu;
	// lockstep = 1: the synchronous-wavefront schedule of wfpp_num_threads > 1 (enc_sched.h): "after the cut" then means a later STEP (c + 2r), not a later CTU
	int32_t lockstep, wctu;
	int32_t last_idr, pad_idr_;            // picture order count of the last IDR picture (the slice header codes the count relative to it)
	double avg_dist, lambda, sqrt_lambda, chroma_weight;
	double sao_lambda[3];
	const int16_t *src[3];                 // source picture, first sample
	const int16_t *ref[3];                 // reference picture (list 0, index 0), first valid sample
	int16_t *rec[3];                       // picture under reconstruction, first valid sample
	// the reference picture as 8-bit phase planes (k_subpel.hip), first valid sample of plane 0: luma plane (mvy & 3) * 4 + (mvx & 3) holds the picture
	// interpolated at that quarter-sample phase, chroma plane (mvy & 7) * 8 + (mvx & 7) likewise in eighth samples; the planes are row-interleaved
	// (row y of plane f starts at (y * 16 + f) * stride_y, chroma (y * 64 + f) * stride_c) and share the reference's row length and margins.  Device only (the checker build interpolates from ref[]).
	const uint8_t *sub_y, *sub_c[2];
	RcFrame rc;
};

struct MvCandList {
	int32_t num;
	MV mv[5];
	int32_t ref_idx[5];
};

// the windows a worker touches less often: seven transform / decoded windows (one per depth + the final and the auxiliary one).
// On the device they stay in HBM (they are 0.5 MB per worker); everything in Work itself is in LDS.
struct WorkSlow {
	int16_t rdec_y[64 * 64], rdec_c[2][32 * 32];
	int16_t tq_y[NWND][64 * 64], tq_c[NWND][2][32 * 32];
	int16_t dec_y[NWND][DEC_ROWS_Y * DEC_STRIDE_Y], dec_c[NWND][2][DEC_ROWS_C * DEC_STRIDE_C];
};

// Source samples of the CTU: bytes on the device (LDS is what limits how many row workers a CU holds: two fit when a worker's state stays under 80 KB),
// the reference's 16-bit width in the checker build.  Block primitives take the source operand as a template parameter.
// The prediction window likewise: what intra prediction and motion compensation write is always a sample value (0 .. 255).
#if defined(__HIPCC__)
typedef uint8_t pred_t;
#else
typedef int16_t pred_t;
#endif
#if defined(__HIPCC__)
typedef uint8_t src_t;
constexpr int TU_SCRATCH = 32 * 32;        // coefficients / remainders / levels of the TU in flight: a TU is at most 32 x 32
#else
typedef int16_t src_t;
constexpr int TU_SCRATCH = 64 * 64;        // (the checker's motion search also interpolates candidate blocks of up to 64 x 64 into pred_aux)
#endif

// scratch of the residual coder (enc_entropy.h) in the worker's fast memory: the coefficient-group scan and flags of the TU being coded
struct EntScratch {
	uint16_t cg[64];
	uint8_t cg_flag[64];
};
constexpr int RD_CTX_BYTES = 192;      // CTX_TOTAL (enc_cabac_tables.h: 179) rounded up
constexpr int RD_RING = 16;            // RD_FULL: frames whose coder states are kept - an engine's objects keep states of its last two frames (enc_rc.h RdCtxSim): 2 x 8 engines (a coder object nobody selects keeps its states: enc_rc.h RdCtxSim)

#if defined(HENC_NHELP)
#define HENC_IQ_SLOTS (HENC_NHELP)
#else
#define HENC_IQ_SLOTS 1
#endif
struct Work {
	src_t curr_y[64 * 64], curr_c[2][32 * 32];
	pred_t pred_y[64 * 64], pred_c[2][32 * 32];
#if defined(__HIPCC__)
	// levels, then dequantised coefficients, of the TU in flight: a slot for the worker (luma; V of an intra chroma TU, while its helper does U) and one for the helper
	// (U, then V of an inter TU; U of an intra chroma TU) - iq_slot below
	int16_t iq_y[32 * 32], iq_c[HENC_IQ_SLOTS][32 * 32];      // (a slot per helper wavefront: two helpers run U and V of a TU at the same time)
#else
	int16_t iq_y[64 * 64], iq_c[2][32 * 32];
#endif
	uint8_t cbf_buffs[3][NDEPTH][NPART];
	uint8_t intra_mode_buffs[2][NDEPTH][NPART];
#if defined(__HIPCC__)
	const uint8_t (*mode_in)[NDEPTH][NPART];   // what intra_mode_buffs held when the CTU started (the values behind the tokens): on the device the thread's row state in HBM itself,
	                                           // which the worker rewrites only after the CTU (k_encode.hip) - a token look-up is rare, 2.5 KB of LDS per worker are not
#else
	uint8_t mode_in[2][NDEPTH][NPART];     // what intra_mode_buffs held when the CTU started (the values behind the tokens)
#endif
	uint8_t tr_idx_buffs[NDEPTH][NPART];
	uint8_t cbf_chroma[2][NPART];
	int16_t adi[264], adi_f[264];
	int16_t pred_aux[TU_SCRATCH];          // transform coefficients of the TU in flight (checker build: between TUs also the motion search's sub-pel candidate block)
	int16_t delta_u[TU_SCRATCH];
#if !defined(__HIPCC__)
	int16_t sub_tmp[(64 + 8) * 72];        // checker build: first interpolation stage of a sub-pel candidate / two-stage motion compensation
#endif
	MvCandList amvp, merge_cands, search_cands;
	uint32_t rsplit[11];                   // performance_mode 3: cu_partition_info_t::recursive_split of the CTU's 341 nodes, one bit each (analyse_recursive_info, enc_ctu.h)
	uint32_t pad_rsplit_;
	alignas(8) int64_t srch_sads[5];       // a round of the intra mode search: its candidates and their SADs (indexed at run time: as locals they lived in private memory)
	int32_t srch_modes[6];
#if !defined(__HIPCC__)
	uint8_t nodes_fast_store[52 * (21 + 16 + 64) + 16];      // checker build: the worker's fast copy of the CTU's partition nodes (enc_common.h NODE_SLOTS; on the device a place in LDS)
#endif
	HENC_GLOBAL_PTR(WorkSlow) slow;
	// Part of the WPP thread's state next to the mode buffers: has this thread ever taken the intra walk?  The reference's thread keeps a shadow CTU whose
	// pred_mode array is set to INTRA by motion_intra_cu (hmr_motion_intra.c:1783) and never cleared; the most-probable-mode look-up of the mode search reads
	// it (homer_loop1_motion_intra :1102-1104).  With one engine every thread has been through the first (intra) frame; the threads of a second engine start
	// on a P frame and see "not intra" -> DC for neighbours inside the CTU until an intra frame or a scene change comes their way.
	int32_t thread_seen_intra;
#if !defined(__HIPCC__)
	struct WorkRd *rd_store;               // checker build: the RD_FULL arrays (on the device a place at the end of the worker's LDS, allocated only for launches that need it)
#endif
};
// RD_FULL (enc_rdo.h): the shadow CTU's own arrays, the counter's working copy of the contexts, the residual coder's scratch
struct WorkRd {
	uint8_t rd_pred_depth[NPART], rd_part_size[NPART], rd_pred_mode[NPART], rd_luma_modes[NPART];
	uint8_t rd_ctx_work[RD_CTX_BYTES];
	EntScratch rd_ent;
	// the views a bit estimate reads its CTU and the neighbours through (enc_rdo.h RdViews: pointers handed on by address, so a local would live in private
	// memory - every look at a neighbour's flag a trip through L2 for the pointer first); rd_views_of() in enc_rdo.h
	alignas(8) unsigned char rd_views[512];
};

HENC_INLINE int16_t *wnd_y(int16_t *base) { return base; }

// accessors ------------------------------------------------------------------------------------------------------
HENC_INLINE int16_t *dec_ptr(Work &w, int wnd, int comp)
{
	return (int16_t *)(comp == COMP_Y ? w.slow->dec_y[wnd] + DEC_ORG_Y : w.slow->dec_c[wnd][comp - 1] + DEC_ORG_C);
}
HENC_INLINE int dec_stride(int comp) { return comp == COMP_Y ? DEC_STRIDE_Y : DEC_STRIDE_C; }
HENC_INLINE int ctu_stride(int comp) { return comp == COMP_Y ? CTU_STRIDE_Y : CTU_STRIDE_C; }
HENC_INLINE int16_t *tq_ptr(Work &w, int wnd, int comp) { return (int16_t *)(comp == COMP_Y ? w.slow->tq_y[wnd] : w.slow->tq_c[wnd][comp - 1]); }
HENC_INLINE src_t *curr_ptr(Work &w, int comp) { return comp == COMP_Y ? w.curr_y : w.curr_c[comp - 1]; }
HENC_INLINE pred_t *pred_ptr(Work &w, int comp) { return comp == COMP_Y ? w.pred_y : w.pred_c[comp - 1]; }
HENC_INLINE int16_t *rdec_ptr(Work &w, int comp) { return (int16_t *)(comp == COMP_Y ? w.slow->rdec_y : w.slow->rdec_c[comp - 1]); }
// the TU's slot of the level / dequantised-coefficient buffer (`off`: its place in a CTU-sized buffer, which only the checker build keeps)
HENC_INLINE int16_t *iq_slot(Work &w, int comp, int off, int on_helper)
{
#if defined(__HIPCC__)
	(void)off; (void)comp;
	return on_helper ? w.iq_c[on_helper - 1] : w.iq_y;      // (on_helper: 0 = the worker, 1 + h = helper h)
#else
	(void)on_helper;
	return (comp == COMP_Y ? w.iq_y : w.iq_c[comp - 1]) + off;
#endif
}

}  // namespace henc
