/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * The CTU encoder core (homerhevc_amd/csrc/enc/, the code the gfx950 kernels are compiled from) instantiated with a one-lane
 * group and driven CTU by CTU in raster order - the order of the reference with wfpp_num_threads = 1.  It exists so that the
 * decision logic can be diffed against the compiled reference (oracle/_ref/ref_ctudump) in the build container, where there
 * is no GPU, and so that the CPU tests can check the host logic.  Only tests/ and tools/ load this library; libhomer_gpu.so
 * never contains this instantiation.
 *
 * Records use the layout of oracle/ref_ctudump.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define HENC_TRACE_ENABLE 1
#define HENC_SAO_TRACE 1
#define HENC_SCHED_CAUSES 1
#include <stdio.h>
static FILE *henc_sao_trace_file = nullptr;
#include <stddef.h>
#include "../homerhevc_amd/csrc/enc/enc_sched.h"
#include "../homerhevc_amd/csrc/enc/enc_host.h"
#include "../homerhevc_amd/csrc/enc/enc_post.h"
#include "hmr_oracle.h"

extern "C" FILE *henc_trace_file = nullptr;
extern "C" int henc_sched_causes[8] = {0};
const DevTables *hmr_host_tables();

using namespace henc;

namespace {

constexpr int REC_BYTES = 32 + 3 * 256 + 2 * 256 + 9 * 256 + 256 + 256 + 2048 + 2048 + 6144 * 2 + 6144 * 2 + 2 * 5 * 256;

struct Cpu {
	HostCfg cfg;
	Seq seq;
	HostState st;
	FrameCtx f;
	Geo geo[NNODES];
	std::vector<CtuInfo> ctus;
	Work *w;
	std::vector<Work *> row_w;   // wavefront emulation: one worker per CTU row
	// engines (enc_host.h): the persistent per-engine state - CTU records, worker(s) - of the engines that are not the active one
	struct EngState { std::vector<CtuInfo> ctus; Work *w = nullptr; std::vector<Work *> row_w; } eng[MAX_ENGINES];
	int active = 0, local_engines = 1;
	std::vector<int16_t> src[3], pic[2][3], coeff;
	// post-decision stage (enc_post.h): the reconstruction before the loop filters, the deblocked picture, unit arrays, row state, sub-streams
	std::vector<int16_t> rec[3], dbk[3], u_mvx, u_mvy;
	std::vector<int8_t> u_ref;
	std::vector<uint8_t> u_qp, u_flags, bs;
	std::vector<PostRow> rows;
	std::vector<RowEnt> ent;
	std::vector<uint32_t> cumbits;
	std::vector<double> sao_lambda;
	std::vector<uint16_t> rc_need;   // rate control: CTUs of each row coded when a wavefront step (sched 2) / a CTU (raster order) starts
	int post_errors[2] = {0, 0};
	// RD_FULL: which coder states each CTU's bit estimates copy (enc_rc.h RdCtxSim), the states after every coded CTU of this frame and of the one before
	RdCtxSim rdsim[MAX_ENGINES];      // (an engine's coder objects are its own: hmr_encoder_lib.c:1096)
	std::vector<RdCtxVersion> rdsrc;
	std::vector<uint8_t> ctx_after[RD_RING];      // (frame f in slot f mod RD_RING)
	uint8_t zero_ctx[RD_CTX_BYTES] = {0}, init_ctx[RD_RING][RD_CTX_BYTES] = {{0}};      // (init_ctx: the slice's initial states of the frames)
	PostPic post = {};
	PostScratch *scratch = nullptr;
	std::vector<uint8_t> records;
	int cur = 0;        // picture under reconstruction: pic[cur], reference: pic[cur ^ 1]
	uint32_t acc_dist = 0;
	uint32_t intra_parts = 0, total_parts = 0;
	EntropyState es;
	FastTables ft;      // the worker's copy of the tables a TU reads (enc_prims.h), refreshed per frame
	// speculative row-parallel schedule (enc_sched.h), emulated: sched = 0 raster order, 1 wavefront with guesses + verification
	int sched = 0, row_guess = 0;          // row_guess: 0 = the truth of the previous frame, 1 = the row above after its second CTU
	std::vector<CtuInfo> ctus_start;
	std::vector<uint8_t> guess, truth, outtok, valid, dirty;
	std::vector<uint32_t> intra_before, used_intra, used_parts, prefix;
	std::vector<uint64_t> hash;
	uint8_t chain_start[MODE_STATE_BYTES] = {0}, chain_end[MODE_STATE_BYTES] = {0};
	int stat_passes = 0, stat_encodes = 0, stat_invalid_first = 0;
};

Work *new_work()
{
	Work *w = (Work *)calloc(1, sizeof(Work));
	w->rd_store = (WorkRd *)calloc(1, sizeof(WorkRd));
	w->slow = (WorkSlow *)calloc(1, sizeof(WorkSlow));
	return w;
}

void activate_engine(Cpu &c, int k)
{
	if (k == c.active) return;
	Cpu::EngState &out = c.eng[c.active], &in = c.eng[k];
	std::swap(c.ctus, out.ctus); std::swap(c.w, out.w); std::swap(c.row_w, out.row_w);      // park the active engine's state
	std::swap(c.ctus, in.ctus); std::swap(c.w, in.w); std::swap(c.row_w, in.row_w);
	c.active = k;
}

int16_t *plane0(Cpu &c, int which, int comp)
{
	const Seq &s = c.seq;
	const int st = comp ? s.stride_c : s.stride_y, m = comp ? s.margin_c : s.margin_y;
	return c.pic[which][comp].data() + (size_t)m * st + m;
}

int16_t *plane0_of(Cpu &c, std::vector<int16_t> *planes, int comp)
{
	const Seq &s = c.seq;
	const int st = comp ? s.stride_c : s.stride_y, m = comp ? s.margin_c : s.margin_y;
	return planes[comp].data() + (size_t)m * st + m;
}

// the post-decision stage of the frame that starts: counters, sub-streams, where its pictures are
void post_begin_frame(Cpu &c)
{
	const Seq &s = c.seq;
	const int us = s.wctu * 16, uh = s.hctu * 16;
	if (c.rows.empty()) {
		for (int k = 0; k < 3; k++) { c.rec[k].assign(c.pic[0][k].size(), 0); c.dbk[k].assign(c.pic[0][k].size(), 0); }
		c.u_mvx.assign((size_t)us * uh, 0); c.u_mvy.assign((size_t)us * uh, 0); c.u_ref.assign((size_t)us * uh, 0); c.u_qp.assign((size_t)us * uh, 0); c.u_flags.assign((size_t)us * uh, 0);
		c.rows.resize(s.hctu); c.ent.resize(s.hctu); c.cumbits.assign(s.nctu, 0);
		c.sao_lambda.assign(52 * 2, 0);
		c.scratch = (PostScratch *)calloc(1, sizeof(PostScratch));
	}
	const int row_cap = s.wctu * 24576;
	c.bs.assign((size_t)row_cap * s.hctu, 0);
	memset((void *)c.rows.data(), 0, sizeof(PostRow) * s.hctu);
	memset((void *)c.ent.data(), 0, sizeof(RowEnt) * s.hctu);
	sao_lambda_table(s, c.f.slice_type, c.sao_lambda.data());
	PostPic &P = c.post;
	for (int k = 0; k < 3; k++) { P.dbk[k] = plane0_of(c, c.dbk, k); P.fin[k] = plane0(c, c.cur, k); }
	P.units_stride = us;
	P.mvx = c.u_mvx.data(); P.mvy = c.u_mvy.data(); P.ref = c.u_ref.data(); P.uqp = c.u_qp.data(); P.flags = c.u_flags.data();
	P.rows = c.rows.data(); P.ent = c.ent.data(); P.bs = c.bs.data(); P.row_cap = row_cap; P.cumbits = c.cumbits.data();
	P.planes[0] = P.planes[1] = P.planes[2] = nullptr; P.prof = nullptr;
	P.sao_lambda = c.sao_lambda.data(); P.errors = c.post_errors; P.rc_need = c.rc_need.empty() ? nullptr : c.rc_need.data();
	P.ctx_after = nullptr;
	if (s.rd_mode == RDM_FULL) {
		const int slot = c.f.num_encoded_frames % RD_RING;
		c.ctx_after[slot].assign((size_t)s.nctu * RD_CTX_BYTES, 0);
		P.ctx_after = c.ctx_after[slot].data();
		c.rdsim[c.local_engines > 1 ? c.f.num_encoded_frames % c.local_engines : 0].frame(c.f.num_encoded_frames, c.rdsrc, c.sched != 2);      // (one thread: raster order)
		for (int i = 0; i < CTX_TOTAL; i++) c.init_ctx[slot][i] = Cabac::init_state(c.f.slice_type, c.f.qp, i);
	}
}
// RD_FULL: the states CTU n's bit estimates copy
const uint8_t *rd_ctx_for(Cpu &c, int n)
{
	const Seq &s = c.seq;
	if (s.rd_mode != RDM_FULL) return nullptr;
	const RdCtxVersion v = c.rdsrc[n];
	if (v.frame < 0) return c.zero_ctx;
	if (v.frame > c.f.num_encoded_frames || v.frame <= c.f.num_encoded_frames - RD_RING) { fprintf(stderr, "RD_FULL: CTU %d copies coder states of frame %d in frame %d\n", n, v.frame, c.f.num_encoded_frames); abort(); }
	if (v.k == 0) return c.init_ctx[v.frame % RD_RING];
	const int idx = v.row * s.wctu + v.k - 1;
	if (v.frame == c.f.num_encoded_frames && c.rows[v.row].p_done < v.k) { fprintf(stderr, "RD_FULL: CTU %d starts before CTU (%d, %d) is coded\n", n, v.row, v.k - 1); abort(); }
	return c.ctx_after[v.frame % RD_RING].data() + (size_t)idx * RD_CTX_BYTES;
}
PostCtx post_ctx(Cpu &c)
{
	PostCtx x;
	x.seq = &c.seq; x.f = &c.f; x.T = hmr_host_tables(); x.geo.p = c.geo; x.ctus = c.ctus.data(); x.coeff = c.coeff.data(); x.pic = &c.post;
	return x;
}
// CTU n has been decided: whatever of the post-decision stage can run now, runs
void post_after_ctu(Cpu &c, int n)
{
	const Seq &s = c.seq;
	c.rows[n / s.wctu].dec = n % s.wctu + 1;
	post_drain(CpuGrp(), post_ctx(c), *c.scratch);
}
// hmr_rc_get_cu_qp for the CTUs whose decisions have index k (the wavefront step / the CTU in raster order): the slice QP, or what the rate control makes of the
// CTUs coded so far.  is_sc: the frame has been found to be a new scene by the time the CTU computes its QP
int ctu_qp_for(Cpu &c, int k, int is_sc)
{
	if (!c.f.rc.on) return c.f.qp;
	const Seq &s = c.seq;
	if (!rc_ready(CpuGrp(), c.post, s.hctu, k)) { fprintf(stderr, "rate control: decisions %d start before the CTUs they read the bits of are coded\n", k); abort(); }
	uint32_t bits = 0;
	int ctus = 0;
	rc_consumed(CpuGrp(), c.post, s.wctu, s.hctu, k, &bits, &ctus);
	return rc_calc_cu_qp(c.f.rc, (double)bits, ctus, c.f.slice_type, is_sc, s.reinit_gop, s.intra_period, c.f.avg_dist, c.f.num_encoded_frames);
}
// a scene change has just been detected by the decisions with index k: hmr_rc_change_pic_mode
void rc_scene_change(Cpu &c, int k)
{
	if (!c.f.rc.on) return;
	const Seq &s = c.seq;
	uint32_t bits = 0;
	int ctus = 0;
	rc_consumed(CpuGrp(), c.post, s.wctu, s.hctu, k, &bits, &ctus);
	rc_change_pic_mode(c.f.rc, s.reinit_gop, s.intra_period, s.nctu, c.f.rc.sqrt_clipped_intra_period, bits, ctus);
}
FrameRcOut rc_frame_out(Cpu &c)
{
	const Seq &s = c.seq;
	FrameRcOut ro = {0, 0.0, c.f.rc.target_pict_size};
	for (int n = 0; n < s.nctu; n++) ro.sum_qp += c.ctus[n].nodes[0].qp;
	for (int r = 0; r < s.hctu; r++) ro.consumed_bits += c.cumbits[(size_t)r * s.wctu + s.wctu - 1];
	return ro;
}

// every CTU has been decided (a schedule that re-encodes CTUs: the stage runs when the decisions are final)
void post_whole_frame(Cpu &c)
{
	const Seq &s = c.seq;
	for (int r = 0; r < s.hctu; r++) c.rows[r].dec = s.wctu;
	post_drain(CpuGrp(), post_ctx(c), *c.scratch);
}

void pad_plane(int16_t *p, int stride, int w, int h, int m)
{
	for (int y = 0; y < h; y++) {
		for (int x = 1; x <= m; x++) {
			p[y * stride - x] = p[y * stride];
			p[y * stride + w - 1 + x] = p[y * stride + w - 1];
		}
	}
	for (int y = 1; y <= m; y++) {
		memcpy(p - y * stride - m, p - m, sizeof(int16_t) * (w + 2 * m));
		memcpy(p + (h - 1 + y) * stride - m, p + (h - 1) * stride - m, sizeof(int16_t) * (w + 2 * m));
	}
}

void make_record(Cpu &c, int n, Enc &e)
{
	uint8_t *o = c.records.data() + (size_t)n * REC_BYTES;
	const CtuInfo &ci = c.ctus[n];
	const Work &w = *c.w;
	int32_t hdr[8] = {0x43545544, c.f.num_encoded_frames, n, c.f.slice_type, (int32_t)ci.nodes[0].cost, (int32_t)ci.nodes[0].distortion, (int32_t)ci.nodes[0].sum,
			  c.f.scene_cut_ctu >= 0 && n >= c.f.scene_cut_ctu};
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
	memcpy(o, w.slow->tq_y[0], 8192); o += 8192;
	memcpy(o, w.slow->tq_c[0][0], 2048); o += 2048;
	memcpy(o, w.slow->tq_c[0][1], 2048); o += 2048;
	for (int comp = 0; comp < 3; comp++) {
		const int nn = comp ? 32 : 64;
		const int16_t *d = comp ? w.slow->dec_c[0][comp - 1] + DEC_ORG_C : w.slow->dec_y[0] + DEC_ORG_Y;
		for (int y = 0; y < nn; y++) { memcpy(o, d + y * dec_stride(comp), nn * 2); o += nn * 2; }
	}
	for (int k = 0; k < 2; k++)
		for (int d = 0; d < 5; d++) { memcpy(o, w.intra_mode_buffs[k][d], 256); o += 256; }
	(void)e;
}

// ---- the row-parallel schedule of the device (k_encode.hip), emulated with one lane: same passes, same guesses, same checks ----------
void record_from_outputs(Cpu &c, int n, const uint8_t *state_after)
{
	const Seq &s = c.seq;
	uint8_t *o = c.records.data() + (size_t)n * REC_BYTES;
	const CtuInfo &ci = c.ctus[n];
	memset(o, 0, REC_BYTES);
	int32_t hdr[8] = {0x43545544, c.f.num_encoded_frames, n, c.f.slice_type, (int32_t)ci.nodes[0].cost, (int32_t)ci.nodes[0].distortion, (int32_t)ci.nodes[0].sum,
			  c.f.scene_cut_ctu >= 0 && n >= c.f.scene_cut_ctu};
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
	memcpy(o, c.coeff.data() + (size_t)n * 6144, 12288); o += 12288;
	for (int comp = 0; comp < 3; comp++) {
		const int nn = comp ? 32 : 64, px = ci.x >> (comp ? 1 : 0), py = ci.y >> (comp ? 1 : 0);
		const int pw = comp ? s.width / 2 : s.width, ph = comp ? s.height / 2 : s.height, rs = comp ? s.stride_c : s.stride_y;
		const int16_t *p = plane0_of(c, c.rec, comp);
		for (int yy = 0; yy < nn; yy++) {
			if (py + yy < ph) memcpy(o, p + (size_t)(py + yy) * rs + px, (px + nn <= pw ? nn : pw - px) * 2);
			o += nn * 2;
		}
	}
	memcpy(o, state_after, MODE_STATE_BYTES);
}

void sched_verify(Cpu &c, int *n_invalid)
{
	const Seq &s = c.seq;
	CpuGrp g;
	uint8_t st[MODE_STATE_BYTES];
	memcpy(st, c.chain_start, MODE_STATE_BYTES);
	uint32_t ib = 0;
	int cut = -1;
	for (int n = 0; n < s.nctu; n++) {
		memcpy(&c.truth[(size_t)n * MODE_STATE_BYTES], st, MODE_STATE_BYTES);
		for (int k = 0; k < NPART; k++) {
			uint8_t col[2][NDEPTH];
			for (int comp = 0; comp < 2; comp++)
				for (int d = 0; d < NDEPTH; d++) col[comp][d] = st[(comp * NDEPTH + d) * NPART + k];
			sched_chain_step(col, &c.outtok[(size_t)n * MODE_STATE_BYTES], k);
			for (int comp = 0; comp < 2; comp++)
				for (int d = 0; d < NDEPTH; d++) st[(comp * NDEPTH + d) * NPART + k] = col[comp][d];
		}
		c.intra_before[n] = ib;
		if (cut < 0 && c.f.slice_type == SLICE_P && scene_cut_fires(s, c.f, ib, (uint32_t)n * NPART)) cut = n;
		ib += c.ctus[n].intra_parts;
	}
	c.f.scene_cut_ctu = cut;      // where the single thread would have detected a scene change, given what the CTUs before it look like now
	memcpy(c.chain_end, st, MODE_STATE_BYTES);
	const int uses_ratio = c.f.slice_type != SLICE_I;
	int bad = 0;
	for (int n = 0; n < s.nctu; n++) {
		c.valid[n] = (uint8_t)sched_guesses_hold(g, c.ctus[n], c.f, &c.truth[(size_t)n * MODE_STATE_BYTES], &c.guess[(size_t)n * MODE_STATE_BYTES], c.intra_before[n],
							 (uint32_t)n * NPART, c.used_intra[n], c.used_parts[n], uses_ratio);
		bad += !c.valid[n];
		if (getenv("HENC_SCHED_DEBUG2")) {
			int dd[NDEPTH] = {0}, tok[NDEPTH] = {0};
			for (int d = 0; d < NDEPTH; d++)
				for (int k = 0; k < NPART; k++) {
					dd[d] += c.truth[(size_t)n * MODE_STATE_BYTES + d * NPART + k] != c.guess[(size_t)n * MODE_STATE_BYTES + d * NPART + k];
					tok[d] += (c.outtok[(size_t)n * MODE_STATE_BYTES + d * NPART + k] & MODE_TOKEN) != 0;
				}
			fprintf(stderr, "  ctu %3d %s guess!=truth by depth: %3d %3d %3d %3d %3d   untouched: %3d %3d %3d %3d %3d  searches %d\n", n, c.valid[n] ? "ok " : "BAD", dd[0], dd[1], dd[2], dd[3], dd[4],
				tok[0], tok[1], tok[2], tok[3], tok[4], c.ctus[n].n_spec_reads);
		}
		if (getenv("HENC_SCHED_DEBUG") && !c.valid[n])
			fprintf(stderr, "  ctu %d invalid: modes %s (%d looked at), ratio used %u true %u of %u (%d comparisons)\n", n,
				sched_guesses_hold(g, c.ctus[n], c.f, &c.truth[(size_t)n * MODE_STATE_BYTES], &c.guess[(size_t)n * MODE_STATE_BYTES], 0, 0, 0, 0, 0) ? "ok" : "WRONG",
				c.ctus[n].n_spec_reads, c.used_intra[n], c.intra_before[n], n * NPART, c.ctus[n].n_ratio_cmp);
	}
	*n_invalid = bad;
	if (getenv("HENC_SCHED_DEBUG")) {
		fprintf(stderr, " verify: %d invalid; causes: replay %d winner %d bits-flip %d ratio-flip %d intra-bits %d\n", bad, henc_sched_causes[0], henc_sched_causes[1],
			henc_sched_causes[2], henc_sched_causes[3], henc_sched_causes[4]);
		memset(henc_sched_causes, 0, sizeof henc_sched_causes);
	}
}

void sched_pass(Cpu &c, Enc &e, int pass)
{
	const Seq &s = c.seq;
	const int W = s.wctu, H = s.hctu;
	CpuGrp g;
	for (int t = 0; t < W + 2 * (H - 1); t++)
		for (int r = 0; r < H; r++) {
			const int col = t - 2 * r;
			if (col < 0 || col >= W) continue;
			const int n = r * W + col;
			if (pass > 0 && c.valid[n] && !c.dirty[n]) continue;
			e.w = c.row_w[r];
			uint8_t *gs = &c.guess[(size_t)n * MODE_STATE_BYTES];
			if (pass > 0) {
				c.ctus[n] = c.ctus_start[n];
				memcpy(gs, &c.truth[(size_t)n * MODE_STATE_BYTES], MODE_STATE_BYTES);
				c.used_intra[n] = c.intra_before[n];
				c.used_parts[n] = (uint32_t)n * NPART;
			} else {
				if (n == 0) memcpy(gs, c.chain_start, MODE_STATE_BYTES);
				sched_known_intra(c.prefix.data(), W, r, col, &c.used_intra[n], &c.used_parts[n]);
			}
			const uint64_t old_hash = c.hash[n];
			memcpy(e.w->mode_in, gs, MODE_STATE_BYTES);
			e.coeff = c.coeff.data() + (size_t)n * 6144;
			e.total_intra_partitions = c.used_intra[n];
			e.total_partitions = c.used_parts[n];
			e.ctu_qp = c.f.qp;
			encode_ctu(g, e, n);
			c.stat_encodes++;
			memcpy(&c.outtok[(size_t)n * MODE_STATE_BYTES], e.w->intra_mode_buffs, MODE_STATE_BYTES);
			c.hash[n] = sched_output_hash(g, s, c.f, c.ctus[n]);
			c.dirty[n] = 0;
			if (pass == 0) {
				// the guesses further on: what this worker's buffers would hold if its own guesses were right
				uint8_t res[MODE_STATE_BYTES];
				for (int i = 0; i < MODE_STATE_BYTES; i++) {
					const uint8_t v = (&e.w->intra_mode_buffs[0][0][0])[i];
					res[i] = (v & MODE_TOKEN) ? gs[(i / (NDEPTH * NPART)) * NDEPTH * NPART + (v & 7) * NPART + i % NPART] : v;
				}
				if (col + 1 < W) memcpy(gs + MODE_STATE_BYTES, res, MODE_STATE_BYTES);
				if (c.row_guess >= 1 && r + 1 < H && col == (W > 1 ? 1 : 0)) {
					uint8_t *dst = &c.guess[(size_t)(r + 1) * W * MODE_STATE_BYTES];
					memcpy(dst, res, MODE_STATE_BYTES);
					if (c.row_guess == 2 && c.f.num_encoded_frames >= 2)   // the depths P slices write: what the previous frame had at this row start
						for (int comp = 0; comp < 2; comp++)
							memcpy(dst + (comp * NDEPTH + 2) * NPART, &c.truth[(size_t)(r + 1) * W * MODE_STATE_BYTES + (comp * NDEPTH + 2) * NPART], 2 * NPART);
				}
				c.prefix[(size_t)r * (W + 1) + col + 1] = c.prefix[(size_t)r * (W + 1) + col] + c.ctus[n].intra_parts;
			} else if (c.hash[n] != old_hash) {
				if (col + 1 < W) c.dirty[n + 1] = 1;
				if (r + 1 < H) {
					if (col > 0) c.dirty[n + W - 1] = 1;
					c.dirty[n + W] = 1;
					if (col + 1 < W) c.dirty[n + W + 1] = 1;
				}
			}
		}
}

// ---- sched = 2: the synchronous wavefront of wfpp_num_threads = N > 1 (the schedule oracle/ref_ctudump.c's HOMER_TURNSTILE forces on the reference): worker t owns the
// rows t, t + N, ... with its own mode buffers; the CTUs of step s = c + 2r all see the counters as of the end of step s - 1; thread 0 alone looks for a scene change ----
void frame_ctus_lockstep(Cpu &c, Enc &e)
{
	const Seq &s = c.seq;
	const int W = s.wctu, H = s.hctu, N = c.cfg.wfpp_num_threads;
	CpuGrp g;
	while ((int)c.row_w.size() < N) c.row_w.push_back(new_work());
	c.f.lockstep = 1;
	uint32_t done_intra = 0, done_ctus = 0;
	std::vector<uint8_t> after(s.nctu * MODE_STATE_BYTES);
	for (int t = 0; t < W + 2 * (H - 1); t++) {
		uint32_t step_intra = 0, step_ctus = 0;
		int fired = -1;
		for (int r = 0; r < H; r++) {
			const int col = t - 2 * r;
			if (col < 0 || col >= W) continue;
			const int n = r * W + col;
			e.w = c.row_w[r % N];
			e.coeff = c.coeff.data() + (size_t)n * 6144;
			e.total_intra_partitions = done_intra;
			e.total_partitions = done_ctus * NPART;
			if (r % N == 0 && c.f.scene_cut_ctu < 0 && fired < 0 && c.f.slice_type == SLICE_P && !ctu_takes_intra_walk(c.f, n) &&
			    scene_cut_fires(s, c.f, done_intra, done_ctus * NPART)) {
				c.f.scene_cut_ctu = fired = n;     // thread 0 decides first in its step: the other CTUs of the step already take the intra walk
				rc_scene_change(c, t);
			}
			e.ctu_qp = ctu_qp_for(c, t, c.f.scene_cut_ctu >= 0 && t >= c.f.scene_cut_ctu % W + 2 * (c.f.scene_cut_ctu / W));
			e.rd_ctx = rd_ctx_for(c, n);
			if (getenv("HENC_WIPE_WORK")) {
				// experiment: a worker that is not the thread (the device's pool, k_encode.hip pool_encode_ctu) - only what travels through the row state
				// there survives from one CTU of a thread to the next: the mode buffers, the prediction window (quirk Q12), "has taken the intra walk";
				// with RD_FULL the shadow CTU's prediction modes are rebuilt from that flag
				Work &w = *e.w;
				std::vector<uint8_t> modes(MODE_STATE_BYTES), pred(sizeof w.pred_y + sizeof w.pred_c);
				memcpy(modes.data(), w.intra_mode_buffs, MODE_STATE_BYTES);
				memcpy(pred.data(), w.pred_y, sizeof w.pred_y);
				memcpy(pred.data() + sizeof w.pred_y, w.pred_c, sizeof w.pred_c);
				const int32_t seen = w.thread_seen_intra;
				WorkSlow *slow = w.slow;
				WorkRd *rd = w.rd_store;
				memset(&w, atoi(getenv("HENC_WIPE_WORK")), sizeof(Work));
				memset(slow, atoi(getenv("HENC_WIPE_WORK")), sizeof(WorkSlow));
				memset(rd, atoi(getenv("HENC_WIPE_WORK")), sizeof(WorkRd));
				w.slow = slow;
				w.rd_store = rd;
				memcpy(w.intra_mode_buffs, modes.data(), MODE_STATE_BYTES);
				memcpy(w.pred_y, pred.data(), sizeof w.pred_y);
				memcpy(w.pred_c, pred.data() + sizeof w.pred_y, sizeof w.pred_c);
				w.thread_seen_intra = seen;
				if (s.rd_mode == RDM_FULL) memset(w.rd_store->rd_pred_mode, seen ? PM_INTRA : PM_INTER, sizeof w.rd_store->rd_pred_mode);
			}
			memcpy(e.w->mode_in, e.w->intra_mode_buffs, MODE_STATE_BYTES);
			encode_ctu(g, e, n);
			resolve_mode_tokens(g, *e.w, c.ctus[n]);
			memcpy(&after[(size_t)n * MODE_STATE_BYTES], e.w->intra_mode_buffs, MODE_STATE_BYTES);
			step_intra += c.ctus[n].intra_parts;
			step_ctus++;
			c.acc_dist += c.ctus[n].distortion;
			post_after_ctu(c, n);
		}
		done_intra += step_intra;
		done_ctus += step_ctus;
	}
	for (int n = 0; n < s.nctu; n++) record_from_outputs(c, n, &after[(size_t)n * MODE_STATE_BYTES]);
}

void frame_ctus_sched(Cpu &c, Enc &e)
{
	const Seq &s = c.seq;
	const size_t nb = (size_t)s.nctu * MODE_STATE_BYTES;
	if (c.guess.size() != nb) {
		c.guess.assign(nb, 0); c.truth.assign(nb, 0); c.outtok.assign(nb, 0);
		c.valid.assign(s.nctu, 0); c.dirty.assign(s.nctu, 0);
		c.intra_before.assign(s.nctu, 0); c.used_intra.assign(s.nctu, 0); c.used_parts.assign(s.nctu, 0); c.hash.assign(s.nctu, 0);
		c.prefix.assign((size_t)s.hctu * (s.wctu + 1), 0);
	}
	while ((int)c.row_w.size() < s.hctu) c.row_w.push_back(new_work());
	c.ctus_start = c.ctus;
	std::fill(c.prefix.begin(), c.prefix.end(), 0);
	if (c.row_guess == 0)   // row starts: what the previous frame found there
		for (int r = 1; r < s.hctu; r++) memcpy(&c.guess[(size_t)r * s.wctu * MODE_STATE_BYTES], &c.truth[(size_t)r * s.wctu * MODE_STATE_BYTES], MODE_STATE_BYTES);
	int bad = 0;
	for (int pass = 0;; pass++) {
		sched_pass(c, e, pass);
		sched_verify(c, &bad);
		c.stat_passes++;
		if (pass == 0) c.stat_invalid_first += bad;
		if (!bad) break;
		if (pass > s.nctu + 2) { fprintf(stderr, "frame_ctus_sched: no convergence\n"); abort(); }
	}
	CpuGrp g;
	for (int n = 0; n < s.nctu; n++) {
		Work *w = c.row_w[0];
		memcpy(w->mode_in, &c.truth[(size_t)n * MODE_STATE_BYTES], MODE_STATE_BYTES);
		resolve_mode_tokens(g, *w, c.ctus[n]);
		record_from_outputs(c, n, n + 1 < s.nctu ? &c.truth[(size_t)(n + 1) * MODE_STATE_BYTES] : c.chain_end);
		c.acc_dist += c.ctus[n].distortion;
	}
	memcpy(c.chain_start, c.chain_end, MODE_STATE_BYTES);
	post_whole_frame(c);
}

}  // namespace

extern "C" {

int henc_cpu_record_bytes(void) { return REC_BYTES; }

void *henc_cpu_create(const HostCfg *cfg)
{
	Cpu *c = new Cpu;
	c->cfg = *cfg;
	const char *why;
	if (!make_seq(*cfg, c->seq, &why)) {
		fprintf(stderr, "henc_cpu_create: unsupported configuration: %s\n", why);
		delete c;
		return nullptr;
	}
	make_geo(c->geo);
	if (cfg->wfpp_num_threads > 1) c->sched = 2;   // one worker per CTU row: the synchronous wavefront
	if (c->seq.rd_mode == RDM_FULL) {
		for (auto &sim : c->rdsim) sim.init(cfg->wfpp_num_threads, c->seq.wctu, c->seq.hctu, c->seq.sao);
		for (auto &v : c->ctx_after) v.assign((size_t)c->seq.nctu * RD_CTX_BYTES, 0);
	}
	if (cfg->bitrate_mode != 0) {
		host_rc_init(*cfg, c->seq, c->st);
		if (!rc_need_table(c->seq.wctu, c->seq.hctu, c->seq.sao, c->sched == 2, c->rc_need)) {
			fprintf(stderr, "henc_cpu_create: rate control: the entropy-coding lag is not row-monotone on this CTU grid\n");
			delete c;
			return nullptr;
		}
	}
	const Seq &s = c->seq;
	c->ctus.resize(s.nctu);
	memset(c->ctus.data(), 0, sizeof(CtuInfo) * s.nctu);
	for (auto &ci : c->ctus) memset(ci.mv_ref_idx, -1, sizeof ci.mv_ref_idx);
	c->w = new_work();
	c->st.engines = clampi(cfg->num_enc_engines, 1, MAX_ENGINES);
	c->local_engines = c->st.engines;
	for (int k = 1; k < c->local_engines; k++) {
		c->eng[k].ctus = c->ctus;
		c->eng[k].w = new_work();
	}
	c->src[0].assign((size_t)s.src_stride_y * s.height, 0);
	c->src[1].assign((size_t)s.src_stride_c * s.height / 2, 0);
	c->src[2].assign((size_t)s.src_stride_c * s.height / 2, 0);
	for (int k = 0; k < 2; k++) {
		c->pic[k][0].assign((size_t)s.stride_y * (s.height + 2 * s.margin_y), 0);
		c->pic[k][1].assign((size_t)s.stride_c * (s.height / 2 + 2 * s.margin_c), 0);
		c->pic[k][2].assign((size_t)s.stride_c * (s.height / 2 + 2 * s.margin_c), 0);
	}
	c->coeff.assign((size_t)s.nctu * 6144, 0);
	c->records.assign((size_t)s.nctu * REC_BYTES, 0);
	return c;
}

// ONE engine of cfg->num_enc_engines (enc_host.h): it is given only its own frames, and the hand-over of the engine before it in front of each of them
void *henc_cpu_create_engine(const HostCfg *cfg, int engine_index)
{
	Cpu *c = (Cpu *)henc_cpu_create(cfg);
	if (!c || engine_index < 0 || engine_index >= c->st.engines) return nullptr;
	c->local_engines = 1;
	return c;
}
int henc_cpu_state_bytes(void) { return (int)sizeof(HostState); }
long henc_cpu_reference_elems(void *h, int comp) { return (long)((Cpu *)h)->pic[0][comp].size(); }
void henc_cpu_export_reference(void *h, int16_t *y, int16_t *u, int16_t *v, void *state)
{
	Cpu &c = *(Cpu *)h;
	int16_t *dst[3] = {y, u, v};
	for (int k = 0; k < 3; k++) memcpy(dst[k], c.pic[c.cur][k].data(), c.pic[c.cur][k].size() * 2);
	memcpy(state, &c.st, sizeof(HostState));
}
void henc_cpu_import_reference(void *h, const int16_t *y, const int16_t *u, const int16_t *v, const void *state)
{
	Cpu &c = *(Cpu *)h;
	const int16_t *src[3] = {y, u, v};
	for (int k = 0; k < 3; k++) memcpy(c.pic[c.cur][k].data(), src[k], c.pic[c.cur][k].size() * 2);
	memcpy(&c.st, state, sizeof(HostState));
}

void henc_cpu_destroy(void *h)
{
	Cpu *c = (Cpu *)h;
	free(c->w->slow);
	free(c->w);
	delete c;
}

// evaluations on a stale prediction window (quirk Q12) over all CTUs of the last frame
long henc_cpu_stale_predictions(void *h)
{
	Cpu &c = *(Cpu *)h;
	long n = 0;
	for (int k = 0; k < c.seq.nctu; k++) n += c.ctus[k].n_stale_pred;
	return n;
}

void henc_cpu_set_sched(void *h, int sched, int row_guess)
{
	if (((Cpu *)h)->cfg.wfpp_num_threads > 1) return;   // the schedule follows from the configuration
	((Cpu *)h)->sched = sched;
	((Cpu *)h)->row_guess = row_guess;
}
void henc_cpu_sched_stats(void *h, int *out3, int reset)
{
	Cpu &c = *(Cpu *)h;
	out3[0] = c.stat_passes; out3[1] = c.stat_encodes; out3[2] = c.stat_invalid_first;
	if (reset) c.stat_passes = c.stat_encodes = c.stat_invalid_first = 0;
}

void henc_cpu_set_sao_trace(const char *path)
{
	if (henc_sao_trace_file) fclose(henc_sao_trace_file);
	henc_sao_trace_file = path && *path ? fopen(path, "w") : nullptr;
}

void henc_cpu_set_trace(const char *path)
{
	if (henc_trace_file) fclose(henc_trace_file);
	henc_trace_file = path && *path ? fopen(path, "w") : nullptr;
}

// CTU decisions of one frame in raster order.  ref_* (8-bit, width x height): the previous frame's final reconstruction as the
// reference gives it (teacher forcing until the loop filters run here too); avg_dist < 0 keeps the encoder's own value.
int henc_cpu_frame_ctus(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
			const uint8_t *ref_v, double avg_dist, int first_ctu, int last_ctu)
{
	Cpu &c = *(Cpu *)h;
	const Seq &s = c.seq;
	const uint8_t *in[3] = {y, u, v}, *rin[3] = {ref_y, ref_u, ref_v};
	if (first_ctu == 0) {
		activate_engine(c, c.local_engines > 1 ? c.st.num_encoded_frames % c.local_engines : 0);
		c.cur ^= 1;
		begin_frame(s, c.st, image_type, c.f);
		if (avg_dist >= 0) c.f.avg_dist = avg_dist;
		c.acc_dist = 0;
		c.intra_parts = c.total_parts = 0;
		for (int comp = 0; comp < 3; comp++) {
			const int w = comp ? s.width / 2 : s.width, hh = comp ? s.height / 2 : s.height, ss = comp ? s.src_stride_c : s.src_stride_y;
			for (int r = 0; r < hh; r++)
				for (int x = 0; x < w; x++) c.src[comp][(size_t)r * ss + x] = in[comp][(size_t)r * w + x];
			if (rin[comp]) {
				int16_t *p = plane0(c, c.cur ^ 1, comp);
				const int rs = comp ? s.stride_c : s.stride_y;
				for (int r = 0; r < hh; r++)
					for (int x = 0; x < w; x++) p[(size_t)r * rs + x] = rin[comp][(size_t)r * w + x];
				pad_plane(p, rs, w, hh, comp ? s.margin_c : s.margin_y);
			}
			c.f.src[comp] = c.src[comp].data();
			c.f.ref[comp] = plane0(c, c.cur ^ 1, comp);
		}
		post_begin_frame(c);
		for (int comp = 0; comp < 3; comp++) c.f.rec[comp] = plane0_of(c, c.rec, comp);
	}
	Enc e;
	memset(&e, 0, sizeof e);
	e.seq = &c.seq;
	e.f = &c.f;
	e.T = hmr_host_tables();
	fast_tables_fill(CpuGrp(), c.ft, e.T, c.f.qp % 6, chroma_qp_table(c.f.qp + c.seq.chroma_qp_offset) % 6);
	e.ft = &c.ft;
	e.geo.p = c.geo;
	e.ctus = c.ctus.data();
	e.w = c.w;
	CpuGrp g;
	if (last_ctu < 0 || last_ctu > s.nctu) last_ctu = s.nctu;
	if (c.sched) {
		if (first_ctu != 0 || last_ctu != s.nctu) return -1;
		if (c.sched == 2) frame_ctus_lockstep(c, e);
		else frame_ctus_sched(c, e);
		first_ctu = last_ctu;
	}
	for (int n = first_ctu; n < last_ctu; n++) {
		e.coeff = c.coeff.data() + (size_t)n * 6144;
		e.total_intra_partitions = c.intra_parts;
		e.total_partitions = c.total_parts;
		if (c.f.scene_cut_ctu < 0 && c.f.slice_type == SLICE_P && scene_cut_fires(s, c.f, c.intra_parts, c.total_parts)) {
			c.f.scene_cut_ctu = n;   // this CTU still takes the inter walk
			rc_scene_change(c, n);
		}
		e.ctu_qp = ctu_qp_for(c, n, c.f.scene_cut_ctu >= 0 && n >= c.f.scene_cut_ctu);
		e.rd_ctx = rd_ctx_for(c, n);
		memcpy(c.w->mode_in, c.w->intra_mode_buffs, MODE_STATE_BYTES);   // one worker in raster order: what the buffers hold IS the inherited state
		if (getenv("HENC_WIPE_NODES")) memset(c.ctus[n].nodes, atoi(getenv("HENC_WIPE_NODES")), sizeof c.ctus[n].nodes);
		if (getenv("HENC_WIPE_PUBLIC")) memset((CtuPublic *)&c.ctus[n], atoi(getenv("HENC_WIPE_PUBLIC")), offsetof(CtuPublic, sao_recon));
		if (getenv("HENC_WIPE_WORK")) {   // experiment: nothing but the mode chain may carry over from CTU to CTU
			uint8_t keep[MODE_STATE_BYTES];
			memcpy(keep, c.w->mode_in, MODE_STATE_BYTES);
			WorkSlow *slow = c.w->slow;
			memset(c.w, atoi(getenv("HENC_WIPE_WORK")), sizeof(Work));
			memset(slow, atoi(getenv("HENC_WIPE_WORK")), sizeof(WorkSlow));
			c.w->slow = slow;
			memcpy(c.w->mode_in, keep, MODE_STATE_BYTES);
		}
		encode_ctu(g, e, n);
		resolve_mode_tokens(g, *c.w, c.ctus[n]);
		c.intra_parts += c.ctus[n].intra_parts;
		c.total_parts += NPART;
		c.acc_dist += c.ctus[n].distortion;
		make_record(c, n, e);
		post_after_ctu(c, n);
	}
	if (last_ctu == s.nctu) {
		const FrameRcOut ro = rc_frame_out(c);
		end_frame(s, c.st, c.f, frame_acc_dist(s, c.cfg.wfpp_num_threads, [&](int n) { return c.ctus[n].distortion; }), &ro);
	}
	return c.f.slice_type;
}

// Debug aid: the same frame in WAVEFRONT order with one worker per CTU row (the device schedule).  `oracle` = the reference's
// records of this frame and `prev_last` = the last record of the previous frame: every row start takes the serial thread's
// mode buffers and every CTU the raster-order intra count from them, so that whatever still differs from the serial run is a
// dependence on worker state that the schedule does not model.
int henc_cpu_frame_ctus_wavefront(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
				  const uint8_t *ref_v, const uint8_t *oracle, const uint8_t *prev_last)
{
	Cpu &c = *(Cpu *)h;
	const Seq &s = c.seq;
	const uint8_t *in[3] = {y, u, v}, *rin[3] = {ref_y, ref_u, ref_v};
	c.cur ^= 1;
	begin_frame(s, c.st, image_type, c.f);
	c.acc_dist = 0;
	for (int comp = 0; comp < 3; comp++) {
		const int w = comp ? s.width / 2 : s.width, hh = comp ? s.height / 2 : s.height, ss = comp ? s.src_stride_c : s.src_stride_y;
		for (int r = 0; r < hh; r++)
			for (int x = 0; x < w; x++) c.src[comp][(size_t)r * ss + x] = in[comp][(size_t)r * w + x];
		if (rin[comp]) {
			int16_t *p = plane0(c, c.cur ^ 1, comp);
			const int rs = comp ? s.stride_c : s.stride_y;
			for (int r = 0; r < hh; r++)
				for (int x = 0; x < w; x++) p[(size_t)r * rs + x] = rin[comp][(size_t)r * w + x];
			pad_plane(p, rs, w, hh, comp ? s.margin_c : s.margin_y);
		}
		c.f.src[comp] = c.src[comp].data();
		c.f.ref[comp] = plane0(c, c.cur ^ 1, comp);
		c.f.rec[comp] = plane0(c, c.cur, comp);
	}
	while ((int)c.row_w.size() < s.hctu) c.row_w.push_back(new_work());
	Enc e;
	memset(&e, 0, sizeof e);
	e.seq = &c.seq; e.f = &c.f; e.T = hmr_host_tables(); e.geo.p = c.geo; e.ctus = c.ctus.data();
	fast_tables_fill(CpuGrp(), c.ft, e.T, c.f.qp % 6, chroma_qp_table(c.f.qp + c.seq.chroma_qp_offset) % 6);
	e.ft = &c.ft;
	CpuGrp g;
	const int mb_off = REC_BYTES - 2560, pm_off = 32 + 768 + 512 + 256 * 4;
	std::vector<uint32_t> intra_prefix(s.nctu + 1, 0);
	for (int n = 0; n < s.nctu; n++) {
		uint32_t cnt = 0;
		const uint8_t *pm = oracle + (size_t)n * REC_BYTES + pm_off;
		for (int i = 0; i < 256; i++) cnt += pm[i] == PM_INTRA;
		intra_prefix[n + 1] = intra_prefix[n] + (c.f.slice_type == SLICE_I ? 256 : cnt);
	}
	std::vector<std::vector<uint8_t>> snap(s.hctu, std::vector<uint8_t>(2560));
	for (int t = 0; t < s.wctu + 2 * (s.hctu - 1); t++)
		for (int r = 0; r < s.hctu; r++) {
			const int col = t - 2 * r;
			if (col < 0 || col >= s.wctu) continue;
			const int n = r * s.wctu + col;
			e.w = c.row_w[r];
			static const int guess_mode = getenv("HENC_WF_GUESS") ? atoi(getenv("HENC_WF_GUESS")) : 0;   // 0 exact state, 1 snapshot of the row above after its CTU 1, 2 per-row chain
			if (col == 0) {
				const uint8_t *src = n == 0 ? prev_last : oracle + (size_t)(n - 1) * REC_BYTES;
				if (guess_mode == 0 || (guess_mode == 1 && r == 0)) { if (src) memcpy(e.w->intra_mode_buffs, src + mb_off, 2560); }
				else if (guess_mode == 1) memcpy(e.w->intra_mode_buffs, snap[r - 1].data(), 2560);
			}
			e.coeff = c.coeff.data() + (size_t)n * 6144;
			e.total_intra_partitions = intra_prefix[n];
			e.total_partitions = (uint32_t)n * NPART;
			memcpy(e.w->mode_in, e.w->intra_mode_buffs, MODE_STATE_BYTES);
			e.ctu_qp = c.f.qp;
			encode_ctu(g, e, n);
			resolve_mode_tokens(g, *e.w, c.ctus[n]);
			c.acc_dist += c.ctus[n].distortion;
			if (col == 1 || s.wctu == 1) memcpy(snap[r].data(), e.w->intra_mode_buffs, 2560);
			Work *keep = c.w;
			c.w = e.w;
			make_record(c, n, e);
			c.w = keep;
		}
	end_frame(s, c.st, c.f, frame_acc_dist(s, c.cfg.wfpp_num_threads, [&](int n) { return c.ctus[n].distortion; }));
	return c.f.slice_type;
}

// the arithmetic coefficient scans of the entropy coder (enc_entropy.h scan_position) against the scan tables built like the reference's (tables.cpp): mismatches
int henc_cpu_scan_mismatches(void)
{
	const DevTables *T = hmr_host_tables();
	int bad = 0;
	for (int mode = SCAN_HOR; mode <= SCAN_DIAG; mode++)
		for (int shift = 2; shift <= 5; shift++) {
			if (mode != SCAN_DIAG && shift > 3) continue;      // find_scan_mode: horizontal / vertical only up to 8 x 8
			const int blk = 1 << (shift - 2), ncg = blk * blk;
			uint16_t cg[64];
			for (int i = 0; i < ncg; i++) {
				if (shift == 3) cg[i] = (mode == SCAN_VER || mode == SCAN_DIAG) ? (uint16_t)(((i & 1) << 1) | (i >> 1)) : (uint16_t)i;
				else if (shift == 5) {
					int d = 0, k = i;
					for (;; d++) { const int len = d < 8 ? d + 1 : 15 - d; if (k < len) break; k -= len; }
					const int row = (d < 8 ? d : 7) - k, col = d - row;
					cg[i] = (uint16_t)(row * 8 + col);
				} else cg[i] = shift > 3 ? (uint16_t)scan4x4_raster(mode, i) : 0;
			}
			for (int i = 0; i < (1 << (2 * shift)); i++)
				if (scan_position(mode, shift, i, cg) != T->scan[mode][shift][i]) bad++;
		}
	return bad;
}

const uint8_t *henc_cpu_records(void *h) { return ((Cpu *)h)->records.data(); }
double henc_cpu_avg_dist(void *h) { return ((Cpu *)h)->st.avg_dist; }

// The whole frame, free running: CTU decisions (raster order), in-loop filters (the round-1 oracle functions), SAO decision + entropy
// coding (homerhevc_amd/csrc/enc/enc_entropy.h, host code of the product) -> Annex-B bytes appended to `stream`; the final picture
// (8 bit, planar) goes to recon when not NULL.  Returns the number of bytes written or a negative value.
long henc_cpu_encode_frame(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, uint8_t *stream, long cap, uint8_t *recon)
{
	Cpu &c = *(Cpu *)h;
	const Seq &s = c.seq;
	henc_cpu_frame_ctus(h, y, u, v, image_type, nullptr, nullptr, nullptr, -1.0, 0, -1);
	// the post-decision stage has run along with the CTU decisions (enc_post.h): deblocked, SAO-filtered and padded picture in pic[cur], one CABAC sub-stream per CTU row
	if (!post_finished(s, c.post) || c.post_errors[0]) { fprintf(stderr, "henc_cpu_encode_frame: the post-decision stage did not finish (errors %d)\n", c.post_errors[0]); return -2; }
	std::vector<const uint8_t *> row_data(s.hctu);
	std::vector<int> row_bytes(s.hctu);
	for (int r = 0; r < s.hctu; r++) { row_data[r] = c.bs.data() + (size_t)r * c.post.row_cap; row_bytes[r] = c.ent[r].bytecnt; }
	std::vector<uint8_t> out;
	assemble_access_unit(c.es, s, c.f, c.cfg.profile, row_data.data(), row_bytes.data(), out);
	if (recon) {
		uint8_t *o = recon;
		for (int k = 0; k < 3; k++) {
			const int w = k ? s.width / 2 : s.width, hh = k ? s.height / 2 : s.height, st = k ? s.stride_c : s.stride_y;
			const int16_t *p = plane0(c, c.cur, k);
			for (int r = 0; r < hh; r++)
				for (int x = 0; x < w; x++) *o++ = (uint8_t)p[(size_t)r * st + x];
		}
	}
	if ((long)out.size() > cap) return -1;
	memcpy(stream, out.data(), out.size());
	return (long)out.size();
}

}  // extern "C"
