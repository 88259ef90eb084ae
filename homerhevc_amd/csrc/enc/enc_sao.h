// SAO parameter decision of one CTU (sao_decide_blk_params, hmr_sao.c:1295: sao_derive_mode_new_rdo :663, sao_derive_mode_merge_rdo :854,
// reconstruct_blk_sao_param :919) and the SAO syntax (code_sao_offset_param / code_sao_blk_param, hmr_arithmetic_encoding.c:1839 / 1971), written once for the
// host entropy stage (enc_entropy.h, and through it the checker build) and for the device kernel k_sao_decide (k_encode.hip).
//
// What the decision takes from the entropy coder is small.  The rate terms come from the counting coder (henc_thread_t.ec) loaded from the real one (ee) at the
// CTU, and the SAO syntax touches two contexts only (sao_merge_flag, sao_type_idx; everything else is bypass-coded), so the decision is a function of
//   * the states of those two contexts in the real coder - which only the SAO syntax of earlier CTUs of the sub-stream changes (+ the WPP hand-over after the
//     second CTU of the row above),
//   * the statistics of the CTU (candidate offsets, their distortion) and the parameters of the left / above CTUs.
// That is what lets the whole decision run on the device behind the statistics pass, with no host in the filter chain.
#pragma once
#include "enc_types.h"
#include "enc_cabac_tables.h"

namespace henc {

enum { SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, SAO_BO = 4 };

// statistics of one CTU: [comp][type][0 diff, 1 count][32] as the frame pass writes them
typedef int32_t SaoStats[3][5][2][32];

struct SaoTables {
	const int32_t *entropy_bits;   // kEntropyBits[128]
	const uint8_t *next_lps;       // kNextStateLps[128]
};

// The counting coder reduced to what the SAO syntax touches (Cabac in counter mode, enc_entropy.h: a context the counter codes with falls to state 0, because
// the reference's transition table for it is never filled; the fraction accumulates across loads of the counter from itself)
struct SaoBits {
	uint64_t frac;
	uint8_t st_merge, st_type;
	const int32_t *bits;
	HENC_INLINE void encode_bin(int ci, uint32_t bin)
	{
		uint8_t &st = ci == CTX_SAO_MERGE ? st_merge : st_type;
		frac += (uint64_t)bits[st ^ bin];
		st = 0;
	}
	HENC_INLINE void encode_ep(uint32_t) { frac += 32768; }
	HENC_INLINE void encode_bins_ep(uint32_t, int n) { frac += (uint64_t)32768 * n; }
	HENC_INLINE uint32_t bitcnt() const { return (uint32_t)(frac >> 15); }
};
// The two contexts in the real coder: their walk through the SAO syntax of a sub-stream
struct SaoContexts {
	uint8_t st_merge, st_type;
	const uint8_t *next_lps;
	HENC_INLINE void encode_bin(int ci, uint32_t bin)
	{
		uint8_t &st = ci == CTX_SAO_MERGE ? st_merge : st_type;
		st = bin != (uint32_t)(st & 1) ? next_lps[st] : (uint8_t)(st < 124 ? st + 2 : st);
	}
	HENC_INLINE void encode_ep(uint32_t) {}
	HENC_INLINE void encode_bins_ep(uint32_t, int) {}
};

// code_sao_offset_param :1839 (8-bit: offsets up to 7)
template <class C>
HENC_INLINE void code_sao_offset_param(C &ee, int comp, const SaoOffset &p, int enabled)
{
	if (!enabled) return;
	if (comp == COMP_Y || comp == COMP_U) {
		const uint32_t sym = p.mode_idc == SAO_OFF ? 0 : (p.type_idc == SAO_BO ? 1 : 2);
		if (sym == 0) ee.encode_bin(CTX_SAO_TYPE, 0);
		else {
			ee.encode_bin(CTX_SAO_TYPE, 1);
			ee.encode_ep(sym == 1 ? 0 : 1);
		}
	}
	if (p.mode_idc == SAO_NEW) {
		const int num_classes = p.type_idc == SAO_BO ? 4 : 5;
		int offset[4], k = 0;
		for (int i = 0; i < num_classes; i++) {
			if (p.type_idc != SAO_BO && i == 2) continue;
			const int cls = p.type_idc == SAO_BO ? (p.type_aux + i) % 32 : i;
			offset[k++] = p.offset[cls];
		}
		for (int i = 0; i < 4; i++) {
			const uint32_t code = (uint32_t)habs(offset[i]), max_symbol = 7;
			const int code_last = max_symbol > code;
			if (code == 0) ee.encode_ep(0);
			else {
				ee.encode_ep(1);
				for (uint32_t j = 0; j + 1 < code; j++) ee.encode_ep(1);
				if (code_last) ee.encode_ep(0);
			}
		}
		if (p.type_idc == SAO_BO) {
			for (int i = 0; i < 4; i++)
				if (offset[i] != 0) ee.encode_ep(offset[i] < 0 ? 1 : 0);
			ee.encode_bins_ep(p.type_aux, 5);
		} else if (comp == COMP_Y || comp == COMP_U) ee.encode_bins_ep(p.type_idc, 2);
	}
}

// code_sao_blk_param :1971
template <class C>
HENC_INLINE void code_sao_blk_param(C &ee, const SaoOffset *p, int left_avail, int above_avail)
{
	int is_left = 0, is_above = 0;
	if (left_avail) {
		is_left = p[0].mode_idc == SAO_MERGE && p[0].type_idc == 0;
		ee.encode_bin(CTX_SAO_MERGE, is_left);
	}
	if (above_avail && !is_left) {
		is_above = p[0].mode_idc == SAO_MERGE && p[0].type_idc == 1;
		ee.encode_bin(CTX_SAO_MERGE, is_above);
	}
	if (!is_left && !is_above)
		for (int comp = 0; comp < 3; comp++) code_sao_offset_param(ee, comp, p[comp], 1);
}

HENC_INLINE int64_t est_sao_dist(int64_t count, int64_t offset, int64_t diff) { return count * offset * offset - diff * offset * 2; }

// sao_get_distortion :620; st = the statistics of the (component, type): [0 diff, 1 count][32]
HENC_INLINE int64_t sao_distortion(int type, int aux, const int32_t *off, const int32_t (*st)[32])
{
	int64_t d = 0;
	if (type != SAO_BO)
		for (int k = 0; k < 5; k++) d += est_sao_dist(st[1][k], off[k], st[0][k]);
	else
		for (int k = aux; k < aux + 4; k++) d += est_sao_dist(st[1][k % 32], off[k % 32], st[0][k % 32]);
	return d;
}

HENC_INLINE void sao_copy(SaoOffset &d, const SaoOffset &s)
{
	d.mode_idc = s.mode_idc; d.type_idc = s.type_idc; d.type_aux = s.type_aux;
	for (int k = 0; k < 32; k++) d.offset[k] = s.offset[k];
}
HENC_INLINE void sao_clear(SaoOffset &d)
{
	d.mode_idc = d.type_idc = d.type_aux = 0;
	for (int k = 0; k < 32; k++) d.offset[k] = 0;
}

// The decision.  `cand` supplies, per (component, type), the candidate of SAO_MODE_NEW: cand.get(comp, type, test) fills test.offset (the offsets as coded,
// sao_derive_offsets :480) and test.type_aux and returns the distortion of their reconstruction (sao_invert_quant_offsets :592 + sao_get_distortion :620).
// st_merge / st_type: the two contexts in the real coder before this CTU's SAO syntax; left / above: the reconstructed parameters ([3]) of the neighbours or nullptr.
// Returns in coded[3] what the syntax carries and in recon[3] what the filter applies.
// ws: nine parameter sets of working space (the candidates under test: in the caller's fast memory, not in locals - their offset arrays are indexed at run time)
constexpr int SAO_DECIDE_WS = 9;
template <class Cand>
HENC_INLINE void sao_decide(const SaoTables &T, uint8_t st_merge, uint8_t st_type, const Cand &cand, const SaoStats &stats, const SaoOffset *left, const SaoOffset *above,
			    const double *lambdas, SaoOffset *coded, SaoOffset *recon, SaoOffset *ws)
{
	const SaoOffset *merge_list[2] = {left, above};
	const int left_avail = left != nullptr, above_avail = above != nullptr;
	const SaoBits ee = {0, st_merge, st_type, T.entropy_bits};   // the real coder as the counter sees it when it is loaded from it: no fraction yet
	SaoBits ec, aux;
	// rd_code_sao_offset_param :2377 / rd_code_sao_blk_param :2391
	auto rate_offset = [&](const SaoBits &src, int comp, const SaoOffset &p) -> uint32_t {
		ec = src;
		const uint32_t init = ec.bitcnt();
		code_sao_offset_param(ec, comp, p, 1);
		return ec.bitcnt() - init;
	};
	auto rate_blk = [&](const SaoOffset *p) -> uint32_t {
		ec = ee;
		const uint32_t init = ec.bitcnt();
		code_sao_blk_param(ec, p, left_avail, above_avail);
		return ec.bitcnt() - init;
	};
	SaoOffset *const mode_param = ws;
	for (int k = 0; k < 3; k++) { sao_clear(mode_param[k]); sao_clear(coded[k]); }
	double min_cost = MAX_COST, mode_cost;
	// ---- SAO_MODE_NEW
	{
		int64_t dist[3], mode_dist[3] = {0, 0, 0};
		SaoOffset *const test = ws + 3;
		double cost, mcost;
		uint32_t rate;
		for (int k = 0; k < 3; k++) sao_clear(test[k]);
		mode_param[0].mode_idc = SAO_OFF;
		rate = rate_offset(ee, 0, mode_param[0]);
		mcost = lambdas[0] * rate;
		aux = ec;
		for (int type = 0; type < 5; type++) {
			test[0].mode_idc = SAO_NEW;
			test[0].type_idc = type;
			dist[0] = cand.get(0, type, test[0]);
			cost = (double)dist[0];
			rate = rate_offset(ee, 0, test[0]);
			cost += lambdas[0] * rate;
			if (cost < mcost) {
				mcost = cost;
				mode_dist[0] = dist[0];
				sao_copy(mode_param[0], test[0]);
				aux = ec;
			}
		}
		cost = 0;
		for (int comp = 1; comp < 3; comp++) {
			mode_param[comp].mode_idc = SAO_OFF;
			mode_dist[comp] = 0;
			rate = comp == 1 ? rate_offset(aux, comp, mode_param[comp]) : rate_offset(ec, comp, mode_param[comp]);
			cost += lambdas[comp] * rate;
		}
		mcost = cost;
		for (int type = 0; type < 5; type++) {
			cost = 0;
			for (int comp = 1; comp < 3; comp++) {
				test[comp].mode_idc = SAO_NEW;
				test[comp].type_idc = type;
				dist[comp] = cand.get(comp, type, test[comp]);
				cost += dist[comp];
				rate = comp == 1 ? rate_offset(aux, comp, test[comp]) : rate_offset(ec, comp, test[comp]);
				cost += lambdas[comp] * rate;
			}
			if (cost < mcost) {
				mcost = cost;
				for (int comp = 1; comp < 3; comp++) { mode_dist[comp] = dist[comp]; sao_copy(mode_param[comp], test[comp]); }
			}
		}
		mode_cost = (double)mode_dist[0] / lambdas[0] + (double)mode_dist[1] / lambdas[1] + (double)mode_dist[2] / lambdas[2];
		mode_cost += rate_blk(mode_param);
		if (mode_cost < min_cost) {
			min_cost = mode_cost;
			for (int k = 0; k < 3; k++) sao_copy(coded[k], mode_param[k]);
		}
	}
	// ---- SAO_MODE_MERGE (the reference's test parameters start from a copy of the candidate; the mode decision only reads what is set here)
	{
		mode_cost = MAX_COST;
		SaoOffset *const best = ws + 6, *const test = ws + 3;
		bool have = false;
		for (int mt = 0; mt < 2; mt++) {
			if (!merge_list[mt]) continue;
			double norm_dist = 0;
			for (int comp = 0; comp < 3; comp++) {
				const SaoOffset &m = merge_list[mt][comp];
				sao_copy(test[comp], m);
				test[comp].mode_idc = SAO_MERGE;
				test[comp].type_idc = mt;
				if (m.mode_idc != SAO_OFF) norm_dist += ((double)sao_distortion(m.type_idc, m.type_aux, m.offset, stats[comp][m.type_idc])) / lambdas[comp];
			}
			const uint32_t rate = rate_blk(test);
			const double cost = norm_dist + (double)rate;
			if (cost < mode_cost) {
				mode_cost = cost;
				for (int k = 0; k < 3; k++) sao_copy(best[k], test[k]);
				have = true;
			}
		}
		if (have && mode_cost < min_cost) {
			min_cost = mode_cost;
			for (int k = 0; k < 3; k++) sao_copy(coded[k], best[k]);
		}
	}
	// reconstruct_blk_sao_param :919
	for (int comp = 0; comp < 3; comp++) {
		SaoOffset &o = recon[comp];
		sao_copy(o, coded[comp]);
		if (o.mode_idc == SAO_OFF) continue;
		if (o.mode_idc == SAO_NEW) {
			// sao_invert_quant_offsets :592 (8 bit: step 1) - also clears what the type does not use
			int32_t *const keep = ws[0].offset;      // (the working sets are free by now)
			for (int k = 0; k < 32; k++) { keep[k] = o.offset[k]; o.offset[k] = 0; }
			if (o.type_idc == SAO_BO)
				for (int i = 0; i < 4; i++) o.offset[(o.type_aux + i) % 32] = keep[(o.type_aux + i) % 32];
			else
				for (int i = 0; i < 5; i++) o.offset[i] = keep[i];
		} else sao_copy(o, merge_list[o.type_idc][comp]);
	}
}

// The SAO Lagrange multipliers, hmr_wpp_sao_ctu :1415 (fixed QP: the same for every CTU)
inline void sao_lambdas(const Seq &S, const FrameCtx &f, double *sao_lambda)
{
	const double qp_temp = (double)f.qp - 12, lambda_scale = 1.0 - hclip(0.05 * (double)(S.gop_size - 1), 0.0, 0.5);
	const double qp_factor = f.slice_type == SLICE_I ? 0.57 * lambda_scale : 0.4624;
	sao_lambda[0] = qp_factor * pow(1.4, qp_temp / 1.4);
	sao_lambda[1] = sao_lambda[2] = qp_factor * pow(1.4, (qp_temp + S.chroma_qp_offset) / 1.4);
}

// the same for every QP a CTU can have: tab[qp][0] luma, [qp][1] chroma (under rate control hmr_wpp_sao_ctu takes the QP of the CTU's first unit, :1420)
inline void sao_lambda_table(const Seq &S, int slice_type, double *tab)
{
	const double lambda_scale = 1.0 - hclip(0.05 * (double)(S.gop_size - 1), 0.0, 0.5);
	const double qp_factor = slice_type == SLICE_I ? 0.57 * lambda_scale : 0.4624;
	for (int qp = 0; qp < 52; qp++) {
		const double qp_temp = (double)qp - 12;
		tab[2 * qp] = qp_factor * pow(1.4, qp_temp / 1.4);
		tab[2 * qp + 1] = qp_factor * pow(1.4, (qp_temp + S.chroma_qp_offset) / 1.4);
	}
}

}  // namespace henc
