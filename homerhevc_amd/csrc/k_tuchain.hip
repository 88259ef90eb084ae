// Fused transform-unit chain: residual -> forward transform -> quantisation (+ sign hiding) -> [dequantisation ->
// inverse transform] -> reconstruction -> SSD, one launch for a batch of TUs.
// Three compositions of the same stages, selected by the kernel's MODE:
//   given prediction - predict, transform, quant, [inv_quant, itransform], reconst, ssd16b: the per-TU sequence of the reference's
//                      intra paths once the prediction exists (hmr_motion_intra.c:1030-1068, hmr_motion_intra_chroma.c:345-365);
//   intra            - the same with the neighbour array and the prediction generated first: encode_intra_cu (hmr_motion_intra.c:970-1069);
//   inter            - encode_inter_cu / encode_inter_cu_chroma (hmr_motion_inter.c:40-230): starts from the CU's residual, SSDs in the
//                      residual domain, keep-or-drop decision on the coded levels.
// The CPU keeps the intermediates of these table calls in scratch windows.  Here a TU stays on chip from the first load to the last store: source and
// prediction are read once, levels, reconstruction, SSD and ac_sum are written once; residual, coefficients and de-quantised
// planes never reach HBM.  The arithmetic of every stage is that of the stand-alone kernels, so outputs are bit-identical to the
// seven calls in sequence.
//
// Transform mapping ("lane = row"): N lanes own one TU (64/N TUs per wave) and each lane holds one row of its TU packed two
// 16-bit samples per register.  A 1-D stage is then, for every basis row k, N/2 v_dot2_i32_i16 per lane against basis words
// that are wave-uniform (all TUs of a wave walk the same k), i.e. scalar loads from the constant tables - no LDS traffic in the
// inner loop.  Only the transposes between stages go through LDS (pitch N+2: conflict-free column writes).  The integer dot
// products are exact (|sum| < 2^31), MFMA is not applicable: the stages need exact 32-bit sums with a saturating 16-bit pack
// in between.  Quantisation runs on the LDS image four coefficients per lane and step; sign hiding with one lane per coefficient group.
#include "common.h"
#include "intra_device.h"
#include "tq_device.h"
#include "vec.h"

namespace {

typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(int a, int b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c, false); }
__device__ __forceinline__ int pack2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }

// out[k] = sat16((sum_i row[i] * basis[k][i] + rnd) >> shift) for k = 0..N-1, written to dst[k * pitch] (column walk)
template <int N>
__device__ __forceinline__ void stage_to_lds(const int (&row)[N / 2], const int *__restrict__ basis, int shift, int16_t *dst, int pitch)
{
	const int rnd = 1 << (shift - 1);
#pragma unroll 4
	for (int k = 0; k < N; k++) {
		int s = 0;
#pragma unroll
		for (int i = 0; i < N / 2; i++) s = dot2(row[i], basis[k * (N / 2) + i], s);
		dst[k * pitch] = (int16_t)sat16i((s + rnd) >> shift);
	}
}

// Inverse DCT stage by even / odd input index (N >= 8): the basis column k and its mirror N-1-k share magnitudes - equal for even input indices, opposite for
// odd ones - so out[k] = E + O and out[N-1-k] = E - O with E, O two half-length dot products: N*N/4 v_dot2 per row instead of N*N/2, plus N/2 v_perm to
// split the packed row into its even and odd samples.  Exact: E + O is the same 32-bit sum in another order.
template <int N>
__device__ __forceinline__ void split_even_odd(const int (&row)[N / 2], int (&ev)[N / 4], int (&od)[N / 4])
{
#pragma unroll
	for (int i = 0; i < N / 4; i++) {
		ev[i] = (int)__builtin_amdgcn_perm((unsigned)row[2 * i + 1], (unsigned)row[2 * i], 0x05040100u);
		od[i] = (int)__builtin_amdgcn_perm((unsigned)row[2 * i + 1], (unsigned)row[2 * i], 0x07060302u);
	}
}
template <int N>
__device__ __forceinline__ void eo_pair(const int (&ev)[N / 4], const int (&od)[N / 4], const int *__restrict__ beo, int k, int &sum, int &dif)
{
	int e = 0, o = 0;
#pragma unroll
	for (int i = 0; i < N / 4; i++) {
		e = dot2(ev[i], beo[k * (N / 2) + i], e);
		o = dot2(od[i], beo[k * (N / 2) + N / 4 + i], o);
	}
	sum = e + o;
	dif = e - o;
}
// out[k] = sat16((sum_i row[i] * M[i][k] + rnd) >> shift), written to dst[k * pitch]
template <int N>
__device__ __forceinline__ void istage_eo_to_lds(const int (&row)[N / 2], const int *__restrict__ beo, int shift, int16_t *dst, int pitch)
{
	const int rnd = 1 << (shift - 1);
	int ev[N / 4], od[N / 4];
	split_even_odd<N>(row, ev, od);
#pragma unroll 4
	for (int k = 0; k < N / 2; k++) {
		int s, d;
		eo_pair<N>(ev, od, beo, k, s, d);
		dst[k * pitch] = (int16_t)sat16i((s + rnd) >> shift);
		dst[(N - 1 - k) * pitch] = (int16_t)sat16i((d + rnd) >> shift);
	}
}
template <int N>
__device__ __forceinline__ void load_row_lds(int (&row)[N / 2], const int16_t *src)
{
#pragma unroll
	for (int i = 0; i < N / 2; i++) row[i] = *reinterpret_cast<const int *>(src + 2 * i);
}

// INTRA: the job also carries the neighbourhood and the mode; the prediction is generated first (neighbour array, smoothing, planar / DC /
// angular, lane = row: each lane predicts its row) into the prediction plane and the chain continues as for a given prediction -
// encode_intra_cu's data path (hmr_motion_intra.c:1011-1068) in one launch.  D = plane under reconstruction (neighbours), Pp = prediction out.
//
// MODE 2 (inter): encode_inter_cu / encode_inter_cu_chroma (hmr_motion_inter.c:40-230).  O addresses the RESIDUAL of the CU (predict ran on
// the whole CU), the transform is always the DCT, and a coded TU is weighed against dropping its levels: ssd_zero = SSD(residual, 0) and
// ssd = SSD(residual, reconstructed residual), both scaled by the job's chroma weight and truncated to uint32, levels dropped when
// ssd_zero <= ssd + zero_thr * sum (doubles).  The returned SSD is the residual-domain one, as in the reference.
enum { TU_GIVEN_PRED = 0, TU_INTRA = 1, TU_INTER = 2 };
constexpr int EO_MIN = 16;      // smallest TU whose inverse stages use the even / odd split (8x8: the split costs registers and saves 16 dot products per row)
// LDS image of one workgroup: carved from one raw buffer so that several (N, MODE) bodies can share a launch (k_tu_chain_multi)
template <int N>
struct alignas(16) TuLds {
	static constexpr int TW = HMR_WAVE / N, P = N + 2, E = N * N;
	int16_t sA[HMR_WAVES_PER_BLOCK][TW][N * P];    // coefficients (linear) -> de-quantised coefficients (transposed, pitched)
	int16_t sT[HMR_WAVES_PER_BLOCK][TW][N * P];    // stage intermediates (pitched) / deltaU (linear) during quantisation
	int16_t sLev[HMR_WAVES_PER_BLOCK][TW][E];
	unsigned long long sNz[HMR_WAVES_PER_BLOCK][TW];
	int sAc[HMR_WAVES_PER_BLOCK][TW];
};
template <int N>
struct alignas(16) TuLdsIntra {
	static constexpr int TW = HMR_WAVE / N;
	int16_t sAdi[HMR_WAVES_PER_BLOCK][TW][2][4 * N + 4];
	int16_t sMain[HMR_WAVES_PER_BLOCK][TW][3 * N + 2];
};
template <int N, int MODE> constexpr int tu_lds_bytes() { return (int)sizeof(TuLds<N>) + (MODE == TU_INTRA ? (int)sizeof(TuLdsIntra<N>) : 0); }

// inter TU jobs whose `reserved` word carries bit 0 address the SOURCE picture: the residual (source - prediction, 16-bit wrap: the `predict` call the
// reference issues per CU ahead of encode_inter) is formed in the kernel and never written out
__device__ __forceinline__ bool inter_from_source(const hmr_gpu_inter_tu_job &jb) { return (jb.reserved & 1u) != 0; }
__device__ __forceinline__ bool inter_from_source(const hmr_gpu_tu_job &) { return false; }
__device__ __forceinline__ bool inter_from_source(const hmr_gpu_itu_job &) { return false; }

template <int N, int MODE>
__device__ __forceinline__ void tu_chain_body(const void *__restrict__ jobs_v, int njobs, const int16_t *__restrict__ O, int16_t *__restrict__ Pp,
					      int16_t *__restrict__ L, int16_t *__restrict__ Rr, uint32_t *__restrict__ ssd_out, int32_t *__restrict__ ac_out,
					      const DevTables *__restrict__ tab, const int16_t *D, const hmr_gpu_intra_result *__restrict__ modes, int rounds,
					      char *lds, unsigned block, unsigned grid)   // D may alias Rr (in-place reconstruction)
{
	constexpr bool INTRA = MODE == TU_INTRA, INTER = MODE == TU_INTER;
	using JobT = typename std::conditional<INTRA, hmr_gpu_itu_job, typename std::conditional<INTER, hmr_gpu_inter_tu_job, hmr_gpu_tu_job>::type>::type;
	const JobT *__restrict__ jobs = static_cast<const JobT *>(jobs_v);
	constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	constexpr int E = N * N, P = N + 2;
	constexpr int TW = HMR_WAVE / N;                       // TUs per wave in the transform mapping (lane = row)
	constexpr int GQ = N == 4 ? 4 : N == 32 ? 64 : 16;     // lanes per TU in the quant mapping (four coefficients per lane and step); measured best
	constexpr int TQ = HMR_WAVE / GQ, QPASSES = TW / TQ;   // TUs per quant pass, passes per wave
	constexpr int JPB = TW * HMR_WAVES_PER_BLOCK;
	constexpr int sh1 = L2 - 1, sh2 = L2 + 6, SIDE = N / 4;
	TuLds<N> &S = *reinterpret_cast<TuLds<N> *>(lds);
	[[maybe_unused]] TuLdsIntra<N> &SI = *reinterpret_cast<TuLdsIntra<N> *>(lds + sizeof(TuLds<N>));
	const int lane = lane_id(), w = wave_in_block(), tu = lane / N, row = lane % N;
	const JobRange jr = xcd_job_range(njobs, JPB, block, grid);
	for (long base = jr.begin; base < jr.end; base += jr.stride)
	// rounds > 1: the job array holds `rounds` sets of njobs jobs; job j of set r + 1 may read what job j of set r reconstructed (the four
	// children of a CU, hmr_motion_intra.c:1441-1477) - the same lanes run them back to back, with the stores of one round made visible
	// to the loads of the next
	for (int rnd = 0; rnd < rounds; rnd++) {
		const JobT *__restrict__ jobs_r = jobs + (long)rnd * njobs;
		uint32_t *__restrict__ ssd_r = ssd_out + (long)rnd * njobs;
		int32_t *__restrict__ ac_r = ac_out + (long)rnd * njobs;
		if (rnd) {      // producer and consumer lanes are in the same wave, hence on the same CU and behind the same vector L1: workgroup scope is enough
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		const long j = base + w * TW + tu;
		const bool ok = j < jr.end;
		JobT jb = {};
		int r[N / 2];
#pragma unroll
		for (int i = 0; i < N / 2; i++) r[i] = 0;
		const int16_t *orow_p = O, *prow_p = Pp;
		if (ok) {
			jb = jobs_r[j];
			if constexpr (INTRA) {
				// mode handed over on the device by the search that ran before this launch: smoothing (hmr_motion_intra.c:1011-1012) and
				// scan (find_scan_mode, hmr_tables.c:398-402) follow from mode and TU size
				if ((jb.flags & 0x100u) && modes) {
					const int m = modes[jb.mode].best_mode & 0xff;
					const bool luma = (jb.flags >> 7) & 1;         // chroma TUs: never smoothed, mode-dependent scan for 4x4 only (hmr_tables.c:403-411)
					const int d10 = m > 10 ? m - 10 : 10 - m, d26 = m > 26 ? m - 26 : 26 - m, dmin = d10 < d26 ? d10 : d26;
					constexpr int thr = N == 4 ? 10 : N == 8 ? 7 : N == 16 ? 1 : 0;
					const unsigned filt = (luma && m != 1 && dmin > thr) ? 0x40u : 0u;
					const unsigned scan = (luma ? N <= 8 : N == 4) ? (d26 < 5 ? 1u : d10 < 5 ? 2u : 3u) : 3u;
					jb.mode = (uint32_t)m;
					jb.flags = (jb.flags & ~0x40u) | filt;
					jb.p0 = (jb.p0 & ~3u) | scan;
				}
			}
			orow_p = O + jb.orig_off + (size_t)row * jb.orig_stride;
			prow_p = Pp + jb.pred_off + (size_t)row * jb.pred_stride;
		}
		if constexpr (INTRA) {
			// N lanes per TU build the 4N+1 neighbours, smooth them when asked, and predict one row each
			int16_t *adi = SI.sAdi[w][tu][0], *adif = SI.sAdi[w][tu][1], *mainr = SI.sMain[w][tu] + N;
			const unsigned fl = ok ? jb.flags : 0u;
			if (ok)
				intra_build_refs<N, N>(adi, D + jb.dec_off, (int)jb.dec_stride, fl & 1, fl & 2, (fl & 4) ? (int)(jb.sizes & 0xffff) : 0,
						       (fl & 8) ? (int)(jb.sizes >> 16) : 0, row);
			wave_sync();
			const bool filt = (fl >> 6) & 1;
			if (ok && filt) intra_filter_refs<N, N>(adi, adif, (fl & 32) != 0, row);
			wave_sync();
			const IntraMode m = intra_mode_setup(ok ? (int)jb.mode : 0);
			const int16_t *mid = (filt ? adif : adi) + 2 * N;
			if (ok) intra_fill_main<N, N>(m, mid, mainr, row);
			const int dc = intra_dc<N, N>(mid, row, ok && m.mode == 1);
			wave_sync();
			if (ok) {
				const bool edge = ((fl >> 7) & 1) && N <= 16;      // luma edge filters (DC, pure horizontal / vertical)
				int16_t *pw = Pp + jb.pred_off + (size_t)row * jb.pred_stride;
#pragma unroll 2
				for (int x4 = 0; x4 < N; x4 += 4) {
					i16x4 pv;
#pragma unroll
					for (int q = 0; q < 4; q++) pv.v[q] = (int16_t)intra_pixel<N>(m, mid, mainr, dc, edge, x4 + q, row);
					st4(pw + x4, pv);
				}
			}
			wave_sync();
		}
		if (ok) {
			// K3 predict: residual row (16-bit wrap per sample).  Source and prediction rows are read again for the reconstruction at the
			// end of the chain (L2-resident) instead of being held in registers across it: the chain is register-bound.
#pragma unroll
			for (int i = 0; i < N / 4; i++) {
				const i16x4 vo = ld4(orow_p + 4 * i);
				if constexpr (INTER) {     // O is the residual plane - or the source picture, the residual then formed here like `predict` forms it
					if (inter_from_source(jb)) {
						const i16x4 vp = ld4(prow_p + 4 * i);
						r[2 * i] = pack2((int16_t)(vo.v[0] - vp.v[0]), (int16_t)(vo.v[1] - vp.v[1]));
						r[2 * i + 1] = pack2((int16_t)(vo.v[2] - vp.v[2]), (int16_t)(vo.v[3] - vp.v[3]));
					} else {
						r[2 * i] = pack2(vo.v[0], vo.v[1]);
						r[2 * i + 1] = pack2(vo.v[2], vo.v[3]);
					}
				} else {
					const i16x4 vp = ld4(prow_p + 4 * i);
					r[2 * i] = pack2((int16_t)(vo.v[0] - vp.v[0]), (int16_t)(vo.v[1] - vp.v[1]));
					r[2 * i + 1] = pack2((int16_t)(vo.v[2] - vp.v[2]), (int16_t)(vo.v[3] - vp.v[3]));
				}
			}
		}
		const bool is_dst = N == 4 && ((jb.p0 >> 7) & 1);
		// basis words: DST only exists for N = 4, where a lane group may need a different basis than its neighbours -> per-lane pointer
		const int *Mf = reinterpret_cast<const int *>(is_dst ? tab->dst4 : tab->dct[L2 - 2]);
		const int *Mt = reinterpret_cast<const int *>(is_dst ? tab->dst4_t : tab->dct_t[L2 - 2]);
		int16_t *tA = S.sA[w][tu], *tT = S.sT[w][tu], *lev = S.sLev[w][tu];
		if (lane % N == 0) { S.sNz[w][tu] = 0; S.sAc[w][tu] = 0; }
		// K12 stage 1: tmp[k][row] = sum_i M[k][i] * res[row][i]
		stage_to_lds<N>(r, Mf, sh1, tT + row, P);
		wave_sync();
		// K12 stage 2: coeff[k2][k1 = row] = sum_j M[k2][j] * tmp[k1][j]; stored linear for the quantiser
		load_row_lds<N>(r, tT + row * P);
		wave_sync();
		stage_to_lds<N>(r, Mf, sh2, tA + row, N);
		wave_sync();
		// K14 quant + sign hiding on the LDS image, TQ TUs per pass with GQ lanes each
		for (int pass = 0; pass < QPASSES; pass++) {
			const int qt = pass * TQ + lane / GQ, l = lane % GQ;          // TU handled by this lane in this pass
			const long qj = base + w * TW + qt;
			const bool qok = qj < jr.end;
			// the TU's parameters live with its transform lanes: fetch them from lane qt * N
			const unsigned p0 = __shfl((int)jb.p0, qt * N, HMR_WAVE), p1 = __shfl((int)jb.p1, qt * N, HMR_WAVE);
			const int scan_mode = p0 & 3, comp = (p0 >> 2) & 3, is_intra = (p0 >> 4) & 1, slice_i = (p0 >> 5) & 1;
			const int per = p1 & 0xff, rem = (p1 >> 8) & 0xff;
			int16_t *qc = S.sA[w][qt], *qd = S.sT[w][qt], *ql = S.sLev[w][qt];
			int ac = 0;
			if (qok) {
				const int32_t *q = tab->quant[L2 - 2][(is_intra ? 0 : 3) + comp][rem];
				const uint8_t *b2c = tab->blk2cg[scan_mode][L2];
				const int qbits = 14 + per + (7 - L2), qbits8 = qbits - 8;
				const uint32_t add = (uint32_t)(slice_i ? 171 : 85) << (qbits - 9);
				uint32_t sum = 0;
				unsigned long long nz = 0;
				// four consecutive coefficients of a row per step (they share a coefficient group): 8-byte LDS accesses, one 16-byte
				// load of the quantiser entries, one group-table lookup
				for (int e0 = 4 * l; e0 < E; e0 += 4 * GQ) {
					int16_t cs[4], lv4[4], du4[4];
					int32_t qv[4];
					__builtin_memcpy(cs, qc + e0, 8);
					__builtin_memcpy(qv, q + e0, 16);
					bool any = false;
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const int s = cs[k];
						const uint32_t mag = (uint16_t)(s < 0 ? -s : s);
						const uint32_t aux = mag * (uint32_t)qv[k];
						const int c = (int)(aux + add) >> qbits;
						const int d = (int)(aux - ((uint32_t)c << qbits)) >> qbits8;
						sum += (uint32_t)c;
						const int m16 = sat16i(c);
						const int lv = (int16_t)(s > 0 ? m16 : (s < 0 ? -m16 : 0));
						lv4[k] = (int16_t)lv;
						du4[k] = (int16_t)sat16i(d);
						any |= lv != 0;
					}
					__builtin_memcpy(ql + e0, lv4, 8);
					__builtin_memcpy(qd + e0, du4, 8);
					if (any) nz |= 1ull << b2c[((e0 / N) >> 2) * SIDE + ((e0 % N) >> 2)];
				}
				ac = (int)group_sum<GQ>(sum);
				if (nz) atomicOr(&S.sNz[w][qt], nz);
				if (l == 0) S.sAc[w][qt] = ac;
			}
		}
		wave_sync();
		// sign hiding: one lane per 16-coefficient group over all TUs of the wave (4N groups)
		{
			constexpr int CGS = E / 16, TOTAL = TW * CGS;
			for (int c0 = 0; c0 < TOTAL; c0 += HMR_WAVE) {
				const int c = c0 + lane, t = c < TOTAL ? c / CGS : 0, cg = c % CGS;
				const unsigned p0 = __shfl((int)jb.p0, t * N, HMR_WAVE);
				const unsigned long long m = S.sNz[w][t];
				const bool run = c < TOTAL && base + w * TW + t < jr.end && ((p0 >> 6) & 1) && S.sAc[w][t] >= 2 && ((m >> cg) & 1);
				if (run) {
					const uint32_t *scan = tab->scan[p0 & 3][L2];
					const bool is_last = cg == 63 - __clzll((long long)m);
					switch (p0 & 3) {
					case 3: sbh_group_block<3, N>(S.sLev[w][t], S.sA[w][t], S.sT[w][t], scan, cg, is_last); break;
					case 1: sbh_group_block<1, N>(S.sLev[w][t], S.sA[w][t], S.sT[w][t], scan, cg, is_last); break;
					case 2: sbh_group_block<2, N>(S.sLev[w][t], S.sA[w][t], S.sT[w][t], scan, cg, is_last); break;
					default: sbh_group_serial(S.sLev[w][t], S.sA[w][t], S.sT[w][t], scan, cg, is_last); break;
					}
				}
			}
		}
		wave_sync();
		const int ac = S.sAc[w][tu];
		const bool coded = ok && ac != 0;                               // the reference skips dequant / inverse transform for all-zero TUs
		{
			const int comp = (jb.p0 >> 2) & 3, is_intra = (jb.p0 >> 4) & 1, per = jb.p1 & 0xff, rem = (jb.p1 >> 8) & 0xff;
			const int32_t *iq = tab->dequant[L2 - 2][is_intra ? 0 : 3 + comp][rem];
			const int iq_shift = 3 + L2;
			if (coded)                                              // lane = coefficient ROW i here: deqT[col][i] = deq[i][col]
				for (int c0 = 0; c0 < N; c0 += 4) {
					int16_t l4[4];
					int32_t q4[4];
					__builtin_memcpy(l4, lev + row * N + c0, 8);
					__builtin_memcpy(q4, iq + row * N + c0, 16);
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t prod = (uint32_t)(int)l4[k] * (uint32_t)q4[k];
						int v;
						if (iq_shift > per) v = (int)(prod + (1u << (iq_shift - per - 1))) >> (iq_shift - per);
						else v = (int)(prod << (per - iq_shift));
						tA[(c0 + k) * P + row] = (int16_t)sat16i(v);
					}
				}
		}
		wave_sync();
		// K13 stage 1: tmp[col = row][k] = sum_i M[i][k] * coeff[i][col]; written transposed (tmpT[k][col]) for stage 2
		const int *Meo = reinterpret_cast<const int *>(tab->dct_eo[L2 - 2]);
		const bool any_coded = __any(coded);                            // a wavefront whose TUs are all uncoded skips the inverse path like the reference does per TU
		if (any_coded) {
			load_row_lds<N>(r, tA + row * P);
			wave_sync();
			if constexpr (N >= EO_MIN) istage_eo_to_lds<N>(r, Meo, 7, tT + row, P);
			else stage_to_lds<N>(r, Mt, 7, tT + row, P);
			wave_sync();
		}
		// K13 stage 2 + K4 reconst + K2 ssd: out[y = row][x] = sum_i M[i][x] * tmp[i][y]
		load_row_lds<N>(r, tT + row * P);
		uint32_t ssd = 0;
		bool drop = false;
		if constexpr (!INTER) {
			if (ok) {
				int16_t *ro = Rr + jb.rec_off + (size_t)row * jb.rec_stride;
				auto emit4 = [&](int xb, const int (&res)[4]) {
					i16x4 outv;
					const i16x4 vo = ld4(orow_p + xb), vp = ld4(prow_p + xb);
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const int rec = clip3i(sat16i((int)vp.v[q] + res[q]), 0, 255);
						outv.v[q] = (int16_t)rec;
						const int d = (int16_t)((int)vo.v[q] - rec);
						ssd += (uint32_t)(d * d);
					}
					st4(ro + xb, outv);
				};
				if (N >= EO_MIN && coded) {
					// even / odd: four outputs from the left half and their four mirrors per step
					int ev[N / 4 > 0 ? N / 4 : 1], od[N / 4 > 0 ? N / 4 : 1];
					if constexpr (N >= EO_MIN) split_even_odd<N>(r, ev, od);
#pragma unroll 2
					for (int x4 = 0; x4 < N / 2; x4 += 4) {
						int a[4], b[4];
#pragma unroll
						for (int q = 0; q < 4; q++) {
							int sm = 0, df = 0;
							if constexpr (N >= EO_MIN) eo_pair<N>(ev, od, Meo, x4 + q, sm, df);
							a[q] = sat16i((sm + 2048) >> 12);
							b[3 - q] = sat16i((df + 2048) >> 12);
						}
						emit4(x4, a);
						emit4(N - 4 - x4, b);
					}
				} else {
#pragma unroll 2
					for (int x4 = 0; x4 < N; x4 += 4) {
						int res[4] = {0, 0, 0, 0};
						if (coded) {
#pragma unroll
							for (int q = 0; q < 4; q++) {
								int s = 0;
#pragma unroll
								for (int i = 0; i < N / 2; i++) s = dot2(r[i], Mt[(x4 + q) * (N / 2) + i], s);
								res[q] = sat16i((s + 2048) >> 12);
							}
						}
						emit4(x4, res);
					}
				}
			}
			ssd = group_sum<N>(ssd);
			if (ok && row == 0) ssd_r[j] = ssd;
		} else {
			// the reconstruction is written as if the levels were kept while both SSDs are accumulated; the (rare) dropped TU is rewritten
			// from the prediction alone once the decision is known
			uint32_t ssd_zero = 0;
			int16_t *ro = ok ? Rr + jb.rec_off + (size_t)row * jb.rec_stride : nullptr;
			if (ok) {
				auto emit4 = [&](int xb, const int (&res)[4]) {
					i16x4 vr = ld4(orow_p + xb);
					const i16x4 vp = ld4(prow_p + xb);
					if (inter_from_source(jb)) {
#pragma unroll
						for (int q = 0; q < 4; q++) vr.v[q] = (int16_t)(vr.v[q] - vp.v[q]);
					}
					i16x4 outv;
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const int d = (int16_t)((int)vr.v[q] - res[q]);
						ssd += (uint32_t)(d * d);
						ssd_zero += (uint32_t)((int)vr.v[q] * (int)vr.v[q]);
						outv.v[q] = (int16_t)clip3i(sat16i((int)vp.v[q] + res[q]), 0, 255);
					}
					st4(ro + xb, outv);
				};
				if (N >= EO_MIN && coded) {
					int ev[N / 4 > 0 ? N / 4 : 1], od[N / 4 > 0 ? N / 4 : 1];
					if constexpr (N >= EO_MIN) split_even_odd<N>(r, ev, od);
#pragma unroll 2
					for (int x4 = 0; x4 < N / 2; x4 += 4) {
						int a[4], b[4];
#pragma unroll
						for (int q = 0; q < 4; q++) {
							int sm = 0, df = 0;
							if constexpr (N >= EO_MIN) eo_pair<N>(ev, od, Meo, x4 + q, sm, df);
							a[q] = sat16i((sm + 2048) >> 12);
							b[3 - q] = sat16i((df + 2048) >> 12);
						}
						emit4(x4, a);
						emit4(N - 4 - x4, b);
					}
				} else {
#pragma unroll 2
					for (int x4 = 0; x4 < N; x4 += 4) {
						int res[4] = {0, 0, 0, 0};
						if (coded) {
#pragma unroll
							for (int q = 0; q < 4; q++) {
								int s = 0;
#pragma unroll
								for (int i = 0; i < N / 2; i++) s = dot2(r[i], Mt[(x4 + q) * (N / 2) + i], s);
								res[q] = sat16i((s + 2048) >> 12);
							}
						}
						emit4(x4, res);
					}
				}
			}
			ssd = group_sum<N>(ssd);
			ssd_zero = group_sum<N>(ssd_zero);
			const uint32_t w_ssd = (uint32_t)(jb.weight * ssd), w_zero = (uint32_t)(jb.weight * ssd_zero);
			const int comp = (jb.p0 >> 2) & 3;
			// luma keeps ssd in an int (hmr_motion_inter.c:42), chroma in a uint32 (:135)
			drop = coded && (double)w_zero <= (comp == 0 ? (double)(int)w_ssd : (double)w_ssd) + jb.zero_thr * ac;
			if (__any(drop)) {
				if (drop) {
#pragma unroll 2
					for (int x4 = 0; x4 < N; x4 += 4) {
						const i16x4 vp = ld4(prow_p + x4);
						i16x4 outv;
#pragma unroll
						for (int q = 0; q < 4; q++) outv.v[q] = (int16_t)clip3i((int)vp.v[q], 0, 255);
						st4(ro + x4, outv);
					}
				}
			}
			if (ok) {
				if (row == 0) {
					ssd_r[j] = coded ? w_ssd : w_zero;
					if (drop) S.sAc[w][tu] = 0;
				}
			}
		}
		if (lane % N == 0) S.sNz[w][tu] = drop ? 1ull : 0ull;     // reused as the "levels dropped" flag of the TU for the store below
		wave_sync();
		// levels out (coalesced)
		for (int pass = 0; pass < QPASSES; pass++) {
			const int qt = pass * TQ + lane / GQ, l = lane % GQ;
			const long qj = base + w * TW + qt;
			const unsigned lev_off = __shfl((int)jb.lev_off, qt * N, HMR_WAVE);
			if (qj < jr.end) {
				int16_t *lo = L + lev_off;
				const int16_t *ql = S.sLev[w][qt];
				const bool zero = INTER && S.sNz[w][qt] != 0;
				const i16x4 z4 = {{0, 0, 0, 0}};
				for (int e0 = 4 * l; e0 < E; e0 += 4 * GQ) st4(lo + e0, zero ? z4 : ld4(ql + e0));
				if (l == 0) ac_r[qj] = S.sAc[w][qt];
			}
		}
		wave_sync();
		wave_sync();
	}
}

template <int N, int MODE>
__global__ __launch_bounds__(HMR_BLOCK, (N == 32 ? 1 : 4)) void k_tu_chain(const void *__restrict__ jobs_v, int njobs, const int16_t *__restrict__ O,
							   int16_t *__restrict__ Pp, int16_t *__restrict__ L, int16_t *__restrict__ Rr,
							   uint32_t *__restrict__ ssd_out, int32_t *__restrict__ ac_out, const DevTables *__restrict__ tab,
							   const int16_t *D, const hmr_gpu_intra_result *__restrict__ modes, int rounds)
{
	__shared__ __attribute__((aligned(16))) char lds[tu_lds_bytes<N, MODE>()];
	tu_chain_body<N, MODE>(jobs_v, njobs, O, Pp, L, Rr, ssd_out, ac_out, tab, D, modes, rounds, lds, blockIdx.x, gridDim.x);
}

// Several TU batches - any mix of block size and kind (given prediction / intra / inter) - as segments of ONE launch: block b runs the body of the
// segment whose block range contains it (ranges start at multiples of 8 blocks, so b % 8 is still the XCD).  The LDS image is the largest body's.
struct TuSegTab {
	const void *jobs[HMR_GPU_MAX_SEGMENTS];
	uint32_t *ssd[HMR_GPU_MAX_SEGMENTS];
	int32_t *ac[HMR_GPU_MAX_SEGMENTS];
	const hmr_gpu_intra_result *modes[HMR_GPU_MAX_SEGMENTS];
	int njobs[HMR_GPU_MAX_SEGMENTS], size[HMR_GPU_MAX_SEGMENTS], kind[HMR_GPU_MAX_SEGMENTS], rounds[HMR_GPU_MAX_SEGMENTS], first[HMR_GPU_MAX_SEGMENTS],
		blocks[HMR_GPU_MAX_SEGMENTS];
	int n;
};
template <int MAXN>
__global__ __launch_bounds__(HMR_BLOCK, (MAXN == 32 ? 1 : 4)) void k_tu_chain_multi(TuSegTab t, const int16_t *__restrict__ O, int16_t *__restrict__ Pp,
									       int16_t *__restrict__ L, int16_t *__restrict__ Rr, const DevTables *__restrict__ tab,
									       const int16_t *D)
{
	constexpr int b4 = tu_lds_bytes<4, TU_INTRA>(), b8 = tu_lds_bytes<8, TU_INTRA>(), b16 = tu_lds_bytes<16, TU_INTRA>(), bmax = tu_lds_bytes<MAXN, TU_INTRA>();
	constexpr int bytes = (b4 > b8 ? b4 : b8) > (b16 > bmax ? b16 : bmax) ? (b4 > b8 ? b4 : b8) : (b16 > bmax ? b16 : bmax);
	__shared__ __attribute__((aligned(16))) char lds[bytes];                              // the largest body's image
	int s = 0;
#pragma unroll
	for (int i = 1; i < HMR_GPU_MAX_SEGMENTS; i++)
		if (i < t.n && (int)blockIdx.x >= t.first[i]) s = i;
	const unsigned vb = blockIdx.x - (unsigned)t.first[s], vg = (unsigned)t.blocks[s];
	if (vb >= vg) return;
	const void *jobs = t.jobs[s];
	const int njobs = t.njobs[s], rounds = t.rounds[s];
	uint32_t *ssd = t.ssd[s];
	int32_t *ac = t.ac[s];
	const hmr_gpu_intra_result *modes = t.modes[s];
#define TU_BODY(N, MODE) tu_chain_body<N, MODE>(jobs, njobs, O, Pp, L, Rr, ssd, ac, tab, D, modes, rounds, lds, vb, vg)
	switch (t.size[s] * 4 + t.kind[s]) {
	case 4 * 4 + 0: TU_BODY(4, TU_GIVEN_PRED); break;
	case 4 * 4 + 1: TU_BODY(4, TU_INTRA); break;
	case 4 * 4 + 2: TU_BODY(4, TU_INTER); break;
	case 8 * 4 + 0: TU_BODY(8, TU_GIVEN_PRED); break;
	case 8 * 4 + 1: TU_BODY(8, TU_INTRA); break;
	case 8 * 4 + 2: TU_BODY(8, TU_INTER); break;
	case 16 * 4 + 0: TU_BODY(16, TU_GIVEN_PRED); break;
	case 16 * 4 + 1: TU_BODY(16, TU_INTRA); break;
	case 16 * 4 + 2: TU_BODY(16, TU_INTER); break;
	default:
		if constexpr (MAXN == 32) {
			switch (t.kind[s]) {
			case 0: TU_BODY(32, TU_GIVEN_PRED); break;
			case 1: TU_BODY(32, TU_INTRA); break;
			default: TU_BODY(32, TU_INTER); break;
			}
		}
		break;
	}
#undef TU_BODY
}

}  // namespace

extern "C" int hmr_gpu_tu_chain_multi(hmr_gpu_ctx *ctx, const hmr_gpu_tu_segment *segs, int nseg, const int16_t *orig_base, const int16_t *decoded_base,
				      int16_t *pred_base, int16_t *level_base, int16_t *recon_base)
{
	if (nseg <= 0) return HMR_GPU_OK;
	if (nseg > HMR_GPU_MAX_SEGMENTS) { hmr_set_error("tu_chain_multi: at most %d segments", HMR_GPU_MAX_SEGMENTS); return HMR_GPU_ERR_ARG; }
	TuSegTab t = {};
	int next = 0, maxn = 0;
	for (int i = 0; i < nseg; i++) {
		const int n = segs[i].size;
		if (segs[i].njobs <= 0) continue;
		if ((n != 4 && n != 8 && n != 16 && n != 32) || segs[i].kind < 0 || segs[i].kind > 2) {
			hmr_set_error("tu_chain_multi: TU size must be 4, 8, 16 or 32 and kind 0 (given prediction), 1 (intra) or 2 (inter)");
			return HMR_GPU_ERR_ARG;
		}
		const int jpb = (HMR_WAVE / n) * HMR_WAVES_PER_BLOCK;
		const int blocks = hmr_grid_for_units(((long)segs[i].njobs + jpb - 1) / jpb);
		t.jobs[t.n] = segs[i].jobs; t.ssd[t.n] = segs[i].ssd; t.ac[t.n] = segs[i].ac_sum; t.modes[t.n] = segs[i].modes;
		t.njobs[t.n] = segs[i].njobs; t.size[t.n] = n; t.kind[t.n] = segs[i].kind; t.rounds[t.n] = segs[i].rounds > 1 ? segs[i].rounds : 1;
		t.first[t.n] = next; t.blocks[t.n] = blocks;
		next = (next + blocks + HMR_XCDS - 1) / HMR_XCDS * HMR_XCDS;
		maxn = n > maxn ? n : maxn;
		t.n++;
	}
	if (!t.n) return HMR_GPU_OK;
	const dim3 grid(t.first[t.n - 1] + t.blocks[t.n - 1]), block(HMR_BLOCK);
	if (maxn == 32)
		hipLaunchKernelGGL((k_tu_chain_multi<32>), grid, block, 0, ctx->stream, t, orig_base, pred_base, level_base, recon_base, ctx->tables, decoded_base);
	else
		hipLaunchKernelGGL((k_tu_chain_multi<16>), grid, block, 0, ctx->stream, t, orig_base, pred_base, level_base, recon_base, ctx->tables, decoded_base);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

template <int MODE>
static int launch_tu_chain(hmr_gpu_ctx *ctx, const void *jobs, int njobs, int size, const int16_t *orig_base, int16_t *pred_base, int16_t *level_base,
			   int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum, const int16_t *decoded_base, const hmr_gpu_intra_result *modes = nullptr, int rounds = 1)
{
	if (njobs <= 0) return HMR_GPU_OK;
#define TU_LAUNCH(N)                                                                                                                               \
	hipLaunchKernelGGL((k_tu_chain<N, MODE>), dim3(hmr_grid_for_units(((long)njobs + (HMR_WAVE / N) * HMR_WAVES_PER_BLOCK - 1) / ((HMR_WAVE / N) * HMR_WAVES_PER_BLOCK))), \
			   dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, orig_base, pred_base, level_base, recon_base, ssd, ac_sum, ctx->tables, decoded_base, modes, rounds)
	switch (size) {
	case 4: TU_LAUNCH(4); break;
	case 8: TU_LAUNCH(8); break;
	case 16: TU_LAUNCH(16); break;
	case 32: TU_LAUNCH(32); break;
	default: hmr_set_error("tu_chain_batch: TU size must be 4, 8, 16 or 32"); return HMR_GPU_ERR_ARG;
	}
#undef TU_LAUNCH
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_tu_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *pred_base,
				      int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum)
{
	return launch_tu_chain<TU_GIVEN_PRED>(ctx, jobs, njobs, size, orig_base, const_cast<int16_t *>(pred_base), level_base, recon_base, ssd, ac_sum, nullptr);
}

extern "C" int hmr_gpu_intra_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int size, const int16_t *orig_base,
					    const int16_t *decoded_base, int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd,
					    int32_t *ac_sum)
{
	return launch_tu_chain<TU_INTRA>(ctx, jobs, njobs, size, orig_base, pred_base, level_base, recon_base, ssd, ac_sum, decoded_base);
}

extern "C" int hmr_gpu_intra_tu_chain_modes_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int size, const int16_t *orig_base,
						  const int16_t *decoded_base, int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd,
						  int32_t *ac_sum, const hmr_gpu_intra_result *modes)
{
	if (!modes) { hmr_set_error("intra_tu_chain_modes_batch: modes is NULL"); return HMR_GPU_ERR_ARG; }
	return launch_tu_chain<TU_INTRA>(ctx, jobs, njobs, size, orig_base, pred_base, level_base, recon_base, ssd, ac_sum, decoded_base, modes);
}

extern "C" int hmr_gpu_intra_tu_chain_rounds_batch(hmr_gpu_ctx *ctx, const hmr_gpu_itu_job *jobs, int njobs, int rounds, int size, const int16_t *orig_base,
						   const int16_t *decoded_base, int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd,
						   int32_t *ac_sum, const hmr_gpu_intra_result *modes)
{
	if (rounds < 1 || rounds > 16) { hmr_set_error("intra_tu_chain_rounds_batch: rounds must be 1..16"); return HMR_GPU_ERR_ARG; }
	return launch_tu_chain<TU_INTRA>(ctx, jobs, njobs, size, orig_base, pred_base, level_base, recon_base, ssd, ac_sum, decoded_base, modes, rounds);
}

extern "C" int hmr_gpu_inter_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_inter_tu_job *jobs, int njobs, int size, const int16_t *residual_base,
					    const int16_t *pred_base, int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum)
{
	return launch_tu_chain<TU_INTER>(ctx, jobs, njobs, size, residual_base, const_cast<int16_t *>(pred_base), level_base, recon_base, ssd, ac_sum, nullptr);
}

