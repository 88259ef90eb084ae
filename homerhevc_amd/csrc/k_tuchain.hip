// Fused transform-unit chain: residual -> forward transform -> quantisation (+ sign hiding) -> [dequantisation ->
// inverse transform] -> reconstruction -> SSD, one launch for a batch of TUs.
// This is the per-TU sequence of the reference's encode_intra_cu (hmr_motion_intra.c:1014-1069) and encode_inter_cu
// (hmr_motion_inter.c:40-230): seven table calls (predict, transform, quant, inv_quant, itransform, reconst, ssd16b) whose
// intermediates the CPU keeps in scratch windows.  Here a TU lives in the LDS region of G lanes from the first load to the
// last store: source and prediction are read once, levels, reconstruction, SSD and ac_sum are written once, and the residual,
// coefficient and de-quantised planes never reach HBM.  Each stage is the same arithmetic as the stand-alone kernels
// (k_transform.hip), so the outputs are bit-identical to the seven calls in sequence.
#include "common.h"
#include "tq_device.h"

namespace {

template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_tu_chain(const hmr_gpu_tu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ O,
							   const int16_t *__restrict__ Pp, int16_t *__restrict__ L, int16_t *__restrict__ Rr,
							   uint32_t *__restrict__ ssd_out, int32_t *__restrict__ ac_out, const DevTables *__restrict__ tab)
{
	using g = Geo<N>;
	__shared__ int16_t sM[2][N * N];
	__shared__ int16_t sA[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];      // residual, later de-quantised coefficients
	__shared__ int16_t sT[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];      // stage intermediates
	__shared__ int16_t sOrig[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sPred[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sCoef[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sLev[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sDu[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ unsigned long long sNzMask[HMR_WAVES_PER_BLOCK][g::JPW];
	const int lane = lane_id(), w = wave_in_block(), sub = lane / g::G, l = lane % g::G;
	constexpr int CG_PER_IT = g::G / 16, SIDE = N / 4;
	constexpr int sh1 = g::L2 - 1, sh2 = g::L2 + 6;
	for (int i = threadIdx.x; i < N * N; i += HMR_BLOCK) {
		sM[0][i] = tab->dct[g::L2 - 2][i];
		sM[1][i] = N == 4 ? tab->dst4[i] : (int16_t)0;
	}
	__syncthreads();
	int16_t *tA = sA[w][sub], *tT = sT[w][sub], *orig = sOrig[w][sub], *pred = sPred[w][sub], *coef = sCoef[w][sub], *lev = sLev[w][sub], *du = sDu[w][sub];
	const JobRange jr = xcd_job_range(njobs, g::JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * g::JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_tu_job jb = {};
		if (l == 0) sNzMask[w][sub] = 0;
		if (ok) {
			jb = jobs[j];
			const int16_t *o = O + jb.orig_off, *p = Pp + jb.pred_off;
			for (int e = l; e < g::E; e += g::G) {      // K3 predict: res = orig - pred
				const int y = e / N, x = e % N;
				const int vo = o[(size_t)y * jb.orig_stride + x], vp = p[(size_t)y * jb.pred_stride + x];
				orig[e] = (int16_t)vo;
				pred[e] = (int16_t)vp;
				tA[y * g::P + x] = (int16_t)(vo - vp);
			}
		}
		wave_sync();
		const int16_t *M = sM[(N == 4 && ((jb.p0 >> 7) & 1)) ? 1 : 0];
		if (ok)                                             // K12 stage 1
			for (int o = l; o < g::E; o += g::G) {
				const int k = o / N, row = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[k * N + i] * tA[row * g::P + i];
				tT[k * g::P + row] = (int16_t)sat16i((s + (1 << (sh1 - 1))) >> sh1);
			}
		wave_sync();
		if (ok)                                             // K12 stage 2
			for (int o = l; o < g::E; o += g::G) {
				const int k2 = o / N, k1 = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[k2 * N + i] * tT[k1 * g::P + i];
				coef[o] = (int16_t)sat16i((s + (1 << (sh2 - 1))) >> sh2);
			}
		wave_sync();
		// K14 quant
		int ac = 0;
		bool sbh = false;
		const uint32_t *scan = tab->scan[3][g::L2];
		const int comp = (jb.p0 >> 2) & 3, is_intra = (jb.p0 >> 4) & 1, per = jb.p1 & 0xff, rem = (jb.p1 >> 8) & 0xff;
		if (ok) {
			const int scan_mode = jb.p0 & 3, slice_i = (jb.p0 >> 5) & 1;
			sbh = (jb.p0 >> 6) & 1;
			const int32_t *q = tab->quant[g::L2 - 2][(is_intra ? 0 : 3) + comp][rem];
			const uint8_t *b2c = tab->blk2cg[scan_mode][g::L2];
			scan = tab->scan[scan_mode][g::L2];
			const int qbits = 14 + per + (7 - g::L2), qbits8 = qbits - 8;
			const uint32_t add = (uint32_t)(slice_i ? 171 : 85) << (qbits - 9);
			uint32_t sum = 0;
			unsigned long long nz = 0;
			for (int e = l; e < g::E; e += g::G) {
				const int s = coef[e];
				const uint32_t mag = (uint16_t)(s < 0 ? -s : s);
				const uint32_t aux = mag * (uint32_t)q[e];
				const int c = (int)(aux + add) >> qbits;
				const int d = (int)(aux - ((uint32_t)c << qbits)) >> qbits8;
				sum += (uint32_t)c;
				const int sgn = s > 0 ? 1 : (s < 0 ? -1 : 0);
				const int lv = (int16_t)(sgn * sat16i(c));
				lev[e] = (int16_t)lv;
				du[e] = (int16_t)sat16i(d);
				if (lv) nz |= 1ull << b2c[((e / N) >> 2) * SIDE + ((e % N) >> 2)];
			}
			ac = (int)group_sum<g::G>(sum);
			if (nz) atomicOr(&sNzMask[w][sub], nz);
		}
		wave_sync();
		{
			const bool run_sbh = ok && sbh && ac >= 2;
			unsigned long long m = run_sbh ? sNzMask[w][sub] : 0ull;
			const int last = m ? 63 - __clzll((long long)m) : -1;
			const int grp = l >> 4;
			while (__any(m != 0)) {
				unsigned long long t = m;
				int cg = -1;
#pragma unroll
				for (int k = 0; k < CG_PER_IT; k++) {
					const int b = t ? __ffsll((long long)t) - 1 : -1;
					if (k == grp) cg = b;
					t &= t - 1;
				}
				m = t;
				sbh_group16(lev, coef, du, scan, cg < 0 ? 0 : cg, cg == last, run_sbh && cg >= 0);
			}
		}
		wave_sync();
		const bool coded = ok && ac != 0;                   // the reference skips dequant / inverse transform for all-zero TUs
		if (ok) {
			int16_t *lo = L + jb.lev_off;
			for (int e = l; e < g::E; e += g::G) lo[e] = lev[e];
			if (l == 0) ac_out[j] = ac;
		}
		if (coded) {                                        // K15 inv_quant into the pitched tile
			const int32_t *iq = tab->dequant[g::L2 - 2][is_intra ? 0 : 3 + comp][rem];
			const int iq_shift = 3 + g::L2;
			for (int e = l; e < g::E; e += g::G) {
				const uint32_t prod = (uint32_t)(int)lev[e] * (uint32_t)iq[e];
				int r;
				if (iq_shift > per) r = (int)(prod + (1u << (iq_shift - per - 1))) >> (iq_shift - per);
				else r = (int)(prod << (per - iq_shift));
				tA[(e / N) * g::P + (e % N)] = (int16_t)sat16i(r);
			}
		}
		wave_sync();
		if (coded)                                          // K13 stage 1
			for (int o = l; o < g::E; o += g::G) {
				const int k = o / N, col = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[i * N + k] * tA[i * g::P + col];
				tT[col * g::P + k] = (int16_t)sat16i((s + 64) >> 7);
			}
		wave_sync();
		uint32_t ssd = 0;
		if (ok) {                                           // K13 stage 2 + K4 reconst + K2 ssd
			int16_t *ro = Rr + jb.rec_off;
			for (int o = l; o < g::E; o += g::G) {
				const int y = o / N, x = o % N;
				int res = 0;
				if (coded) {
					int s = 0;
#pragma unroll
					for (int i = 0; i < N; i++) s += M[i * N + x] * tT[i * g::P + y];
					res = sat16i((s + 2048) >> 12);
				}
				const int rec = clip3i(sat16i(pred[o] + res), 0, 255);
				ro[(size_t)y * jb.rec_stride + x] = (int16_t)rec;
				const int d = (int16_t)(orig[o] - rec);
				ssd += (uint32_t)(d * d);
			}
		}
		ssd = group_sum<g::G>(ssd);
		if (ok && l == 0) ssd_out[j] = ssd;
		wave_sync();
	}
}

}  // namespace

extern "C" int hmr_gpu_tu_chain_batch(hmr_gpu_ctx *ctx, const hmr_gpu_tu_job *jobs, int njobs, int size, const int16_t *orig_base, const int16_t *pred_base,
				      int16_t *level_base, int16_t *recon_base, uint32_t *ssd, int32_t *ac_sum)
{
	if (njobs <= 0) return HMR_GPU_OK;
#define TU_LAUNCH(N)                                                                                                              \
	hipLaunchKernelGGL((k_tu_chain<N>), dim3(hmr_grid_for_units(((long)njobs + Geo<N>::JPB - 1) / Geo<N>::JPB)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, \
			   njobs, orig_base, pred_base, level_base, recon_base, ssd, ac_sum, ctx->tables)
	switch (size) {
	case 4: TU_LAUNCH(4); break;
	case 8: TU_LAUNCH(8); break;
	case 16: TU_LAUNCH(16); break;
	case 32: TU_LAUNCH(32); break;
	default: hmr_set_error("tu_chain_batch: TU size must be 4, 8, 16 or 32"); return HMR_GPU_ERR_ARG;
	}
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
