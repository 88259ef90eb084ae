// Chroma mode search of an intra CU: the candidate loop of encode_intra_chroma (hmr_motion_intra_chroma.c:176-233, non-HM path, rd_mode != RD_FULL).
//
// Five candidates - planar, vertical, horizontal, DC, DM (the luma mode; a list entry equal to the luma mode is replaced by 34, create_chroma_dir_list :92) -
// are each predicted for U and V from the UNFILTERED neighbours of the CU (is_luma = 0: no edge filters) and compared by SAD.  The reference issues
// 10 x {fill_reference_samples, create_intra_*_prediction, sad} through the table per CU; here G = min(64, N*N) lanes own the CU for the whole loop, both
// neighbour arrays live in LDS, predictions are never written out and the source samples of both components stay in registers.
// cost = dU + (dU + dV) (the running distortion is added once per component, :211-213) + (uint32)(bits * sqrt_lambda + .5), bits = 1 for DM, 12 otherwise;
// the winner is the first minimum (homer_update_cand_list, hmr_motion_intra.c:893, displaces on strict >).
#include "intra_device.h"

namespace {

template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_chroma_search(const hmr_gpu_chroma_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ O,
								const int16_t *__restrict__ D, const hmr_gpu_intra_result *__restrict__ luma,
								hmr_gpu_intra_result *__restrict__ out)
{
	constexpr int E = N * N, G = N == 4 ? 4 : N == 8 ? 16 : HMR_WAVE;
	constexpr int JPB = HMR_BLOCK / G, PPL = E / G, YSTEP = G / N;
	constexpr int l2 = N == 4 ? 2 : N == 8 ? 3 : 4, total = 4 * N + 1;
	__shared__ int16_t sAdi[JPB][2][total + 3];       // U, V
	__shared__ int16_t sMainBuf[JPB][3 * N + 2];
	const int tid = threadIdx.x, sub = tid / G, l = tid % G;
	int16_t *mainr = sMainBuf[sub] + N;
	const int x = l & (N - 1), y0 = l >> l2;
	const JobRange jr = xcd_job_range(njobs, JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + sub;
		const bool ok = j < jr.end;
		hmr_gpu_chroma_job jb = {};
		if (ok) {
			jb = jobs[j];
			const bool left = jb.flags & 1, top = jb.flags & 2, bl = jb.flags & 4, tr = jb.flags & 8;
			intra_build_refs<N, G>(sAdi[sub][0], D + jb.dec_u_off, (int)jb.dec_stride, left, top, bl ? (int)(jb.sizes & 0xffff) : 0, tr ? (int)(jb.sizes >> 16) : 0, l);
			intra_build_refs<N, G>(sAdi[sub][1], D + jb.dec_v_off, (int)jb.dec_stride, left, top, bl ? (int)(jb.sizes & 0xffff) : 0, tr ? (int)(jb.sizes >> 16) : 0, l);
		}
		wave_sync();
		int dc[2], og[2][PPL];
#pragma unroll
		for (int c = 0; c < 2; c++) {
			int dcs = 0;
			if (ok)
				for (int i = 1 + l; i <= N; i += G) dcs += sAdi[sub][c][2 * N + i] + sAdi[sub][c][2 * N - i];
			dc[c] = ((group_sum<G>(dcs) + N) / (2 * N)) & 0xff;
			const int16_t *org = O + (c ? jb.orig_v_off : jb.orig_u_off) + x;
			const int os = (int)jb.orig_stride;
#pragma unroll
			for (int i = 0; i < PPL; i++) og[c][i] = ok ? org[(size_t)(y0 + i * YSTEP) * os] : 0;
		}
		int luma_mode = (int)jb.luma_mode;
		if (ok && (jb.flags & 0x100u) && luma) luma_mode = luma[jb.luma_mode].best_mode & 0xff;       // handed over on the device by the luma search
		int best_coded = 0, best_mode = 0, best_bits = 0;
		unsigned best_cost = 0;
		bool have = false, replaced = false;
#pragma unroll
		for (int k = 0; k < 5; k++) {
			constexpr int base_list[5] = {0, 26, 10, 1, 36};
			int coded = base_list[k];
			if (k < 4 && !replaced && coded == luma_mode) { coded = 34; replaced = true; }
			const int mode = coded == 36 ? luma_mode : coded;
			const IntraMode m = intra_mode_setup(ok ? mode : 0);
			unsigned dist = 0, cost = 0;
#pragma unroll
			for (int c = 0; c < 2; c++) {
				const int16_t *mid = sAdi[sub][c] + 2 * N;
				if (ok) intra_fill_main<N, G>(m, mid, mainr, l);
				wave_sync();
				int s = 0;
				if (ok) {
#pragma unroll
					for (int i = 0; i < PPL; i++) {
						const int d = og[c][i] - intra_pixel<N>(m, mid, mainr, dc[c], false, x, y0 + i * YSTEP);
						s += d < 0 ? -d : d;
					}
				}
				s = group_sum<G>(s);
				wave_sync();
				dist += (unsigned)s;
				cost += dist;
			}
			const unsigned bits = coded == 36 ? 1u : 12u;
			cost += (unsigned)((double)bits * jb.sqrt_lambda + .5);
			if (!have || cost < best_cost) {
				have = true;
				best_cost = cost; best_coded = coded; best_mode = mode; best_bits = (int)bits;
			}
		}
		if (ok && l == 0) {
			hmr_gpu_intra_result r;
			r.best_mode = best_mode | (best_coded << 8);
			r.bits = best_bits;
			r.cost = (double)best_cost;
			out[j] = r;
		}
		wave_sync();
	}
}

}  // namespace

extern "C" int hmr_gpu_chroma_search_batch(hmr_gpu_ctx *ctx, const hmr_gpu_chroma_job *jobs, int njobs, int size, const int16_t *orig_base,
					    const int16_t *decoded_base, const hmr_gpu_intra_result *luma_modes, hmr_gpu_intra_result *out)
{
	if (njobs <= 0) return HMR_GPU_OK;
#define LAUNCH(NN, JPB) \
	hipLaunchKernelGGL((k_chroma_search<NN>), dim3(hmr_grid_for_units(((long)njobs + JPB - 1) / JPB)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, orig_base, \
			   decoded_base, luma_modes, out)
	switch (size) {
	case 4: LAUNCH(4, 64); break;
	case 8: LAUNCH(8, 16); break;
	case 16: LAUNCH(16, 4); break;
	default: hmr_set_error("chroma search: size must be 4, 8 or 16 (a 64x64 CU is searched on its first 16x16 chroma quadrant)"); return HMR_GPU_ERR_ARG;
	}
#undef LAUNCH
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
