// SAO offset derivation from the CTU statistics: sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion (hmr_sao.c:480-659), 8-bit.
// One wavefront per (CTU, component): lanes 0-31 own the 32 band classes, lanes 32-51 the 4 x 5 edge classes, so the initial offset (a double division,
// rounded half away from zero), the sign rule and the iterative refinement (est_iter_offset, :445: dist + lambda * rate in IEEE double, strict <) run
// once per class in parallel; the band position is the first minimum of the four-band cost sums, added in the reference's order.
#include "common.h"

namespace {

__global__ __launch_bounds__(HMR_BLOCK) void k_sao_offsets(const int32_t *__restrict__ stats, int njobs, const double *__restrict__ lambdas, int32_t *__restrict__ offsets,
							    int32_t *__restrict__ aux, long long *__restrict__ dist)
{
	__shared__ double sCost[HMR_WAVES_PER_BLOCK][32];
	__shared__ long long sDist[HMR_WAVES_PER_BLOCK][64];
	__shared__ int sBand[HMR_WAVES_PER_BLOCK];
	const int lane = lane_id(), w = wave_in_block();
	const JobRange jr = xcd_job_range(njobs, HMR_WAVES_PER_BLOCK);
	for (long j = jr.begin + w; j < jr.end; j += jr.stride) {      // j = ctu * 3 + component
		const bool bo = lane < 32, eo = lane >= 32 && lane < 52;
		const int type = bo ? 4 : (lane - 32) / 5, cls = bo ? lane : (lane - 32) % 5;
		const int32_t *st = stats + ((size_t)j * 5 + type) * 64;
		const double lambda = lambdas[j];
		int q = 0;
		long long d = 0;
		double cost = lambda;
		if (bo || eo) {
			const long long df = st[cls], cn = st[32 + cls];
			if (cn != 0 && (bo || cls != 2)) {
				const double x = (double)df / (double)cn;
				int v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
				v = v < -7 ? -7 : v > 7 ? 7 : v;
				if (eo && ((cls < 2 && v < 0) || (cls > 2 && v > 0))) v = 0;
				// est_iter_offset: towards zero, keep the cheapest; an offset that never beats lambda alone becomes 0
				double min_cost = lambda;
				for (int it = v; it != 0; it = it > 0 ? it - 1 : it + 1) {
					const int a = it < 0 ? -it : it;
					const long long rate = (bo ? a + 2 : a + 1) - (a == 7 ? 1 : 0);
					const long long dd = cn * it * it - df * it * 2;
					const double c = (double)dd + lambda * (double)rate;
					if (c < min_cost) { min_cost = c; q = it; d = dd; cost = c; }
				}
			}
		}
		if (bo) sCost[w][lane] = cost;
		sDist[w][lane] = d;
		wave_sync();
		if (lane == 0) {
			double min_cost = (double)(0xffffffffu / 8);
			int band = 0;
			for (int i = 0; i < 29; i++) {
				double s = sCost[w][i];
				s += sCost[w][i + 1]; s += sCost[w][i + 2]; s += sCost[w][i + 3];
				if (s < min_cost) { min_cost = s; band = i; }
			}
			sBand[w] = band;
		}
		wave_sync();
		const int band = sBand[w];
		int32_t *o = offsets + (size_t)j * 5 * 32;
		if (bo) o[4 * 32 + lane] = (lane >= band && lane < band + 4) ? q : 0;
		else if (eo) o[type * 32 + cls] = q;
		for (int e = lane; e < 4 * 27; e += HMR_WAVE) o[(e / 27) * 32 + 5 + e % 27] = 0;      // entries 5..31 of the edge types
		if (lane < 5) {
			long long s = 0;
			if (lane == 4) for (int i = band; i < band + 4; i++) s += sDist[w][i];
			else for (int c = 0; c < 5; c++) s += sDist[w][32 + lane * 5 + c];
			dist[(size_t)j * 5 + lane] = s;
			aux[(size_t)j * 5 + lane] = lane == 4 ? band : 0;
		}
		wave_sync();
	}
}

}  // namespace

extern "C" int hmr_gpu_sao_offsets_frame(hmr_gpu_ctx *ctx, const int32_t *stats, int n_ctu, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist)
{
	if (n_ctu <= 0) return HMR_GPU_OK;
	hipLaunchKernelGGL(k_sao_offsets, dim3(hmr_grid_for_waves(3L * n_ctu)), dim3(HMR_BLOCK), 0, ctx->stream, stats, 3 * n_ctu, lambdas, offsets, aux, (long long *)dist);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
