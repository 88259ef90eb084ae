// Batched sub-pel interpolation: 8-tap luma / 4-tap chroma separable FIR with the HM stage rules.
// Reference semantics: hmr_sse42_functions_inter_prediction.c:796,818 (scalar spec hmr_motion_inter.c:262-391,878).
//
// One wave per job; consecutive lanes produce consecutive samples of an output row, so the tap reads of a
// wave are overlapping 128-byte row segments served by L1/L2.  Exactly `width` columns are written (the SSE
// code overshoots to a multiple of 8).
#include "common.h"

namespace {

__constant__ int16_t cLuma[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int16_t cChroma[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
				      {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

template <int TAPS>
__global__ __launch_bounds__(HMR_BLOCK) void k_interpolate(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							      int16_t *__restrict__ Cc)
{
	const int lane = lane_id();
	const long wave = (long)blockIdx.x * HMR_WAVES_PER_BLOCK + wave_in_block();
	const long nwaves = (long)gridDim.x * HMR_WAVES_PER_BLOCK;
	for (long j = wave; j < njobs; j += nwaves) {
		const hmr_gpu_job jb = jobs[j];
		const int w = jb.w, h = jb.h, frac = (int)jb.p0;
		const bool vert = jb.p1 & 1, first = jb.p1 & 2, last = jb.p1 & 4;
		const int ss = (int)jb.a_stride, ds = (int)jb.c_stride;
		const int16_t *src = A + jb.a_off;
		int16_t *dst = Cc + jb.c_off;
		if (frac == 0) {
			if (TAPS == 4 && w < 4) continue;   // hmr_sse42_functions_inter_prediction.c:822-825: silent no-op
			for (int e = lane; e < w * h; e += HMR_WAVE) {
				const int y = e / w, x = e - y * w;
				const int v = src[(size_t)y * ss + x];
				int r;
				if (first == last) r = v;
				else if (first) r = (int16_t)((int16_t)(v << 6) - 8192);
				else r = clip3i((v + 8192 + 32) >> 6, 0, 255);
				dst[(size_t)y * ds + x] = (int16_t)r;
			}
			continue;
		}
		int c[TAPS];
#pragma unroll
		for (int t = 0; t < TAPS; t++) c[t] = TAPS == 8 ? cLuma[frac][t] : cChroma[frac][t];
		const int rs = vert ? ss : 1;
		int shift = 6, offset;
		if (last) {
			shift += first ? 0 : 6;
			offset = (1 << (shift - 1)) + (first ? 0 : 8192 << 6);
		} else {
			shift -= first ? 6 : 0;
			offset = first ? -(8192 << shift) : 0;
		}
		const int16_t *s0 = src - (TAPS / 2 - 1) * rs;
		for (int e = lane; e < w * h; e += HMR_WAVE) {
			const int y = e / w, x = e - y * w;
			const int16_t *p = s0 + (size_t)y * ss + x;
			int sum = 0;
#pragma unroll
			for (int t = 0; t < TAPS; t++) sum += p[t * rs] * c[t];
			int v = sat16i((sum + offset) >> shift);
			if (last) v = clip3i(v, 0, 255);
			dst[(size_t)y * ds + x] = (int16_t)v;
		}
	}
}

}  // namespace

extern "C" int hmr_gpu_interpolate_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int is_luma, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	dim3 grid(hmr_grid_for_waves(njobs)), block(HMR_BLOCK);
	if (is_luma) hipLaunchKernelGGL((k_interpolate<8>), grid, block, 0, ctx->stream, jobs, njobs, a, c);
	else hipLaunchKernelGGL((k_interpolate<4>), grid, block, 0, ctx->stream, jobs, njobs, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
