// Batched sub-pel interpolation: 8-tap luma / 4-tap chroma separable FIR with the HM stage rules.
// Reference semantics: hmr_sse42_functions_inter_prediction.c:796,818 (scalar spec hmr_motion_inter.c:262-391,878).
//
// One wave per job; consecutive lanes produce consecutive samples of an output row, so the tap reads of a
// wave are overlapping 128-byte row segments served by L1/L2.  Exactly `width` columns are written (the SSE
// code overshoots to a multiple of 8).
#include "common.h"
#include "vec.h"

namespace {

__constant__ int16_t cLuma[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int16_t cChroma[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
				      {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

__device__ __forceinline__ int finish(int sum, int offset, int shift, bool last)
{
	int v = sat16i((sum + offset) >> shift);
	return last ? clip3i(v, 0, 255) : v;
}

// One wave per job; a lane produces 4 consecutive samples of an output row from 8-byte loads: 2 loads + 3 scalars
// (8-tap, horizontal) or TAPS loads (vertical) instead of 4*TAPS scalar loads.  Row tails (w % 4) take the scalar path.
template <int TAPS>
__global__ __launch_bounds__(HMR_BLOCK) void k_interpolate(const hmr_gpu_job *__restrict__ jobs, int njobs, int lanes_per_job,
							      const int16_t *__restrict__ A, int16_t *__restrict__ Cc)
{
	// lanes_per_job (16, 32 or 64): small blocks share a wave (an 8x8 block is only 16 four-sample chunks)
	const int G = lanes_per_job, JPW = HMR_WAVE / G;
	const int sub = lane_id() / G, lane = lane_id() % G;
	const JobRange jr = xcd_job_range(njobs, JPW * HMR_WAVES_PER_BLOCK);
	for (long j0 = jr.begin + wave_in_block() * JPW; j0 < jr.end; j0 += jr.stride) {
		const long j = j0 + sub;
		if (j >= jr.end) continue;
		const hmr_gpu_job jb = jobs[j];
		const int w = jb.w, h = jb.h, frac = (int)jb.p0;
		const bool vert = jb.p1 & 1, first = jb.p1 & 2, last = jb.p1 & 4;
		const int ss = (int)jb.a_stride, ds = (int)jb.c_stride;
		const int16_t *src = A + jb.a_off;
		int16_t *dst = Cc + jb.c_off;
		const int cpr = (w + 3) >> 2, total = cpr * h;
		if (frac == 0) {
			if (TAPS == 4 && w < 4) continue;   // hmr_sse42_functions_inter_prediction.c:822-825: silent no-op
			for (int e = lane; e < total; e += G) {
				const int y = e / cpr, x = (e - y * cpr) * 4, nv = w - x < 4 ? w - x : 4;
				for (int k = 0; k < nv; k++) {
					const int v = src[(size_t)y * ss + x + k];
					int r;
					if (first == last) r = v;
					else if (first) r = (int16_t)((int16_t)(v << 6) - 8192);
					else r = clip3i((v + 8192 + 32) >> 6, 0, 255);
					dst[(size_t)y * ds + x + k] = (int16_t)r;
				}
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
		for (int e = lane; e < total; e += G) {
			const int y = e / cpr, x = (e - y * cpr) * 4, nv = w - x < 4 ? w - x : 4;
			const int16_t *p = s0 + (size_t)y * ss + x;
			int16_t *o = dst + (size_t)y * ds + x;
			if (nv == 4) {
				int sum[4] = {0, 0, 0, 0};
				if (vert) {
#pragma unroll
					for (int t = 0; t < TAPS; t++) {
						const i16x4 r = ld4(p + (size_t)t * ss);
#pragma unroll
						for (int k = 0; k < 4; k++) sum[k] += r.v[k] * c[t];
					}
				} else {
					int in[TAPS + 3];
					const i16x4 v0 = ld4(p);
#pragma unroll
					for (int k = 0; k < 4; k++) in[k] = v0.v[k];
					if (TAPS == 8) {
						const i16x4 v1 = ld4(p + 4);
#pragma unroll
						for (int k = 0; k < 4; k++) in[4 + k] = v1.v[k];
					}
#pragma unroll
					for (int k = TAPS; k < TAPS + 3; k++) in[k] = p[k];
#pragma unroll
					for (int k = 0; k < 4; k++)
#pragma unroll
						for (int t = 0; t < TAPS; t++) sum[k] += in[k + t] * c[t];
				}
				i16x4 r;
#pragma unroll
				for (int k = 0; k < 4; k++) r.v[k] = (int16_t)finish(sum[k], offset, shift, last);
				st4(o, r);
			} else {
				for (int k = 0; k < nv; k++) {
					int sum = 0;
#pragma unroll
					for (int t = 0; t < TAPS; t++) sum += p[k + t * rs] * c[t];
					o[k] = (int16_t)finish(sum, offset, shift, last);
				}
			}
		}
	}
}

}  // namespace

extern "C" int hmr_gpu_interpolate_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int flags, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	const int is_luma = flags & 1;
	int g = (flags >> 8) & 0xff;   // lanes per job hint: 16 / 32 / 64 (0 = 64)
	if (g != 16 && g != 32) g = 64;
	const int jpw = HMR_WAVE / g;
	dim3 grid(hmr_grid_for_waves(((long)njobs + jpw - 1) / jpw)), block(HMR_BLOCK);
	if (is_luma) hipLaunchKernelGGL((k_interpolate<8>), grid, block, 0, ctx->stream, jobs, njobs, g, a, c);
	else hipLaunchKernelGGL((k_interpolate<4>), grid, block, 0, ctx->stream, jobs, njobs, g, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
