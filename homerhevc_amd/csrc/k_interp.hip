// Batched sub-pel interpolation: 8-tap luma / 4-tap chroma separable FIR with the HM stage rules.
// Reference semantics: hmr_sse42_functions_inter_prediction.c:796,818 (scalar spec hmr_motion_inter.c:262-391,878).
//
// Exactly `width` columns are written (the SSE code overshoots to a multiple of 8) and no sample
// outside the taps' footprint is read.
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

// Work item = a register window, so that every input sample is loaded once per item instead of once per tap:
//   vertical   : 4 columns x 4 output rows  -> 4+TAPS-1 row loads of 8 bytes feed 16 outputs
//   horizontal : 8 consecutive outputs      -> TAPS+7 samples (8-byte loads) feed 8 outputs
// `lanes_per_job` lanes share a job (a 16x16 vertical job is 16 items), so small blocks fill a wave together.
template <int TAPS>
__global__ __launch_bounds__(HMR_BLOCK) void k_interpolate(const hmr_gpu_job *__restrict__ jobs, int njobs, int lanes_per_job,
							      const int16_t *__restrict__ A, int16_t *__restrict__ Cc)
{
	const int G = lanes_per_job, JPW = HMR_WAVE / G;
	const int sub = lane_id() / G, lane = lane_id() % G;
	const JobRange jr = xcd_job_range(njobs, JPW * HMR_WAVES_PER_BLOCK);
	for (long j0 = jr.begin + wave_in_block() * JPW; j0 < jr.end; j0 += jr.stride) {
		const long j = j0 + sub;
		if (j >= jr.end) continue;
		const hmr_gpu_job jb = JPW == 1 ? load_job_uniform(jobs, j) : jobs[j];
		const int w = jb.w, h = jb.h, frac = (int)jb.p0;
		const bool vert = jb.p1 & 1, first = jb.p1 & 2, last = jb.p1 & 4;
		const int ss = (int)jb.a_stride, ds = (int)jb.c_stride;
		const int16_t *src = A + jb.a_off;
		int16_t *dst = Cc + jb.c_off;
		if (frac == 0) {
			if (TAPS == 4 && w < 4) continue;   // hmr_sse42_functions_inter_prediction.c:822-825: silent no-op
			const int cpr = (w + 3) >> 2, total = cpr * h;
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
		int shift = 6, offset;
		if (last) {
			shift += first ? 0 : 6;
			offset = (1 << (shift - 1)) + (first ? 0 : 8192 << 6);
		} else {
			shift -= first ? 6 : 0;
			offset = first ? -(8192 << shift) : 0;
		}
		if (vert) {
			const int cpr = (w + 3) >> 2, segs = (h + 3) >> 2, total = cpr * segs;
			const int16_t *s0 = src - (TAPS / 2 - 1) * ss;
			for (int e = lane; e < total; e += G) {
				const int sg = e / cpr, x = (e - sg * cpr) * 4, y0 = sg * 4;
				const int nv = w - x < 4 ? w - x : 4, nr = h - y0 < 4 ? h - y0 : 4;
				int sum[4][4];
#pragma unroll
				for (int r = 0; r < 4; r++)
#pragma unroll
					for (int k = 0; k < 4; k++) sum[r][k] = 0;
#pragma unroll
				for (int t = 0; t < TAPS + 3; t++) {
					if (t < nr + TAPS - 1) {
						const int16_t *p = s0 + (size_t)(y0 + t) * ss + x;
						i16x4 v;
						if (nv == 4) v = ld4(p);
						else {
#pragma unroll
							for (int k = 0; k < 4; k++) v.v[k] = k < nv ? p[k] : (int16_t)0;
						}
#pragma unroll
						for (int r = 0; r < 4; r++)
							if (t - r >= 0 && t - r < TAPS) {
#pragma unroll
								for (int k = 0; k < 4; k++) sum[r][k] += v.v[k] * c[t - r];
							}
					}
				}
#pragma unroll
				for (int r = 0; r < 4; r++)
					if (r < nr) {
						int16_t *o = dst + (size_t)(y0 + r) * ds + x;
						if (nv == 4) {
							i16x4 q;
#pragma unroll
							for (int k = 0; k < 4; k++) q.v[k] = (int16_t)finish(sum[r][k], offset, shift, last);
							st4(o, q);
						} else {
							for (int k = 0; k < nv; k++) o[k] = (int16_t)finish(sum[r][k], offset, shift, last);
						}
					}
			}
		} else {
			const int spr = (w + 7) >> 3, total = spr * h;
			const int16_t *s0 = src - (TAPS / 2 - 1);
			for (int e = lane; e < total; e += G) {
				const int y = e / spr, x = (e - y * spr) * 8, nv = w - x < 8 ? w - x : 8;
				const int16_t *p = s0 + (size_t)y * ss + x;
				int16_t *o = dst + (size_t)y * ds + x;
				int in[TAPS + 7];
				const int need = nv + TAPS - 1;                 // samples this item may touch
#pragma unroll
				for (int q = 0; q < (TAPS + 7 + 3) / 4; q++) {
					if (4 * q + 4 <= need) {
						const i16x4 v = ld4(p + 4 * q);
#pragma unroll
						for (int k = 0; k < 4; k++)
							if (4 * q + k < TAPS + 7) in[4 * q + k] = v.v[k];
					} else {
#pragma unroll
						for (int k = 0; k < 4; k++)
							if (4 * q + k < TAPS + 7) in[4 * q + k] = 4 * q + k < need ? p[4 * q + k] : 0;
					}
				}
				int res[8];
#pragma unroll
				for (int k = 0; k < 8; k++) {
					int sm = 0;
#pragma unroll
					for (int t = 0; t < TAPS; t++) sm += in[k + t] * c[t];
					res[k] = finish(sm, offset, shift, last);
				}
				if (nv == 8) {
					i16x4 q0, q1;
#pragma unroll
					for (int k = 0; k < 4; k++) { q0.v[k] = (int16_t)res[k]; q1.v[k] = (int16_t)res[4 + k]; }
					st4(o, q0);
					st4(o + 4, q1);
				} else {
#pragma unroll
					for (int k = 0; k < 8; k++)
						if (k < nv) o[k] = (int16_t)res[k];
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
	int g = (flags >> 8) & 0xff;   // lanes per job hint: 4 / 8 / 16 / 32 / 64 (0 = 64)
	if (g != 4 && g != 8 && g != 16 && g != 32) g = 64;
	const int jpw = HMR_WAVE / g;
	dim3 grid(hmr_grid_for_waves(((long)njobs + jpw - 1) / jpw)), block(HMR_BLOCK);
	if (is_luma) hipLaunchKernelGGL((k_interpolate<8>), grid, block, 0, ctx->stream, jobs, njobs, g, a, c);
	else hipLaunchKernelGGL((k_interpolate<4>), grid, block, 0, ctx->stream, jobs, njobs, g, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
