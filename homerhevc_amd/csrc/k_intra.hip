// Batched intra prediction (planar / DC / 33 angular modes) and intra reference-sample construction.
// Reference semantics: hmr_sse42_functions_prediction.c:199,926 (scalar spec hmr_motion_intra.c:408-625),
// fill_reference_samples hmr_motion_intra.c:246 and adi_filter :189.
//
// One wave per block.  The 4N+1 neighbour samples are staged in LDS once, the projected main reference of
// the angular modes is built there in closed form (no serial running sums), and every lane then produces
// pixels so that consecutive lanes write consecutive samples of an output row, also for the horizontal
// modes the SSE code computes transposed.
#include "common.h"

namespace {

constexpr int kAng[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};                    // hmr_encoder_lib.c:35
constexpr int kInvAng[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};    // hmr_encoder_lib.c:36
__constant__ int cAng[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
__constant__ int cInvAng[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};

__global__ __launch_bounds__(HMR_BLOCK) void k_intra_pred(const hmr_gpu_job *__restrict__ jobs, int njobs, int N, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	__shared__ int16_t sAdi[HMR_WAVES_PER_BLOCK][4 * 64 + 1];
	__shared__ int16_t sMainBuf[HMR_WAVES_PER_BLOCK][3 * 64 + 2];   // main reference, index -N+1 .. 2N, origin at 64
	const int lane = lane_id(), w = wave_in_block();
	const int l2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
	int16_t *adi = sAdi[w];
	int16_t *mainr = sMainBuf[w] + 64;
	const int16_t *mid = adi + 2 * N;
	for (long base = (long)blockIdx.x * HMR_WAVES_PER_BLOCK; base < njobs; base += (long)gridDim.x * HMR_WAVES_PER_BLOCK) {
		const long j = base + w;
		const bool ok = j < njobs;
		hmr_gpu_job jb;
		if (ok) {
			jb = jobs[j];
			const int16_t *a = A + jb.a_off;
			for (int i = lane; i < 4 * N + 1; i += HMR_WAVE) adi[i] = a[i];
		}
		__syncthreads();
		const int mode = ok ? (int)jb.p0 : 0;
		const bool luma = ok && jb.p1 != 0;
		const bool is_hor = mode >= 2 && mode < 18, is_ver = mode >= 18;
		int angle = is_ver ? mode - 26 : is_hor ? -(mode - 10) : 0;
		int inv_angle = 0;
		if (mode >= 2) {
			const int aa = angle < 0 ? -angle : angle;
			inv_angle = cInvAng[aa];
			angle = angle < 0 ? -cAng[aa] : cAng[aa];
		}
		// main[idx] = mid[sgn_main*idx], side[k] = mid[-sgn_main*k] with sgn_main = +1 for vertical modes
		const int sm = is_ver ? 1 : -1;
		if (ok && mode >= 2) {
			for (int idx = lane; idx <= 2 * N; idx += HMR_WAVE) mainr[idx] = mid[sm * idx];
			if (angle < 0) {
				const int last = (N * angle) >> 5;   // projected entries idx = -1 .. last+1
				for (int t = 1 + lane; -t > last; t += HMR_WAVE) mainr[-t] = mid[-sm * ((128 + t * inv_angle) >> 8)];
			}
		}
		int dc = 0;
		if (ok && mode == 1) {
			int s = 0;
			for (int i = 1 + lane; i <= N; i += HMR_WAVE) s += mid[i] + mid[-i];
			s = wave_sum(s);
			dc = ((s + N) / (2 * N)) & 0xff;
		}
		__syncthreads();
		if (ok) {
			int16_t *c = Cc + jb.c_off;
			const int cs = (int)jb.c_stride;
			const bool edge = luma && N <= 16;
			for (int e = lane; e < N * N; e += HMR_WAVE) {
				const int y = e >> l2, x = e & (N - 1);
				int v;
				if (mode == 0) {
					const int left = mid[-(y + 1)], top = mid[x + 1], bl = mid[-(N + 1)], tr = mid[N + 1];
					v = ((N - 1 - x) * left + (x + 1) * tr + (N - 1 - y) * top + (y + 1) * bl + N) >> (l2 + 1);
				} else if (mode == 1) {
					v = dc;
					if (edge) {
						if (x == 0 && y == 0) v = (mid[-1] + mid[1] + 2 * dc + 2) >> 2;
						else if (y == 0) v = (mid[1 + x] + 3 * dc + 2) >> 2;
						else if (x == 0) v = (mid[-1 - y] + 3 * dc + 2) >> 2;
					}
				} else {
					// (line, pos) in the mode's own orientation: vertical modes line = row, horizontal modes line = column
					const int line = is_ver ? y : x, i = is_ver ? x : y;
					if (angle == 0) {
						v = mainr[i + 1] & 0xff;
						if (edge && i == 0) v = clip3i(v + ((mid[-sm * (line + 1)] - mid[0]) >> 1), 0, 255);
					} else {
						const int pos = (line + 1) * angle, delta = pos >> 5, fract = pos & 31, idx = i + delta + 1;
						v = fract ? (((32 - fract) * mainr[idx] + fract * mainr[idx + 1] + 16) >> 5) & 0xff : mainr[idx] & 0xff;
					}
				}
				c[(size_t)y * cs + x] = (int16_t)v;
			}
		}
		__syncthreads();
	}
}

// Intra reference build.  Flags must come from the partition tree (bottom_left implies left, top_right implies top).
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_refs(const hmr_gpu_job *__restrict__ jobs, int njobs, int N, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	__shared__ int16_t sAdi[HMR_WAVES_PER_BLOCK][4 * 64 + 1];
	const int lane = lane_id(), w = wave_in_block();
	int16_t *adi = sAdi[w];
	const int total = 4 * N + 1;
	for (long base = (long)blockIdx.x * HMR_WAVES_PER_BLOCK; base < njobs; base += (long)gridDim.x * HMR_WAVES_PER_BLOCK) {
		const long j = base + w;
		const bool ok = j < njobs;
		hmr_gpu_job jb;
		if (ok) {
			jb = jobs[j];
			const int16_t *d = A + jb.a_off;   // corner sample (-1,-1)
			const int st = (int)jb.a_stride;
			const bool left = jb.p0 & 1, top = jb.p0 & 2, bl = jb.p0 & 4, tr = jb.p0 & 8;
			const int bl_size = bl ? (int)(jb.p1 & 0xffff) : 0, tr_size = tr ? (int)(jb.p1 >> 16) : 0;
			// substitution samples (hmr_motion_intra.c:277,301,324-338)
			int first_sample, last_sample;
			if (left) first_sample = d[(size_t)(N + bl_size) * st];          // lowest available left / bottom-left sample
			else first_sample = d[1];                                        // top[0]
			if (top) last_sample = d[N + tr_size];                           // right-most available top / top-right sample
			else last_sample = d[(size_t)st];                                // top of the left column
			for (int i = lane; i < total; i += HMR_WAVE) {
				int v;
				if (!left && !top) v = 128;
				else if (i < N) {                         // bottom-left, adi[N-1-r] = row N+1+r
					const int r = N - 1 - i;
					v = (r < bl_size) ? d[(size_t)(N + 1 + r) * st] : first_sample;
				} else if (i < 2 * N) {                   // left, adi[N+r'] = row N-r'
					v = left ? d[(size_t)(2 * N - i) * st] : first_sample;
				} else if (i == 2 * N) {
					v = (left && top) ? d[0] : (left ? last_sample : first_sample);
				} else if (i <= 3 * N) {
					v = top ? d[i - 2 * N] : last_sample;
				} else {
					v = (i - 3 * N - 1 < tr_size) ? d[i - 2 * N] : last_sample;
				}
				adi[i] = (int16_t)v;
			}
		}
		__syncthreads();
		if (ok) {
			int16_t *o = Cc + jb.c_off;
			for (int i = lane; i < total; i += HMR_WAVE) o[i] = adi[i];
			if (jb.p0 & 16) {
				int16_t *f = Cc + jb.b_off;
				const int bls = adi[0], tl = adi[2 * N], trs = adi[total - 1];
				bool strong = false;
				if ((jb.p0 & 32) && N >= 32) {
					const int dl = bls + tl - 2 * adi[N], dt = tl + trs - 2 * adi[3 * N];
					strong = (dl < 0 ? -dl : dl) < 8 && (dt < 0 ? -dt : dt) < 8;
				}
				const int l2n = N == 32 ? 6 : 7;   // log2(2N) for the sizes that reach the strong branch
				for (int i = lane; i < total; i += HMR_WAVE) {
					int v;
					if (i == 0 || i == total - 1 || (strong && i == 2 * N)) v = adi[i];
					else if (strong) {
						v = i < 2 * N ? ((2 * N - i) * bls + i * tl + N) >> l2n : ((4 * N - i) * tl + (i - 2 * N) * trs + N) >> l2n;
					} else v = (adi[i - 1] + 2 * adi[i] + adi[i + 1] + 2) >> 2;
					f[i] = (int16_t)v;
				}
			}
		}
		__syncthreads();
	}
}

}  // namespace

extern "C" int hmr_gpu_intra_pred_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	if (size != 4 && size != 8 && size != 16 && size != 32 && size != 64) return HMR_GPU_ERR_ARG;
	hipLaunchKernelGGL(k_intra_pred, dim3(hmr_grid_for_waves(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, size, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_intra_refs_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	if (size != 4 && size != 8 && size != 16 && size != 32 && size != 64) return HMR_GPU_ERR_ARG;
	hipLaunchKernelGGL(k_intra_refs, dim3(hmr_grid_for_waves(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, size, a, c);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
