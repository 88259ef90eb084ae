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

// G = min(64, N*N) lanes own one prediction block: four 4x4 blocks per wave, one larger block per wave.
template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_pred(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	constexpr int E = N * N, G = E < HMR_WAVE ? E : HMR_WAVE, JPW = HMR_WAVE / G, JPB = JPW * HMR_WAVES_PER_BLOCK;
	constexpr int l2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
	__shared__ int16_t sAdi[HMR_WAVES_PER_BLOCK][JPW][4 * N + 1 + 3];
	__shared__ int16_t sMainBuf[HMR_WAVES_PER_BLOCK][JPW][3 * N + 2];   // main reference, index -N+1 .. 2N, origin at N
	const int lane = lane_id(), w = wave_in_block(), sub = lane / G, l = lane % G;
	int16_t *adi = sAdi[w][sub];
	int16_t *mainr = sMainBuf[w][sub] + N;
	const int16_t *mid = adi + 2 * N;
	const JobRange jr = xcd_job_range(njobs, JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_job jb = {};
		if (ok) {
			jb = jobs[j];
			const int16_t *a = A + jb.a_off;
			for (int i = l; i < 4 * N + 1; i += G) adi[i] = a[i];
		}
		wave_sync();
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
		// main[idx] = mid[sm*idx], side[k] = mid[-sm*k] with sm = +1 for vertical modes
		const int sm = is_ver ? 1 : -1;
		if (ok && mode >= 2) {
			for (int idx = l; idx <= 2 * N; idx += G) mainr[idx] = mid[sm * idx];
			if (angle < 0) {
				const int last = (N * angle) >> 5;   // projected entries idx = -1 .. last+1
				for (int t = 1 + l; -t > last; t += G) mainr[-t] = mid[-sm * ((128 + t * inv_angle) >> 8)];
			}
		}
		int s = 0;
		if (ok && mode == 1)
			for (int i = 1 + l; i <= N; i += G) s += mid[i] + mid[-i];
		s = group_sum<G>(s);
		const int dc = ((s + N) / (2 * N)) & 0xff;
		wave_sync();
		if (ok) {
			int16_t *c = Cc + jb.c_off;
			const int cs = (int)jb.c_stride;
			const bool edge = luma && N <= 16;
			for (int e = l; e < E; e += G) {
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
		wave_sync();
	}
}

// Intra reference build.  Flags must come from the partition tree (bottom_left implies left, top_right implies top).
// G lanes (32 for N = 4, else 64) own one 4N+1 array.
template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_refs(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	constexpr int total = 4 * N + 1, G = N == 4 ? 32 : HMR_WAVE, JPW = HMR_WAVE / G, JPB = JPW * HMR_WAVES_PER_BLOCK;
	__shared__ int16_t sAdi[HMR_WAVES_PER_BLOCK][JPW][total + 3];
	const int lane = lane_id(), w = wave_in_block(), sub = lane / G, l = lane % G;
	int16_t *adi = sAdi[w][sub];
	const JobRange jr = xcd_job_range(njobs, JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_job jb = {};
		if (ok) {
			jb = jobs[j];
			const int16_t *d = A + jb.a_off;   // corner sample (-1,-1)
			const int st = (int)jb.a_stride;
			const bool left = jb.p0 & 1, top = jb.p0 & 2, bl = jb.p0 & 4, tr = jb.p0 & 8;
			const int bl_size = bl ? (int)(jb.p1 & 0xffff) : 0, tr_size = tr ? (int)(jb.p1 >> 16) : 0;
			// substitution samples (hmr_motion_intra.c:277,301,324-338)
			int first_sample = 128, last_sample = 128;
			if (left) first_sample = d[(size_t)(N + bl_size) * st];          // lowest available left / bottom-left sample
			else if (top) first_sample = d[1];                               // top[0]
			if (top) last_sample = d[N + tr_size];                           // right-most available top / top-right sample
			else if (left) last_sample = d[(size_t)st];                      // top of the left column
			for (int i = l; i < total; i += G) {
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
		wave_sync();
		if (ok) {
			int16_t *o = Cc + jb.c_off;
			for (int i = l; i < total; i += G) o[i] = adi[i];
			if (jb.p0 & 16) {
				int16_t *f = Cc + jb.b_off;
				const int bls = adi[0], tl = adi[2 * N], trs = adi[total - 1];
				bool strong = false;
				if ((jb.p0 & 32) && N >= 32) {
					const int dl = bls + tl - 2 * adi[N], dt = tl + trs - 2 * adi[3 * N];
					strong = (dl < 0 ? -dl : dl) < 8 && (dt < 0 ? -dt : dt) < 8;
				}
				constexpr int l2n = N == 32 ? 6 : 7;   // log2(2N) for the sizes that reach the strong branch
				for (int i = l; i < total; i += G) {
					int v;
					if (i == 0 || i == total - 1 || (strong && i == 2 * N)) v = adi[i];
					else if (strong) {
						v = i < 2 * N ? ((2 * N - i) * bls + i * tl + N) >> l2n : ((4 * N - i) * tl + (i - 2 * N) * trs + N) >> l2n;
					} else v = (adi[i - 1] + 2 * adi[i] + adi[i + 1] + 2) >> 2;
					f[i] = (int16_t)v;
				}
			}
		}
		wave_sync();
	}
}

template <int N> int intra_grid(int njobs, int jpw)
{
	return hmr_grid_for_units(((long)njobs + jpw * HMR_WAVES_PER_BLOCK - 1) / (jpw * HMR_WAVES_PER_BLOCK));
}

}  // namespace

#define INTRA_DISPATCH(KERNEL, JPW_OF)                                                                                              \
	switch (size) {                                                                                                             \
	case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(intra_grid<4>(njobs, JPW_OF(4))), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, a, c); break;     \
	case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(intra_grid<8>(njobs, JPW_OF(8))), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, a, c); break;     \
	case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(intra_grid<16>(njobs, JPW_OF(16))), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, a, c); break; \
	case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(intra_grid<32>(njobs, JPW_OF(32))), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, a, c); break; \
	case 64: hipLaunchKernelGGL((KERNEL<64>), dim3(intra_grid<64>(njobs, JPW_OF(64))), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, a, c); break; \
	default: return HMR_GPU_ERR_ARG;                                                                                            \
	}
#define PRED_JPW(n) ((n) == 4 ? 4 : 1)
#define REFS_JPW(n) ((n) == 4 ? 2 : 1)

extern "C" int hmr_gpu_intra_pred_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	INTRA_DISPATCH(k_intra_pred, PRED_JPW)
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_intra_refs_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	INTRA_DISPATCH(k_intra_refs, REFS_JPW)
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
