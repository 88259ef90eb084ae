// Batched intra prediction (planar / DC / 33 angular modes) and intra reference-sample construction.
// Reference semantics: hmr_sse42_functions_prediction.c:199,926 (scalar spec hmr_motion_intra.c:408-625),
// fill_reference_samples hmr_motion_intra.c:246 and adi_filter :189.
//
// One wave per block.  The 4N+1 neighbour samples are staged in LDS once, the projected main reference of
// the angular modes is built there in closed form (no serial running sums), and every lane then produces
// pixels so that consecutive lanes write consecutive samples of an output row, also for the horizontal
// modes the SSE code computes transposed.
#include "intra_device.h"

namespace {

// G = 4 / 16 / 32 / 64 lanes own one N = 4 / 8 / 16 / larger prediction block (16 / 4 / 2 / 1 blocks per wave): the per-block set-up
// (neighbour staging, mode decode, main reference) is wave-wide work, so packing blocks divides it.
template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_pred(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	constexpr int E = N * N, G = N == 4 ? 4 : N == 8 ? 16 : N == 16 ? 32 : HMR_WAVE, JPW = HMR_WAVE / G, JPB = JPW * HMR_WAVES_PER_BLOCK;
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
		const IntraMode m = intra_mode_setup(ok ? (int)jb.p0 : 0);
		const bool luma = ok && jb.p1 != 0;
		if (ok) intra_fill_main<N, G>(m, mid, mainr, l);
		const int dc = intra_dc<N, G>(mid, l, ok && m.mode == 1);
		wave_sync();
		if (ok) {
			int16_t *c = Cc + jb.c_off;
			const int cs = (int)jb.c_stride;
			const bool edge = luma && N <= 16;
			for (int e = l; e < E; e += G) {
				const int y = e >> l2, x = e & (N - 1);
				c[(size_t)y * cs + x] = (int16_t)intra_pixel<N>(m, mid, mainr, dc, edge, x, y);
			}
		}
		wave_sync();
	}
}

// Intra reference build.  Flags must come from the partition tree (bottom_left implies left, top_right implies top).
// G = 8 / 16 / 32 / 64 lanes own one 4N+1 array for N = 4 / 8 / 16 / larger.
template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_refs(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc)
{
	constexpr int total = 4 * N + 1, G = N == 4 ? 8 : N == 8 ? 16 : N == 16 ? 32 : HMR_WAVE, JPW = HMR_WAVE / G, JPB = JPW * HMR_WAVES_PER_BLOCK;
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
			intra_build_refs<N, G>(adi, d, st, left, top, bl_size, tr_size, l);
		}
		wave_sync();
		if (ok) {
			int16_t *o = Cc + jb.c_off;
			for (int i = l; i < total; i += G) o[i] = adi[i];
			if (jb.p0 & 16) intra_filter_refs<N, G>(adi, Cc + jb.b_off, (jb.p0 & 32) != 0, l);
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
#define PRED_JPW(n) ((n) == 4 ? 16 : (n) == 8 ? 4 : (n) == 16 ? 2 : 1)
#define REFS_JPW(n) ((n) == 4 ? 8 : (n) == 8 ? 4 : (n) == 16 ? 2 : 1)

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
