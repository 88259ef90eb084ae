// Transform-tree consolidation (include/homer_gpu.h section 9): the comparison and buffer synchronisation of encode_intra_luma's tree walk
// (hmr_motion_intra.c:1479-1557).  One wavefront per CU: lane 0's scalars decide, all 64 lanes copy.  Element-wise, HBM/L2 bound:
// at most 2 * size^2 samples + size^2 levels move per CU.
#include "common.h"
#include "vec.h"

namespace {

__global__ __launch_bounds__(HMR_BLOCK) void k_tree_decide(const hmr_gpu_tree_job *__restrict__ jobs, int njobs, const uint32_t *__restrict__ ssd,
							    const int32_t *__restrict__ ac, int16_t *__restrict__ R, int16_t *__restrict__ L,
							    hmr_gpu_tree_result *__restrict__ out)
{
	const int lane = lane_id(), w = wave_in_block();
	const JobRange jr = xcd_job_range(njobs, HMR_WAVES_PER_BLOCK);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w;
		if (j >= jr.end) continue;
		const hmr_gpu_tree_job jb = jobs[j];
		const int n = (int)jb.size;
		const bool has_parent = jb.parent != HMR_GPU_TREE_NO_PARENT;
		uint32_t cs[4], ca[4];
#pragma unroll
		for (int k = 0; k < 4; k++) { cs[k] = ssd[jb.child[k]]; ca[k] = (uint32_t)ac[jb.child[k]]; }
		const uint32_t dist = cs[0] + cs[1] + cs[2] + cs[3], sum = ca[0] + ca[1] + ca[2] + ca[3];          // uint32 like the node fields
		const uint32_t pcost = has_parent ? ssd[jb.parent] : 0x7fffffffu, psum = has_parent ? (uint32_t)ac[jb.parent] : 0u;
		bool split;
		if (!has_parent) split = true;
		else if (jb.rule == 1) {
			// 1.25 * ((double)dist + (uint32)(45 * sum)) < (double)(uint32)(pcost + 45 * psum), exact in 64-bit integers as 5 * lhs < 4 * rhs
			const unsigned long long lhs = (unsigned long long)dist + (uint32_t)(45u * sum), rhs = (uint32_t)(pcost + 45u * psum);
			split = 5ull * lhs < 4ull * rhs;
		} else
			split = dist < pcost;
		int16_t *rp = R + jb.par_rec_off, *rc = R + jb.chl_rec_off;
		const int sp = (int)jb.par_rec_stride, sc = (int)jb.chl_rec_stride;
		if (split) {
			const int per_row = n / 4;                                   // 4-sample pieces per row
			for (int e = lane; e < n * per_row; e += HMR_WAVE) {
				const int y = e / per_row, x = (e % per_row) * 4;
				st4(rp + (size_t)y * sp + x, ld4(rc + (size_t)y * sc + x));
			}
			int16_t *lp = L + jb.par_lev_off;
			const int16_t *lc = L + jb.chl_lev_off;
			for (int e = lane * 4; e < n * n; e += HMR_WAVE * 4) st4(lp + e, ld4(lc + e));
		} else {
			for (int x = lane * 4; x < n; x += HMR_WAVE * 4) st4(rc + (size_t)(n - 1) * sc + x, ld4(rp + (size_t)(n - 1) * sp + x));
			for (int y = lane; y < n - 1; y += HMR_WAVE) rc[(size_t)y * sc + n - 1] = rp[(size_t)y * sp + n - 1];
		}
		if (lane == 0) {
			hmr_gpu_tree_result r;
			r.split = split ? 1u : 0u;
			r.cost = split ? dist : pcost;
			r.sum = split ? sum : psum;
			const uint8_t any = (ca[0] | ca[1] | ca[2] | ca[3]) ? 1 : 0;
#pragma unroll
			for (int k = 0; k < 4; k++) r.cbf[k] = split ? (uint8_t)(((ca[k] ? 1 : 0) << 1) | any) : (uint8_t)(psum ? 1 : 0);
			out[j] = r;
		}
	}
}

}  // namespace

extern "C" int hmr_gpu_tree_decide_batch(hmr_gpu_ctx *ctx, const hmr_gpu_tree_job *jobs, int njobs, const uint32_t *ssd, const int32_t *ac_sum, int16_t *recon_base,
					 int16_t *level_base, hmr_gpu_tree_result *out)
{
	if (njobs <= 0) return HMR_GPU_OK;
	hipLaunchKernelGGL(k_tree_decide, dim3(hmr_grid_for_waves(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, ssd, ac_sum, recon_base, level_base, out);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
