// Fused intra mode search of one PU: homer_loop1_motion_intra (hmr_motion_intra.c:1084-1179).
//
// The reference issues, per PU, one fill_reference_samples (raw + smoothed neighbour arrays), then up to 13 rounds of
// {create_intra_*_prediction into prediction_wnd, sad against the source} through the table, each result compared on the host.
// Here G = min(64, N*N) lanes own the PU for the whole search: both neighbour arrays live in LDS, a candidate's prediction is
// never written out - each lane predicts its pixels and accumulates |orig - pred| directly - and the strict-< cost comparison
// (SAD + bits * sqrt_lambda in IEEE double, like the reference) is evaluated redundantly by every lane of the group, so the
// next round's candidates need no broadcast.  Four 4x4 PUs share a wavefront; the loop structure is uniform over the
// workgroup, the candidate modes are per PU.
#include "intra_device.h"

namespace {

// search_points / num_search_points (hmr_motion_intra.c:1076-1080) and intra_filter (:148) as compile-time constants: the 13-candidate
// schedule unrolls and no table load sits in the dependent chain of a candidate
struct SearchPlan {
	static constexpr int point(int loop, int k)
	{
		constexpr int p[4][5] = {{0, 1, 0, 8, 16}, {2, 10, 16, 22, 30}, {-4, -2, 2, 4, 0}, {-1, 1, 0, 0, 0}};
		return p[loop][k];
	}
	static constexpr int count(int loop)
	{
		constexpr int n[4] = {2, 5, 4, 2};
		return n[loop];
	}
	static constexpr int filter_thr(int l2)
	{
		constexpr int t[5] = {10, 7, 1, 0, 10};
		return t[l2 - 2];
	}
};

// WPJ wavefronts cooperate on one PU (2 for N = 32, 4 for N = 64: a 64x64 PU is 13 x 4096 predicted samples, too long a
// dependent chain for one wave; 1 otherwise).  A lane owns PPL samples of one column, so the source samples stay in registers for all
// candidates.
template <int N, int WPJ>
__global__ __launch_bounds__(HMR_BLOCK) void k_intra_search(const hmr_gpu_intra_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ O,
							       const int16_t *__restrict__ D, int16_t *__restrict__ Cc, hmr_gpu_intra_result *__restrict__ out)
{
	// lanes per PU: 4 / 16 for N = 4 / 8 (16 / 4 PUs share a wavefront: the per-candidate set-up, synchronisation,
	// reduction and cost arithmetic are paid once per wavefront, so packing PUs divides them), a wavefront or more above
	constexpr int E = N * N, G = WPJ > 1 ? HMR_WAVE * WPJ : (N == 4 ? 4 : N == 8 ? 16 : HMR_WAVE);
	constexpr int JPB = HMR_BLOCK / G;                     // jobs per workgroup
	constexpr int PPL = E / G, YSTEP = G / N;              // samples per lane, row distance between a lane's samples
	constexpr int l2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6, total = 4 * N + 1;
	__shared__ int16_t sAdi[JPB][2][total + 3];   // raw, smoothed
	__shared__ int16_t sMainBuf[JPB][3 * N + 2];
	__shared__ int sRed[HMR_WAVES_PER_BLOCK];
	const int tid = threadIdx.x, sub = tid / G, l = tid % G, w = wave_in_block();
	int16_t *adi = sAdi[sub][0], *adif = sAdi[sub][1];
	int16_t *mainr = sMainBuf[sub] + N;
	const int x = l & (N - 1), y0 = l >> l2;
	auto sync = [&]() {
		if constexpr (WPJ == 1) wave_sync();
		else __syncthreads();
	};
	auto job_sum = [&](int v) -> int {      // called by every lane of the workgroup
		if constexpr (WPJ == 1) return group_sum<G>(v);
		else {
			v = wave_sum(v);
			if (lane_id() == 0) sRed[w] = v;
			__syncthreads();
			int t = 0;
#pragma unroll
			for (int i = 0; i < WPJ; i++) t += sRed[(w / WPJ) * WPJ + i];
			__syncthreads();
			return t;
		}
	};
	const JobRange jr = xcd_job_range(njobs, JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + sub;
		const bool ok = j < jr.end;
		hmr_gpu_intra_job jb = {};
		if (ok) {
			jb = jobs[j];
			const bool left = jb.flags & 1, top = jb.flags & 2, bl = jb.flags & 4, tr = jb.flags & 8;
			intra_build_refs<N, G>(adi, D + jb.dec_off, (int)jb.dec_stride, left, top, bl ? (int)(jb.sizes & 0xffff) : 0, tr ? (int)(jb.sizes >> 16) : 0, l);
		}
		sync();
		if (ok) {
			intra_filter_refs<N, G>(adi, adif, (jb.flags & 32) != 0, l);
			int16_t *o = Cc + jb.adi_off;
			for (int i = l; i < total; i += G) o[i] = adi[i];
		}
		// DC is never predicted from the smoothed array (:1128), so its value is a property of the job
		int dcs = 0;
		if (ok)
			for (int i = 1 + l; i <= N; i += G) dcs += adi[2 * N + i] + adi[2 * N - i];
		const int dc = ((job_sum(dcs) + N) / (2 * N)) & 0xff;
		sync();
		if (ok) {
			int16_t *o = Cc + jb.adif_off;
			for (int i = l; i < total; i += G) o[i] = adif[i];
		}
		int og[PPL];
		{
			const int16_t *org = O + jb.orig_off + x;
			const int os = (int)jb.orig_stride;
#pragma unroll
			for (int i = 0; i < PPL; i++) og[i] = ok ? org[(size_t)(y0 + i * YSTEP) * os] : 0;
		}
		int best = 0, new_best = 0, best_bits = 0, min_mode = 0, max_mode = 1, last_mode = 0;
		double best_cost = (double)(0xffffffffu / 8);   // MAX_COST, hmr_private.h:54
#pragma unroll
		for (int loop = 0; loop < 4; loop++) {
			if (loop == 1) {
				best = 2;
				min_mode = 2;
				max_mode = 34;
			}
#pragma unroll
			for (int k = 0; k < 5; k++) {
				if (k >= SearchPlan::count(loop)) continue;
				const int mode = best + SearchPlan::point(loop, k);
				const bool valid = ok && mode >= min_mode && mode <= max_mode;
				const IntraMode m = intra_mode_setup(valid ? mode : 0);
				const int d10 = mode > 10 ? mode - 10 : 10 - mode, d26 = mode > 26 ? mode - 26 : 26 - mode;
				const bool filtered = mode != 1 && (d10 < d26 ? d10 : d26) > SearchPlan::filter_thr(l2);
				const int16_t *mid = (filtered ? adif : adi) + 2 * N;
				if (valid) intra_fill_main<N, G>(m, mid, mainr, l);
				sync();
				int s = 0;
				if (valid) {
#pragma unroll
					for (int i = 0; i < PPL; i++) {
						const int d = og[i] - intra_pixel<N>(m, mid, mainr, dc, N <= 16, x, y0 + i * YSTEP);
						s += d < 0 ? -d : d;
					}
				}
				s = job_sum(s);
				sync();          // mainr is rebuilt by the next candidate
				if (valid) {
					const unsigned bits = (int)jb.preds[0] == mode ? jb.pred_bits[0] : (int)jb.preds[1] == mode ? jb.pred_bits[1] : (int)jb.preds[2] == mode ? jb.pred_bits[2] : jb.other_bits;
					const double cost = (double)(unsigned)s + (double)bits * jb.sqrt_lambda;
					if (cost < best_cost) {
						best_cost = cost;
						new_best = mode;
						best_bits = (int)bits;
					}
					last_mode = mode;
				}
			}
			best = new_best;
		}
		// the reference leaves the prediction of the last candidate it evaluated in the prediction window
		{
			const IntraMode m = intra_mode_setup(last_mode);
			const int d10 = last_mode > 10 ? last_mode - 10 : 10 - last_mode, d26 = last_mode > 26 ? last_mode - 26 : 26 - last_mode;
			const bool filtered = last_mode != 1 && (d10 < d26 ? d10 : d26) > SearchPlan::filter_thr(l2);
			const int16_t *mid = (filtered ? adif : adi) + 2 * N;
			if (ok) intra_fill_main<N, G>(m, mid, mainr, l);
			sync();
			if (ok) {
				int16_t *c = Cc + jb.pred_off + x;
				const int cs = (int)jb.pred_stride;
#pragma unroll
				for (int i = 0; i < PPL; i++) c[(size_t)(y0 + i * YSTEP) * cs] = (int16_t)intra_pixel<N>(m, mid, mainr, dc, N <= 16, x, y0 + i * YSTEP);
				if (l == 0) {
					hmr_gpu_intra_result r;
					r.best_mode = best;
					r.bits = best_bits;
					r.cost = best_cost;
					out[j] = r;
				}
			}
		}
		sync();
	}
}

}  // namespace

extern "C" int hmr_gpu_intra_search_batch(hmr_gpu_ctx *ctx, const hmr_gpu_intra_job *jobs, int njobs, int size, const int16_t *orig_base,
					   const int16_t *decoded_base, int16_t *out_base, hmr_gpu_intra_result *out)
{
	if (njobs <= 0) return HMR_GPU_OK;
#define LAUNCH(NN, WPJ, JPB) \
	hipLaunchKernelGGL((k_intra_search<NN, WPJ>), dim3(hmr_grid_for_units(((long)njobs + JPB - 1) / JPB)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, orig_base, \
			   decoded_base, out_base, out)
	switch (size) {
	case 4: LAUNCH(4, 1, 64); break;
	case 8: LAUNCH(8, 1, 16); break;
	case 16: LAUNCH(16, 1, 4); break;
	case 32: LAUNCH(32, 2, 2); break;
	case 64: LAUNCH(64, 4, 1); break;
	default: hmr_set_error("intra search: unsupported size %d", size); return HMR_GPU_ERR_ARG;
	}
#undef LAUNCH
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
