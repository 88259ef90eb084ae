// Batched forward / inverse core transforms and (de)quantisation with sign-data hiding.
// Reference semantics: hmr_sse42_functions_transform.c:1670,1700 (scalar spec hmr_transform.c:133-587),
// hmr_sse42_functions_quant.c:34,135 and sign_bit_hidding hmr_quant.c:61.
//
// A TU is owned by G = clamp(N*N/4, 16, 64) lanes of a wave (four 4x4 or 8x8 TUs per wave, one larger TU per wave); waves own
// private LDS regions and synchronise only with themselves.  The
// two separable stages go through LDS tiles with a row pitch of N+2 samples (17 dwords for N = 32) so that
// the "lanes walk rows" reads of stage 1 and the transposed writes are bank-conflict free; the basis
// matrix sits in LDS once per workgroup.  Integer multiply-accumulate on 16-bit data: VALU, not MFMA
// (the products must be exact 32-bit integers with saturating 16-bit packs between the stages).
#include "common.h"
#include "tq_device.h"

namespace {

template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_transform(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							    int16_t *__restrict__ Cc, const DevTables *__restrict__ tab)
{
	using g = Geo<N>;
	__shared__ int16_t sM[2][N * N];
	__shared__ int16_t sIn[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];
	__shared__ int16_t sTmp[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];
	const int lane = lane_id(), w = wave_in_block(), sub = lane / g::G, l = lane % g::G;
	load_basis<N>(sM, tab);
	__syncthreads();
	constexpr int sh1 = g::L2 - 1, sh2 = g::L2 + 6;
	const JobRange jr = xcd_job_range(njobs, g::JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * g::JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_job jb;
		if (ok) {
			jb = jobs[j];
			const int16_t *a = A + jb.a_off;
			for (int e = l; e < g::E; e += g::G) {
				const int y = e / N, x = e % N;
				sIn[w][sub][y * g::P + x] = a[(size_t)y * jb.a_stride + x];
			}
		}
		wave_sync();
		if (ok) {
			const int16_t *M = sM[(N == 4 && jb.p0) ? 1 : 0];
			for (int o = l; o < g::E; o += g::G) {     // tmp[k][row] = sum_i M[k][i] * in[row][i]
				const int k = o / N, row = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[k * N + i] * sIn[w][sub][row * g::P + i];
				sTmp[w][sub][k * g::P + row] = (int16_t)sat16i((s + (1 << (sh1 - 1))) >> sh1);
			}
		}
		wave_sync();
		if (ok) {
			const int16_t *M = sM[(N == 4 && jb.p0) ? 1 : 0];
			int16_t *c = Cc + jb.c_off;
			for (int o = l; o < g::E; o += g::G) {     // coeff[k2][k1] = sum_j M[k2][j] * tmp[k1][j]
				const int k2 = o / N, k1 = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[k2 * N + i] * sTmp[w][sub][k1 * g::P + i];
				c[o] = (int16_t)sat16i((s + (1 << (sh2 - 1))) >> sh2);
			}
		}
		wave_sync();
	}
}

template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_itransform(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							     int16_t *__restrict__ Cc, const DevTables *__restrict__ tab)
{
	using g = Geo<N>;
	__shared__ int16_t sM[2][N * N];
	__shared__ int16_t sIn[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];
	__shared__ int16_t sTmp[HMR_WAVES_PER_BLOCK][g::JPW][N * g::P];
	const int lane = lane_id(), w = wave_in_block(), sub = lane / g::G, l = lane % g::G;
	load_basis<N>(sM, tab);
	__syncthreads();
	const JobRange jr = xcd_job_range(njobs, g::JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * g::JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_job jb;
		if (ok) {
			jb = jobs[j];
			const int16_t *a = A + jb.a_off;
			for (int e = l; e < g::E; e += g::G) sIn[w][sub][(e / N) * g::P + (e % N)] = a[e];
		}
		wave_sync();
		if (ok) {
			const int16_t *M = sM[(N == 4 && jb.p0) ? 1 : 0];
			for (int o = l; o < g::E; o += g::G) {     // tmp[col][k] = sum_i M[i][k] * coeff[i][col]
				const int k = o / N, col = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[i * N + k] * sIn[w][sub][i * g::P + col];
				sTmp[w][sub][col * g::P + k] = (int16_t)sat16i((s + 64) >> 7);
			}
		}
		wave_sync();
		if (ok) {
			const int16_t *M = sM[(N == 4 && jb.p0) ? 1 : 0];
			int16_t *c = Cc + jb.c_off;
			for (int o = l; o < g::E; o += g::G) {     // out[y][x] = sum_i M[i][x] * tmp[i][y]
				const int y = o / N, x = o % N;
				int s = 0;
#pragma unroll
				for (int i = 0; i < N; i++) s += M[i * N + x] * sTmp[w][sub][i * g::P + y];
				c[(size_t)y * jb.c_stride + x] = (int16_t)sat16i((s + 2048) >> 12);
			}
		}
		wave_sync();
	}
}

template <int N>
__global__ __launch_bounds__(HMR_BLOCK) void k_quant(const hmr_gpu_job *__restrict__ jobs, int njobs, const int16_t *__restrict__ A,
							int16_t *__restrict__ Cc, int16_t *__restrict__ DU, int32_t *__restrict__ ac_out,
							const DevTables *__restrict__ tab)
{
	using g = Geo<N>;
	__shared__ int16_t sSrc[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sDst[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ int16_t sDu[HMR_WAVES_PER_BLOCK][g::JPW][g::E];
	__shared__ unsigned long long sNzMask[HMR_WAVES_PER_BLOCK][g::JPW];   // coefficient groups (scan order) that hold a level
	const int lane = lane_id(), w = wave_in_block(), sub = lane / g::G, l = lane % g::G;
	constexpr int CG_PER_IT = g::G / 16, SIDE = N / 4;
	const JobRange jr = xcd_job_range(njobs, g::JPB);
	for (long base = jr.begin; base < jr.end; base += jr.stride) {
		const long j = base + w * g::JPW + sub;
		const bool ok = j < jr.end;
		hmr_gpu_job jb = {};
		int ac = 0;
		bool sbh = false;
		const uint32_t *scan = tab->scan[3][g::L2];
		if (l == 0) sNzMask[w][sub] = 0;
		wave_sync();
		if (ok) {
			jb = jobs[j];
			const int scan_mode = jb.p0 & 3, comp = (jb.p0 >> 2) & 3, is_intra = (jb.p0 >> 4) & 1, slice_i = (jb.p0 >> 5) & 1;
			const int per = jb.p1 & 0xff, rem = (jb.p1 >> 8) & 0xff;
			sbh = (jb.p0 >> 6) & 1;
			const int32_t *q = tab->quant[g::L2 - 2][(is_intra ? 0 : 3) + comp][rem];
			const uint8_t *b2c = tab->blk2cg[scan_mode][g::L2];
			scan = tab->scan[scan_mode][g::L2];
			const int qbits = 14 + per + (7 - g::L2), qbits8 = qbits - 8;
			const uint32_t add = (uint32_t)(slice_i ? 171 : 85) << (qbits - 9);
			const int16_t *a = A + jb.a_off;
			uint32_t sum = 0;
			unsigned long long nz = 0;
			for (int e = l; e < g::E; e += g::G) {
				const int s = a[e];
				const uint32_t mag = (uint16_t)(s < 0 ? -s : s);
				const uint32_t aux = mag * (uint32_t)q[e];
				const int c = (int)(aux + add) >> qbits;
				const int d = (int)(aux - ((uint32_t)c << qbits)) >> qbits8;
				sum += (uint32_t)c;
				const int sgn = s > 0 ? 1 : (s < 0 ? -1 : 0);
				const int lv = (int16_t)(sgn * sat16i(c));
				sSrc[w][sub][e] = (int16_t)s;
				sDst[w][sub][e] = (int16_t)lv;
				sDu[w][sub][e] = (int16_t)sat16i(d);
				if (lv) nz |= 1ull << b2c[((e / N) >> 2) * SIDE + ((e % N) >> 2)];
			}
			ac = (int)group_sum<g::G>(sum);
			if (nz) atomicOr(&sNzMask[w][sub], nz);
		}
		wave_sync();
		// sign hiding only visits groups that hold a level; CG_PER_IT groups per pass (16 lanes each)
		const bool run_sbh = ok && sbh && ac >= 2;
		unsigned long long m = run_sbh ? sNzMask[w][sub] : 0ull;
		const int last = m ? 63 - __clzll((long long)m) : -1;
		const int grp = l >> 4;
		while (__any(m != 0)) {
			unsigned long long t = m;
			int cg = -1;
#pragma unroll
			for (int k = 0; k < CG_PER_IT; k++) {
				const int b = t ? __ffsll((long long)t) - 1 : -1;
				if (k == grp) cg = b;
				t &= t - 1;
			}
			m = t;
			sbh_group16(sDst[w][sub], sSrc[w][sub], sDu[w][sub], scan, cg < 0 ? 0 : cg, cg == last, run_sbh && cg >= 0);
		}
		wave_sync();
		if (ok) {
			int16_t *c = Cc + jb.c_off;
			for (int e = l; e < g::E; e += g::G) c[e] = sDst[w][sub][e];
			if (DU) {
				int16_t *du = DU + jb.b_off;
				for (int e = l; e < g::E; e += g::G) du[e] = sDu[w][sub][e];
			}
			if (l == 0) ac_out[j] = ac;
		}
		wave_sync();
	}
}

__global__ __launch_bounds__(HMR_BLOCK) void k_inv_quant(const hmr_gpu_job *__restrict__ jobs, int njobs, int size, const int16_t *__restrict__ A,
							    int16_t *__restrict__ Cc, const DevTables *__restrict__ tab)
{
	const int lane = lane_id();
	const JobRange jr = xcd_job_range(njobs, HMR_WAVES_PER_BLOCK);
	const int l2 = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : 5;
	const int iq_shift = 3 + l2;   // 20 - 14 - (15 - 8 - log2N) + 4
	for (long j = jr.begin + wave_in_block(); j < jr.end; j += jr.stride) {
		const hmr_gpu_job jb = jobs[j];
		const int comp = (jb.p0 >> 2) & 3, is_intra = (jb.p0 >> 4) & 1;
		const int per = jb.p1 & 0xff, rem = (jb.p1 >> 8) & 0xff;
		// list index keeps the reference's `is_intra?0:3 + comp` precedence (hmr_sse42_functions_quant.c:138)
		const int32_t *iq = tab->dequant[l2 - 2][is_intra ? 0 : 3 + comp][rem];
		const int16_t *a = A + jb.a_off;
		int16_t *c = Cc + jb.c_off;
		for (int e = lane; e < size * size; e += HMR_WAVE) {
			const uint32_t prod = (uint32_t)(int)a[e] * (uint32_t)iq[e];
			int r;
			if (iq_shift > per) r = (int)(prod + (1u << (iq_shift - per - 1))) >> (iq_shift - per);
			else r = (int)(prod << (per - iq_shift));
			c[e] = (int16_t)sat16i(r);
		}
	}
}

template <int N> int grid_for(int njobs) { return hmr_grid_for_units(((long)njobs + Geo<N>::JPB - 1) / Geo<N>::JPB); }

}  // namespace

#define DISPATCH_N(KERNEL, ...)                                                                                             \
	switch (size) {                                                                                                     \
	case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(grid_for<4>(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, __VA_ARGS__); break;   \
	case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(grid_for<8>(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, __VA_ARGS__); break;   \
	case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(grid_for<16>(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, __VA_ARGS__); break; \
	case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(grid_for<32>(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, __VA_ARGS__); break; \
	default: return HMR_GPU_OK; /* unsupported sizes silently do nothing, hmr_sse42_functions_transform.c:1672-1694 */ \
	}

extern "C" int hmr_gpu_transform_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	DISPATCH_N(k_transform, jobs, njobs, a, c, ctx->tables)
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_itransform_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	DISPATCH_N(k_itransform, jobs, njobs, a, c, ctx->tables)
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_quant_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c, int16_t *du, int32_t *ac)
{
	if (njobs <= 0) return HMR_GPU_OK;
	if (size != 4 && size != 8 && size != 16 && size != 32) return HMR_GPU_ERR_ARG;
	DISPATCH_N(k_quant, jobs, njobs, a, c, du, ac, ctx->tables)
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_inv_quant_batch(hmr_gpu_ctx *ctx, const hmr_gpu_job *jobs, int njobs, int size, const int16_t *a, int16_t *c)
{
	if (njobs <= 0) return HMR_GPU_OK;
	if (size != 4 && size != 8 && size != 16 && size != 32) return HMR_GPU_ERR_ARG;
	hipLaunchKernelGGL(k_inv_quant, dim3(hmr_grid_for_waves(njobs)), dim3(HMR_BLOCK), 0, ctx->stream, jobs, njobs, size, a, c, ctx->tables);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
