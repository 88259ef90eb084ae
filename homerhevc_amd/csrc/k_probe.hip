// Issue-rate probe (include/homer_gpu.h section 14): what a CU's SIMDs deliver in plain vector instructions, measured on the device the bench runs on.
//
// bench.py prices k_encode_pool - a kernel bound by the instruction issue of a few wavefronts per CU and by the latency of its dependent chains, not by HBM
// bandwidth - against "wave-instructions per second".  The guide's figure for that ceiling (MI355X_MICROARCH.md: 1024 SIMDs, a wave64 VALU instruction occupies
// its SIMD's 16 lanes for four cycles) depends on the clock the part actually sustains; this kernel measures it: every wavefront executes `iters` rounds of 64
// instructions of one kind - op 0 v_mad_u32_u24, 1 v_add_u32, 2 v_mov_b32, 3 v_perm_b32, 4 s_add_u32 (the scalar unit) - on eight independent accumulators (no memory access inside the loop), `waves_per_simd` wavefronts per SIMD on every CU.  With
// `dependent` the 64 instructions form ONE chain (each needs the result of the one before): what a single dependent stream - a row worker walking a decision
// chain - can issue.
#include "common.h"

namespace {

// OP: 0 v_mad_u32_u24, 1 v_add_u32, 2 v_mov_b32, 3 v_perm_b32, 4 s_add_u32 (the scalar unit)
#define PROBE_STEP(acc, other)                                                                                 \
	do {                                                                                                   \
		if (OP == 0) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc) : "v"(m));               \
		else if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(m));                   \
		else if (OP == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(acc) : "v"(other));                   \
		else if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc) : "v"(m), "v"(other)); \
	} while (0)
template <bool DEPENDENT, int OP>
__global__ __launch_bounds__(1024) void k_probe_valu(int iters, uint32_t *out)
{
	uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const uint32_t m = 2654435761u + blockIdx.x;
	uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
	for (int i = 0; i < iters; i++) {
		if (OP == 4) {
#pragma unroll
			for (int k = 0; k < 16; k++) {
				if (DEPENDENT) {
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
				} else {
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2));
					asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(s0));
				}
			}
		} else if (DEPENDENT) {
#pragma unroll
			for (int k = 0; k < 64; k++) PROBE_STEP(a0, a1);
		} else {
#pragma unroll
			for (int k = 0; k < 8; k++) {
				PROBE_STEP(a0, a1); PROBE_STEP(a1, a2); PROBE_STEP(a2, a3); PROBE_STEP(a3, a4);
				PROBE_STEP(a4, a5); PROBE_STEP(a5, a6); PROBE_STEP(a6, a7); PROBE_STEP(a7, a0);
			}
		}
	}
	const uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
	if (r == 0x12345u) out[0] = r;      // (keeps the loop alive; practically never true)
}

template <int OP>
void probe_launch(bool dependent, dim3 grid, dim3 block, hipStream_t st, int iters, uint32_t *out)
{
	if (dependent) hipLaunchKernelGGL((k_probe_valu<true, OP>), grid, block, 0, st, iters, out);
	else hipLaunchKernelGGL((k_probe_valu<false, OP>), grid, block, 0, st, iters, out);
}

}  // namespace

// waves_per_simd in 1 .. 4 (a workgroup of waves_per_simd x 4 wavefronts per CU); *wave_instr_per_s = vector instructions issued per second by all wavefronts together
extern "C" int hmr_gpu_probe_issue(hmr_gpu_ctx *ctx, int op, int waves_per_simd, int dependent, double *wave_instr_per_s, double *ms)
{
	if (!ctx || !wave_instr_per_s || waves_per_simd < 1 || waves_per_simd > 4 || op < 0 || op > 4) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipSetDevice(ctx->device));
	int cus = 0;
	HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
	uint32_t *d_out = nullptr;
	HIP_TRY(hipMalloc((void **)&d_out, 4));
	const int threads = waves_per_simd * 4 * 64, iters = 200000 / (dependent ? 4 : 1);
	const dim3 grid(cus), block(threads);
	float best = 0.f;
	for (int rep = 0; rep < 4; rep++) {      // (the first launch carries the clock ramp)
		HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
		switch (op) {
		case 0: probe_launch<0>(dependent != 0, grid, block, ctx->stream, iters, d_out); break;
		case 1: probe_launch<1>(dependent != 0, grid, block, ctx->stream, iters, d_out); break;
		case 2: probe_launch<2>(dependent != 0, grid, block, ctx->stream, iters, d_out); break;
		case 3: probe_launch<3>(dependent != 0, grid, block, ctx->stream, iters, d_out); break;
		default: probe_launch<4>(dependent != 0, grid, block, ctx->stream, iters, d_out); break;
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
		HIP_TRY(hipEventSynchronize(ctx->ev1));
		float t = 0.f;
		HIP_TRY(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
		if (rep == 0 || t < best) best = t;
	}
	(void)hipFree(d_out);
	const double instr = (double)cus * waves_per_simd * 4 * (double)iters * 64.0;
	*wave_instr_per_s = instr / (best * 1e-3);
	if (ms) *ms = best;
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_probe_valu_issue(hmr_gpu_ctx *ctx, int waves_per_simd, int dependent, double *wave_instr_per_s, double *ms)
{
	return hmr_gpu_probe_issue(ctx, 0, waves_per_simd, dependent, wave_instr_per_s, ms);
}
