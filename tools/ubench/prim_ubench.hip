// GPU box tool: cycles per call of the CTU encoder's block primitives as ONE wavefront sees them (the situation inside k_encode_ctus):
// operands in LDS or in global memory, calls back to back.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
// -I homerhevc_amd/csrc tools/ubench/prim_ubench.hip homerhevc_amd/csrc/tables.cpp -o tools/ubench/prim_ubench
#include <stdio.h>
#include <vector>
#include "common.h"
#include "enc/enc_prims.h"

using namespace henc;

struct Args {
	const DevTables *T;
	int16_t *ga, *gb, *gc;      // global scratch planes (64 x 64 each, stride 64) and a "reference picture" (gc, stride 2080)
	unsigned long long *out;    // [case] cycles for `reps` calls
	uint32_t *sink;
	int reps;
};

#define CASE(idx, stmt)                                                     \
	do {                                                                \
		__syncthreads();                                            \
		const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
		for (int r = 0; r < a.reps; r++) { stmt; }                  \
		__syncthreads();                                            \
		const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
		if (threadIdx.x == 0) a.out[idx] = t1 - t0;                 \
	} while (0)

__global__ __launch_bounds__(64) void k_ubench(Args a)
{
	extern __shared__ __align__(16) uint8_t lds[];
	int16_t *la = (int16_t *)lds, *lb = la + 4096, *lc = lb + 4096, *ld = lc + 4096;
	WaveGrp g{(int)threadIdx.x};
	for (int i = g.tid; i < 4096; i += 64) { la[i] = (int16_t)((i * 7) & 255); lb[i] = (int16_t)((i * 13) & 255); lc[i] = 0; ld[i] = 0; }
	__syncthreads();
	uint32_t acc = 0;
	int n;
	// 0-3: SAD LDS/LDS n = 8,16,32,64
	n = 8;  CASE(0, acc += blk_sad(g, la, 64, lb, 64, n));
	n = 16; CASE(1, acc += blk_sad(g, la, 64, lb, 64, n));
	n = 32; CASE(2, acc += blk_sad(g, la, 64, lb, 64, n));
	n = 64; CASE(3, acc += blk_sad(g, la, 64, lb, 64, n));
	// 4-5: SAD LDS vs global reference (the motion search)
	n = 8;  CASE(4, acc += blk_sad(g, la, 64, a.gc + (r & 63) * 2080 + (r & 31), 2080, n));
	n = 16; CASE(5, acc += blk_sad(g, la, 64, a.gc + (r & 63) * 2080 + (r & 31), 2080, n));
	// 6-9: forward transform 4, 8, 16, 32 (LDS)
	n = 4;  CASE(6, tr_forward(g, a.T, la, 64, lc, ld, n, 0));
	n = 8;  CASE(7, tr_forward(g, a.T, la, 64, lc, ld, n, 0));
	n = 16; CASE(8, tr_forward(g, a.T, la, 64, lc, ld, n, 0));
	n = 32; CASE(9, tr_forward(g, a.T, la, 64, lc, ld, n, 0));
	// 10-11: quantize 8, 32 (LDS in, global out like the tq windows)
	n = 8;  CASE(10, acc += quantize(g, a.T, lc, a.ga, ld, 3, 3, 0, 0, 0, 1, n, 5, 2));
	n = 32; CASE(11, acc += quantize(g, a.T, lc, a.ga, ld, 3, 1, 0, 0, 0, 1, n, 5, 2));
	// 12-13: luma interpolation stage 8x8 and 32x32 horizontal from global, vertical from LDS
	n = 8;  CASE(12, interp_stage<8>(g, a.gc + 8 * 2080 + 8, 2080, lc, 72, 2, n, n + 7, 0, 1, 0));
	n = 8;  CASE(13, interp_stage<8>(g, lc + 3 * 72, 72, ld, 64, 2, n, n, 1, 0, 1));
	n = 32; CASE(14, interp_stage<8>(g, a.gc + 8 * 2080 + 8, 2080, lc, 72, 2, n, n + 7, 0, 1, 0));
	// 15-16: intra prediction + SAD 8x8 angular / planar
	n = 8;  CASE(15, acc += intra_predict_sad(g, lc, 64, la, 64, lb, n, 20 + (r & 7), 1));
	n = 8;  CASE(16, acc += intra_predict_sad(g, lc, 64, la, 64, lb, n, 0, 1));
	n = 32; CASE(17, acc += intra_predict_sad(g, lc, 64, la, 64, lb, n, 20 + (r & 7), 1));
	// 18: predict (residual) 8x8; 19: reconst to global; 20: copy 8x8 global->global
	n = 8;  CASE(18, blk_predict(g, la, 64, lb, 64, lc, 64, n));
	n = 8;  CASE(19, blk_reconst(g, la, 64, lc, 64, a.ga, 144, n));
	n = 8;  CASE(20, blk_copy(g, a.ga, 144, a.gb, 144, n, n));
	// 21: empty sync; 22: wave sum alone
	CASE(21, g.sync());
	CASE(22, acc += g.sum(acc));
	// 23: inverse transform 8; 24: dequantize 8 from global
	n = 8;  CASE(23, tr_inverse(g, a.T, lc, 64, la, ld, n, 0));
	n = 8;  CASE(24, dequantize(g, a.T, a.ga, lc, 3, 0, 0, n, 5, 2));
	if (acc == 0x12345678) a.sink[0] = acc;
}

int main()
{
	Args a;
	DevTables *T;
	hipMalloc((void **)&T, sizeof(DevTables));
	hipMemcpy(T, hmr_host_tables(), sizeof(DevTables), hipMemcpyHostToDevice);
	a.T = T;
	hipMalloc((void **)&a.ga, 144 * 144 * 2); hipMalloc((void **)&a.gb, 144 * 144 * 2); hipMalloc((void **)&a.gc, 2080 * 256 * 2);
	hipMemset(a.ga, 1, 144 * 144 * 2); hipMemset(a.gb, 2, 144 * 144 * 2); hipMemset(a.gc, 3, 2080 * 256 * 2);
	hipMalloc((void **)&a.out, 32 * 8); hipMalloc((void **)&a.sink, 4);
	a.reps = 200;
	hipFuncSetAttribute((const void *)k_ubench, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k_ubench, dim3(1), dim3(64), 4 * 4096 * 2, 0, a);
	hipDeviceSynchronize();
	unsigned long long out[32];
	hipMemcpy(out, a.out, sizeof out, hipMemcpyDeviceToHost);
	const char *names[25] = {"sad8 lds", "sad16 lds", "sad32 lds", "sad64 lds", "sad8 lds/global", "sad16 lds/global", "tr_fwd4", "tr_fwd8", "tr_fwd16", "tr_fwd32", "quant8", "quant32",
				 "interp8 H 8x15 global", "interp8 V 8x8 lds", "interp32 H 32x39 global", "intra_sad8 ang", "intra_sad8 planar", "intra_sad32 ang", "predict8", "reconst8 ->global",
				 "copy8 global", "sync", "wave sum", "tr_inv8", "dequant8 global"};
	for (int i = 0; i < 25; i++) printf("%-28s %8.0f cycles/call (s_memtime ticks)\n", names[i], (double)out[i] / a.reps);
	return 0;
}
