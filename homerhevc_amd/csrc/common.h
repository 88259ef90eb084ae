// Shared declarations for the gfx950 backend (context, device tables, wave helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/homer_gpu.h"

#define HMR_WAVE 64
#define HMR_BLOCK 256               // 4 waves per workgroup
#define HMR_WAVES_PER_BLOCK (HMR_BLOCK / HMR_WAVE)
#define HMR_MAX_GRID 4096           // default cap, >> 256 CUs; every batched kernel grid-strides over its jobs
extern int g_hmr_max_grid;          // run-time cap (hmr_gpu_set_max_grid), context.cpp

#include "tables_layout.h"

struct hmr_gpu_ctx {
	int device;
	hipStream_t stream;
	bool owns_stream;
	DevTables *tables;              // device
	hipEvent_t ev0, ev1;
	// staging for the host-pointer (drop-in) entries
	uint8_t *h_stage;               // pinned
	uint8_t *d_stage;
	size_t stage_bytes;             // (h_stage / d_stage: allocated by hmr_ctx_need_stage when a drop-in entry first runs)
	int num_cus;
};

void hmr_set_error(const char *fmt, ...);
int hmr_ctx_need_stage(hmr_gpu_ctx *c);      // context.cpp: the drop-in layer's staging buffers, on first use
const DevTables *hmr_host_tables();      // host copy (tables.cpp)

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess) {                                                                \
			hmr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return HMR_GPU_ERR_HIP;                                                            \
		}                                                                                      \
	} while (0)

#define HMR_XCDS 8                  // MI355X: 8 XCDs, each with a private 4 MiB L2; block b is dispatched to XCD b % 8

// grid for `units` block-iterations of work: capped, and a multiple of the XCD count so that xcd_job_range() can give
// every XCD a contiguous share of the batch
static inline int hmr_grid_for_units(long units)
{
	long blocks = units < 1 ? 1 : units;
	if (blocks > g_hmr_max_grid) blocks = g_hmr_max_grid;
	if (blocks >= HMR_XCDS) blocks = (blocks + HMR_XCDS - 1) / HMR_XCDS * HMR_XCDS;
	return (int)blocks;
}
static inline int hmr_grid_for_waves(long njobs_waves)
{
	return hmr_grid_for_units((njobs_waves + HMR_WAVES_PER_BLOCK - 1) / HMR_WAVES_PER_BLOCK);
}

#ifdef __HIPCC__
// XCD-aware batch partition.  Hosts enumerate jobs in CTU order, so neighbouring jobs touch neighbouring samples;
// giving XCD x the contiguous job range [x*per, (x+1)*per) keeps each picture region in ONE XCD's L2 instead of
// spreading it over all eight (placement b % 8 is only used for speed: any placement computes the same result).
struct JobRange { long begin, end, stride; };
// (block, grid) are the launch's blockIdx.x / gridDim.x, or a segment's block index and block count inside a multi-segment launch
// (segments start at multiples of 8 blocks, so block % 8 is still the XCD)
__device__ __forceinline__ JobRange xcd_job_range(long njobs, int jobs_per_block, unsigned block, unsigned grid)
{
	JobRange r;
	if (grid % HMR_XCDS) {   // tiny grids: plain grid-stride
		r.begin = (long)block * jobs_per_block; r.end = njobs; r.stride = (long)grid * jobs_per_block;
		return r;
	}
	const int xcd = block % HMR_XCDS, bi = block / HMR_XCDS, bpx = grid / HMR_XCDS;
	const long per = ((njobs + (long)HMR_XCDS * jobs_per_block - 1) / ((long)HMR_XCDS * jobs_per_block)) * jobs_per_block;
	r.begin = xcd * per + (long)bi * jobs_per_block;
	r.end = (xcd + 1) * per < njobs ? (xcd + 1) * per : njobs;
	r.stride = (long)bpx * jobs_per_block;
	return r;
}
__device__ __forceinline__ JobRange xcd_job_range(long njobs, int jobs_per_block) { return xcd_job_range(njobs, jobs_per_block, blockIdx.x, gridDim.x); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & (HMR_WAVE - 1); }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }

// Butterfly sums.  Inside a row of 16 lanes the exchange is a DPP operand modifier (quad_perm for xor 1 / 2, row_half_mirror and
// row_mirror for the 8- and 16-lane steps: after the quad steps every lane of a quad holds the quad's sum, so mirroring pairs it with
// a lane of the other half) - one VALU instruction per step instead of a ds_bpermute round trip; only the 32- and 64-lane steps go
// through the LDS crossbar.  Every lane of the group ends up with the group's sum.
__device__ __forceinline__ int dpp_xor_sum_row(int v, int steps)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);                     // quad_perm [1,0,3,2]
	if (steps >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
	if (steps >= 3) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);    // row_half_mirror
	if (steps >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);    // row_mirror
	return v;
}
// sum over aligned groups of G lanes (G power of two <= 64)
template <int G, typename T>
__device__ __forceinline__ T group_sum(T v)
{
	if constexpr (sizeof(T) == 4 && __is_integral(T) && G >= 2) {
		constexpr int steps = G >= 16 ? 4 : G == 8 ? 3 : G == 4 ? 2 : 1;
		int x = dpp_xor_sum_row((int)v, steps);
		if constexpr (G >= 32) x += __shfl_xor(x, 16, HMR_WAVE);
		if constexpr (G >= 64) x += __shfl_xor(x, 32, HMR_WAVE);
		return (T)x;
	} else {
#pragma unroll
		for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, HMR_WAVE);
		return v;
	}
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
	return group_sum<HMR_WAVE, T>(v);
}
// Order LDS traffic between the lanes of ONE wave (the compiler only sees per-lane dependences).  Kernels whose waves own
// private LDS regions use this instead of __syncthreads(), so waves of a workgroup never wait for each other.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int clip3i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sat16i(int v) { return clip3i(v, -32768, 32767); }
#endif
