// Small device helpers shared by the batched kernels: 4-sample (8-byte) vector access at 2-byte alignment
// (gfx950 runs with unaligned global access enabled; hipcc emits one global_load_dwordx2 for these copies),
// and wave-uniform job fetch.
#pragma once
#include "common.h"

#ifdef __HIPCC__
struct i16x4 { int16_t v[4]; };

__device__ __forceinline__ i16x4 ld4(const int16_t *p)
{
	i16x4 r;
	__builtin_memcpy(&r, p, 8);
	return r;
}
__device__ __forceinline__ void st4(int16_t *p, const i16x4 &v) { __builtin_memcpy(p, &v, 8); }

// A job index that is the same for the whole wave: moving it to an SGPR lets the descriptor fetch be scalar loads.
__device__ __forceinline__ hmr_gpu_job load_job_uniform(const hmr_gpu_job *__restrict__ jobs, long j)
{
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(j & 0xffffffffu));
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)j >> 32));
	return jobs[((unsigned long)hi << 32) | lo];
}
#endif
