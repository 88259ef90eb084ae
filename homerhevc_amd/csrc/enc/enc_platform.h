// Execution-group abstraction of the CTU encoder.
//
// The encoder core (enc_*.h) is written once as SPMD code: every function is executed by all lanes of a group with
// group-uniform control flow; pixel loops are strided by the lane index and reductions go through the group.
//   * product: hipcc, HENC_HD = __device__, group = one 64-lane wavefront that owns a CTU (k_encode.hip);
//   * checker: g++ (oracle/enc_cpu.cpp, test infrastructure only), group = one lane - the same decision logic runs
//     serially there so that it can be diffed against the compiled reference in the build container, where there is no GPU.
// The product library never contains the one-lane instantiation: there is no CPU fallback.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HENC_HD __device__
#if defined(HENC_PRIM_NOINLINE)
#define HENC_PRIM __device__ __attribute__((noinline))
#else
#define HENC_PRIM __device__   // block primitives: the compiler decides (forcing them out of line does not pay: all of them +3 %, only the large ones - intra prediction, SSD, reference fill - +2 %)
#endif
#define HENC_INLINE __host__ __device__ __forceinline__   // the small helpers are also used by the host entropy stage
// Are the TU tables (FastTables, enc_prims.h) kept in the worker's fast memory?  Not on the device since two workers share a CU's LDS (k_encode.hip:
// LDS_KEEPS_TU_TABLES must agree) - a constant, so that the transform / quantiser primitives carry one table source instead of a run-time choice of two.
#define HENC_TU_TABLES_IN_LDS 0
#define HENC_FT(e) ((const FastTables *)nullptr)
#else
#define HENC_HD
#define HENC_PRIM
#define HENC_INLINE inline
#define HENC_TU_TABLES_IN_LDS 1
#define HENC_FT(e) ((e).ft)
#endif

// The three CU-level evaluations (encode_inter, encode_intra_luma, encode_intra_chroma): left to the compiler (it keeps them out of line), or forced into their callers
// (the default; -DHENC_WALK_OUTLINE leaves them to the compiler: no call frames - a call saves and restores up to a hundred callee-saved registers through private memory - against a larger body)
#if defined(__HIPCC__) && !defined(HENC_WALK_OUTLINE)
#define HENC_WALK_FN __attribute__((always_inline))
#else
#define HENC_WALK_FN
#endif

namespace henc {

#if defined(__HIPCC__)
// One wavefront.  Scalar state is computed redundantly by all lanes (uniform), so stores of uniform values by every
// lane are benign; sync() orders the data-parallel producer / consumer phases that go through memory.
struct WaveGrp {
	int tid;
	static constexpr int n = 64;
	static constexpr bool bg = false;      // (WaveGrpLat: the walk of the latency kernel, whose helper runs background intra searches - enc_common.h bg_post)
	// the group is ONE wavefront: its lanes run in lockstep, so ordering its own memory operations is all a "barrier" has to do (the workgroup may hold
	// helper wavefronts that are doing something else, see HelperBox in enc_common.h)
	__device__ __forceinline__ void sync() const
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	// wave-wide sum without LDS traffic: row shifts (DPP) inside the 16-lane rows, then the four row totals through readlane
	__device__ __forceinline__ uint32_t sum(uint32_t v) const
	{
		int x = (int)v;
		x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
		x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
		x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
		x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8 -> lane 15 of every row holds the row total
		return (uint32_t)(__builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63));
	}
	__device__ __forceinline__ int64_t sum64(int64_t v) const
	{
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
		return v;
	}
	__device__ __forceinline__ uint32_t any(bool p) const { return __ballot(p) != 0; }
	__device__ __forceinline__ uint64_t ballot(bool p) const { return __ballot(p); }
	// smallest key over the lanes (ties: the key itself breaks them)
	__device__ __forceinline__ uint64_t min64(uint64_t v) const
	{
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) {
			const uint64_t o = __shfl_xor(v, m, 64);
			v = o < v ? o : v;
		}
		return v;
	}
};
// the same wavefront as the group of the latency kernel's walk (k_encode_pool_lat): a type of its own, so that the walk is compiled a second time WITH the background-search
// hooks and the throughput kernel's code stays exactly what it was (a batch pays for every instruction the hooks add: -1.3 % when both kernels shared the walk)
struct WaveGrpLat : WaveGrp {
	static constexpr bool bg = true;
};

// Two groups of 32 lanes in one wavefront, each with a block of its own (the helper's two chroma planes of a small TU: a 4 x 4 or 8 x 8 chain keeps 4 - 16 lanes busy,
// and two of them one after the other made the helper the slower side of a small CU).  The halves run the same code on different operands; where their control flow
// parts (one plane has levels, the other has not) the hardware masks the lanes, and nothing in the group operations crosses the halves: sums add up a half's two
// 16-lane rows, the ballot is the half's 32 bits.
struct PairGrp {
	int tid, half;           // lane within the half, the half (0 / 1)
	static constexpr int n = 32;
	static constexpr bool bg = false;
	__device__ __forceinline__ void sync() const
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	__device__ __forceinline__ uint32_t sum(uint32_t v) const
	{
		int x = (int)v;
		x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
		x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // lane 15 of every 16-lane row: the row total
		const int lo = __builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31), hi = __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63);
		return (uint32_t)(half ? hi : lo);
	}
	__device__ __forceinline__ uint64_t ballot(bool p) const { const uint64_t m = __ballot(p); return half ? m >> 32 : m & 0xffffffffull; }
	__device__ __forceinline__ uint32_t any(bool p) const { return ballot(p) != 0; }
};

#endif

// The row worker's state - its Work, the CTU's nodes and record, the geometry, the sequence / frame parameters, the helper mailbox and scratch - lives in
// LDS on the device (k_encode.hip).  The pointers to it travel through Enc as generic pointers, which the compiler can only serve with flat_* instructions:
// 64-bit address arithmetic, and a wait on every outstanding store before any load result is used.  FastPtr says where they point (a cast to the LDS address
// space and back, which address-space inference propagates into everything derived from the pointer), so these accesses become ds_* instructions.
// On the CPU (checker build, host pass) it is an ordinary pointer.
// A value every lane of the group holds (it was read at a group-uniform address, or computed from such values): on the device this says so - the value moves to a
// scalar register, what is computed from it is scalar arithmetic and branches on it are scalar branches instead of execution-mask regions.  The compiler cannot
// see it by itself where the address is a generic pointer (a flat load is a source of divergence for it).
template <class T>
HENC_INLINE T uni(T x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	static_assert(sizeof(T) <= 4, "one scalar register");
	return (T)__builtin_amdgcn_readfirstlane((int)x);
#else
	return x;
#endif
}
// ... and a pointer every lane holds (an argument of a function the compiler keeps out of line arrives in vector registers even when the caller had it in scalar ones:
// uniform again, the address arithmetic behind it is scalar and a switch on a uniform size is a scalar branch instead of execution-mask regions)
template <class T>
HENC_INLINE T *uni_ptr(T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const uintptr_t v = (uintptr_t)p;
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
	return (T *)(((uintptr_t)hi << 32) | (uintptr_t)lo);
#else
	return p;
#endif
}
template <class T>
HENC_INLINE T *in_fast_memory(T *p)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(HENC_CHECK_ADDRSPACE)
	if (p && !__builtin_amdgcn_is_shared((const void *)p)) __builtin_trap();
	return p;
#elif defined(__HIP_DEVICE_COMPILE__)
	__builtin_assume(__builtin_amdgcn_is_shared((const void *)p));
	return p;
#else
	return p;
#endif
}
// The worker's context (Enc, enc_common.h) is ONE object per wavefront at a fixed place in LDS; it reaches the functions of the walk as a reference, which the compiler
// can only take for a generic pointer (flat_* accesses, and - when the object was a local of the kernel - private memory behind it).  Every function that gets it says
// where it is: its members then are ds_* accesses at an address the lanes share.
// -DHENC_CHECK_ADDRSPACE (a debug build, tools/build_variant.sh): the promises become checks - a pointer that is not in LDS traps instead of being read through
// ds_* instructions at a wrong address (the GPU stream tests run once on such a build per round: profiles/r06_history.md)
#if defined(__HIP_DEVICE_COMPILE__) && defined(HENC_CHECK_ADDRSPACE)
#define HENC_ENC_IN_LDS(e) do { if (!__builtin_amdgcn_is_shared((const void *)&(e))) __builtin_trap(); } while (0)
#elif defined(__HIP_DEVICE_COMPILE__)
#define HENC_ENC_IN_LDS(e) __builtin_assume(__builtin_amdgcn_is_shared((const void *)&(e)))
#else
#define HENC_ENC_IN_LDS(e) do { } while (0)
#endif
// The operands of the TU primitives (source / prediction windows, coefficient, level and remainder buffers of the TU in flight) are in the worker's LDS wherever the
// encoder kernel calls them; the primitives are functions of their own with generic pointer parameters, i.e. flat_* accesses that wait for LDS and memory together.
// k_encode.hip defines HENC_TU_OPERANDS_IN_LDS and the primitives say so per operand; the test harness (k_primtest.hip: operands in HBM) does not.
#if defined(__HIP_DEVICE_COMPILE__) && defined(HENC_TU_OPERANDS_IN_LDS) && defined(HENC_CHECK_ADDRSPACE)
#define HENC_OP_IN_LDS(p) do { if (!__builtin_amdgcn_is_shared((const void *)(p))) __builtin_trap(); } while (0)
#elif defined(__HIP_DEVICE_COMPILE__) && defined(HENC_TU_OPERANDS_IN_LDS)
#define HENC_OP_IN_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p)))
#else
#define HENC_OP_IN_LDS(p) do { } while (0)
#endif
// ... and a pointer into HBM (the worker's slow windows): kept as a pointer of the global address space, what is derived from it are global_* accesses instead of
// flat_* ones (which wait for LDS and memory operations alike and decide per lane where they go)
#if defined(__HIP_DEVICE_COMPILE__)
#define HENC_GLOBAL_PTR(T) __attribute__((address_space(1))) T *
#else
#define HENC_GLOBAL_PTR(T) T *
#endif
template <class T>
struct FastPtr {
	T *p;
	HENC_INLINE T *get() const { return in_fast_memory(p); }
	HENC_INLINE T *operator->() const { return get(); }
	HENC_INLINE T &operator*() const { return *get(); }
	HENC_INLINE T &operator[](int i) const { return get()[i]; }
	HENC_INLINE operator T *() const { return get(); }
	HENC_INLINE explicit operator bool() const { return p != nullptr; }
	HENC_INLINE FastPtr &operator=(T *q) { p = q; return *this; }
};

#if defined(__HIP_DEVICE_COMPILE__)
// A member that IS a fixed place in the workgroup's LDS (the worker's Work, the CTU's nodes, the sequence / frame parameters, the helper mailbox: k_encode.hip lays
// them out at constant offsets).  As FastPtr members of Enc these pointers were data - and Enc, whose address goes to every out-of-line function of the walk, lives
// in private memory: each `e.w->` began with a scratch load.  An LdsAt holds nothing; assignments to it are accepted and ignored (the kernels still "set" them).
extern __shared__ __align__(16) unsigned char henc_lds_base[];
template <class T, int OFFSET>
struct LdsAt {
	HENC_INLINE T *get() const { return (T *)(henc_lds_base + OFFSET); }
	HENC_INLINE T *operator->() const { return get(); }
	HENC_INLINE T &operator*() const { return *get(); }
	HENC_INLINE T &operator[](int i) const { return get()[i]; }
	HENC_INLINE operator T *() const { return get(); }
	HENC_INLINE explicit operator bool() const { return true; }
	HENC_INLINE LdsAt &operator=(T *) { return *this; }
	HENC_INLINE LdsAt &operator=(decltype(nullptr)) { return *this; }
};
#endif

struct CpuGrp {
	static constexpr int tid = 0;
	static constexpr int n = 1;
	static constexpr bool bg = false;
	void sync() const {}
	uint32_t sum(uint32_t v) const { return v; }
	int64_t sum64(int64_t v) const { return v; }
	uint32_t any(bool p) const { return p; }
	uint64_t ballot(bool p) const { return p ? 1 : 0; }
	uint64_t min64(uint64_t v) const { return v; }
};

template <class T> HENC_INLINE T hmin(T a, T b) { return a < b ? a : b; }
template <class T> HENC_INLINE T hmax(T a, T b) { return a > b ? a : b; }
template <class T> HENC_INLINE T hclip(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
HENC_INLINE int habs(int v) { return v < 0 ? -v : v; }
HENC_INLINE double hsqrt(double v) { return ::sqrt(v); }      // (correctly rounded on both sides: libm here, __ocml_sqrt_f64 on the device)
HENC_INLINE int16_t sat16(int v) { return (int16_t)hclip(v, -32768, 32767); }

}  // namespace henc
