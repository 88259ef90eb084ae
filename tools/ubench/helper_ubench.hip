// GPU box tool: round-trip cost of handing a job to a helper wavefront through LDS (the HelperBox protocol of enc_common.h), one workgroup of 192 threads.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(192) void k(unsigned long long *out, int reps)
{
	__shared__ int cmd[2], done[2], work[2];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (threadIdx.x < 2) { cmd[threadIdx.x] = 0; done[threadIdx.x] = 0; }
	__syncthreads();
	if (wave > 0) {
		const int h = wave - 1;
		for (int seq = 1; seq <= 2 * reps + 1; seq++) {
			while (__hip_atomic_load(&cmd[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
			if (lane == 0) { work[h] += seq; __hip_atomic_store(&done[h], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
		}
		return;
	}
	int seq = 0;
	// one helper, empty job
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int r = 0; r < reps; r++) {
		seq++;
		if (lane == 0) __hip_atomic_store(&cmd[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		while (__hip_atomic_load(&done[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	// same without s_sleep in the master's wait
	for (int r = 0; r < reps; r++) {
		seq++;
		if (lane == 0) __hip_atomic_store(&cmd[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		while (__hip_atomic_load(&done[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) {}
	}
	unsigned long long t2 = __builtin_amdgcn_s_memtime();
	if (lane == 0) { out[0] = (t1 - t0) / reps; out[1] = (t2 - t1) / reps; }
	seq++;
	if (lane == 0) __hip_atomic_store(&cmd[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	if (lane == 0) for (int s2 = 1; s2 <= 2 * reps + 1; s2++) { __hip_atomic_store(&cmd[1], s2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); while (__hip_atomic_load(&done[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != s2) {} }
}
int main()
{
	unsigned long long *d, h[2];
	hipMalloc((void **)&d, 16);
	hipLaunchKernelGGL(k, dim3(1), dim3(192), 0, 0, d, 1000);
	hipDeviceSynchronize();
	hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("helper round trip: %llu ticks (master sleeps), %llu ticks (master spins)\n", h[0], h[1]);
	return 0;
}
