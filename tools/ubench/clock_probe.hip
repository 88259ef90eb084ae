// How fast does a dependent ALU chain run when 17 workgroups are resident, and when 255 are?  (Does the shader clock depend on how busy the chip is.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_chain(int iters, unsigned *out, unsigned long long *ticks)
{
	unsigned v = threadIdx.x + 1;
	const unsigned long long t0 = wall_clock64(), c0 = clock64();
	for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;   // 2 dependent VALU ops per iteration
	const unsigned long long t1 = wall_clock64(), c1 = clock64();
	if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = c1 - c0; }
	out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main()
{
	unsigned *out; unsigned long long *ticks;
	hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&ticks, 1024 * 16);
	const int iters = 20 << 20;
	for (int rep = 0; rep < 3; rep++)
		for (int wgs : {1, 17, 64, 136, 255, 17}) {
			hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
			hipEventRecord(a);
			k_chain<<<wgs, 64>>>(iters, out, ticks);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			std::vector<unsigned long long> h(2 * wgs);
			hipMemcpy(h.data(), ticks, 16 * wgs, hipMemcpyDeviceToHost);
			printf("rep %d wgs %3d: %.1f ms, wall ticks %llu (100 MHz -> %.1f ms), clock64 %llu, iterations/us %.1f\n", rep, wgs, ms, h[0], h[0] / 1e5, h[1], iters / (ms * 1e3));
		}
	return 0;
}
