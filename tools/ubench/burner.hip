// Background load for latency experiments: `wgs` workgroups stream through a buffer (mode 0) or run a dependent ALU chain (mode 1) for `seconds`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
__global__ void k_stream(const uint4 *in, uint4 *out, size_t n, int rounds)
{
	uint4 acc = {0, 0, 0, 0};
	for (int r = 0; r < rounds; r++)
		for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
			const uint4 v = in[i];
			acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
			out[i] = acc;
		}
}
__global__ void k_alu(unsigned *out, int iters)
{
	unsigned v = threadIdx.x;
	for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
	out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main(int argc, char **argv)
{
	const int wgs = argc > 1 ? atoi(argv[1]) : 64, mode = argc > 2 ? atoi(argv[2]) : 0;
	const double seconds = argc > 3 ? atof(argv[3]) : 30;
	const size_t n = (size_t)1 << 26;   // 1 GiB each way
	uint4 *a, *b;
	if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) return 1;
	(void)hipMemset(a, 1, n * 16);
	const auto t0 = std::chrono::steady_clock::now();
	int launches = 0;
	while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
		if (mode == 0) k_stream<<<wgs, 256>>>(a, b, n, 1);
		else k_alu<<<wgs, 256>>>((unsigned *)b, 1 << 22);
		(void)hipDeviceSynchronize();
		launches++;
	}
	printf("burner: %d launches of %d workgroups, mode %d\n", launches, wgs, mode);
	return 0;
}
