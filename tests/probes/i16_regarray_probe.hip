// Regression probe for profiles/r04_history.md ("a code-generation problem with 16-bit register arrays on this toolchain"): intra_fill_refs once read a lane's (up
// to five) neighbour samples into an int16_t array in registers before the first store - correct in the one-lane build, different streams on the device - and was
// correct again with 32-bit temporaries.  Both forms of that loop, on the reference-sample gather of a 32 x 32 block; prints the number of entries each form got wrong.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ int sample_off(int k, int n, int stride)
{
	if (k < n) return (n + 1 + (n - 1 - k)) * stride;
	if (k < 2 * n) return (n - (k - n)) * stride;
	if (k == 2 * n) return 0;
	if (k <= 3 * n) return k - 2 * n;
	return 1 + n + (k - 3 * n - 1);
}
template <class T>
__global__ void gather(const int16_t *corner, int stride, int n, int16_t *adi)
{
	const int tid = threadIdx.x, size = 4 * n + 1;
	T v[5];
#pragma unroll
	for (int u = 0; u < 5; u++) {
		const int k = tid + 64 * u;
		v[u] = k < size ? (T)corner[sample_off(k, n, stride)] : (T)0;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int u = 0; u < 5; u++) {
		const int k = tid + 64 * u;
		if (k < size) adi[k] = (int16_t)v[u];
	}
}
int main()
{
	const int n = 32, stride = 144, rows = 2 * n + 2, size = 4 * n + 1;
	std::vector<int16_t> h(rows * stride);
	for (size_t i = 0; i < h.size(); i++) h[i] = (int16_t)((i * 7919u + 13u) % 256u);
	int16_t *d, *o;
	hipMalloc(&d, h.size() * 2); hipMalloc(&o, size * 2);
	hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
	int bad[2] = {0, 0};
	for (int form = 0; form < 2; form++) {
		hipMemset(o, 0xff, size * 2);
		if (form == 0) gather<int16_t><<<1, 64>>>(d, stride, n, o); else gather<int32_t><<<1, 64>>>(d, stride, n, o);
		std::vector<int16_t> got(size);
		if (hipMemcpy(got.data(), o, size * 2, hipMemcpyDeviceToHost) != hipSuccess) return 2;
		for (int k = 0; k < size; k++) {
			int off = k < n ? (n + 1 + (n - 1 - k)) * stride : k < 2 * n ? (n - (k - n)) * stride : k == 2 * n ? 0 : k <= 3 * n ? k - 2 * n : 1 + n + (k - 3 * n - 1);
			bad[form] += got[k] != h[off];
		}
	}
	printf("int16_array_wrong=%d int32_temporaries_wrong=%d\n", bad[0], bad[1]);
	return 0;
}
