// Where do the wavefronts of the CTU pool's workgroups land?  512 workgroups x 192 threads with 80 KB of LDS each (two per CU, as k_encode_pool), every wavefront
// reports its HW_ID (SIMD, CU, SE) and XCC_ID.  Answers: do the two workgroups of a CU put their wavefront 0 (the row worker) on the same SIMD, and which
// wavefronts share a SIMD.      hipcc --offload-arch=gfx950 -O2 tools/ubench/placement.hip -o /tmp/placement && /tmp/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(192) void k_where(unsigned *out, int spin)
{
	extern __shared__ unsigned char lds[];
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) {
		const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
		const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID, 4 bits
		out[(blockIdx.x * 3 + wave) * 2] = hw;
		out[(blockIdx.x * 3 + wave) * 2 + 1] = xcc;
		lds[wave] = (unsigned char)hw;
	}
	// stay resident long enough for every workgroup of the grid to have been placed
	for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
}

int main()
{
	const int wgs = 512;
	unsigned *d;
	hipMalloc(&d, wgs * 3 * 2 * sizeof(unsigned));
	hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
	hipLaunchKernelGGL(k_where, dim3(wgs), dim3(192), 80 * 1024, 0, d, 20000);
	hipDeviceSynchronize();
	std::vector<unsigned> h(wgs * 3 * 2);
	hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
	// CU key: xcc, se, sh, cu
	std::map<unsigned, std::vector<std::pair<int, int>>> cus;      // -> (workgroup, wave) with their simd
	std::map<unsigned, std::vector<int>> simd_of;
	int patterns[4][4][4] = {};
	for (int b = 0; b < wgs; b++) {
		int simd[3];
		unsigned key = 0;
		for (int w = 0; w < 3; w++) {
			const unsigned hw = h[(b * 3 + w) * 2], xcc = h[(b * 3 + w) * 2 + 1];
			simd[w] = (hw >> 4) & 3;
			key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
		}
		patterns[simd[0]][simd[1]][simd[2]]++;
		cus[key].push_back({b, simd[0]});
		simd_of[key].push_back(simd[0] | (simd[1] << 2) | (simd[2] << 4));
	}
	printf("CUs used: %zu\n", cus.size());
	for (int a = 0; a < 4; a++)
		for (int b = 0; b < 4; b++)
			for (int c = 0; c < 4; c++)
				if (patterns[a][b][c]) printf("waves 0,1,2 on SIMDs %d,%d,%d: %d workgroups\n", a, b, c, patterns[a][b][c]);
	int same = 0, pairs = 0, w0_shared = 0;
	std::map<int, int> per_cu;
	for (auto &kv : simd_of) {
		per_cu[(int)kv.second.size()]++;
		if (kv.second.size() == 2) {
			pairs++;
			const int a = kv.second[0], b = kv.second[1];
			if ((a & 3) == (b & 3)) same++;
			// is wave 0 of either workgroup on a SIMD that holds any wave of the other?
			for (int x = 0; x < 2; x++) {
				const int me = x ? b : a, other = x ? a : b;
				for (int w = 0; w < 3; w++)
					if (((other >> (2 * w)) & 3) == (me & 3)) { w0_shared++; break; }
			}
		}
	}
	for (auto &kv : per_cu) printf("CUs with %d workgroups: %d\n", kv.first, kv.second);
	printf("CU pairs: %d; both wave 0 on the same SIMD: %d; wave 0 shares its SIMD with a wave of the other workgroup: %d of %d\n", pairs, same, w0_shared, 2 * pairs);
	int shown = 0;
	for (auto &kv : simd_of)
		if (shown++ < 6) {
			printf("cu %06x:", kv.first);
			for (int v : kv.second) printf("  [%d %d %d]", v & 3, (v >> 2) & 3, (v >> 4) & 3);
			printf("\n");
		}
	return 0;
}
