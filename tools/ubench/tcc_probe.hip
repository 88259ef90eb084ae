// Known-byte access patterns for calibrating the L2's memory-side request counters (TCC_EA0_RDREQ / TCC_EA0_WRREQ, MI355X_MICROARCH.md "HBM": widths other than
// the wide coalesced read are uncalibrated).  One kernel per pattern, each over a buffer larger than the 256 MB Infinity Cache, so that
//     rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RD_UNCACHED_32B_sum -- tools/ubench/tcc_probe
// gives requests per kernel next to the bytes each kernel is known to move (printed by the program; tools/tcc_calibrate.py joins the two).
// The patterns are those of the CTU kernel: 16-bit samples in short rows of a strided window, single bytes of a side-info record, 4-byte reads of 8-bit plane rows.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t BUF = (size_t)1 << 30;      // 1 GiB

// 16 bytes per lane, fully coalesced
__global__ void w_wide16(uint4 *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(i, 1, 2, 3); }
__global__ void r_wide16(const uint4 *p, size_t n, uint32_t *sink) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = p[i]; if (v.x == 0xdeadbeef) *sink = v.y; } }
// 2 bytes per lane, contiguous (a wavefront writes 128 contiguous bytes)
__global__ void w_short_contig(int16_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (int16_t)i; }
// an 8 x 8 block of int16 in a window of 144 samples per row: 8 lanes x 2 bytes = 16 bytes per row, rows 288 bytes apart; one block per wavefront, blocks 4 KB apart
__global__ void w_block8_s16(int16_t *p, size_t blocks)
{
	size_t b = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	int l = threadIdx.x & 63;
	if (b < blocks) p[b * 2048 + (l >> 3) * 144 + (l & 7)] = (int16_t)l;
}
__global__ void r_block8_s16(const int16_t *p, size_t blocks, uint32_t *sink)
{
	size_t b = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	int l = threadIdx.x & 63;
	if (b < blocks) { int16_t v = p[b * 2048 + (l >> 3) * 144 + (l & 7)]; if (v == 12345) *sink = 1; }
}
// 64 single bytes per wavefront, contiguous (a side-info array of a CTU record)
__global__ void w_bytes_contig(uint8_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i * 1] = (uint8_t)i; }
// 64 single bytes per wavefront, one per 64-byte line
__global__ void w_bytes_sparse(uint8_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i * 64] = (uint8_t)i; }
// 4 bytes per lane from rows of an 8-bit plane: 16 lanes x 4 bytes = a 64-byte row, rows 2080 bytes apart (a 64 x 4 piece of a phase plane per wavefront)
__global__ void r_plane_rows(const uint8_t *p, size_t pieces, uint32_t *sink)
{
	size_t b = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	int l = threadIdx.x & 63;
	if (b < pieces) { uint32_t v = *(const uint32_t *)(p + b * 8320 + (l >> 4) * 2080 + (l & 15) * 4); if (v == 0xdeadbeef) *sink = 1; }
}

int main()
{
	void *buf;
	uint32_t *sink;
	hipMalloc(&buf, BUF);
	hipMalloc(&sink, 4);
	hipMemset(buf, 1, BUF);
	hipDeviceSynchronize();
	const int T = 256;
	struct { const char *name; double bytes; } rows[8];
	int k = 0;
	{ size_t n = BUF / 16; hipLaunchKernelGGL(w_wide16, dim3((n + T - 1) / T), dim3(T), 0, 0, (uint4 *)buf, n); rows[k++] = {"w_wide16", (double)n * 16}; }
	{ size_t n = BUF / 16; hipLaunchKernelGGL(r_wide16, dim3((n + T - 1) / T), dim3(T), 0, 0, (const uint4 *)buf, n, sink); rows[k++] = {"r_wide16", (double)n * 16}; }
	{ size_t n = BUF / 2; hipLaunchKernelGGL(w_short_contig, dim3((n + T - 1) / T), dim3(T), 0, 0, (int16_t *)buf, n); rows[k++] = {"w_short_contig", (double)n * 2}; }
	{ size_t blocks = BUF / 4096; hipLaunchKernelGGL(w_block8_s16, dim3((blocks * 64 + T - 1) / T), dim3(T), 0, 0, (int16_t *)buf, blocks); rows[k++] = {"w_block8_s16", (double)blocks * 128}; }
	{ size_t blocks = BUF / 4096; hipLaunchKernelGGL(r_block8_s16, dim3((blocks * 64 + T - 1) / T), dim3(T), 0, 0, (const int16_t *)buf, blocks, sink); rows[k++] = {"r_block8_s16", (double)blocks * 128}; }
	{ size_t n = BUF / 4; hipLaunchKernelGGL(w_bytes_contig, dim3((n + T - 1) / T), dim3(T), 0, 0, (uint8_t *)buf, n); rows[k++] = {"w_bytes_contig", (double)n}; }
	{ size_t n = BUF / 64; hipLaunchKernelGGL(w_bytes_sparse, dim3((n + T - 1) / T), dim3(T), 0, 0, (uint8_t *)buf, n); rows[k++] = {"w_bytes_sparse", (double)n}; }
	{ size_t pieces = BUF / 8320; hipLaunchKernelGGL(r_plane_rows, dim3((pieces * 64 + T - 1) / T), dim3(T), 0, 0, (const uint8_t *)buf, pieces, sink); rows[k++] = {"r_plane_rows", (double)pieces * 256}; }
	hipDeviceSynchronize();
	for (int i = 0; i < k; i++) printf("%s %.0f\n", rows[i].name, rows[i].bytes);
	return 0;
}
