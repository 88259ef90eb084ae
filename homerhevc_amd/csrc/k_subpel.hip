// Phase planes of a reference picture (include/homer_gpu.h section 13).
//
// The reference interpolates a block every time a vector is tried: the 16 planes of hmr_half/quarter_pixel_estimation_luma_hm
// (hmr_motion_inter.c:395,442) per motion search, hmr_motion_compensation_luma / _chroma (:1779,1860) per merge candidate and
// per coded vector - all through sse_interpolate_luma / _chroma (inter_prediction.c:796,818; stage rules hmr_motion_inter.c:240-391).
// The value of a prediction sample depends only on the vector's phase and on the reference samples around it, not on the block it
// is asked for, so here the whole padded reference is interpolated ONCE per frame for every phase:
//   luma    16 planes, plane (fy * 4 + fx) = the picture displaced by (fx, fy) quarter samples,
//   chroma  64 planes per component, plane (fy * 8 + fx) in eighth samples,
// stored as bytes with the picture's own row length and margins, ROW-INTERLEAVED: row y of plane f starts at byte (y * planes + f) * stride, so that what a
// CTU reads of all the planes (its search window, a few hundred rows) is one contiguous stretch of memory - a handful of pages instead of one per plane.  Inside the CTU walk (enc/enc_inter.h) motion compensation is then a
// copy and a motion-search candidate is v_sad_u8 against a plane.
//
// These kernels are streaming, store-bound work: a luma sample is read once (2 bytes at the picture's int16 width) and 16 bytes are
// written; a chroma sample 2 in, 64 out.  One workgroup = a 64 x 16 tile: the tile and its filter margin are staged in LDS, the
// horizontal first stages (14-bit intermediates, as the reference keeps them) are computed once per row and phase into LDS, the
// vertical stage reads them back; every thread finishes four adjacent samples of all phases and stores one dword per plane, so a
// wavefront writes 256 contiguous bytes per plane and row.  Addressing is linear over the allocation (taps that run over a row end
// read the neighbouring row's memory exactly as the reference's pointer arithmetic does); taps outside the allocation read zero.
#include "common.h"

namespace {

constexpr int TW = 64, TH = 16;

// One output byte.  The values clipped here provably fit 16 bits and the compiler clips them with a 16-bit median (v_med3_i16), which leaves the upper half
// of its destination register as it was; packed with shift-and-or that garbage lands in the neighbouring byte (seen on gfx950, ROCm 7.2: bits of byte 2 set
// from sample 0's register).  The empty asm hides the value's range, so the mask after it is a real instruction.
__device__ __forceinline__ uint32_t clip255(int v)
{
	v = v < 0 ? 0 : (v > 255 ? 255 : v);
	asm volatile("" : "+v"(v));
	return (uint32_t)v & 255u;
}

// Workgroups go to the eight XCDs in turn (blockIdx mod 8), each with an L2 of its own.  A tile row is 64 bytes of a plane row, half a 128-byte line: with
// neighbouring tiles on different XCDs every line leaves two L2s half written.  This gives XCD k the k-th eighth of the tiles instead, so that the tiles that share
// lines (and the rows of the picture that neighbouring tiles both read) meet in one L2.  The grid is rounded up to a multiple of eight; -1: no tile.
__device__ __forceinline__ int xcd_contiguous(unsigned block, unsigned blocks)
{
	const unsigned per_xcd = blocks / 8, tile = (block % 8) * per_xcd + block / 8;
	return (int)tile;
}

typedef short short2_t __attribute__((ext_vector_type(2)));
// four results (each inside the 16-bit range) clipped to 0..255 and packed into a dword: two v_perm_b32 to pair them up as 16-bit halves, a packed max and a packed
// min per pair, one v_perm_b32 to pick the four low bytes - 7 instructions instead of a median, a mask, a shift and an or per byte
__device__ __forceinline__ uint32_t pack4_clip(int r0, int r1, int r2, int r3)
{
	const short2_t lo = {0, 0}, hi = {255, 255};
	uint32_t a = __builtin_amdgcn_perm((uint32_t)r1, (uint32_t)r0, 0x05040100u), b = __builtin_amdgcn_perm((uint32_t)r3, (uint32_t)r2, 0x05040100u);
	short2_t va, vb;
	__builtin_memcpy(&va, &a, 4);
	__builtin_memcpy(&vb, &b, 4);
	va = __builtin_elementwise_min(__builtin_elementwise_max(va, lo), hi);
	vb = __builtin_elementwise_min(__builtin_elementwise_max(vb, lo), hi);
	__builtin_memcpy(&a, &va, 4);
	__builtin_memcpy(&b, &vb, 4);
	return __builtin_amdgcn_perm(b, a, 0x06040200u);
}

typedef short short4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ short4_t lds4(const int16_t *p) { return *(const short4_t *)p; }      // four samples with one 8-byte LDS read (p 8-byte aligned)

// The tiles keep sample x0 + c at column c + 4 (a left margin of four, of which the filters use three / one): the four samples a thread finishes start on an
// 8-byte boundary, so every LDS access of the hot loops is a 64-bit one.

// out = [rows][16][stride] bytes; pic = allocation start; elems = stride * rows
__global__ __launch_bounds__(256) void k_subpel_luma(const int16_t *__restrict__ pic, int stride, int rows, uint8_t *__restrict__ out)
{
	__shared__ __attribute__((aligned(16))) int16_t in[TH + 7][TW + 8];          // rows y0-3 .. y0+TH+3, columns x0-4 .. x0+TW+3
	__shared__ __attribute__((aligned(16))) int16_t hs[3][TH + 7][TW];           // horizontal stage (sum - 8192) for fx = 1, 2, 3
	const int tiles_x = (stride + TW - 1) / TW;
	const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
	const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
	if (y0 >= rows) return;
	const long elems = (long)stride * rows;
	const int t = (int)threadIdx.x;
	for (int i = t; i < (TH + 7) * (TW + 8); i += 256) {
		const int r = i / (TW + 8), c = i - r * (TW + 8);
		const long li = (long)(y0 - 3 + r) * stride + (x0 - 4 + c);
		in[r][c] = (li >= 0 && li < elems) ? pic[li] : (int16_t)0;
	}
	__syncthreads();
	const int c1[8] = {-1, 4, -10, 58, 17, -5, 1, 0}, c2[8] = {-1, 4, -11, 40, 40, -11, 4, -1}, c3[8] = {0, 1, -5, 17, 58, -10, 4, -1};
	// horizontal stage: four adjacent outputs per work item from twelve samples (three 64-bit reads)
	for (int i = t; i < (TH + 7) * (TW / 4); i += 256) {
		const int r = i / (TW / 4), c = (i - r * (TW / 4)) << 2;
		const short4_t a = lds4(&in[r][c]), b = lds4(&in[r][c + 4]), d = lds4(&in[r][c + 8]);
		const int sm[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};      // columns c .. c + 11 of the tile = samples x0 + c - 4 ...
		short4_t o1, o2, o3;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			int s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int v = sm[j + 1 + k];                                                        // sample (c + j) - 3 + k
				s1 += v * c1[k]; s2 += v * c2[k]; s3 += v * c3[k];
			}
			o1[j] = (short)(s1 - 8192); o2[j] = (short)(s2 - 8192); o3[j] = (short)(s3 - 8192);
		}
		*(short4_t *)&hs[0][r][c] = o1; *(short4_t *)&hs[1][r][c] = o2; *(short4_t *)&hs[2][r][c] = o3;
	}
	__syncthreads();
	const int ty = t >> 4, tx = (t & 15) << 2;
	const int y = y0 + ty, x = x0 + tx;
	if (y >= rows || x >= stride) return;
	uint32_t pk[16];
	short4_t v0[8], v1[8], v2[8], v3[8];
#pragma unroll
	for (int k = 0; k < 8; k++) { v0[k] = lds4(&in[ty + k][tx + 4]); v1[k] = lds4(&hs[0][ty + k][tx]); v2[k] = lds4(&hs[1][ty + k][tx]); v3[k] = lds4(&hs[2][ty + k][tx]); }
	// fy = 0: the integer sample and the three horizontal phases (single stage: (sum + 32) >> 6)
	pk[0] = pack4_clip(v0[3][0], v0[3][1], v0[3][2], v0[3][3]);
	pk[1] = pack4_clip((v1[3][0] + 8192 + 32) >> 6, (v1[3][1] + 8192 + 32) >> 6, (v1[3][2] + 8192 + 32) >> 6, (v1[3][3] + 8192 + 32) >> 6);
	pk[2] = pack4_clip((v2[3][0] + 8192 + 32) >> 6, (v2[3][1] + 8192 + 32) >> 6, (v2[3][2] + 8192 + 32) >> 6, (v2[3][3] + 8192 + 32) >> 6);
	pk[3] = pack4_clip((v3[3][0] + 8192 + 32) >> 6, (v3[3][1] + 8192 + 32) >> 6, (v3[3][2] + 8192 + 32) >> 6, (v3[3][3] + 8192 + 32) >> 6);
	// fy = 1..3: vertical filter over the integer column (single stage) and over the horizontal intermediates (second stage: >> 12; its results lie within
	// -168 .. 434, so the reference's saturation to 16 bits never acts and the clip to 8 bits is all there is)
#pragma unroll
	for (int fy = 1; fy < 4; fy++) {
		const int *cf = fy == 1 ? c1 : (fy == 2 ? c2 : c3);
		int r[4][4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
			for (int k = 0; k < 8; k++) { s0 += v0[k][j] * cf[k]; s1 += v1[k][j] * cf[k]; s2 += v2[k][j] * cf[k]; s3 += v3[k][j] * cf[k]; }
			r[0][j] = (s0 + 32) >> 6;
			r[1][j] = (s1 + 2048 + (8192 << 6)) >> 12;
			r[2][j] = (s2 + 2048 + (8192 << 6)) >> 12;
			r[3][j] = (s3 + 2048 + (8192 << 6)) >> 12;
		}
#pragma unroll
		for (int fx = 0; fx < 4; fx++) pk[fy * 4 + fx] = pack4_clip(r[fx][0], r[fx][1], r[fx][2], r[fx][3]);
	}
	uint8_t *o = out + ((size_t)y * 16 * stride + x);      // row y of phase f starts at (y * 16 + f) * stride: the phases of a row lie side by side
#pragma unroll
	for (int f = 0; f < 16; f++) *(uint32_t *)(o + (size_t)f * stride) = pk[f];
}

// one chroma component: out = [rows][64][stride]
__global__ __launch_bounds__(256) void k_subpel_chroma(const int16_t *__restrict__ pic, int stride, int rows, uint8_t *__restrict__ out)
{
	__shared__ __attribute__((aligned(16))) int16_t in[TH + 3][TW + 8];          // rows y0-1 .. y0+TH+1, columns x0-4 .. x0+TW+3
	__shared__ __attribute__((aligned(16))) int16_t hs[7][TH + 3][TW];           // horizontal stage (sum - 8192) for fx = 1 .. 7
	const int tiles_x = (stride + TW - 1) / TW;
	const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
	const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
	if (y0 >= rows) return;
	const long elems = (long)stride * rows;
	const int t = (int)threadIdx.x;
	for (int i = t; i < (TH + 3) * (TW + 8); i += 256) {
		const int r = i / (TW + 8), c = i - r * (TW + 8);
		const long li = (long)(y0 - 1 + r) * stride + (x0 - 4 + c);
		in[r][c] = (li >= 0 && li < elems) ? pic[li] : (int16_t)0;
	}
	__syncthreads();
	const int cf[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
	for (int i = t; i < (TH + 3) * (TW / 4); i += 256) {
		const int r = i / (TW / 4), c = (i - r * (TW / 4)) << 2;
		const short4_t a = lds4(&in[r][c]), b = lds4(&in[r][c + 4]), d = lds4(&in[r][c + 8]);
		const int sm[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
		for (int fx = 1; fx < 8; fx++) {
			short4_t o;
#pragma unroll
			for (int j = 0; j < 4; j++) o[j] = (short)(sm[j + 3] * cf[fx][0] + sm[j + 4] * cf[fx][1] + sm[j + 5] * cf[fx][2] + sm[j + 6] * cf[fx][3] - 8192);   // samples (c + j) - 1 ...
			*(short4_t *)&hs[fx - 1][r][c] = o;
		}
	}
	__syncthreads();
	const int ty = t >> 4, tx = (t & 15) << 2;
	const int y = y0 + ty, x = x0 + tx;
	if (y >= rows || x >= stride) return;
	uint8_t *o = out + ((size_t)y * 64 * stride + x);      // row y of phase f starts at (y * 64 + f) * stride
	short4_t in4[4];
#pragma unroll
	for (int k = 0; k < 4; k++) in4[k] = lds4(&in[ty + k][tx + 4]);
	// fy = 0
	{
		uint32_t pk[8];
#pragma unroll
		for (int fx = 1; fx < 8; fx++) {
			const short4_t h = lds4(&hs[fx - 1][ty + 1][tx]);
			pk[fx] = pack4_clip((h[0] + 8192 + 32) >> 6, (h[1] + 8192 + 32) >> 6, (h[2] + 8192 + 32) >> 6, (h[3] + 8192 + 32) >> 6);
		}
		pk[0] = pack4_clip(in4[1][0], in4[1][1], in4[1][2], in4[1][3]);
#pragma unroll
		for (int f = 0; f < 8; f++) *(uint32_t *)(o + (size_t)f * stride) = pk[f];
	}
	for (int fx = 0; fx < 8; fx++) {
		// the column of intermediates (fx = 0: of integer samples) once, the seven vertical phases from it
		short4_t h[4];
#pragma unroll
		for (int k = 0; k < 4; k++) h[k] = fx ? lds4(&hs[fx - 1][ty + k][tx]) : in4[k];
#pragma unroll
		for (int fy = 1; fy < 8; fy++) {
			int r[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int sum = h[0][j] * cf[fy][0] + h[1][j] * cf[fy][1] + h[2][j] * cf[fy][2] + h[3][j] * cf[fy][3];
				r[j] = fx ? (sum + 2048 + (8192 << 6)) >> 12 : (sum + 32) >> 6;      // (second stage: within -130 .. 390, the saturation to 16 bits never acts)
			}
			*(uint32_t *)(o + (size_t)(fy * 8 + fx) * stride) = pack4_clip(r[0], r[1], r[2], r[3]);
		}
	}
}

}  // namespace

// device pointers; pic_* = start of the padded allocations (stride x rows elements), out_y = [rows][16][stride], out_u / out_v = [rows][64][stride]
int hmr_subpel_planes_on(hipStream_t stream, const int16_t *pic_y, const int16_t *pic_u, const int16_t *pic_v, int stride_y, int rows_y, int stride_c, int rows_c, uint8_t *out_y,
			 uint8_t *out_u, uint8_t *out_v)
{
	if (!pic_y || !out_y || (stride_y & 3) || (stride_c & 3) || stride_y <= 0 || rows_y <= 0) return HMR_GPU_ERR_ARG;
	const int gy = (((stride_y + TW - 1) / TW) * ((rows_y + TH - 1) / TH) + 7) / 8 * 8;      // (a multiple of eight: xcd_contiguous)
	hipLaunchKernelGGL(k_subpel_luma, dim3(gy), dim3(256), 0, stream, pic_y, stride_y, rows_y, out_y);
	if (pic_u && pic_v && out_u && out_v) {
		const int gc = (((stride_c + TW - 1) / TW) * ((rows_c + TH - 1) / TH) + 7) / 8 * 8;
		hipLaunchKernelGGL(k_subpel_chroma, dim3(gc), dim3(256), 0, stream, pic_u, stride_c, rows_c, out_u);
		hipLaunchKernelGGL(k_subpel_chroma, dim3(gc), dim3(256), 0, stream, pic_v, stride_c, rows_c, out_v);
	}
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
// one component's planes on `stream` (comp 0: luma, 16 planes; 1 / 2: a chroma component, 64 planes)
int hmr_subpel_plane_on(hipStream_t stream, int comp, const int16_t *pic, int stride, int rows, uint8_t *out)
{
	if (!pic || !out || (stride & 3) || stride <= 0 || rows <= 0) return HMR_GPU_ERR_ARG;
	const int grid = (((stride + TW - 1) / TW) * ((rows + TH - 1) / TH) + 7) / 8 * 8;
	if (comp == 0) hipLaunchKernelGGL(k_subpel_luma, dim3(grid), dim3(256), 0, stream, pic, stride, rows, out);
	else hipLaunchKernelGGL(k_subpel_chroma, dim3(grid), dim3(256), 0, stream, pic, stride, rows, out);
	HIP_TRY(hipGetLastError());
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_subpel_planes(hmr_gpu_ctx *ctx, const int16_t *pic_y, const int16_t *pic_u, const int16_t *pic_v, int stride_y, int rows_y, int stride_c, int rows_c,
				     uint8_t *out_y, uint8_t *out_u, uint8_t *out_v)
{
	if (!ctx) return HMR_GPU_ERR_ARG;
	return hmr_subpel_planes_on(ctx->stream, pic_y, pic_u, pic_v, stride_y, rows_y, stride_c, rows_c, out_y, out_u, out_v);
}
