// The phase planes of a reference picture (k_subpel.hip, include/homer_gpu.h section 13) produced CTU by CTU as a task of the CTU kernel: when a sequence's
// frames overlap (hmr_gpu_enc_encode_chain) the next frame's CTUs start as soon as the part of this frame they can reach is final, so the planes cannot wait
// for the picture to be complete.  S(r, c) covers CTU (r, c) of the FINAL picture - with the margins beside it when the CTU lies on the picture's edge - and
// needs the final samples of the CTUs around it (the filters reach three / four samples out).
//
// Same arithmetic as k_subpel_luma / k_subpel_chroma (sse_interpolate_luma / _chroma, inter_prediction.c:796,818; stage rules hmr_motion_inter.c:240-391), same plane
// layout (row y of phase f at (y * phases + f) * stride).  One difference that no vector can see: the frame kernels address the padded allocation linearly (a tap that
// runs over a row end reads the neighbouring row's memory, as the reference's pointer arithmetic does), here a tap stops at the allocation's edge.  The search keeps a
// block inside the picture (hmr_motion_estimation :1424-1427), sub-sample refinement and merge candidates move it at most 64 + 1 samples out, the taps four more: the
// outermost columns of the 80-sample margins are never read (that is what the margin's extra 16 samples are for).
#pragma once
#include "common.h"
#include "enc/enc_types.h"

namespace henc {

constexpr int SPT_W = 64, SPT_H = 16;      // a luma tile; a chroma tile is as many samples, 32 x 32
struct SubpelScratch {
	alignas(16) int16_t in[(32 + 7) * (SPT_W + 8)];       // the tile with its filter margin: rows y0 - 3 .. y0 + TH + 3 (chroma: y0 - 1 .. + 1), columns x0 - 4 .. x0 + TW + 3
	alignas(16) int16_t hs[7 * (32 + 3) * SPT_W];         // horizontal stage (sum - 8192): 3 phases x (16 + 7) rows x 64 (luma), 7 phases x (32 + 3) rows x 32 (chroma)
};

__device__ __forceinline__ uint32_t spt_pack4_clip(int r0, int r1, int r2, int r3)
{
	typedef short short2_t __attribute__((ext_vector_type(2)));
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
typedef short spt_short4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ spt_short4 spt_lds4(const int16_t *p) { return *(const spt_short4 *)p; }

// one tile of the luma planes: allocation coordinates (x0, y0), tw x th samples (tw a multiple of 4, <= 64; th <= 16), by one wavefront
__device__ void subpel_luma_tile(int tid, SubpelScratch &sc, const int16_t *__restrict__ pic, int stride, int rows, uint8_t *__restrict__ out, int x0, int y0, int tw, int th)
{
	constexpr int IW = SPT_W + 8;
	int16_t *in = sc.in, *hs = sc.hs;
	const int hrows = th + 7;
	for (int i = tid; i < hrows * IW; i += 64) {
		const int r = i / IW, c = i - r * IW;
		const int y = y0 - 3 + r, x = x0 - 4 + c;
		in[i] = (y >= 0 && y < rows && x >= 0 && x < stride) ? pic[(size_t)y * stride + x] : (int16_t)0;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int c1[8] = {-1, 4, -10, 58, 17, -5, 1, 0}, c2[8] = {-1, 4, -11, 40, 40, -11, 4, -1}, c3[8] = {0, 1, -5, 17, 58, -10, 4, -1};
	const int w4 = tw >> 2;
	for (int i = tid; i < hrows * w4; i += 64) {
		const int r = i / w4, c = (i - r * w4) << 2;
		const spt_short4 a = spt_lds4(&in[r * IW + c]), b = spt_lds4(&in[r * IW + c + 4]), d = spt_lds4(&in[r * IW + c + 8]);
		const int sm[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
		spt_short4 o1, o2, o3;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			int s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int v = sm[j + 1 + k];
				s1 += v * c1[k]; s2 += v * c2[k]; s3 += v * c3[k];
			}
			o1[j] = (short)(s1 - 8192); o2[j] = (short)(s2 - 8192); o3[j] = (short)(s3 - 8192);
		}
		*(spt_short4 *)&hs[(0 * (SPT_H + 7) + r) * SPT_W + c] = o1;
		*(spt_short4 *)&hs[(1 * (SPT_H + 7) + r) * SPT_W + c] = o2;
		*(spt_short4 *)&hs[(2 * (SPT_H + 7) + r) * SPT_W + c] = o3;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	for (int i = tid; i < th * w4; i += 64) {
		const int ty = i / w4, tx = (i - ty * w4) << 2;
		const int y = y0 + ty, x = x0 + tx;
		uint32_t pk[16];
		spt_short4 v0[8], v1[8], v2[8], v3[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			v0[k] = spt_lds4(&in[(ty + k) * IW + tx + 4]);
			v1[k] = spt_lds4(&hs[(0 * (SPT_H + 7) + ty + k) * SPT_W + tx]);
			v2[k] = spt_lds4(&hs[(1 * (SPT_H + 7) + ty + k) * SPT_W + tx]);
			v3[k] = spt_lds4(&hs[(2 * (SPT_H + 7) + ty + k) * SPT_W + tx]);
		}
		pk[0] = spt_pack4_clip(v0[3][0], v0[3][1], v0[3][2], v0[3][3]);
		pk[1] = spt_pack4_clip((v1[3][0] + 8192 + 32) >> 6, (v1[3][1] + 8192 + 32) >> 6, (v1[3][2] + 8192 + 32) >> 6, (v1[3][3] + 8192 + 32) >> 6);
		pk[2] = spt_pack4_clip((v2[3][0] + 8192 + 32) >> 6, (v2[3][1] + 8192 + 32) >> 6, (v2[3][2] + 8192 + 32) >> 6, (v2[3][3] + 8192 + 32) >> 6);
		pk[3] = spt_pack4_clip((v3[3][0] + 8192 + 32) >> 6, (v3[3][1] + 8192 + 32) >> 6, (v3[3][2] + 8192 + 32) >> 6, (v3[3][3] + 8192 + 32) >> 6);
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
			for (int fx = 0; fx < 4; fx++) pk[fy * 4 + fx] = spt_pack4_clip(r[fx][0], r[fx][1], r[fx][2], r[fx][3]);
		}
		uint8_t *o = out + ((size_t)y * 16 * stride + x);
#pragma unroll
		for (int f = 0; f < 16; f++) *(uint32_t *)(o + (size_t)f * stride) = pk[f];
	}
	__builtin_amdgcn_wave_barrier();
}

// one tile of a chroma component's planes: tw <= 32 (a multiple of 4), th <= 32
__device__ void subpel_chroma_tile(int tid, SubpelScratch &sc, const int16_t *__restrict__ pic, int stride, int rows, uint8_t *__restrict__ out, int x0, int y0, int tw, int th)
{
	constexpr int IW = SPT_W + 8, HW = 32, HR = 32 + 3;
	int16_t *in = sc.in, *hs = sc.hs;
	const int hrows = th + 3;
	for (int i = tid; i < hrows * IW; i += 64) {
		const int r = i / IW, c = i - r * IW;
		const int y = y0 - 1 + r, x = x0 - 4 + c;
		in[i] = (c < tw + 8 && y >= 0 && y < rows && x >= 0 && x < stride) ? pic[(size_t)y * stride + x] : (int16_t)0;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int cf[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
	const int w4 = tw >> 2;
	for (int i = tid; i < hrows * w4; i += 64) {
		const int r = i / w4, c = (i - r * w4) << 2;
		const spt_short4 a = spt_lds4(&in[r * IW + c]), b = spt_lds4(&in[r * IW + c + 4]), d = spt_lds4(&in[r * IW + c + 8]);
		const int sm[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
		for (int fx = 1; fx < 8; fx++) {
			spt_short4 o;
#pragma unroll
			for (int j = 0; j < 4; j++) o[j] = (short)(sm[j + 3] * cf[fx][0] + sm[j + 4] * cf[fx][1] + sm[j + 5] * cf[fx][2] + sm[j + 6] * cf[fx][3] - 8192);
			*(spt_short4 *)&hs[((fx - 1) * HR + r) * HW + c] = o;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	for (int i = tid; i < th * w4; i += 64) {
		const int ty = i / w4, tx = (i - ty * w4) << 2;
		const int y = y0 + ty, x = x0 + tx;
		uint8_t *o = out + ((size_t)y * 64 * stride + x);
		spt_short4 in4[4];
#pragma unroll
		for (int k = 0; k < 4; k++) in4[k] = spt_lds4(&in[(ty + k) * IW + tx + 4]);
		{
			uint32_t pk[8];
#pragma unroll
			for (int fx = 1; fx < 8; fx++) {
				const spt_short4 h = spt_lds4(&hs[((fx - 1) * HR + ty + 1) * HW + tx]);
				pk[fx] = spt_pack4_clip((h[0] + 8192 + 32) >> 6, (h[1] + 8192 + 32) >> 6, (h[2] + 8192 + 32) >> 6, (h[3] + 8192 + 32) >> 6);
			}
			pk[0] = spt_pack4_clip(in4[1][0], in4[1][1], in4[1][2], in4[1][3]);
#pragma unroll
			for (int f = 0; f < 8; f++) *(uint32_t *)(o + (size_t)f * stride) = pk[f];
		}
		for (int fx = 0; fx < 8; fx++) {
			spt_short4 h[4];
#pragma unroll
			for (int k = 0; k < 4; k++) h[k] = fx ? spt_lds4(&hs[((fx - 1) * HR + ty + k) * HW + tx]) : in4[k];
#pragma unroll
			for (int fy = 1; fy < 8; fy++) {
				int r[4];
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const int sum = h[0][j] * cf[fy][0] + h[1][j] * cf[fy][1] + h[2][j] * cf[fy][2] + h[3][j] * cf[fy][3];
					r[j] = fx ? (sum + 2048 + (8192 << 6)) >> 12 : (sum + 32) >> 6;
				}
				*(uint32_t *)(o + (size_t)(fy * 8 + fx) * stride) = spt_pack4_clip(r[0], r[1], r[2], r[3]);
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
}

// S(r, c): the planes of CTU (cx, cy) of the final picture `fin` (first valid sample of each padded plane) and of the margins beside it on the picture's edges
__device__ __attribute__((noinline)) void subpel_task_ctu(int tid, SubpelScratch &sc, const Seq &S, int16_t *const *fin, uint8_t *out_y, uint8_t *out_u, uint8_t *out_v, int cx, int cy)
{
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, m = comp ? S.margin_c : S.margin_y, stride = comp ? S.stride_c : S.stride_y;
		const int pw = S.width >> sh, ph = S.height >> sh, rows = ph + 2 * m, cs = 64 >> sh;
		// the region in allocation coordinates
		const int X0 = cx == 0 ? 0 : m + cx * cs, X1 = cx == S.wctu - 1 ? stride : m + hmin((cx + 1) * cs, pw);
		const int Y0 = cy == 0 ? 0 : m + cy * cs, Y1 = cy == S.hctu - 1 ? rows : m + hmin((cy + 1) * cs, ph);
		const int16_t *pic = fin[comp] - ((size_t)m * stride + m);       // allocation start
		uint8_t *out = comp == 0 ? out_y : (comp == 1 ? out_u : out_v);
		const int TW = comp ? 32 : SPT_W, TH = comp ? 32 : SPT_H;
		for (int y0 = Y0; y0 < Y1; y0 += TH)
			for (int x0 = X0; x0 < X1; x0 += TW) {
				const int tw = hmin(TW, X1 - x0), th = hmin(TH, Y1 - y0);
				if (comp == 0) subpel_luma_tile(tid, sc, pic, stride, rows, out, x0, y0, tw, th);
				else subpel_chroma_tile(tid, sc, pic, stride, rows, out, x0, y0, tw, th);
			}
	}
}

}  // namespace henc
