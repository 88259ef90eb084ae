// Test harness for the block primitives of the frame encoder (enc/enc_prims.h): the product's CTU walk (k_encode_pool) runs THESE primitives, not the table kernels of
// k_pixel / k_transform / k_intra ... that the drop-in and batched ABI expose - and until now they were only checked through whole streams.  Every entry below has the
// flat signature of the table function it restates (include/homer_gpu.h section 1; low_level_funcs_t, hmr_private.h:1063-1092) under the prefix hmr_gpu_prim_, so that the
// case generator of tests/kernel_cases.py drives them like the oracle's and the table kernels' (tests/test_gpu_prims.py).  One wavefront runs the primitive as the worker
// does (WaveGrp).  hmr_gpu_prim_bytes(1): the sample operands the worker keeps as bytes (source, prediction: src_t / pred_t) are narrowed before the call, i.e. the
// <uint8_t> instantiations run - valid for cases whose samples are 0 .. 255; (0): the 16-bit instantiations (what the one-lane checker build compiles).
// Host pointers in, synchronous; not a performance path.
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"
#include "enc/enc_prims.h"

using namespace henc;

hmr_gpu_ctx *hmr_default_ctx();

namespace {

enum { PT_SAD = 1, PT_SSD, PT_PREDICT, PT_RECONST, PT_VARIANCE, PT_INTRA_PRED, PT_FILL_REFS, PT_ADI_FILTER, PT_TRANSFORM, PT_ITRANSFORM, PT_QUANT, PT_INV_QUANT, PT_SAD_U8 };

struct PtArgs {
	int op, bytes;
	int p[12];
	long off[6];      // element offsets of the operands in the arena (int16 elements; byte operands use the same element offsets in the byte arena)
};

int g_bytes = 0;

// arena: 16-bit image of every operand; arena8: the same elements narrowed to bytes (for the operands a primitive takes as bytes)
__global__ __launch_bounds__(64) void k_primtest(PtArgs a, int16_t *arena, uint8_t *arena8, const DevTables *T, uint32_t *ret)
{
	const WaveGrp g{(int)threadIdx.x};
	const int *p = a.p;
	auto A = [&](int k) { return arena + a.off[k]; };
	auto B = [&](int k) { return arena8 + a.off[k]; };
	uint32_t r = 0;
	switch (a.op) {
	case PT_SAD: r = blk_sad(g, A(0), p[0], A(1), p[1], p[2]); break;
	case PT_SAD_U8: {      // the device's motion search: candidates against byte planes (multi_sad_u8; the source block at pitch 64)
		const uint8_t *cand[4] = {B(1), nullptr, B(1) + 1, nullptr};
		uint32_t out[4];
		multi_sad_u8<4>(g, B(0), p[2], cand, p[1], out);
		r = out[0];
		break;
	}
	case PT_SSD: r = a.bytes ? blk_ssd(g, B(0), p[0], B(1), p[1], p[2]) : blk_ssd(g, A(0), p[0], A(1), p[1], p[2]); break;
	case PT_PREDICT:
		if (a.bytes) blk_predict(g, B(0), p[0], B(1), p[1], A(2), p[2], p[3]);
		else blk_predict(g, A(0), p[0], A(1), p[1], A(2), p[2], p[3]);
		break;
	case PT_RECONST: {
		const int16_t *res = p[1] ? A(1) : nullptr;      // (residual stride 0: the reference passes a zeroed row)
		if (a.bytes) blk_reconst(g, B(0), p[0], res, p[1], A(2), p[2], p[3]);
		else blk_reconst(g, A(0), p[0], res, p[1], A(2), p[2], p[3]);
		break;
	}
	case PT_VARIANCE: r = a.bytes ? blk_modified_variance(g, B(0), p[1], p[0], p[2]) : blk_modified_variance(g, A(0), p[1], p[0], p[2]); break;
	case PT_INTRA_PRED:      // p: pred stride, n, mode, is_luma
		if (a.bytes) {
			intra_predict(g, B(0), p[0], A(1), p[1], p[2], p[3]);
			g.sync();
			for (int i = g.tid; i < p[1] * p[1]; i += 64) A(0)[(i / p[1]) * p[0] + i % p[1]] = B(0)[(i / p[1]) * p[0] + i % p[1]];
		} else intra_predict(g, A(0), p[0], A(1), p[1], p[2], p[3]);
		break;
	case PT_FILL_REFS: intra_fill_refs(g, A(0), p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], A(1)); break;
	case PT_ADI_FILTER: intra_adi_filter(g, A(0), A(1), p[0], p[1]); break;
	case PT_TRANSFORM:      // block - 0 through the residual-forming first stage (RowDiff), as the worker's TUs call it
		if (a.bytes) tr_forward(g, (const FastTables *)nullptr, T, B(0), p[0], B(3), p[0], A(1), A(2), p[1], p[2]);
		else tr_forward(g, (const FastTables *)nullptr, T, A(0), p[0], A(3), p[0], A(1), A(2), p[1], p[2]);
		break;
	case PT_ITRANSFORM: tr_inverse(g, (const FastTables *)nullptr, T, A(0), p[0], A(1), A(2), p[1], p[2]); break;
	case PT_QUANT: r = (uint32_t)quantize(g, (const FastTables *)nullptr, T, A(0), A(1), A(2), p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]); break;
	case PT_INV_QUANT: dequantize(g, (const FastTables *)nullptr, T, A(0), A(1), p[0], p[1], p[2], p[3], p[4], p[5]); break;
	default: break;
	}
	g.sync();
	if (g.tid == 0) *ret = r;
}

// one operand of a call: a host block of `elems` int16 elements starting at `host` (nullptr: scratch of that size)
struct Operand {
	const int16_t *in;
	int16_t *out;
	size_t elems;
};

uint32_t run(int op, const int *p, int np, Operand *ops, int nops)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	if (!c) { fprintf(stderr, "hmr_gpu_prim_*: no device\n"); abort(); }
	PtArgs a;
	memset(&a, 0, sizeof a);
	a.op = op;
	a.bytes = g_bytes;
	for (int i = 0; i < np; i++) a.p[i] = p[i];
	size_t total = 0;
	for (int k = 0; k < nops; k++) {
		a.off[k] = (long)total;
		total += (ops[k].elems + 31) & ~(size_t)31;      // (64-byte aligned operands)
	}
	std::vector<int16_t> h(total, 0);
	std::vector<uint8_t> h8(total, 0);
	for (int k = 0; k < nops; k++)
		if (ops[k].in) memcpy(h.data() + a.off[k], ops[k].in, ops[k].elems * 2);
	for (size_t i = 0; i < total; i++) h8[i] = (uint8_t)h[i];
	int16_t *d = nullptr;
	uint8_t *d8 = nullptr;
	uint32_t *dr = nullptr, r = 0;
	if (hipMalloc((void **)&d, total * 2 + 64) != hipSuccess || hipMalloc((void **)&d8, total + 64) != hipSuccess || hipMalloc((void **)&dr, 4) != hipSuccess) abort();
	(void)hipMemcpy(d, h.data(), total * 2, hipMemcpyHostToDevice);
	(void)hipMemcpy(d8, h8.data(), total, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_primtest, dim3(1), dim3(64), 0, c->stream, a, d, d8, (const DevTables *)c->tables, dr);
	if (hipStreamSynchronize(c->stream) != hipSuccess) { fprintf(stderr, "hmr_gpu_prim_*: %s\n", hipGetErrorString(hipGetLastError())); abort(); }
	(void)hipMemcpy(h.data(), d, total * 2, hipMemcpyDeviceToHost);
	(void)hipMemcpy(&r, dr, 4, hipMemcpyDeviceToHost);
	for (int k = 0; k < nops; k++)
		if (ops[k].out) memcpy(ops[k].out, h.data() + a.off[k], ops[k].elems * 2);
	(void)hipFree(d); (void)hipFree(d8); (void)hipFree(dr);
	return r;
}

size_t span(int stride, int rows, int cols) { return stride ? (size_t)stride * (rows - 1) + cols : (size_t)cols; }

}  // namespace

extern "C" {

void hmr_gpu_prim_bytes(int on) { g_bytes = on ? 1 : 0; }

uint32_t hmr_gpu_prim_sad(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size)
{
	if (g_bytes) {      // the device's search: the source block at the worker's pitch of 64, the candidate in a byte plane
		std::vector<int16_t> blk(64 * 64, 0);
		for (int y = 0; y < size; y++) memcpy(&blk[y * 64], src + (size_t)y * src_stride, size * 2);
		Operand ops[2] = {{blk.data(), nullptr, blk.size()}, {pred, nullptr, span((int)pred_stride, size, size) + 8}};
		const int p[3] = {64, (int)pred_stride, size};
		return run(PT_SAD_U8, p, 3, ops, 2);
	}
	Operand ops[2] = {{src, nullptr, span((int)src_stride, size, size)}, {pred, nullptr, span((int)pred_stride, size, size)}};
	const int p[3] = {(int)src_stride, (int)pred_stride, size};
	return run(PT_SAD, p, 3, ops, 2);
}
uint32_t hmr_gpu_prim_ssd16b(int16_t *src, uint32_t src_stride, int16_t *pred, uint32_t pred_stride, int size)
{
	Operand ops[2] = {{src, nullptr, span((int)src_stride, size, size)}, {pred, nullptr, span((int)pred_stride, size, size)}};
	const int p[3] = {(int)src_stride, (int)pred_stride, size};
	return run(PT_SSD, p, 3, ops, 2);
}
void hmr_gpu_prim_predict(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int size)
{
	Operand ops[3] = {{orig, nullptr, span(orig_stride, size, size)}, {pred, nullptr, span(pred_stride, size, size)}, {residual, residual, span(residual_stride, size, size)}};
	const int p[4] = {orig_stride, pred_stride, residual_stride, size};
	run(PT_PREDICT, p, 4, ops, 3);
}
void hmr_gpu_prim_reconst(int16_t *pred, int pred_stride, int16_t *residual, int residual_stride, int16_t *decoded, int decoded_stride, int size)
{
	Operand ops[3] = {{pred, nullptr, span(pred_stride, size, size)}, {residual, nullptr, span(residual_stride, size, size)}, {decoded, decoded, span(decoded_stride, size, size)}};
	const int p[4] = {pred_stride, residual_stride, decoded_stride, size};
	run(PT_RECONST, p, 4, ops, 3);
}
uint32_t hmr_gpu_prim_modified_variance(int16_t *ptr, int size, int stride, int modif)
{
	Operand ops[1] = {{ptr, nullptr, span(stride, size, size)}};
	const int p[3] = {size, stride, modif};
	return run(PT_VARIANCE, p, 3, ops, 1);
}
void hmr_gpu_prim_intra_angular(int16_t *prediction, int pred_stride, int16_t *adi, int adi_size, int cu_size, int cu_mode, int is_luma)
{
	Operand ops[2] = {{prediction, prediction, span(pred_stride, cu_size, cu_size)}, {adi, nullptr, (size_t)adi_size}};
	const int p[4] = {pred_stride, cu_size, cu_mode, is_luma};
	run(PT_INTRA_PRED, p, 4, ops, 2);
}
void hmr_gpu_prim_intra_planar(int16_t *prediction, int pred_stride, int16_t *adi, int adi_size, int cu_size)
{
	hmr_gpu_prim_intra_angular(prediction, pred_stride, adi, adi_size, cu_size, PLANAR_IDX, 1);
}
void hmr_gpu_prim_fill_reference_samples(int16_t *decoded_corner, int stride, int n, int left, int top, int bottom_left, int top_right, int bl_size, int tr_size, int16_t *adi)
{
	Operand ops[2] = {{decoded_corner, nullptr, span(stride, 2 * n + 1, 2 * n + 1)}, {adi, adi, (size_t)(4 * n + 1)}};
	const int p[8] = {stride, n, left, top, bottom_left, top_right, bl_size, tr_size};
	run(PT_FILL_REFS, p, 8, ops, 2);
}
void hmr_gpu_prim_adi_filter(int16_t *adi, int16_t *out, int adi_size, int n, int strong_enabled)
{
	Operand ops[2] = {{adi, nullptr, (size_t)adi_size}, {out, out, (size_t)adi_size}};
	const int p[2] = {n, strong_enabled};
	run(PT_ADI_FILTER, p, 2, ops, 2);
}
void hmr_gpu_prim_transform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst)
{
	if (g_bytes) {      // the worker's call: source and prediction windows of bytes (the transforms on the matrix cores) - block = source - prediction, both 0 .. 255
		const size_t el = span(block_stride, n, n);
		std::vector<int16_t> o(el, 0), q(el, 0);
		for (int y = 0; y < n; y++)
			for (int x = 0; x < n; x++) {
				const int b = block[(size_t)y * block_stride + x];
				if (b < -255 || b > 255) { fprintf(stderr, "hmr_gpu_prim_transform: a residual of %d is not the difference of two bytes\n", b); abort(); }
				o[(size_t)y * block_stride + x] = (int16_t)(b > 0 ? b : 0);
				q[(size_t)y * block_stride + x] = (int16_t)(b < 0 ? -b : 0);
			}
		Operand ops[4] = {{o.data(), nullptr, el}, {nullptr, coeff, (size_t)n * n}, {nullptr, nullptr, (size_t)n * n}, {q.data(), nullptr, el}};
		const int p[3] = {block_stride, n, is_dst};
		run(PT_TRANSFORM, p, 3, ops, 4);
		return;
	}
	Operand ops[4] = {{block, nullptr, span(block_stride, n, n)}, {nullptr, coeff, (size_t)n * n}, {nullptr, nullptr, (size_t)n * n}, {nullptr, nullptr, span(block_stride, n, n)}};
	const int p[3] = {block_stride, n, is_dst};
	run(PT_TRANSFORM, p, 3, ops, 4);
}
void hmr_gpu_prim_itransform(int16_t *block, int16_t *coeff, int block_stride, int n, int is_dst)
{
	Operand ops[3] = {{block, block, span(block_stride, n, n)}, {coeff, nullptr, (size_t)n * n}, {nullptr, nullptr, (size_t)n * n}};
	const int p[3] = {block_stride, n, is_dst};
	run(PT_ITRANSFORM, p, 3, ops, 3);
}
void hmr_gpu_prim_quant(int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra, int slice_is_intra, int sign_hiding, int *ac_sum,
			int cu_size, int per, int rem)
{
	Operand ops[3] = {{src, nullptr, (size_t)cu_size * cu_size}, {nullptr, dst, (size_t)cu_size * cu_size}, {nullptr, delta_u, (size_t)cu_size * cu_size}};
	const int p[9] = {scan_mode, depth, comp, is_intra, slice_is_intra, sign_hiding, cu_size, per, rem};
	*ac_sum = (int)run(PT_QUANT, p, 9, ops, 3);
}
void hmr_gpu_prim_inv_quant(int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem)
{
	Operand ops[2] = {{src, nullptr, (size_t)cu_size * cu_size}, {nullptr, dst, (size_t)cu_size * cu_size}};
	const int p[6] = {depth, comp, is_intra, cu_size, per, rem};
	run(PT_INV_QUANT, p, 6, ops, 2);
}

}  // extern "C"
