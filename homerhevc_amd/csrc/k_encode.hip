// Frame encoder on the device (include/homer_gpu.h section 12).
//
// k_encode_ctus is ONE persistent launch per frame that walks the picture the way the reference's WPP threads do
// (wfpp_encoder_thread, hmr_encoder_lib.c:2849-2975): workgroup r = one wavefront = the worker of CTU row r; it encodes its row
// left to right and may start CTU (r, c) once row r-1 has finished CTU c+1 (:2885-2898, two CTUs of lag).  Rows publish their
// progress with release stores at agent scope and wait with acquire loads, so a row sees the reconstruction, side-info and
// counters of the rows above it.  The decision code is enc/enc_ctu.h, instantiated for the 64-lane group.
#include <vector>

#include "common.h"
#include "enc/enc_ctu.h"
#include "enc/enc_host.h"

using namespace henc;

struct EncDev {
	const Seq *seq;
	const FrameCtx *frame;
	const DevTables *tables;
	const Geo *geo;
	CtuInfo *ctus;
	Work *work;            // one per CTU row
	int16_t *coeff;
	int *progress;         // [hctu] CTUs finished per row
	uint32_t *intra_prefix;   // [hctu][wctu + 1] running count of intra partitions along each row
	unsigned long long *prof; // [hctu][PF_COUNT] phase timers (profiling build)
};

__global__ __launch_bounds__(64) void k_encode_ctus(EncDev d)
{
	const Seq &S = *d.seq;
	const int row = blockIdx.x, W = S.wctu;
	WaveGrp g{(int)threadIdx.x};
	Enc e;
	e.seq = d.seq;
	e.f = d.frame;
	e.T = d.tables;
	e.geo = d.geo;
	e.ctus = d.ctus;
	e.ctu = nullptr;
	e.w = d.work + row;
	e.prof = d.prof ? d.prof + (size_t)row * PF_COUNT : nullptr;
	uint32_t *my_prefix = d.intra_prefix + (size_t)row * (W + 1);
	uint32_t run = 0;
	if (g.tid == 0) my_prefix[0] = 0;
	for (int c = 0; c < W; c++) {
		if (row > 0) {
			HENC_PROF_T0();
			const int need = c + 2 < W ? c + 2 : W;
			while (__hip_atomic_load(&d.progress[row - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(16);
			HENC_PROF_ADD(e, PF_WAIT);
		}
		HENC_PROF_T0();
		// running intra statistics (hmr_motion_inter.c:3769-3776) from the CTUs the wavefront order guarantees to be finished:
		// row r-k has completed at least c + 2k CTUs
		uint32_t ti = run;
		for (int k = 1; k <= row; k++) {
			const int have = c + 2 * k < W ? c + 2 * k : W;
			ti += d.intra_prefix[(size_t)(row - k) * (W + 1) + have];
		}
		e.total_intra_partitions = ti;
		e.total_partitions = (uint32_t)(row * W + c) * NPART;
		const int n = row * W + c;
		e.coeff = d.coeff + (size_t)n * 6144;
		encode_ctu(g, e, n);
		HENC_PROF_ADD(e, PF_TOTAL);
		run += d.ctus[n].intra_parts;
		if (g.tid == 0) my_prefix[c + 1] = run;
		__syncthreads();
		if (g.tid == 0) __hip_atomic_store(&d.progress[row], c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}

struct hmr_gpu_enc {
	hmr_gpu_ctx *ctx;
	HostCfg cfg;
	Seq seq;
	HostState st;
	FrameCtx f;
	EncDev d;
	Seq *d_seq;
	FrameCtx *d_frame;
	Geo *d_geo;
	int16_t *d_src[3], *d_pic[2][3];
	size_t src_elems[3], pic_elems[3];
	int cur;
	float last_ms;
	std::vector<int16_t> stage;
};

namespace {
constexpr int REC_BYTES = 32 + 3 * 256 + 2 * 256 + 9 * 256 + 256 + 256 + 2048 + 2048 + 6144 * 2 + 6144 * 2 + 2 * 5 * 256;

int16_t *plane0(hmr_gpu_enc *e, int which, int comp)
{
	const Seq &s = e->seq;
	const int st = comp ? s.stride_c : s.stride_y, m = comp ? s.margin_c : s.margin_y;
	return e->d_pic[which][comp] + (size_t)m * st + m;
}
}  // namespace

extern "C" int hmr_gpu_enc_record_bytes(void) { return REC_BYTES; }

extern "C" int hmr_gpu_enc_create(hmr_gpu_ctx *ctx, const hmr_gpu_enc_cfg *cfg, hmr_gpu_enc **out)
{
	if (!ctx || !cfg || !out) return HMR_GPU_ERR_ARG;
	static_assert(sizeof(hmr_gpu_enc_cfg) == sizeof(HostCfg), "configuration layouts must match");
	hmr_gpu_enc *e = new hmr_gpu_enc();
	e->ctx = ctx;
	memcpy(&e->cfg, cfg, sizeof(HostCfg));
	const char *why = "";
	if (!make_seq(e->cfg, e->seq, &why)) {
		hmr_set_error("hmr_gpu_enc_create: configuration outside the built rows: %s", why);
		delete e;
		return HMR_GPU_ERR_ARG;
	}
	const Seq &s = e->seq;
	HIP_TRY(hipSetDevice(ctx->device));
	std::vector<Geo> geo(NNODES);
	make_geo(geo.data());
	HIP_TRY(hipMalloc((void **)&e->d_seq, sizeof(Seq)));
	HIP_TRY(hipMemcpy(e->d_seq, &s, sizeof(Seq), hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc((void **)&e->d_frame, sizeof(FrameCtx)));
	HIP_TRY(hipMalloc((void **)&e->d_geo, sizeof(Geo) * NNODES));
	HIP_TRY(hipMemcpy(e->d_geo, geo.data(), sizeof(Geo) * NNODES, hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc((void **)&e->d.ctus, sizeof(CtuInfo) * s.nctu));
	{
		std::vector<CtuInfo> init(s.nctu);
		memset(init.data(), 0, sizeof(CtuInfo) * s.nctu);
		for (auto &c : init) memset(c.mv_ref_idx, -1, sizeof c.mv_ref_idx);
		HIP_TRY(hipMemcpy(e->d.ctus, init.data(), sizeof(CtuInfo) * s.nctu, hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMalloc((void **)&e->d.work, sizeof(Work) * s.hctu));
	HIP_TRY(hipMemset(e->d.work, 0, sizeof(Work) * s.hctu));
	HIP_TRY(hipMalloc((void **)&e->d.coeff, sizeof(int16_t) * 6144 * s.nctu));
	HIP_TRY(hipMemset(e->d.coeff, 0, sizeof(int16_t) * 6144 * s.nctu));
	HIP_TRY(hipMalloc((void **)&e->d.progress, sizeof(int) * s.hctu));
	HIP_TRY(hipMalloc((void **)&e->d.intra_prefix, sizeof(uint32_t) * s.hctu * (s.wctu + 1)));
	HIP_TRY(hipMalloc((void **)&e->d.prof, sizeof(unsigned long long) * s.hctu * PF_COUNT));
	HIP_TRY(hipMemset(e->d.prof, 0, sizeof(unsigned long long) * s.hctu * PF_COUNT));
	for (int c = 0; c < 3; c++) {
		e->src_elems[c] = (size_t)(c ? s.src_stride_c : s.src_stride_y) * (c ? s.height / 2 : s.height);
		e->pic_elems[c] = (size_t)(c ? s.stride_c : s.stride_y) * ((c ? s.height / 2 : s.height) + 2 * (c ? s.margin_c : s.margin_y));
		HIP_TRY(hipMalloc((void **)&e->d_src[c], e->src_elems[c] * 2));
		for (int k = 0; k < 2; k++) {
			HIP_TRY(hipMalloc((void **)&e->d_pic[k][c], e->pic_elems[c] * 2));
			HIP_TRY(hipMemset(e->d_pic[k][c], 0, e->pic_elems[c] * 2));
		}
	}
	e->d.seq = e->d_seq;
	e->d.frame = e->d_frame;
	e->d.tables = ctx->tables;
	e->d.geo = e->d_geo;
	e->cur = 0;
	e->last_ms = 0;
	*out = e;
	return HMR_GPU_OK;
}

extern "C" void hmr_gpu_enc_destroy(hmr_gpu_enc *e)
{
	if (!e) return;
	(void)hipSetDevice(e->ctx->device);
	(void)hipStreamSynchronize(e->ctx->stream);
	(void)hipFree(e->d_seq); (void)hipFree(e->d_frame); (void)hipFree(e->d_geo);
	(void)hipFree(e->d.ctus); (void)hipFree(e->d.work); (void)hipFree(e->d.coeff); (void)hipFree(e->d.progress); (void)hipFree(e->d.intra_prefix);
	for (int c = 0; c < 3; c++) {
		(void)hipFree(e->d_src[c]);
		(void)hipFree(e->d_pic[0][c]);
		(void)hipFree(e->d_pic[1][c]);
	}
	delete e;
}

extern "C" float hmr_gpu_enc_last_ctu_ms(hmr_gpu_enc *e) { return e ? e->last_ms : 0.f; }

// profiling build (-DHENC_PROFILE): per-row phase timers in s_memtime ticks (100 MHz), [hctu][12]; all zero otherwise
extern "C" int hmr_gpu_enc_profile(hmr_gpu_enc *e, unsigned long long *out, int reset)
{
	if (!e || !out) return HMR_GPU_ERR_ARG;
	HIP_TRY(hipMemcpy(out, e->d.prof, sizeof(unsigned long long) * e->seq.hctu * PF_COUNT, hipMemcpyDeviceToHost));
	if (reset) HIP_TRY(hipMemset(e->d.prof, 0, sizeof(unsigned long long) * e->seq.hctu * PF_COUNT));
	return HMR_GPU_OK;
}

// host 8-bit plane -> device int16 plane (sse_copy_8_16 at frame entry, hmr_encoder_lib.c:295-305); pad > 0 also replicates the borders
static int upload_plane(hmr_gpu_enc *e, const uint8_t *src, int w, int h, int16_t *dst_base, size_t elems, int stride, int margin)
{
	e->stage.assign(elems, 0);
	int16_t *p = e->stage.data() + (size_t)margin * stride + margin;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) p[(size_t)y * stride + x] = src[(size_t)y * w + x];
	if (margin) {
		for (int y = 0; y < h; y++)
			for (int x = 1; x <= margin; x++) {
				p[(size_t)y * stride - x] = p[(size_t)y * stride];
				p[(size_t)y * stride + w - 1 + x] = p[(size_t)y * stride + w - 1];
			}
		for (int y = 1; y <= margin; y++) {
			memcpy(p - (size_t)y * stride - margin, p - margin, sizeof(int16_t) * (w + 2 * margin));
			memcpy(p + (size_t)(h - 1 + y) * stride - margin, p + (size_t)(h - 1) * stride - margin, sizeof(int16_t) * (w + 2 * margin));
		}
	}
	HIP_TRY(hipMemcpyAsync(dst_base, e->stage.data(), elems * 2, hipMemcpyHostToDevice, e->ctx->stream));
	HIP_TRY(hipStreamSynchronize(e->ctx->stream));
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_enc_frame_ctus(hmr_gpu_enc *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
				      const uint8_t *ref_v, double avg_dist, uint8_t *records)
{
	if (!e || !y || !u || !v) return HMR_GPU_ERR_ARG;
	const Seq &s = e->seq;
	hipStream_t st = e->ctx->stream;
	HIP_TRY(hipSetDevice(e->ctx->device));
	const uint8_t *in[3] = {y, u, v}, *rin[3] = {ref_y, ref_u, ref_v};
	e->cur ^= 1;
	begin_frame(s, e->st, image_type, e->f);
	if (avg_dist >= 0) e->f.avg_dist = avg_dist;
	for (int c = 0; c < 3; c++) {
		const int w = c ? s.width / 2 : s.width, h = c ? s.height / 2 : s.height;
		int rc = upload_plane(e, in[c], w, h, e->d_src[c], e->src_elems[c], c ? s.src_stride_c : s.src_stride_y, 0);
		if (rc) return rc;
		if (rin[c]) {
			rc = upload_plane(e, rin[c], w, h, e->d_pic[e->cur ^ 1][c], e->pic_elems[c], c ? s.stride_c : s.stride_y, c ? s.margin_c : s.margin_y);
			if (rc) return rc;
		}
		e->f.src[c] = e->d_src[c];
		e->f.ref[c] = plane0(e, e->cur ^ 1, c);
		e->f.rec[c] = plane0(e, e->cur, c);
	}
	HIP_TRY(hipMemcpyAsync(e->d_frame, &e->f, sizeof(FrameCtx), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemsetAsync(e->d.progress, 0, sizeof(int) * s.hctu, st));
	HIP_TRY(hipEventRecord(e->ctx->ev0, st));
	hipLaunchKernelGGL(k_encode_ctus, dim3(s.hctu), dim3(64), 0, st, e->d);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e->ctx->ev1, st));
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipEventElapsedTime(&e->last_ms, e->ctx->ev0, e->ctx->ev1));
	// frame statistics (encoder_engine_thread :3217-3238)
	std::vector<CtuInfo> ctus(s.nctu);
	HIP_TRY(hipMemcpy(ctus.data(), e->d.ctus, sizeof(CtuInfo) * s.nctu, hipMemcpyDeviceToHost));
	uint32_t acc = 0;
	for (int n = 0; n < s.nctu; n++) acc += ctus[n].distortion;
	end_frame(s, e->st, e->f, acc);
	if (records) {
		std::vector<int16_t> coeff((size_t)s.nctu * 6144), rec[3];
		HIP_TRY(hipMemcpy(coeff.data(), e->d.coeff, coeff.size() * 2, hipMemcpyDeviceToHost));
		for (int c = 0; c < 3; c++) {
			rec[c].resize(e->pic_elems[c]);
			HIP_TRY(hipMemcpy(rec[c].data(), e->d_pic[e->cur][c], e->pic_elems[c] * 2, hipMemcpyDeviceToHost));
		}
		memset(records, 0, (size_t)REC_BYTES * s.nctu);
		for (int n = 0; n < s.nctu; n++) {
			uint8_t *o = records + (size_t)n * REC_BYTES;
			const CtuInfo &ci = ctus[n];
			int32_t hdr[8] = {0x43545544, e->f.num_encoded_frames, n, e->f.slice_type, (int32_t)ci.nodes[0].cost, (int32_t)ci.nodes[0].distortion, (int32_t)ci.nodes[0].sum,
					  e->f.is_scene_change};
			memcpy(o, hdr, 32); o += 32;
			for (int k = 0; k < 3; k++) { memcpy(o, ci.cbf[k], 256); o += 256; }
			memcpy(o, ci.intra_mode[0], 256); o += 256;
			memcpy(o, ci.intra_mode[1], 256); o += 256;
			const uint8_t *arrs[9] = {ci.inter_mode, ci.tr_idx, ci.pred_depth, ci.part_size_type, ci.pred_mode, ci.skipped, ci.merge, ci.merge_idx, ci.qp};
			for (int k = 0; k < 9; k++) { memcpy(o, arrs[k], 256); o += 256; }
			memcpy(o, ci.mv_ref_idx, 256); o += 256;
			memcpy(o, ci.mv_diff_ref_idx, 256); o += 256;
			memcpy(o, ci.mv_ref, 2048); o += 2048;
			memcpy(o, ci.mv_diff, 2048); o += 2048;
			memcpy(o, coeff.data() + (size_t)n * 6144, 12288); o += 12288;
			// reconstruction before the loop filters: the part of the CTU inside the picture (the rest stays zero)
			for (int c = 0; c < 3; c++) {
				const int nn = c ? 32 : 64, px = (ci.x >> (c ? 1 : 0)), py = (ci.y >> (c ? 1 : 0));
				const int pw = c ? s.width / 2 : s.width, ph = c ? s.height / 2 : s.height, rs = c ? s.stride_c : s.stride_y, m = c ? s.margin_c : s.margin_y;
				const int16_t *p = rec[c].data() + (size_t)m * rs + m;
				for (int yy = 0; yy < nn; yy++) {
					if (py + yy < ph) {
						const int ww = px + nn <= pw ? nn : pw - px;
						memcpy(o, p + (size_t)(py + yy) * rs + px, ww * 2);
					}
					o += nn * 2;
				}
			}
		}
	}
	return e->f.slice_type;
}
