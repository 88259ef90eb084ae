// Drop-in entries: the low_level_funcs_t signatures (hmr_private.h:1066-1091) on HOST pointers.
// Each call packs its operands into the context's pinned staging buffer, does one H2D copy, launches the
// same batched kernel the performance path uses with a one-job batch, copies the result back and waits.
// This is the compatibility surface (a maintainer can store these pointers in hvenc_enc_t.funcs, see
// INTEGRATION.md); it is launch- and PCIe-bound by construction and is not what bench.py measures.
#include <mutex>

#include "common.h"

hmr_gpu_ctx *hmr_default_ctx();

// The reference calls its table from up to num_enc_engines x wfpp_num_threads host threads at once (hmr_private.h:1232-1234).  The drop-in
// entries share one default context and its staging buffer, so a call holds this lock from staging to the final copy back.
static std::recursive_mutex g_dropin_mutex;

namespace {

struct Stager {
	std::unique_lock<std::recursive_mutex> lock{g_dropin_mutex};
	hmr_gpu_ctx *c;
	size_t in_end = 0, out_begin = 0, out_end = 0;
	explicit Stager(hmr_gpu_ctx *ctx) : c(ctx)
	{
		// the reference's WPP / engine threads have device 0 current; the context may live on another GPU (one engine per GPU)
		(void)hipSetDevice(c->device);
		if (hmr_ctx_need_stage(c) != HMR_GPU_OK) {
			fprintf(stderr, "homer_gpu: no memory for the drop-in staging buffers: %s\n", hmr_gpu_last_error());
			abort();      // (the table's entries have no error channel)
		}
		in_end = align(sizeof(hmr_gpu_job));
	}
	static size_t align(size_t v) { return (v + 63) & ~(size_t)63; }
	hmr_gpu_job *job() { return (hmr_gpu_job *)c->h_stage; }
	void check(size_t end)
	{
		if (end > c->stage_bytes) {
			fprintf(stderr, "homer_gpu: drop-in operand exceeds the %zu-byte staging buffer\n", c->stage_bytes);
			abort();
		}
	}
	// copy an h x w region (elements of `es` bytes, host stride in elements) densely; returns the byte offset
	size_t put2d(const void *host, size_t stride, int h, int w, size_t es)
	{
		size_t off = in_end;
		check(off + (size_t)h * w * es);
		for (int y = 0; y < h; y++) memcpy(c->h_stage + off + (size_t)y * w * es, (const uint8_t *)host + (size_t)y * stride * es, (size_t)w * es);
		in_end = align(off + (size_t)h * w * es);
		return off;
	}
	size_t zeros(size_t bytes)
	{
		size_t off = in_end;
		check(off + bytes);
		memset(c->h_stage + off, 0, bytes);
		in_end = align(off + bytes);
		return off;
	}
	void begin_outputs() { out_begin = out_end = in_end; }
	size_t out(size_t bytes)
	{
		size_t off = out_end;
		check(off + bytes);
		out_end = align(off + bytes);
		return off;
	}
	hmr_gpu_job *djob() { return (hmr_gpu_job *)c->d_stage; }
	template <typename T> T *dev(size_t off = 0) { return (T *)(c->d_stage + off); }
	template <typename T> T *host(size_t off) { return (T *)(c->h_stage + off); }
	void upload()
	{
		if (hipMemcpyAsync(c->d_stage, c->h_stage, in_end, hipMemcpyHostToDevice, c->stream) != hipSuccess) die("H2D");
	}
	// in/out operands (in-place kernels): allocate with out(), fill the host side, then upload_all()
	void upload_all()
	{
		const size_t end = out_end > in_end ? out_end : in_end;
		if (hipMemcpyAsync(c->d_stage, c->h_stage, end, hipMemcpyHostToDevice, c->stream) != hipSuccess) die("H2D");
	}
	void finish()
	{
		if (out_end > out_begin &&
		    hipMemcpyAsync(c->h_stage + out_begin, c->d_stage + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
			die("D2H");
		if (hipStreamSynchronize(c->stream) != hipSuccess) die("sync");
	}
	// scatter a dense h x w result back to the caller's strided buffer
	void get2d(size_t off, void *host_dst, size_t stride, int h, int w, size_t es)
	{
		for (int y = 0; y < h; y++) memcpy((uint8_t *)host_dst + (size_t)y * stride * es, c->h_stage + off + (size_t)y * w * es, (size_t)w * es);
	}
	[[noreturn]] static void die(const char *what)
	{
		fprintf(stderr, "homer_gpu: %s failed in a drop-in entry: %s\n", what, hipGetErrorString(hipGetLastError()));
		abort();
	}
};

void must(int rc, const char *what)
{
	if (rc != HMR_GPU_OK) {
		fprintf(stderr, "homer_gpu: %s failed: %s\n", what, hmr_gpu_last_error());
		abort();
	}
}

int norm_size(int size) { return (size == 4 || size == 8 || size == 16 || size == 32) ? size : 64; }

uint32_t sad_like(bool ssd, int16_t *src, uint32_t ss, int16_t *pred, uint32_t ps, int size)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	const int n = norm_size(size);
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(src, ss, n, n, 2) / 2);
	jb.a_stride = n;
	jb.b_off = (uint32_t)(st.put2d(pred, ps, ps ? n : 1, n, 2) / 2);
	jb.b_stride = ps ? n : 0;
	*st.job() = jb;
	st.begin_outputs();
	size_t o = st.out(4);
	st.upload();
	must(ssd ? hmr_gpu_ssd16b_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<uint32_t>(o))
		 : hmr_gpu_sad_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<uint32_t>(o)),
	     "sad/ssd");
	st.finish();
	return *st.host<uint32_t>(o);
}

}  // namespace

extern "C" {

uint32_t hmr_gpu_sad(int16_t *src, uint32_t ss, int16_t *pred, uint32_t ps, int size) { return sad_like(false, src, ss, pred, ps, size); }
uint32_t hmr_gpu_ssd16b(int16_t *src, uint32_t ss, int16_t *pred, uint32_t ps, int size) { return sad_like(true, src, ss, pred, ps, size); }

void hmr_gpu_predict(int16_t *orig, int os, int16_t *pred, int ps, int16_t *res, int rs, int n)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(orig, os, n, n, 2) / 2); jb.a_stride = n;
	jb.b_off = (uint32_t)(st.put2d(pred, ps, n, n, 2) / 2); jb.b_stride = n;
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = n;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_predict_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>()), "predict");
	st.finish();
	st.get2d(o, res, rs, n, n, 2);
}

void hmr_gpu_reconst(int16_t *pred, int ps, int16_t *res, int rs, int16_t *dec, int ds, int n)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(pred, ps, n, n, 2) / 2); jb.a_stride = n;
	jb.b_off = (uint32_t)(st.put2d(res, rs, rs ? n : 1, n, 2) / 2); jb.b_stride = rs ? n : 0;
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = n;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_reconst_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>()), "reconst");
	st.finish();
	st.get2d(o, dec, ds, n, n, 2);
}

static void copy_any(int kind, void *src, uint32_t ss, void *dst, uint32_t ds, int h, int w)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	const size_t es = kind == 1 ? 1 : 2, ed = kind == 2 ? 1 : 2;
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(src, ss, h, w, es) / es); jb.a_stride = w;
	jb.w = (uint16_t)w; jb.h = (uint16_t)h;
	st.begin_outputs();
	size_t o = st.out((size_t)h * w * ed);
	jb.c_off = (uint32_t)(o / ed); jb.c_stride = w;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_copy_batch(c, st.djob(), 1, kind, st.dev<void>(), st.dev<void>()), "copy");
	st.finish();
	st.get2d(o, dst, ds, h, w, ed);
}
void hmr_gpu_copy_16_16(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { copy_any(0, s, ss, d, ds, h, w); }
void hmr_gpu_copy_8_16(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { copy_any(1, s, ss, d, ds, h, w); }
void hmr_gpu_copy_16_8(void *s, uint32_t ss, void *d, uint32_t ds, int h, int w) { copy_any(2, s, ss, d, ds, h, w); }

uint32_t hmr_gpu_modified_variance(int16_t *p, int size, int stride, int modif)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	// the SSE code reads whole 16-byte vectors: size x max(size,8) samples cover every byte it uses
	const int wcopy = size < 8 ? 8 : size;
	jb.a_off = (uint32_t)(st.put2d(p, stride, size, wcopy, 2) / 2); jb.a_stride = wcopy;
	jb.p0 = (uint32_t)modif;
	*st.job() = jb;
	st.begin_outputs();
	size_t o = st.out(4);
	st.upload();
	must(hmr_gpu_modified_variance_batch(c, st.djob(), 1, size, st.dev<int16_t>(), st.dev<uint32_t>(o)), "modified_variance");
	st.finish();
	return *st.host<uint32_t>(o);
}

static void intra_pred(int16_t *pred, int ps, int16_t *adi, int adi_size, int n, int mode, int luma)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	(void)adi_size;
	jb.a_off = (uint32_t)(st.put2d(adi, 0, 1, 4 * n + 1, 2) / 2);
	jb.p0 = (uint32_t)mode; jb.p1 = (uint32_t)luma;
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = n;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_intra_pred_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "intra_pred");
	st.finish();
	st.get2d(o, pred, ps, n, n, 2);
}
void hmr_gpu_intra_planar(int16_t *pred, int ps, int16_t *adi, int adi_size, int n) { intra_pred(pred, ps, adi, adi_size, n, 0, 1); }
void hmr_gpu_intra_angular(int16_t *pred, int ps, int16_t *adi, int adi_size, int n, int mode, int luma) { intra_pred(pred, ps, adi, adi_size, n, mode, luma); }

void hmr_gpu_fill_reference_samples(int16_t *corner, int stride, int n, int left, int top, int bl, int tr, int bl_size, int tr_size, int16_t *adi_out)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	// stage only the L-shaped neighbourhood the kernel reads: row 0 and column 0 of a (2n+1)^2 tile
	const int S = 2 * n + 1;
	size_t off = st.zeros((size_t)S * S * 2);
	int16_t *tile = st.host<int16_t>(off);
	const int rows = left ? n + (bl ? bl_size : 0) : 0, cols = top ? n + (tr ? tr_size : 0) : 0;
	if (left || top) tile[0] = corner[0];
	for (int y = 1; y <= rows; y++) tile[y * S] = corner[(size_t)y * stride];
	for (int x = 1; x <= cols; x++) tile[x] = corner[x];
	jb.a_off = (uint32_t)(off / 2); jb.a_stride = S;
	jb.p0 = (left ? 1u : 0u) | (top ? 2u : 0u) | (bl ? 4u : 0u) | (tr ? 8u : 0u);
	jb.p1 = (uint32_t)bl_size | ((uint32_t)tr_size << 16);
	st.begin_outputs();
	size_t o = st.out((size_t)(4 * n + 1) * 2);
	jb.c_off = (uint32_t)(o / 2);
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_intra_refs_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "intra_refs");
	st.finish();
	memcpy(adi_out, st.host<int16_t>(o), (size_t)(4 * n + 1) * 2);
}

// adi_filter alone: rebuild the adi array from a synthetic neighbourhood is not possible, so this entry runs the
// filter stage of the same kernel on an already-built array placed as the top row/left column of a tile.
void hmr_gpu_adi_filter(int16_t *adi, int16_t *out, int adi_size, int n, int strong)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	(void)adi_size;
	const int S = 2 * n + 1;
	size_t off = st.zeros((size_t)S * S * 2);
	int16_t *tile = st.host<int16_t>(off);
	// inverse of the gather: adi[2n] corner, adi[2n-r] = row r (r = 1..2n), adi[2n+x] = column x
	tile[0] = adi[2 * n];
	for (int r = 1; r <= 2 * n; r++) tile[r * S] = adi[2 * n - r];
	for (int x = 1; x <= 2 * n; x++) tile[x] = adi[2 * n + x];
	jb.a_off = (uint32_t)(off / 2); jb.a_stride = S;
	jb.p0 = 1u | 2u | 4u | 8u | 16u | (strong ? 32u : 0u);
	jb.p1 = (uint32_t)n | ((uint32_t)n << 16);
	st.begin_outputs();
	size_t o = st.out((size_t)(4 * n + 1) * 2);
	size_t f = st.out((size_t)(4 * n + 1) * 2);
	jb.c_off = (uint32_t)(o / 2);
	jb.b_off = (uint32_t)(f / 2);
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_intra_refs_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "adi_filter");
	st.finish();
	memcpy(out, st.host<int16_t>(f), (size_t)(4 * n + 1) * 2);
}

static void interp(int is_luma, int16_t *src, int ss, int16_t *dst, int ds, int frac, int w, int h, int vert, int first, int last)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	if (!is_luma && frac == 0 && w < 4) return;
	Stager st(c);
	hmr_gpu_job jb = {};
	const int taps = is_luma ? 8 : 4, before = frac ? taps / 2 - 1 : 0, after = frac ? taps / 2 : 0;
	const int mx0 = vert ? 0 : before, mx1 = vert ? 0 : after, my0 = vert ? before : 0, my1 = vert ? after : 0;
	const int tw = w + mx0 + mx1, th = h + my0 + my1;
	size_t off = st.put2d(src - (ptrdiff_t)my0 * ss - mx0, ss, th, tw, 2);
	jb.a_off = (uint32_t)(off / 2 + (size_t)my0 * tw + mx0); jb.a_stride = tw;
	jb.w = (uint16_t)w; jb.h = (uint16_t)h;
	jb.p0 = (uint32_t)frac; jb.p1 = (vert ? 1u : 0u) | (first ? 2u : 0u) | (last ? 4u : 0u);
	st.begin_outputs();
	size_t o = st.out((size_t)w * h * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = w;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_interpolate_batch(c, st.djob(), 1, is_luma, st.dev<int16_t>(), st.dev<int16_t>()), "interpolate");
	st.finish();
	st.get2d(o, dst, ds, h, w, 2);
}
void hmr_gpu_interpolate_luma(int16_t *s, int ss, int16_t *d, int ds, int frac, int w, int h, int v, int f, int l) { interp(1, s, ss, d, ds, frac, w, h, v, f, l); }
void hmr_gpu_interpolate_chroma(int16_t *s, int ss, int16_t *d, int ds, int frac, int w, int h, int v, int f, int l) { interp(0, s, ss, d, ds, frac, w, h, v, f, l); }

void hmr_gpu_weighted_average(int16_t *a, int as, int16_t *b, int bs, int16_t *d, int ds, int h, int w)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(a, as, h, w, 2) / 2); jb.a_stride = w;
	jb.b_off = (uint32_t)(st.put2d(b, bs, h, w, 2) / 2); jb.b_stride = w;
	jb.w = (uint16_t)w; jb.h = (uint16_t)h;
	st.begin_outputs();
	size_t o = st.out((size_t)w * h * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = w;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_weighted_average_batch(c, st.djob(), 1, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>()), "weighted_average");
	st.finish();
	st.get2d(o, d, ds, h, w, 2);
}

void hmr_gpu_transform(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	if (n != 4 && n != 8 && n != 16 && n != 32) return;
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(block, stride, n, n, 2) / 2); jb.a_stride = n;
	jb.p0 = (uint32_t)is_dst;
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2);
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_transform_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "transform");
	st.finish();
	memcpy(coeff, st.host<int16_t>(o), (size_t)n * n * 2);
}

void hmr_gpu_itransform(int16_t *block, int16_t *coeff, int stride, int n, int is_dst)
{
	if (n != 4 && n != 8 && n != 16 && n != 32) return;
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(coeff, 0, 1, n * n, 2) / 2);
	jb.p0 = (uint32_t)is_dst;
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = n;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_itransform_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "itransform");
	st.finish();
	st.get2d(o, block, stride, n, n, 2);
}

void hmr_gpu_quant(int16_t *src, int16_t *dst, int16_t *delta_u, int scan_mode, int depth, int comp, int is_intra, int slice_is_intra, int sign_hiding,
		   int *ac_sum, int cu_size, int per, int rem)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	const int n = 1 << (6 - (depth + (comp != 0)));   // inv_depth, hmr_sse42_functions_quant.c:39
	(void)cu_size;
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(src, 0, 1, n * n, 2) / 2);
	jb.p0 = (uint32_t)(scan_mode & 3) | ((uint32_t)comp << 2) | ((uint32_t)(is_intra != 0) << 4) | ((uint32_t)(slice_is_intra != 0) << 5) |
		((uint32_t)(sign_hiding != 0) << 6);
	jb.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2), du = st.out((size_t)n * n * 2), ac = st.out(4);
	jb.c_off = (uint32_t)(o / 2);
	jb.b_off = (uint32_t)(du / 2);
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_quant_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int32_t>(ac)), "quant");
	st.finish();
	memcpy(dst, st.host<int16_t>(o), (size_t)n * n * 2);
	if (delta_u) memcpy(delta_u, st.host<int16_t>(du), (size_t)n * n * 2);
	*ac_sum = *st.host<int32_t>(ac);
}

void hmr_gpu_inv_quant(int16_t *src, int16_t *dst, int depth, int comp, int is_intra, int cu_size, int per, int rem)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	const int n = 1 << (6 - (depth + (comp != 0)));
	(void)cu_size;
	Stager st(c);
	hmr_gpu_job jb = {};
	jb.a_off = (uint32_t)(st.put2d(src, 0, 1, n * n, 2) / 2);
	jb.p0 = ((uint32_t)comp << 2) | ((uint32_t)(is_intra != 0) << 4);
	jb.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	st.begin_outputs();
	size_t o = st.out((size_t)n * n * 2);
	jb.c_off = (uint32_t)(o / 2);
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_inv_quant_batch(c, st.djob(), 1, n, st.dev<int16_t>(), st.dev<int16_t>()), "inv_quant");
	st.finish();
	memcpy(dst, st.host<int16_t>(o), (size_t)n * n * 2);
}

uint32_t hmr_gpu_motion_estimation(int16_t *orig, int orig_stride, int16_t *ref, int ref_stride, int gx, int gy, int init_x, int init_y, int size, int range_x,
				   int range_y, int frame_w, int frame_h, const int32_t *amvp, int n_amvp, const int32_t *search, int n_search, double corr,
				   int action, int32_t *out_mv4)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	// the search window the kernel can touch: integer candidates inside [low, high], sub-pel taps 4 further
	const int xlow = (gx - range_x) < 0 ? -gx : -range_x, xhigh = (gx + range_x) > (frame_w - size) ? frame_w - gx - size : range_x;
	const int ylow = (gy - range_y) < 0 ? -gy : -range_y, yhigh = (gy + range_y) > (frame_h - size) ? frame_h - gy - size : range_y;
	// without the integer stage the caller's vector is used unclamped (hmr_motion_inter.c:1668): keep it inside the staged window
	const int xl = init_x < xlow ? init_x : xlow, xh = init_x > xhigh ? init_x : xhigh;
	const int yl = init_y < ylow ? init_y : ylow, yh = init_y > yhigh ? init_y : yhigh;
	const int x0 = xl - 5, y0 = yl - 5, ww = xh - xl + size + 10, wh = yh - yl + size + 10;
	hmr_gpu_me_job jb = {};
	jb.corr = corr;
	jb.orig_off = (uint32_t)(st.put2d(orig, orig_stride, size, size, 2) / 2); jb.orig_stride = size;
	const size_t woff = st.put2d(ref + (ptrdiff_t)y0 * ref_stride + x0, ref_stride, wh, ww, 2);
	jb.ref_off = (uint32_t)(woff / 2 + (size_t)(-y0) * ww + (-x0)); jb.ref_stride = ww;
	jb.gx = (int16_t)gx; jb.gy = (int16_t)gy; jb.init_x = (int16_t)init_x; jb.init_y = (int16_t)init_y;
	jb.n_amvp = (int16_t)n_amvp; jb.n_search = (int16_t)n_search;
	for (int i = 0; i < n_amvp && i < 2; i++) { jb.amvp[i][0] = (int16_t)amvp[2 * i]; jb.amvp[i][1] = (int16_t)amvp[2 * i + 1]; }
	for (int i = 0; i < n_search && i < 5; i++) { jb.search[i][0] = (int16_t)search[2 * i]; jb.search[i][1] = (int16_t)search[2 * i + 1]; }
	jb.action = (uint32_t)action;
	const size_t joff = st.zeros(sizeof jb);
	memcpy(st.host<uint8_t>(joff), &jb, sizeof jb);
	st.begin_outputs();
	const size_t o = st.out(sizeof(hmr_gpu_me_result));
	st.upload();
	must(hmr_gpu_motion_estimation_batch(c, st.dev<hmr_gpu_me_job>(joff), 1, size, st.dev<int16_t>(), st.dev<int16_t>(), range_x, range_y, frame_w, frame_h,
					     st.dev<hmr_gpu_me_result>(o)),
	     "motion_estimation");
	st.finish();
	const hmr_gpu_me_result *r = st.host<hmr_gpu_me_result>(o);
	out_mv4[0] = r->mvx; out_mv4[1] = r->mvy; out_mv4[2] = r->subx; out_mv4[3] = r->suby;
	return r->sad;
}

static void mc_any(int is_luma, int16_t *ref, int ref_stride, int16_t *pred, int pred_stride, int w, int h, int mvx, int mvy, int is_bi)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_job jb = {};
	const int fs = is_luma ? 2 : 3, m = is_luma ? 4 : 2;   // taps reach m-1 before / m after the block
	const int ix = mvx >> fs, iy = mvy >> fs;
	const int tw = w + 2 * m, th = h + 2 * m;
	const size_t off = st.put2d(ref + (ptrdiff_t)(iy - m) * ref_stride + ix - m, ref_stride, th, tw, 2);
	// the staged tile already starts at the integer vector: pass only the fractional part on
	jb.a_off = (uint32_t)(off / 2 + (size_t)m * tw + m); jb.a_stride = tw;
	jb.w = (uint16_t)w; jb.h = (uint16_t)h; jb.p0 = (uint32_t)(mvx & (is_luma ? 3 : 7)); jb.p1 = (uint32_t)(mvy & (is_luma ? 3 : 7));
	st.begin_outputs();
	const size_t o = st.out((size_t)w * h * 2);
	jb.c_off = (uint32_t)(o / 2); jb.c_stride = w;
	*st.job() = jb;
	st.upload();
	must(hmr_gpu_mc_batch(c, st.djob(), 1, is_luma, is_bi, st.dev<int16_t>(), st.dev<int16_t>()), "mc");
	st.finish();
	st.get2d(o, pred, pred_stride, h, w, 2);
}
void hmr_gpu_mc_luma(int16_t *ref, int rs, int16_t *pred, int ps, int w, int h, int mvx, int mvy, int bi) { mc_any(1, ref, rs, pred, ps, w, h, mvx, mvy, bi); }
void hmr_gpu_mc_chroma(int16_t *ref, int rs, int16_t *pred, int ps, int size, int mvx, int mvy, int bi) { mc_any(0, ref, rs, pred, ps, size, size, mvx, mvy, bi); }

void hmr_gpu_get_sao_stats(const int16_t *const orig[3], const int orig_stride[3], const int16_t *const recon[3], const int recon_stride[3], int pict_width,
			   int pict_height, int ctu_x, int ctu_y, int64_t *stats)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_frame fo = {}, fr = {};
	fo.width = fr.width = pict_width; fo.height = fr.height = pict_height;
	int16_t **po[3] = {&fo.y, &fo.u, &fo.v}, **pr[3] = {&fr.y, &fr.u, &fr.v};
	int tw_l = 0, tw_c = 0;
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, pw = pict_width >> sh, ph = pict_height >> sh, x = ctu_x >> sh, y = ctu_y >> sh, n = 64 >> sh;
		// the CTU plus a one-sample ring, clipped to the picture
		const int x0 = x > 0 ? x - 1 : 0, y0 = y > 0 ? y - 1 : 0, x1 = x + n + 1 < pw ? x + n + 1 : pw, y1 = y + n + 1 < ph ? y + n + 1 : ph;
		const int w = comp ? 34 : 66;   // common tile pitch per plane type, so that one stride serves U and V
		if (comp == 0) tw_l = w; else tw_c = w;
		const size_t or_off = st.zeros((size_t)w * w * 2), re_off = st.zeros((size_t)w * w * 2);
		for (int yy = y0; yy < y1; yy++) {
			memcpy(st.host<int16_t>(or_off) + (size_t)(yy - y0) * w, orig[comp] + (size_t)yy * orig_stride[comp] + x0, (size_t)(x1 - x0) * 2);
			memcpy(st.host<int16_t>(re_off) + (size_t)(yy - y0) * w, recon[comp] + (size_t)yy * recon_stride[comp] + x0, (size_t)(x1 - x0) * 2);
		}
		// virtual picture origin: the kernel indexes with picture coordinates, only the tile is ever dereferenced
		*po[comp] = st.dev<int16_t>(or_off) - ((ptrdiff_t)y0 * w + x0);
		*pr[comp] = st.dev<int16_t>(re_off) - ((ptrdiff_t)y0 * w + x0);
	}
	fo.stride_y = fr.stride_y = tw_l; fo.stride_c = fr.stride_c = tw_c;
	st.begin_outputs();
	const size_t o = st.out(3 * 5 * 2 * 32 * 4);
	st.upload();
	const int ctus_x = (pict_width + 63) / 64;
	must(hmr_gpu_sao_stats_ctu(c, &fo, &fr, (ctu_y / 64) * ctus_x + ctu_x / 64, st.dev<int32_t>(o)), "get_sao_stats");
	st.finish();
	const int32_t *r = st.host<int32_t>(o);
	for (int i = 0; i < 3 * 5 * 2 * 32; i++) stats[i] = r[i];
}

uint32_t hmr_gpu_tu_chain(int16_t *orig, int orig_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size, int is_dst,
			  int scan_mode, int comp, int is_intra, int slice_is_intra, int sign_hiding, int per, int rem, int *ac_sum)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_tu_job jb = {};
	const int n = size;
	jb.orig_off = (uint32_t)(st.put2d(orig, orig_stride, n, n, 2) / 2); jb.orig_stride = n;
	jb.pred_off = (uint32_t)(st.put2d(pred, pred_stride, n, n, 2) / 2); jb.pred_stride = n;
	jb.p0 = (uint32_t)(scan_mode & 3) | ((uint32_t)comp << 2) | ((uint32_t)(is_intra != 0) << 4) | ((uint32_t)(slice_is_intra != 0) << 5) |
		((uint32_t)(sign_hiding != 0) << 6) | ((uint32_t)(is_dst != 0) << 7);
	jb.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	const size_t joff = st.zeros(sizeof jb);
	st.begin_outputs();
	const size_t lo = st.out((size_t)n * n * 2), ro = st.out((size_t)n * n * 2), so = st.out(4), ao = st.out(4);
	jb.lev_off = (uint32_t)(lo / 2);
	jb.rec_off = (uint32_t)(ro / 2); jb.rec_stride = n;
	memcpy(st.host<uint8_t>(joff), &jb, sizeof jb);
	st.upload();
	must(hmr_gpu_tu_chain_batch(c, st.dev<hmr_gpu_tu_job>(joff), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<uint32_t>(so),
				    st.dev<int32_t>(ao)),
	     "tu_chain");
	st.finish();
	memcpy(levels, st.host<int16_t>(lo), (size_t)n * n * 2);
	st.get2d(ro, recon, recon_stride, n, n, 2);
	*ac_sum = *st.host<int32_t>(ao);
	return *st.host<uint32_t>(so);
}

void hmr_gpu_intra_search(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int n, int left, int top, int bottom_left, int top_right,
			  int bl_size, int tr_size, int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits, double sqrt_lambda,
			  int16_t *adi, int16_t *adi_filtered, int16_t *pred, int pred_stride, int32_t *out, double *best_cost)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_intra_job jb = {};
	const int total = 4 * n + 1, ring = 2 * n + 1;
	jb.sqrt_lambda = sqrt_lambda;
	jb.orig_off = (uint32_t)(st.put2d(orig, orig_stride, n, n, 2) / 2); jb.orig_stride = n;
	// stage only the L-shaped neighbourhood the build reads: row 0 and column 0 of a (2n+1)^2 tile
	const size_t doff = st.zeros((size_t)ring * ring * 2);
	int16_t *tile = st.host<int16_t>(doff);
	const int rows = left ? n + (bottom_left ? bl_size : 0) : 0, cols = top ? n + (top_right ? tr_size : 0) : 0;
	if (left || top) tile[0] = decoded_corner[0];
	for (int y = 1; y <= rows; y++) tile[(size_t)y * ring] = decoded_corner[(size_t)y * decoded_stride];
	for (int x = 1; x <= cols; x++) tile[x] = decoded_corner[x];
	jb.dec_off = (uint32_t)(doff / 2); jb.dec_stride = ring;
	jb.flags = (uint32_t)(left != 0) | ((uint32_t)(top != 0) << 1) | ((uint32_t)(bottom_left != 0) << 2) | ((uint32_t)(top_right != 0) << 3) |
		   ((uint32_t)(strong_enabled != 0) << 5);
	jb.sizes = (uint32_t)bl_size | ((uint32_t)tr_size << 16);
	for (int i = 0; i < 3; i++) { jb.preds[i] = preds[i]; jb.pred_bits[i] = (uint32_t)pred_bits[i]; }
	jb.other_bits = (uint32_t)other_bits;
	const size_t joff = st.zeros(sizeof jb);
	st.begin_outputs();
	const size_t ao = st.out((size_t)total * 2), fo = st.out((size_t)total * 2), po = st.out((size_t)n * n * 2), ro = st.out(sizeof(hmr_gpu_intra_result));
	jb.adi_off = (uint32_t)(ao / 2); jb.adif_off = (uint32_t)(fo / 2);
	jb.pred_off = (uint32_t)(po / 2); jb.pred_stride = n;
	memcpy(st.host<uint8_t>(joff), &jb, sizeof jb);
	st.upload();
	must(hmr_gpu_intra_search_batch(c, st.dev<hmr_gpu_intra_job>(joff), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<hmr_gpu_intra_result>(ro)),
	     "intra_search");
	st.finish();
	memcpy(adi, st.host<int16_t>(ao), (size_t)total * 2);
	memcpy(adi_filtered, st.host<int16_t>(fo), (size_t)total * 2);
	st.get2d(po, pred, pred_stride, n, n, 2);
	const hmr_gpu_intra_result *r = st.host<hmr_gpu_intra_result>(ro);
	out[0] = r->best_mode;
	out[1] = r->bits;
	*best_cost = r->cost;
}

/* ---- in-loop filters at the reference's own call granularity (one CTU per call).  Only the CTU's neighbourhood travels: the tile is addressed
 * through a virtual picture origin, the kernels index with picture coordinates and touch nothing outside the region of interest. ---- */
void hmr_gpu_deblock_filter_ctu(int16_t *const planes[3], const int strides[3], int width, int height, int units_stride, const int16_t *mvx, const int16_t *mvy,
				const int8_t *ref_idx, const uint8_t *qp, const uint8_t *unit_flags, const uint8_t *pred_depth, const uint8_t *tr_idx, int ctu_x,
				int ctu_y, int ctu_size, int dir, int cb_qp_offset, int cr_qp_offset, int beta_offset_div2, int tc_offset_div2)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	// luma tile: the CTU grown by 8 samples to the left / top (edges on the CTU boundary modify 3 samples of the neighbour and read 4)
	const int x0 = ctu_x >= 8 ? ctu_x - 8 : 0, y0 = ctu_y >= 8 ? ctu_y - 8 : 0;
	const int x1 = ctu_x + ctu_size < width ? ctu_x + ctu_size : width, y1 = ctu_y + ctu_size < height ? ctu_y + ctu_size : height;
	const int tw = x1 - x0, th = y1 - y0, uw = tw / 4, uh = th / 4, ux0 = x0 / 4, uy0 = y0 / 4;
	// side-info tiles (inputs)
	size_t o_mvx = st.zeros((size_t)uw * uh * 2), o_mvy = st.zeros((size_t)uw * uh * 2), o_ref = st.zeros((size_t)uw * uh), o_qp = st.zeros((size_t)uw * uh),
	       o_fl = st.zeros((size_t)uw * uh), o_pd = st.zeros((size_t)uw * uh), o_tr = st.zeros((size_t)uw * uh);
	for (int y = 0; y < uh; y++) {
		const size_t src = (size_t)(uy0 + y) * units_stride + ux0, dst = (size_t)y * uw;
		memcpy(st.host<int16_t>(o_mvx) + dst, mvx + src, (size_t)uw * 2);
		memcpy(st.host<int16_t>(o_mvy) + dst, mvy + src, (size_t)uw * 2);
		memcpy(st.host<int8_t>(o_ref) + dst, ref_idx + src, (size_t)uw);
		memcpy(st.host<uint8_t>(o_qp) + dst, qp + src, (size_t)uw);
		for (int x = 0; x < uw; x++) st.host<uint8_t>(o_fl)[dst + x] = unit_flags[src + x] & (HMR_GPU_UNIT_INTRA | HMR_GPU_UNIT_CBF_Y);
		memcpy(st.host<uint8_t>(o_pd) + dst, pred_depth + src, (size_t)uw);
		memcpy(st.host<uint8_t>(o_tr) + dst, tr_idx + src, (size_t)uw);
	}
	// sample tiles (in place)
	st.begin_outputs();
	size_t o_pl[3];
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, w = tw >> sh, h = th >> sh;
		o_pl[comp] = st.out((size_t)w * h * 2);
		for (int y = 0; y < h; y++)
			memcpy(st.host<int16_t>(o_pl[comp]) + (size_t)y * w, planes[comp] + (size_t)((y0 >> sh) + y) * strides[comp] + (x0 >> sh), (size_t)w * 2);
	}
	st.upload_all();
	hmr_gpu_frame f = {};
	f.width = width; f.height = height; f.stride_y = tw; f.stride_c = tw / 2;
	f.y = st.dev<int16_t>(o_pl[0]) - ((ptrdiff_t)y0 * tw + x0);
	f.u = st.dev<int16_t>(o_pl[1]) - ((ptrdiff_t)(y0 / 2) * (tw / 2) + x0 / 2);
	f.v = st.dev<int16_t>(o_pl[2]) - ((ptrdiff_t)(y0 / 2) * (tw / 2) + x0 / 2);
	hmr_gpu_units u = {};
	const ptrdiff_t uo = (ptrdiff_t)uy0 * uw + ux0;
	u.units_stride = uw;
	u.mvx = st.dev<int16_t>(o_mvx) - uo; u.mvy = st.dev<int16_t>(o_mvy) - uo; u.ref_idx = st.dev<int8_t>(o_ref) - uo;
	u.qp = st.dev<uint8_t>(o_qp) - uo; u.flags = st.dev<uint8_t>(o_fl) - uo;
	must(hmr_gpu_deblock_ctu(c, &f, &u, st.dev<uint8_t>(o_pd) - uo, st.dev<uint8_t>(o_tr) - uo, cb_qp_offset, cr_qp_offset, beta_offset_div2, tc_offset_div2, ctu_x,
				 ctu_y, ctu_size, dir),
	     "deblock_ctu");
	st.finish();
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, w = tw >> sh, h = th >> sh;
		st.get2d(o_pl[comp], planes[comp] + (size_t)(y0 >> sh) * strides[comp] + (x0 >> sh), strides[comp], h, w, 2);
	}
}

void hmr_gpu_sao_offset_ctu(const int16_t *const src[3], const int src_stride[3], int16_t *const dst[3], const int dst_stride[3], int width, int height, int ctu_x,
			    int ctu_y, const int32_t *params)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_frame fs = {}, fd = {};
	fs.width = fd.width = width; fs.height = fd.height = height;
	int16_t **ps[3] = {&fs.y, &fs.u, &fs.v}, **pd[3] = {&fd.y, &fd.u, &fd.v};
	const size_t o_par = st.put2d(params, 3 * 34, 1, 3 * 34, 4);
	size_t o_src[3], o_dst[3];
	int x0[3], y0[3], x1[3], y1[3], pitch[3];
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, pw = width >> sh, ph = height >> sh, x = ctu_x >> sh, y = ctu_y >> sh, n = 64 >> sh;
		x0[comp] = x > 0 ? x - 1 : 0; y0[comp] = y > 0 ? y - 1 : 0;
		x1[comp] = x + n + 1 < pw ? x + n + 1 : pw; y1[comp] = y + n + 1 < ph ? y + n + 1 : ph;
		pitch[comp] = comp ? 34 : 66;     // one pitch per plane type, so that one stride serves U and V
		o_src[comp] = st.zeros((size_t)pitch[comp] * pitch[comp] * 2);
		for (int yy = y0[comp]; yy < y1[comp]; yy++)
			memcpy(st.host<int16_t>(o_src[comp]) + (size_t)(yy - y0[comp]) * pitch[comp], src[comp] + (size_t)yy * src_stride[comp] + x0[comp],
			       (size_t)(x1[comp] - x0[comp]) * 2);
	}
	st.begin_outputs();
	for (int comp = 0; comp < 3; comp++) {
		o_dst[comp] = st.out((size_t)pitch[comp] * pitch[comp] * 2);
		for (int yy = y0[comp]; yy < y1[comp]; yy++)    // the samples a class does not touch keep the destination's own value
			memcpy(st.host<int16_t>(o_dst[comp]) + (size_t)(yy - y0[comp]) * pitch[comp], dst[comp] + (size_t)yy * dst_stride[comp] + x0[comp],
			       (size_t)(x1[comp] - x0[comp]) * 2);
		*ps[comp] = st.dev<int16_t>(o_src[comp]) - ((ptrdiff_t)y0[comp] * pitch[comp] + x0[comp]);
		*pd[comp] = st.dev<int16_t>(o_dst[comp]) - ((ptrdiff_t)y0[comp] * pitch[comp] + x0[comp]);
	}
	fs.stride_y = fd.stride_y = pitch[0]; fs.stride_c = fd.stride_c = pitch[1];
	st.upload_all();
	const int ctus_x = (width + 63) / 64;
	must(hmr_gpu_sao_apply_ctu(c, &fs, &fd, (ctu_y / 64) * ctus_x + ctu_x / 64, st.dev<int32_t>(o_par)), "sao_offset_ctu");
	st.finish();
	for (int comp = 0; comp < 3; comp++) {
		if (!params[comp * 34]) continue;       // SAO_MODE_OFF: the reference does not touch the component
		const int sh = comp ? 1 : 0, pw = width >> sh, ph = height >> sh, x = ctu_x >> sh, y = ctu_y >> sh, n = 64 >> sh;
		const int xe = x + n < pw ? x + n : pw, ye = y + n < ph ? y + n : ph;
		for (int yy = y; yy < ye; yy++)
			memcpy(dst[comp] + (size_t)yy * dst_stride[comp] + x, st.host<int16_t>(o_dst[comp]) + (size_t)(yy - y0[comp]) * pitch[comp] + (x - x0[comp]),
			       (size_t)(xe - x) * 2);
	}
}

void hmr_gpu_pad_ctu(int16_t *const planes[3], const int strides[3], int width, int height, int pad_x, int pad_y, int ctu_x, int ctu_y, int ctu_size)
{
	const bool left = ctu_x == 0, top = ctu_y == 0, right = ctu_x + ctu_size >= width, bottom = ctu_y + ctu_size >= height;
	if (!(left || top || right || bottom)) return;     // interior CTU: nothing to replicate (hmr_encoder_lib.c:1730-1740)
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	const int bw = (right ? width : ctu_x + ctu_size) - ctu_x, bh = (bottom ? height : ctu_y + ctu_size) - ctu_y;
	st.begin_outputs();
	size_t off[3];
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, w = bw >> sh, h = bh >> sh, px = pad_x >> sh, py = pad_y >> sh, pitch = w + 2 * px;
		off[comp] = st.out((size_t)pitch * (h + 2 * py) * 2);
		for (int y = 0; y < h; y++)
			memcpy(st.host<int16_t>(off[comp]) + (size_t)(y + py) * pitch + px, planes[comp] + (size_t)((ctu_y >> sh) + y) * strides[comp] + (ctu_x >> sh), (size_t)w * 2);
	}
	st.upload_all();
	// the CTU's block as a picture of its own: replicating ITS edges equals replicating the picture's edges on the sides where they coincide
	hmr_gpu_frame f = {};
	f.width = bw; f.height = bh; f.stride_y = bw + 2 * pad_x; f.stride_c = bw / 2 + 2 * (pad_x / 2);
	f.y = st.dev<int16_t>(off[0]) + (size_t)pad_y * f.stride_y + pad_x;
	f.u = st.dev<int16_t>(off[1]) + (size_t)(pad_y / 2) * f.stride_c + pad_x / 2;
	f.v = st.dev<int16_t>(off[2]) + (size_t)(pad_y / 2) * f.stride_c + pad_x / 2;
	must(hmr_gpu_pad_frame(c, &f, pad_x, pad_y), "pad_ctu");
	st.finish();
	for (int comp = 0; comp < 3; comp++) {
		const int sh = comp ? 1 : 0, w = bw >> sh, h = bh >> sh, px = pad_x >> sh, py = pad_y >> sh, pitch = w + 2 * px;
		const int xa = left ? -px : 0, xb = right ? w + px : w, ya = top ? -py : 0, yb = bottom ? h + py : h;
		for (int y = ya; y < yb; y++)
			memcpy(planes[comp] + ((ptrdiff_t)((ctu_y >> sh) + y)) * strides[comp] + (ctu_x >> sh) + xa,
			       st.host<int16_t>(off[comp]) + (size_t)(y + py) * pitch + px + xa, (size_t)(xb - xa) * 2);
	}
}

uint32_t hmr_gpu_intra_tu_chain(int16_t *orig, int orig_stride, int16_t *decoded_corner, int decoded_stride, int left, int top, int bottom_left, int top_right,
				int bl_size, int tr_size, int strong_enabled, int is_filtered, int mode, int is_luma, int16_t *pred, int pred_stride, int16_t *levels,
				int16_t *recon, int recon_stride, int size, int is_dst, int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem,
				int *ac_sum)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_itu_job jb = {};
	const int n = size, ring = 2 * n + 1;
	jb.orig_off = (uint32_t)(st.put2d(orig, orig_stride, n, n, 2) / 2); jb.orig_stride = n;
	// the L-shaped neighbourhood: row 0 and column 0 of a (2n+1)^2 tile
	const size_t doff = st.zeros((size_t)ring * ring * 2);
	int16_t *tile = st.host<int16_t>(doff);
	const int rows = left ? n + (bottom_left ? bl_size : 0) : 0, cols = top ? n + (top_right ? tr_size : 0) : 0;
	if (left || top) tile[0] = decoded_corner[0];
	for (int y = 1; y <= rows; y++) tile[(size_t)y * ring] = decoded_corner[(size_t)y * decoded_stride];
	for (int x = 1; x <= cols; x++) tile[x] = decoded_corner[x];
	jb.dec_off = (uint32_t)(doff / 2); jb.dec_stride = ring;
	jb.flags = (uint32_t)(left != 0) | ((uint32_t)(top != 0) << 1) | ((uint32_t)(bottom_left != 0) << 2) | ((uint32_t)(top_right != 0) << 3) |
		   ((uint32_t)(strong_enabled != 0) << 5) | ((uint32_t)(is_filtered != 0) << 6) | ((uint32_t)(is_luma != 0) << 7);
	jb.sizes = (uint32_t)bl_size | ((uint32_t)tr_size << 16);
	jb.mode = (uint32_t)mode;
	jb.p0 = (uint32_t)(scan_mode & 3) | ((uint32_t)comp << 2) | (1u << 4) | ((uint32_t)(slice_is_intra != 0) << 5) | ((uint32_t)(sign_hiding != 0) << 6) |
		((uint32_t)(is_dst != 0) << 7);
	jb.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	const size_t joff = st.zeros(sizeof jb);
	st.begin_outputs();
	const size_t po = st.out((size_t)n * n * 2), lo = st.out((size_t)n * n * 2), ro = st.out((size_t)n * n * 2), so = st.out(4), ao = st.out(4);
	jb.pred_off = (uint32_t)(po / 2); jb.pred_stride = n;
	jb.lev_off = (uint32_t)(lo / 2);
	jb.rec_off = (uint32_t)(ro / 2); jb.rec_stride = n;
	memcpy(st.host<uint8_t>(joff), &jb, sizeof jb);
	st.upload();
	must(hmr_gpu_intra_tu_chain_batch(c, st.dev<hmr_gpu_itu_job>(joff), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(),
					  st.dev<int16_t>(), st.dev<uint32_t>(so), st.dev<int32_t>(ao)),
	     "intra_tu_chain");
	st.finish();
	st.get2d(po, pred, pred_stride, n, n, 2);
	memcpy(levels, st.host<int16_t>(lo), (size_t)n * n * 2);
	st.get2d(ro, recon, recon_stride, n, n, 2);
	*ac_sum = *st.host<int32_t>(ao);
	return *st.host<uint32_t>(so);
}

uint32_t hmr_gpu_inter_tu_chain(int16_t *residual, int residual_stride, int16_t *pred, int pred_stride, int16_t *levels, int16_t *recon, int recon_stride, int size,
				int scan_mode, int comp, int slice_is_intra, int sign_hiding, int per, int rem, double weight, double zero_thr, int *ac_sum)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_inter_tu_job jb = {};
	const int n = size;
	jb.orig_off = (uint32_t)(st.put2d(residual, residual_stride, n, n, 2) / 2); jb.orig_stride = n;
	jb.pred_off = (uint32_t)(st.put2d(pred, pred_stride, n, n, 2) / 2); jb.pred_stride = n;
	jb.p0 = (uint32_t)(scan_mode & 3) | ((uint32_t)comp << 2) | ((uint32_t)(slice_is_intra != 0) << 5) | ((uint32_t)(sign_hiding != 0) << 6);
	jb.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	jb.weight = weight; jb.zero_thr = zero_thr;
	const size_t joff = st.zeros(sizeof jb);
	st.begin_outputs();
	const size_t lo = st.out((size_t)n * n * 2), ro = st.out((size_t)n * n * 2), so = st.out(4), ao = st.out(4);
	jb.lev_off = (uint32_t)(lo / 2);
	jb.rec_off = (uint32_t)(ro / 2); jb.rec_stride = n;
	memcpy(st.host<uint8_t>(joff), &jb, sizeof jb);
	st.upload();
	must(hmr_gpu_inter_tu_chain_batch(c, st.dev<hmr_gpu_inter_tu_job>(joff), 1, n, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(),
					  st.dev<uint32_t>(so), st.dev<int32_t>(ao)),
	     "inter_tu_chain");
	st.finish();
	memcpy(levels, st.host<int16_t>(lo), (size_t)n * n * 2);
	st.get2d(ro, recon, recon_stride, n, n, 2);
	*ac_sum = *st.host<int32_t>(ao);
	return *st.host<uint32_t>(so);
}

}  // extern "C"

/* ---- encode_intra_luma's data path for one 2Nx2N CU (hmr_motion_intra.c:1226-1632): search -> parent TU -> four child TUs (one launch, four rounds) -> consolidation, four
 * launches in stream order on one staged image; the mode never leaves the device between them. ---- */
void hmr_gpu_intra_luma_cu(int16_t *orig, int orig_stride, int16_t *dec_par, int dec_par_stride, int16_t *dec_chl, int dec_chl_stride, const int32_t *nb,
			   int strong_enabled, const int32_t *preds, const int32_t *pred_bits, int other_bits, double sqrt_lambda, int16_t *adi, int16_t *adi_filtered,
			   int16_t *pred, int pred_stride, int16_t *lev_par, int16_t *lev_chl, int size, int slice_is_intra, int sign_hiding, int per, int rem, int rule,
			   int32_t *out, double *best_cost)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	const int n = size, h = n / 2, ring = 2 * n + 1, total = 4 * n + 1;
	const bool has_parent = n <= 32;
	const size_t oo = st.put2d(orig, orig_stride, n, n, 2);
	// node geometry inside the CU and how far the L-shaped neighbourhood is read
	const int gx[5] = {0, 0, h, 0, h}, gy[5] = {0, 0, 0, h, h}, gs[5] = {n, h, h, h, h};
	int rows = 0, cols = 0;
	for (int k = 0; k < 5; k++) {
		const int32_t *f = nb + 6 * k;
		if (gx[k] == 0 && f[0]) { const int r = gy[k] + gs[k] + (f[2] ? f[4] : 0); rows = r > rows ? r : rows; }
		if (gy[k] == 0 && f[1]) { const int q = gx[k] + gs[k] + (f[3] ? f[5] : 0); cols = q > cols ? q : cols; }
	}
	hmr_gpu_intra_job sj = {};
	hmr_gpu_itu_job tj[5] = {};
	hmr_gpu_tree_job dj = {};
	const size_t so = st.zeros(sizeof sj), to = st.zeros(sizeof tj), d_o = st.zeros(sizeof dj);
	st.begin_outputs();
	// in/out: the two planes (neighbourhood + CU interior)
	const size_t pp = st.out((size_t)ring * ring * 2), pc = st.out((size_t)ring * ring * 2);
	for (int w = 0; w < 2; w++) {
		int16_t *tile = st.host<int16_t>(w ? pc : pp);
		const int16_t *src = w ? dec_chl : dec_par;
		const int ss = w ? dec_chl_stride : dec_par_stride;
		memset(tile, 0, (size_t)ring * ring * 2);
		if (rows || cols) tile[0] = src[-ss - 1];
		for (int y = 0; y < rows; y++) tile[(size_t)(y + 1) * ring] = src[(ptrdiff_t)y * ss - 1];
		for (int x = 0; x < cols; x++) tile[x + 1] = src[-ss + x];
		for (int y = 0; y < n; y++) memcpy(tile + (size_t)(y + 1) * ring + 1, src + (ptrdiff_t)y * ss, (size_t)n * 2);
	}
	const size_t ao = st.out((size_t)total * 2), fo = st.out((size_t)total * 2), po = st.out((size_t)n * n * 2), ro = st.out(sizeof(hmr_gpu_intra_result));
	const size_t lp = st.out((size_t)n * n * 2), lc = st.out((size_t)n * n * 2), sso = st.out(5 * 4), aco = st.out(5 * 4), tro = st.out(sizeof(hmr_gpu_tree_result));
	memset(st.host<uint8_t>(sso), 0, 20);
	memset(st.host<uint8_t>(aco), 0, 20);
	auto flags_of = [&](const int32_t *f) {
		return (uint32_t)(f[0] != 0) | ((uint32_t)(f[1] != 0) << 1) | ((uint32_t)(f[2] != 0) << 2) | ((uint32_t)(f[3] != 0) << 3) | ((uint32_t)(strong_enabled != 0) << 5);
	};
	sj.sqrt_lambda = sqrt_lambda;
	sj.orig_off = (uint32_t)(oo / 2); sj.orig_stride = n;
	sj.dec_off = (uint32_t)(pp / 2); sj.dec_stride = ring;
	sj.flags = flags_of(nb);
	sj.sizes = (uint32_t)nb[4] | ((uint32_t)nb[5] << 16);
	for (int i = 0; i < 3; i++) { sj.preds[i] = preds[i]; sj.pred_bits[i] = (uint32_t)pred_bits[i]; }
	sj.other_bits = (uint32_t)other_bits;
	sj.adi_off = (uint32_t)(ao / 2); sj.adif_off = (uint32_t)(fo / 2);
	sj.pred_off = (uint32_t)(po / 2); sj.pred_stride = n;
	for (int k = 0; k < 5; k++) {
		const int32_t *f = nb + 6 * k;
		const size_t plane = k ? pc : pp;
		hmr_gpu_itu_job &j = tj[k];
		j.orig_off = (uint32_t)(oo / 2 + (size_t)gy[k] * n + gx[k]); j.orig_stride = n;
		j.pred_off = (uint32_t)(po / 2 + (size_t)gy[k] * n + gx[k]); j.pred_stride = n;
		j.dec_off = (uint32_t)(plane / 2 + (size_t)gy[k] * ring + gx[k]); j.dec_stride = ring;
		j.rec_off = j.dec_off + ring + 1; j.rec_stride = ring;
		j.lev_off = (uint32_t)(k ? lc / 2 + (size_t)(k - 1) * h * h : lp / 2);
		j.flags = flags_of(f) | (1u << 7) | HMR_GPU_ITU_MODE_FROM_SEARCH;
		j.sizes = (uint32_t)f[4] | ((uint32_t)f[5] << 16);
		j.mode = 0;
		j.p0 = (1u << 4) | ((uint32_t)(slice_is_intra != 0) << 5) | ((uint32_t)(sign_hiding != 0) << 6) | ((uint32_t)(gs[k] == 4) << 7);
		j.p1 = (uint32_t)per | ((uint32_t)rem << 8);
	}
	dj.parent = has_parent ? 0u : HMR_GPU_TREE_NO_PARENT;
	for (int k = 0; k < 4; k++) dj.child[k] = (uint32_t)(k + 1);
	dj.par_rec_off = (uint32_t)(pp / 2 + ring + 1); dj.par_rec_stride = ring;
	dj.chl_rec_off = (uint32_t)(pc / 2 + ring + 1); dj.chl_rec_stride = ring;
	dj.par_lev_off = (uint32_t)(lp / 2); dj.chl_lev_off = (uint32_t)(lc / 2);
	dj.size = (uint32_t)n; dj.rule = (uint32_t)rule;
	memcpy(st.host<uint8_t>(so), &sj, sizeof sj);
	memcpy(st.host<uint8_t>(to), tj, sizeof tj);
	memcpy(st.host<uint8_t>(d_o), &dj, sizeof dj);
	st.upload_all();
	int16_t *base = st.dev<int16_t>();
	hmr_gpu_intra_result *modes = st.dev<hmr_gpu_intra_result>(ro);
	must(hmr_gpu_intra_search_batch(c, st.dev<hmr_gpu_intra_job>(so), 1, n, base, base, base, modes), "intra_luma_cu: search");
	if (has_parent)
		must(hmr_gpu_intra_tu_chain_modes_batch(c, st.dev<hmr_gpu_itu_job>(to), 1, n, base, base, base, base, base, st.dev<uint32_t>(sso), st.dev<int32_t>(aco), modes),
		     "intra_luma_cu: parent TU");
	// the four children back to back in one launch (each reads its siblings' reconstruction)
	must(hmr_gpu_intra_tu_chain_rounds_batch(c, st.dev<hmr_gpu_itu_job>(to) + 1, 1, 4, h, base, base, base, base, base, st.dev<uint32_t>(sso) + 1,
						 st.dev<int32_t>(aco) + 1, modes),
	     "intra_luma_cu: child TUs");
	must(hmr_gpu_tree_decide_batch(c, st.dev<hmr_gpu_tree_job>(d_o), 1, st.dev<uint32_t>(sso), st.dev<int32_t>(aco), base, base, st.dev<hmr_gpu_tree_result>(tro)),
	     "intra_luma_cu: consolidation");
	st.finish();
	for (int w = 0; w < 2; w++) {
		const int16_t *tile = st.host<int16_t>(w ? pc : pp);
		int16_t *dst = w ? dec_chl : dec_par;
		const int ds = w ? dec_chl_stride : dec_par_stride;
		for (int y = 0; y < n; y++) memcpy(dst + (ptrdiff_t)y * ds, tile + (size_t)(y + 1) * ring + 1, (size_t)n * 2);
	}
	memcpy(adi, st.host<int16_t>(ao), (size_t)total * 2);
	memcpy(adi_filtered, st.host<int16_t>(fo), (size_t)total * 2);
	st.get2d(po, pred, pred_stride, n, n, 2);
	memcpy(lev_par, st.host<int16_t>(lp), (size_t)n * n * 2);
	memcpy(lev_chl, st.host<int16_t>(lc), (size_t)n * n * 2);
	const hmr_gpu_intra_result *sr = st.host<hmr_gpu_intra_result>(ro);
	const hmr_gpu_tree_result *tr = st.host<hmr_gpu_tree_result>(tro);
	const uint32_t *ssd = st.host<uint32_t>(sso);
	const int32_t *ac = st.host<int32_t>(aco);
	out[0] = (int32_t)tr->split;
	out[1] = out[2] = (int32_t)tr->cost;
	out[3] = (int32_t)tr->sum;
	for (int k = 0; k < 4; k++) out[4 + k] = tr->cbf[k];
	out[8] = (int32_t)tr->split;
	for (int k = 0; k < 5; k++) { out[9 + k] = (int32_t)ssd[k]; out[14 + k] = ac[k]; }
	if (tr->split) { out[9] = (int32_t)tr->cost; out[14] = (int32_t)tr->sum; }
	out[19] = sr->best_mode;
	out[20] = sr->bits;
	*best_cost = sr->cost;
}

/* ---- encode_intra_chroma's data path for one 2Nx2N CU (hmr_motion_intra_chroma.c:114-471): chroma search -> the U / V TUs of the winner, two launches in
 * stream order on one staged image; the mode never leaves the device between them. ---- */
void hmr_gpu_intra_chroma_cu(int16_t *orig_u, int16_t *orig_v, int orig_stride, int16_t *dec_u, int16_t *dec_v, int dec_stride, const int32_t *nb, int luma_mode,
			     int split, double sqrt_lambda, double weight, int16_t *pred_u, int16_t *pred_v, int pred_stride, int16_t *lev_u, int16_t *lev_v, int size,
			     int slice_is_intra, int sign_hiding, int per, int rem, int32_t *out)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	const int n = size, ring = 2 * n + 1;
	const bool do_split = split && n > 4;
	const int tn = do_split ? n / 2 : n, ntu = do_split ? 4 : 1;
	const int ss = n == 32 ? 16 : n;                        // a 64x64 CU is searched on its first quadrant only (:165-169)
	const int32_t *snb = n == 32 ? nb + 6 : nb;
	int16_t *orig[2] = {orig_u, orig_v}, *dec[2] = {dec_u, dec_v}, *pred[2] = {pred_u, pred_v}, *lev[2] = {lev_u, lev_v};
	size_t oo[2];
	for (int k = 0; k < 2; k++) oo[k] = st.put2d(orig[k], orig_stride, n, n, 2);
	// how far the L-shaped neighbourhood is read (CU node and quadrants, as in hmr_gpu_intra_luma_cu)
	const int h = n / 2;
	const int gx[5] = {0, 0, h, 0, h}, gy[5] = {0, 0, 0, h, h}, gs[5] = {n, h, h, h, h};
	int rows = 0, cols = 0;
	for (int k = 0; k < 5; k++) {
		const int32_t *f = nb + 6 * k;
		if (gx[k] == 0 && f[0]) { const int r = gy[k] + gs[k] + (f[2] ? f[4] : 0); rows = r > rows ? r : rows; }
		if (gy[k] == 0 && f[1]) { const int q = gx[k] + gs[k] + (f[3] ? f[5] : 0); cols = q > cols ? q : cols; }
	}
	hmr_gpu_chroma_job sj = {};
	hmr_gpu_itu_job tj[8] = {};
	const size_t so = st.zeros(sizeof sj), to = st.zeros(sizeof tj);
	st.begin_outputs();
	size_t pl[2];
	for (int k = 0; k < 2; k++) {
		pl[k] = st.out((size_t)ring * ring * 2);
		int16_t *tile = st.host<int16_t>(pl[k]);
		const int16_t *src = dec[k];
		memset(tile, 0, (size_t)ring * ring * 2);
		if (rows || cols) tile[0] = src[-dec_stride - 1];
		for (int y = 0; y < rows; y++) tile[(size_t)(y + 1) * ring] = src[(ptrdiff_t)y * dec_stride - 1];
		for (int x = 0; x < cols; x++) tile[x + 1] = src[-dec_stride + x];
		for (int y = 0; y < n; y++) memcpy(tile + (size_t)(y + 1) * ring + 1, src + (ptrdiff_t)y * dec_stride, (size_t)n * 2);
	}
	size_t po[2], lo[2];
	for (int k = 0; k < 2; k++) { po[k] = st.out((size_t)n * n * 2); lo[k] = st.out((size_t)n * n * 2); }
	const size_t ro = st.out(sizeof(hmr_gpu_intra_result)), sso = st.out(8 * 4), aco = st.out(8 * 4);
	memset(st.host<uint8_t>(sso), 0, 32);
	memset(st.host<uint8_t>(aco), 0, 32);
	auto flags_of = [&](const int32_t *f) { return (uint32_t)(f[0] != 0) | ((uint32_t)(f[1] != 0) << 1) | ((uint32_t)(f[2] != 0) << 2) | ((uint32_t)(f[3] != 0) << 3); };
	sj.sqrt_lambda = sqrt_lambda;
	sj.orig_u_off = (uint32_t)(oo[0] / 2); sj.orig_v_off = (uint32_t)(oo[1] / 2); sj.orig_stride = n;
	sj.dec_u_off = (uint32_t)(pl[0] / 2); sj.dec_v_off = (uint32_t)(pl[1] / 2); sj.dec_stride = ring;
	sj.flags = flags_of(snb); sj.sizes = (uint32_t)snb[4] | ((uint32_t)snb[5] << 16);
	sj.luma_mode = (uint32_t)luma_mode;
	// TU jobs: round r (= quadrant when split) holds the U and the V job
	for (int r = 0; r < ntu; r++)
		for (int k = 0; k < 2; k++) {
			const int x0 = do_split ? (r & 1) * tn : 0, y0 = do_split ? (r >> 1) * tn : 0;
			const int32_t *f = nb + (do_split ? 6 * (r + 1) : 0);
			hmr_gpu_itu_job &j = tj[2 * r + k];
			j.orig_off = (uint32_t)(oo[k] / 2 + (size_t)y0 * n + x0); j.orig_stride = n;
			j.pred_off = (uint32_t)(po[k] / 2 + (size_t)y0 * n + x0); j.pred_stride = n;
			j.dec_off = (uint32_t)(pl[k] / 2 + (size_t)y0 * ring + x0); j.dec_stride = ring;
			j.rec_off = j.dec_off + ring + 1; j.rec_stride = ring;
			j.lev_off = (uint32_t)(lo[k] / 2 + (size_t)r * tn * tn);
			j.flags = flags_of(f) | HMR_GPU_ITU_MODE_FROM_SEARCH;           /* is_luma = 0 */
			j.sizes = (uint32_t)f[4] | ((uint32_t)f[5] << 16);
			j.mode = 0;
			j.p0 = ((uint32_t)(k + 1) << 2) | (1u << 4) | ((uint32_t)(slice_is_intra != 0) << 5) | ((uint32_t)(sign_hiding != 0) << 6);
			j.p1 = (uint32_t)per | ((uint32_t)rem << 8);
		}
	memcpy(st.host<uint8_t>(so), &sj, sizeof sj);
	memcpy(st.host<uint8_t>(to), tj, sizeof tj);
	st.upload_all();
	int16_t *base = st.dev<int16_t>();
	hmr_gpu_intra_result *modes = st.dev<hmr_gpu_intra_result>(ro);
	must(hmr_gpu_chroma_search_batch(c, st.dev<hmr_gpu_chroma_job>(so), 1, ss, base, base, nullptr, modes), "intra_chroma_cu: search");
	must(hmr_gpu_intra_tu_chain_rounds_batch(c, st.dev<hmr_gpu_itu_job>(to), 2, ntu, tn, base, base, base, base, base, st.dev<uint32_t>(sso), st.dev<int32_t>(aco), modes),
	     "intra_chroma_cu: TUs");
	st.finish();
	for (int k = 0; k < 2; k++) {
		const int16_t *tile = st.host<int16_t>(pl[k]);
		for (int y = 0; y < n; y++) memcpy(dec[k] + (ptrdiff_t)y * dec_stride, tile + (size_t)(y + 1) * ring + 1, (size_t)n * 2);
		st.get2d(po[k], pred[k], pred_stride, n, n, 2);
		memcpy(lev[k], st.host<int16_t>(lo[k]), (size_t)n * n * 2);
	}
	const hmr_gpu_intra_result *sr = st.host<hmr_gpu_intra_result>(ro);
	const uint32_t *ssd = st.host<uint32_t>(sso);
	const int32_t *ac = st.host<int32_t>(aco);
	uint32_t distortion = 0, sum = 0;
	for (int k = 0; k < 8; k++) out[6 + k] = 0;
	for (int r = 0; r < ntu; r++) {
		int part = 0;
		for (int k = 0; k < 2; k++) {
			part += (int)(weight * ssd[2 * r + k]);                 /* :331 */
			sum += (uint32_t)ac[2 * r + k];
			out[6 + 4 * k + r] = ac[2 * r + k];
		}
		distortion += (uint32_t)part;
	}
	out[0] = sr->best_mode >> 8;
	out[1] = sr->best_mode & 0xff;
	out[2] = sr->bits;
	out[3] = (int32_t)(uint32_t)sr->cost;
	out[4] = (int32_t)distortion;
	out[5] = (int32_t)sum;
}

/* ---- sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion for the 15 (component, type) pairs of one CTU (hmr_sao.c:480-659) ---- */
void hmr_gpu_sao_offsets_ctu(const int32_t *stats, const double *lambdas, int32_t *offsets, int32_t *aux, int64_t *dist)
{
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	const size_t so = st.put2d(stats, 960, 1, 960, 4), lo = st.put2d(lambdas, 3, 1, 3, 8);
	st.begin_outputs();
	const size_t oo = st.out(3 * 5 * 32 * 4), ao = st.out(15 * 4), dd = st.out(15 * 8);
	st.upload();
	must(hmr_gpu_sao_offsets_frame(c, st.dev<int32_t>(so), 1, st.dev<double>(lo), st.dev<int32_t>(oo), st.dev<int32_t>(ao), st.dev<int64_t>(dd)), "sao_offsets");
	st.finish();
	memcpy(offsets, st.host<int32_t>(oo), 3 * 5 * 32 * 4);
	memcpy(aux, st.host<int32_t>(ao), 15 * 4);
	memcpy(dist, st.host<int64_t>(dd), 15 * 8);
}

/* ---- the inter TUs of one CU's transform tree (encode_inter, hmr_motion_inter.c:3069: encode_inter_cu + encode_inter_cu_chroma per node) in ONE submission:
 * they only read the CU's residual and prediction, so the whole tree can be computed ahead of the walk that compares and consolidates it. ---- */
void hmr_gpu_inter_tu_chain_n(hmr_gpu_inter_tu_host *tus, int n)
{
	if (n <= 0) return;
	if (n > 32) { fprintf(stderr, "homer_gpu: inter_tu_chain_n takes at most 32 TUs\n"); abort(); }
	hmr_gpu_ctx *c = hmr_default_ctx();
	Stager st(c);
	hmr_gpu_inter_tu_job jb[32] = {};
	int order[32], cnt = 0;
	const int sizes[4] = {32, 16, 8, 4};
	int seg_first[4], seg_n[4];
	for (int s = 0; s < 4; s++) {            // jobs grouped by TU size: one segment of the multi launch each
		seg_first[s] = cnt;
		for (int i = 0; i < n; i++)
			if (tus[i].size == sizes[s]) order[cnt++] = i;
		seg_n[s] = cnt - seg_first[s];
	}
	if (cnt != n) { fprintf(stderr, "homer_gpu: inter_tu_chain_n: TU size must be 4, 8, 16 or 32\n"); abort(); }
	for (int k = 0; k < n; k++) {
		const hmr_gpu_inter_tu_host &t = tus[order[k]];
		hmr_gpu_inter_tu_job &j = jb[k];
		j.orig_off = (uint32_t)(st.put2d(t.residual, t.residual_stride, t.size, t.size, 2) / 2); j.orig_stride = t.size;
		j.pred_off = (uint32_t)(st.put2d(t.pred, t.pred_stride, t.size, t.size, 2) / 2); j.pred_stride = t.size;
		j.p0 = (uint32_t)(t.scan_mode & 3) | ((uint32_t)t.comp << 2) | ((uint32_t)(t.slice_is_intra != 0) << 5) | ((uint32_t)(t.sign_hiding != 0) << 6);
		j.p1 = (uint32_t)t.per | ((uint32_t)t.rem << 8);
		j.weight = t.weight; j.zero_thr = t.zero_thr;
	}
	const size_t joff = st.zeros(sizeof jb);
	st.begin_outputs();
	size_t lo[32], ro[32];
	for (int k = 0; k < n; k++) {
		const int sz = tus[order[k]].size;
		lo[k] = st.out((size_t)sz * sz * 2); ro[k] = st.out((size_t)sz * sz * 2);
		jb[k].lev_off = (uint32_t)(lo[k] / 2);
		jb[k].rec_off = (uint32_t)(ro[k] / 2); jb[k].rec_stride = sz;
	}
	const size_t so = st.out(32 * 4), ao = st.out(32 * 4);
	memcpy(st.host<uint8_t>(joff), jb, sizeof jb);
	st.upload();
	hmr_gpu_tu_segment segs[4];
	int nseg = 0;
	for (int s = 0; s < 4; s++)
		if (seg_n[s]) {
			hmr_gpu_tu_segment &g = segs[nseg++];
			g.jobs = st.dev<hmr_gpu_inter_tu_job>(joff) + seg_first[s];
			g.ssd = st.dev<uint32_t>(so) + seg_first[s]; g.ac_sum = st.dev<int32_t>(ao) + seg_first[s]; g.modes = nullptr;
			g.njobs = seg_n[s]; g.size = sizes[s]; g.kind = 2; g.rounds = 0;
		}
	must(hmr_gpu_tu_chain_multi(c, segs, nseg, st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>(), st.dev<int16_t>()), "inter_tu_chain_n");
	st.finish();
	for (int k = 0; k < n; k++) {
		hmr_gpu_inter_tu_host &t = tus[order[k]];
		memcpy(t.levels, st.host<int16_t>(lo[k]), (size_t)t.size * t.size * 2);
		st.get2d(ro[k], t.recon, t.recon_stride, t.size, t.size, 2);
		t.ssd = st.host<uint32_t>(so)[k];
		t.ac_sum = st.host<int32_t>(ao)[k];
	}
}
