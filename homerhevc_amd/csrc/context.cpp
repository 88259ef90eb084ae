// Context: device selection, stream, constant-table upload, device memory helpers, HIP-event timer.
#include <mutex>
#include <stdlib.h>

#include "common.h"

static hmr_gpu_ctx *g_default = nullptr;

static std::recursive_mutex g_default_mutex;   // guards g_default (recursive: a failing create under hmr_default_ctx destroys its context)

static int ctx_init(hmr_gpu_ctx *c, void *stream)
{
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, c->device));
	c->num_cus = prop.multiProcessorCount;
	if (stream) {
		c->stream = (hipStream_t)stream;
		c->owns_stream = false;
	} else {
		const int every = getenv("HENC_CU_MASK_EVERY") ? atoi(getenv("HENC_CU_MASK_EVERY")) : 0;      // (experiment: the stream's kernels on every k-th CU only - how workers that share a CU slow each other, tools/cu_sharing.sh)
		if (every > 1) {
			uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int i = 0; i < c->num_cus && i < 256; i++)
				if (i % every == 0) mask[i >> 5] |= 1u << (i & 31);
			HIP_TRY(hipExtStreamCreateWithCUMask(&c->stream, 8, mask));
		} else HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->owns_stream = true;
	}
	HIP_TRY(hipEventCreate(&c->ev0));
	HIP_TRY(hipEventCreate(&c->ev1));
	HIP_TRY(hipMalloc((void **)&c->tables, sizeof(DevTables)));
	HIP_TRY(hipMemcpy(c->tables, hmr_host_tables(), sizeof(DevTables), hipMemcpyHostToDevice));
	c->stage_bytes = 64u << 20;   // largest drop-in operand: a whole 2160p picture going through sse_copy_8_16 (8 MB in, 17 MB out)
	return HMR_GPU_OK;
}

// The staging buffers of the drop-in layer, allocated when a drop-in entry first needs them: a frame encoder has a context per sequence (its stream), and an engine
// ring keeps hundreds of them per GPU - 64 MB of page-locked host memory and of HBM each would be most of the host's memory for nothing.
int hmr_ctx_need_stage(hmr_gpu_ctx *c)
{
	if (c->h_stage && c->d_stage) return HMR_GPU_OK;
	if (!c->h_stage) HIP_TRY(hipHostMalloc((void **)&c->h_stage, c->stage_bytes, hipHostMallocDefault));
	if (!c->d_stage) HIP_TRY(hipMalloc((void **)&c->d_stage, c->stage_bytes));
	return HMR_GPU_OK;
}

extern "C" void hmr_gpu_destroy(hmr_gpu_ctx *c);

extern "C" int hmr_gpu_create(hmr_gpu_ctx **out, int device, void *stream)
{
	if (!out) return HMR_GPU_ERR_ARG;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		hmr_set_error("no HIP device visible: the MI355X backend has no CPU fallback");
		return HMR_GPU_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= count) {
		hmr_set_error("device %d out of range (%d visible)", device, count);
		return HMR_GPU_ERR_ARG;
	}
	HIP_TRY(hipSetDevice(device));
	hmr_gpu_ctx *c = new hmr_gpu_ctx();
	memset(c, 0, sizeof *c);
	c->device = device;
	const int rc = ctx_init(c, stream);
	if (rc != HMR_GPU_OK) {
		hmr_gpu_destroy(c);   // releases whatever was created before the failing call
		return rc;
	}
	*out = c;
	return HMR_GPU_OK;
}

extern "C" void hmr_gpu_destroy(hmr_gpu_ctx *c)
{
	if (!c) return;
	{
		std::lock_guard<std::recursive_mutex> guard(g_default_mutex);
		if (g_default == c) g_default = nullptr;
	}
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->tables) (void)hipFree(c->tables);
	if (c->d_stage) (void)hipFree(c->d_stage);
	if (c->h_stage) (void)hipHostFree(c->h_stage);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int hmr_gpu_set_default(hmr_gpu_ctx *ctx)
{
	std::lock_guard<std::recursive_mutex> guard(g_default_mutex);
	g_default = ctx;
	return HMR_GPU_OK;
}

// default context for the drop-in entries: created on first use on device 0
hmr_gpu_ctx *hmr_default_ctx()
{
	std::lock_guard<std::recursive_mutex> guard(g_default_mutex);
	if (!g_default) {
		hmr_gpu_ctx *c = nullptr;
		if (hmr_gpu_create(&c, 0, nullptr) != HMR_GPU_OK) {
			fprintf(stderr, "homer_gpu: cannot create the default context: %s\n", hmr_gpu_last_error());
			abort();   // the table has no error channel (SURVEY.md §8-b "Errors"); never fall back silently
		}
		g_default = c;
	}
	return g_default;
}

extern "C" int hmr_gpu_sync(hmr_gpu_ctx *c)
{
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HMR_GPU_OK;
}
extern "C" void *hmr_gpu_stream(hmr_gpu_ctx *c) { return (void *)c->stream; }

extern "C" int hmr_gpu_malloc(hmr_gpu_ctx *c, void **p, size_t bytes)
{
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipMalloc(p, bytes));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_free(hmr_gpu_ctx *c, void *p)
{
	HIP_TRY(hipFree(p));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_upload(hmr_gpu_ctx *c, void *d, const void *h, size_t bytes)
{
	HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_download(hmr_gpu_ctx *c, void *h, const void *d, size_t bytes)
{
	HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_memset(hmr_gpu_ctx *c, void *d, int value, size_t bytes)
{
	HIP_TRY(hipMemsetAsync(d, value, bytes, c->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_timer_start(hmr_gpu_ctx *c)
{
	HIP_TRY(hipEventRecord(c->ev0, c->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_timer_stop(hmr_gpu_ctx *c, float *ms)
{
	HIP_TRY(hipEventRecord(c->ev1, c->stream));
	HIP_TRY(hipEventSynchronize(c->ev1));
	HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
	return HMR_GPU_OK;
}

// Event pairs for per-kernel timing on the context's stream (bench.py, roofline accounting)
extern "C" int hmr_gpu_event_create(hmr_gpu_ctx *c, void **ev)
{
	hipEvent_t e;
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipEventCreate(&e));
	*ev = (void *)e;
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_event_record(hmr_gpu_ctx *c, void *ev)
{
	HIP_TRY(hipEventRecord((hipEvent_t)ev, c->stream));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_event_elapsed(void *ev0, void *ev1, float *ms)
{
	HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)ev0, (hipEvent_t)ev1));
	return HMR_GPU_OK;
}
extern "C" int hmr_gpu_event_destroy(void *ev)
{
	HIP_TRY(hipEventDestroy((hipEvent_t)ev));
	return HMR_GPU_OK;
}

// Cap on the workgroups of one batched launch (default HMR_MAX_GRID): the kernels grid-stride over their jobs, so any cap computes the
// same results; a host may lower it to leave CUs to concurrent streams, the tests lower it to force many iterations per workgroup.
int g_hmr_max_grid = HMR_MAX_GRID;
extern "C" int hmr_gpu_set_max_grid(int blocks)
{
	if (blocks < 1) { hmr_set_error("hmr_gpu_set_max_grid: need at least one workgroup"); return HMR_GPU_ERR_ARG; }
	g_hmr_max_grid = blocks;
	return HMR_GPU_OK;
}

