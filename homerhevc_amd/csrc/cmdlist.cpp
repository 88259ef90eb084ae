// Command lists: the host describes one frame's launches once (kernel, job array, bases) and replays them from C, optionally
// bracketing every command with an event pair (per-kernel timing) or capturing the whole list into a hipGraph - a frame is a
// fixed launch sequence, so graph replay removes the per-launch host cost and the inter-kernel gaps.
#include <vector>

#include "common.h"

struct hmr_gpu_cmdlist {
	std::vector<hmr_gpu_cmd> cmds;
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	// branches > 0 of a captured list run on side streams forked from / joined to the context's stream
	std::vector<hipStream_t> side;
	std::vector<hipEvent_t> join;
	hipEvent_t fork = nullptr;
};

static int run_one(hmr_gpu_ctx *ctx, const hmr_gpu_cmd &c)
{
	const hmr_gpu_job *jobs = (const hmr_gpu_job *)c.jobs;
	const int16_t *a = (const int16_t *)c.a, *b = (const int16_t *)c.b;
	int16_t *o = (int16_t *)c.c;
	switch (c.op) {
	case HMR_GPU_OP_SAD: return hmr_gpu_sad_batch(ctx, jobs, c.njobs, c.size, a, b, (uint32_t *)c.out);
	case HMR_GPU_OP_SSD16B: return hmr_gpu_ssd16b_batch(ctx, jobs, c.njobs, c.size, a, b, (uint32_t *)c.out);
	case HMR_GPU_OP_PREDICT: return hmr_gpu_predict_batch(ctx, jobs, c.njobs, c.size, a, b, o);
	case HMR_GPU_OP_RECONST: return hmr_gpu_reconst_batch(ctx, jobs, c.njobs, c.size, a, b, o);
	case HMR_GPU_OP_COPY: return hmr_gpu_copy_batch(ctx, jobs, c.njobs, c.size, c.a, c.c);
	case HMR_GPU_OP_VARIANCE: return hmr_gpu_modified_variance_batch(ctx, jobs, c.njobs, c.size, a, (uint32_t *)c.out);
	case HMR_GPU_OP_INTRA_PRED: return hmr_gpu_intra_pred_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_INTRA_REFS: return hmr_gpu_intra_refs_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_INTERPOLATE: return hmr_gpu_interpolate_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_WAVG: return hmr_gpu_weighted_average_batch(ctx, jobs, c.njobs, a, b, o);
	case HMR_GPU_OP_TRANSFORM: return hmr_gpu_transform_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_ITRANSFORM: return hmr_gpu_itransform_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_QUANT: return hmr_gpu_quant_batch(ctx, jobs, c.njobs, c.size, a, o, (int16_t *)c.b, (int32_t *)c.out);
	case HMR_GPU_OP_INV_QUANT: return hmr_gpu_inv_quant_batch(ctx, jobs, c.njobs, c.size, a, o);
	case HMR_GPU_OP_MC: return hmr_gpu_mc_batch(ctx, jobs, c.njobs, c.size, c.p[0], a, o);   /* size = flags, p[0] = is_bi */
	case HMR_GPU_OP_ME:
		return hmr_gpu_motion_estimation_batch(ctx, (const hmr_gpu_me_job *)c.jobs, c.njobs, c.size, a, b, c.p[0], c.p[1], c.p[2], c.p[3], (hmr_gpu_me_result *)c.out);
	case HMR_GPU_OP_EDGE_FLAGS: return hmr_gpu_edge_flags_frame(ctx, (const uint8_t *)c.a, (const uint8_t *)c.b, c.p[0], c.p[1], c.p[2], (uint8_t *)c.c);
	case HMR_GPU_OP_DEBLOCK: return hmr_gpu_deblock_frame(ctx, (const hmr_gpu_frame *)c.a, (const hmr_gpu_units *)c.b, c.p[0], c.p[1], c.p[2], c.p[3], nullptr, nullptr);
	case HMR_GPU_OP_SAO_STATS: return hmr_gpu_sao_stats_frame(ctx, (const hmr_gpu_frame *)c.a, (const hmr_gpu_frame *)c.b, (int32_t *)c.out);
	case HMR_GPU_OP_SAO_APPLY: return hmr_gpu_sao_apply_frame(ctx, (const hmr_gpu_frame *)c.a, (const hmr_gpu_frame *)c.b, (const int32_t *)c.c);
	case HMR_GPU_OP_PAD: return hmr_gpu_pad_frame(ctx, (const hmr_gpu_frame *)c.a, c.p[0], c.p[1]);
	case HMR_GPU_OP_TU_CHAIN:
		return hmr_gpu_tu_chain_batch(ctx, (const hmr_gpu_tu_job *)c.jobs, c.njobs, c.size, a, b, o, (int16_t *)c.p64[0], (uint32_t *)c.out, (int32_t *)c.p64[1]);
	case HMR_GPU_OP_INTRA_SEARCH:
		return hmr_gpu_intra_search_batch(ctx, (const hmr_gpu_intra_job *)c.jobs, c.njobs, c.size, a, b, o, (hmr_gpu_intra_result *)c.out);
	case HMR_GPU_OP_PIXEL_MULTI: return hmr_gpu_pixel_multi(ctx, c.size, (const hmr_gpu_segment *)c.jobs, c.njobs, a, b, o);
	case HMR_GPU_OP_SAO_OFFSETS:
		return hmr_gpu_sao_offsets_frame(ctx, (const int32_t *)c.a, c.njobs, (const double *)c.b, (int32_t *)c.c, (int32_t *)c.out, (int64_t *)c.p64[0]);
	case HMR_GPU_OP_CHROMA_SEARCH:
		return hmr_gpu_chroma_search_batch(ctx, (const hmr_gpu_chroma_job *)c.jobs, c.njobs, c.size, a, b, (const hmr_gpu_intra_result *)c.p64[0],
						   (hmr_gpu_intra_result *)c.out);
	case HMR_GPU_OP_TU_MULTI:
		return hmr_gpu_tu_chain_multi(ctx, (const hmr_gpu_tu_segment *)c.jobs, c.njobs, a, b, (int16_t *)c.p64[0], o, (int16_t *)c.p64[0]);
	case HMR_GPU_OP_TREE_DECIDE:
		return hmr_gpu_tree_decide_batch(ctx, (const hmr_gpu_tree_job *)c.jobs, c.njobs, (const uint32_t *)c.a, (const int32_t *)c.b, o, (int16_t *)c.p64[0],
						 (hmr_gpu_tree_result *)c.out);
	case HMR_GPU_OP_INTRA_TU_CHAIN:
		if (c.p[0] > 1)
			return hmr_gpu_intra_tu_chain_rounds_batch(ctx, (const hmr_gpu_itu_job *)c.jobs, c.njobs, c.p[0], c.size, a, b, (int16_t *)c.p64[0], o, (int16_t *)c.p64[0],
								   (uint32_t *)c.out, (int32_t *)c.p64[1], (const hmr_gpu_intra_result *)c.p64[2]);
		if (c.p64[2])
			return hmr_gpu_intra_tu_chain_modes_batch(ctx, (const hmr_gpu_itu_job *)c.jobs, c.njobs, c.size, a, b, (int16_t *)c.p64[0], o, (int16_t *)c.p64[0],
								  (uint32_t *)c.out, (int32_t *)c.p64[1], (const hmr_gpu_intra_result *)c.p64[2]);
		return hmr_gpu_intra_tu_chain_batch(ctx, (const hmr_gpu_itu_job *)c.jobs, c.njobs, c.size, a, b, (int16_t *)c.p64[0], o, (int16_t *)c.p64[0], (uint32_t *)c.out,
						    (int32_t *)c.p64[1]);
	case HMR_GPU_OP_INTER_TU_CHAIN:
		return hmr_gpu_inter_tu_chain_batch(ctx, (const hmr_gpu_inter_tu_job *)c.jobs, c.njobs, c.size, a, b, o, (int16_t *)c.p64[0], (uint32_t *)c.out, (int32_t *)c.p64[1]);
	default: hmr_set_error("command list: unknown op %d", c.op); return HMR_GPU_ERR_ARG;
	}
}

extern "C" int hmr_gpu_cmdlist_create(hmr_gpu_ctx *ctx, const hmr_gpu_cmd *cmds, int n, hmr_gpu_cmdlist **out)
{
	(void)ctx;
	if (!cmds || n <= 0 || !out) return HMR_GPU_ERR_ARG;
	hmr_gpu_cmdlist *l = new hmr_gpu_cmdlist;
	l->cmds.assign(cmds, cmds + n);
	*out = l;
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_cmdlist_run(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *l, void **event_pairs)
{
	for (size_t i = 0; i < l->cmds.size(); i++) {
		if (event_pairs) HIP_TRY(hipEventRecord((hipEvent_t)event_pairs[2 * i], ctx->stream));
		const int rc = run_one(ctx, l->cmds[i]);
		if (rc != HMR_GPU_OK) return rc;
		if (event_pairs) HIP_TRY(hipEventRecord((hipEvent_t)event_pairs[2 * i + 1], ctx->stream));
	}
	return HMR_GPU_OK;
}

// (event-record nodes inside a captured graph cannot be read back with hipEventElapsedTime on ROCm 7.2 - "invalid resource
// handle" - so per-kernel timing uses hmr_gpu_cmdlist_run with event pairs instead)
extern "C" int hmr_gpu_cmdlist_capture(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *l)
{
	if (l->exec) return HMR_GPU_OK;
	int nbranch = 0;
	for (const hmr_gpu_cmd &c : l->cmds) {
		if (c.branch < 0 || c.branch > HMR_GPU_MAX_BRANCHES) { hmr_set_error("command list: branch %d out of range", c.branch); return HMR_GPU_ERR_ARG; }
		nbranch = c.branch > nbranch ? c.branch : nbranch;
	}
	while ((int)l->side.size() < nbranch) {
		hipStream_t s;
		hipEvent_t e;
		HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		l->side.push_back(s);
		l->join.push_back(e);
	}
	if (nbranch && !l->fork) HIP_TRY(hipEventCreateWithFlags(&l->fork, hipEventDisableTiming));
	hipStream_t main_stream = ctx->stream;
	HIP_TRY(hipStreamBeginCapture(main_stream, hipStreamCaptureModeThreadLocal));
	int rc = HMR_GPU_OK;
	hipError_t fe = hipSuccess;
	if (nbranch) {      // fork: every side stream joins the capture by waiting on an event of the origin stream
		fe = hipEventRecord(l->fork, main_stream);
		for (int b = 0; b < nbranch && fe == hipSuccess; b++) fe = hipStreamWaitEvent(l->side[b], l->fork, 0);
	}
	for (size_t i = 0; i < l->cmds.size() && rc == HMR_GPU_OK && fe == hipSuccess; i++) {
		ctx->stream = l->cmds[i].branch ? l->side[l->cmds[i].branch - 1] : main_stream;
		rc = run_one(ctx, l->cmds[i]);
	}
	ctx->stream = main_stream;
	for (int b = 0; b < nbranch && fe == hipSuccess; b++) {   // join
		fe = hipEventRecord(l->join[b], l->side[b]);
		if (fe == hipSuccess) fe = hipStreamWaitEvent(main_stream, l->join[b], 0);
	}
	hipError_t e = hipStreamEndCapture(main_stream, &l->graph);
	if (rc != HMR_GPU_OK) return rc;
	if (fe != hipSuccess) { hmr_set_error("command list fork/join: %s", hipGetErrorString(fe)); return HMR_GPU_ERR_HIP; }
	if (e != hipSuccess) { hmr_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return HMR_GPU_ERR_HIP; }
	HIP_TRY(hipGraphInstantiate(&l->exec, l->graph, nullptr, nullptr, 0));
	return HMR_GPU_OK;
}

extern "C" int hmr_gpu_cmdlist_replay(hmr_gpu_ctx *ctx, hmr_gpu_cmdlist *l)
{
	if (!l->exec) {
		const int rc = hmr_gpu_cmdlist_capture(ctx, l);
		if (rc != HMR_GPU_OK) return rc;
	}
	HIP_TRY(hipGraphLaunch(l->exec, ctx->stream));
	return HMR_GPU_OK;
}

extern "C" void hmr_gpu_cmdlist_destroy(hmr_gpu_cmdlist *l)
{
	if (!l) return;
	if (l->exec) (void)hipGraphExecDestroy(l->exec);
	if (l->graph) (void)hipGraphDestroy(l->graph);
	for (hipStream_t st : l->side) (void)hipStreamDestroy(st);
	for (hipEvent_t ev : l->join) (void)hipEventDestroy(ev);
	if (l->fork) (void)hipEventDestroy(l->fork);
	delete l;
}
