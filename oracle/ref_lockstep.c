/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * Lockstep driver over the reference's public C API (homer_hevc_enc_api.h:169-174):
 * feed one frame, drain one frame.  The reference's own CLI polls its output
 * queue non-blockingly and loses NAL units, so stream-level
 * golden vectors and the CPU baseline are produced with this driver instead.
 *
 * Built by oracle/Makefile against the reference sources where they lie
 * (/root/reference/src/homer_lib) into oracle/_ref/ref_lockstep; nothing of
 * the reference is copied into this repository.
 *
 * usage: ref_lockstep in.yuv out.265|- W H frames key=value ...
 *   keys: gop_size num_b intra_period qp bitrate_mode bitrate wpp engines sao
 *         perf rd force_intra intra_tr inter_tr recon=path
 * prints one line:  LOCKSTEP frames=N seconds=S fps=F bytes=B
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>
#include "homer_hevc_enc_api.h"

/* hook for oracle/ref_swap.c: called right after HOMER_enc_init, before HOMER_SETCFG */
__attribute__((weak)) void lockstep_post_init(void *handle) { (void)handle; }
/* hook for oracle/ref_ctudump.c (engine turnstile): every input frame has been handed to the encoder */
__attribute__((weak)) void lockstep_all_fed(int frames) { (void)frames; }

static double now(void)
{
	struct timeval tv;
	gettimeofday(&tv, 0);
	return tv.tv_sec + tv.tv_usec * 1e-6;
}

static int drain_one(void *h, encoder_in_out_t *rec, encoder_in_out_t *os, FILE *fo, FILE *frec, int W, int H, long *bytes)
{
	nalu_t *nal[8];
	unsigned nn = 0;
	HOMER_enc_get_coded_frame(h, rec, nal, &nn);
	if (!nn)
		return 0;
	HOMER_enc_write_annex_b_output(nal, nn, os);
	if (fo)
		fwrite(os->stream.streams[0], 1, os->stream.data_size[0], fo);
	*bytes += os->stream.data_size[0];
	if (frec) {
		fwrite(rec->stream.streams[0], 1, (size_t)W * H, frec);
		fwrite(rec->stream.streams[1], 1, (size_t)W * H / 4, frec);
		fwrite(rec->stream.streams[2], 1, (size_t)W * H / 4, frec);
	}
	return 1;
}

int main(int argc, char **argv)
{
	if (argc < 6) {
		fprintf(stderr, "usage: %s in.yuv out.265|- W H frames [key=value ...]\n", argv[0]);
		return 2;
	}
	const char *in = argv[1], *out = argv[2], *recpath = NULL;
	int W = atoi(argv[3]), H = atoi(argv[4]), N = atoi(argv[5]);
	int force_intra = 0, i;
	HVENC_Cfg c;
	memset(&c, 0, sizeof c);
	c.size = sizeof c;
	c.width = W; c.height = H; c.profile = PROFILE_MAIN;
	/* BASELINE.json configs[1] (SURVEY.md §8-d "cfg2") */
	c.gop_size = 1; c.num_b = 0; c.intra_period = 100; c.qp = 32;
	c.bitrate_mode = BR_FIXED_QP; c.bitrate = 20000;
	c.wfpp_num_threads = 1; c.wfpp_enable = 1; c.num_enc_engines = 1;
	c.sample_adaptive_offset = 1; c.performance_mode = 2; c.rd_mode = 2;
	c.max_intra_tr_depth = 2; c.max_inter_tr_depth = 1;
	c.motion_estimation_precision = QUARTER_PEL; c.frame_rate = 25;
	c.num_ref_frames = 1; c.cu_size = 64; c.max_pred_partition_depth = 4;
	c.sign_hiding = 1; c.chroma_qp_offset = 2; c.reinit_gop_on_scene_change = 1;
	for (i = 6; i < argc; i++) {
		char *eq = strchr(argv[i], '=');
		if (!eq) continue;
		*eq = 0;
		const char *k = argv[i], *v = eq + 1;
		if (!strcmp(k, "gop_size")) c.gop_size = atoi(v);
		else if (!strcmp(k, "num_b")) c.num_b = atoi(v);
		else if (!strcmp(k, "intra_period")) c.intra_period = atoi(v);
		else if (!strcmp(k, "qp")) c.qp = atoi(v);
		else if (!strcmp(k, "bitrate_mode")) c.bitrate_mode = atoi(v);
		else if (!strcmp(k, "bitrate")) c.bitrate = atoi(v);
		else if (!strcmp(k, "wpp")) c.wfpp_num_threads = atoi(v);
		else if (!strcmp(k, "engines")) c.num_enc_engines = atoi(v);
		else if (!strcmp(k, "sao")) c.sample_adaptive_offset = atoi(v);
		else if (!strcmp(k, "perf")) c.performance_mode = atoi(v);
		else if (!strcmp(k, "rd")) c.rd_mode = atoi(v);
		else if (!strcmp(k, "force_intra")) force_intra = atoi(v);
		else if (!strcmp(k, "intra_tr")) c.max_intra_tr_depth = atoi(v);
		else if (!strcmp(k, "inter_tr")) c.max_inter_tr_depth = atoi(v);
		else if (!strcmp(k, "me")) c.motion_estimation_precision = atoi(v);      /* 0 PEL, 1 HALF_PEL, 2 QUARTER_PEL */
		else if (!strcmp(k, "cqo")) c.chroma_qp_offset = atoi(v);
		else if (!strcmp(k, "sign_hiding")) c.sign_hiding = atoi(v);
		else if (!strcmp(k, "recon")) recpath = v;
		else { fprintf(stderr, "unknown key %s\n", k); return 2; }
	}
	c.vbv_size = c.bitrate;
	c.vbv_init = (int)(c.bitrate * 0.35);
	c.wfpp_enable = c.wfpp_num_threads > 0;

	/* the library prints a banner and per-frame traces on stdout: keep ours on stderr+last line */
	void *h = HOMER_enc_init();
	lockstep_post_init(h);
	FILE *fi = fopen(in, "rb");
	FILE *fo = strcmp(out, "-") ? fopen(out, "wb") : NULL;
	FILE *frec = recpath ? fopen(recpath, "wb") : NULL;
	if (!fi) { fprintf(stderr, "cannot open %s\n", in); return 1; }
	if (!HOMER_enc_control(h, HOMER_SETCFG, &c)) { fprintf(stderr, "SETCFG failed\n"); return 1; }

	unsigned char *y = malloc((size_t)W * H), *u = malloc((size_t)W * H / 4), *v = malloc((size_t)W * H / 4);
	encoder_in_out_t inf, os, rec;
	memset(&inf, 0, sizeof inf); memset(&os, 0, sizeof os); memset(&rec, 0, sizeof rec);
	os.stream.streams[0] = malloc(0x4000000);
	if (frec) {
		rec.stream.streams[0] = malloc((size_t)W * H);
		rec.stream.streams[1] = malloc((size_t)W * H / 4);
		rec.stream.streams[2] = malloc((size_t)W * H / 4);
	}
	int got = 0, fed = 0, lag = c.num_enc_engines - 1;
	long bytes = 0;
	double t0 = now();
	while (fed < N && fread(y, 1, (size_t)W * H, fi) == (size_t)W * H && fread(u, 1, (size_t)W * H / 4, fi) == (size_t)W * H / 4 &&
	       fread(v, 1, (size_t)W * H / 4, fi) == (size_t)W * H / 4) {
		inf.stream.streams[0] = y; inf.stream.streams[1] = u; inf.stream.streams[2] = v;
		inf.stream.data_stride[0] = W; inf.stream.data_stride[1] = inf.stream.data_stride[2] = W / 2;
		inf.pts = fed;
		inf.image_type = force_intra ? IMAGE_I : IMAGE_AUTO;
		HOMER_enc_encode(h, &inf);
		fed++;
		while (got < fed - lag) {
			if (drain_one(h, &rec, &os, fo, frec, W, H, &bytes)) got++;
			else usleep(100);
		}
	}
	lockstep_all_fed(fed);
	while (got < fed) {
		if (drain_one(h, &rec, &os, fo, frec, W, H, &bytes)) got++;
		else usleep(100);
	}
	double t1 = now();
	if (fo) fclose(fo);
	if (frec) fclose(frec);
	fflush(stdout);
	printf("\nLOCKSTEP frames=%d seconds=%.4f fps=%.4f bytes=%ld\n", got, t1 - t0, got / (t1 - t0), bytes);
	fflush(stdout);
	_exit(0); /* engine threads are parked on semaphores; skip the reference's close path */
}
