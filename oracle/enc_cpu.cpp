/*
 * TEST INFRASTRUCTURE - not part of the product path.
 *
 * The CTU encoder core (homerhevc_amd/csrc/enc/, the code the gfx950 kernels are compiled from) instantiated with a one-lane
 * group and driven CTU by CTU in raster order - the order of the reference with wfpp_num_threads = 1.  It exists so that the
 * decision logic can be diffed against the compiled reference (oracle/_ref/ref_ctudump) in the build container, where there
 * is no GPU, and so that the CPU tests can check the host logic.  Only tests/ and tools/ load this library; libhomer_gpu.so
 * never contains this instantiation.
 *
 * Records use the layout of oracle/ref_ctudump.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define HENC_TRACE_ENABLE 1
#include "../homerhevc_amd/csrc/enc/enc_ctu.h"
#include "../homerhevc_amd/csrc/enc/enc_host.h"

extern "C" FILE *henc_trace_file = nullptr;
const DevTables *hmr_host_tables();

using namespace henc;

namespace {

constexpr int REC_BYTES = 32 + 3 * 256 + 2 * 256 + 9 * 256 + 256 + 256 + 2048 + 2048 + 6144 * 2 + 6144 * 2 + 2 * 5 * 256;

struct Cpu {
	HostCfg cfg;
	Seq seq;
	HostState st;
	FrameCtx f;
	Geo geo[NNODES];
	std::vector<CtuInfo> ctus;
	Work *w;
	std::vector<int16_t> src[3], pic[2][3], coeff;
	std::vector<uint8_t> records;
	int cur = 0;        // picture under reconstruction: pic[cur], reference: pic[cur ^ 1]
	uint32_t acc_dist = 0;
	uint32_t intra_parts = 0, total_parts = 0;
};

int16_t *plane0(Cpu &c, int which, int comp)
{
	const Seq &s = c.seq;
	const int st = comp ? s.stride_c : s.stride_y, m = comp ? s.margin_c : s.margin_y;
	return c.pic[which][comp].data() + (size_t)m * st + m;
}

void pad_plane(int16_t *p, int stride, int w, int h, int m)
{
	for (int y = 0; y < h; y++) {
		for (int x = 1; x <= m; x++) {
			p[y * stride - x] = p[y * stride];
			p[y * stride + w - 1 + x] = p[y * stride + w - 1];
		}
	}
	for (int y = 1; y <= m; y++) {
		memcpy(p - y * stride - m, p - m, sizeof(int16_t) * (w + 2 * m));
		memcpy(p + (h - 1 + y) * stride - m, p + (h - 1) * stride - m, sizeof(int16_t) * (w + 2 * m));
	}
}

void make_record(Cpu &c, int n, Enc &e)
{
	uint8_t *o = c.records.data() + (size_t)n * REC_BYTES;
	const CtuInfo &ci = c.ctus[n];
	const Work &w = *c.w;
	int32_t hdr[8] = {0x43545544, c.f.num_encoded_frames, n, c.f.slice_type, (int32_t)ci.nodes[0].cost, (int32_t)ci.nodes[0].distortion, (int32_t)ci.nodes[0].sum,
			  c.f.is_scene_change};
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
	memcpy(o, w.tq_y[0], 8192); o += 8192;
	memcpy(o, w.tq_c[0][0], 2048); o += 2048;
	memcpy(o, w.tq_c[0][1], 2048); o += 2048;
	for (int comp = 0; comp < 3; comp++) {
		const int nn = comp ? 32 : 64;
		const int16_t *d = comp ? w.dec_c[0][comp - 1] + DEC_ORG_C : w.dec_y[0] + DEC_ORG_Y;
		for (int y = 0; y < nn; y++) { memcpy(o, d + y * dec_stride(comp), nn * 2); o += nn * 2; }
	}
	for (int k = 0; k < 2; k++)
		for (int d = 0; d < 5; d++) { memcpy(o, w.intra_mode_buffs[k][d], 256); o += 256; }
	(void)e;
}

}  // namespace

extern "C" {

int henc_cpu_record_bytes(void) { return REC_BYTES; }

void *henc_cpu_create(const HostCfg *cfg)
{
	Cpu *c = new Cpu;
	c->cfg = *cfg;
	const char *why;
	if (!make_seq(*cfg, c->seq, &why)) {
		fprintf(stderr, "henc_cpu_create: unsupported configuration: %s\n", why);
		delete c;
		return nullptr;
	}
	make_geo(c->geo);
	const Seq &s = c->seq;
	c->ctus.resize(s.nctu);
	memset(c->ctus.data(), 0, sizeof(CtuInfo) * s.nctu);
	for (auto &ci : c->ctus) memset(ci.mv_ref_idx, -1, sizeof ci.mv_ref_idx);
	c->w = (Work *)calloc(1, sizeof(Work));
	c->src[0].assign((size_t)s.src_stride_y * s.height, 0);
	c->src[1].assign((size_t)s.src_stride_c * s.height / 2, 0);
	c->src[2].assign((size_t)s.src_stride_c * s.height / 2, 0);
	for (int k = 0; k < 2; k++) {
		c->pic[k][0].assign((size_t)s.stride_y * (s.height + 2 * s.margin_y), 0);
		c->pic[k][1].assign((size_t)s.stride_c * (s.height / 2 + 2 * s.margin_c), 0);
		c->pic[k][2].assign((size_t)s.stride_c * (s.height / 2 + 2 * s.margin_c), 0);
	}
	c->coeff.assign((size_t)s.nctu * 6144, 0);
	c->records.assign((size_t)s.nctu * REC_BYTES, 0);
	return c;
}

void henc_cpu_destroy(void *h)
{
	Cpu *c = (Cpu *)h;
	free(c->w);
	delete c;
}

void henc_cpu_set_trace(const char *path)
{
	if (henc_trace_file) fclose(henc_trace_file);
	henc_trace_file = path && *path ? fopen(path, "w") : nullptr;
}

// CTU decisions of one frame in raster order.  ref_* (8-bit, width x height): the previous frame's final reconstruction as the
// reference gives it (teacher forcing until the loop filters run here too); avg_dist < 0 keeps the encoder's own value.
int henc_cpu_frame_ctus(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int image_type, const uint8_t *ref_y, const uint8_t *ref_u,
			const uint8_t *ref_v, double avg_dist, int first_ctu, int last_ctu)
{
	Cpu &c = *(Cpu *)h;
	const Seq &s = c.seq;
	const uint8_t *in[3] = {y, u, v}, *rin[3] = {ref_y, ref_u, ref_v};
	if (first_ctu == 0) {
		c.cur ^= 1;
		begin_frame(s, c.st, image_type, c.f);
		if (avg_dist >= 0) c.f.avg_dist = avg_dist;
		c.acc_dist = 0;
		c.intra_parts = c.total_parts = 0;
		for (int comp = 0; comp < 3; comp++) {
			const int w = comp ? s.width / 2 : s.width, hh = comp ? s.height / 2 : s.height, ss = comp ? s.src_stride_c : s.src_stride_y;
			for (int r = 0; r < hh; r++)
				for (int x = 0; x < w; x++) c.src[comp][(size_t)r * ss + x] = in[comp][(size_t)r * w + x];
			if (rin[comp]) {
				int16_t *p = plane0(c, c.cur ^ 1, comp);
				const int rs = comp ? s.stride_c : s.stride_y;
				for (int r = 0; r < hh; r++)
					for (int x = 0; x < w; x++) p[(size_t)r * rs + x] = rin[comp][(size_t)r * w + x];
				pad_plane(p, rs, w, hh, comp ? s.margin_c : s.margin_y);
			}
			c.f.src[comp] = c.src[comp].data();
			c.f.ref[comp] = plane0(c, c.cur ^ 1, comp);
			c.f.rec[comp] = plane0(c, c.cur, comp);
		}
	}
	Enc e;
	memset(&e, 0, sizeof e);
	e.seq = &c.seq;
	e.f = &c.f;
	e.T = hmr_host_tables();
	e.geo = c.geo;
	e.ctus = c.ctus.data();
	e.w = c.w;
	CpuGrp g;
	if (last_ctu < 0 || last_ctu > s.nctu) last_ctu = s.nctu;
	for (int n = first_ctu; n < last_ctu; n++) {
		e.coeff = c.coeff.data() + (size_t)n * 6144;
		e.total_intra_partitions = c.intra_parts;
		e.total_partitions = c.total_parts;
		encode_ctu(g, e, n);
		c.intra_parts += c.ctus[n].intra_parts;
		c.total_parts += NPART;
		c.acc_dist += c.ctus[n].distortion;
		make_record(c, n, e);
	}
	if (last_ctu == s.nctu) end_frame(s, c.st, c.f, c.acc_dist);
	return c.f.slice_type;
}

const uint8_t *henc_cpu_records(void *h) { return ((Cpu *)h)->records.data(); }
double henc_cpu_avg_dist(void *h) { return ((Cpu *)h)->st.avg_dist; }

}  // extern "C"
