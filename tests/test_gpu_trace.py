"""The BATCHED GPU entries on real-encode data: every group of tests/golden/trace_200x136.npz (inputs and outputs of the reference's block
drivers logged during a real encode) is issued as ONE launch and compared with what the reference produced."""
import ctypes as C
import sys

import numpy as np
import pytest

import libs
import trace_cases as tc

sys.path.insert(0, libs.ROOT)
import gpu_abi as gh  # noqa: E402

pytestmark = pytest.mark.gpu
VP = C.c_void_p


class Dev:
    def __init__(self, gpu):
        self.gpu, self.ctx, self.bufs = gpu, VP(), []
        assert gpu.hmr_gpu_create(C.byref(self.ctx), 0, None) == 0

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        p = VP()
        assert self.gpu.hmr_gpu_malloc(self.ctx, C.byref(p), C.c_size_t(max(arr.nbytes, 16))) == 0
        assert self.gpu.hmr_gpu_upload(self.ctx, p, VP(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0
        self.bufs.append(p)
        return p

    def down(self, p, shape, dtype):
        out = np.zeros(shape, dtype)
        assert self.gpu.hmr_gpu_download(self.ctx, VP(out.ctypes.data), p, C.c_size_t(out.nbytes)) == 0
        return out

    def close(self):
        for p in self.bufs:
            self.gpu.hmr_gpu_free(self.ctx, p)
        self.gpu.hmr_gpu_destroy(self.ctx)


@pytest.fixture()
def dev():
    d = Dev(libs.load_gpu())
    yield d
    d.close()


def test_trace_groups_as_single_launches(dev):
    gpu = dev.gpu
    seen = set()
    groups = tc.load()
    planes = tc.ref_planes(groups)
    for g in groups:
        cnt, kind, hdr, dbl, b = g["count"], g["kind"], g["hdr"], g["dbl"], g["blobs"]
        if kind == "ref_plane":
            continue
        seen.add(kind)
        idx = np.arange(cnt, dtype=np.int64)
        if kind == "inter_tu":
            n = g["w"]; e = n * n
            arena = np.concatenate([b[0].ravel(), b[1].ravel(), np.full(2 * cnt * e, 0x1234, np.int16)])
            jb = np.zeros(cnt, gh.INTER_TU_JOB_DTYPE)
            jb["orig_off"] = idx * e; jb["pred_off"] = cnt * e + idx * e; jb["lev_off"] = 2 * cnt * e + idx * e; jb["rec_off"] = 3 * cnt * e + idx * e
            jb["orig_stride"] = jb["pred_stride"] = jb["rec_stride"] = n
            jb["p0"] = hdr[:, 2] | (hdr[:, 1] << 2) | (hdr[:, 3] << 5) | (hdr[:, 4] << 6); jb["p1"] = hdr[:, 5] | (hdr[:, 6] << 8)
            jb["weight"] = dbl[:, 0]; jb["zero_thr"] = dbl[:, 1]
            d_a, d_ssd, d_ac = dev.up(arena), dev.up(np.zeros(cnt, np.uint32)), dev.up(np.zeros(cnt, np.int32))
            assert gpu.hmr_gpu_inter_tu_chain_batch(dev.ctx, dev.up(jb), cnt, n, d_a, d_a, d_a, d_a, d_ssd, d_ac) == 0
            out = dev.down(d_a, arena.shape, np.int16)
            assert np.array_equal(out[2 * cnt * e:3 * cnt * e].reshape(cnt, e), b[2]), (g["tag"], "levels")
            assert np.array_equal(out[3 * cnt * e:].reshape(cnt, e), b[3]), (g["tag"], "recon")
            assert np.array_equal(dev.down(d_ac, cnt, np.int32), hdr[:, 7]) and np.array_equal(dev.down(d_ssd, cnt, np.uint32).view(np.int32), hdr[:, 8]), g["tag"]
        elif kind in ("intra_tu", "intra_search"):
            n = g["w"]; e = n * n; s = 2 * n + 1
            tiles = np.stack([tc.lshape_tile(b[1][i], n) for i in range(cnt)]).ravel()
            flags = hdr[:, 1] | (hdr[:, 2] << 1) | (hdr[:, 3] << 2) | (hdr[:, 4] << 3) | (hdr[:, 7] << 5)
            if kind == "intra_tu":
                arena = np.concatenate([b[0].ravel(), tiles, np.full(3 * cnt * e, 0x1234, np.int16)])
                o0 = cnt * e + tiles.size
                jb = np.zeros(cnt, gh.ITU_JOB_DTYPE)
                jb["orig_off"] = idx * e; jb["orig_stride"] = n
                jb["dec_off"] = cnt * e + idx * s * s; jb["dec_stride"] = s
                jb["pred_off"] = o0 + idx * e; jb["lev_off"] = o0 + cnt * e + idx * e; jb["rec_off"] = o0 + 2 * cnt * e + idx * e
                jb["pred_stride"] = jb["rec_stride"] = n
                jb["flags"] = flags | (hdr[:, 8] << 6) | (1 << 7); jb["sizes"] = hdr[:, 5] | (hdr[:, 6] << 16); jb["mode"] = hdr[:, 9]
                jb["p0"] = hdr[:, 10] | (1 << 4) | (hdr[:, 11] << 5) | (hdr[:, 12] << 6) | ((1 if n == 4 else 0) << 7); jb["p1"] = hdr[:, 13] | (hdr[:, 14] << 8)
                d_a, d_ssd, d_ac = dev.up(arena), dev.up(np.zeros(cnt, np.uint32)), dev.up(np.zeros(cnt, np.int32))
                assert gpu.hmr_gpu_intra_tu_chain_batch(dev.ctx, dev.up(jb), cnt, n, d_a, d_a, d_a, d_a, d_a, d_ssd, d_ac) == 0
                out = dev.down(d_a, arena.shape, np.int16)[o0:].reshape(3, cnt, e)
                for k, name in enumerate(("pred", "levels", "recon")):
                    assert np.array_equal(out[k], b[2 + k]), (g["tag"], name)
                assert np.array_equal(dev.down(d_ac, cnt, np.int32), hdr[:, 15]) and np.array_equal(dev.down(d_ssd, cnt, np.uint32).view(np.int32), hdr[:, 16]), g["tag"]
            else:
                a1 = 4 * n + 4
                arena = np.concatenate([b[0].ravel(), tiles, np.full(cnt * (2 * a1 + e), 0x1234, np.int16)])
                o0 = cnt * e + tiles.size
                jb = np.zeros(cnt, gh.INTRA_JOB_DTYPE)
                jb["orig_off"] = idx * e; jb["orig_stride"] = n
                jb["dec_off"] = cnt * e + idx * s * s; jb["dec_stride"] = s
                jb["adi_off"] = o0 + idx * a1; jb["adif_off"] = o0 + cnt * a1 + idx * a1; jb["pred_off"] = o0 + 2 * cnt * a1 + idx * e; jb["pred_stride"] = n
                jb["flags"] = flags; jb["sizes"] = hdr[:, 5] | (hdr[:, 6] << 16)
                jb["preds"] = hdr[:, 8:11]; jb["pred_bits"] = hdr[:, 11:14]; jb["other_bits"] = hdr[:, 14]; jb["sqrt_lambda"] = dbl[:, 0]
                d_a = dev.up(arena)
                d_res = dev.up(np.zeros(cnt, np.dtype([("best", "<i4"), ("bits", "<i4"), ("cost", "<f8")])))
                assert gpu.hmr_gpu_intra_search_batch(dev.ctx, dev.up(jb), cnt, n, d_a, d_a, d_a, d_res) == 0
                out = dev.down(d_a, arena.shape, np.int16)[o0:]
                assert np.array_equal(out[:cnt * a1].reshape(cnt, a1)[:, :4 * n + 1], b[2]), (g["tag"], "adi")
                assert np.array_equal(out[cnt * a1:2 * cnt * a1].reshape(cnt, a1)[:, :4 * n + 1], b[3]), (g["tag"], "adi filtered")
                assert np.array_equal(out[2 * cnt * a1:].reshape(cnt, e), b[4]), (g["tag"], "last prediction")
                res = dev.down(d_res, cnt, np.dtype([("best", "<i4"), ("bits", "<i4"), ("cost", "<f8")]))
                assert np.array_equal(res["best"], hdr[:, 15]) and np.array_equal(res["bits"], hdr[:, 16]) and np.array_equal(res["cost"], dbl[:, 1]), g["tag"]
        elif kind == "me":
            n = g["w"]; e = n * n
            ids = sorted(planes)
            psize = planes[ids[0]].size
            arena = np.concatenate([planes[k] for k in ids] + [b[0].ravel()])
            fw, fh = int(hdr[0, 8]), int(hdr[0, 9]); st = fw + 160
            assert (hdr[:, 8] == fw).all() and (hdr[:, 9] == fh).all() and (hdr[:, 6] == hdr[0, 6]).all() and (hdr[:, 7] == hdr[0, 7]).all()
            jb = np.zeros(cnt, gh.ME_JOB_DTYPE)
            jb["corr"] = dbl[:, 0]
            jb["orig_off"] = len(ids) * psize + idx * e; jb["orig_stride"] = n
            jb["ref_off"] = np.array([ids.index(int(p)) for p in hdr[:, 1]]) * psize + (80 + hdr[:, 3]) * st + 80 + hdr[:, 2]; jb["ref_stride"] = st
            jb["gx"] = hdr[:, 2]; jb["gy"] = hdr[:, 3]; jb["init_x"] = hdr[:, 4]; jb["init_y"] = hdr[:, 5]
            jb["n_amvp"] = hdr[:, 11]; jb["n_search"] = hdr[:, 12]
            jb["amvp"] = hdr[:, 13:17].reshape(cnt, 2, 2); jb["search"] = hdr[:, 17:27].reshape(cnt, 5, 2); jb["action"] = hdr[:, 10]
            d_a, d_out = dev.up(arena), dev.up(np.zeros((cnt, 5), np.int32))
            assert gpu.hmr_gpu_motion_estimation_batch(dev.ctx, dev.up(jb), cnt, n, d_a, d_a, int(hdr[0, 6]), int(hdr[0, 7]), fw, fh, d_out) == 0
            assert np.array_equal(dev.down(d_out, (cnt, 5), np.int32), hdr[:, 27:32]), g["tag"]
        else:
            w, h = g["w"], g["h"]; ws = (h + 8) * (w + 8)
            arena = np.concatenate([b[0].ravel(), np.full(cnt * w * h, 0x1234, np.int16)])
            for luma in (1, 0):
                for bi in (0, 1):
                    sel = np.flatnonzero((hdr[:, 0] == luma) & (hdr[:, 5] == bi))
                    if not sel.size:
                        continue
                    jb = np.zeros(sel.size, gh.JOB_DTYPE)
                    jb["a_off"] = sel * ws + 4 * (w + 8) + 4; jb["a_stride"] = w + 8
                    jb["c_off"] = cnt * ws + sel * w * h; jb["c_stride"] = w
                    jb["w"] = w; jb["h"] = h; jb["p0"] = hdr[sel, 3]; jb["p1"] = hdr[sel, 4]
                    d_a = dev.up(arena)
                    lanes = 4 if w == 4 else 16 if w == 8 else 64
                    assert gpu.hmr_gpu_mc_batch(dev.ctx, dev.up(jb), int(sel.size), luma | (lanes << 8), bi, d_a, d_a) == 0
                    out = dev.down(d_a, arena.shape, np.int16)[cnt * ws:].reshape(cnt, w * h)
                    assert np.array_equal(out[sel], b[1][sel]), (g["tag"], luma, bi)
    assert seen == {"inter_tu", "intra_tu", "intra_search", "mc", "me"}
