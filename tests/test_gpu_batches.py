"""GPU parity of the BATCHED entries (include/homer_gpu.h section 3, 5, 7): hundreds of jobs per launch on one device arena, so
lane packing (several jobs per wavefront), the XCD job partition and the unaligned vector paths are exercised the way bench.py
uses them.  Every job is replayed through the CPU oracle on a host copy of the same arena and the WHOLE arena is compared, so a
stray write outside a job's output block fails too."""
import ctypes as C
import sys

import numpy as np
import pytest

import libs
from kernel_cases import quant_depth

sys.path.insert(0, libs.ROOT)
import gpu_abi as gpu_host  # noqa: E402  (descriptor layouts of include/homer_gpu.h)

pytestmark = pytest.mark.gpu

JOB = gpu_host.JOB_DTYPE
TU_JOB = gpu_host.TU_JOB_DTYPE
ME_JOB = gpu_host.ME_JOB_DTYPE
VP = C.c_void_p
SEED = int(__import__("os").environ.get("HOMER_TEST_SEED", "0"))     # fuzzing: HOMER_TEST_SEED=k reseeds every case
NJ = 611                      # odd on purpose: ragged last wavefront / last XCD chunk
PW, PH = 512, 320             # input planes
SLOT = 80 * 80                # one output slot per job


class Rig:
    """Host arena (int16) mirrored on the device: [pixels | residuals | 14-bit intermediates | out1 slots | out2 slots]."""

    def __init__(self, gpu, rng, nj=NJ):
        self.gpu, self.nj = gpu, nj
        self.ctx = VP()
        assert gpu.hmr_gpu_create(C.byref(self.ctx), 0, None) == 0, gpu.hmr_gpu_last_error()
        self.pix, self.res, self.mid = 0, PW * PH, 2 * PW * PH
        self.out1, self.out2 = 3 * PW * PH, 3 * PW * PH + nj * SLOT
        self.size = 3 * PW * PH + 2 * nj * SLOT
        h = np.full(self.size, 0x1234, np.int16)
        h[self.pix:self.res] = rng.integers(0, 256, PW * PH)
        h[self.res:self.mid] = rng.integers(-300, 301, PW * PH)
        h[self.mid:self.out1] = rng.integers(-8192, 8129, PW * PH)
        self.host = h
        self.dev = self.malloc(self.size * 2)
        self.bufs = [self.dev]

    def malloc(self, nbytes):
        p = VP()
        assert self.gpu.hmr_gpu_malloc(self.ctx, C.byref(p), C.c_size_t(max(nbytes, 4))) == 0
        return p

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.bufs.append(p)
        assert self.gpu.hmr_gpu_upload(self.ctx, p, VP(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0
        return p

    def down(self, p, shape, dtype):
        out = np.zeros(shape, dtype)
        assert self.gpu.hmr_gpu_download(self.ctx, VP(out.ctypes.data), p, C.c_size_t(out.nbytes)) == 0
        return out

    def launch(self, name, *args):
        """upload the arena, run one batched call, return the device arena afterwards"""
        assert self.gpu.hmr_gpu_upload(self.ctx, self.dev, VP(self.host.ctypes.data), C.c_size_t(self.size * 2)) == 0
        rc = getattr(self.gpu, name)(self.ctx, *args)
        assert rc == 0, (name, self.gpu.hmr_gpu_last_error())
        assert self.gpu.hmr_gpu_sync(self.ctx) == 0, (name, self.gpu.hmr_gpu_last_error())
        return self.down(self.dev, self.size, np.int16)

    def block(self, rng, base, bw, bh, margin=0):
        """random element offsets of bw x bh blocks inside an input plane, `margin` samples away from its border"""
        x = rng.integers(margin, PW - bw - margin + 1, self.nj)
        y = rng.integers(margin, PH - bh - margin + 1, self.nj)
        return (base + y * PW + x).astype(np.uint32)

    def slots(self, base):
        return (base + np.arange(self.nj, dtype=np.int64) * SLOT).astype(np.uint32)

    def close(self):
        for p in self.bufs:
            self.gpu.hmr_gpu_free(self.ctx, p)
        self.gpu.hmr_gpu_destroy(self.ctx)


@pytest.fixture(params=[4096, 8, 3], ids=["grid4096", "grid8", "grid3"])
def rig(request):
    """Every test runs with the default launch cap and with caps of 8 and 3 workgroups: the same 611 jobs then take many grid-stride
    iterations per workgroup (8 = one workgroup per XCD chunk, 3 = the plain grid-stride form), which is how the kernels run on
    frame-sized batches."""
    gpu = libs.load_gpu()
    assert gpu.hmr_gpu_set_max_grid(request.param) == 0
    r = Rig(gpu, np.random.default_rng(77 + SEED))
    yield r
    r.close()
    gpu.hmr_gpu_set_max_grid(4096)


def at(arr, off):
    return VP(arr.ctypes.data + int(off) * arr.itemsize)


def same(got, exp, what):
    if not np.array_equal(got, exp):
        bad = np.flatnonzero(got != exp)
        raise AssertionError(f"{what}: {bad.size} elements differ, first at {bad[0]} (got {got[bad[0]]}, expected {exp[bad[0]]})")


@pytest.mark.parametrize("which", ["sad", "ssd16b"])
@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_sad_ssd(rig, oracle, which, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.block(rng, rig.pix, n, n); jb["a_stride"] = PW
    jb["b_off"] = rig.block(rng, rig.res if which == "ssd16b" else rig.pix, n, n); jb["b_stride"] = PW
    if which == "ssd16b":
        jb["b_stride"][::3] = 0
    else:   # a few candidates with negative samples: the packed-unsigned fast path must hand over to the general form
        neg = rng.random(rig.nj) < 0.1
        jb["b_off"] = np.where(neg, rig.block(rng, rig.res, n, n), jb["b_off"])
    d_out = rig.malloc(4 * rig.nj); rig.bufs.append(d_out)
    g = rig.launch(f"hmr_gpu_{which}_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev, d_out)
    same(g, rig.host, "arena untouched")
    f = getattr(oracle, "ora_" + which); f.restype = C.c_uint32
    exp = np.array([f(at(rig.host, j["a_off"]), int(j["a_stride"]), at(rig.host, j["b_off"]), int(j["b_stride"]), n) for j in jb], np.uint32)
    same(rig.down(d_out, rig.nj, np.uint32), exp, which)


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_predict_reconst(rig, oracle, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.block(rng, rig.pix, n, n); jb["a_stride"] = PW
    jb["b_off"] = rig.block(rng, rig.pix, n, n); jb["b_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = np.where(np.arange(rig.nj) % 2, n, 80)
    g = rig.launch("hmr_gpu_predict_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:
        oracle.ora_predict(at(o, j["a_off"]), int(j["a_stride"]), at(o, j["b_off"]), int(j["b_stride"]), at(o, j["c_off"]), int(j["c_stride"]), n)
    same(g, o, "predict")
    jb["b_off"] = rig.block(rng, rig.res, n, n)
    jb["b_stride"][::4] = 0
    g = rig.launch("hmr_gpu_reconst_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:
        oracle.ora_reconst(at(o, j["a_off"]), int(j["a_stride"]), at(o, j["b_off"]), int(j["b_stride"]), at(o, j["c_off"]), int(j["c_stride"]), n)
    same(g, o, "reconst")


@pytest.mark.parametrize("square", [0, 4, 8, 16, 32, 64])
def test_copy(rig, oracle, square):
    rng = np.random.default_rng(square + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    if square:
        jb["w"] = jb["h"] = square
    else:
        jb["w"] = rng.integers(1, 17, rig.nj) * 4; jb["h"] = rng.integers(1, 65, rig.nj)
    x = rng.integers(0, PW - 64, rig.nj); y = rng.integers(0, PH - 64, rig.nj)
    jb["a_off"] = rig.res + y * PW + x; jb["a_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = 80
    g = rig.launch("hmr_gpu_copy_batch", rig.up(jb), rig.nj, square << 8, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:     # the SSE copy rounds the width up to its vector; the contract is the h x w block
        src = o[j["a_off"]:j["a_off"] + (j["h"] - 1) * PW + j["w"]]
        for r in range(j["h"]):
            o[j["c_off"] + r * 80:j["c_off"] + r * 80 + j["w"]] = src[r * PW:r * PW + j["w"]]
    same(g, o, "copy_16_16")
    # the oracle's copy agrees on the block itself
    j = jb[0]
    blk = np.zeros((int(j["h"]), 96), np.int16)
    oracle.ora_copy_16_16(at(rig.host, j["a_off"]), PW, at(blk, 0), 96, int(j["h"]), int(j["w"]))
    same(blk[:, :j["w"]].ravel(), g[j["c_off"]:j["c_off"] + 80 * j["h"]].reshape(-1, 80)[:, :j["w"]].ravel(), "copy vs oracle")


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_modified_variance(rig, oracle, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.block(rng, rig.pix, n, n); jb["a_stride"] = PW
    jb["p0"] = rng.integers(1, 3, rig.nj)
    d_out = rig.malloc(4 * rig.nj); rig.bufs.append(d_out)
    g = rig.launch("hmr_gpu_modified_variance_batch", rig.up(jb), rig.nj, n, rig.dev, d_out)
    same(g, rig.host, "arena untouched")
    oracle.ora_modified_variance.restype = C.c_uint32
    exp = np.array([oracle.ora_modified_variance(at(rig.host, j["a_off"]), n, PW, int(j["p0"])) for j in jb], np.uint32)
    same(rig.down(d_out, rig.nj, np.uint32), exp, "modified_variance")


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_intra_pred_and_refs(rig, oracle, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.pix + rng.integers(0, PW * PH - 4 * n - 1, rig.nj)
    jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = 80
    jb["p0"] = rng.integers(0, 35, rig.nj); jb["p1"] = rng.integers(0, 2, rig.nj)
    g = rig.launch("hmr_gpu_intra_pred_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:
        if j["p0"] == 0:
            oracle.ora_intra_planar(at(o, j["c_off"]), 80, at(o, j["a_off"]), 4 * n + 1, n)
        else:
            oracle.ora_intra_angular(at(o, j["c_off"]), 80, at(o, j["a_off"]), 4 * n + 1, n, int(j["p0"]), int(j["p1"]))
    same(g, o, "intra_pred")

    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.block(rng, rig.pix, 2 * n + 1, 2 * n + 1); jb["a_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["b_off"] = rig.slots(rig.out2)
    avail = rng.integers(0, 16, rig.nj)
    left, top = (avail & 1) | ((avail >> 2) & 1), ((avail >> 1) & 1) | ((avail >> 3) & 1)
    bl, tr = (avail >> 2) & 1, (avail >> 3) & 1
    filt, strong = rng.integers(0, 2, rig.nj), rng.integers(0, 2, rig.nj)
    bl_size = np.where(bl, np.where(rng.random(rig.nj) < 0.5, n, max(n // 2, 4)), 0)
    tr_size = np.where(tr, np.where(rng.random(rig.nj) < 0.5, n, max(n // 2, 4)), 0)
    jb["p0"] = left | (top << 1) | (bl << 2) | (tr << 3) | (filt << 4) | (strong << 5)
    jb["p1"] = bl_size | (tr_size << 16)
    g = rig.launch("hmr_gpu_intra_refs_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev)
    o = rig.host.copy()
    for i, j in enumerate(jb):
        oracle.ora_fill_reference_samples(at(o, j["a_off"]), PW, n, int(left[i]), int(top[i]), int(bl[i]), int(tr[i]), int(bl_size[i]), int(tr_size[i]),
                                          at(o, j["c_off"]))
        if filt[i]:
            oracle.ora_adi_filter(at(o, j["c_off"]), at(o, j["b_off"]), 4 * n + 1, n, int(strong[i]))
    same(g, o, "intra_refs")


@pytest.mark.parametrize("luma", [1, 0])
@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
def test_interpolate(rig, oracle, luma, lanes):
    rng = np.random.default_rng(luma * 100 + lanes + 1000 * SEED)
    taps = 8 if luma else 4
    for first in (1, 0):
        jb = np.zeros(rig.nj, JOB)
        wmax = {4: 8, 8: 12, 16: 17, 32: 33, 64: 65}[lanes]
        jb["w"] = rng.integers(2 if not luma else 4, wmax + 1, rig.nj); jb["h"] = rng.integers(2, wmax + 8, rig.nj)
        jb["a_off"] = rig.block(rng, rig.pix if first else rig.mid, 80, 80, margin=4) ; jb["a_stride"] = PW
        jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = 80
        jb["p0"] = rng.integers(0, 4 if luma else 8, rig.nj)
        vert, last = rng.integers(0, 2, rig.nj), rng.integers(0, 2, rig.nj)
        jb["p1"] = vert | (first << 1) | (last << 2)
        g = rig.launch("hmr_gpu_interpolate_batch", rig.up(jb), rig.nj, luma | (lanes << 8), rig.dev, rig.dev)
        o = rig.host.copy()
        f = oracle.ora_interpolate_luma if luma else oracle.ora_interpolate_chroma
        for i, j in enumerate(jb):
            f(at(o, j["a_off"]), PW, at(o, j["c_off"]), 80, int(j["p0"]), int(j["w"]), int(j["h"]), int(vert[i]), first, int(last[i]))
        same(g, o, f"interpolate taps={taps} first={first}")


def test_weighted_average(rig, oracle):
    rng = np.random.default_rng(5 + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    n = 1 << rng.integers(2, 7, rig.nj)
    jb["w"] = n; jb["h"] = np.where(rng.random(rig.nj) < 0.2, np.maximum(n // 2, 4), n)
    jb["a_off"] = rig.block(rng, rig.mid, 64, 64); jb["a_stride"] = PW
    jb["b_off"] = rig.block(rng, rig.mid, 64, 64); jb["b_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = 80
    g = rig.launch("hmr_gpu_weighted_average_batch", rig.up(jb), rig.nj, rig.dev, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:
        oracle.ora_weighted_average(at(o, j["a_off"]), PW, at(o, j["b_off"]), PW, at(o, j["c_off"]), 80, int(j["h"]), int(j["w"]))
    same(g, o, "weighted_average")


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_transform_quant(rig, oracle, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    is_dst = rng.integers(0, 2, rig.nj) if n == 4 else np.zeros(rig.nj, np.int64)
    jb = np.zeros(rig.nj, JOB)
    jb["a_off"] = rig.block(rng, rig.res, n, n); jb["a_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["p0"] = is_dst
    g = rig.launch("hmr_gpu_transform_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in jb:
        oracle.ora_transform(at(o, j["a_off"]), at(o, j["c_off"]), PW, n, int(j["p0"]))
    same(g, o, "transform")

    rig.host[:] = o            # the coefficients just produced feed quant; weaken two thirds of the blocks so sparse paths run
    coef = rig.host[rig.out1:rig.out1 + rig.nj * SLOT].reshape(rig.nj, SLOT)
    coef[1::3, :n * n] //= 16
    coef[2::3, :n * n] //= 64
    q = np.zeros(rig.nj, JOB)
    q["a_off"] = rig.slots(rig.out1); q["c_off"] = rig.slots(rig.out2); q["b_off"] = q["c_off"] + n * n
    comp = rng.integers(0, 3 if n < 32 else 1, rig.nj); intra = rng.integers(0, 2, rig.nj); slice_i = intra | rng.integers(0, 2, rig.nj)
    sbh = rng.integers(0, 2, rig.nj); scan = np.where(intra == 1, rng.integers(1, 4, rig.nj), 3)
    per, rem = rng.integers(0, 9, rig.nj), rng.integers(0, 6, rig.nj)
    q["p0"] = scan | (comp << 2) | (intra << 4) | (slice_i << 5) | (sbh << 6); q["p1"] = per | (rem << 8)
    d_ac = rig.malloc(4 * rig.nj); rig.bufs.append(d_ac)
    g = rig.launch("hmr_gpu_quant_batch", rig.up(q), rig.nj, n, rig.dev, rig.dev, rig.dev, d_ac)
    o = rig.host.copy()
    ac = np.zeros(rig.nj, np.int32)
    for i, j in enumerate(q):
        v = C.c_int(0)
        oracle.ora_quant(at(o, j["a_off"]), at(o, j["c_off"]), at(o, j["b_off"]), int(scan[i]), quant_depth(n, int(comp[i])), int(comp[i]), int(intra[i]),
                         int(slice_i[i]), int(sbh[i]), C.byref(v), n, int(per[i]), int(rem[i]))
        ac[i] = v.value
    same(g, o, "quant levels + deltaU")
    same(rig.down(d_ac, rig.nj, np.int32), ac, "ac_sum")
    # without the deltaU scratch
    g2 = rig.launch("hmr_gpu_quant_batch", rig.up(q), rig.nj, n, rig.dev, rig.dev, None, d_ac)
    lev = lambda a: a[rig.out2:rig.out2 + rig.nj * SLOT].reshape(rig.nj, SLOT)[:, :n * n]   # noqa: E731
    same(lev(g2), lev(o), "quant levels (no deltaU)")

    rig.host[:] = o            # levels -> inv_quant -> itransform
    iq = np.zeros(rig.nj, JOB)
    iq["a_off"] = rig.slots(rig.out2); iq["c_off"] = rig.slots(rig.out1)
    iq["p0"] = (comp << 2) | (intra << 4); iq["p1"] = per | (rem << 8)
    g = rig.launch("hmr_gpu_inv_quant_batch", rig.up(iq), rig.nj, n, rig.dev, rig.dev)
    o = rig.host.copy()
    for i, j in enumerate(iq):
        oracle.ora_inv_quant(at(o, j["a_off"]), at(o, j["c_off"]), quant_depth(n, int(comp[i])), int(comp[i]), int(intra[i]), n, int(per[i]), int(rem[i]))
    same(g, o, "inv_quant")
    rig.host[:] = o
    it = np.zeros(rig.nj, JOB)
    it["a_off"] = rig.slots(rig.out1); it["c_off"] = rig.slots(rig.out2); it["c_stride"] = 80; it["p0"] = is_dst
    g = rig.launch("hmr_gpu_itransform_batch", rig.up(it), rig.nj, n, rig.dev, rig.dev)
    o = rig.host.copy()
    for j in it:
        oracle.ora_itransform(at(o, j["c_off"]), at(o, j["a_off"]), 80, n, int(j["p0"]))
    same(g, o, "itransform")


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_tu_chain(rig, oracle, n):
    rng = np.random.default_rng(n + 1000 * SEED)
    # prediction = source + noise of a per-job strength, so coded and all-zero TUs mix inside one wavefront
    src = rig.host[rig.pix:rig.res].reshape(PH, PW)
    noise = rng.integers(-1, 2, (PH, PW)) * np.repeat(np.repeat(rng.choice([0, 2, 12, 60], (PH // 32, PW // 32)), 32, 0), 32, 1)
    rig.host[rig.res:rig.mid] = np.clip(src + noise, 0, 255).ravel()
    jb = np.zeros(rig.nj, TU_JOB)
    pos = rig.block(rng, 0, n, n)
    jb["orig_off"] = rig.pix + pos; jb["pred_off"] = rig.res + pos; jb["orig_stride"] = jb["pred_stride"] = PW
    jb["rec_off"] = rig.slots(rig.out1); jb["rec_stride"] = 80
    jb["lev_off"] = rig.slots(rig.out2)
    comp = rng.integers(0, 3 if n < 32 else 1, rig.nj); intra = rng.integers(0, 2, rig.nj); sbh = rng.integers(0, 2, rig.nj)
    per, rem = rng.integers(2, 7, rig.nj), rng.integers(0, 6, rig.nj)
    is_dst = ((n == 4) & (intra == 1) & (comp == 0)).astype(np.int64)
    jb["p0"] = 3 | (comp << 2) | (intra << 4) | (intra << 5) | (sbh << 6) | (is_dst << 7); jb["p1"] = per | (rem << 8)
    d_ssd = rig.malloc(4 * rig.nj); d_ac = rig.malloc(4 * rig.nj); rig.bufs += [d_ssd, d_ac]
    g = rig.launch("hmr_gpu_tu_chain_batch", rig.up(jb), rig.nj, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
    o = rig.host.copy()
    ssd, ac = np.zeros(rig.nj, np.uint32), np.zeros(rig.nj, np.int32)
    oracle.ora_tu_chain.restype = C.c_uint32
    for i, j in enumerate(jb):
        v = C.c_int(0)
        ssd[i] = oracle.ora_tu_chain(at(o, j["orig_off"]), PW, at(o, j["pred_off"]), PW, at(o, j["lev_off"]), at(o, j["rec_off"]), 80, n, int(is_dst[i]), 3,
                                     int(comp[i]), int(intra[i]), int(intra[i]), int(sbh[i]), int(per[i]), int(rem[i]), C.byref(v))
        ac[i] = v.value
    same(g, o, "tu_chain levels + reconstruction")
    same(rig.down(d_ssd, rig.nj, np.uint32), ssd, "tu_chain ssd")
    same(rig.down(d_ac, rig.nj, np.int32), ac, "tu_chain ac_sum")


@pytest.mark.parametrize("luma", [1, 0])
@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("is_bi", [0, 1])
def test_motion_compensation(rig, oracle, luma, lanes, is_bi):
    rng = np.random.default_rng(luma * 10 + lanes + is_bi + 1000 * SEED)
    jb = np.zeros(rig.nj, JOB)
    if luma:   # deliberately includes blocks larger than the hint's LDS share (the kernel's recompute path)
        n = np.array([8, 16, 32, 64])[rng.integers(0, 4, rig.nj)]
        jb["w"] = n; jb["h"] = np.where(rng.random(rig.nj) < 0.3, n // 2, n)
    else:
        n = np.array([2, 4, 8, 16, 32])[rng.integers(0, 5, rig.nj)]
        jb["w"] = jb["h"] = n
    jb["a_off"] = rig.block(rng, rig.pix, 64, 64, margin=24); jb["a_stride"] = PW
    jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = 80
    fb = 2 if luma else 3
    frac = lambda: np.where(rng.random(rig.nj) < 0.35, 0, rng.integers(0, 1 << fb, rig.nj))   # noqa: E731
    mvx = (rng.integers(-16, 17, rig.nj) << fb) + frac(); mvy = (rng.integers(-16, 17, rig.nj) << fb) + frac()
    jb["p0"] = mvx.astype(np.int32).view(np.uint32); jb["p1"] = mvy.astype(np.int32).view(np.uint32)
    g = rig.launch("hmr_gpu_mc_batch", rig.up(jb), rig.nj, luma | (lanes << 8), is_bi, rig.dev, rig.dev)
    o = rig.host.copy()
    for i, j in enumerate(jb):
        if luma:
            oracle.ora_mc_luma(at(o, j["a_off"]), PW, at(o, j["c_off"]), 80, int(j["w"]), int(j["h"]), int(mvx[i]), int(mvy[i]), is_bi)
        else:
            oracle.ora_mc_chroma(at(o, j["a_off"]), PW, at(o, j["c_off"]), 80, int(j["w"]), int(mvx[i]), int(mvy[i]), is_bi)
    same(g, o, "motion compensation")


@pytest.mark.parametrize("n", [8, 16, 32, 64])
@pytest.mark.parametrize("action", [7, 6, 1, 3])
def test_motion_estimation(rig, oracle, n, action):
    rng = np.random.default_rng(n + action + 1000 * SEED)
    nj = rig.nj if n < 64 else 97
    # a reference with structure (so the search has something to find): the source plane is the reference shifted + noise
    ref = rig.host[rig.pix:rig.res].reshape(PH, PW)
    sm = ref.astype(np.int64)
    for _ in range(2):
        sm = (sm + np.roll(sm, 1, 0) + np.roll(sm, 1, 1) + np.roll(sm, (1, 1), (0, 1))) // 4
    ref[...] = sm
    src = np.roll(ref, (3, -5), (0, 1)) + rng.integers(-3, 4, (PH, PW))
    rig.host[rig.res:rig.mid] = np.clip(src, 0, 255).ravel()
    FW, FH = PW - 160, PH - 160         # picture = the plane minus an 80-sample margin on every side
    jb = np.zeros(nj, ME_JOB)
    gx = rng.integers(0, (FW - n) // 4 + 1, nj) * 4; gy = rng.integers(0, (FH - n) // 4 + 1, nj) * 4
    pos = (gy + 80) * PW + gx + 80
    jb["orig_off"] = rig.res + pos; jb["ref_off"] = rig.pix + pos; jb["orig_stride"] = jb["ref_stride"] = PW
    jb["gx"] = gx; jb["gy"] = gy
    jb["init_x"] = rng.integers(-12, 13, nj); jb["init_y"] = rng.integers(-8, 9, nj)
    jb["n_amvp"] = rng.integers(1, 3, nj); jb["n_search"] = rng.integers(0, 6, nj)
    jb["amvp"] = rng.integers(-48, 49, (nj, 2, 2)); jb["search"] = rng.integers(-40, 41, (nj, 5, 2))
    qp, avg = rng.integers(20, 40, nj), rng.integers(0, 4000, nj)
    jb["corr"] = qp * np.clip(avg / 2000.0, 0.15, 1.4)
    jb["action"] = action
    d_out = rig.malloc(20 * nj); rig.bufs.append(d_out)
    g = rig.launch("hmr_gpu_motion_estimation_batch", rig.up(jb), nj, n, rig.dev, rig.dev, 128, 64, FW, FH, d_out)
    same(g, rig.host, "arena untouched")
    got = rig.down(d_out, (nj, 5), np.int32)
    oracle.ora_motion_estimation.restype = C.c_uint32
    exp = np.zeros((nj, 5), np.int32)
    I32 = C.c_int32
    for i, j in enumerate(jb):
        amvp = (I32 * 4)(*[int(v) for v in j["amvp"].ravel()]); search = (I32 * 10)(*[int(v) for v in j["search"].ravel()])
        out = (I32 * 4)()
        sad = oracle.ora_motion_estimation(at(rig.host, j["orig_off"]), PW, at(rig.host, j["ref_off"]), PW, int(j["gx"]), int(j["gy"]), int(j["init_x"]),
                                           int(j["init_y"]), n, 128, 64, FW, FH, amvp, int(j["n_amvp"]), search, int(j["n_search"]),
                                           C.c_double(float(j["corr"])), action, out)
        exp[i] = [out[0], out[1], out[2], out[3], sad]
    same(got.view(np.uint32), exp.view(np.uint32), "motion estimation (mv, sub-pel mv, sad)")


INTRA_JOB = gpu_host.INTRA_JOB_DTYPE
INTRA_RES = np.dtype([("best_mode", "<i4"), ("bits", "<i4"), ("cost", "<f8")])


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_intra_search(rig, oracle, n):
    from kernel_cases import mpm_list
    assert INTRA_JOB.itemsize == 80 and INTRA_RES.itemsize == 16
    rng = np.random.default_rng(n + 1000 * SEED)
    nj = rig.nj if n < 64 else 151
    # directional texture so that different modes win; the source is the same texture plus noise
    yy, xx = np.mgrid[0:PH, 0:PW]
    th = np.repeat(np.repeat(rng.uniform(0, np.pi, (PH // 64, PW // 64)), 64, 0), 64, 1)
    tex = 128 + 70 * np.sin((xx * np.cos(th) + yy * np.sin(th)) / 6.0)
    rig.host[rig.pix:rig.res] = np.clip(tex + rng.integers(-4, 5, (PH, PW)), 0, 255).ravel()
    rig.host[rig.res:rig.mid] = np.clip(tex + rng.integers(-4, 5, (PH, PW)), 0, 255).ravel()
    jb = np.zeros(nj, INTRA_JOB)
    x = rng.integers(1, PW - 2 * n - 1, nj); y = rng.integers(1, PH - 2 * n - 1, nj)
    jb["orig_off"] = rig.res + y * PW + x; jb["orig_stride"] = PW
    jb["dec_off"] = rig.pix + (y - 1) * PW + x - 1; jb["dec_stride"] = PW
    slot = rig.slots(rig.out1)[:nj]
    jb["adi_off"] = slot; jb["adif_off"] = slot + 4 * n + 4; jb["pred_off"] = rig.slots(rig.out2)[:nj]; jb["pred_stride"] = 80
    avail = np.where(rng.random(nj) < 0.7, 15, rng.integers(0, 16, nj))
    left, top = (avail & 1) | ((avail >> 2) & 1), ((avail >> 1) & 1) | ((avail >> 3) & 1)
    bl, tr = (avail >> 2) & 1, (avail >> 3) & 1
    strong = rng.integers(0, 2, nj)
    bl_size = np.where(bl, np.where(rng.random(nj) < 0.6, n, max(n // 2, 4)), 0); tr_size = np.where(tr, np.where(rng.random(nj) < 0.6, n, max(n // 2, 4)), 0)
    jb["flags"] = left | (top << 1) | (bl << 2) | (tr << 3) | (strong << 5); jb["sizes"] = bl_size | (tr_size << 16)
    fast = rng.random(nj) < 0.7
    for i in range(nj):
        jb["preds"][i] = mpm_list(int(rng.integers(-1, 35)), int(rng.integers(-1, 35)))
        jb["pred_bits"][i] = [1, 1, 1] if fast[i] else rng.integers(0, 9, 3)
    jb["other_bits"] = np.where(fast, 12, 6)
    jb["sqrt_lambda"] = rng.uniform(1.0, 50.0, nj)
    d_out = rig.malloc(16 * nj); rig.bufs.append(d_out)
    g = rig.launch("hmr_gpu_intra_search_batch", rig.up(jb), nj, n, rig.dev, rig.dev, rig.dev, d_out)
    o = rig.host.copy()
    exp = np.zeros(nj, INTRA_RES)
    I32 = C.c_int32
    for i, j in enumerate(jb):
        out = (I32 * 2)(); cost = C.c_double(0)
        oracle.ora_intra_search(at(o, j["orig_off"]), PW, at(o, j["dec_off"]), PW, n, int(left[i]), int(top[i]), int(bl[i]), int(tr[i]), int(bl_size[i]),
                                int(tr_size[i]), int(strong[i]), (I32 * 3)(*[int(v) for v in j["preds"]]), (I32 * 3)(*[int(v) for v in j["pred_bits"]]),
                                int(j["other_bits"]), C.c_double(float(j["sqrt_lambda"])), at(o, j["adi_off"]), at(o, j["adif_off"]), at(o, j["pred_off"]), 80,
                                out, C.byref(cost))
        exp[i] = (out[0], out[1], cost.value)
    same(g, o, "intra search: neighbour arrays + last prediction")
    got = rig.down(d_out, nj, INTRA_RES)
    same(got["best_mode"], exp["best_mode"], "best mode")
    same(got["bits"], exp["bits"], "bits")
    same(got["cost"].view(np.uint64), exp["cost"].view(np.uint64), "cost (bit pattern)")
    assert len(set(exp["best_mode"].tolist())) > 8


ITU_JOB = gpu_host.ITU_JOB_DTYPE


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_intra_tu_chain(rig, oracle, n):
    from kernel_cases import intra_is_filtered
    assert ITU_JOB.itemsize == 56
    rng = np.random.default_rng(n + 1000 * SEED)
    nj = rig.nj
    yy, xx = np.mgrid[0:PH, 0:PW]
    th = np.repeat(np.repeat(rng.uniform(0, np.pi, (PH // 64, PW // 64)), 64, 0), 64, 1)
    amp = np.repeat(np.repeat(rng.choice([0, 3, 20, 70], (PH // 64, PW // 64)), 64, 0), 64, 1)     # flat areas give all-zero TUs
    tex = 128 + amp * np.sin((xx * np.cos(th) + yy * np.sin(th)) / 6.0)
    rig.host[rig.pix:rig.res] = np.clip(tex + rng.integers(-2, 3, (PH, PW)), 0, 255).ravel()
    rig.host[rig.res:rig.mid] = np.clip(tex + rng.integers(-2, 3, (PH, PW)), 0, 255).ravel()
    jb = np.zeros(nj, ITU_JOB)
    x = rng.integers(1, PW - 2 * n - 1, nj); y = rng.integers(1, PH - 2 * n - 1, nj)
    jb["orig_off"] = rig.res + y * PW + x; jb["orig_stride"] = PW
    jb["dec_off"] = rig.pix + (y - 1) * PW + x - 1; jb["dec_stride"] = PW
    s1, s2 = rig.slots(rig.out1), rig.slots(rig.out2)
    jb["pred_off"] = s1; jb["pred_stride"] = n
    jb["lev_off"] = s2; jb["rec_off"] = s2 + 2048; jb["rec_stride"] = 80
    avail = np.where(rng.random(nj) < 0.7, 15, rng.integers(0, 16, nj))
    left, top = (avail & 1) | ((avail >> 2) & 1), ((avail >> 1) & 1) | ((avail >> 3) & 1)
    bl, tr = (avail >> 2) & 1, (avail >> 3) & 1
    strong = rng.integers(0, 2, nj)
    bl_size = np.where(bl, np.where(rng.random(nj) < 0.6, n, max(n // 2, 4)), 0); tr_size = np.where(tr, np.where(rng.random(nj) < 0.6, n, max(n // 2, 4)), 0)
    comp = rng.integers(0, 3 if n < 32 else 1, nj); luma = (comp == 0).astype(np.int64)
    mode = rng.integers(0, 35, nj)
    filt = np.array([intra_is_filtered(n, int(m)) for m in mode]) * luma
    jb["flags"] = left | (top << 1) | (bl << 2) | (tr << 3) | (strong << 5) | (filt << 6) | (luma << 7); jb["sizes"] = bl_size | (tr_size << 16)
    jb["mode"] = mode
    scan, slice_i, sbh = rng.integers(1, 4, nj), rng.integers(0, 2, nj), rng.integers(0, 2, nj)
    per, rem = rng.integers(2, 7, nj), rng.integers(0, 6, nj)
    is_dst = ((n == 4) & (luma == 1)).astype(np.int64)
    jb["p0"] = scan | (comp << 2) | (1 << 4) | (slice_i << 5) | (sbh << 6) | (is_dst << 7); jb["p1"] = per | (rem << 8)
    d_ssd = rig.malloc(4 * nj); d_ac = rig.malloc(4 * nj); rig.bufs += [d_ssd, d_ac]
    g = rig.launch("hmr_gpu_intra_tu_chain_batch", rig.up(jb), nj, n, rig.dev, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
    o = rig.host.copy()
    ssd, ac = np.zeros(nj, np.uint32), np.zeros(nj, np.int32)
    oracle.ora_intra_tu_chain.restype = C.c_uint32
    for i, j in enumerate(jb):
        v = C.c_int(0)
        ssd[i] = oracle.ora_intra_tu_chain(at(o, j["orig_off"]), PW, at(o, j["dec_off"]), PW, int(left[i]), int(top[i]), int(bl[i]), int(tr[i]), int(bl_size[i]),
                                           int(tr_size[i]), int(strong[i]), int(filt[i]), int(mode[i]), int(luma[i]), at(o, j["pred_off"]), n, at(o, j["lev_off"]),
                                           at(o, j["rec_off"]), 80, n, int(is_dst[i]), int(scan[i]), int(comp[i]), int(slice_i[i]), int(sbh[i]), int(per[i]),
                                           int(rem[i]), C.byref(v))
        ac[i] = v.value
    same(g, o, "intra TU chain: prediction, levels, reconstruction")
    same(rig.down(d_ssd, nj, np.uint32), ssd, "ssd")
    same(rig.down(d_ac, nj, np.int32), ac, "ac_sum")
    assert 0.1 < (ac != 0).mean() < 0.98


INTER_TU_JOB = gpu_host.INTER_TU_JOB_DTYPE


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_inter_tu_chain(rig, oracle, n):
    assert INTER_TU_JOB.itemsize == 56
    rng = np.random.default_rng(n + 1000 * SEED)
    nj = rig.nj
    # residual plane: smooth texture of per-region strength (kept / dropped / all-zero TUs all occur)
    yy, xx = np.mgrid[0:PH, 0:PW]
    amp = np.repeat(np.repeat(rng.choice([0, 2, 5, 15, 60], (PH // 32, PW // 32)), 32, 0), 32, 1)
    rig.host[rig.res:rig.mid] = (amp * np.sin(xx / 5.0 + yy / 7.0) + rng.integers(-2, 3, (PH, PW)) * (amp > 0)).astype(np.int16).ravel()
    jb = np.zeros(nj, INTER_TU_JOB)
    jb["orig_off"] = rig.block(rng, rig.res, n, n); jb["orig_stride"] = PW
    jb["pred_off"] = rig.block(rng, rig.pix, n, n); jb["pred_stride"] = PW
    jb["rec_off"] = rig.slots(rig.out1); jb["rec_stride"] = 80
    jb["lev_off"] = rig.slots(rig.out2)
    comp = rng.integers(0, 3 if n < 32 else 1, nj); sbh = rng.integers(0, 2, nj)
    per, rem = rng.integers(2, 7, nj), rng.integers(0, 6, nj)
    jb["p0"] = 3 | (comp << 2) | (sbh << 6); jb["p1"] = per | (rem << 8)
    jb["weight"] = np.where(comp == 0, 1.0, 2.0 ** (rng.integers(-2, 5, nj) / 3.0))
    jb["zero_thr"] = np.clip(rng.uniform(0, 3000, nj) / 2.5 - 5.0, 1.0, 20000.0)
    d_ssd = rig.malloc(4 * nj); d_ac = rig.malloc(4 * nj); rig.bufs += [d_ssd, d_ac]
    g = rig.launch("hmr_gpu_inter_tu_chain_batch", rig.up(jb), nj, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
    o = rig.host.copy()
    ssd, ac = np.zeros(nj, np.uint32), np.zeros(nj, np.int32)
    oracle.ora_inter_tu_chain.restype = C.c_uint32
    for i, j in enumerate(jb):
        v = C.c_int(0)
        ssd[i] = oracle.ora_inter_tu_chain(at(o, j["orig_off"]), PW, at(o, j["pred_off"]), PW, at(o, j["lev_off"]), at(o, j["rec_off"]), 80, n, 3, int(comp[i]), 0,
                                           int(sbh[i]), int(per[i]), int(rem[i]), C.c_double(float(j["weight"])), C.c_double(float(j["zero_thr"])), C.byref(v))
        ac[i] = v.value
    same(g, o, "inter TU chain: levels, reconstruction")
    same(rig.down(d_ssd, nj, np.uint32), ssd, "ssd")
    same(rig.down(d_ac, nj, np.int32), ac, "ac_sum")
    dropped = (ac == 0) & (ssd != 0)
    assert (ac != 0).sum() > 20 and dropped.sum() > 20


TREE_JOB, TREE_RES = gpu_host.TREE_JOB_DTYPE, gpu_host.TREE_RESULT_DTYPE


@pytest.mark.parametrize("rounds", [1, 4], ids=["children_4_launches", "children_1_launch"])
@pytest.mark.parametrize("n", [8, 16, 32, 64])
def test_intra_luma_cu_tree(rig, oracle, n, rounds):
    """The luma intra CU as a chain of launches with no host step in between (include/homer_gpu.h section 9): mode search -> parent TUs -> children 0..3 ->
    consolidation, the mode handed from the search to the TU launches on the device.  Every CU has its own pair of planes (parent / child level) in one
    device arena; the oracle's ora_intra_luma_cu runs on a host copy and the whole arena is compared."""
    from kernel_cases import cu_tree_neighbours, mpm_list
    rng = np.random.default_rng(n + 1000 * SEED)
    nj = 301 if n <= 16 else 151 if n == 32 else 61
    h, ring, adi = n // 2, 2 * n + 1, 4 * n + 4
    per_cu = 2 * ring * ring + 4 * n * n + 2 * adi
    host = np.zeros(nj * per_cu, np.int16)
    o_pp = np.arange(nj) * per_cu; o_pc = o_pp + ring * ring; o_orig = o_pc + ring * ring; o_pred = o_orig + n * n
    o_lp = o_pred + n * n; o_lc = o_lp + n * n; o_adi = o_lc + n * n; o_adif = o_adi + adi
    yy, xx = np.mgrid[0:ring, 0:ring]
    nbs, rules = [], rng.integers(0, 2, nj)
    for i in range(nj):
        th, amp, noise = rng.uniform(0, np.pi), rng.choice([3, 15, 40, 80]), int(rng.choice([0, 1, 3, 8, 20]))
        tex = 128 + amp * np.sin((xx * np.cos(th) + yy * np.sin(th)) / rng.uniform(3, 20))
        img = np.clip(tex + rng.integers(-noise, noise + 1, (ring, ring)), 0, 255).astype(np.int16)
        for o in (o_pp[i], o_pc[i]):
            host[o:o + ring * ring] = np.clip(img + rng.integers(-2, 3, (ring, ring)), 0, 255).ravel()
        host[o_orig[i]:o_orig[i] + n * n] = img[1:n + 1, 1:n + 1].ravel()
        left, top = (int(rng.integers(0, 2)), int(rng.integers(0, 2))) if i % 5 == 0 else (1, 1)
        bl, tr = int(rng.integers(0, 2)) & left, int(rng.integers(0, 2)) & top
        flags = [(left, top, bl, tr), (left, top, left, top), (1, top, 0, tr), (left, 1, bl, 1), (1, 1, 0, 0)]
        nbs.append(cu_tree_neighbours(n, flags, int(rng.choice([n, n + h, 2 * n, 4 * n])), int(rng.choice([n, n + h, 2 * n, 4 * n]))))
    nbs = np.array(nbs, np.int32).reshape(nj, 5, 6)
    strong, slice_i, sbh = rng.integers(0, 2, nj), rng.integers(0, 2, nj), rng.integers(0, 2, nj)
    qp = rng.integers(18, 45, nj)
    sj = np.zeros(nj, INTRA_JOB)
    sj["sqrt_lambda"] = rng.uniform(2.0, 60.0, nj)
    sj["orig_off"] = o_orig; sj["orig_stride"] = n; sj["dec_off"] = o_pp; sj["dec_stride"] = ring
    sj["adi_off"] = o_adi; sj["adif_off"] = o_adif; sj["pred_off"] = o_pred; sj["pred_stride"] = n
    fl = lambda f: f[:, 0] | (f[:, 1] << 1) | (f[:, 2] << 2) | (f[:, 3] << 3) | (strong << 5)
    sj["flags"] = fl(nbs[:, 0]); sj["sizes"] = nbs[:, 0, 4] | (nbs[:, 0, 5] << 16)
    for i in range(nj):
        sj["preds"][i] = mpm_list(int(rng.integers(-1, 35)), int(rng.integers(-1, 35)))
    sj["pred_bits"] = np.where(rules[:, None] == 1, 1, 0); sj["other_bits"] = np.where(rules == 1, 12, 0)
    gx, gy, gs = [0, 0, h, 0, h], [0, 0, 0, h, h], [n, h, h, h, h]
    tj = np.zeros((5, nj), ITU_JOB)
    for k in range(5):
        plane = o_pc if k else o_pp
        t = tj[k]
        t["orig_off"] = o_orig + gy[k] * n + gx[k]; t["orig_stride"] = n
        t["pred_off"] = o_pred + gy[k] * n + gx[k]; t["pred_stride"] = n
        t["dec_off"] = plane + gy[k] * ring + gx[k]; t["dec_stride"] = ring
        t["rec_off"] = t["dec_off"] + ring + 1; t["rec_stride"] = ring
        t["lev_off"] = (o_lc + (k - 1) * h * h) if k else o_lp
        t["flags"] = fl(nbs[:, k]) | (1 << 7) | gpu_host.ITU_MODE_FROM_SEARCH; t["sizes"] = nbs[:, k, 4] | (nbs[:, k, 5] << 16)
        t["mode"] = np.arange(nj)
        t["p0"] = (1 << 4) | (slice_i << 5) | (sbh << 6) | (int(gs[k] == 4) << 7); t["p1"] = (qp // 6) | ((qp % 6) << 8)
    dj = np.zeros(nj, TREE_JOB)
    dj["parent"] = np.arange(nj) if n <= 32 else gpu_host.TREE_NO_PARENT
    for k in range(4):
        dj["child"][:, k] = (k + 1) * nj + np.arange(nj)
    dj["par_rec_off"] = o_pp + ring + 1; dj["par_rec_stride"] = ring; dj["chl_rec_off"] = o_pc + ring + 1; dj["chl_rec_stride"] = ring
    dj["par_lev_off"] = o_lp; dj["chl_lev_off"] = o_lc; dj["size"] = n; dj["rule"] = rules
    gpu, ctx = rig.gpu, rig.ctx
    d_arena = rig.up(host)
    d_modes = rig.malloc(16 * nj); d_ssd = rig.malloc(4 * 5 * nj); d_ac = rig.malloc(4 * 5 * nj); d_res = rig.malloc(16 * nj)
    rig.bufs += [d_modes, d_ssd, d_ac, d_res]
    zero = np.zeros(5 * nj, np.uint32)
    for d in (d_ssd, d_ac):
        assert gpu.hmr_gpu_upload(ctx, d, VP(zero.ctypes.data), C.c_size_t(zero.nbytes)) == 0
    ok = lambda rc, what: (_ for _ in ()).throw(AssertionError((what, gpu.hmr_gpu_last_error()))) if rc else None
    ok(gpu.hmr_gpu_intra_search_batch(ctx, rig.up(sj), nj, n, d_arena, d_arena, d_arena, d_modes), "search")
    for k in range(0 if n <= 32 else 1, 5 if rounds == 1 else 1):
        ok(gpu.hmr_gpu_intra_tu_chain_modes_batch(ctx, rig.up(tj[k]), nj, gs[k], d_arena, d_arena, d_arena, d_arena, d_arena, VP(d_ssd.value + 4 * k * nj),
                                                  VP(d_ac.value + 4 * k * nj), d_modes), f"TU level {k}")
    if rounds == 4:     # the four children of every CU back to back in ONE launch
        ok(gpu.hmr_gpu_intra_tu_chain_rounds_batch(ctx, rig.up(np.concatenate([tj[1], tj[2], tj[3], tj[4]])), nj, 4, h, d_arena, d_arena, d_arena, d_arena, d_arena,
                                                   VP(d_ssd.value + 4 * nj), VP(d_ac.value + 4 * nj), d_modes), "children")
    ok(gpu.hmr_gpu_tree_decide_batch(ctx, rig.up(dj), nj, d_ssd, d_ac, d_arena, d_arena, d_res), "consolidation")
    assert gpu.hmr_gpu_sync(ctx) == 0, gpu.hmr_gpu_last_error()
    got = rig.down(d_arena, host.size, np.int16)
    res = rig.down(d_res, nj, TREE_RES); modes = rig.down(d_modes, nj, gpu_host.INTRA_RESULT_DTYPE)
    ssd = rig.down(d_ssd, 5 * nj, np.uint32).reshape(5, nj); ac = rig.down(d_ac, 5 * nj, np.int32).reshape(5, nj)
    o = host.copy()
    I32 = C.c_int32
    splits = 0
    for i in range(nj):
        out = (I32 * 24)(); cost = C.c_double(0)
        nb = np.ascontiguousarray(nbs[i].ravel())
        oracle.ora_intra_luma_cu(at(o, o_orig[i]), n, at(o, o_pp[i] + ring + 1), ring, at(o, o_pc[i] + ring + 1), ring, VP(nb.ctypes.data), int(strong[i]),
                                 (I32 * 3)(*[int(v) for v in sj["preds"][i]]), (I32 * 3)(*[int(v) for v in sj["pred_bits"][i]]), int(sj["other_bits"][i]),
                                 C.c_double(float(sj["sqrt_lambda"][i])), at(o, o_adi[i]), at(o, o_adif[i]), at(o, o_pred[i]), n, at(o, o_lp[i]), at(o, o_lc[i]), n,
                                 int(slice_i[i]), int(sbh[i]), int(qp[i]) // 6, int(qp[i]) % 6, int(rules[i]), out, C.byref(cost))
        exp = list(out)
        assert (int(res["split"][i]), int(res["cost"][i]), int(res["sum"][i])) == (exp[0], exp[1] & 0xFFFFFFFF, exp[3]), (i, res[i], exp[:9])
        assert list(res["cbf"][i]) == exp[4:8], (i, res[i], exp[:9])
        assert int(modes["best_mode"][i]) == exp[19] and int(modes["bits"][i]) == exp[20]
        assert modes["cost"][i:i + 1].view(np.uint64)[0] == np.array([cost.value]).view(np.uint64)[0]
        for k in range(1, 5):
            assert (int(ssd[k, i]), int(ac[k, i])) == (exp[9 + k] & 0xFFFFFFFF, exp[14 + k]), (i, k)
        splits += exp[0]
    same(got, o, "luma CU tree: planes, prediction, levels, neighbour arrays")
    if n <= 32:
        assert 0.1 < splits / nj < 0.9, splits


class Segment(C.Structure):
    _fields_ = [("jobs", VP), ("out", VP), ("njobs", C.c_int), ("size", C.c_int)]


@pytest.mark.parametrize("op", ["sad", "ssd16b", "predict", "reconst", "copy"])
def test_pixel_multi(rig, oracle, op):
    """hmr_gpu_pixel_multi: the five block sizes of one pixel kernel as segments of ONE launch (segment block ranges, XCD partition inside a segment,
    ragged and tiny segments) against the oracle, whole arena compared."""
    rng = np.random.default_rng(91 + 1000 * SEED)
    opcode = {"sad": 1, "ssd16b": 2, "predict": 3, "reconst": 4, "copy": 5}[op]
    sizes, counts = [4, 8, 16, 32, 64], [rig.nj, 3, 200, 77, 41]
    segs, jobs, outs = (Segment * 5)(), [], []
    slot0 = 0
    for i, (n, cnt) in enumerate(zip(sizes, counts)):
        jb = np.zeros(cnt, JOB)
        x = rng.integers(0, PW - n + 1, cnt); y = rng.integers(0, PH - n + 1, cnt)
        jb["a_off"] = rig.pix + y * PW + x; jb["a_stride"] = PW
        x = rng.integers(0, PW - n + 1, cnt); y = rng.integers(0, PH - n + 1, cnt)
        jb["b_off"] = (rig.res if op in ("ssd16b", "reconst") else rig.pix) + y * PW + x; jb["b_stride"] = PW
        jb["c_off"] = rig.out1 + (slot0 + np.arange(cnt)) * SLOT; jb["c_stride"] = 80
        jb["w"] = jb["h"] = n
        slot0 += cnt
        d_out = rig.malloc(4 * cnt); rig.bufs.append(d_out)
        segs[i] = Segment(rig.up(jb), d_out, cnt, n)
        jobs.append(jb); outs.append(d_out)
    assert slot0 <= 2 * rig.nj
    g = rig.launch("hmr_gpu_pixel_multi", opcode, segs, 5, rig.dev, rig.dev, rig.dev)
    o = rig.host.copy()
    for n, jb, d_out in zip(sizes, jobs, outs):
        if op in ("sad", "ssd16b"):
            f = getattr(oracle, "ora_" + op); f.restype = C.c_uint32
            exp = np.array([f(at(o, j["a_off"]), PW, at(o, j["b_off"]), PW, n) for j in jb], np.uint32)
            same(rig.down(d_out, len(jb), np.uint32), exp, f"{op} {n}")
        else:
            for j in jb:
                if op == "copy":
                    for r in range(n):
                        dst = int(j["c_off"]) + r * 80; src = int(j["a_off"]) + r * PW
                        o[dst:dst + n] = o[src:src + n]
                else:
                    getattr(oracle, "ora_" + op)(at(o, j["a_off"]), PW, at(o, j["b_off"]), PW, at(o, j["c_off"]), 80, n)
    same(g, o, f"{op} multi: arena")


class TuSegment(C.Structure):
    _fields_ = [("jobs", VP), ("ssd", VP), ("ac_sum", VP), ("modes", VP), ("njobs", C.c_int), ("size", C.c_int), ("kind", C.c_int), ("rounds", C.c_int)]


@pytest.mark.parametrize("with32", [0, 1], ids=["upto16", "with32"])
def test_tu_chain_multi(rig, oracle, with32):
    """hmr_gpu_tu_chain_multi: given-prediction, intra and inter TU batches of different sizes as segments of ONE launch must leave exactly what the
    single-batch entries leave (those are checked against the oracle above): same arena, same SSD / sum arrays."""
    rng = np.random.default_rng(5 + 1000 * SEED + with32)
    yy, xx = np.mgrid[0:PH, 0:PW]
    amp = np.repeat(np.repeat(rng.choice([0, 3, 20, 70], (PH // 64, PW // 64)), 64, 0), 64, 1)
    tex = 128 + amp * np.sin((xx * 0.7 + yy * 0.4) / 6.0)
    rig.host[rig.pix:rig.res] = np.clip(tex + rng.integers(-2, 3, (PH, PW)), 0, 255).ravel()
    rig.host[rig.res:rig.mid] = (amp * np.sin(xx / 5.0 + yy / 7.0) / 3 + rng.integers(-2, 3, (PH, PW))).astype(np.int16).ravel()
    plan = [(4, 0, 300), (8, 1, 157), (16, 2, 90), (8, 2, 33), (4, 1, 5), (16, 0, 64)] + ([(32, 2, 40), (32, 1, 17)] if with32 else [])
    segs, singles, slot = [], [], 0
    for n, kind, cnt in plan:
        dt = {0: TU_JOB, 1: ITU_JOB, 2: INTER_TU_JOB}[kind]
        jb = np.zeros(cnt, dt)
        x = rng.integers(1, PW - 2 * n - 1, cnt); y = rng.integers(1, PH - 2 * n - 1, cnt)
        jb["orig_off"] = (rig.res if kind == 2 else rig.pix) + y * PW + x; jb["orig_stride"] = PW
        sl = rig.out1 + (slot + np.arange(cnt)) * SLOT
        slot += cnt
        jb["rec_off"] = sl; jb["rec_stride"] = 80; jb["lev_off"] = sl + 40 * 80
        if kind == 1:
            jb["pred_off"] = sl + 40; jb["pred_stride"] = 80
            jb["dec_off"] = rig.pix + (y - 1) * PW + x - 1; jb["dec_stride"] = PW
            mode = rng.integers(0, 35, cnt)
            jb["flags"] = 15 | 32 | (rng.integers(0, 2, cnt) << 6) | (1 << 7); jb["sizes"] = n | (n << 16); jb["mode"] = mode
            jb["p0"] = rng.integers(1, 4, cnt) | (1 << 4) | (rng.integers(0, 2, cnt) << 5) | (1 << 6) | (int(n == 4) << 7)
        else:
            x2 = rng.integers(0, PW - n, cnt); y2 = rng.integers(0, PH - n, cnt)
            jb["pred_off"] = rig.pix + y2 * PW + x2; jb["pred_stride"] = PW
            jb["p0"] = 3 | ((1 << 4) if kind == 0 else 0) | (1 << 6)
            if kind == 2:
                jb["weight"] = 1.0; jb["zero_thr"] = np.clip(rng.uniform(0, 3000, cnt) / 2.5 - 5.0, 1.0, 20000.0)
        jb["p1"] = rng.integers(2, 7, cnt) | (rng.integers(0, 6, cnt) << 8)
        d_jobs = rig.up(jb)
        bufs = [rig.malloc(4 * cnt) for _ in range(4)]; rig.bufs += bufs
        segs.append(TuSegment(d_jobs, bufs[0], bufs[1], None, cnt, n, kind, 0))
        singles.append((kind, n, d_jobs, cnt, bufs[2], bufs[3]))
    assert slot <= 2 * rig.nj
    arr = (TuSegment * len(segs))(*segs)
    g_multi = rig.launch("hmr_gpu_tu_chain_multi", arr, len(segs), rig.dev, rig.dev, rig.dev, rig.dev, rig.dev)
    # the same batches one launch each
    assert rig.gpu.hmr_gpu_upload(rig.ctx, rig.dev, VP(rig.host.ctypes.data), C.c_size_t(rig.size * 2)) == 0
    for kind, n, d_jobs, cnt, d_ssd, d_ac in singles:
        if kind == 0:
            rc = rig.gpu.hmr_gpu_tu_chain_batch(rig.ctx, d_jobs, cnt, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
        elif kind == 1:
            rc = rig.gpu.hmr_gpu_intra_tu_chain_batch(rig.ctx, d_jobs, cnt, n, rig.dev, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
        else:
            rc = rig.gpu.hmr_gpu_inter_tu_chain_batch(rig.ctx, d_jobs, cnt, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
        assert rc == 0, rig.gpu.hmr_gpu_last_error()
    assert rig.gpu.hmr_gpu_sync(rig.ctx) == 0
    g_single = rig.down(rig.dev, rig.size, np.int16)
    same(g_multi, g_single, "arena after the multi launch vs after the single launches")
    assert not np.array_equal(g_single, rig.host)
    for seg, (kind, n, d_jobs, cnt, d_ssd, d_ac) in zip(segs, singles):
        same(rig.down(VP(seg.ssd), cnt, np.uint32), rig.down(d_ssd, cnt, np.uint32), f"ssd {n}/{kind}")
        same(rig.down(VP(seg.ac_sum), cnt, np.int32), rig.down(d_ac, cnt, np.int32), f"ac_sum {n}/{kind}")


@pytest.mark.parametrize("n", [4, 8, 16])
def test_chroma_search_and_tus(rig, oracle, n):
    """The chroma half of intra CUs as two launches: hmr_gpu_chroma_search_batch (luma mode read from a luma search result array on the device), then the U / V
    TUs of every CU in one rounds launch with their mode taken from the chroma search; every CU owns a pair of planes.  Oracle: ora_intra_chroma_cu."""
    from kernel_cases import cu_tree_neighbours
    CH = gpu_host.CHROMA_JOB_DTYPE
    rng = np.random.default_rng(n + 17 + 1000 * SEED)
    nj = 301
    h, ring = n // 2, 2 * n + 1
    split = (rng.integers(0, 2, nj) if n > 4 else np.zeros(nj, np.int64))
    per_cu = 2 * ring * ring + 6 * n * n
    host = np.zeros(nj * per_cu, np.int16)
    o_pl = [np.arange(nj) * per_cu, np.arange(nj) * per_cu + ring * ring]
    o_org = [o_pl[1] + ring * ring, o_pl[1] + ring * ring + n * n]
    o_prd = [o_org[1] + n * n, o_org[1] + 2 * n * n]
    o_lev = [o_prd[1] + n * n, o_prd[1] + 2 * n * n]
    yy, xx = np.mgrid[0:ring, 0:ring]
    nbs = []
    for i in range(nj):
        for c in range(2):
            th, amp, noise = rng.uniform(0, np.pi), rng.choice([3, 15, 40, 80]), int(rng.choice([0, 1, 3, 8, 20]))
            img = np.clip(128 + amp * np.sin((xx * np.cos(th) + yy * np.sin(th)) / rng.uniform(3, 20)) + rng.integers(-noise, noise + 1, (ring, ring)), 0, 255).astype(np.int16)
            host[o_pl[c][i]:o_pl[c][i] + ring * ring] = np.clip(img + rng.integers(-2, 3, (ring, ring)), 0, 255).ravel()
            host[o_org[c][i]:o_org[c][i] + n * n] = img[1:n + 1, 1:n + 1].ravel()
        left, top = (int(rng.integers(0, 2)), int(rng.integers(0, 2))) if i % 5 == 0 else (1, 1)
        bl, tr = int(rng.integers(0, 2)) & left, int(rng.integers(0, 2)) & top
        flags = [(left, top, bl, tr), (left, top, left, top), (1, top, 0, tr), (left, 1, bl, 1), (1, 1, 0, 0)]
        nbs.append(cu_tree_neighbours(n, flags, int(rng.choice([n, n + h, 2 * n, 4 * n])), int(rng.choice([n, n + h, 2 * n, 4 * n]))))
    nbs = np.array(nbs, np.int32).reshape(nj, 5, 6)
    luma = np.zeros(nj, gpu_host.INTRA_RESULT_DTYPE)
    luma["best_mode"] = rng.choice([0, 1, 10, 26, 2, 7, 18, 33, 34], nj)
    slice_i, sbh, per, rem = rng.integers(0, 2, nj), rng.integers(0, 2, nj), rng.integers(2, 7, nj), rng.integers(0, 6, nj)
    sj = np.zeros(nj, CH)
    sj["sqrt_lambda"] = rng.uniform(2.0, 60.0, nj)
    sj["orig_u_off"] = o_org[0]; sj["orig_v_off"] = o_org[1]; sj["orig_stride"] = n
    sj["dec_u_off"] = o_pl[0]; sj["dec_v_off"] = o_pl[1]; sj["dec_stride"] = ring
    fl = lambda f: f[:, 0] | (f[:, 1] << 1) | (f[:, 2] << 2) | (f[:, 3] << 3)
    sj["flags"] = fl(nbs[:, 0]) | 0x100; sj["sizes"] = nbs[:, 0, 4] | (nbs[:, 0, 5] << 16)
    sj["luma_mode"] = np.arange(nj)
    # TU jobs: unsplit CUs go out as one launch of size n, split ones as four rounds of size n / 2; [round][cu * 2 + comp]
    def tu_jobs(sel, rounds, tn):
        idx = np.flatnonzero(sel)
        t = np.zeros((rounds, 2 * len(idx)), ITU_JOB)
        for r in range(rounds):
            x0, y0 = ((r & 1) * tn, (r >> 1) * tn) if rounds == 4 else (0, 0)
            f = nbs[idx, r + 1 if rounds == 4 else 0]
            for c in range(2):
                q = t[r, c::2]
                q["orig_off"] = o_org[c][idx] + y0 * n + x0; q["orig_stride"] = n
                q["pred_off"] = o_prd[c][idx] + y0 * n + x0; q["pred_stride"] = n
                q["dec_off"] = o_pl[c][idx] + y0 * ring + x0; q["dec_stride"] = ring
                q["rec_off"] = q["dec_off"] + ring + 1; q["rec_stride"] = ring
                q["lev_off"] = o_lev[c][idx] + r * tn * tn
                q["flags"] = fl(f) | gpu_host.ITU_MODE_FROM_SEARCH; q["sizes"] = f[:, 4] | (f[:, 5] << 16)
                q["mode"] = idx
                q["p0"] = ((c + 1) << 2) | (1 << 4) | (slice_i[idx] << 5) | (sbh[idx] << 6); q["p1"] = per[idx] | (rem[idx] << 8)
        return idx, t
    gpu, ctx = rig.gpu, rig.ctx
    d_arena = rig.up(host)
    d_luma = rig.up(luma)
    d_modes = rig.malloc(16 * nj); rig.bufs.append(d_modes)
    ok = lambda rc, what: (_ for _ in ()).throw(AssertionError((what, gpu.hmr_gpu_last_error()))) if rc else None
    ok(gpu.hmr_gpu_chroma_search_batch(ctx, rig.up(sj), nj, n, d_arena, d_arena, d_luma, d_modes), "chroma search")
    outs = []
    for sel, rounds, tn in ((split == 0, 1, n), (split == 1, 4, h)):
        idx, t = tu_jobs(sel, rounds, tn)
        if not len(idx):
            continue
        m = 2 * len(idx)
        d_ssd = rig.malloc(4 * rounds * m); d_ac = rig.malloc(4 * rounds * m); rig.bufs += [d_ssd, d_ac]
        ok(gpu.hmr_gpu_intra_tu_chain_rounds_batch(ctx, rig.up(t.reshape(-1)), m, rounds, tn, d_arena, d_arena, d_arena, d_arena, d_arena, d_ssd, d_ac, d_modes), "chroma TUs")
        outs.append((idx, rounds, m, d_ssd, d_ac))
    assert gpu.hmr_gpu_sync(ctx) == 0, gpu.hmr_gpu_last_error()
    got = rig.down(d_arena, host.size, np.int16)
    modes = rig.down(d_modes, nj, gpu_host.INTRA_RESULT_DTYPE)
    ssd_ac = {}
    for idx, rounds, m, d_ssd, d_ac in outs:
        s_ = rig.down(d_ssd, rounds * m, np.uint32).reshape(rounds, len(idx), 2); a_ = rig.down(d_ac, rounds * m, np.int32).reshape(rounds, len(idx), 2)
        for k, i in enumerate(idx):
            ssd_ac[int(i)] = (s_[:, k, :], a_[:, k, :])
    o = host.copy()
    I32 = C.c_int32
    coded = set()
    for i in range(nj):
        out = (I32 * 16)()
        nb = np.ascontiguousarray(nbs[i].ravel())
        oracle.ora_intra_chroma_cu(at(o, o_org[0][i]), at(o, o_org[1][i]), n, at(o, o_pl[0][i] + ring + 1), at(o, o_pl[1][i] + ring + 1), ring, VP(nb.ctypes.data),
                                   int(luma["best_mode"][i]), int(split[i]), C.c_double(float(sj["sqrt_lambda"][i])), C.c_double(1.0), at(o, o_prd[0][i]), at(o, o_prd[1][i]),
                                   n, at(o, o_lev[0][i]), at(o, o_lev[1][i]), n, int(slice_i[i]), int(sbh[i]), int(per[i]), int(rem[i]), out)
        exp = list(out)
        assert (int(modes["best_mode"][i]) >> 8, int(modes["best_mode"][i]) & 0xff, int(modes["bits"][i]), int(modes["cost"][i])) == tuple(exp[0:4]), (i, modes[i], exp[:6])
        s_, a_ = ssd_ac[i]
        assert int(s_.sum()) == exp[4] & 0xFFFFFFFF and int(a_.sum()) == exp[5], (i, s_, a_, exp[:6])      # weight 1.0: distortion = sum of the SSDs
        coded.add(exp[0])
    same(got, o, "chroma CUs: planes, predictions, levels")
    assert len(coded) >= 4


def test_sao_offsets_frame(rig, oracle):
    """hmr_gpu_sao_offsets_frame: the offsets, band positions and distortions of every (CTU, component, type) of a 1080p frame's worth of statistics against
    ora_sao_offsets_ctu."""
    rng = np.random.default_rng(31 + 1000 * SEED)
    n_ctu = 510
    stats = np.zeros((n_ctu, 3, 5, 2, 32), np.int32)
    for t in range(5):
        ncls = 32 if t == 4 else 5
        cnt = rng.integers(0, 600, (n_ctu, 3, ncls)) * (rng.random((n_ctu, 3, ncls)) > 0.25)
        sign = rng.choice([-1, 1], (n_ctu, 3, ncls)) if t == 4 else np.array([1, 1, 0, -1, -1]) * rng.choice([1, 1, 1, -1], (n_ctu, 3, ncls))
        stats[:, :, t, 1, :ncls] = cnt
        stats[:, :, t, 0, :ncls] = np.round(cnt * sign * rng.gamma(1.0, rng.choice([0.2, 1.0, 4.0], (n_ctu, 3, 1)), (n_ctu, 3, ncls)))
    lambdas = np.ascontiguousarray(rng.choice([1.0, 8.0, 56.0, 300.0], (n_ctu, 1)) * rng.uniform(0.6, 1.5, (n_ctu, 3)))
    d_off = rig.malloc(n_ctu * 15 * 32 * 4); d_aux = rig.malloc(n_ctu * 15 * 4); d_dist = rig.malloc(n_ctu * 15 * 8); rig.bufs += [d_off, d_aux, d_dist]
    rc = rig.gpu.hmr_gpu_sao_offsets_frame(rig.ctx, rig.up(stats), n_ctu, rig.up(lambdas), d_off, d_aux, d_dist)
    assert rc == 0 and rig.gpu.hmr_gpu_sync(rig.ctx) == 0, rig.gpu.hmr_gpu_last_error()
    e_off, e_aux, e_dist = np.zeros((n_ctu, 3, 5, 32), np.int32), np.zeros((n_ctu, 3, 5), np.int32), np.zeros((n_ctu, 3, 5), np.int64)
    for c in range(n_ctu):
        oracle.ora_sao_offsets_ctu(at(stats, c * 960), at(lambdas, c * 3), at(e_off, c * 480), at(e_aux, c * 15), at(e_dist, c * 15))
    same(rig.down(d_off, (n_ctu, 3, 5, 32), np.int32), e_off, "offsets")
    same(rig.down(d_aux, (n_ctu, 3, 5), np.int32), e_aux, "band positions")
    same(rig.down(d_dist, (n_ctu, 3, 5), np.int64), e_dist, "distortions")
    assert (e_off != 0).sum() > 1000 and len(set(e_aux[:, :, 4].ravel().tolist())) > 20


@pytest.mark.parametrize("n", [16, 32])
def test_inter_tu_chain_wide_residuals(rig, oracle, n):
    """Residuals beyond what 8-bit pictures produce (up to +-32767 in some regions): the contract is the full int16 range, saturating packs included."""
    rng = np.random.default_rng(n + 7 + 1000 * SEED)
    nj = rig.nj
    amp = np.repeat(np.repeat(rng.choice([3, 300, 16383, 16384, 20000, 32767], (PH // 32, PW // 32)), 32, 0), 32, 1)
    rig.host[rig.res:rig.mid] = np.clip(rng.integers(-32768, 32768, (PH, PW)) * amp // 32768, -32768, 32767).astype(np.int16).ravel()
    jb = np.zeros(nj, INTER_TU_JOB)
    jb["orig_off"] = rig.block(rng, rig.res, n, n); jb["orig_stride"] = PW
    jb["pred_off"] = rig.block(rng, rig.pix, n, n); jb["pred_stride"] = PW
    jb["rec_off"] = rig.slots(rig.out1); jb["rec_stride"] = 80
    jb["lev_off"] = rig.slots(rig.out2)
    per, rem = rng.integers(2, 9, nj), rng.integers(0, 6, nj)
    jb["p0"] = 3 | (1 << 6); jb["p1"] = per | (rem << 8)
    jb["weight"] = 1.0; jb["zero_thr"] = 1.0
    d_ssd = rig.malloc(4 * nj); d_ac = rig.malloc(4 * nj); rig.bufs += [d_ssd, d_ac]
    g = rig.launch("hmr_gpu_inter_tu_chain_batch", rig.up(jb), nj, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
    o = rig.host.copy()
    ssd, ac = np.zeros(nj, np.uint32), np.zeros(nj, np.int32)
    oracle.ora_inter_tu_chain.restype = C.c_uint32
    for i, j in enumerate(jb):
        v = C.c_int(0)
        ssd[i] = oracle.ora_inter_tu_chain(at(o, j["orig_off"]), PW, at(o, j["pred_off"]), PW, at(o, j["lev_off"]), at(o, j["rec_off"]), 80, n, 3, 0, 0, 1, int(per[i]),
                                           int(rem[i]), C.c_double(1.0), C.c_double(1.0), C.byref(v))
        ac[i] = v.value
    same(g, o, "levels, reconstruction")
    same(rig.down(d_ssd, nj, np.uint32), ssd, "ssd")
    same(rig.down(d_ac, nj, np.int32), ac, "ac_sum")


class InterTuHost(C.Structure):
    _fields_ = [("residual", VP), ("residual_stride", C.c_int), ("pred", VP), ("pred_stride", C.c_int), ("levels", VP), ("recon", VP), ("recon_stride", C.c_int),
                ("size", C.c_int), ("scan_mode", C.c_int), ("comp", C.c_int), ("slice_is_intra", C.c_int), ("sign_hiding", C.c_int), ("per", C.c_int), ("rem", C.c_int),
                ("weight", C.c_double), ("zero_thr", C.c_double), ("ssd", C.c_uint32), ("ac_sum", C.c_int)]


def test_inter_tu_chain_n(oracle):
    """hmr_gpu_inter_tu_chain_n: the inter TUs of a CU's transform tree (luma 16 + chroma 8 parent level, luma 8 + chroma 4 child level: 15 TUs of three sizes) in one
    submission against the oracle TU by TU."""
    gpu = libs.load_gpu()
    rng = np.random.default_rng(77 + SEED)
    oracle.ora_inter_tu_chain.restype = C.c_uint32
    for trial in range(6):
        plan = [(16, 0), (8, 1), (8, 2)] + [(8, 0)] * 4 + [(4, 1)] * 4 + [(4, 2)] * 4 + ([(32, 0)] if trial % 2 else [])
        amp = [0, 2, 8, 30, 90, 250][trial]
        tus = (InterTuHost * len(plan))()
        keep, exp = [], []
        for i, (n, comp) in enumerate(plan):
            res = np.ascontiguousarray((amp * np.sin(np.arange(n * 40).reshape(n, 40) / 5.0) + rng.integers(-3, 4, (n, 40))).astype(np.int16))
            pred = np.ascontiguousarray(rng.integers(0, 256, (n, 48)).astype(np.int16))
            lev, rec = np.zeros(n * n, np.int16), np.zeros((n, 36), np.int16)
            per, rem, sbh = int(rng.integers(2, 7)), int(rng.integers(0, 6)), int(rng.integers(0, 2))
            w, thr = (1.0 if comp == 0 else 2.0 ** (int(rng.integers(-2, 5)) / 3.0)), float(np.clip(rng.uniform(0, 3000) / 2.5 - 5.0, 1.0, 20000.0))
            tus[i] = InterTuHost(VP(res.ctypes.data), 40, VP(pred.ctypes.data), 48, VP(lev.ctypes.data), VP(rec.ctypes.data), 36, n, 3, comp, 0, sbh, per, rem, w, thr, 0, 0)
            elev, erec, v = np.zeros(n * n, np.int16), np.zeros((n, 36), np.int16), C.c_int(0)
            essd = oracle.ora_inter_tu_chain(VP(res.ctypes.data), 40, VP(pred.ctypes.data), 48, VP(elev.ctypes.data), VP(erec.ctypes.data), 36, n, 3, comp, 0, sbh, per, rem,
                                             C.c_double(w), C.c_double(thr), C.byref(v))
            keep.append((res, pred, lev, rec)); exp.append((elev, erec, essd, v.value))
        gpu.hmr_gpu_inter_tu_chain_n.restype = None
        gpu.hmr_gpu_inter_tu_chain_n(tus, len(plan))
        for i, ((res, pred, lev, rec), (elev, erec, essd, eac)) in enumerate(zip(keep, exp)):
            same(lev, elev, f"levels of TU {i}"); same(rec, erec, f"reconstruction of TU {i}")
            assert (tus[i].ssd, tus[i].ac_sum) == (essd, eac), (trial, i, tus[i].ssd, tus[i].ac_sum, essd, eac)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_inter_tu_chain_from_source(rig, oracle, n):
    """Inter TU jobs that address the SOURCE block (HMR_GPU_INTER_TU_FROM_SOURCE): the kernel forms the residual the reference's `predict` call would have
    written (source - prediction, 16-bit wrap) and must leave what predict + the inter TU chain leave."""
    rng = np.random.default_rng(n + 23 + 1000 * SEED)
    nj = rig.nj
    yy, xx = np.mgrid[0:PH, 0:PW]
    amp = np.repeat(np.repeat(rng.choice([0, 2, 5, 15, 60], (PH // 32, PW // 32)), 32, 0), 32, 1)
    src = np.clip(128 + 50 * np.sin(xx / 9.0 + yy / 13.0), 0, 255)
    rig.host[rig.pix:rig.res] = src.astype(np.int16).ravel()                                                              # source picture
    rig.host[rig.res:rig.mid] = np.clip(src - amp * np.sin(xx / 5.0 + yy / 7.0) - rng.integers(-2, 3, (PH, PW)) * (amp > 0), 0, 255).astype(np.int16).ravel()   # prediction
    jb = np.zeros(nj, INTER_TU_JOB)
    x = rng.integers(0, PW - n + 1, nj); y = rng.integers(0, PH - n + 1, nj)
    jb["orig_off"] = rig.pix + y * PW + x; jb["orig_stride"] = PW
    jb["pred_off"] = rig.res + y * PW + x; jb["pred_stride"] = PW
    jb["rec_off"] = rig.slots(rig.out1); jb["rec_stride"] = 80
    jb["lev_off"] = rig.slots(rig.out2)
    jb["reserved"] = 1
    comp = rng.integers(0, 3 if n < 32 else 1, nj); sbh = rng.integers(0, 2, nj)
    per, rem = rng.integers(2, 7, nj), rng.integers(0, 6, nj)
    jb["p0"] = 3 | (comp << 2) | (sbh << 6); jb["p1"] = per | (rem << 8)
    jb["weight"] = np.where(comp == 0, 1.0, 2.0 ** (rng.integers(-2, 5, nj) / 3.0))
    jb["zero_thr"] = np.clip(rng.uniform(0, 3000, nj) / 2.5 - 5.0, 1.0, 20000.0)
    d_ssd = rig.malloc(4 * nj); d_ac = rig.malloc(4 * nj); rig.bufs += [d_ssd, d_ac]
    g = rig.launch("hmr_gpu_inter_tu_chain_batch", rig.up(jb), nj, n, rig.dev, rig.dev, rig.dev, rig.dev, d_ssd, d_ac)
    o = rig.host.copy()
    ssd, ac = np.zeros(nj, np.uint32), np.zeros(nj, np.int32)
    oracle.ora_inter_tu_chain.restype = C.c_uint32
    res = np.zeros((n, n), np.int16)
    for i, j in enumerate(jb):
        v = C.c_int(0)
        oracle.ora_predict(at(o, j["orig_off"]), PW, at(o, j["pred_off"]), PW, VP(res.ctypes.data), n, n)
        ssd[i] = oracle.ora_inter_tu_chain(VP(res.ctypes.data), n, at(o, j["pred_off"]), PW, at(o, j["lev_off"]), at(o, j["rec_off"]), 80, n, 3, int(comp[i]), 0,
                                           int(sbh[i]), int(per[i]), int(rem[i]), C.c_double(float(j["weight"])), C.c_double(float(j["zero_thr"])), C.byref(v))
        ac[i] = v.value
    same(g, o, "levels, reconstruction")
    same(rig.down(d_ssd, nj, np.uint32), ssd, "ssd")
    same(rig.down(d_ac, nj, np.int32), ac, "ac_sum")
    assert (ac != 0).sum() > 20 and ((ac == 0) & (ssd != 0)).sum() > 5
