"""Command lists (include/homer_gpu.h, hmr_gpu_cmdlist_*): a fixed sequence of batched calls described once, run eagerly, and captured into a hipGraph and replayed -
every way of running the list must leave what the direct calls leave, and that is what the CPU oracle computes."""
import ctypes as C

import numpy as np
import pytest

import libs
from test_gpu_batches import JOB, PW, Rig, VP, at, same

pytestmark = pytest.mark.gpu


class Cmd(C.Structure):          # hmr_gpu_cmd
    _fields_ = [("op", C.c_int), ("njobs", C.c_int), ("size", C.c_int), ("p", C.c_int * 4), ("jobs", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p),
                ("out", C.c_void_p), ("p64", C.c_void_p * 3), ("branch", C.c_int)]


OP_SAD, OP_SSD16B, OP_PREDICT = 1, 2, 3


@pytest.fixture
def rig():
    gpu = libs.load_gpu()
    r = Rig(gpu, np.random.default_rng(4242))
    yield r
    r.close()


@pytest.mark.parametrize("branches", [False, True], ids=["one-stream", "two-branches"])
def test_command_list_eager_and_graph_replay(rig, oracle, branches):
    gpu, n = rig.gpu, 16
    rng = np.random.default_rng(9)
    jobs = []
    for base_b in (rig.pix, rig.res, rig.pix):
        jb = np.zeros(rig.nj, JOB)
        jb["a_off"] = rig.block(rng, rig.pix, n, n); jb["a_stride"] = PW
        jb["b_off"] = rig.block(rng, base_b, n, n); jb["b_stride"] = PW
        jb["c_off"] = rig.slots(rig.out1); jb["c_stride"] = n
        jobs.append(jb)
    d_jobs = [rig.up(j) for j in jobs]
    d_sad, d_ssd = rig.malloc(4 * rig.nj), rig.malloc(4 * rig.nj)
    rig.bufs += [d_sad, d_ssd]
    cmds = (Cmd * 3)()
    for k, (op, out) in enumerate(((OP_SAD, d_sad), (OP_SSD16B, d_ssd), (OP_PREDICT, None))):
        cmds[k].op, cmds[k].njobs, cmds[k].size = op, rig.nj, n
        cmds[k].jobs, cmds[k].a, cmds[k].b, cmds[k].c, cmds[k].out = d_jobs[k].value, rig.dev.value, rig.dev.value, rig.dev.value, out.value if out else None
        cmds[k].branch = k if branches else 0        # (the three commands are independent of each other: as graph branches they may overlap)
    lst = VP()
    assert gpu.hmr_gpu_cmdlist_create(rig.ctx, cmds, 3, C.byref(lst)) == 0, gpu.hmr_gpu_last_error()
    f_sad = oracle.ora_sad; f_sad.restype = C.c_uint32
    f_ssd = oracle.ora_ssd16b; f_ssd.restype = C.c_uint32
    exp_sad = np.array([f_sad(at(rig.host, j["a_off"]), PW, at(rig.host, j["b_off"]), PW, n) for j in jobs[0]], np.uint32)
    exp_ssd = np.array([f_ssd(at(rig.host, j["a_off"]), PW, at(rig.host, j["b_off"]), PW, n) for j in jobs[1]], np.uint32)
    exp = rig.host.copy()
    for j in jobs[2]:
        a = rig.host[int(j["a_off"]):].reshape(-1)[:(n - 1) * PW + n]
        b = rig.host[int(j["b_off"]):].reshape(-1)[:(n - 1) * PW + n]
        for y in range(n):
            exp[int(j["c_off"]) + y * n:int(j["c_off"]) + y * n + n] = a[y * PW:y * PW + n] - b[y * PW:y * PW + n]

    def check(run, what):
        assert gpu.hmr_gpu_upload(rig.ctx, rig.dev, VP(rig.host.ctypes.data), C.c_size_t(rig.size * 2)) == 0
        for d in (d_sad, d_ssd):
            z = np.zeros(rig.nj, np.uint32)
            assert gpu.hmr_gpu_upload(rig.ctx, d, VP(z.ctypes.data), C.c_size_t(z.nbytes)) == 0
        assert run() == 0, (what, gpu.hmr_gpu_last_error())
        assert gpu.hmr_gpu_sync(rig.ctx) == 0
        same(rig.down(d_sad, rig.nj, np.uint32), exp_sad, what + ": sad")
        same(rig.down(d_ssd, rig.nj, np.uint32), exp_ssd, what + ": ssd16b")
        same(rig.down(rig.dev, rig.size, np.int16), exp, what + ": predict / arena")

    check(lambda: gpu.hmr_gpu_cmdlist_run(rig.ctx, lst, None), "eager run")
    assert gpu.hmr_gpu_cmdlist_capture(rig.ctx, lst) == 0, gpu.hmr_gpu_last_error()
    check(lambda: gpu.hmr_gpu_cmdlist_replay(rig.ctx, lst), "graph replay")
    check(lambda: gpu.hmr_gpu_cmdlist_replay(rig.ctx, lst), "second graph replay")
    gpu.hmr_gpu_cmdlist_destroy(lst)
