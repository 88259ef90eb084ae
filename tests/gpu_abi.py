"""Test-side ctypes mirror of the batched / frame-level C ABI (include/homer_gpu.h layers 1, 3, 4): descriptor layouts and a thin driver for the kernel tests.
Not part of the product package.

Device memory is owned by the caller (torch tensors in bench.py, hmr_gpu_malloc in tests); this module only
marshals pointers.  Everything here fails loudly when the native library or the GPU is missing.
"""
import ctypes as C

import numpy as np

import libs


def load_library():
    return libs.load_gpu()


JOB_DTYPE = np.dtype([("a_off", "<u4"), ("a_stride", "<u4"), ("b_off", "<u4"), ("b_stride", "<u4"), ("c_off", "<u4"), ("c_stride", "<u4"),
                      ("w", "<u2"), ("h", "<u2"), ("p0", "<u4"), ("p1", "<u4")])
assert JOB_DTYPE.itemsize == 36
# the other descriptors of include/homer_gpu.h
TU_JOB_DTYPE = np.dtype([("orig_off", "<u4"), ("orig_stride", "<u4"), ("pred_off", "<u4"), ("pred_stride", "<u4"), ("rec_off", "<u4"), ("rec_stride", "<u4"),
                         ("lev_off", "<u4"), ("p0", "<u4"), ("p1", "<u4")])                                                   # hmr_gpu_tu_job
ITU_JOB_DTYPE = np.dtype(TU_JOB_DTYPE.descr + [("dec_off", "<u4"), ("dec_stride", "<u4"), ("flags", "<u4"), ("sizes", "<u4"), ("mode", "<u4")])   # hmr_gpu_itu_job
INTER_TU_JOB_DTYPE = np.dtype(TU_JOB_DTYPE.descr + [("reserved", "<u4"), ("weight", "<f8"), ("zero_thr", "<f8")])             # hmr_gpu_inter_tu_job
ME_JOB_DTYPE = np.dtype([("corr", "<f8"), ("orig_off", "<u4"), ("orig_stride", "<u4"), ("ref_off", "<u4"), ("ref_stride", "<u4"), ("gx", "<i2"), ("gy", "<i2"),
                         ("init_x", "<i2"), ("init_y", "<i2"), ("n_amvp", "<i2"), ("n_search", "<i2"), ("amvp", "<i2", (2, 2)), ("search", "<i2", (5, 2)),
                         ("action", "<u4"), ("reserved", "<u4")])                                                             # hmr_gpu_me_job
INTRA_JOB_DTYPE = np.dtype([("sqrt_lambda", "<f8"), ("orig_off", "<u4"), ("orig_stride", "<u4"), ("dec_off", "<u4"), ("dec_stride", "<u4"), ("adi_off", "<u4"),
                            ("adif_off", "<u4"), ("pred_off", "<u4"), ("pred_stride", "<u4"), ("flags", "<u4"), ("sizes", "<u4"), ("preds", "<i4", (3,)),
                            ("pred_bits", "<u4", (3,)), ("other_bits", "<u4"), ("reserved", "<u4")])                          # hmr_gpu_intra_job
TREE_JOB_DTYPE = np.dtype([("parent", "<u4"), ("child", "<u4", (4,)), ("par_rec_off", "<u4"), ("par_rec_stride", "<u4"), ("chl_rec_off", "<u4"),
                           ("chl_rec_stride", "<u4"), ("par_lev_off", "<u4"), ("chl_lev_off", "<u4"), ("size", "<u4"), ("rule", "<u4")])   # hmr_gpu_tree_job
TREE_RESULT_DTYPE = np.dtype([("split", "<u4"), ("cost", "<u4"), ("sum", "<u4"), ("cbf", "u1", (4,))])                            # hmr_gpu_tree_result
INTRA_RESULT_DTYPE = np.dtype([("best_mode", "<i4"), ("bits", "<i4"), ("cost", "<f8")])                                          # hmr_gpu_intra_result
CHROMA_JOB_DTYPE = np.dtype([("sqrt_lambda", "<f8"), ("orig_u_off", "<u4"), ("orig_v_off", "<u4"), ("orig_stride", "<u4"), ("dec_u_off", "<u4"), ("dec_v_off", "<u4"),
                             ("dec_stride", "<u4"), ("flags", "<u4"), ("sizes", "<u4"), ("luma_mode", "<u4"), ("reserved", "<u4")])   # hmr_gpu_chroma_job
assert CHROMA_JOB_DTYPE.itemsize == 48
ITU_MODE_FROM_SEARCH = 0x100
TREE_NO_PARENT = 0xFFFFFFFF
assert (TREE_JOB_DTYPE.itemsize, TREE_RESULT_DTYPE.itemsize, INTRA_RESULT_DTYPE.itemsize) == (52, 16, 16)
assert (TU_JOB_DTYPE.itemsize, ITU_JOB_DTYPE.itemsize, INTER_TU_JOB_DTYPE.itemsize, ME_JOB_DTYPE.itemsize, INTRA_JOB_DTYPE.itemsize) == (36, 56, 56, 72, 80)


class Frame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("stride_y", C.c_int), ("stride_c", C.c_int)]


class Units(C.Structure):
    _fields_ = [("units_stride", C.c_int), ("mvx", C.c_void_p), ("mvy", C.c_void_p), ("ref_idx", C.c_void_p), ("qp", C.c_void_p), ("flags", C.c_void_p)]


class GpuError(RuntimeError):
    pass


class Context:
    """One context per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=0, stream=None, lib=None):
        self.lib = lib or load_library()
        self.lib.hmr_gpu_last_error.restype = C.c_char_p
        self.ctx = C.c_void_p()
        rc = self.lib.hmr_gpu_create(C.byref(self.ctx), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise GpuError(f"hmr_gpu_create failed ({rc}): {self.lib.hmr_gpu_last_error().decode()}")

    def check(self, rc, what):
        if rc != 0:
            raise GpuError(f"{what} failed ({rc}): {self.lib.hmr_gpu_last_error().decode()}")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.ctx, *args), name)

    def sync(self):
        self.call("hmr_gpu_sync")

    def timer_start(self):
        self.call("hmr_gpu_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self.call("hmr_gpu_timer_stop", C.byref(ms))
        return ms.value

    def event(self):
        ev = C.c_void_p()
        self.call("hmr_gpu_event_create", C.byref(ev))
        return ev

    def record(self, ev):
        self.call("hmr_gpu_event_record", ev)

    def elapsed(self, ev0, ev1):
        ms = C.c_float()
        self.check(self.lib.hmr_gpu_event_elapsed(ev0, ev1, C.byref(ms)), "hmr_gpu_event_elapsed")
        return ms.value

    def close(self):
        if self.ctx:
            self.lib.hmr_gpu_destroy(self.ctx)
            self.ctx = None
