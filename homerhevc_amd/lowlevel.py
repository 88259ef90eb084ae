"""ctypes mirror of the reference's plug-in surface, `struct low_level_funcs_t` (hmr_private.h:1063-1092).

`LowLevelFuncs` exposes the table's member names (sad, ssd16b, predict, reconst, transform, quant, ...) bound
to the drop-in entries of libhomer_gpu.so, which take HOST pointers and are synchronous exactly like the
SSE4.2 functions the reference stores there (hmr_encoder_lib.c:159-187).  Members whose reference signature
starts with `henc_thread_t*` take the scalars that struct is read for instead (see include/homer_gpu.h).
"""
import ctypes as C
import os

from .build import LIB_PATH

_I16P = C.c_void_p


class NativeLibraryMissing(RuntimeError):
    pass


def load_library():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(f"{LIB_PATH} not built: run __graft_entry__.build() (there is no CPU fallback)")
    return C.CDLL(LIB_PATH)


# member name in low_level_funcs_t -> (exported symbol, restype, argtypes)
TABLE = {
    "sse_copy_16_16": ("hmr_gpu_copy_16_16", None, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "sse_copy_16_8": ("hmr_gpu_copy_16_8", None, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "sse_copy_8_16": ("hmr_gpu_copy_8_16", None, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "sad": ("hmr_gpu_sad", C.c_uint32, [_I16P, C.c_uint32, _I16P, C.c_uint32, C.c_int]),
    "ssd16b": ("hmr_gpu_ssd16b", C.c_uint32, [_I16P, C.c_uint32, _I16P, C.c_uint32, C.c_int]),
    "predict": ("hmr_gpu_predict", None, [_I16P, C.c_int, _I16P, C.c_int, _I16P, C.c_int, C.c_int]),
    "reconst": ("hmr_gpu_reconst", None, [_I16P, C.c_int, _I16P, C.c_int, _I16P, C.c_int, C.c_int]),
    "modified_variance": ("hmr_gpu_modified_variance", C.c_uint32, [_I16P, C.c_int, C.c_int, C.c_int]),
    "create_intra_planar_prediction": ("hmr_gpu_intra_planar", None, [_I16P, C.c_int, _I16P, C.c_int, C.c_int]),
    "create_intra_angular_prediction": ("hmr_gpu_intra_angular", None, [_I16P, C.c_int, _I16P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "interpolate_luma_m_compensation": ("hmr_gpu_interpolate_luma", None, [_I16P, C.c_int, _I16P, C.c_int] + [C.c_int] * 6),
    "interpolate_chroma_m_compensation": ("hmr_gpu_interpolate_chroma", None, [_I16P, C.c_int, _I16P, C.c_int] + [C.c_int] * 6),
    "interpolate_luma_m_estimation": ("hmr_gpu_interpolate_luma", None, [_I16P, C.c_int, _I16P, C.c_int] + [C.c_int] * 6),
    "weighted_average_motion": ("hmr_gpu_weighted_average", None, [_I16P, C.c_int, _I16P, C.c_int, _I16P, C.c_int, C.c_int, C.c_int]),
    "quant": ("hmr_gpu_quant", None, [_I16P, _I16P, _I16P] + [C.c_int] * 6 + [C.POINTER(C.c_int)] + [C.c_int] * 3),
    "inv_quant": ("hmr_gpu_inv_quant", None, [_I16P, _I16P] + [C.c_int] * 6),
    "transform": ("hmr_gpu_transform", None, [_I16P, _I16P, C.c_int, C.c_int, C.c_int]),
    "itransform": ("hmr_gpu_itransform", None, [_I16P, _I16P, C.c_int, C.c_int, C.c_int]),
    "get_sao_stats": ("hmr_gpu_get_sao_stats", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}


# functions the reference calls directly (no table slot): reference symbol -> (exported symbol, restype).  A C host interposes them by symbol
# (INTEGRATION.md); argument lists are in include/homer_gpu.h.
OFF_TABLE = {
    "fill_reference_samples": ("hmr_gpu_fill_reference_samples", None),
    "adi_filter": ("hmr_gpu_adi_filter", None),
    "homer_loop1_motion_intra": ("hmr_gpu_intra_search", None),
    "encode_intra_cu": ("hmr_gpu_intra_tu_chain", C.c_uint32),
    "encode_intra_luma": ("hmr_gpu_intra_luma_cu", None),
    "encode_intra_chroma": ("hmr_gpu_intra_chroma_cu", None),
    "encode_inter_cu": ("hmr_gpu_inter_tu_chain", C.c_uint32),
    "encode_inter": ("hmr_gpu_inter_tu_chain_n", None),
    "hmr_motion_estimation": ("hmr_gpu_motion_estimation", C.c_uint32),
    "hmr_motion_compensation_luma": ("hmr_gpu_mc_luma", None),
    "hmr_motion_compensation_chroma": ("hmr_gpu_mc_chroma", None),
    "hmr_deblock_filter_cu": ("hmr_gpu_deblock_filter_ctu", None),
    "sao_offset_ctu": ("hmr_gpu_sao_offset_ctu", None),
    "sao_derive_offsets": ("hmr_gpu_sao_offsets_ctu", None),
    "reference_picture_border_padding_ctu": ("hmr_gpu_pad_ctu", None),
}


def _pointer_safe(f):
    """The off-table entries have long mixed argument lists (include/homer_gpu.h) and no argtypes here: without them ctypes passes a bare Python
    int as a C int, which would cut a 64-bit address to 32 bits.  Addresses given as plain ints are therefore wrapped as pointers."""
    def call(*args):
        return f(*[C.c_void_p(a) if isinstance(a, int) and not isinstance(a, bool) and (a > 0x7fffffff or a < -0x80000000) else a for a in args])
    call.__name__ = getattr(f, "__name__", "off_table_entry")
    return call


class LowLevelFuncs:
    """The function table, populated the way HOMER_enc_init populates hvenc->funcs; the off-table functions are attributes under the
    reference's own symbol names."""

    def __init__(self, lib=None):
        self.lib = lib or load_library()
        for member, (sym, restype, argtypes) in TABLE.items():
            f = getattr(self.lib, sym)
            f.restype = restype
            f.argtypes = argtypes
            setattr(self, member, f)
        for name, (sym, restype) in OFF_TABLE.items():
            f = getattr(self.lib, sym)
            f.restype = restype
            setattr(self, name, _pointer_safe(f))
