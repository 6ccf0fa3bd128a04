"""ctypes binding of libcuking_amd.so (the C ABI of include/cuking_amd.h).

There is no fallback: if the library has not been built, importing the binding
raises, and every device entry point fails without a gfx950 GPU.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libcuking_amd.so"

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_FAILED_PRECONDITION = 2
ERR_RESOURCE_EXHAUSTED = 3
ERR_OUT_OF_MEMORY = 4
ERR_DEVICE = 5

KERNEL_TILED = 0
KERNEL_STREAM = 1


class CukingError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"cuking_amd status {status}: {message}")
        self.status = status
        self.message = message


class CSubmatrix(C.Structure):
    """cuking_submatrix (cuking.cu:129-179)."""
    _fields_ = [("i_begin", C.c_uint32), ("i_end", C.c_uint32),
                ("j_begin", C.c_uint32), ("j_end", C.c_uint32)]


_u32, _u64, _i64, _f32 = C.c_uint32, C.c_uint64, C.c_int64, C.c_float
_vp, _sz, _int = C.c_void_p, C.c_size_t, C.c_int
_SM = C.POINTER(CSubmatrix)

# name -> (restype, argtypes); every symbol include/cuking_amd.h declares.
SIGNATURES = {
    "cuking_submatrix_init": (_int, [_SM, _u32, _u32, _u32]),
    "cuking_submatrix_num_rows": (_u32, [_SM]),
    "cuking_submatrix_num_cols": (_u32, [_SM]),
    "cuking_submatrix_num_samples": (_u32, [_SM]),
    "cuking_submatrix_contains": (_u32, [_SM, _u32]),
    "cuking_submatrix_sample_offset": (_u32, [_SM, _u32]),
    "cuking_submatrix_num_pairs": (_u64, [_SM]),
    "cuking_padded_sites": (_u32, [_u32]),
    "cuking_words_per_sample": (_u32, [_u32]),
    "cuking_bytes_per_pair": (_u64, [_u32]),
    "cuking_pack_host": (_int, [_SM, _u32, _vp, _vp, _vp, _vp, _sz]),
    "cuking_narrow_triples": (_int, [_SM, _u32, _vp, _vp, _vp, _sz, _vp, _vp,
                                     C.POINTER(_sz)]),
    "cuking_schedule_tile_partition": (None, [_u64, _u32, _vp]),
    "cuking_schedule_weighted_tile_partition": (_int, [_u64, _vp, _u32, _vp]),
    "cuking_schedule_calibration_tiles": (_u64, [_u64, _u32]),
    "cuking_schedule_chunk_ranges": (_u32, [_u32, _u32, _u32, _vp]),
    "cuking_schedule_staged_steps": (_u32, [_u32, _u32, _u32, _u32, _u32, _vp]),
    "cuking_last_error": (C.c_char_p, []),
    "cuking_abi_version": (_u32, []),
    "cuking_device_count": (_int, []),
    "cuking_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "cuking_ctx_destroy": (None, [_vp]),
    "cuking_device_alloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "cuking_device_free": (_int, [_vp, _vp]),
    "cuking_memset_async": (_int, [_vp, _vp, _int, _sz, _vp]),
    "cuking_copy_to_device": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "cuking_copy_to_host": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "cuking_stream_synchronize": (_int, [_vp, _vp]),
    "cuking_stream_create": (_int, [_vp, C.POINTER(_vp)]),
    "cuking_stream_destroy": (_int, [_vp, _vp]),
    "cuking_event_create": (_int, [_vp, C.POINTER(_vp)]),
    "cuking_event_record": (_int, [_vp, _vp, _vp]),
    "cuking_event_synchronize": (_int, [_vp, _vp]),
    "cuking_event_destroy": (_int, [_vp, _vp]),
    "cuking_host_alloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "cuking_host_free": (_int, [_vp, _vp]),
    "cuking_pack_device": (_int, [_vp, _SM, _u32, _vp, _vp, _vp, _vp, _sz,
                                  _vp, _vp]),
    "cuking_pack_device_compact": (_int, [_vp, _SM, _u32, _vp, _vp, _vp, _sz, _vp,
                                          _vp]),
    "cuking_ctx_set_kernel": (_int, [_vp, _int]),
    "cuking_ctx_set_option": (_int, [_vp, C.c_char_p, _i64]),
    "cuking_ctx_get_option": (_int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "cuking_num_variants": (_int, []),
    "cuking_variant_name": (C.c_char_p, [_int]),
    "cuking_compute_king": (_int, [_vp, _SM, _u32, _vp, _f32, _u32, _vp, _vp,
                                   _vp, _vp]),
    "cuking_num_tiles": (_u64, [_vp, _SM]),
    "cuking_tile_samples": (_u32, [_vp]),
    "cuking_tile_bounds": (_int, [_vp, _SM, _u64, C.POINTER(_u32),
                                  C.POINTER(_u32), C.POINTER(_u32),
                                  C.POINTER(_u32)]),
    "cuking_compute_king_tiles": (_int, [_vp, _SM, _u32, _vp, _u64, _u64, _f32,
                                         _u32, _vp, _vp, _vp, _vp]),
    "cuking_prepare_samples": (_int, [_vp, _SM, _u32, _vp, _u32, _u32, _vp]),
    "cuking_compute_king_rect": (_int, [_vp, _SM, _u32, _vp, _u32, _u32, _u32,
                                        _u32, _u32, _f32, _u32, _vp, _vp, _vp,
                                        _vp]),
    "cuking_ctx_reserve": (_int, [_vp, _SM, _u32, C.POINTER(_vp), _sz]),
    "cuking_invalidate": (_int, [_vp]),
    "cuking_compute_counts": (_int, [_vp, _SM, _u32, _vp, _vp, _vp]),
    "cuking_sort_results": (None, [_vp, _sz]),
    "cuking_timing_enable": (_int, [_vp, _int]),
    "cuking_timing_reset": (_int, [_vp]),
    "cuking_timing_collect": (_int, [_vp, C.POINTER(C.c_double),
                                     C.POINTER(_u64), C.POINTER(C.c_double),
                                     C.POINTER(_u64)]),
    "cuking_clock_probe": (_int, [_vp, _u64, _vp, _vp]),
    "cuking_synth_bitset": (_int, [_vp, _u64, _vp, _vp, _vp, _u32, _u32, _u32,
                                   _u32, _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the HIP library; raises if it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        # A source-only checkout: build the HIP library in-tree (hipcc cross-
        # compiles gfx950 without a GPU).  This is a build step, not a fallback:
        # without hipcc there is nothing to run and the import fails.
        try:
            from . import build as _build
            _build.build_library()
        except Exception as e:  # noqa: BLE001
            raise ImportError(
                f"{LIB_PATH} is missing and could not be built ({e}): run "
                "`python -m cuking_amd.build` (hipcc, gfx950).  cuking_amd has no "
                "CPU fallback.") from e
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  A
    # process must hold exactly one HIP runtime, so when torch is going to be
    # used for device memory it has to be loaded first; our library then binds
    # to that copy.  (A host without torch, e.g. the C++ CLI, gets /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.cuking_abi_version() != 2:
        raise ImportError("libcuking_amd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != OK:
        raise CukingError(status, load().cuking_last_error().decode())
