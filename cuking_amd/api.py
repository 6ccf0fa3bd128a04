"""Host-side mirror of the reference's interface for the KING hot path.

Names follow cuking.cu: ``Submatrix`` (:129-179) with ``NumRows / NumCols /
NumSamples / Contains / SampleOffset``, ``KingResult`` records (:182-186), and
``KingContext.compute_king`` taking exactly the arguments of
``ComputeKingKernel`` (:191-195).  Everything is executed by libcuking_amd.so
through the C ABI (include/cuking_amd.h); torch only provides device memory
and the stream handle.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import CSubmatrix, CukingError, check

# cuking.cu:182-186
KING_RESULT_DTYPE = np.dtype(
    [("sample_i", "<u4"), ("sample_j", "<u4"), ("kin", "<f4"),
     ("ibs0", "<u4"), ("ibs1", "<u4"), ("ibs2", "<u4")])
KING_COUNTS_DTYPE = np.dtype(
    [("het_i", "<u4"), ("het_j", "<u4"), ("both_het", "<u4"),
     ("opposing_hom", "<u4"), ("concordant_hom", "<u4"), ("shared", "<u4")])

DEFAULT_KIN_THRESHOLD = 0.0884   # cuking.cu:43
DEFAULT_MAX_RESULTS = 10 << 20   # cuking.cu:40


class ResourceExhaustedError(RuntimeError):
    """cuking.cu:747-751."""


class Submatrix:
    """cuking.cu:129-179: block (block_i <= block_j) of the relatedness matrix
    selected by ``shard_index`` out of ``split_factor*(split_factor+1)/2``."""

    def __init__(self, num_samples: int, split_factor: int = 1,
                 shard_index: int = 0):
        self.c = CSubmatrix()
        check(_lib.load().cuking_submatrix_init(
            C.byref(self.c), num_samples, split_factor, shard_index))

    @classmethod
    def from_ranges(cls, i_begin, i_end, j_begin, j_end) -> "Submatrix":
        self = cls.__new__(cls)
        self.c = CSubmatrix(i_begin, i_end, j_begin, j_end)
        return self

    i_begin = property(lambda s: s.c.i_begin)
    i_end = property(lambda s: s.c.i_end)
    j_begin = property(lambda s: s.c.j_begin)
    j_end = property(lambda s: s.c.j_end)

    def NumRows(self) -> int:
        return _lib.load().cuking_submatrix_num_rows(C.byref(self.c))

    def NumCols(self) -> int:
        return _lib.load().cuking_submatrix_num_cols(C.byref(self.c))

    def NumSamples(self) -> int:
        return _lib.load().cuking_submatrix_num_samples(C.byref(self.c))

    def Contains(self, index: int) -> bool:
        return bool(_lib.load().cuking_submatrix_contains(C.byref(self.c), index))

    def SampleOffset(self, index: int) -> int:
        return _lib.load().cuking_submatrix_sample_offset(C.byref(self.c), index)

    def NumPairs(self) -> int:
        return _lib.load().cuking_submatrix_num_pairs(C.byref(self.c))

    def as_tuple(self):
        return (self.i_begin, self.i_end, self.j_begin, self.j_end)

    def __repr__(self):
        return "Submatrix(i=[%d,%d), j=[%d,%d))" % self.as_tuple()


def padded_sites(num_sites: int) -> int:
    return _lib.load().cuking_padded_sites(num_sites)


def words_per_sample(num_sites: int) -> int:
    return _lib.load().cuking_words_per_sample(num_sites)


def bytes_per_pair(wps: int) -> int:
    return _lib.load().cuking_bytes_per_pair(wps)


def new_host_bitset(sm: Submatrix, num_sites: int) -> np.ndarray:
    """All-missing host bitset (cuking.cu:513-523)."""
    return np.full((sm.NumSamples(), words_per_sample(num_sites)),
                   np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)


def pack_host(sm: Submatrix, bit_set: np.ndarray, row_idx, col_idx,
              n_alt_alleles) -> None:
    """cuking.cu:675-703 on host memory (thread-safe relaxed atomics)."""
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int64)
    n_alt = np.ascontiguousarray(n_alt_alleles, dtype=np.int32)
    assert bit_set.dtype == np.uint64 and bit_set.flags.c_contiguous
    assert row_idx.size == col_idx.size == n_alt.size
    check(_lib.load().cuking_pack_host(
        C.byref(sm.c), bit_set.shape[1], bit_set.ctypes.data,
        row_idx.ctypes.data, col_idx.ctypes.data, n_alt.ctypes.data,
        row_idx.size))


def sort_results(results: np.ndarray) -> np.ndarray:
    """cuking.cu:761-765 (in place)."""
    assert results.dtype == KING_RESULT_DTYPE and results.flags.c_contiguous
    _lib.load().cuking_sort_results(results.ctypes.data, results.size)
    return results


def device_count() -> int:
    return _lib.load().cuking_device_count()


@dataclass
class KernelTiming:
    king_ms: float
    king_launches: int
    prepare_ms: float
    prepare_launches: int


def _stream_handle(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


class KingContext:
    """One per GPU (cuking_ctx).  Device buffers are torch tensors on that GPU;
    bitsets are int64 tensors holding the reference's uint64 words."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        self.device = device
        h = C.c_void_p()
        check(self.lib.cuking_ctx_create(device, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cuking_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- options ------------------------------------------------------------
    def set_kernel(self, name: str) -> None:
        kernel = {"tiled": _lib.KERNEL_TILED, "stream": _lib.KERNEL_STREAM}[name]
        check(self.lib.cuking_ctx_set_kernel(self.handle, kernel))

    def set_option(self, key: str, value: int) -> None:
        check(self.lib.cuking_ctx_set_option(self.handle, key.encode(), value))

    def get_option(self, key: str) -> int:
        value = C.c_int64(0)
        check(self.lib.cuking_ctx_get_option(self.handle, key.encode(), C.byref(value)))
        return int(value.value)

    def variant_name(self, variant: int | None = None) -> str:
        v = self.get_option("variant") if variant is None else variant
        return self.lib.cuking_variant_name(v).decode()

    def num_tiles(self, sm: Submatrix) -> int:
        return self.lib.cuking_num_tiles(self.handle, C.byref(sm.c))

    def tile_samples(self) -> int:
        return self.lib.cuking_tile_samples(self.handle)

    # -- memory -------------------------------------------------------------
    def _tensor(self, shape, dtype):
        import torch
        return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")

    def upload_bitset(self, host_bits: np.ndarray):
        import torch
        t = torch.from_numpy(host_bits.view(np.int64))
        return t.to(f"cuda:{self.device}", non_blocking=False)

    # -- hot path -----------------------------------------------------------
    def compute_king(self, submatrix: Submatrix, words_per_sample: int,
                     bit_sets, kin_threshold: float, max_results: int, results,
                     result_index, result_overflow, stream=None,
                     tile_range=None) -> None:
        """ComputeKingKernel (cuking.cu:191-195) + its launch (:734-741):
        appends to ``results`` (device, [max_results, 6] int32 = KingResult
        records) and bumps ``result_index`` / ``result_overflow`` (device u32,
        not reset here).  Asynchronous on the stream."""
        self._check_bits(submatrix, words_per_sample, bit_sets)
        assert results.numel() * results.element_size() >= max_results * 24
        args = (self.handle, C.byref(submatrix.c), words_per_sample,
                bit_sets.data_ptr())
        tail = (kin_threshold, max_results, results.data_ptr(),
                result_index.data_ptr(), result_overflow.data_ptr(),
                _stream_handle(stream))
        if tile_range is None:
            check(self.lib.cuking_compute_king(*args, *tail))
        else:
            check(self.lib.cuking_compute_king_tiles(
                *args, tile_range[0], tile_range[1], *tail))

    def prepare_samples(self, submatrix: Submatrix, words_per_sample: int,
                        bit_sets, sample_begin: int, sample_end: int,
                        stream=None) -> None:
        """Staged operator, step 1 (diagonal block): convert samples
        [sample_begin, sample_end) into the kernel layout."""
        self._check_bits(submatrix, words_per_sample, bit_sets)
        check(self.lib.cuking_prepare_samples(
            self.handle, C.byref(submatrix.c), words_per_sample,
            bit_sets.data_ptr(), sample_begin, sample_end,
            _stream_handle(stream)))

    def compute_king_rect(self, submatrix: Submatrix, words_per_sample: int,
                          bit_sets, rows, cols, kin_threshold: float,
                          max_results: int, results, result_index,
                          result_overflow, stream=None) -> None:
        """Staged operator, step 2: pairs (i < j) of rows x cols from the
        prepared layout; appends like compute_king.  rows = (begin, end) or
        (begin, end, step): every tile row, or every (step / tile)-th one;
        cols = (begin, end); sample indices."""
        self._check_bits(submatrix, words_per_sample, bit_sets)
        assert results.numel() * results.element_size() >= max_results * 24
        step = rows[2] if len(rows) > 2 else 0
        check(self.lib.cuking_compute_king_rect(
            self.handle, C.byref(submatrix.c), words_per_sample,
            bit_sets.data_ptr(), rows[0], rows[1], step, cols[0], cols[1],
            kin_threshold, max_results, results.data_ptr(),
            result_index.data_ptr(), result_overflow.data_ptr(),
            _stream_handle(stream)))

    def reserve(self, submatrix: Submatrix, words_per_sample: int, streams=()) -> None:
        """Sizes the workspace for the block (and the split slabs of the
        streams named) up front: later compute / prepare calls on those streams
        neither allocate nor wait for the device."""
        handles = (C.c_void_p * max(len(streams), 1))(
            *[_stream_handle(s) for s in streams])
        check(self.lib.cuking_ctx_reserve(
            self.handle, C.byref(submatrix.c), words_per_sample, handles, len(streams)))

    def invalidate(self) -> None:
        """The bitset behind the pointer last converted has been rewritten in
        place (only matters with the option "reuse_prepared")."""
        check(self.lib.cuking_invalidate(self.handle))

    def compute_counts(self, submatrix: Submatrix, words_per_sample: int,
                       bit_sets, stream=None) -> np.ndarray:
        """Diagnostic: the six sums (cuking.cu:232-239) of every pair, as a
        [NumRows, NumCols] array (entries with i >= j are zero)."""
        import torch
        self._check_bits(submatrix, words_per_sample, bit_sets)
        r, c = submatrix.NumRows(), submatrix.NumCols()
        out = torch.zeros((r, c, 6), dtype=torch.int32,
                          device=f"cuda:{self.device}")
        check(self.lib.cuking_compute_counts(
            self.handle, C.byref(submatrix.c), words_per_sample,
            bit_sets.data_ptr(), out.data_ptr(), _stream_handle(stream)))
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy().view(np.uint32).reshape(r, c, 6).view(
            KING_COUNTS_DTYPE).reshape(r, c)

    def run(self, submatrix: Submatrix, words_per_sample: int, bit_sets,
            kin_threshold: float = DEFAULT_KIN_THRESHOLD,
            max_results: int = DEFAULT_MAX_RESULTS, tile_range=None,
            sort: bool = True) -> np.ndarray:
        """cuking.cu:713-765: allocate + zero the result buffer, launch, wait,
        raise on overflow, return the (sorted) host records."""
        import torch
        dev = f"cuda:{self.device}"
        results = torch.zeros((max(max_results, 1), 6), dtype=torch.int32,
                              device=dev)
        index_and_flag = torch.zeros(2, dtype=torch.int32, device=dev)
        self.compute_king(submatrix, words_per_sample, bit_sets, kin_threshold,
                          max_results, results, index_and_flag[0:1],
                          index_and_flag[1:2], tile_range=tile_range)
        torch.cuda.synchronize(self.device)
        count, overflow = (int(x) & 0xFFFFFFFF for x in index_and_flag.tolist())
        if overflow:
            raise ResourceExhaustedError(
                "Could not store all results: try increasing the "
                "--max_results parameter.")
        host = results[:count].cpu().numpy().view(np.uint32).reshape(-1)
        recs = host.view(KING_RESULT_DTYPE).copy()
        return sort_results(recs) if sort else recs

    def pack_device(self, submatrix: Submatrix, words_per_sample: int,
                    bit_set, row_idx, col_idx, n_alt_alleles, status,
                    stream=None) -> None:
        """cuking.cu:675-703 as a device kernel; all arguments device tensors
        (int64, int64, int32; status one int32, zeroed by the caller)."""
        import torch
        assert row_idx.dtype == torch.int64 and col_idx.dtype == torch.int64
        assert n_alt_alleles.dtype == torch.int32
        assert row_idx.numel() == col_idx.numel() == n_alt_alleles.numel()
        check(self.lib.cuking_pack_device(
            self.handle, C.byref(submatrix.c), words_per_sample,
            bit_set.data_ptr(), row_idx.data_ptr(), col_idx.data_ptr(),
            n_alt_alleles.data_ptr(), row_idx.numel(), status.data_ptr(),
            _stream_handle(stream)))

    def synth_bitset(self, seed: int, kind, pa, pb, sample_begin: int,
                     sample_end: int, num_sites: int, out=None, stream=None):
        """Synthetic reference-layout bitset rows [sample_begin, sample_end)
        (bit-identical to oracle/synth_oracle.c)."""
        import torch
        wps = words_per_sample(num_sites)
        if out is None:
            out = self._tensor((sample_end - sample_begin, wps), torch.int64)
        assert out.numel() >= (sample_end - sample_begin) * wps
        check(self.lib.cuking_synth_bitset(
            self.handle, seed, kind.data_ptr(), pa.data_ptr(), pb.data_ptr(),
            sample_begin, sample_end, num_sites, wps, out.data_ptr(),
            _stream_handle(stream)))
        return out

    # -- measurement ----------------------------------------------------------
    def timing_enable(self, enabled: bool = True) -> None:
        check(self.lib.cuking_timing_enable(self.handle, int(enabled)))

    def timing_reset(self) -> None:
        check(self.lib.cuking_timing_reset(self.handle))

    def timing_collect(self) -> KernelTiming:
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_uint64(), C.c_uint64()
        check(self.lib.cuking_timing_collect(
            self.handle, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return KernelTiming(a.value, na.value, b.value, nb.value)

    def clock_probe(self, microseconds: int, stream):
        """Starts the sustained-clock probe on ``stream`` (a side stream);
        returns a function that waits for it and gives the clock in MHz."""
        import torch
        ticks = torch.zeros(2, dtype=torch.int64, device=f"cuda:{self.device}")
        stream.wait_stream(torch.cuda.current_stream())
        check(self.lib.cuking_clock_probe(self.handle, int(microseconds), ticks.data_ptr(),
                                          _stream_handle(stream)))

        def result() -> float:
            stream.synchronize()
            shader, real = ticks.tolist()
            return 100.0 * shader / real if real else 0.0
        return result

    # -- helpers --------------------------------------------------------------
    def _check_bits(self, sm: Submatrix, wps: int, bit_sets) -> None:
        import torch
        if not bit_sets.is_cuda or bit_sets.device.index != self.device:
            raise ValueError("bit_sets must live on this context's GPU")
        if bit_sets.dtype != torch.int64 or not bit_sets.is_contiguous():
            raise ValueError("bit_sets must be a contiguous int64 tensor")
        if bit_sets.numel() < sm.NumSamples() * wps:
            raise ValueError(
                f"bit_sets holds {bit_sets.numel()} words, the block needs "
                f"{sm.NumSamples()} x {wps}")


__all__ = [
    "Submatrix", "KingContext", "KING_RESULT_DTYPE", "KING_COUNTS_DTYPE",
    "ResourceExhaustedError", "CukingError", "padded_sites",
    "words_per_sample", "bytes_per_pair", "new_host_bitset", "pack_host",
    "sort_results", "device_count", "DEFAULT_KIN_THRESHOLD",
    "DEFAULT_MAX_RESULTS",
]
