"""ctypes bindings for oracle/king_oracle.c and oracle/synth_oracle.c.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

RESULT_DTYPE = np.dtype(
    [("sample_i", "<u4"), ("sample_j", "<u4"), ("kin", "<f4"),
     ("ibs0", "<u4"), ("ibs1", "<u4"), ("ibs2", "<u4")])  # cuking.cu:182-186
COUNTS_DTYPE = np.dtype(
    [("het_i", "<u4"), ("het_j", "<u4"), ("both_het", "<u4"),
     ("opposing_hom", "<u4"), ("concordant_hom", "<u4"), ("shared", "<u4")])


class Submatrix(C.Structure):
    _fields_ = [("i_begin", C.c_uint32), ("i_end", C.c_uint32),
                ("j_begin", C.c_uint32), ("j_end", C.c_uint32)]

    def as_tuple(self):
        return (self.i_begin, self.i_end, self.j_begin, self.j_end)


def build(native: bool = False, out_dir: os.PathLike | None = None) -> Path:
    """Compiles the oracle with gcc (oracle/Makefile) and returns the .so path."""
    if native and out_dir is None:
        # -march=native code must never travel to another machine: build it in
        # a scratch directory of the machine that runs it.
        import tempfile
        out_dir = tempfile.mkdtemp(prefix="king_oracle_native_")
    out = Path(out_dir) if out_dir else _HERE
    out.mkdir(parents=True, exist_ok=True)
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", str(_HERE), target, f"OUT={out}"],
                   check=True)
    return out / ("libking_oracle_native.so" if native else "libking_oracle.so")


_LIBS: dict[str, C.CDLL] = {}


def load(native: bool = False, out_dir: os.PathLike | None = None) -> C.CDLL:
    key = f"{native}:{out_dir}"
    if key in _LIBS:
        return _LIBS[key]
    lib = C.CDLL(str(build(native=native, out_dir=out_dir)))
    u32, u64, f32, vp = C.c_uint32, C.c_uint64, C.c_float, C.c_void_p
    SM = C.POINTER(Submatrix)
    lib.orc_submatrix_init.argtypes = [SM, u32, u32, u32]
    lib.orc_submatrix_init.restype = C.c_int
    for name in ("orc_num_rows", "orc_num_cols", "orc_num_samples"):
        getattr(lib, name).argtypes = [SM]
        getattr(lib, name).restype = u32
    for name in ("orc_contains", "orc_sample_offset"):
        getattr(lib, name).argtypes = [SM, u32]
        getattr(lib, name).restype = u32
    lib.orc_padded_sites.argtypes = [u32]
    lib.orc_padded_sites.restype = u32
    lib.orc_words_per_sample.argtypes = [u32]
    lib.orc_words_per_sample.restype = u32
    lib.orc_bitset_init.argtypes = [vp, C.c_size_t]
    lib.orc_bitset_init.restype = None
    lib.orc_pack.argtypes = [SM, u32, vp, vp, vp, vp, C.c_size_t]
    lib.orc_pack.restype = C.c_int
    lib.orc_pair_counts.argtypes = [vp, vp, u32, vp]
    lib.orc_pair_counts.restype = None
    lib.orc_kin.argtypes = [vp]
    lib.orc_kin.restype = f32
    lib.orc_compute.argtypes = [SM, u32, vp, f32, u32, vp, C.POINTER(u32)]
    lib.orc_compute.restype = u64
    lib.orc_compute_mt.argtypes = [SM, u32, vp, f32, u32, vp, C.POINTER(u32),
                                   C.c_int]
    lib.orc_compute_mt.restype = u64
    lib.orc_all_pairs.argtypes = [SM, u32, vp, u64, vp, vp, vp, vp]
    lib.orc_all_pairs.restype = u64
    lib.orc_sort.argtypes = [vp, C.c_size_t]
    lib.orc_sort.restype = None
    lib.orc_max_threads.argtypes = []
    lib.orc_max_threads.restype = C.c_int
    lib.syn_genotype.argtypes = [u64, vp, vp, vp, u32, u32]
    lib.syn_genotype.restype = u32
    lib.syn_fill_bitset.argtypes = [u64, vp, vp, vp, u32, u32, u32, u32, vp]
    lib.syn_fill_bitset.restype = None
    _LIBS[key] = lib
    return lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def submatrix(num_samples: int, split_factor: int = 1, shard_index: int = 0,
              lib=None) -> Submatrix:
    lib = lib or load()
    sm = Submatrix()
    if lib.orc_submatrix_init(C.byref(sm), num_samples, split_factor,
                              shard_index) != 0:
        raise ValueError("Invalid split factor / shard index")
    return sm


def words_per_sample(num_sites: int) -> int:
    return int(load().orc_words_per_sample(num_sites))


def new_bitset(sm: Submatrix, num_sites: int) -> np.ndarray:
    """All-missing bitset [NumSamples, words_per_sample] (cuking.cu:513-523)."""
    lib = load()
    wps = lib.orc_words_per_sample(num_sites)
    bits = np.empty((lib.orc_num_samples(C.byref(sm)), wps), dtype=np.uint64)
    lib.orc_bitset_init(_p(bits), bits.size)
    return bits


def pack(sm: Submatrix, bits: np.ndarray, row_idx, col_idx, n_alt) -> None:
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int64)
    n_alt = np.ascontiguousarray(n_alt, dtype=np.int32)
    assert bits.flags.c_contiguous and bits.dtype == np.uint64
    rc = load().orc_pack(C.byref(sm), bits.shape[1], _p(bits), _p(row_idx),
                         _p(col_idx), _p(n_alt), row_idx.size)
    if rc == -2:
        raise ValueError("Invalid value for n_alt_alleles")
    if rc == -3:
        raise ValueError("row_idx out of range")
    if rc != 0:
        raise RuntimeError(f"orc_pack failed: {rc}")


def bitset_from_genotypes(geno: np.ndarray, sm: Submatrix | None = None):
    """geno: int8 [N, M], -1 = missing.  Goes through the triple format
    (absent triple = missing), exactly like the Parquet path."""
    n, m = geno.shape
    sm = sm or submatrix(n)
    bits = new_bitset(sm, m)
    col, row = np.nonzero(geno >= 0)
    pack(sm, bits, row, col, geno[col, row].astype(np.int32))
    return bits


def compute(sm: Submatrix, bits: np.ndarray, kin_threshold: float,
            max_results: int = 10 << 20, threads: int = 0, native: bool = False):
    """Returns (sorted results, overflow flag, number of qualifying pairs)."""
    lib = load(native=native)
    res = np.zeros(max_results, dtype=RESULT_DTYPE)
    ovf = C.c_uint32(0)
    if threads == 0:
        n = lib.orc_compute(C.byref(sm), bits.shape[1], _p(bits),
                            kin_threshold, max_results, _p(res), C.byref(ovf))
    else:
        n = lib.orc_compute_mt(C.byref(sm), bits.shape[1], _p(bits),
                               kin_threshold, max_results, _p(res),
                               C.byref(ovf), threads)
    k = min(int(n), max_results)
    res = res[:k].copy()
    lib.orc_sort(_p(res), k)
    return res, int(ovf.value), int(n)


def all_pairs(sm: Submatrix, bits: np.ndarray):
    """(i, j, counts, kin) for every pair of the submatrix, (i, j) order."""
    lib = load()
    r, c = lib.orc_num_rows(C.byref(sm)), lib.orc_num_cols(C.byref(sm))
    cap = r * c
    oi = np.zeros(cap, dtype=np.uint32)
    oj = np.zeros(cap, dtype=np.uint32)
    oc = np.zeros(cap, dtype=COUNTS_DTYPE)
    ok = np.zeros(cap, dtype=np.float32)
    n = int(lib.orc_all_pairs(C.byref(sm), bits.shape[1], _p(bits), cap,
                              _p(oi), _p(oj), _p(oc), _p(ok)))
    return oi[:n], oj[:n], oc[:n], ok[:n]


def synth_bitset(seed: int, kind, pa, pb, sample_begin: int, sample_end: int,
                 num_sites: int) -> np.ndarray:
    lib = load()
    kind = np.ascontiguousarray(kind, dtype=np.uint32)
    pa = np.ascontiguousarray(pa, dtype=np.uint32)
    pb = np.ascontiguousarray(pb, dtype=np.uint32)
    wps = lib.orc_words_per_sample(num_sites)
    bits = np.empty((sample_end - sample_begin, wps), dtype=np.uint64)
    lib.syn_fill_bitset(seed, _p(kind), _p(pa), _p(pb), sample_begin,
                        sample_end, num_sites, wps, _p(bits))
    return bits


def synth_genotypes(seed: int, kind, pa, pb, samples, num_sites: int):
    """int8 genotype matrix (-1 missing) for the listed samples (slow)."""
    lib = load()
    kind = np.ascontiguousarray(kind, dtype=np.uint32)
    pa = np.ascontiguousarray(pa, dtype=np.uint32)
    pb = np.ascontiguousarray(pb, dtype=np.uint32)
    out = np.empty((len(samples), num_sites), dtype=np.int8)
    for r, s in enumerate(samples):
        for site in range(num_sites):
            g = lib.syn_genotype(seed, _p(kind), _p(pa), _p(pb), int(s), site)
            out[r, site] = -1 if g == 3 else g
    return out
