"""cuking_amd: MI355X-native KING-robust kinship (all-pairs popcount path).

Hot path only: bitset pack -> all-pairs AND/popcount kernel -> thresholded
KingResult records, behind the C ABI of include/cuking_amd.h.  The HIP library
is loaded on first use and there is no CPU fallback.
"""
from .api import (DEFAULT_KIN_THRESHOLD, DEFAULT_MAX_RESULTS,  # noqa: F401
                  KING_COUNTS_DTYPE, KING_RESULT_DTYPE, CukingError,
                  KingContext, ResourceExhaustedError, Submatrix,
                  bytes_per_pair, device_count, new_host_bitset, pack_host,
                  padded_sites, sort_results, words_per_sample)

__version__ = "0.1.0"
