// Internal: what the host-only half of the C ABI (king_host.cc) shares with the
// HIP half (king_abi.hip).
#ifndef CUKING_AMD_KING_HOST_H_
#define CUKING_AMD_KING_HOST_H_

#include "cuking_amd.h"

// Records the calling thread's error message (cuking_last_error) and returns
// `code`.
cuking_status cuking_fail(cuking_status code, const char *fmt, ...)
    __attribute__((format(printf, 2, 3)));
// Argument checks every entry point that takes a block makes.
cuking_status cuking_check_block(const cuking_submatrix *sm, uint32_t words_per_sample);

#endif  // CUKING_AMD_KING_HOST_H_
