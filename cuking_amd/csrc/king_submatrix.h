// Submatrix helpers usable on the host (any C++ compiler) and on the device.
#ifndef CUKING_AMD_KING_SUBMATRIX_H_
#define CUKING_AMD_KING_SUBMATRIX_H_

#include <stdint.h>

#include "cuking_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CUKING_HD __host__ __device__
#else
#define CUKING_HD
#endif

namespace cuking {

// Submatrix helpers usable on both sides (cuking.cu:154-175).
CUKING_HD inline uint32_t sm_num_rows(const cuking_submatrix &s) {
  return s.i_end - s.i_begin;
}
CUKING_HD inline uint32_t sm_num_cols(const cuking_submatrix &s) {
  return s.j_end - s.j_begin;
}
CUKING_HD inline bool sm_is_diag(const cuking_submatrix &s) {
  return s.i_begin == s.j_begin;
}
CUKING_HD inline uint32_t sm_num_samples(const cuking_submatrix &s) {
  return sm_is_diag(s) ? sm_num_rows(s) : sm_num_rows(s) + sm_num_cols(s);
}
CUKING_HD inline bool sm_contains(const cuking_submatrix &s,
                                            uint32_t index) {
  return (s.i_begin <= index && index < s.i_end) ||
         (s.j_begin <= index && index < s.j_end);
}
CUKING_HD inline uint32_t sm_sample_offset(const cuking_submatrix &s,
                                                     uint32_t index) {
  return index < s.i_end ? index - s.i_begin
                         : (s.i_end - s.i_begin) + (index - s.j_begin);
}

}  // namespace cuking

#endif  // CUKING_AMD_KING_SUBMATRIX_H_
