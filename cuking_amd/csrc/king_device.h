// Device-side pieces shared by the pair kernels (king_kernels.hip,
// king_mfma.hip): tile decoding, the kinship arithmetic, the record append and
// the per-emitted-pair recount.
#ifndef CUKING_AMD_KING_DEVICE_H_
#define CUKING_AMD_KING_DEVICE_H_

#include <hip/hip_runtime.h>

#include <atomic>

#include "king_common.h"

namespace cuking {

typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *global_void_ptr;

// cuking.cu:289-294: two float32 roundings (divide, add).  Numerator and
// denominator are exact integers (< 2^24 for < 2^22 sites); the divide is the
// IEEE-correct one (no fast-math, see build flags).  min == 0 gives -inf or
// NaN, which fails `kin > threshold`.
__device__ __forceinline__ float king_kinship(uint32_t het_i, uint32_t het_j,
                                              uint32_t both_het,
                                              uint32_t opposing_hom) {
  const uint32_t min_hets = het_i < het_j ? het_i : het_j;
  const float num = 2.f * (float)both_het - 4.f * (float)opposing_hom -
                    (float)het_i - (float)het_j;
  const float den = 4.f * (float)min_hets;
  return 0.5f + num / den;
}

// cuking.cu:297-313: reserve a slot, store or flag overflow.
__device__ __forceinline__ void emit_result(uint32_t i, uint32_t j, float kin,
                                            uint32_t ibs0, uint32_t ibs1,
                                            uint32_t ibs2, uint32_t max_results,
                                            cuking_result *results,
                                            uint32_t *result_index,
                                            uint32_t *result_overflow) {
  const uint32_t slot = atomicAdd(result_index, 1u);
  if (slot < max_results) {
    cuking_result r;
    r.sample_i = i;
    r.sample_j = j;
    r.kin = kin;
    r.ibs0 = ibs0;
    r.ibs1 = ibs1;
    r.ibs2 = ibs2;
    results[slot] = r;
  } else {
    atomicMax(result_overflow, 1u);
  }
}

// Sites where both samples are homozygous (hom-ref or hom-alt), counted by one
// whole wavefront straight from the reference layout: ~het is "homozygous and
// defined" (missing and padding sites have the het bit set, cuking.cu:688-697).
// Every lane gets the sum.
__device__ __forceinline__ uint32_t wave_hom_hom_count(
    const uint64_t *__restrict__ bits, uint32_t words_per_sample,
    uint32_t offset_i, uint32_t offset_j, uint32_t lane) {
  typedef const __attribute__((address_space(1))) uint64_t *gptr;
  const uint32_t n = words_per_sample / 2;
  // The two het planes: wave-uniform bases (SGPR pairs) + one lane offset, so
  // that a trip's loads share their address registers.
  auto uniform = [](const uint64_t *p) {
    const uint64_t v = (uint64_t)p;
    // (the builtin returns int: without the casts the low word is sign-extended)
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    return (gptr)(((uint64_t)hi << 32) | lo);
  };
  const gptr het_i = uniform(bits + (uint64_t)offset_i * words_per_sample);
  const gptr het_j = uniform(bits + (uint64_t)offset_j * words_per_sample);
  uint32_t c = 0;
  // The loop is one memory latency per trip (the planes of an arbitrary pair
  // are cold), so a trip requests 2 x 8 words per lane before it counts any:
  // a 100k-site pair takes 4 trips instead of 25 (tiles with ~40 related pairs
  // spent 350 us here, archive/profiles/r02_tail.txt).
  constexpr uint32_t kAhead = 8;
  for (uint32_t w0 = 0; w0 < n; w0 += 64 * kAhead) {  // (uniform trip count)
    uint64_t x[kAhead], y[kAhead];
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) {
      const uint32_t w = w0 + 64 * k + lane;
      const bool in = w < n;  // beyond the plane: counts as "het" = nothing
      x[k] = in ? het_i[w] : ~0ull;
      y[k] = in ? het_j[w] : ~0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) c += __popcll(~(x[k] | y[k]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  return c;
}

// Which tile workgroup `t` of the launch owns (uniform across the workgroup).
// Returns false when there is nothing to do (rectangle mode, below the
// diagonal of a diagonal block): the whole workgroup must then leave before
// any barrier.
__device__ __forceinline__ bool decode_tile_space(const TiledArgs &a, uint64_t t,
                                                  uint32_t *tr, uint32_t *tc);
__device__ __forceinline__ bool decode_tile(const TiledArgs &a, uint64_t t,
                                            uint32_t *tr, uint32_t *tc) {
  if (a.tile_list != nullptr) {  // tile-list mode (king_common.h)
    const uint2 e = a.tile_list[t];
    *tr = e.x;
    *tc = e.y;
    return true;
  }
  if (a.quad == 0) return decode_tile_space(a, t, tr, tc);
  // Quadrant mode: quadrant t % 4 of the 256-sample tile t / 4.
  // (the filter's fallback launch: tiles the filter kernel has dealt with)
  if (a.skip_tiles != nullptr && a.skip_tiles[(t >> 2) - a.skip_base] != 0) return false;
  uint32_t r, c;
  if (!decode_tile_space(a, t >> 2, &r, &c)) return false;
  *tr = 2 * r + (uint32_t)((t >> 1) & 1);
  *tc = 2 * c + (uint32_t)(t & 1);
  // (a diagonal tile's lower-left quadrant holds no pair with i < j)
  return !(a.tiles.diag && *tc < *tr);
}
__device__ __forceinline__ bool decode_tile_space(const TiledArgs &a, uint64_t t,
                                                  uint32_t *tr, uint32_t *tc) {
  if (a.rect_rows != 0) {
    // Rectangle mode: bands of band_rows rows of the rectangle, column-major
    // inside a band (same locality as the whole-block enumeration).  Only the
    // last band can be shorter.
    const uint32_t g = a.tiles.band_rows;
    const uint64_t per_band = (uint64_t)g * a.rect_cols;
    const uint32_t b = (uint32_t)(t / per_band);
    const uint32_t r0 = b * g;
    const uint32_t h = a.rect_rows - r0 < g ? a.rect_rows - r0 : g;
    const uint64_t u = t - (uint64_t)b * per_band;
    *tr = a.rect_row0 + (r0 + (uint32_t)(u % h)) * a.rect_row_stride;
    *tc = a.rect_col0 + (uint32_t)(u / h);
    return !(a.tiles.diag && *tc < *tr);
  }
  uint32_t lo = 0, hi = a.tiles.num_bands();
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.band_prefix[mid] <= t) lo = mid; else hi = mid;
  }
  a.tiles.decode(lo, t - a.band_prefix[lo], tr, tc);
  return true;
}

// What the record-emitting epilogues need of the launch arguments (small, so
// that an out-of-line epilogue can take it by value and the kernel's argument
// struct never has its address taken).
struct EmitCtx {
  float kin_threshold;
  uint32_t max_results;
  cuking_result *results;
  uint32_t *result_index;
  uint32_t *result_overflow;
  const uint64_t *bits;
  uint32_t words_per_sample;
  uint32_t diag, num_rows;
  uint32_t i_begin, j_begin;
};
// ... plus the sample order of the layout (king_common.h `perm`, or nullptr): the
// four-product kernel's epilogues, which may run on a sorted layout.  (A type of its own:
// two more live registers in the five-product full form's epilogue were enough for the
// allocator to park its k loop's LDS offsets in scratch and reload them inside the loop.)
struct EmitCtxP : EmitCtx {
  const uint32_t *perm;
};

// The stored sample (index into `bits`: rows first, then an off-diagonal block's columns)
// behind row / column plane index li / lj of the block.
__device__ __forceinline__ uint32_t stored_row(const EmitCtx &, uint32_t li) { return li; }
__device__ __forceinline__ uint32_t stored_col(const EmitCtx &c, uint32_t lj) {
  return c.diag ? lj : c.num_rows + lj;
}
__device__ __forceinline__ uint32_t stored_row(const EmitCtxP &c, uint32_t li) {
  return c.perm != nullptr ? c.perm[li] : li;
}
__device__ __forceinline__ uint32_t stored_col(const EmitCtxP &c, uint32_t lj) {
  // (a sorted layout is the filter variant's: 256-sample tiles, the columns of an
  //  off-diagonal block behind the padded rows)
  return c.perm != nullptr
             ? c.perm[(c.diag ? 0u : (c.num_rows + kFilterTile - 1) / kFilterTile * kFilterTile) + lj]
             : (c.diag ? lj : c.num_rows + lj);
}
// The record's (sample_i, sample_j) for stored samples (si, sj): global indices, smaller
// first (a sorted layout enumerates a diagonal block's pairs in plane order; every field
// of a record is symmetric in the two samples).
__device__ __forceinline__ void record_pair(const EmitCtx &c, uint32_t si, uint32_t sj,
                                            uint32_t *gi, uint32_t *gj) {
  const uint32_t a = c.i_begin + si;
  const uint32_t b = c.diag ? c.j_begin + sj : c.j_begin + (sj - c.num_rows);
  *gi = a < b ? a : b;
  *gj = a < b ? b : a;
}

__device__ __forceinline__ EmitCtx make_emit_ctx(const TiledArgs &a) {
  EmitCtx c;
  c.kin_threshold = a.kin_threshold;
  c.max_results = a.max_results;
  c.results = a.results;
  c.result_index = a.result_index;
  c.result_overflow = a.result_overflow;
  c.bits = a.bits;
  c.words_per_sample = a.words_per_sample;
  c.diag = a.geo.diag;
  c.num_rows = a.geo.num_rows;
  c.i_begin = a.i_begin;
  c.j_begin = a.j_begin;
  return c;
}
__device__ __forceinline__ EmitCtxP make_emit_ctx_p(const TiledArgs &a) {
  EmitCtxP c;
  static_cast<EmitCtx &>(c) = make_emit_ctx(a);
  c.perm = a.perm;
  return c;
}

// Lean epilogue of one pair held by every lane (cuking.cu:284-313): the main
// loop kept the four sums kinship needs.  IBS0 and IBS1 follow from them
// (ibs1 = exactly one het = hi + hj - 2 bh); IBS2 needs the hom/hom count,
// which the whole wavefront sums for each of the (few) pairs that pass the
// threshold.  Must be called by all 64 lanes.
__device__ __forceinline__ void lean_epilogue_pair(
    const EmitCtx &a, bool valid, uint32_t li, uint32_t lj, uint32_t het_i,
    uint32_t het_j, uint32_t both_het, uint32_t opp, uint32_t lane) {
  const float kin = king_kinship(het_i, het_j, both_het, opp);
  const bool emit = valid && kin > a.kin_threshold;
  unsigned long long pending = __ballot(emit);  // wave-uniform
  uint32_t hom_hom = 0;
  while (pending) {
    const int src = __builtin_ctzll(pending);
    pending &= pending - 1;
    const uint32_t p_li = __builtin_amdgcn_readlane(li, src);
    const uint32_t p_lj = __builtin_amdgcn_readlane(lj, src);
    const uint32_t sum = wave_hom_hom_count(a.bits, a.words_per_sample, stored_row(a, p_li),
                                            stored_col(a, p_lj), lane);
    if ((int)lane == src) hom_hom = sum;
  }
  if (emit) {
    const uint32_t ibs0 = opp, ibs2 = hom_hom - opp + both_het;
    uint32_t gi, gj;
    record_pair(a, stored_row(a, li), stored_col(a, lj), &gi, &gj);
    emit_result(gi, gj, kin, ibs0, het_i + het_j - 2 * both_het, ibs2, a.max_results,
                a.results, a.result_index, a.result_overflow);
  }
}

// Cheap, conservative stand-in for `king_kinship(...) > threshold` on the exact
// float sums (integers < 2^24): false only when the pair certainly fails.  The
// approximate quotient num * rcp(den) is within 3e-7 relative of the correctly
// rounded one, the margin is 1e-5 (1 + |q|); den == 0 gives -inf or NaN, which
// also fail the exact test (both_het <= min(het_i, het_j) = 0 makes num <= 0).
// Lets the epilogue skip the IEEE divide for the ~all pairs under the
// threshold; candidates still take the exact path.
__device__ __forceinline__ bool kinship_may_pass(float het_i, float het_j,
                                                 float both_het, float opp,
                                                 float threshold) {
  const float num = 2.f * both_het - 4.f * opp - het_i - het_j;
  const float den = 4.f * fminf(het_i, het_j);
  const float q = num * __builtin_amdgcn_rcpf(den);
  return q >= (threshold - 0.5f) - 1e-5f * (1.f + fabsf(q));
}

// The same decision from the numerator itself (four-product kernel: num = hi +
// hj - 2 dd + 2 q as a float sum of exact terms, at most one rounding of 6e-8
// relative; min_hets = min(hi, hj)).
__device__ __forceinline__ bool kinship_may_pass_num(float num, float min_hets,
                                                     float threshold) {
  const float q = num * __builtin_amdgcn_rcpf(4.f * min_hets);
  return q >= (threshold - 0.5f) - 1e-5f * (1.f + fabsf(q));
}

// Lean epilogue of the four-product kernel (king_mfma.hip): the main loop kept
// hi, hj, dd = both defined and q = hom_hom - 2 opp.  Kinship's numerator
// 2 bh - 4 opp - hi - hj (cuking.cu:291) equals hi + hj - 2 dd + 2 q; below 2^22
// sites (kMfmaN4MaxSites) that integer is below 2^24 in magnitude like every
// partial sum of the reference's float expression, so (float)num IS that
// expression's value and kin comes out bit for bit the same.  bh and opp
// themselves are only needed for the records: bh = hi + hj - dd + hom_hom and
// opp = (hom_hom - q) / 2 from the recounted hom_hom.  Must be called by all 64
// lanes.
__device__ __forceinline__ void lean_epilogue_pair_n4(
    const EmitCtxP &a, bool valid, uint32_t li, uint32_t lj, uint32_t het_i,
    uint32_t het_j, uint32_t dd, int32_t q, uint32_t lane) {
  const uint32_t min_hets = het_i < het_j ? het_i : het_j;
  const int32_t num = (int32_t)(het_i + het_j) - 2 * (int32_t)dd + 2 * q;
  const float kin = 0.5f + (float)num / (4.f * (float)min_hets);
  const bool emit = valid && kin > a.kin_threshold;
  unsigned long long pending = __ballot(emit);  // wave-uniform
  uint32_t hom_hom = 0;
  while (pending) {
    const int src = __builtin_ctzll(pending);
    pending &= pending - 1;
    const uint32_t p_li = __builtin_amdgcn_readlane(li, src);
    const uint32_t p_lj = __builtin_amdgcn_readlane(lj, src);
    const uint32_t sum = wave_hom_hom_count(a.bits, a.words_per_sample, stored_row(a, p_li),
                                            stored_col(a, p_lj), lane);
    if ((int)lane == src) hom_hom = sum;
  }
  if (emit) {
    const uint32_t both_het = het_i + het_j - dd + hom_hom;
    const uint32_t opp = (uint32_t)((int32_t)hom_hom - q) >> 1;
    const uint32_t ibs2 = hom_hom - opp + both_het;
    uint32_t gi, gj;
    record_pair(a, stored_row(a, li), stored_col(a, lj), &gi, &gj);
    emit_result(gi, gj, kin, opp, het_i + het_j - 2 * both_het, ibs2, a.max_results,
                a.results, a.result_index, a.result_overflow);
  }
}

// Full epilogue of one pair: all six reference sums from the five kept ones.
__device__ __forceinline__ void full_epilogue_pair(
    const TiledArgs &a, bool valid, uint32_t li, uint32_t lj, uint32_t het_i,
    uint32_t het_j, uint32_t both_het, uint32_t opp, uint32_t hom_hom) {
  if (!valid) return;
  const uint32_t conc = hom_hom - opp;
  const uint32_t shared = het_i + het_j - both_het + hom_hom;
  // (a sorted layout: the pair's stored samples; in a diagonal block the smaller one is
  //  "i", and het_i / het_j follow it)
  uint32_t si = li, sj = lj;  // row / column index within the block
  if (a.perm != nullptr) {
    si = a.perm[li];
    sj = a.perm[a.geo.col_base + lj];
    if (!a.geo.diag) sj -= a.geo.num_rows;
    if (a.geo.diag && si > sj) {
      const uint32_t t = si; si = sj; sj = t;
      const uint32_t h = het_i; het_i = het_j; het_j = h;
    }
  }
  if (a.dense_counts != nullptr) {
    cuking_counts c;
    c.het_i = het_i;
    c.het_j = het_j;
    c.both_het = both_het;
    c.opposing_hom = opp;
    c.concordant_hom = conc;
    c.shared = shared;
    a.dense_counts[(uint64_t)si * a.geo.num_cols + sj] = c;
    return;
  }
  const float kin = king_kinship(het_i, het_j, both_het, opp);
  if (kin > a.kin_threshold) {
    const uint32_t ibs0 = opp, ibs2 = conc + both_het;
    emit_result(a.i_begin + si, a.j_begin + sj, kin, ibs0, shared - ibs0 - ibs2,
                ibs2, a.max_results, a.results, a.result_index,
                a.result_overflow);
  }
}

// "Done once per device" flag for per-function attributes
// (hipFuncSetAttribute applies to the current device's function object only).
// Lock-free; a race merely sets the attribute twice.
struct DeviceOnce {
  std::atomic<uint64_t> mask[4] = {};  // up to 256 devices
  static int current() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 255;
  }
  bool done() const {
    const int d = current();
    return (mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1;
  }
  void mark() {
    const int d = current();
    mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release);
  }
};

// Workgroups per launch: one launch may not exceed 2^32 - 1 threads in x (HIP
// truncates silently beyond that); tests can lower the cap.
uint64_t max_blocks_per_launch(uint32_t threads);

}  // namespace cuking

#endif  // CUKING_AMD_KING_DEVICE_H_
