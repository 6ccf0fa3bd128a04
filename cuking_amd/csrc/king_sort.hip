// The sample order of the filter variant's kernel layout (king_common.h, TiledArgs::perm):
// the samples of a prepared range sorted by their share of missing calls.
//
// The slack of the filter's bound is the missingness of the two samples of a pair
// (king_filter.hip: about the missing rate in kinship), so a cohort with a FEW
// low-call-rate samples -- 2 % of the samples at 20 % missing calls is an ordinary exome
// batch effect -- has a few samples whose pairs the bound cannot rule out.  In stored order
// they sit two or three to every 128 x 128 quadrant, and every quadrant of the cohort goes
// to the exact kernel; sorted, they fill the last tile rows and columns and the other 96 %
// of the tiles keep the filter's speed.  The reference has no counterpart (it evaluates
// every pair at constant cost, cuking.cu:216-240).
//
// Steps, per prepared range of plane samples (the whole row / column side of a block, or
// one broadcast chunk of the staged multi-GPU pass -- any tile-aligned range is a valid
// unit, since pairs are enumerated in plane order):
//   1. sample_stats_kernel (king_filter.hip) has left the statistics in STORED order;
//   2. keys: the missing share in 1/64ths (8 bits: one radix pass) -- coarse on purpose: the
//      samples of an ordinary cohort (1 % missing calls, give or take) all get key 0 and
//      the layout IS the stored order;
//   3. a STABLE sort of (key, sample) pairs -- deterministic, and samples of equal share
//      (an ordinary cohort: all of them) keep their stored order;
//   4. perm and the statistics written in plane order; padding samples behind the real
//      ones (kNoSample, zero statistics).
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "king_common.h"

namespace cuking {

namespace {

// Keys of the real samples [begin, begin + n) of a side: stored sample = plane sample
// before the sort.  |M| follows from the statistics: per stored site exactly one of
// hom-and-defined, het, missing, so u = |Y| - |M| = sites - |H| - 2 |M|.
__global__ void sort_keys_kernel(const float2 *__restrict__ tmp_stats, uint32_t begin, uint32_t n,
                                 float stored_sites, uint32_t *__restrict__ keys,
                                 uint32_t *__restrict__ vals) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float2 st = tmp_stats[begin + p];
  const float missing = 0.5f * (stored_sites - st.y - st.x);
  uint32_t key = (uint32_t)(missing * 64.f / stored_sites);
  keys[p] = key > 255u ? 255u : key;
  vals[p] = begin + p;
}

__global__ void identity_order_kernel(uint32_t begin, uint32_t n, uint32_t *__restrict__ vals) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) vals[p] = begin + p;
}

// Plane samples [begin, end): the first n take the sorted real samples, the rest padding.
__global__ void apply_order_kernel(PlaneGeometry geo, uint32_t begin, uint32_t end, uint32_t n,
                                   const uint32_t *__restrict__ order,
                                   const float2 *__restrict__ tmp_stats,
                                   const float *__restrict__ tmp_prefix,
                                   uint32_t *__restrict__ perm, float2 *__restrict__ stats,
                                   float *__restrict__ prefix) {
  const uint32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= end) return;
  if (p - begin >= n) {
    perm[p] = kNoSample;
    stats[p] = make_float2(0.f, 0.f);
#pragma unroll
    for (uint32_t c = 0; c < kNumCum; ++c) prefix[(size_t)c * geo.s_stride + p] = 0.f;
    return;
  }
  const uint32_t src = order[p - begin];  // plane sample in stored order
  // the stored sample behind it: rows as they are, columns of an off-diagonal block behind
  // the rows (cuking.cu:171-175)
  perm[p] = (geo.diag || src < geo.rows_padded) ? src : geo.num_rows + (src - geo.col_base);
  stats[p] = tmp_stats[src];
#pragma unroll 7
  for (uint32_t c = 0; c < kNumCum; ++c)
    prefix[(size_t)c * geo.s_stride + p] = tmp_prefix[(size_t)c * geo.s_stride + src];
}

}  // namespace

size_t sort_temp_bytes_for(uint32_t n) {
  size_t bytes = 0;
  uint32_t *nil = nullptr;
  if (n == 0) return 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, nil, nil, nil, nil, (int)n, 0, 8);
  return (bytes + 255) / 256 * 256;
}

hipError_t launch_sample_order(const PlaneGeometry &geo, uint32_t words_per_sample,
                               uint4 *d_planes, uint32_t s_begin, uint32_t s_end, bool sort,
                               void *sort_temp, size_t sort_temp_bytes, hipStream_t stream) {
  if (s_end > geo.s_stride) s_end = geo.s_stride;
  if (s_begin >= s_end) return hipSuccess;
  const float2 *tmp_stats = plane_tmp_stats(d_planes, geo);
  const float *tmp_prefix = plane_tmp_prefix(d_planes, geo);
  uint32_t *perm = plane_perm(d_planes, geo);
  float2 *stats = const_cast<float2 *>(plane_stats(d_planes, geo));
  float *prefix = const_cast<float *>(plane_prefix_u(d_planes, geo));
  uint32_t *words = plane_sort_words(d_planes, geo);
  const float stored_sites = 32.f * (float)words_per_sample;  // (what the statistics count over)
  // The sides of the block inside the range: rows [0, rows_padded), columns behind them.
  struct Side { uint32_t begin, end, real_end; };
  Side sides[2];
  int num_sides = 0;
  if (geo.diag) {
    sides[num_sides++] = {0, geo.rows_padded, geo.num_rows};
  } else {
    sides[num_sides++] = {0, geo.rows_padded, geo.num_rows};
    sides[num_sides++] = {geo.col_base, geo.col_base + geo.cols_padded, geo.col_base + geo.num_cols};
  }
  for (int k = 0; k < num_sides; ++k) {
    const uint32_t b = s_begin > sides[k].begin ? s_begin : sides[k].begin;
    const uint32_t e = s_end < sides[k].end ? s_end : sides[k].end;
    if (b >= e) continue;
    const uint32_t real_e = e < sides[k].real_end ? e : sides[k].real_end;
    const uint32_t n = real_e > b ? real_e - b : 0;
    // the sort's four arrays: indexed from the range's first plane sample
    uint32_t *keys_in = words + b, *keys_out = words + geo.s_stride + b;
    uint32_t *vals_in = words + 2 * (size_t)geo.s_stride + b;
    uint32_t *vals_out = words + 3 * (size_t)geo.s_stride + b;
    const uint32_t *order = vals_in;
    if (n != 0) {
      if (sort) {
        sort_keys_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(
            tmp_stats, b, n, stored_sites, keys_in, vals_in);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        size_t bytes = sort_temp_bytes;
        e0 = hipcub::DeviceRadixSort::SortPairs(sort_temp, bytes, keys_in, keys_out, vals_in,
                                                vals_out, (int)n, 0, 8, stream);
        if (e0 != hipSuccess) return e0;
        order = vals_out;
      } else {
        identity_order_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(b, n, vals_in);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
      }
    }
    apply_order_kernel<<<dim3((e - b + 255) / 256), dim3(256), 0, stream>>>(
        geo, b, e, n, order, tmp_stats, tmp_prefix, perm, stats, prefix);
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) return e1;
  }
  return hipSuccess;
}

}  // namespace cuking
