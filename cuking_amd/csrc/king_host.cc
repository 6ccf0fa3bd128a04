// Host-only half of the C ABI (include/cuking_amd.h): Submatrix arithmetic,
// bitset sizing, the host pack with its relaxed atomics, the narrowing step of
// the device pack, the record sort, the per-thread error message.  Plain C++
// (no HIP): hipcc compiles it into libcuking_amd.so, and the sanitizer tests
// compile the same file with g++ -fsanitize=thread / address so that the code
// that runs on many reader threads is instrumented (tests/test_cli.py).
#include "king_host.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <tuple>
#include <vector>

#include "../host/schedule.h"
#include "king_submatrix.h"

using namespace cuking;

namespace {

thread_local std::string g_last_error;

uint32_t ceil_div(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a + b - 1) / b);
}
uint32_t round_up(uint32_t a, uint32_t b) { return ceil_div(a, b) * b; }

}  // namespace

cuking_status cuking_fail(cuking_status code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

cuking_status cuking_check_block(const cuking_submatrix *sm,
                          uint32_t words_per_sample) {
  if (sm == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null submatrix");
  if (sm->i_end < sm->i_begin || sm->j_end < sm->j_begin)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "submatrix ranges are reversed");
  if (!sm_is_diag(*sm) && sm->j_begin < sm->i_end && sm_num_rows(*sm) != 0 &&
      sm_num_cols(*sm) != 0)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "row and column ranges must be identical or disjoint with "
                "rows first");
  if (sm_is_diag(*sm) && sm->i_end != sm->j_end)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "a diagonal block needs identical row and column ranges");
  if (words_per_sample == 0 || (words_per_sample & 1))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "words_per_sample must be a positive even number");
  return CUKING_OK;
}

extern "C" {

const char *cuking_last_error(void) { return g_last_error.c_str(); }
uint32_t cuking_abi_version(void) { return CUKING_ABI_VERSION; }

// ---- host-only helpers ----------------------------------------------------

cuking_status cuking_submatrix_init(cuking_submatrix *sm, uint32_t num_samples,
                                    uint32_t split_factor,
                                    uint32_t shard_index) {
  if (sm == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null submatrix");
  if (split_factor == 0)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "Invalid split factor");
  const uint64_t shards = (uint64_t)split_factor * ((uint64_t)split_factor + 1) / 2;
  if (shard_index >= shards)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "Invalid shard index");
  // Row r of the block triangle starts at shard r*k - r(r-1)/2.
  uint32_t block_i = 0;
  uint64_t first = 0;
  while (first + (split_factor - block_i) <= shard_index) {
    first += split_factor - block_i;
    ++block_i;
  }
  const uint32_t block_j = block_i + (uint32_t)(shard_index - first);
  const uint64_t size = ceil_div(num_samples, split_factor);
  auto clamp = [&](uint64_t x) {
    return (uint32_t)std::min<uint64_t>(x, num_samples);
  };
  sm->i_begin = clamp(block_i * size);
  sm->i_end = clamp(block_i * size + size);
  sm->j_begin = clamp(block_j * size);
  sm->j_end = clamp(block_j * size + size);
  return CUKING_OK;
}

uint32_t cuking_submatrix_num_rows(const cuking_submatrix *sm) { return sm_num_rows(*sm); }
uint32_t cuking_submatrix_num_cols(const cuking_submatrix *sm) { return sm_num_cols(*sm); }
uint32_t cuking_submatrix_num_samples(const cuking_submatrix *sm) { return sm_num_samples(*sm); }
uint32_t cuking_submatrix_contains(const cuking_submatrix *sm, uint32_t index) {
  return sm_contains(*sm, index) ? 1u : 0u;
}
uint32_t cuking_submatrix_sample_offset(const cuking_submatrix *sm, uint32_t index) {
  return sm_sample_offset(*sm, index);
}

uint64_t cuking_submatrix_num_pairs(const cuking_submatrix *sm) {
  const uint64_t r = sm_num_rows(*sm), c = sm_num_cols(*sm);
  if (sm_is_diag(*sm)) return r * (r - (r ? 1 : 0)) / 2;
  // Off-diagonal blocks lie strictly above the diagonal: every (i, j) counts.
  uint64_t n = 0;
  if (sm->j_begin >= sm->i_end) return r * c;
  for (uint32_t i = sm->i_begin; i < sm->i_end; ++i) {
    const uint32_t lo = std::max(sm->j_begin, i + 1);
    if (lo < sm->j_end) n += sm->j_end - lo;
  }
  return n;
}

uint32_t cuking_padded_sites(uint32_t num_sites) { return round_up(num_sites, 32u); }
uint32_t cuking_words_per_sample(uint32_t num_sites) {
  return 2u * ceil_div(cuking_padded_sites(num_sites), 64u);
}
uint64_t cuking_bytes_per_pair(uint32_t words_per_sample) {
  return 2ull * words_per_sample * sizeof(uint64_t);
}

namespace {

// Per reader thread: the AND masks of ONE 64-site word column, per stored sample.
// Input tables are site-major (the Spark writer's order, mt_to_cuking_inputs.py:24-34:
// all samples of a site, then the next site), so consecutive triples fall into the
// same word of different samples' planes: 64 sites x n samples of them per column.
// Collecting their bits here and clearing each (sample, plane) word once replaces
// ~32 atomic read-modify-writes per word (15 ns per triple, most of the host
// pack's time) by one.
struct WordColumn {
  std::vector<uint64_t> het, hom;   // bits to KEEP (all ones = untouched)
  std::vector<uint8_t> dirty;
  std::vector<uint32_t> touched;    // samples with dirty != 0, in arrival order
  void Reserve(uint32_t samples) {
    if (het.size() < samples) {
      het.resize(samples, ~0ull);
      hom.resize(samples, ~0ull);
      dirty.resize(samples, 0);
    }
  }
};

}  // namespace

cuking_status cuking_pack_host(const cuking_submatrix *sm,
                               uint32_t words_per_sample, uint64_t *bit_set,
                               const int64_t *row_idx, const int64_t *col_idx,
                               const int32_t *n_alt_alleles,
                               size_t num_triples) {
  cuking_status st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  const uint32_t plane_words = words_per_sample / 2;
  const uint64_t plane_bits = (uint64_t)plane_words * 64;
  auto clear_bit = [](uint64_t *plane, uint64_t index) {
    __atomic_and_fetch(plane + (index >> 6), ~(1ull << (index & 63)),
                       __ATOMIC_RELAXED);
  };
  // cuking.cu:675-703 per triple; `direct` = straight into the bitset (one atomic AND
  // per bit, cuking.cu:317-323), otherwise through the word column.
  thread_local WordColumn wc_of_thread;
  WordColumn &wc = wc_of_thread;  // (one thread-local lookup per call, not per triple)
  // (short calls, and tables that are not site-major -- detected below -- go direct)
  bool direct = num_triples < 1024;
  if (!direct) wc.Reserve(cuking_submatrix_num_samples(sm));
  uint64_t *const keep_het = wc.het.data(), *const keep_hom = wc.hom.data();
  uint8_t *const dirty = wc.dirty.data();
  uint64_t column = ~0ull;      // word index the masks belong to
  size_t flushes = 0;
  auto flush = [&]() {
    for (const uint32_t s : wc.touched) {
      uint64_t *het = bit_set + (uint64_t)s * words_per_sample + column;
      if (keep_het[s] != ~0ull) __atomic_and_fetch(het, keep_het[s], __ATOMIC_RELAXED);
      if (keep_hom[s] != ~0ull) __atomic_and_fetch(het + plane_words, keep_hom[s], __ATOMIC_RELAXED);
      keep_het[s] = keep_hom[s] = ~0ull;
      dirty[s] = 0;
    }
    wc.touched.clear();
    ++flushes;
  };
  for (size_t t = 0; t < num_triples; ++t) {
    const int64_t col = col_idx[t];
    if (col < 0 || col > 0xFFFFFFFFll || !sm_contains(*sm, (uint32_t)col))
      continue;
    const int64_t row = row_idx[t];
    if (row < 0 || (uint64_t)row >= plane_bits) {
      if (!direct) flush();
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                  "row_idx %lld outside the %llu padded sites", (long long)row,
                  (unsigned long long)plane_bits);
    }
    const int32_t alt = n_alt_alleles[t];
    if (alt < 0 || alt > 2) {
      if (!direct) flush();
      return cuking_fail(CUKING_ERR_FAILED_PRECONDITION,
                  "Invalid value for n_alt_alleles (%d) encountered", alt);
    }
    const uint32_t sample = sm_sample_offset(*sm, (uint32_t)col);
    if (direct) {
      uint64_t *het = bit_set + (uint64_t)sample * words_per_sample;
      if (alt != 1) clear_bit(het, (uint64_t)row);                // 0 and 2: not het
      if (alt != 2) clear_bit(het + plane_words, (uint64_t)row);  // 0 and 1: not hom-alt
      continue;
    }
    const uint64_t w = (uint64_t)row >> 6;
    if (w != column) {
      if (!wc.touched.empty()) flush();
      column = w;
      // Not site-major after all (a column change every few triples): the masks
      // buy nothing, the rest of the call goes direct.
      if (flushes >= 256 && t < 8 * flushes) {
        direct = true;
        --t;  // this triple again, on the direct path
        continue;
      }
    }
    if (!dirty[sample]) {
      dirty[sample] = 1;
      wc.touched.push_back(sample);
    }
    // (branch-free: a mask of all ones where the plane keeps its bit)
    const uint64_t bit = 1ull << (row & 63);
    keep_het[sample] &= ~(alt != 1 ? bit : 0ull);
    keep_hom[sample] &= ~(alt != 2 ? bit : 0ull);
  }
  if (!direct && !wc.touched.empty()) flush();
  return CUKING_OK;
}

cuking_status cuking_narrow_triples(const cuking_submatrix *sm, uint32_t words_per_sample,
                                    const int64_t *row_idx, const int64_t *col_idx,
                                    const int32_t *n_alt_alleles, size_t num_triples,
                                    uint32_t *site, uint32_t *sample_alt,
                                    size_t *num_out) {
  cuking_status st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  if (num_out == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  *num_out = 0;
  if (cuking_submatrix_num_samples(sm) > 0x3FFFFFFFu)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "block holds more than 2^30 samples");
  const uint64_t plane_bits = (uint64_t)(words_per_sample / 2) * 64;
  size_t w = 0;
  for (size_t t = 0; t < num_triples; ++t) {
    const int64_t col = col_idx[t];
    if (col < 0 || col > 0xFFFFFFFFll || !sm_contains(*sm, (uint32_t)col)) continue;
    const int64_t row = row_idx[t];
    if (row < 0 || (uint64_t)row >= plane_bits)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                  "row_idx %lld outside the %llu padded sites", (long long)row,
                  (unsigned long long)plane_bits);
    const int32_t g = n_alt_alleles[t];
    if (g < 0 || g > 2)
      return cuking_fail(CUKING_ERR_FAILED_PRECONDITION,
                  "Invalid value for n_alt_alleles (%d) encountered", g);
    site[w] = (uint32_t)row;
    sample_alt[w] = sm_sample_offset(*sm, (uint32_t)col) | ((uint32_t)g << 30);
    ++w;
  }
  *num_out = w;
  return CUKING_OK;
}

// ---- schedules of a block over the GPUs of a node (host/schedule.h) ----------
void cuking_schedule_tile_partition(uint64_t num_tiles, uint32_t world, uint64_t *out) {
  if (world == 0 || out == nullptr) return;
  const auto parts = cuking_host::TilePartition(num_tiles, world);
  for (uint32_t r = 0; r < world; ++r) {
    out[2 * r] = parts[r].begin;
    out[2 * r + 1] = parts[r].end;
  }
}

cuking_status cuking_schedule_weighted_tile_partition(uint64_t num_tiles, const double *weights,
                                                      uint32_t world, uint64_t *out) {
  if (world == 0 || weights == nullptr || out == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<double> w(weights, weights + world);
  for (double x : w)
    if (!(x > 0) || !std::isfinite(x))
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "weights must be positive");
  const auto parts = cuking_host::WeightedTilePartition(num_tiles, w);
  for (uint32_t r = 0; r < world; ++r) {
    out[2 * r] = parts[r].begin;
    out[2 * r + 1] = parts[r].end;
  }
  return CUKING_OK;
}

uint64_t cuking_schedule_calibration_tiles(uint64_t num_tiles, uint32_t world) {
  return cuking_host::CalibrationTiles(num_tiles, world);
}

uint32_t cuking_schedule_chunk_ranges(uint32_t num_samples, uint32_t tile, uint32_t num_chunks,
                                      uint32_t *out) {
  if (tile == 0 || out == nullptr) return 0;
  const auto chunks = cuking_host::ChunkRanges(num_samples, tile, num_chunks);
  for (size_t c = 0; c < chunks.size(); ++c) {
    out[2 * c] = chunks[c].begin;
    out[2 * c + 1] = chunks[c].end;
  }
  return (uint32_t)chunks.size();
}

uint32_t cuking_schedule_staged_steps(uint32_t num_samples, uint32_t tile, uint32_t world,
                                      uint32_t rank, uint32_t num_chunks, uint32_t *out) {
  if (tile == 0 || world == 0 || rank >= world || out == nullptr) return 0;
  const auto steps = cuking_host::StagedSchedule(num_samples, tile, world, rank, num_chunks);
  for (size_t k = 0; k < steps.size(); ++k) {
    uint32_t *o = out + 6 * k;
    o[0] = steps[k].chunk.begin;
    o[1] = steps[k].chunk.end;
    o[2] = steps[k].has_rect ? 1u : 0u;
    o[3] = steps[k].row_begin;
    o[4] = steps[k].row_end;
    o[5] = steps[k].row_step;
  }
  return (uint32_t)steps.size();
}

void cuking_sort_results(cuking_result *results, size_t num_results) {
  std::sort(results, results + num_results,
            [](const cuking_result &a, const cuking_result &b) {
              return std::tie(a.sample_i, a.sample_j, a.kin) <
                     std::tie(b.sample_i, b.sample_j, b.kin);
            });
}

}  // extern "C"
