/*
 * king_oracle.c -- CPU restatement of the reference's KING hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see king_oracle.h.
 *
 * Plain C, one translation unit, no dependencies beyond libc (+ OpenMP for
 * the multi-threaded baseline).  Built by oracle/Makefile.
 */
#include "king_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static uint32_t ceil_div_u32(uint32_t a, uint32_t b) { /* cuking.cu:123-126 */
  return (uint32_t)(((uint64_t)a + b - 1) / b);
}

static uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* cuking.cu:130-152 */
int orc_submatrix_init(orc_submatrix *sm, uint32_t num_samples,
                       uint32_t split_factor, uint32_t shard_index) {
  if (split_factor == 0) return -1;
  const uint64_t num_shards =
      (uint64_t)split_factor * ((uint64_t)split_factor + 1) / 2;
  if (shard_index >= num_shards) return -1;

  /* Walk the rows of the upper-triangular block matrix (diagonal included)
   * until the running block count passes shard_index (:136-144). */
  uint32_t block_i = 0, block_j = 0, blocks_so_far = 0;
  for (uint32_t row = 0; row < split_factor; ++row) {
    blocks_so_far += split_factor - row;
    if (shard_index < blocks_so_far) {
      block_i = row;
      block_j = split_factor - blocks_so_far + shard_index;
      break;
    }
  }

  const uint32_t size = ceil_div_u32(num_samples, split_factor); /* :147 */
  /* 64-bit products + clamp: the reference's u32 arithmetic wraps when
   * block*size > N (App. C item 4); a clamped empty range is the sane value. */
  const uint64_t ib = (uint64_t)block_i * size, jb = (uint64_t)block_j * size;
  sm->i_begin = (uint32_t)(ib < num_samples ? ib : num_samples);
  sm->i_end = (uint32_t)(ib + size < num_samples ? ib + size : num_samples);
  sm->j_begin = (uint32_t)(jb < num_samples ? jb : num_samples);
  sm->j_end = (uint32_t)(jb + size < num_samples ? jb + size : num_samples);
  return 0;
}

uint32_t orc_num_rows(const orc_submatrix *sm) { return sm->i_end - sm->i_begin; }
uint32_t orc_num_cols(const orc_submatrix *sm) { return sm->j_end - sm->j_begin; }

uint32_t orc_num_samples(const orc_submatrix *sm) { /* :159-162 */
  return sm->i_begin == sm->j_begin ? orc_num_rows(sm)
                                    : orc_num_rows(sm) + orc_num_cols(sm);
}

uint32_t orc_contains(const orc_submatrix *sm, uint32_t index) { /* :165-168 */
  return (sm->i_begin <= index && index < sm->i_end) ||
         (sm->j_begin <= index && index < sm->j_end);
}

uint32_t orc_sample_offset(const orc_submatrix *sm, uint32_t index) { /* :171-175 */
  /* Row samples come first, then the column samples. */
  if (index < sm->i_end) return index - sm->i_begin;
  return (sm->i_end - sm->i_begin) + (index - sm->j_begin);
}

uint32_t orc_padded_sites(uint32_t num_sites) { /* :498-500 */
  return ceil_div_u32(num_sites, 32u) * 32u;
}

uint32_t orc_words_per_sample(uint32_t num_sites) { /* :513 */
  return 2u * ceil_div_u32(orc_padded_sites(num_sites), 64u);
}

void orc_bitset_init(uint64_t *bit_set, size_t num_words) { /* :523 */
  memset(bit_set, 0xFF, num_words * sizeof(uint64_t));
}

static void clear_bit(uint64_t *plane, uint64_t index) { /* :317-323 */
  plane[index >> 6] &= ~((uint64_t)1 << (index & 63u));
}

/* cuking.cu:675-703 */
int orc_pack(const orc_submatrix *sm, uint32_t words_per_sample,
             uint64_t *bit_set, const int64_t *row_idx, const int64_t *col_idx,
             const int32_t *n_alt_alleles, size_t num_triples) {
  const uint32_t plane_words = words_per_sample / 2;
  const uint64_t plane_bits = (uint64_t)plane_words * 64u;
  for (size_t t = 0; t < num_triples; ++t) {
    const int64_t col = col_idx[t];
    if (col < 0 || col > (int64_t)UINT32_MAX ||
        !orc_contains(sm, (uint32_t)col)) {
      continue; /* :677-679: sample not part of this shard */
    }
    const int64_t row = row_idx[t];
    if (row < 0 || (uint64_t)row >= plane_bits) return -3;
    uint64_t *het = bit_set + (uint64_t)orc_sample_offset(sm, (uint32_t)col) *
                                  words_per_sample; /* :683-685 */
    uint64_t *hom_var = het + plane_words;           /* :686 */
    switch (n_alt_alleles[t]) {
      case 0: /* hom-ref: neither bit (:688-691) */
        clear_bit(het, (uint64_t)row);
        clear_bit(hom_var, (uint64_t)row);
        break;
      case 1: /* het: keep the het bit (:692-694) */
        clear_bit(hom_var, (uint64_t)row);
        break;
      case 2: /* hom-var: keep the hom_var bit (:695-697) */
        clear_bit(het, (uint64_t)row);
        break;
      default:
        return -2; /* :698-702 */
    }
  }
  return 0;
}

/* cuking.cu:216-240 */
void orc_pair_counts(const uint64_t *sample_i, const uint64_t *sample_j,
                     uint32_t words_per_sample, orc_counts *out) {
  const uint32_t n = words_per_sample / 2; /* :204 */
  const uint64_t *het_i_w = sample_i, *alt_i_w = sample_i + n; /* :209-210 */
  const uint64_t *het_j_w = sample_j, *alt_j_w = sample_j + n; /* :211-212 */
  uint32_t het_i = 0, het_j = 0, both_het = 0, opp = 0, conc = 0, shared = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint64_t hi = het_i_w[k], ai = alt_i_w[k];
    const uint64_t hj = het_j_w[k], aj = alt_j_w[k];
    const uint64_t ri = ~hi & ~ai; /* hom-ref planes (:221, :225) */
    const uint64_t rj = ~hj & ~aj;
    /* Missing = both bits set; count only sites defined in both (:229). */
    const uint64_t defined = ~(hi & ai) & ~(hj & aj);
    het_i += (uint32_t)__builtin_popcountll(hi & defined);
    het_j += (uint32_t)__builtin_popcountll(hj & defined);
    both_het += (uint32_t)__builtin_popcountll(hi & hj & defined);
    opp += (uint32_t)__builtin_popcountll(((ri & aj) | (ai & rj)) & defined);
    conc += (uint32_t)__builtin_popcountll(((ri & rj) | (ai & aj)) & defined);
    shared += (uint32_t)__builtin_popcountll(defined);
  }
  out->het_i = het_i;
  out->het_j = het_j;
  out->both_het = both_het;
  out->opposing_hom = opp;
  out->concordant_hom = conc;
  out->shared = shared;
}

/* cuking.cu:289-294.  Two float32 roundings (divide, then add); the operands
 * are integers below 2^24 for < 2^22 sites, so numerator and denominator are
 * exact whatever the association order.  min_hets == 0
 * yields -inf or NaN, which fails every `kin > threshold` test. */
float orc_kin(const orc_counts *c) {
  const uint32_t min_hets = min_u32(c->het_i, c->het_j);
  volatile float num = 2.f * (float)c->both_het - 4.f * (float)c->opposing_hom -
                       (float)c->het_i - (float)c->het_j;
  volatile float den = 4.f * (float)min_hets;
  volatile float q = num / den;
  return 0.5f + q;
}

static void fill_result(orc_result *r, uint32_t i, uint32_t j, float kin,
                        const orc_counts *c) { /* :301-307 */
  r->sample_i = i;
  r->sample_j = j;
  r->kin = kin;
  r->ibs0 = c->opposing_hom;
  r->ibs2 = c->concordant_hom + c->both_het;
  r->ibs1 = c->shared - r->ibs0 - r->ibs2;
}

/* Column range of row i: the pairs (i, j) with i < j and j inside the block
 * (:197-199).  Blocks are either diagonal (same range) or strictly above. */
static void col_range(const orc_submatrix *sm, uint32_t i, uint32_t *j0,
                      uint32_t *j1) {
  uint32_t lo = sm->j_begin;
  if (lo <= i) lo = i + 1;
  *j0 = lo;
  *j1 = sm->j_end;
}

uint64_t orc_compute(const orc_submatrix *sm, uint32_t words_per_sample,
                     const uint64_t *bit_set, float kin_threshold,
                     uint32_t max_results, orc_result *results,
                     uint32_t *overflow) {
  uint64_t found = 0;
  *overflow = 0;
  for (uint32_t i = sm->i_begin; i < sm->i_end; ++i) {
    const uint64_t *si =
        bit_set + (uint64_t)orc_sample_offset(sm, i) * words_per_sample;
    uint32_t j0, j1;
    col_range(sm, i, &j0, &j1);
    for (uint32_t j = j0; j < j1; ++j) {
      const uint64_t *sj =
          bit_set + (uint64_t)orc_sample_offset(sm, j) * words_per_sample;
      orc_counts c;
      orc_pair_counts(si, sj, words_per_sample, &c);
      const float kin = orc_kin(&c);
      if (kin > kin_threshold) { /* strict, :297 */
        if (found < max_results) {
          fill_result(&results[found], i, j, kin, &c);
        } else {
          *overflow = 1; /* :308-312 */
        }
        ++found;
      }
    }
  }
  return found;
}

uint64_t orc_compute_mt(const orc_submatrix *sm, uint32_t words_per_sample,
                        const uint64_t *bit_set, float kin_threshold,
                        uint32_t max_results, orc_result *results,
                        uint32_t *overflow, int num_threads) {
  uint64_t found = 0;
  uint32_t ovf = 0;
#ifdef _OPENMP
  if (num_threads < 1) num_threads = omp_get_max_threads();
#else
  (void)num_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4) num_threads(num_threads)
  for (int64_t ii = (int64_t)sm->i_begin; ii < (int64_t)sm->i_end; ++ii) {
    const uint32_t i = (uint32_t)ii;
    const uint64_t *si =
        bit_set + (uint64_t)orc_sample_offset(sm, i) * words_per_sample;
    uint32_t j0, j1;
    col_range(sm, i, &j0, &j1);
    for (uint32_t j = j0; j < j1; ++j) {
      const uint64_t *sj =
          bit_set + (uint64_t)orc_sample_offset(sm, j) * words_per_sample;
      orc_counts c;
      orc_pair_counts(si, sj, words_per_sample, &c);
      const float kin = orc_kin(&c);
      if (kin > kin_threshold) {
        uint64_t slot;
#pragma omp atomic capture
        slot = found++;
        if (slot < max_results) {
          fill_result(&results[slot], i, j, kin, &c);
        } else {
#pragma omp atomic write
          ovf = 1;
        }
      }
    }
  }
  *overflow = ovf;
  return found;
}

uint64_t orc_all_pairs(const orc_submatrix *sm, uint32_t words_per_sample,
                       const uint64_t *bit_set, uint64_t capacity,
                       uint32_t *out_i, uint32_t *out_j, orc_counts *out_counts,
                       float *out_kin) {
  uint64_t n = 0;
  for (uint32_t i = sm->i_begin; i < sm->i_end; ++i) {
    const uint64_t *si =
        bit_set + (uint64_t)orc_sample_offset(sm, i) * words_per_sample;
    uint32_t j0, j1;
    col_range(sm, i, &j0, &j1);
    for (uint32_t j = j0; j < j1; ++j) {
      if (n >= capacity) return n;
      const uint64_t *sj =
          bit_set + (uint64_t)orc_sample_offset(sm, j) * words_per_sample;
      orc_pair_counts(si, sj, words_per_sample, &out_counts[n]);
      out_i[n] = i;
      out_j[n] = j;
      if (out_kin) out_kin[n] = orc_kin(&out_counts[n]);
      ++n;
    }
  }
  return n;
}

static int result_less(const void *pa, const void *pb) { /* :762-764 */
  const orc_result *a = (const orc_result *)pa, *b = (const orc_result *)pb;
  if (a->sample_i != b->sample_i) return a->sample_i < b->sample_i ? -1 : 1;
  if (a->sample_j != b->sample_j) return a->sample_j < b->sample_j ? -1 : 1;
  if (a->kin < b->kin) return -1;
  if (a->kin > b->kin) return 1;
  return 0;
}

void orc_sort(orc_result *results, size_t n) {
  qsort(results, n, sizeof(orc_result), result_less);
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
