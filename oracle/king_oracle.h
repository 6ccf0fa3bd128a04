/*
 * king_oracle.h -- CPU restatement of the reference's KING hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker (or as the timed CPU baseline).
 * The product (cuking_amd/, include/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (populationgenomics/cuKING) ships no tests,
 * golden vectors or fixtures of any kind, cannot be
 * compiled here (needs nvcc, abseil, google-cloud-cpp, nlohmann/json and a
 * CUDA device) and has no importable Python on this path.  This restatement
 * is therefore pinned only by (1) a naive per-genotype numpy oracle that
 * shares no code or bit tricks with it (oracle/naive_oracle.py), and (2) the
 * hand-checked known-answer case of SURVEY.md App. A.4
 * (tests/golden/kat_4x10.json).
 *
 * Every function cites the reference lines (cuking.cu) it follows.
 */
#ifndef KING_ORACLE_H_
#define KING_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cuking.cu:129-179 (Submatrix): bounds of one block of the upper-triangular
 * block matrix of sample pairs. */
typedef struct {
  uint32_t i_begin, i_end;
  uint32_t j_begin, j_end;
} orc_submatrix;

/* cuking.cu:182-186 (KingResult). */
typedef struct {
  uint32_t sample_i, sample_j;
  float kin;
  uint32_t ibs0, ibs1, ibs2;
} orc_result;

/* The six per-pair sums of cuking.cu:216-240, in the reference's order. */
typedef struct {
  uint32_t het_i, het_j, both_het, opposing_hom, concordant_hom, shared;
} orc_counts;

/* cuking.cu:130-152. Returns 0, or -1 if split_factor == 0 or shard_index is
 * not below k(k+1)/2 (cuking.cu:455-462). Ranges are clamped to num_samples
 * (the reference wraps around when block*size > N). */
int orc_submatrix_init(orc_submatrix *sm, uint32_t num_samples,
                       uint32_t split_factor, uint32_t shard_index);
uint32_t orc_num_rows(const orc_submatrix *sm);                /* :154 */
uint32_t orc_num_cols(const orc_submatrix *sm);                /* :156 */
uint32_t orc_num_samples(const orc_submatrix *sm);             /* :159-162 */
uint32_t orc_contains(const orc_submatrix *sm, uint32_t index);      /* :165-168 */
uint32_t orc_sample_offset(const orc_submatrix *sm, uint32_t index); /* :171-175 */

/* cuking.cu:498-500: sites padded up to a multiple of 32. */
uint32_t orc_padded_sites(uint32_t num_sites);
/* cuking.cu:513: 2 * ceil(padded_sites / 64) 64-bit words per sample. */
uint32_t orc_words_per_sample(uint32_t num_sites);
/* cuking.cu:520-523: every bit set = every genotype missing. */
void orc_bitset_init(uint64_t *bit_set, size_t num_words);

/* cuking.cu:675-703 + :317-323: clears bits for each (row_idx, col_idx,
 * n_alt_alleles) triple whose sample is in the submatrix.  Returns 0; -2 for
 * an n_alt_alleles outside {0,1,2} (:698-702); -3 for a row_idx outside the
 * padded site range (the reference does not check). */
int orc_pack(const orc_submatrix *sm, uint32_t words_per_sample,
             uint64_t *bit_set, const int64_t *row_idx, const int64_t *col_idx,
             const int32_t *n_alt_alleles, size_t num_triples);

/* cuking.cu:216-240: the six masked popcount sums for one pair of samples,
 * each given as its [het words | hom_alt words] block. */
void orc_pair_counts(const uint64_t *sample_i, const uint64_t *sample_j,
                     uint32_t words_per_sample, orc_counts *out);

/* cuking.cu:289-294: float32 between-family kinship from the counts. */
float orc_kin(const orc_counts *c);

/* cuking.cu:197-201 + :284-313 over the whole submatrix, single thread, pairs
 * visited in (i, j) order.  Stores at most max_results records; sets
 * *overflow to 1 if more qualified (:308-312).  Returns the number of
 * qualifying pairs (may exceed max_results, like the reference's counter). */
uint64_t orc_compute(const orc_submatrix *sm, uint32_t words_per_sample,
                     const uint64_t *bit_set, float kin_threshold,
                     uint32_t max_results, orc_result *results,
                     uint32_t *overflow);

/* Same pairs and arithmetic, rows spread over num_threads OpenMP threads
 * (the CPU baseline timed by bench.py).  Output order is unspecified until
 * orc_sort().  Returns the number of qualifying pairs. */
uint64_t orc_compute_mt(const orc_submatrix *sm, uint32_t words_per_sample,
                        const uint64_t *bit_set, float kin_threshold,
                        uint32_t max_results, orc_result *results,
                        uint32_t *overflow, int num_threads);

/* All pairs of a submatrix without threshold: writes counts for every pair
 * (i<j) in (i, j) order into out_counts (and kin into out_kin if non-NULL).
 * Returns the number of pairs written (bounded by capacity). */
uint64_t orc_all_pairs(const orc_submatrix *sm, uint32_t words_per_sample,
                       const uint64_t *bit_set, uint64_t capacity,
                       uint32_t *out_i, uint32_t *out_j, orc_counts *out_counts,
                       float *out_kin);

/* cuking.cu:761-765: sort by (sample_i, sample_j, kin). */
void orc_sort(orc_result *results, size_t n);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* KING_ORACLE_H_ */
