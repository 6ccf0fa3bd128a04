/*
 * synth_oracle.h -- CPU twin of the synthetic genotype generator.
 * TEST INFRASTRUCTURE ONLY (see king_oracle.h).  The constants below are the
 * generator's specification; cuking_amd/csrc/synth.hip restates them.
 */
#ifndef SYNTH_ORACLE_H_
#define SYNTH_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYN_TAG_SITE 1u
#define SYN_TAG_GENO 2u
#define SYN_TAG_MISS 3u
#define SYN_TAG_TRANS 4u

#define SYN_AF_LO 214748364u    /* floor(0.05 * 2^32) */
#define SYN_AF_SPAN 1932735283u /* floor(0.45 * 2^32) */
#define SYN_MISS_THR 42949672u  /* floor(0.01 * 2^32) */

#define SYN_KIND_FOUNDER 0u
#define SYN_KIND_DUP 1u   /* copy of founder pa[s] (own missingness) */
#define SYN_KIND_CHILD 2u /* child of founders pa[s], pb[s] */

/* Genotype (0, 1, 2; 3 = missing) of `sample` at `site`. */
uint32_t syn_genotype(uint64_t seed, const uint32_t *kind, const uint32_t *pa,
                      const uint32_t *pb, uint32_t sample, uint32_t site);

/* Fills the reference-layout bitset rows of samples [sample_begin,
 * sample_end) (row 0 of bit_set = sample_begin). */
void syn_fill_bitset(uint64_t seed, const uint32_t *kind, const uint32_t *pa,
                     const uint32_t *pb, uint32_t sample_begin,
                     uint32_t sample_end, uint32_t num_sites,
                     uint32_t words_per_sample, uint64_t *bit_set);

#ifdef __cplusplus
}
#endif
#endif /* SYNTH_ORACLE_H_ */
