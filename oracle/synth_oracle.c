/*
 * synth_oracle.c -- CPU twin of the device-side synthetic genotype generator
 * (cuking_amd/csrc/synth.hip).  TEST INFRASTRUCTURE ONLY (see king_oracle.h).
 *
 * The reference has no synthetic inputs; the workload is the one SURVEY.md
 * section 8(d) defines: Hardy-Weinberg genotypes with per-site allele frequency
 * ~ U(0.05, 0.5), 1 % missing, planted duplicates / parent-child / full-sib /
 * half-sib samples.  All randomness is a counter-based integer hash, so this
 * file and the HIP generator produce bit-identical bitsets (integer compares
 * only, no floating point).
 *
 * Output layout = the reference's bitset (cuking.cu:507-523): per sample
 * [het plane | hom_var plane], site s -> bit s&63 of word s>>6, both bits set
 * = missing, padding sites missing.
 */
#include "synth_oracle.h"

#include <string.h>

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser */
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

static uint64_t hash3(uint64_t seed, uint64_t tag, uint64_t a, uint64_t b) {
  return mix64(mix64(seed + tag * 0x9E3779B97F4A7C15ull + a) ^
               (b * 0xD1B54A32D192ED03ull));
}

/* Alt-allele threshold of a site on the u32 scale: AF ~ U(0.05, 0.5). */
static uint32_t site_af_threshold(uint64_t seed, uint32_t site) {
  const uint32_t u = (uint32_t)(hash3(seed, SYN_TAG_SITE, site, 0) >> 32);
  return SYN_AF_LO + (uint32_t)(((uint64_t)u * SYN_AF_SPAN) >> 32);
}

static uint32_t founder_genotype(uint64_t seed, uint32_t founder, uint32_t site,
                                 uint32_t af_thr) {
  const uint64_t h = hash3(seed, SYN_TAG_GENO, founder, site);
  return ((uint32_t)h < af_thr) + ((uint32_t)(h >> 32) < af_thr);
}

/* Allele passed on by a parent with genotype g; `coin` decides for a het. */
static uint32_t transmit(uint32_t g, uint32_t coin) {
  return g == 1 ? coin : (g >> 1);
}

uint32_t syn_genotype(uint64_t seed, const uint32_t *kind, const uint32_t *pa,
                      const uint32_t *pb, uint32_t sample, uint32_t site) {
  const uint32_t af_thr = site_af_threshold(seed, site);
  const uint64_t hm = hash3(seed, SYN_TAG_MISS, sample, site);
  if ((uint32_t)hm < SYN_MISS_THR) return 3; /* missing */
  switch (kind[sample]) {
    case SYN_KIND_DUP:
      return founder_genotype(seed, pa[sample], site, af_thr);
    case SYN_KIND_CHILD: {
      const uint64_t ht = hash3(seed, SYN_TAG_TRANS, sample, site);
      const uint32_t ga = founder_genotype(seed, pa[sample], site, af_thr);
      const uint32_t gb = founder_genotype(seed, pb[sample], site, af_thr);
      return transmit(ga, (uint32_t)(ht & 1)) +
             transmit(gb, (uint32_t)((ht >> 1) & 1));
    }
    default:
      return founder_genotype(seed, sample, site, af_thr);
  }
}

void syn_fill_bitset(uint64_t seed, const uint32_t *kind, const uint32_t *pa,
                     const uint32_t *pb, uint32_t sample_begin,
                     uint32_t sample_end, uint32_t num_sites,
                     uint32_t words_per_sample, uint64_t *bit_set) {
  const uint32_t plane = words_per_sample / 2;
  for (uint32_t s = sample_begin; s < sample_end; ++s) {
    uint64_t *het = bit_set + (uint64_t)(s - sample_begin) * words_per_sample;
    uint64_t *hom = het + plane;
    for (uint32_t w = 0; w < plane; ++w) {
      uint64_t hw = 0, aw = 0;
      for (uint32_t b = 0; b < 64; ++b) {
        const uint64_t site = (uint64_t)w * 64 + b;
        uint32_t g = 3;
        if (site < num_sites) {
          g = syn_genotype(seed, kind, pa, pb, s, (uint32_t)site);
        }
        /* (het, hom_var): 0 -> 00, 1 -> 10, 2 -> 01, missing -> 11 */
        if (g == 1 || g == 3) hw |= (uint64_t)1 << b;
        if (g == 2 || g == 3) aw |= (uint64_t)1 << b;
      }
      het[w] = hw;
      hom[w] = aw;
    }
  }
}
