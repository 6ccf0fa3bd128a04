// Device-side synthetic genotype generator for benchmarks and parity runs at
// sizes whose COO Parquet form would be 10^9 .. 10^11 rows.
// The reference has no counterpart.  Specification (integer only, so the CPU
// twin oracle/synth_oracle.c is bit-identical):
//   mix64      splitmix64 finaliser
//   hash3      mix64(mix64(seed + tag*GOLD + a) ^ (b * 0xD1B54A32D192ED03))
//   site AF    AF_LO + (hi32(hash3(seed,1,site,0)) * AF_SPAN >> 32)   (u32 scale)
//   founder    two alleles: lo32 / hi32 of hash3(seed,2,founder,site) < AF
//   missing    lo32(hash3(seed,3,sample,site)) < 1 % (u32 scale)
//   child      one allele from each founder parent; a het parent transmits
//              bit 0 / bit 1 of hash3(seed,4,child,site)
// Output: the reference bitset layout (cuking.cu:507-523).
#include <hip/hip_runtime.h>

#include "king_common.h"

namespace cuking {

namespace {

constexpr uint64_t kTagSite = 1, kTagGeno = 2, kTagMiss = 3, kTagTrans = 4;
constexpr uint32_t kAfLo = 214748364u;     // floor(0.05 * 2^32)
constexpr uint32_t kAfSpan = 1932735283u;  // floor(0.45 * 2^32)
constexpr uint32_t kMissThr = 42949672u;   // floor(0.01 * 2^32)
constexpr uint32_t kKindDup = 1, kKindChild = 2;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t tag,
                                          uint64_t a, uint64_t b) {
  return mix64(mix64(seed + tag * 0x9E3779B97F4A7C15ull + a) ^
               (b * 0xD1B54A32D192ED03ull));
}

__device__ __forceinline__ uint32_t founder_genotype(uint64_t seed,
                                                     uint32_t founder,
                                                     uint32_t site,
                                                     uint32_t af_thr) {
  const uint64_t h = hash3(seed, kTagGeno, founder, site);
  return ((uint32_t)h < af_thr) + ((uint32_t)(h >> 32) < af_thr);
}

__device__ __forceinline__ uint32_t transmit(uint32_t g, uint32_t coin) {
  return g == 1 ? coin : (g >> 1);
}

// One thread = one 64-site word of one sample (both planes).
__global__ __launch_bounds__(256) void synth_kernel(
    const uint64_t seed, const uint32_t *__restrict__ kind,
    const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb,
    const uint32_t sample_begin, const uint32_t sample_end,
    const uint32_t num_sites, const uint32_t words_per_sample,
    uint64_t *__restrict__ bit_set, const uint64_t block_offset) {
  const uint32_t plane = words_per_sample / 2;
  const uint64_t idx = block_offset * blockDim.x + (uint64_t)blockIdx.x * blockDim.x +
                       threadIdx.x;
  const uint64_t total = (uint64_t)(sample_end - sample_begin) * plane;
  if (idx >= total) return;
  const uint32_t row = (uint32_t)(idx / plane);
  const uint32_t w = (uint32_t)(idx % plane);
  const uint32_t s = sample_begin + row;
  const uint32_t k = kind[s];
  const uint32_t a = pa[s], b = pb[s];

  uint64_t het_word = 0, hom_word = 0;
  for (uint32_t bit = 0; bit < 64; ++bit) {
    const uint64_t site64 = (uint64_t)w * 64 + bit;
    uint32_t g = 3;
    if (site64 < num_sites) {
      const uint32_t site = (uint32_t)site64;
      const uint32_t u = (uint32_t)(hash3(seed, kTagSite, site, 0) >> 32);
      const uint32_t af_thr =
          kAfLo + (uint32_t)(((uint64_t)u * kAfSpan) >> 32);
      if ((uint32_t)hash3(seed, kTagMiss, s, site) >= kMissThr) {
        if (k == kKindDup) {
          g = founder_genotype(seed, a, site, af_thr);
        } else if (k == kKindChild) {
          const uint64_t ht = hash3(seed, kTagTrans, s, site);
          g = transmit(founder_genotype(seed, a, site, af_thr),
                       (uint32_t)(ht & 1)) +
              transmit(founder_genotype(seed, b, site, af_thr),
                       (uint32_t)((ht >> 1) & 1));
        } else {
          g = founder_genotype(seed, s, site, af_thr);
        }
      }
    }
    // (het, hom_var): 0 -> 00, 1 -> 10, 2 -> 01, missing -> 11
    het_word |= (uint64_t)((g == 1) | (g == 3)) << bit;
    hom_word |= (uint64_t)((g == 2) | (g == 3)) << bit;
  }
  uint64_t *dst = bit_set + (uint64_t)row * words_per_sample;
  dst[w] = het_word;
  dst[plane + w] = hom_word;
}

// One wavefront that watches the two clocks of its compute unit for a fixed
// wall time: s_memtime ticks with the shader clock, s_memrealtime at a constant
// 100 MHz, so delta / delta x 100 MHz is the clock the chip sustains while
// whatever else is running (MI355X_MICROARCH.md, "DVFS give-back" item 6).  It
// sleeps between looks and leaves after `ticks_100mhz` of real time whatever
// happens.  out = {shader ticks, real ticks}.
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t ticks_100mhz,
                                                         uint64_t *out) {
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  uint64_t r1 = r0;
  while (r1 - r0 < ticks_100mhz) {
    __builtin_amdgcn_s_sleep(127);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
  }
}

}  // namespace

hipError_t launch_clock_probe(uint64_t microseconds, uint64_t *d_out,
                              hipStream_t stream) {
  clock_probe_kernel<<<dim3(1), dim3(64), 0, stream>>>(microseconds * 100, d_out);
  return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, const uint32_t *d_kind,
                        const uint32_t *d_pa, const uint32_t *d_pb,
                        uint32_t sample_begin, uint32_t sample_end,
                        uint32_t num_sites, uint32_t words_per_sample,
                        uint64_t *d_bit_set, hipStream_t stream) {
  const uint64_t total =
      (uint64_t)(sample_end - sample_begin) * (words_per_sample / 2);
  if (total == 0) return hipSuccess;
  const uint64_t blocks = (total + 255) / 256;
  const uint64_t cap = 0xFFFFFFFFull / 256;  // < 2^32 threads per launch
  for (uint64_t done = 0; done < blocks; done += cap) {
    const uint64_t n = blocks - done < cap ? blocks - done : cap;
    synth_kernel<<<dim3((uint32_t)n), dim3(256), 0, stream>>>(
        seed, d_kind, d_pa, d_pb, sample_begin, sample_end, num_sites,
        words_per_sample, d_bit_set, done);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace cuking
