// Synthetic cohort description for `cuking --synthetic=N,M[,seed]` (no reference
// counterpart): founders first, planted relatives at the end --
// 0.5 % duplicates, 1 % parent-child trios, 1 % full siblings, 1 % half
// siblings.  The same plan, number for number, as cuking_amd/synth.py
// (plan_cohort); the genotypes come from the device generator
// (cuking_synth_bitset), whose CPU twin is oracle/synth_oracle.c.
#ifndef CUKING_AMD_HOST_SYNTH_PLAN_H_
#define CUKING_AMD_HOST_SYNTH_PLAN_H_

#include <cstdint>
#include <vector>

namespace cuking_host {

struct CohortPlan {
  std::vector<uint32_t> kind, pa, pb;  // 0 founder, 1 duplicate of pa, 2 child of pa x pb
  uint32_t num_founders = 0;
};

inline uint64_t Mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

inline CohortPlan PlanCohort(uint32_t n, uint64_t seed) {
  CohortPlan plan;
  plan.kind.assign(n, 0);
  plan.pa.assign(n, 0);
  plan.pb.assign(n, 0);
  const uint32_t n_dup = n / 200, n_po = n / 100;
  const uint32_t n_sib = (n / 100) / 2 * 2, n_half = (n / 100) / 2 * 2;
  const uint32_t n_derived = n_dup + n_po + n_sib + n_half;
  const uint32_t n_f = n - n_derived;
  plan.num_founders = n_f;
  if (n_f < 3) {
    plan.num_founders = n;
    return plan;
  }
  uint64_t state = Mix64(seed ^ 0xC0FFEEull);
  auto pick = [&](int64_t ex0 = -1, int64_t ex1 = -1) {
    while (true) {
      state = Mix64(state + 0x9E3779B97F4A7C15ull);
      const uint64_t f = state % n_f;
      if ((int64_t)f != ex0 && (int64_t)f != ex1) return (uint32_t)f;
    }
  };
  uint32_t s = n_f;
  auto child = [&](uint32_t a, uint32_t b) {
    plan.kind[s] = 2;
    plan.pa[s] = a;
    plan.pb[s] = b;
    ++s;
  };
  for (uint32_t k = 0; k < n_dup; ++k) {
    const uint32_t a = pick();
    plan.kind[s] = 1;
    plan.pa[s] = plan.pb[s] = a;
    ++s;
  }
  for (uint32_t k = 0; k < n_po; ++k) {
    const uint32_t a = pick();
    const uint32_t b = pick(a);
    child(a, b);
  }
  for (uint32_t k = 0; k < n_sib / 2; ++k) {
    const uint32_t a = pick();
    const uint32_t b = pick(a);
    child(a, b);
    child(a, b);
  }
  for (uint32_t k = 0; k < n_half / 2; ++k) {
    const uint32_t a = pick();
    const uint32_t b = pick(a);
    const uint32_t c = pick(a, b);
    child(a, b);
    child(a, c);
  }
  return plan;
}

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_SYNTH_PLAN_H_
