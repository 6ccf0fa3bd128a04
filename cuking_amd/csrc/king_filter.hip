// The filter variant of the KING pair kernel on the gfx950 matrix cores: ONE
// plane product per pair and site instead of four, as a rigorous upper bound on
// kinship, and the exact sums only for the pairs the bound lets through.
//
// With g = -1 / 0 / +1 for hom-ref / het / hom-alt, the reference's numerator
// (cuking.cu:289-294) is -X with
//     X = het_i + het_j - 2 both_het + 4 opposing_hom
//       = sum over the sites defined in both samples of (g_i - g_j)^2
//       = Y_i.D_j + D_i.Y_j - 2 T_i.T_j
// (Y homozygous and defined, D defined, T = R - A; king_mfma.hip "Four
// products").  Y_i.D_j = |Y_i| - Y_i.M_j >= |Y_i| - |M_j| (M missing), and
// het_i = H_i.D_j <= |H_i|, so with the per-sample counts u = |Y| - |M| and |H|
//     X >= u_i + u_j - 2 q,   q = T_i.T_j,
//     kin = 1/2 - X / (4 min(het_i, het_j)) <= 1/2 - (u_i + u_j - 2 q) / (4 min(|H_i|, |H_j|)).
// A pair can only pass `kin > threshold` (0 < threshold < 1/2) when
//     u_i + u_j - 2 q < (2 - 4 threshold) min(|H_i|, |H_j|) + margin,
// the margin (8 sites) covering the float32 roundings of the reference's divide
// and add and of this test (launch_filter's callers keep bitsets below 2^22
// sites, where every term is an exact float).  The slack of the bound is the
// missingness: |M_i| + |M_j| sites, i.e. ~0.01 in kinship at a 1 % missing rate
// with a third of the sites heterozygous; unrelated pairs sit around 0, so at
// the usual thresholds (>= 0.04) next to nothing but real records gets through.
//
// Pipeline per launch chunk (<= kFilterChunkTiles tiles of 256 x 256 pairs):
//   1. king_filter_kernel: q for every pair on the matrix cores (T from the
//      two-bit T2 layout, king_common.h: +-2.0 in fp4 after one v_and -- or a
//      shift and a v_and -- per fragment dword; the accumulators carry 4 q), the
//      test above per pair; candidates are appended to a list, or, when a 128 x
//      128 quadrant has more than quadrant_cap of them (or the list is full), the
//      quadrant is put on the dense list.  Once most finished quadrants of a
//      launch have gone dense the remaining tiles hand theirs over unexamined.
//   2. king_refine_kernel: one wavefront per candidate, the reference's own six
//      sums straight from the bitset (king_kernels.hip stream kernel), exact
//      kinship, record.
//   3. the four-product kernel (king_mfma.hip, tile-list mode) over the dense
//      quadrants -- nothing, on ordinary cohorts.
// Same records as every other variant, whatever the data: the bound only
// decides WHO computes a pair exactly.
//
// Round 4, all inside filter_tile() below: check points inside the k loop (a tile
// none of whose pairs can still become a candidate leaves, and so does one that
// holds a few, handing them to the candidate list: "Check points"); rotated tiles
// (a tile starts its k loop where the tiles of its XCD are and wraps around, so
// that they share their operands through the XCD's L2: "Rotated tiles"); tiles
// that give up leave for a gated launch of the four-product kernel instead of
// appending to a list; king_filter_persistent_kernel, the same launch from one
// resident workgroup per CU (measured, off by default).
//
// Workgroup = 256 x 256 pairs, 4 wavefronts of 128 x 128 = 4 x 4 MFMA blocks (256
// accumulator registers), k-step = 256 sites = 4 slices of 64, 5 LDS stages of
// 32 KiB by LDS-DMA.  Per k-step and wavefront: 64 MFMAs, 16 ds_read_b128, 192
// VALU (128 v_and + 64 v_lshl), 8 requests of 1 KiB.  DESIGN.md 4.0 has the
// measurements; profiles/r03_ablation.txt, r03_filter_curve.txt, r04_l2_probe.txt and
// r04_tile_gaps.txt the raw numbers.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <atomic>
#include <stdlib.h>

#include "king_common.h"
#include "king_device.h"

// The LDS-DMA statements below write M0 and say so in their clobber lists.
#pragma clang diagnostic ignored "-Winline-asm"

namespace cuking {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kT = (int)kFilterTile;
// LDS stages: 5 of 32 KiB (two units of 64 sites per k-half: a k-step of 256 sites), or
// -DCUKING_FILTER_FINE=1: 10 of 16 KiB (one unit: 128 sites per k-step, a hand-over per
// 32 MFMAs, 8 instead of 3 k-steps between a request and the hand-over that needs it).
#ifndef CUKING_FILTER_FINE
#define CUKING_FILTER_FINE 0
#endif
constexpr int kUnits = CUKING_FILTER_FINE ? 1 : 2;   // units per k-half and stage
constexpr int kStages = CUKING_FILTER_FINE ? 10 : 5;
constexpr int kSliceU4 = kT;                          // one (side, k-half, unit): 256 samples
constexpr int kStageU4 = 2 * 2 * kUnits * kSliceU4;   // uint4 per stage
constexpr int kStageReqs = 4 * kUnits;                // requests per wavefront and stage
// requests that may be in flight at a hand-over: the stages after next, plus what the
// k-step has issued before its last slice
constexpr int kSyncVm = (kStages - 3) * kStageReqs + (kStageReqs - 2);
static_assert(kSyncVm < 64, "vmcnt is a 6-bit counter");
static_assert(kStages * kStageU4 * 16 == (int)kFilterLdsBytes, "LDS size");
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
constexpr uint32_t kNoPair = 0xFFFFFFFFu;

// Which stored sample of the reference bitset plane sample `ps` is (the same
// mapping as the prepare kernels), or kNoPair for padding.
__device__ __forceinline__ uint32_t source_sample(const PlaneGeometry &geo, uint32_t ps) {
  if (geo.diag || ps < geo.rows_padded) return ps < geo.num_rows ? ps : kNoPair;
  const uint32_t c = ps - geo.col_base;
  return c < geo.num_cols ? geo.num_rows + c : kNoPair;
}

// One wavefront per plane sample: (|Y| - |M|, |H|) as floats (exact below 2^24
// sites).  Padding sites of the last word are missing (cuking.cu:513-523) and
// count as such; padding samples get (0, 0).  Beside them the same |Y| - |M|
// CUMULATIVE at every phase boundary of the k-steps (king_common.h phase_step:
// cum[x - 1][sample] = over the first 256 phase_step(x) sites, x = 1 .. 127) -- what a
// check point of a tile that started at any phase needs --, and the cohort's sums
// (samples, missing calls, het calls) the kernel picks a check from.
struct CheckWords {
  uint32_t w[kNumCheckShares];  // k-steps behind each share from the first site on (0: no checks)
};
constexpr uint32_t kStatsMaxSteps = kMfmaN4MaxSites / 256;  // k-steps of the widest bitset
__global__ __launch_bounds__(256) void sample_stats_kernel(
    const uint64_t *__restrict__ bits, uint32_t words_per_sample, PlaneGeometry geo,
    float2 *__restrict__ stats, float *__restrict__ cum, unsigned long long *__restrict__ sums,
    uint32_t *__restrict__ steps_out, CheckWords cw, uint32_t s_begin, uint32_t s_end) {
  __shared__ uint8_t phase_of[kStatsMaxSteps];  // the phase a k-step belongs to
  __shared__ int32_t phase_sum[4][kNumPhases];  // per wavefront: |Y| - |M| per phase
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x < kNumCheckShares) steps_out[threadIdx.x] = cw.w[threadIdx.x];
  const uint32_t all_steps = geo.k_words / 8;  // k-steps of 256 sites
  // k-step s is in phase x <=> phase_step(x) <= s < phase_step(x + 1)
  //                        <=> x = ceil(64 (s + 1) / all_steps) - 1
  for (uint32_t st = threadIdx.x; st < all_steps; st += 256) {
    const uint32_t x = (kNumPhases * (st + 1) - 1) / all_steps;
    phase_of[st] = (uint8_t)(x < kNumPhases ? x : kNumPhases - 1);
  }
#pragma unroll
  for (uint32_t k = 0; k < kNumPhases / 64; ++k) phase_sum[wave][lane + 64 * k] = 0;
  __syncthreads();
  const uint32_t ps = s_begin + blockIdx.x * 4 + wave;
  if (ps >= s_end) return;  // whole wavefront
  const uint32_t src = source_sample(geo, ps);
  int32_t yc = 0, mc = 0, hc = 0;
  if (src != kNoPair) {
    const uint32_t n = words_per_sample / 2;
    const uint64_t *het = bits + (uint64_t)src * words_per_sample;
    const uint64_t *hom = het + n;
    constexpr uint32_t kAhead = 4;  // words per lane and plane requested before any is counted
    for (uint32_t w0 = 0; w0 < n; w0 += 64 * kAhead) {
      uint64_t h[kAhead], v[kAhead];
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint32_t w = w0 + 64 * k + lane;
        const bool in = w < n;
        h[k] = in ? het[w] : 0ull;
        v[k] = in ? hom[w] : 0ull;
      }
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint32_t w = w0 + 64 * k + lane;
        const bool in = w < n;  // (beyond the plane: contributes nothing)
        const int32_t y = in ? __popcll(~h[k]) : 0;  // homozygous and defined (missing has the het bit set)
        const int32_t m = __popcll(h[k] & v[k]);     // missing
        yc += y;
        mc += m;
        hc += __popcll(h[k] & ~v[k]);  // het
        // (a k-step is 4 words of 64 sites = 4 neighbouring lanes: their sum first, then ONE
        //  LDS add per k-step -- a dozen lanes share a phase)
        int32_t d = y - m;
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        if ((lane & 3) == 0 && in) atomicAdd(&phase_sum[wave][phase_of[w >> 2]], d);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      yc += __shfl_xor(yc, off);
      mc += __shfl_xor(mc, off);
      hc += __shfl_xor(hc, off);
    }
  }
  // (the wavefront's own LDS adds are done: same wavefront, in order) inclusive scan over
  // the phases -- lane l holds phases 2 l and 2 l + 1 --: the count in front of boundary x + 1
  // is the scan's value at phase x
  static_assert(kNumPhases == 128, "two phases per lane");
  const int32_t p0 = phase_sum[wave][2 * lane], p1 = phase_sum[wave][2 * lane + 1];
  int32_t run = p0 + p1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t up = __shfl_up(run, off);
    if ((int)lane >= off) run += up;
  }
  cum[(size_t)(2 * lane) * geo.s_stride + ps] = (float)(run - p1);
  if (2 * lane + 1 < kNumCum) cum[(size_t)(2 * lane + 1) * geo.s_stride + ps] = (float)run;
  if (lane == 0) {
    stats[ps] = make_float2((float)(yc - mc), (float)hc);
    // (the cohort's sums feed a choice, not a result: a sample of the samples will do --
    //  three device-scope atomics on three addresses for EVERY sample cost 3 ms at 100k)
    if (src != kNoPair && ((blockIdx.x & 15) == 0 || gridDim.x < 64)) {
      __hip_atomic_fetch_add(sums, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sums + 1, (unsigned long long)mc, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sums + 2, (unsigned long long)hc, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  // scale operands 0: the unscaled instruction
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4 /* fp4 */, 4, 0, 0, 0, 0);
}

__device__ __forceinline__ v8i tfrag(const uint4 w, uint32_t mask) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)(w.x & mask);
  r[1] = (int)(w.y & mask);
  r[2] = (int)(w.z & mask);
  r[3] = (int)(w.w & mask);
  return r;
}

// Timing-only builds (wrong results, never shipped; tools/ab_flags.sh):
// -DCUKING_FILTER_ABLATE=1 no LDS-DMA requests, =2 no stage barrier either,
// =3 the shipped loop without the epilogue, =4 the shipped kernel with every tile
// reading tile (0, 0)'s operands.
#ifndef CUKING_FILTER_ABLATE
#define CUKING_FILTER_ABLATE 0
#endif
#ifndef CUKING_FILTER_TIMING
#define CUKING_FILTER_TIMING 0
#endif

__device__ __forceinline__ uint4 shl2(const uint4 w) {
  return make_uint4(w.x << 2, w.y << 2, w.z << 2, w.w << 2);
}

// One LDS-DMA request: lane l's 16 bytes of SRC + OFF land at DST + OFF + 16 l
// (the immediate offset moves source and destination alike).
#define F_ISSUE(SRC, DST, OFF)                                                 \
  if (CUKING_FILTER_ABLATE != 1 && CUKING_FILTER_ABLATE != 2)                  \
  asm volatile("s_mov_b32 m0, %0\n\t"                                          \
               "s_nop 0\n\t"                                                   \
               "global_load_lds_dwordx4 %1, %2 offset:" #OFF                   \
               :                                                               \
               : "s"(DST), "v"(lane16), "s"(SRC)                               \
               : "memory", "m0")

// One tile (or one piece of the k range of a tile) of the launch: what workgroup `wg` of a
// grid of one workgroup per tile does.  Returns 1 when there was nothing left to take (the
// dynamic tail is through: uniform, before anything else), 0 otherwise -- wavefronts return
// from the epilogue one by one.  lds: [kStages][side][k-half][unit][256].
__device__ __forceinline__ int filter_tile(const TiledArgs &a, const uint32_t wg, uint4 *const lds) {

#if CUKING_FILTER_TIMING
  // Timing build (tools/tile_gaps.sh): how long a CU waits for its next workgroup, and how
  // long a workgroup takes to its first request.  Totals 8 .. 11 (100 MHz ticks, summed by
  // thread 0): gaps between a workgroup's exit at the check point and the entry of the next
  // workgroup on the same CU, their number, entry -> first request, their number.  The
  // table of last exits per CU sits in the (idle) slabs of the remainder pieces.
  const uint32_t t_entry = (uint32_t)__builtin_amdgcn_s_memrealtime();
  uint32_t cu_key;
  {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    cu_key = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);  // XCC | SE, SH, CU
  }
  uint32_t *const cu_last_exit = reinterpret_cast<uint32_t *>(a.fsplit_slabs);
  if (threadIdx.x == 0 && cu_last_exit != nullptr && a.tile_done != nullptr) {
    const uint32_t last = __hip_atomic_load(cu_last_exit + cu_key, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    if (last != 0) {
      __hip_atomic_fetch_add(a.filter_totals + 8, (unsigned long long)(t_entry - last),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(a.filter_totals + 9, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
  uint32_t bid = wg;
  // Remainder of a short launch (king_common.h, fsplit_*): piece `part` of the k
  // range of one of the launch's last tiles.
  const bool split = a.fsplit_parts != 0 && wg >= a.fsplit_first;
  uint32_t part = 0, piece = 0;
  if (split) {
    piece = wg - a.fsplit_first;
    part = __builtin_amdgcn_readfirstlane(piece % a.fsplit_parts);
    bid = a.fsplit_tile0 + piece / a.fsplit_parts;
  } else
  if (a.dyn_tiles != 0 && wg >= a.launch_tiles) {
    // dynamic tail (king_common.h): the next of the launch's last dyn_tiles tiles
    // nobody has taken yet -- the XCDs run at rates a few percent apart, and an
    // XCD that gets through its static share early takes more of these.  (The
    // counter is filter_ctrl[2], zeroed in front of every launch.)
    uint32_t *slot = reinterpret_cast<uint32_t *>(lds);
    if (threadIdx.x == 0)
      *slot = __hip_atomic_fetch_add(a.filter_ctrl + kCtrlDyn, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t t = __builtin_amdgcn_readfirstlane(*slot);
    __syncthreads();  // the word is stage memory from here on
    if (t >= a.dyn_tiles) return 1;  // uniform: nothing left
    bid = a.launch_tiles + t;
  } else if (a.xcd_chunk == 1) {
    // patches of 32 consecutive tiles dealt round-robin to the XCDs (king_common.h)
    const uint32_t x = bid & 7, j = bid >> 3;
    bid = (((j >> 5) * 8 + x) << 5) + (j & 31);
    if (bid >= a.launch_tiles) return 0;  // padding (uniform)
  }
  bid = __builtin_amdgcn_readfirstlane(bid);
  uint32_t tr, tc;
  if (!decode_tile_space(a, a.tile_begin + bid, &tr, &tc)) return 0;  // uniform
  tr = __builtin_amdgcn_readfirstlane(tr);
  tc = __builtin_amdgcn_readfirstlane(tc);

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wy = wave >> 1, wx = wave & 1;  // the wavefront's quadrant
  // When the bound does not thin this cohort out (a threshold inside the noise of
  // unrelated pairs, heavy missingness) nearly every quadrant ends on the dense list
  // anyway: once most of at least 512 finished quadrants of the launch have (or their
  // tiles have left at check 0, below), every remaining tile leaves at once -- it
  // touches nothing, and the fallback launch behind this one (the four-product kernel
  // over the chunk in its own order, king_mfma.hip persistent mode) computes every tile
  // that has not set its tile_done flag.  The worst case costs the exact kernel's time
  // plus the first round of this one (short launches: plus an eighth of it, check 0).
// (A/B: k-steps a new tile starts ahead of where the tiles of its XCD are)
#ifndef CUKING_ROTATE_LEAD
#define CUKING_ROTATE_LEAD 0
#endif
#ifndef CUKING_FILTER_GIVE_UP
#define CUKING_FILTER_GIVE_UP 1  // (A/B: 0 = tiles never give up)
#endif
  uint32_t xcd_pos = 0;  // the k-step the tiles of this XCD are at (rotated tiles)
  if (CUKING_FILTER_GIVE_UP && !split && a.tile_done != nullptr) {
    // ONE decision per workgroup (the counters move while the wavefronts read them,
    // and a wavefront that left alone would take its quarter of every stage's
    // requests with it): thread 0 reads, the stage memory carries the verdict.
    uint32_t *verdict = reinterpret_cast<uint32_t *>(lds);
    if (threadIdx.x == 0) {
      uint32_t leave = __hip_atomic_load(a.filter_ctrl + kCtrlAllLeave, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
      // The cohort's own verdict first: where the bound's level for unrelated pairs (from the
      // cohort's mean missing and het rates, as for the check points below) lies four
      // standard deviations above the threshold, it lets every pair through, whatever the
      // tile -- the first tiles need not find that out by computing their product.
      if (leave == 0 && a.cohort_sums != nullptr && a.check_steps != nullptr) {
        const float ns = (float)a.cohort_sums[0], nm = (float)a.cohort_sums[1],
                    nh = (float)a.cohort_sums[2];
        if (ns > 0.f && nh > 0.f) {
          const float sites = 32.f * (float)a.geo.k_words;
          const float m = nm / (ns * sites), h = nh / (ns * sites);
          if (m * (1.f + m / (2.f * h * (1.f - m))) - 4.f * rsqrtf(sites) > a.kin_threshold &&
              (a.check1 >> 16) != 0) {
            leave = 1;
            __hip_atomic_store(a.filter_ctrl + kCtrlAllLeave, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.filter_ctrl + kCtrlGate, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (leave == 0) {
        const uint32_t dense_so_far =
            __hip_atomic_load(a.filter_ctrl + kCtrlDense, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
            __hip_atomic_load(a.filter_ctrl + kCtrlLeft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t finished = __hip_atomic_load(a.filter_ctrl + kCtrlFinished, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        if (finished >= 512 && 2 * dense_so_far > finished) {
          leave = 1;
          __hip_atomic_store(a.filter_ctrl + kCtrlAllLeave, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.filter_ctrl + kCtrlGate, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      verdict[0] = leave;
    }
    // Rotated tiles (below): where the tiles of this XCD are -- workgroups go to the XCDs
    // round-robin by their index.  Every slot says where one of them was and when; brought
    // forward to now by the measured k-step time, the most advanced one counts.
    if (a.rotate == 1 && threadIdx.x < kPosSlots) {
      const uint32_t x = wg & 7;
      const unsigned long long said = __hip_atomic_load(
          reinterpret_cast<const unsigned long long *>(a.filter_ctrl + kCtrlPos) + x * kPosSlots +
              threadIdx.x,
          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t ticks16 = __hip_atomic_load(a.filter_ctrl + kCtrlStepTicks + x,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t pos = (uint32_t)(said >> 32);
      const uint32_t ago = (uint32_t)__builtin_amdgcn_s_memrealtime() - (uint32_t)said;
      if (said != 0 && ticks16 != 0 && ago < 16384u)
        pos += ago * 16u / ticks16 + CUKING_ROTATE_LEAD;
      verdict[1 + threadIdx.x] = pos;
    }
    __syncthreads();
    const bool give_up = verdict[0] != 0;
    if (a.rotate == 1) {
#pragma unroll
      for (uint32_t k = 0; k < kPosSlots; ++k) xcd_pos = max(xcd_pos, verdict[1 + k]);
    }
    __syncthreads();  // the words are stage memory from here on
    if (give_up) {  // uniform across the workgroup; tile_done stays 0
      // (its quadrants count as handed over: "filter_dense_quadrants")
      if (threadIdx.x == 0)
        __hip_atomic_fetch_add(a.filter_totals + kTotalDense, 4ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
  }
  xcd_pos = __builtin_amdgcn_readfirstlane(xcd_pos);
  const uint32_t g = lane >> 5;                  // k-half of the MFMA operand
  const uint32_t lr = lane & 31;                 // row / column inside a block
  uint32_t lane16 = lane * 16;
  const uint32_t s_stride = a.geo.s_stride;
  // k-steps of 256 sites: all of them, or this piece's share
  const uint32_t all_steps = a.geo.k_words / (4 * kUnits);
  // (wave-uniform, but divisions run in vector registers: pinned to SGPRs for the
  //  request addresses)
  const uint32_t k_first = __builtin_amdgcn_readfirstlane(
      split ? part * all_steps / a.fsplit_parts : 0u);
  const uint32_t num_steps = __builtin_amdgcn_readfirstlane(
      split ? (part + 1) * all_steps / a.fsplit_parts - k_first : all_steps);
  // (=4: every tile reads the operands of tile (0, 0) -- all requests hit the L2)
  const uint4 *g_rows = a.t2 + (CUKING_FILTER_ABLATE == 4 ? 0u : (uint64_t)tr * kT);
  const uint4 *g_cols = a.t2 + a.geo.col_base + (CUKING_FILTER_ABLATE == 4 ? 0u : (uint64_t)tc * kT);

  uint32_t mT;
  asm volatile("s_mov_b32 %0, 0xcccccccc" : "=s"(mT));

  // LDS-DMA: wavefront (side, k-half) fetches that quarter of a stage: 2 units x
  // 4 runs of 64 samples, 1 KiB each.  Unit c of k-step s, k-half h is unit
  // 4 s + 2 h + c of the T2 layout (the order of the sites inside k does not
  // matter as long as rows and columns agree); every unit feeds TWO slices of 64
  // sites per k-half: its bits 2-3 (set B) and its bits 0-1 (set A).
  const uint32_t dma_side = wave >> 1, dma_h = wave & 1;
  const uint32_t row_bytes = s_stride * 16;  // one unit of the layout
  const char *const g_wave = reinterpret_cast<const char *>(
      (dma_side ? g_cols : g_rows) + ((uint64_t)2 * kUnits * k_first + kUnits * dma_h) * s_stride);
  const uint32_t l_wave = (uint32_t)(uintptr_t)(lds_void_ptr)(
      lds + ((dma_side * 2 + dma_h) * kUnits) * kSliceU4);
  struct Addr { const char *src; uint32_t dst; };  // of unit 0; unit 1: + row_bytes, + 4 KiB
  const uint32_t kstep_bytes = 2 * kUnits * row_bytes;
  // The pipeline runs over one SEGMENT of the piece's k-steps at a time (one segment,
  // unless the tile has check points, below): `seg_src` is the wavefront's first request
  // of the segment, `seg_steps` its k-steps.
  const char *seg_src = g_wave;
  uint32_t seg_steps = num_steps;
  // (a rotated tile, below: k-step `seg_wrap` of the segment is the bitset's FIRST again --
  //  all_steps k-steps back; 32-bit scalar selects and one signed product: a select between
  //  two 64-bit addresses goes through vector registers, which the scalar pins cannot take)
  uint32_t seg_wrap = 0xFFFFFFFFu;
  auto addr_of = [&](uint32_t step, uint32_t buf) {
    Addr pa;
    if (step >= seg_steps) step = seg_steps - 1;  // clamped repeats (see king_mfma.hip)
    const int32_t rel = (int32_t)step - (int32_t)(step >= seg_wrap ? all_steps : 0u);
    pa.src = seg_src + (int64_t)rel * (int64_t)kstep_bytes;
    pa.dst = l_wave + buf * (kStageU4 * 16);
    asm volatile("" : "+s"(pa.src), "+s"(pa.dst));
    return pa;
  };
  auto addr_next = [&](const Addr &cur, uint32_t step, uint32_t buf) {
    Addr pa;
    const int32_t adv = (int32_t)(step < seg_steps ? 1u : 0u) -
                        (int32_t)(step == seg_wrap ? all_steps : 0u);
    pa.src = cur.src + (int64_t)adv * (int64_t)kstep_bytes;
    pa.dst = l_wave + buf * (kStageU4 * 16);
    asm volatile("" : "+s"(pa.src), "+s"(pa.dst));
    return pa;
  };
  // The four requests of unit c of the stage `pa` names.
#define F_ISSUE4(PA, C)                                                        \
  {                                                                            \
    const char *src_ = (PA).src + (C) * row_bytes;                             \
    const uint32_t dst_ = (PA).dst + (C) * (kSliceU4 * 16);                    \
    F_ISSUE(src_, dst_, 0);                                                    \
    F_ISSUE(src_, dst_, 1024);                                                 \
    F_ISSUE(src_, dst_, 2048);                                                 \
    F_ISSUE(src_, dst_, 3072);                                                 \
  }

  v16f acc[4][4];
#pragma unroll
  for (int bi = 0; bi < 4; ++bi)
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;

  // This lane's operand words inside a stage (uint4 units).
  uint32_t row_off = ((0 * 2 + g) * kUnits) * kSliceU4 + wy * 128 + lr;
  uint32_t col_off = ((1 * 2 + g) * kUnits) * kSliceU4 + wx * 128 + lr;
  asm volatile("" : "+v"(row_off), "+v"(col_off), "+v"(lane16));
  v8i FA[2][4], FB[2][4];  // T fragments [slice parity][block]
  uint4 RAW[8];            // the words of one unit: rows 0-3, columns 4-7
#define F_READ(K, BUF, C)                                                      \
  RAW[K] = lds[(BUF) * kStageU4 + ((K) < 4 ? row_off : col_off) + (C) * kSliceU4 + ((K) & 3) * 32];
// (plain ANDs and shifts: nothing but data orders them against the MFMAs, and
// left alone the compiler builds every fragment right behind its LDS read, i.e.
// waits for the read it has just issued.  The empty asm statements tie a build
// to the place it is written in: not above the pin of its input, not below the
// pin of its result -- as in king_mfma.hip.)
#define F_PIN4(W) asm volatile("" : "+v"((W).x), "+v"((W).y), "+v"((W).z), "+v"((W).w));
#define F_PINF(F) asm volatile("" : "+v"((F)[0]), "+v"((F)[1]), "+v"((F)[2]), "+v"((F)[3]));
// Set B of word K: bits 2-3 of every nibble as they are.
#define F_BUILD_B(NXT, K)                                                      \
  F_PIN4(RAW[K])                                                               \
  if ((K) < 4) {                                                               \
    FA[NXT][(K) & 3] = tfrag(RAW[K], mT);                                      \
    F_PINF(FA[NXT][(K) & 3])                                                   \
  } else {                                                                     \
    FB[NXT][(K) & 3] = tfrag(RAW[K], mT);                                      \
    F_PINF(FB[NXT][(K) & 3])                                                   \
  }
// Set A of word K: bits 0-1 moved up to bits 2-3.
#define F_BUILD_A(NXT, K)                                                      \
  F_PIN4(RAW[K])                                                               \
  if ((K) < 4) {                                                               \
    FA[NXT][(K) & 3] = tfrag(shl2(RAW[K]), mT);                                \
    F_PINF(FA[NXT][(K) & 3])                                                   \
  } else {                                                                     \
    FB[NXT][(K) & 3] = tfrag(shl2(RAW[K]), mT);                                \
    F_PINF(FB[NXT][(K) & 3])                                                   \
  }
#define F_MMA(CUR, BI, BJ) acc[BI][BJ] = mma(FA[CUR][BI], FB[CUR][BJ], acc[BI][BJ]);
#define F_BAR __builtin_amdgcn_sched_barrier(0);
  // Slice B of unit c (fragment set CUR): 16 MFMAs; in their gaps the set-A
  // fragments of the same words (8 VALU per fragment: every second gap) and two
  // requests.
#define F_SLICE_B(CUR, NXT, PA, DC, OFF0, OFF1)                                \
  {                                                                            \
    const char *src_ = (PA).src + (DC) * row_bytes;                            \
    const uint32_t dst_ = (PA).dst + (DC) * (kSliceU4 * 16);                   \
    F_ISSUE(src_, dst_, OFF0);                                                 \
    F_MMA(CUR, 0, 0) F_BAR                                                     \
    F_BUILD_A(NXT, 0) F_MMA(CUR, 0, 1) F_BAR                                   \
    F_MMA(CUR, 0, 2) F_BAR                                                     \
    F_BUILD_A(NXT, 1) F_MMA(CUR, 0, 3) F_BAR                                   \
    F_MMA(CUR, 1, 0) F_BAR                                                     \
    F_BUILD_A(NXT, 2) F_MMA(CUR, 1, 1) F_BAR                                   \
    F_MMA(CUR, 1, 2) F_BAR                                                     \
    F_BUILD_A(NXT, 3) F_MMA(CUR, 1, 3) F_BAR                                   \
    F_ISSUE(src_, dst_, OFF1);                                                 \
    F_MMA(CUR, 2, 0) F_BAR                                                     \
    F_BUILD_A(NXT, 4) F_MMA(CUR, 2, 1) F_BAR                                   \
    F_MMA(CUR, 2, 2) F_BAR                                                     \
    F_BUILD_A(NXT, 5) F_MMA(CUR, 2, 3) F_BAR                                   \
    F_MMA(CUR, 3, 0) F_BAR                                                     \
    F_BUILD_A(NXT, 6) F_MMA(CUR, 3, 1) F_BAR                                   \
    F_MMA(CUR, 3, 2) F_BAR                                                     \
    F_BUILD_A(NXT, 7) F_MMA(CUR, 3, 3) F_BAR                                   \
  }
  // Slice A of a unit (fragment set CUR): 16 MFMAs; in the gaps of the first
  // eight the LDS reads of the NEXT unit's words (RBUF, RC; behind the stage
  // hand-over if SYNC: that read is the first of a new stage) and two requests,
  // in the gaps of the last eight that unit's set-B fragments.
#define F_SLICE_A(CUR, NXT, RBUF, RC, SYNC, PA, DC, OFF0, OFF1)                \
  {                                                                            \
    if ((SYNC) && CUKING_FILTER_ABLATE != 2) {                                 \
      if (CUKING_FILTER_ABLATE != 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(kSyncVm)); \
      __syncthreads();                                                         \
    }                                                                          \
    const char *src_ = (PA).src + (DC) * row_bytes;                            \
    const uint32_t dst_ = (PA).dst + (DC) * (kSliceU4 * 16);                   \
    F_READ(0, RBUF, RC) F_READ(1, RBUF, RC)                                    \
    F_ISSUE(src_, dst_, OFF0);                                                 \
    F_MMA(CUR, 0, 0) F_MMA(CUR, 0, 1) F_BAR                                    \
    F_READ(2, RBUF, RC) F_READ(3, RBUF, RC)                                    \
    F_MMA(CUR, 0, 2) F_MMA(CUR, 0, 3) F_BAR                                    \
    F_READ(4, RBUF, RC) F_READ(5, RBUF, RC)                                    \
    F_ISSUE(src_, dst_, OFF1);                                                 \
    F_MMA(CUR, 1, 0) F_MMA(CUR, 1, 1) F_BAR                                    \
    F_READ(6, RBUF, RC) F_READ(7, RBUF, RC)                                    \
    F_MMA(CUR, 1, 2) F_MMA(CUR, 1, 3) F_BAR                                    \
    F_BUILD_B(NXT, 0) F_MMA(CUR, 2, 0) F_BAR                                   \
    F_BUILD_B(NXT, 1) F_MMA(CUR, 2, 1) F_BAR                                   \
    F_BUILD_B(NXT, 2) F_MMA(CUR, 2, 2) F_BAR                                   \
    F_BUILD_B(NXT, 3) F_MMA(CUR, 2, 3) F_BAR                                   \
    F_BUILD_B(NXT, 4) F_MMA(CUR, 3, 0) F_BAR                                   \
    F_BUILD_B(NXT, 5) F_MMA(CUR, 3, 1) F_BAR                                   \
    F_BUILD_B(NXT, 6) F_MMA(CUR, 3, 2) F_BAR                                   \
    F_BUILD_B(NXT, 7) F_MMA(CUR, 3, 3) F_BAR                                   \
  }
  // k-step s requests stage s + kStages - 1 into the buffer stage s - 1 left: every
  // wavefront finished reading it before the hand-over of k-step s - 1.  The
  // hand-over of k-step s (stage s + 1 must have landed) comes in its last slice:
  // in flight then may be the stages after s + 1 and the requests of the newest one
  // that the slices before the last have issued (kSyncVm: 2 x 8 + 6 = 22).
#if CUKING_FILTER_FINE
#define F_KSTEP                                                                \
  {                                                                            \
    const uint32_t nbuf = buf == kStages - 1 ? 0 : buf + 1;                    \
    F_SLICE_B(0, 1, pa, 0, 0, 1024)                                            \
    F_SLICE_A(1, 0, nbuf, 0, true, pa, 0, 2048, 3072)                          \
    pa = addr_next(pa, step + kStages, buf);                                   \
    buf = nbuf;                                                                \
    ++step;                                                                    \
  }
#else
#define F_KSTEP                                                                \
  {                                                                            \
    const uint32_t nbuf = buf == kStages - 1 ? 0 : buf + 1;                    \
    F_SLICE_B(0, 1, pa, 0, 0, 1024)                                            \
    F_SLICE_A(1, 0, buf, 1, false, pa, 0, 2048, 3072)                          \
    F_SLICE_B(0, 1, pa, 1, 0, 1024)                                            \
    F_SLICE_A(1, 0, nbuf, 0, true, pa, 1, 2048, 3072)                          \
    pa = addr_next(pa, step + kStages, buf);                                   \
    buf = nbuf;                                                                \
    ++step;                                                                    \
  }
#endif

  // --- Check points (king_common.h).  X = sum over the sites of (g_i - g_j)^2 has only
  // non-negative terms, so its sum over a PREFIX of the sites is a lower bound of X, and
  // so is this kernel's bound of that prefix sum, u'_i + u'_j - 2 q' (u' over the prefix,
  // sample_stats_kernel).  A pair can only pass the threshold when X < t min(|H_i|, |H_j|) +
  // margin (|H| over ALL sites: the epilogue's own test): a tile none of whose 65,536 pairs
  // satisfies u'_i + u'_j - 2 q' < that bound at the check point holds no record, whatever
  // the remaining sites say -- it leaves.  For unrelated samples that happens from a share
  // (1 - 2 thr) / (1 - 2 kappa) of the sites on, kappa = the level of the bound for unrelated
  // pairs (their mean m (1 + m / (2 h (1 - m))) at missing rate m and het rate h, plus 4.6
  // standard deviations 1 / sqrt(sites): tools/bound_tiers.py, profiles/r04_bound_tiers.txt):
  // 0.88 of the sites at the default threshold and 1 % missing calls.  Every workgroup picks
  // the same entry of the share menu from the cohort's sums.  (Check 0 is a FORECAST for
  // short launches: the same test with the bound scaled to an eighth of the sites, counted
  // per quadrant; a tile whose quadrants mostly look dense leaves for the exact kernel
  // there instead of at its end.)  The pipeline is DRAINED at a check point -- the segment
  // before it ends like a tile (every request landed, no fragment built ahead), the one
  // behind it starts like a tile -- so that the check has the LDS for its per-sample
  // values and the register file for its sweep, and the k loop's registers are not live
  // across it: ~8 us per check of a 500 us tile.
  //
  // --- Rotated tiles.  The 32 tiles an XCD holds at a time are a patch of the tile space
  // (8 rows x 4 columns: 12 strips of operands for 32 tiles), but they share those strips
  // through the XCD's 4 MiB L2 only while they read the same k-steps at about the same
  // time -- 21 k-steps of the patch fit.  Tiles that all start at k-step 0 do so in the
  // first round of a launch and drift apart from there (L2 hit rate 0.36, 555 GB from the
  // fabric per pass of configs[2]; one launch per round: 0.80 and 170 GB, and the chip
  // holds 1.96-1.99 GHz instead of 1.86: tools/l2_probe.sh).  The order of the sites
  // inside a sum does not matter, so a tile STARTS where the tiles of its XCD are: at the
  // phase boundary (king_common.h phase_step: 128 phases) nearest to the k-step the most
  // advanced of them has published, runs to the end of the sites, wraps around (a segment
  // boundary like a check point's) and ends where it started.  A check point sits behind a
  // share of the k-steps as before; the per-sample counts over "phases [p, p + e)" are
  // differences of the cumulative counts sample_stats_kernel leaves.
  uint32_t chk0 = 0, chk1 = 0, entry1 = 0, share1 = 0;
  uint32_t phase = 0, k0 = 0, wrap = 0, start_abs = 0;
  if (!CUKING_FILTER_FINE && !split && a.check_steps != nullptr && a.tile_done != nullptr) {
    if (a.rotate != 0 && all_steps >= a.rotate_min_steps) {
      uint32_t base = 0;
      if (a.rotate == 1) {
        const uint32_t r = xcd_pos % all_steps;
        base = xcd_pos - r;
        phase = (r * kNumPhases + all_steps / 2) / all_steps;
        if (phase >= kNumPhases) {
          phase = 0;
          base += all_steps;
        }
      } else if (a.rotate == 2) {
        phase = (bid * 37u + 11u) & (kNumPhases - 1);  // (test hook)
      } else {
        phase = (a.rotate - 3u) & (kNumPhases - 1);    // (test hook)
      }
      k0 = phase_step(all_steps, phase);
      start_abs = base + k0;
      wrap = k0 != 0 ? all_steps - k0 : 0u;
    }
    // kappa: the level of the bound for this cohort's unrelated pairs (from its mean missing
    // and het rates) plus 4.6 standard deviations and a little
    float kappa = -1.f;
    {
      const float ns = (float)a.cohort_sums[0], nm = (float)a.cohort_sums[1],
                  nh = (float)a.cohort_sums[2];
      if (ns > 0.f && nh > 0.f) {
        const float sites = 256.f * (float)all_steps;
        const float m = nm / (ns * sites), h = nh / (ns * sites);
        kappa = m * (1.f + m / (2.f * h * (1.f - m))) + 4.6f * rsqrtf(sites) + 0.003f;
      }
    }
    // Check 0 (the forecast; a.check0: 1 = this is a short launch, 2 = forced): only for a
    // cohort whose unrelated pairs come anywhere near the threshold -- a drain and a sweep
    // per tile (1.5-3 % of configs[1]) that a clean cohort need not pay; a cohort that is
    // clean on average but holds a few bad samples then carries its dense tiles to their
    // end before the quadrant lists take them over.
    // k-steps from phase boundary `phase` on that cover `share` phases (around the end)
    auto steps_of = [&](uint32_t share) {
      const uint32_t hi = phase + share * kPhasesPerShare;
      return hi <= kNumPhases ? phase_step(all_steps, hi) - k0
                              : (all_steps - k0) + phase_step(all_steps, hi - kNumPhases);
    };
    if ((a.check0 == 2 || (a.check0 == 1 && kappa > 0.7f * a.kin_threshold)) &&
        a.check_steps[0] != 0)
      chk0 = steps_of(kCheckShares64[0]);
    const uint32_t sw1 = a.check1 & 0xFFu;
    if (sw1 >= 2) {
      entry1 = sw1 - 2;
    } else if (sw1 == 1 && kappa >= 0.f) {
      // The share behind which no unrelated pair of this cohort is still under the bound, in
      // 64ths, rounded up -- any of them: the counts are there at every phase boundary.  (A
      // check costs a tile that leaves ~1.5 % -- the drain, the sweep -- and one that stays
      // ~3 % -- the refill as well; a tile that still holds a few live pairs leaves all the
      // same, handing them over, so the share needs no margin: up to 62/64.  The menu of
      // king_common.h is what "filter_check1" = 2 + k forces.)
      const float f64 = 64.f * (1.f - 2.f * a.kin_threshold) / (1.f - 2.f * kappa);
      if (f64 <= 62.f) {
        share1 = (uint32_t)ceilf(f64);
        if (share1 < 32) share1 = 32;
        entry1 = 1;  // (any entry of the menu: whether the bitset is long enough for checks)
      }
    }
#pragma unroll
    for (uint32_t k = 1; k < kNumCheckShares; ++k)
      if (sw1 >= 2 && k == entry1) share1 = kCheckShares64[k];
    if (share1 != 0 && a.check_steps[entry1] != 0) chk1 = steps_of(share1);
    if (chk0 >= num_steps) chk0 = 0;
    if (chk1 >= num_steps || chk1 <= chk0) chk1 = 0;
  }
  chk0 = __builtin_amdgcn_readfirstlane(chk0);
  chk1 = __builtin_amdgcn_readfirstlane(chk1);
  share1 = __builtin_amdgcn_readfirstlane(share1);
  phase = __builtin_amdgcn_readfirstlane(phase);
  k0 = __builtin_amdgcn_readfirstlane(k0);
  wrap = __builtin_amdgcn_readfirstlane(wrap);
  start_abs = __builtin_amdgcn_readfirstlane(start_abs);
  // u of plane sample idx over the `share` phases from this tile's first on (`total`: over
  // all sites): cumulative counts in front of the inner boundaries, nothing in front of 0
  auto prefix_u_of = [&](uint32_t share, size_t idx, float total) {
    const uint32_t hi = phase + share * kPhasesPerShare;
    const uint32_t xb = hi > kNumPhases ? hi - kNumPhases : hi;  // (uniform)
    float u = hi > kNumPhases ? total : 0.f;
    if (xb == kNumPhases)
      u += total;
    else if (xb != 0)
      u += a.prefix_u[(size_t)(xb - 1) * s_stride + idx];
    if (phase != 0) u -= a.prefix_u[(size_t)(phase - 1) * s_stride + idx];
    return u;
  };
  // The k-step this tile has reached, for the tiles of the XCD that start next (one lane of
  // one wavefront: the branch is scalar, the lane mask is set by hand -- a divergent branch
  // here makes the compiler treat the segment loop as divergent)
  unsigned long long *const pos_word =
      reinterpret_cast<unsigned long long *>(a.filter_ctrl + kCtrlPos) +
      (wg & 7) * kPosSlots + ((wg >> 3) & (kPosSlots - 1));
  uint32_t *const ticks_word = a.filter_ctrl + kCtrlStepTicks + (wg & 7);
  // (the first request of the tile goes out about now)
  const uint32_t t_start = a.rotate == 1 ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
#if CUKING_FILTER_TIMING
  if (threadIdx.x == 0 && a.tile_done != nullptr && !split) {
    __hip_atomic_fetch_add(a.filter_totals + 10,
                           (unsigned long long)((uint32_t)__builtin_amdgcn_s_memrealtime() - t_entry),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a.filter_totals + 11, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
#define F_PUBLISH(STEPS)                                                       \
  if (a.rotate == 1 && wave == 0) {                                            \
    const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memrealtime();          \
    const unsigned long long val_ =                                            \
        ((unsigned long long)(start_abs + (STEPS)) << 32) | now_;              \
    /* ticks per k-step x 16, once the tile has made enough of them to tell */ \
    const uint32_t t16_ = (STEPS) >= 32 ? (now_ - t_start) * 16u / (STEPS) : 0u; \
    const uint32_t zero_ = 0;                                                  \
    unsigned long long save_;                                                  \
    asm volatile("s_mov_b64 %0, exec\n\t"                                      \
                 "s_mov_b64 exec, 1\n\t"                                       \
                 "global_store_dwordx2 %1, %2, %3\n\t"                         \
                 "s_mov_b64 exec, %0"                                          \
                 : "=&s"(save_)                                                \
                 : "v"(zero_), "v"(val_), "s"(pos_word)                        \
                 : "memory");                                                  \
    if (t16_ != 0)                                                             \
      asm volatile("s_mov_b64 %0, exec\n\t"                                    \
                   "s_mov_b64 exec, 1\n\t"                                     \
                   "global_store_dword %1, %2, %3\n\t"                         \
                   "s_mov_b64 exec, %0"                                        \
                   : "=&s"(save_)                                              \
                   : "v"(zero_), "v"(t16_), "s"(ticks_word)                    \
                   : "memory");                                                \
  }

  uint32_t seg_first = 0;  // k-steps of the piece behind us
  // the tile left at a check point (uniform): at the forecast, or at the rigorous check --
  // with nothing alive, or with a few live pairs that the epilogue below (run on the prefix
  // counts) hands to the candidate list
  bool left = false, left_forecast = false, left_emit = false;
#pragma nounroll
  while (true) {
    uint32_t seg_end = num_steps;
    if (chk1 > seg_first) seg_end = chk1;
    if (chk0 > seg_first) seg_end = chk0;
    seg_end = __builtin_amdgcn_readfirstlane(seg_end);
    seg_steps = seg_end - seg_first;
    // (the segment's first k-step of the bitset: behind the end of the sites it counts from
    //  0 again; a segment that holds the end goes around it without a pause)
    const uint32_t seg_k = seg_first >= wrap && wrap != 0 ? seg_first - wrap : k0 + seg_first;
    seg_src = g_wave + (uint64_t)seg_k * kstep_bytes;
    seg_wrap = __builtin_amdgcn_readfirstlane(
        wrap > seg_first && wrap < seg_end ? wrap - seg_first : 0xFFFFFFFFu);
    asm volatile("" : "+s"(seg_src), "+s"(seg_steps), "+s"(seg_wrap));

    // Stages 0 .. 3 of the segment requested, stage 0 landed.
#pragma unroll
    for (int st = 0; st < kStages - 1; ++st) {
      const Addr p0 = addr_of(st, st);
      F_ISSUE4(p0, 0)
      if (kUnits == 2) F_ISSUE4(p0, 1)
    }
    if (CUKING_FILTER_ABLATE != 1 && CUKING_FILTER_ABLATE != 2)
      __builtin_amdgcn_s_waitcnt(vmcnt_imm((kStages - 2) * kStageReqs));
    __syncthreads();
    // unit 0 of stage 0, set B
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      F_READ(k, 0, 0)
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      F_BUILD_B(0, k)
    }
    uint32_t buf = 0;  // buffer of the k-step being multiplied
    Addr pa = addr_of(kStages - 1, kStages - 1);
    uint32_t step = 0;
#if CUKING_FILTER_FINE
    while (step + 3 < seg_steps) {
      F_KSTEP
      F_KSTEP
      F_KSTEP
      F_KSTEP
    }
    while (step < seg_steps) F_KSTEP
#else
    while (step + 1 < seg_steps) {
      F_KSTEP
      F_KSTEP
    }
    if (step < seg_steps) F_KSTEP
#endif
    // The clamped repeats of the last stage must have landed before the stages
    // become the check's or the epilogue's scratch.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    F_PUBLISH(seg_end)
    if (seg_end == num_steps) break;

    // --- the check behind k-step seg_end of the tile
    const bool forecast = seg_end == chk0;  // uniform
    const float scale = forecast ? (float)kCheckShares64[0] * (1.f / 64.f) : 1.f;
    float2 *const ck_rows = reinterpret_cast<float2 *>(lds);  // (u over the prefix, scaled bound)
    float2 *const ck_cols = ck_rows + kT;
    uint32_t *const ck_words = reinterpret_cast<uint32_t *>(ck_cols + kT);  // one per wavefront
    {
      const uint32_t share = forecast ? kCheckShares64[0] : share1;  // (uniform)
      // (the forecast counts the pairs that WILL be candidates from an eighth of the sites:
      //  the bound of an unrelated pair scatters sqrt(8) times as widely there as at the
      //  end, 1 / sqrt(sites).  A quadrant goes dense from 2.3 % candidates on -- pairs two
      //  standard deviations out --, so the prefix count matches the final one at that
      //  point when the prefix is tested against a threshold 2 (sqrt(8) - 1) standard
      //  deviations higher; tested against the threshold itself it called cohorts dense that
      //  the list handles at a third of the cost: 7 % missing calls at the default threshold,
      //  profiles/r04_missing_curve.txt)
      const float thr_f = forecast ? a.kin_threshold + 3.66f * rsqrtf(256.f * (float)all_steps)
                                   : a.kin_threshold;
      const float t = 2.f - 4.f * thr_f;
      const size_t ir = (size_t)tr * kT + threadIdx.x;
      const size_t ic = (size_t)a.geo.col_base + (size_t)tc * kT + threadIdx.x;
      const float2 sr = a.sample_stats[ir], sc = a.sample_stats[ic];
      ck_rows[threadIdx.x] = make_float2(prefix_u_of(share, ir, sr.x), scale * fmaf(t, sr.y, 8.f));
      ck_cols[threadIdx.x] = make_float2(prefix_u_of(share, ic, sc.x), scale * fmaf(t, sc.y, 8.f));
    }
    __syncthreads();
    const uint32_t emit_cap = (a.check1 >> 8) & 0xFFu;  // (uniform; king_common.h check1)
    uint32_t cnt = 0;  // pairs of this lane still under the bound
    {
      float2 cc[4];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) cc[bj] = ck_cols[wx * 128 + bj * 32 + lr];
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 cr = ck_rows[wy * 128 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];
          // (every element of the tile counts, the ones outside the block too: a tile on the
          //  diagonal holds each sample against itself and stays -- 0.5 % of the tiles at
          //  configs[2]; the test on the indices made the kernel a quarter longer)
#pragma unroll
          for (int bj = 0; bj < 4; ++bj)
            cnt += fmaf(-0.5f, acc[bi][bj][r], cr.x + cc[bj].x) < fminf(cr.y, cc[bj].y) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    // (forecast: twice the cap -- the pairs of a quadrant share their samples, so the count
    //  of a quadrant scatters more widely than independent pairs would; at 7 % missing calls
    //  and the default threshold a sixth of the tiles left with the cap itself, for a launch
    //  the candidate list handles in half the time)
    // (rigorous check: the word carries the quadrant's live pairs, capped -- a tile with
    //  only a FEW of them, the relatives it holds, hands exactly those to the candidate list
    //  and leaves as well: a cohort with some relatedness in every tile keeps its early exits)
    if (lane == 0)
      ck_words[wave] = forecast ? (cnt > 2 * a.quadrant_cap ? 1u : 0u)
                                : (cnt > emit_cap ? emit_cap + 1 : cnt);
    __syncthreads();
    uint32_t found = forecast ? ck_words[0] + ck_words[1] + ck_words[2] + ck_words[3]
                              : max(max(ck_words[0], ck_words[1]), max(ck_words[2], ck_words[3]));
    found = __builtin_amdgcn_readfirstlane(found);
    __syncthreads();  // (the words are stage memory again from here on)
    if (forecast ? found >= 3 : found <= emit_cap) {
      // The tile leaves (uniform; nothing is in flight; the book-keeping follows behind
      // the loop: a divergent branch on the way out makes the compiler treat the whole
      // loop as divergent, request addresses and all).
      left = true;
      left_forecast = forecast;
      left_emit = !forecast && found != 0;
      break;
    }
    seg_first = seg_end;
  }
#undef F_KSTEP
#undef F_SLICE_A
#undef F_SLICE_B
#undef F_BAR
#undef F_MMA
#undef F_BUILD_A
#undef F_BUILD_B
#undef F_PIN4
#undef F_PINF
#undef F_READ
#undef F_ISSUE4
#undef F_PUBLISH
  if (left && left_emit && threadIdx.x == 0)
    __hip_atomic_fetch_add(a.filter_totals + kTotalEarly, 1ull, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (phase != 0 && threadIdx.x == 0)
    __hip_atomic_fetch_add(a.filter_totals + kTotalRotated, 1ull, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (left && !left_emit) {
    // Forecast: the tile leaves for the exact kernel -- its quadrants count as handed
    // over, the fallback launch is needed.  Rigorous check: for good.
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(a.filter_ctrl + kCtrlFinished, 4u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      if (left_forecast) {
        __hip_atomic_fetch_add(a.filter_ctrl + kCtrlLeft, 4u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.filter_ctrl + kCtrlGate, 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(
            a.filter_totals + kTotalDense, 4ull,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a.tile_done[bid] = 1;
        __hip_atomic_fetch_add(
            a.filter_totals + kTotalEarly, 1ull,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if CUKING_FILTER_TIMING
        if (cu_last_exit != nullptr)
          __hip_atomic_store(cu_last_exit + cu_key, (uint32_t)__builtin_amdgcn_s_memrealtime() | 1u,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      }
    }
    return 0;
  }

  // This tile runs to its end here: the fallback launch has nothing to do for it.
  // (The tiles of remainder pieces are marked by the host.)
  if (!split && a.tile_done != nullptr && threadIdx.x == 0) a.tile_done[bid] = 1;

  if (split) {
    // Park this piece (16-byte write-through stores, [4 registers][thread]), take
    // a ticket of the tile; the last piece in adds the others to its own.
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr uint32_t kSlabU4 = 64 * 256;  // float4 per slab
    {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          a.fsplit_slabs + (size_t)piece * kSlabU4, 0, (int)(kSlabU4 * 16), 0x00020000);
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            v4u v;
            v[0] = __float_as_uint(acc[bi][bj][4 * r4]);
            v[1] = __float_as_uint(acc[bi][bj][4 * r4 + 1]);
            v[2] = __float_as_uint(acc[bi][bj][4 * r4 + 2]);
            v[3] = __float_as_uint(acc[bi][bj][4 * r4 + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(
                v, rsrc, (int)((((bi * 4 + bj) * 4 + r4) * 256 + threadIdx.x) * 16), 0,
                16 /* sc1 */);
          }
    }
    // every wavefront's stores are done (and written through), then the ticket
    // (cdna_hip_programming.md, Guideline 16: sc1 payload + counter)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t *flag = reinterpret_cast<uint32_t *>(lds);  // the stages are idle now
    if (threadIdx.x == 0) {
      uint32_t *counter = a.fsplit_tickets + piece / a.fsplit_parts;
      const uint32_t ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == a.fsplit_parts - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *counter = 0;  // ready for the next launch
      }
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();  // (the flag word becomes the epilogue's scratch)
    if (!last) return 0;
    const uint32_t first_piece = piece - part;
    for (uint32_t p = 0; p < a.fsplit_parts; ++p) {
      if (p == part) continue;
      const float4 *src = a.fsplit_slabs + (size_t)(first_piece + p) * kSlabU4 + threadIdx.x;
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = src[((bi * 4 + bj) * 4 + r4) * 256];
            acc[bi][bj][4 * r4] += v.x;
            acc[bi][bj][4 * r4 + 1] += v.y;
            acc[bi][bj][4 * r4 + 2] += v.z;
            acc[bi][bj][4 * r4 + 3] += v.w;
          }
    }
  }

  if (CUKING_FILTER_ABLATE != 0) {
    float sum = 0.f;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[bi][bj][r];
    if (sum == -1.f) a.results[0].kin = sum;  // never true, keeps the sums alive
    return 0;
  }

  // --- epilogue: the bound, per pair.  C layout of the 32 x 32 MFMA: column =
  // lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  float2 *const st_rows = reinterpret_cast<float2 *>(lds);  // (u, t |H| + margin) per row
  float2 *const st_cols = st_rows + kT;
  {
    const float t = 2.f - 4.f * a.kin_threshold;
    const size_t ir = (size_t)tr * kT + threadIdx.x;
    const size_t ic = (size_t)a.geo.col_base + (size_t)tc * kT + threadIdx.x;
    float2 r = a.sample_stats[ir];
    float2 c = a.sample_stats[ic];
    if (left_emit) {
      // (the tile left at the rigorous check with a few live pairs: the same test on the
      //  sums and the per-sample counts of the sites so far -- every record is among the
      //  pairs it admits, king_filter.hip "Check points")
      r.x = prefix_u_of(share1, ir, r.x);
      c.x = prefix_u_of(share1, ic, c.x);
    }
    r.y = fmaf(t, r.y, 8.f);
    c.y = fmaf(t, c.y, 8.f);
    st_rows[threadIdx.x] = r;
    st_cols[threadIdx.x] = c;
  }
  __syncthreads();
  float2 sc[4];
#pragma unroll
  for (int bj = 0; bj < 4; ++bj) sc[bj] = st_cols[wx * 128 + bj * 32 + lr];

  // Nearly every wavefront has no candidate at all: a first sweep with the bare test
  // (5 VALU per pair, the largest `bound - x_lb` of the lane: positive <=> the test
  // below holds for some pair, the sign of a float difference is exact; pairs outside
  // the block may raise a false alarm, which only costs the sweeps below), then out.
  {
    float best = -1.f;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 sr = st_rows[wy * 128 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
          best = fmaxf(best, fminf(sr.y, sc[bj].y) -
                                 fmaf(-0.5f, acc[bi][bj][r], sr.x + sc[bj].x));
      }
    if (__ballot(best > 0.f) == 0) {  // wave-uniform
      if (lane == 0)
        __hip_atomic_fetch_add(a.filter_ctrl + kCtrlFinished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
  }
  if (lane == 0)
    __hip_atomic_fetch_add(a.filter_ctrl + kCtrlFinished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  uint32_t total = 0, base = 0, run = 0;  // wave-uniform
#pragma nounroll
  for (int pass = 0; pass < 2; ++pass) {  // 0: count the candidates, 1: append them
    // (opaque per pass: otherwise the compiler computes the 256 pairs' indices and
    //  validity once in front of the loop and parks them in scratch memory)
    uint32_t tr_p = tr, tc_p = tc;
    asm volatile("" : "+s"(tr_p), "+s"(tc_p));
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = wy * 128 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const float2 sr = st_rows[row];
        const uint32_t li = tr_p * kT + row;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
          const uint32_t lj = tc_p * kT + wx * 128 + bj * 32 + lr;
          // cuking.cu:199 plus the tile padding
          const bool valid = li < a.geo.num_rows && lj < a.geo.num_cols &&
                             a.i_begin + li < a.j_begin + lj;
          // u_i + u_j - 2 q  <  t min(|H_i|, |H_j|) + margin   (acc = 4 q)
          const float x_lb = fmaf(-0.5f, acc[bi][bj][r], sr.x + sc[bj].x);
          const bool cand = valid && x_lb < fminf(sr.y, sc[bj].y);
          const unsigned long long b = __ballot(cand);
          if (b != 0) {  // wave-uniform
            if (pass == 0) {
              total += (uint32_t)__popcll(b);
            } else {
              const uint32_t before = __builtin_amdgcn_mbcnt_hi(
                  (uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
              if (cand) a.cand_list[base + run + before] = make_uint2(li, lj);
              run += (uint32_t)__popcll(b);
            }
          }
        }
      }
    }
    if (pass == 0) {
      if (total == 0) break;  // wave-uniform: the usual case
      bool dense = total > a.quadrant_cap;
      if (!dense) {
        uint32_t got = 0;
        if (lane == 0) {
          got = __hip_atomic_fetch_add(a.filter_ctrl, total, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
          // (running total since the scratch was allocated: "filter_candidates")
          __hip_atomic_fetch_add(a.filter_totals + kTotalCand,
                                 (unsigned long long)total, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        }
        base = (uint32_t)__builtin_amdgcn_readfirstlane(got);
        if (base >= a.cand_cap || total > a.cand_cap - base) {
          // list full: the slots taken (if any) must not be read as pairs
          for (uint32_t k = base + lane; k < a.cand_cap; k += 64)
            a.cand_list[k] = make_uint2(kNoPair, kNoPair);
          dense = true;
        }
      }
      if (dense) {
        if (lane == 0) {
          const uint32_t slot = __hip_atomic_fetch_add(a.filter_ctrl + kCtrlDense, 1u, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
          if (slot < a.dense_cap) a.dense_list[slot] = make_uint2(2 * tr + wy, 2 * tc + wx);
          __hip_atomic_fetch_add(a.filter_totals + kTotalDense, 1ull,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
    }
  }
  return 0;
}

__global__ __launch_bounds__(256, 1) void king_filter_kernel(const TiledArgs a) {
  extern __shared__ uint4 lds[];
  (void)filter_tile(a, blockIdx.x, lds);
}

// The same launch from ONE workgroup per CU that takes tile after tile (king_common.h
// persist_wgs; option "filter_persistent", OFF by default).  A CU waits 32 us of a 430 us tile
// for the dispatcher between two workgroups of the one-tile-per-workgroup grid
// (tools/tile_gaps.sh: exit at the check point -> entry of the next workgroup on the same
// CU; 160 KiB of LDS and 512 registers per workgroup) -- and yet this kernel, which does not
// wait, runs configs[2] in 135.9-136.0 ms against 134.3-134.4 (same box, interleaved): the chip
// is power-bound under the matrix pipes, and what an idle CU does not draw the others clock
// on.  Kept as a measured alternative and for the tests.  The workgroups of XCD x (workgroups go to the XCDs round-robin by their index)
// take the indices x, 8 + x, 16 + x, ... of the one-tile-per-workgroup grid through a ticket
// counter of the XCD -- the same tiles in the same order as the dispatcher would hand them
// out --, then the dynamic tail's tiles through its counter like everybody else.
__global__ __launch_bounds__(256, 1) void king_filter_persistent_kernel(const TiledArgs a) {
  extern __shared__ uint4 lds[];
  const uint32_t x = blockIdx.x & 7;
  bool tail = false;  // (uniform)
  for (;;) {
    uint32_t wg = a.launch_tiles + x;  // (an index of the dynamic tail)
    if (!tail) {
      uint32_t *slot = reinterpret_cast<uint32_t *>(lds);
      if (threadIdx.x == 0)
        *slot = __hip_atomic_fetch_add(a.filter_ctrl + kCtrlTickets + x, 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const uint32_t j = __builtin_amdgcn_readfirstlane(*slot);
      __syncthreads();  // the word is stage memory from here on
      if (j * 8 + x < a.persist_wgs) {
        wg = j * 8 + x;
      } else {
        if (a.dyn_tiles == 0) break;
        tail = true;
      }
    }
    if (filter_tile(a, wg, lds) != 0) break;
    __syncthreads();  // every wavefront is through with the tile's stage memory
  }
}

// One wavefront per candidate pair: the reference's six sums (cuking.cu:219-239)
// straight from the bitset, kinship, threshold, record (cuking.cu:284-313).
__global__ __launch_bounds__(256) void king_refine_kernel(const TiledArgs a) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t count = *a.filter_ctrl;
  if (count > a.cand_cap) count = a.cand_cap;
  const uint32_t n = a.words_per_sample / 2;
  const uint32_t stride = gridDim.x * 4;
  for (uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < count; p += stride) {
    const uint2 e = a.cand_list[p];
    if (e.x == kNoPair) continue;  // (uniform: a slot of a quadrant that went dense)
    // (plane indices of the pair -> the stored samples behind them, king_common.h `perm`)
    const uint32_t off_i = a.perm != nullptr ? a.perm[e.x] : e.x;
    const uint32_t off_j = a.perm != nullptr ? a.perm[a.geo.col_base + e.y]
                                             : (a.geo.diag ? e.y : a.geo.num_rows + e.y);
    const uint64_t *het_i_w = a.bits + (uint64_t)off_i * a.words_per_sample;
    const uint64_t *alt_i_w = het_i_w + n;
    const uint64_t *het_j_w = a.bits + (uint64_t)off_j * a.words_per_sample;
    const uint64_t *alt_j_w = het_j_w + n;
    uint32_t s_het_i = 0, s_het_j = 0, s_both = 0, s_opp = 0, s_conc = 0, s_shared = 0;
    // words per lane and plane requested before any is counted: the planes of an arbitrary
    // pair are cold, a trip is one memory latency (100k sites: 4 trips; 21 us for the 770
    // candidates of configs[1] with 4 words in flight)
    constexpr uint32_t kAhead = 8;
    for (uint32_t w0 = 0; w0 < n; w0 += 64 * kAhead) {
      uint64_t hi[kAhead], ai[kAhead], hj[kAhead], aj[kAhead];
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint32_t w = w0 + 64 * k + lane;
        const bool in = w < n;  // beyond the plane: missing
        hi[k] = in ? het_i_w[w] : ~0ull;
        ai[k] = in ? alt_i_w[w] : ~0ull;
        hj[k] = in ? het_j_w[w] : ~0ull;
        aj[k] = in ? alt_j_w[w] : ~0ull;
      }
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint64_t ri = ~(hi[k] | ai[k]), rj = ~(hj[k] | aj[k]);
        const uint64_t defined = ~((hi[k] & ai[k]) | (hj[k] & aj[k]));
        s_het_i += __popcll(hi[k] & defined);
        s_het_j += __popcll(hj[k] & defined);
        s_both += __popcll(hi[k] & hj[k] & defined);
        s_opp += __popcll(((ri & aj[k]) | (ai[k] & rj)) & defined);
        s_conc += __popcll(((ri & rj) | (ai[k] & aj[k])) & defined);
        s_shared += __popcll(defined);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s_het_i += __shfl_xor(s_het_i, off);
      s_het_j += __shfl_xor(s_het_j, off);
      s_both += __shfl_xor(s_both, off);
      s_opp += __shfl_xor(s_opp, off);
      s_conc += __shfl_xor(s_conc, off);
      s_shared += __shfl_xor(s_shared, off);
    }
    if (lane == 0) {
      const float kin = king_kinship(s_het_i, s_het_j, s_both, s_opp);
      if (kin > a.kin_threshold) {
        const uint32_t ibs0 = s_opp, ibs2 = s_conc + s_both;
        const uint32_t gi = a.i_begin + off_i;
        const uint32_t gj = a.geo.diag ? a.j_begin + off_j : a.j_begin + (off_j - a.geo.num_rows);
        emit_result(gi < gj ? gi : gj, gi < gj ? gj : gi, kin, ibs0, s_shared - ibs0 - ibs2, ibs2,
                    a.max_results, a.results, a.result_index, a.result_overflow);
      }
    }
  }
}

}  // namespace

static std::atomic<uint32_t> g_check_min_steps{64};
void set_filter_check_min_steps(uint32_t steps) { g_check_min_steps.store(steps); }

hipError_t launch_sample_stats(const uint64_t *d_bit_sets, uint32_t words_per_sample,
                               const PlaneGeometry &geo, uint4 *d_planes, uint32_t s_begin,
                               uint32_t s_end, hipStream_t stream) {
  if (s_end > geo.s_stride) s_end = geo.s_stride;
  if (s_begin >= s_end) return hipSuccess;
  // (in stored order, into the arrays the sample order is built from: king_sort.hip puts
  //  them into plane order)
  float2 *stats = plane_tmp_stats(d_planes, geo);
  float *prefix = plane_tmp_prefix(d_planes, geo);
  unsigned long long *sums = const_cast<unsigned long long *>(plane_cohort_sums(d_planes, geo));
  uint32_t *steps = const_cast<uint32_t *>(plane_check_steps(d_planes, geo));
  CheckWords cw;
  const uint32_t all_steps = geo.k_words / 8;  // k-steps of 256 sites
  if (all_steps > kStatsMaxSteps) return hipErrorInvalidValue;
  for (uint32_t k = 0; k < kNumCheckShares; ++k)
    cw.w[k] = check_step_of(all_steps, k, g_check_min_steps.load());
  sample_stats_kernel<<<dim3((s_end - s_begin + 3) / 4), dim3(256), 0, stream>>>(
      d_bit_sets, words_per_sample, geo, stats, prefix, sums, steps, cw, s_begin, s_end);
  return hipGetLastError();
}

hipError_t launch_filter(const TiledArgs &args, uint64_t num_tiles, hipStream_t stream) {
  if ((uint64_t)args.geo.k_words * 32 > kMfmaN4MaxSites || args.filter_ctrl == nullptr ||
      args.cand_list == nullptr || args.dense_list == nullptr || args.sample_stats == nullptr)
    return hipErrorInvalidValue;
  static DeviceOnce attr_set;  // per device, see king_device.h
  if (!attr_set.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(king_filter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kFilterLdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(king_filter_persistent_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFilterLdsBytes);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  uint64_t cap = max_blocks_per_launch(256);
  if (cap > kFilterChunkTiles) cap = kFilterChunkTiles;
  // (a chunk's quadrants must fit the dense list)
  if (cap > args.dense_cap / 4) cap = args.dense_cap / 4;
  if (cap == 0) return hipErrorInvalidValue;
  const uint32_t wgs = args.split_wgs != 0 ? args.split_wgs : 256;  // one per CU
  uint64_t done = 0;
  while (done < num_tiles) {
    const uint64_t n = num_tiles - done < cap ? num_tiles - done : cap;
    // (the chunk's control words and, directly behind them, its tile flags: one memset)
    const bool checks = args.tile_done != nullptr;
    hipError_t e = hipMemsetAsync(args.filter_ctrl, 0, kCtrlChunkBytes + (checks ? n : 0), stream);
    if (e != hipSuccess) return e;
#if CUKING_FILTER_TIMING
    if (args.fsplit_slabs != nullptr) {  // (the timing build's table of last exits per CU)
      e = hipMemsetAsync(args.fsplit_slabs, 0, 4096 * sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
    }
#endif
    TiledArgs a = args;
    a.tile_begin = args.tile_begin + done;
    a.quad = 0;
    a.tile_list = nullptr;
    a.gate = nullptr;
    a.skip_tiles = nullptr;
    // check 0 (the forecast): for launches of fewer than 16 rounds, where the tiles that
    // would have to give up make up most of the launch before anybody has finished
    a.check0 = !checks ? 0u : args.check0 == 2 ? 2u : (args.check0 == 1 && n < 16ull * wgs) ? 1u : 0u;
    a.check1 = checks ? args.check1 : 0u;
    // (rotated tiles: for launches of many rounds -- the tiles of a few rounds have not
    //  drifted apart yet, configs[1] has an L2 hit rate of 0.76 without)
    a.rotate = checks && (n >= args.rotate_min_tiles || args.rotate >= 2) ? args.rotate : 0u;
    // Short launches: the tiles beyond whole rounds of one per CU would leave most
    // CUs idle for a whole tile time; each of them is cut into `parts` pieces of k
    // instead (same launch, behind the whole tiles).
    uint32_t rest = 0, parts = 0;
    if (args.fsplit_slabs != nullptr && args.fsplit_tickets != nullptr && n < 16ull * wgs) {
      rest = (uint32_t)(n % wgs);
      if (rest != 0 && 2 * rest <= wgs && rest <= kFilterSplitSlabs / 2) {
        parts = wgs / rest;
        if (parts > 8) parts = 8;
        if (parts * rest > kFilterSplitSlabs) parts = kFilterSplitSlabs / rest;
        // (pieces of at least 8 k-steps -- on entry args.fsplit_first, a test hook --:
        //  the pipeline's fill and the slab are per piece)
        const uint32_t min_steps = args.fsplit_first != 0 ? args.fsplit_first : 8;
        while (parts > 1 && args.geo.k_words / 8 / parts < min_steps) --parts;  // (k-steps of 256 sites)
      }
      if (parts < 2) rest = parts = 0;
    }
    const uint64_t n_whole = n - rest;  // tiles that go out whole
    uint64_t grid = n_whole;
    a.dyn_tiles = a.dyn_wgs = 0;
    if (args.xcd_chunk == 2 && args.dyn_tiles != 0 && n_whole >= args.dyn_tiles / 4 &&
        n_whole >= 288) {
      // whole rounds of patches, then the last ~6 % through the counter with half
      // as many workgroups again as tiles (launch_shape in king_mfma.hip; the
      // threshold is the context's, in 128-sample tiles there)
      const uint64_t fixed = (n_whole - n_whole / 16) / 256 * 256;
      a.launch_tiles = (uint32_t)fixed;
      a.xcd_chunk = 1;
      a.dyn_tiles = (uint32_t)(n_whole - fixed);
      a.dyn_wgs = a.dyn_tiles + a.dyn_tiles / 2;
      grid = fixed + a.dyn_wgs;
    } else if (args.xcd_chunk == 2 && n_whole >= 64) {  // patches of 32, dealt round-robin to the XCDs
      a.launch_tiles = (uint32_t)n_whole;
      a.xcd_chunk = 1;
      grid = 8ull * 32 * ((((n_whole + 31) / 32) + 7) / 8);
    } else {
      a.xcd_chunk = 0;
      a.launch_tiles = (uint32_t)n_whole;
    }
    // ... and behind them the pieces of the remainder
    a.fsplit_parts = parts;
    a.fsplit_tile0 = (uint32_t)n_whole;
    a.fsplit_first = (uint32_t)grid;
    grid += (uint64_t)rest * parts;
    // (the tiles of the remainder pieces never leave early: done as far as the fallback
    //  launch is concerned)
    if (checks && rest != 0) {
      e = hipMemsetAsync(args.tile_done + n_whole, 1, rest, stream);
      if (e != hipSuccess) return e;
    }
    // On request ("filter_persistent"), launches of many rounds without remainder pieces:
    // one resident workgroup per CU takes the grid's indices in turn.
    a.persist_wgs = 0;
    if (args.persist_wgs != 0 && rest == 0 && n >= args.persist_min_tiles) {
      a.persist_wgs = (uint32_t)(a.dyn_tiles != 0 ? a.launch_tiles : grid);
      king_filter_persistent_kernel<<<dim3(wgs), dim3(256), kFilterLdsBytes, stream>>>(a);
    } else {
      king_filter_kernel<<<dim3((uint32_t)grid), dim3(256), kFilterLdsBytes, stream>>>(a);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    king_refine_kernel<<<dim3(wgs * 4), dim3(256), 0, stream>>>(a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.codes_ready != nullptr) {
      // Lazy codes (king_common.h): the four-product kernel's layout for the whole block,
      // if this chunk handed anything to that kernel and the codes are not there yet.
      e = launch_prepare_nibbles(true, false, a.bits, a.words_per_sample, a.geo,
                                 const_cast<uint4 *>(a.planes), a.perm, 0, 0xFFFFFFFFu,
                                 a.filter_ctrl, a.codes_ready, stream);
      if (e != hipSuccess) return e;
      e = launch_mark_codes_ready(a.filter_ctrl, a.codes_ready, stream);
      if (e != hipSuccess) return e;
    }
    TiledArgs d = a;
    d.tile_list = a.dense_list;
    d.tile_list_count = a.filter_ctrl + kCtrlDense;
    d.tile_list_cap = a.dense_cap;
    e = launch_mfma_list(d, wgs, stream);
    if (e != hipSuccess) return e;
    if (checks) {
      // The fallback: every tile of the chunk that has not set its flag (it left at
      // check 0, or never started because most quadrants of the launch had gone dense),
      // in the four-product kernel's own order -- if there is any: the gate word.
      TiledArgs f = a;
      f.quad = 1;
      f.tile_begin = a.tile_begin * 4;
      f.gate = a.filter_ctrl + kCtrlGate;
      f.skip_tiles = a.tile_done;
      f.skip_base = a.tile_begin;
      e = launch_mfma_gated(f, n * 4, wgs, stream);
      if (e != hipSuccess) return e;
    }
    done += n;
  }
  return hipSuccess;
}

}  // namespace cuking
