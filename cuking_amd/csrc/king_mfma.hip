// The KING pair kernel on the gfx950 matrix cores.
//
// popcount(x & y) over the sites of two bit planes is the dot product of the
// two 0/1 vectors, so the four sums kinship needs (cuking.cu:232-239) are five
// plane products per pair:
//     opp = A_i.R_j + R_i.A_j      bh = H_i.H_j
//     hi  = H_i.D_j                hj = D_i.H_j
// (A hom-alt, R hom-ref, H het, D defined; the full form adds
// hom_hom = (A|R)_i.(A|R)_j).  v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1)
// operands does 32 x 32 pairs x 64 sites per instruction in 32 cycles, four
// times the bf16 rate; products and sums are small integers, exact in the
// float32 accumulators while every sum stays below 2^24 (kMfmaMaxSites).
//
// Operand expansion costs ONE VALU instruction per dword.  A lane's fragment
// is 32 fp4 values = 4 dwords.  From four 32-site words of the reference's two
// planes (het, hom_var) one v_bitop3_b32 per dword computes e.g.
// ~het & hom_var & (0x11111111 << f): site 4q+f of each word lands in nibble q
// as the fp4 code 1 << f, i.e. the value 2^(f-1) (0.5, 1, 2).  Both operands
// carry the same factor, and the instruction's E8M0 block scale (2^(1-f) on
// each side) takes it out again, so every product is exactly 1.0 (f = 1 needs
// no scale and uses the unscaled instruction).  f = 3 would be the sign bit:
// those sites are shifted down to position 0 first.  The order of the sites
// inside the k dimension is irrelevant as long as both operands agree.
//
// Workgroup = 128 x 128 pairs, 4 wavefronts (one per SIMD, up to 512
// registers each), each 64 x 64 pairs = 2 x 2 MFMA blocks x 4 float32
// accumulator sets (lean form; the full form keeps 5 sets for 1 x 2 blocks and
// makes two passes over k).  One k-step = 256 sites = for every lane one uint4
// (four 32-site words) per plane and block, read from LDS with ds_read_b128
// and expanded four times (f = 0..3): 80 MFMAs per k-step and wavefront.  The
// planes come from the quad layout (king_common.h) by LDS-DMA, 16 KiB per
// k-step, kStages stages deep.  DESIGN.md 4.1 has the measurements behind the
// choices; archive/profiles/r01_mfma_microbench.txt the raw numbers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef CUKING_MFMA_TIMELINE
#include <algorithm>
#include <vector>
#endif

#include "king_common.h"
#include "king_device.h"

// The LDS-DMA statements below write M0 and say so in their clobber lists; the
// compiler notes that it keeps no value of its own there (M0 is reserved).
#pragma clang diagnostic ignored "-Winline-asm"

#ifndef CUKING_MFMA_STAGES
#define CUKING_MFMA_STAGES 6
#endif
#ifndef CUKING_MFMA_PREFILTER
#define CUKING_MFMA_PREFILTER 1  // 0: exact kinship for every pair (A/B experiments)
#endif

namespace cuking {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;
constexpr int kStageU4 = 2 * 2 * 2 * kTile;  // sides x k-groups x planes x samples
constexpr int kPiecesPerWave = 4;            // 16 x 1 KiB per stage, 4 wavefronts
// LDS stages (16 KiB each): stage s + kStages - 1 is requested while stage s
// is multiplied, i.e. kStages - 2 k-steps (~1.4 us each) of HBM latency are
// covered.
constexpr int kStages = CUKING_MFMA_STAGES;
// Lean form (no LDS needed for a parked sum): more stages, at least the two it
// takes to hand stages over with one barrier per TWO k-steps.
#ifndef CUKING_MFMA_PAIRED
#define CUKING_MFMA_PAIRED 1
#endif
constexpr bool kPairedSync = CUKING_MFMA_PAIRED != 0 && kStages == 6;
#ifndef CUKING_MFMA_PAIRED_STAGES
// even, 8 or 10.  10 x 16 KiB is the CU's whole LDS and puts 5.5 instead of 3.5
// k-steps between a request and the hand-over that needs it: configs[2] 593 ->
// 590 ms, 40k x 100k 95.2 -> 94.7 ms, configs[1] (bitset in the Infinity Cache) equal
// (archive/experiments/exp25.sh).
#define CUKING_MFMA_PAIRED_STAGES 10
#endif
constexpr int kStagesPaired = CUKING_MFMA_PAIRED_STAGES;
static_assert(kStagesPaired % 2 == 0 && kStagesPaired >= 8 && kStagesPaired <= 10, "stages");
// s_waitcnt vmcnt(N) immediate for N requests that may stay in flight.
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

// v_bitop3_b32 truth tables over (het, hom_var, mask), index = 4 het + 2 hom + mask.
constexpr int kA = 0x08;  // hom-alt:  ~het &  hom & mask
constexpr int kR = 0x02;  // hom-ref:  ~het & ~hom & mask
constexpr int kH = 0x20;  // het:       het & ~hom & mask
constexpr int kD = 0x2A;  // defined:  ~(het & hom) & mask
constexpr int kY = 0x0A;  // A | R:    ~het & mask

// One operand fragment: plane KIND of four 32-site words at nibble position
// given by `mask` (an SGPR: a 32-bit literal would cost half a cycle more per
// instruction, tools/micro/mfma_fill).
template <int KIND>
__device__ __forceinline__ v8i frag(const uint4 het, const uint4 hom, uint32_t mask) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)__builtin_amdgcn_bitop3_b32(het.x, hom.x, mask, KIND);
  r[1] = (int)__builtin_amdgcn_bitop3_b32(het.y, hom.y, mask, KIND);
  r[2] = (int)__builtin_amdgcn_bitop3_b32(het.z, hom.z, mask, KIND);
  r[3] = (int)__builtin_amdgcn_bitop3_b32(het.w, hom.w, mask, KIND);
  return r;
}

// Nibble layout: a fragment is the stored dwords masked to one kind's bits.
__device__ __forceinline__ v8i nfrag(const uint4 w, uint32_t mask) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)(w.x & mask);
  r[1] = (int)(w.y & mask);
  r[2] = (int)(w.z & mask);
  r[3] = (int)(w.w & mask);
  return r;
}

__device__ __forceinline__ uint4 shr3(const uint4 w) {
  return make_uint4(w.x >> 3, w.y >> 3, w.z >> 3, w.w >> 3);
}

// acc += sum over the 64 sites of the fragment of a_site * b_site.  The E8M0
// scale 2^(1-F) on each side (F == 3 sits at position 0 again) undoes the
// 2^(F-1) of the expansion.  F == 1 needs none: scale operands 0 select the
// unscaled instruction, which holds the issue port 5 cycles less.
template <int F>
__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  constexpr int scale = F == 0 ? 128 : F == 1 ? 0 : F == 2 ? 126 : 128;
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
      a, b, c, 4 /* A is fp4 */, 4 /* B is fp4 */, 0, scale, 0, scale);
}

// The exact lean epilogue of one register's pairs, out of line: it is reached
// for the few pairs that may pass the threshold, and inlined 64 times (with
// its recount loop and record append) it made the epilogue ~90 KB of code,
// more than the instruction cache.
__device__ __attribute__((noinline)) void lean_epilogue_call(
    const EmitCtx c, bool valid, uint32_t li, uint32_t lj, uint32_t het_i,
    uint32_t het_j, uint32_t both_het, uint32_t opp, uint32_t lane) {
  lean_epilogue_pair(c, valid, li, lj, het_i, het_j, both_het, opp, lane);
}

__device__ __attribute__((noinline)) void lean_epilogue_call_n4(
    const EmitCtxP c, bool valid, uint32_t li, uint32_t lj, uint32_t het_i,
    uint32_t het_j, uint32_t dd, int32_t q, uint32_t lane) {
  lean_epilogue_pair_n4(c, valid, li, lj, het_i, het_j, dd, q, lane);
}

// Full form: a wavefront reserves the slots for ALL records of its 64 x 64 pairs
// with one atomic (one lane), out of line.  History: the per-pair append
// (atomicAdd in every lane, which the compiler turns into a wave-level
// aggregation with whole-wave-mode temporaries) inlined 64 times among ~500 live
// registers made the remainder-split instantiation return wrong sums for whole
// tiles whenever only a few lanes emitted (tools/fuzz_split.py seed 1 case 3);
// out of line per pair it was right but cost one atomic round trip per call --
// 11 ms per 10^6 records at configs[1] when most pairs pass.
// ... and the record itself (cuking.cu:297-313), out of line as well: the kernel
// around it has no registers to spare for 64 inlined copies.
template <class Ctx>
__device__ __attribute__((noinline)) void full_store_call(
    const Ctx c, uint32_t slot, uint32_t li, uint32_t lj, uint32_t het_i, uint32_t het_j,
    uint32_t both_het, uint32_t opp, uint32_t hom_hom) {
  if (slot >= c.max_results) {
    atomicMax(c.result_overflow, 1u);
    return;
  }
  const uint32_t ibs2 = hom_hom - opp + both_het;
  const uint32_t shared = het_i + het_j - both_het + hom_hom;
  cuking_result rec;
  record_pair(c, stored_row(c, li), stored_col(c, lj), &rec.sample_i, &rec.sample_j);
  rec.kin = king_kinship(het_i, het_j, both_het, opp);
  rec.ibs0 = opp;
  rec.ibs1 = shared - opp - ibs2;
  rec.ibs2 = ibs2;
  c.results[slot] = rec;
}

__device__ __attribute__((noinline)) uint32_t reserve_slots(uint32_t *result_index, uint32_t n) {
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0)
    base = __hip_atomic_fetch_add(result_index, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (uint32_t)__builtin_amdgcn_readfirstlane(base);
}

// `n` MFMAs, each followed by `v` VALU instructions (scheduling request).
#define CUKING_PACE(n, v)                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
    if ((v) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);          \
  }

// Diagnostic build (-DCUKING_MFMA_TIMELINE, never shipped): when the workgroups
// of a launch start and finish, and where a piece's time goes (100 MHz
// s_memrealtime, comparable across the chip).  [workgroup][12]: 0 entry,
// per segment s (0, 1): 1+5s start, 2+5s loop entered, 3+5s loop done,
// 4+5s slab / totals done, 5+5s epilogue done, 11 exit.
#ifdef CUKING_MFMA_TIMELINE
__device__ unsigned long long *g_timeline;
constexpr uint32_t kTimelineBlocks = 1u << 19;
void timeline_arm(uint32_t whole, uint32_t blocks);
#define CUKING_TL(K)                                                           \
  if (threadIdx.x == 0 && g_timeline != nullptr && blockIdx.x < kTimelineBlocks) { \
    unsigned long long t_;                                                     \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    g_timeline[(size_t)blockIdx.x * 12 + (K)] = t_;                            \
  }
#else
#define CUKING_TL(K)
#endif

// Diagnostic build (-DCUKING_MFMA_STAMPS, never shipped): where a wavefront's
// time goes inside a k-step.  One s_memtime per phase boundary (ONE asm statement
// with its own lgkmcnt(0), MI355X guide, In-kernel stamps), summed per phase and
// written by wave 0 of the first 1024 workgroups into the (otherwise unused)
// split scratch; cuking_timing_collect prints the averages.  The stamps cost
// ~40 cycles each.
#ifdef CUKING_MFMA_STAMPS
#define CUKING_STAMP(K)                                                        \
  {                                                                            \
    unsigned long long now_;                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    stamp_sum[(K) + stamp_row] += now_ - stamp_last;                           \
    stamp_last = now_;                                                         \
  }
#define CUKING_STAMP_ROW(SYNC) stamp_row = (SYNC) ? 6 : 0;
#define CUKING_STAMP_SLICE(C) stamp_row = 4 * (C);
#else
#define CUKING_STAMP(K)
#define CUKING_STAMP_ROW(SYNC)
#define CUKING_STAMP_SLICE(C)
#endif

// Tickets: one per workgroup and pass (the full form makes two).
__host__ __device__ inline size_t split_counter_bytes(uint32_t wgs) {
  return ((size_t)wgs * 2 * sizeof(uint32_t) + 255) / 256 * 256 + 256;
}
// ... and, in the last 256 bytes, the tile counter of a launch's dynamic tail.
__host__ __device__ inline size_t dyn_counter_index(uint32_t wgs) {
  return (split_counter_bytes(wgs) - 256) / sizeof(uint32_t);
}

// First work unit of split workgroup w: floor(w * units / wgs).
__device__ __forceinline__ uint64_t split_bound(uint64_t w, uint64_t units,
                                                uint32_t wgs) {
  return w * units / wgs;
}
// The split workgroup that owns unit u (needs units >= wgs).
__device__ __forceinline__ uint32_t split_owner(uint64_t u, uint64_t units,
                                                uint32_t wgs) {
  const uint32_t w = (uint32_t)(u * wgs / units);
  return split_bound(w + 1, units, wgs) <= u ? w + 1 : w;
}

// SPLIT = false: workgroup = one tile, all k-steps.
// SPLIT = true ("stream-k" remainder): the launch's tiles x k-steps are one
// line of work units cut into equal pieces, one per workgroup, so a remainder
// of tiles that would leave most CUs idle for a whole tile time still fills
// the chip.  A piece covers the end of one tile and/or the start of the next;
// each partial result (exact integers) is parked in a scratch slab, and the
// workgroup that delivers a tile's last part adds the others to its own and
// runs the epilogue.
// ABLATE (tuning builds, wrong results): 1 = no LDS-DMA, 2 = no barrier either,
// 3 = epilogue reduced to one store per lane (prices the kinship/threshold pass).
// N4 = the four-product form on the nibble layout (below, "Four products").
template <bool FULL, bool SPLIT, int ABLATE = 0, bool N4 = false>
__global__ __launch_bounds__(256, 1) void king_mfma_kernel(const TiledArgs a) {
  // Five-product form: the lean form has 10 LDS stages and ONE stage barrier per
  // two k-steps (below); the full form parks its fifth sum in the LDS behind the
  // stages and keeps 6 stages with a barrier per k-step.  Four-product form:
  // 5 stages of 32 KiB, every sum in registers.
  constexpr bool PAIRED = !N4 && !FULL && kPairedSync;
  constexpr int NSTAGE = N4 ? kMfmaN4Stages : PAIRED ? kStagesPaired : kStages;
  constexpr int NSUM = FULL ? 5 : 4; // sums per pair
  // ... of which the main loop keeps NQ = 4 in its accumulators (five products:
  // opp, bh, hi, hj; four products: hi / 2, hj / 2, dd, 4 q).  The full form's
  // fifth sum, hom_hom, comes from a pass of its own in front of the main loop and
  // waits for the epilogue PARKED in LDS (five products) or in 64 registers that
  // the main loop does not touch (four products: `hh5`).
  constexpr int NQ = 4;
  constexpr bool PARKED = FULL && !N4;
  constexpr bool HH5 = FULL && N4;
  constexpr int BI = 2;              // 32-row blocks of the wavefront
  extern __shared__ uint4 lds[];  // [NSTAGE][side][k-group][plane | slice][128]

  // Which tile (SPLIT: which piece) this workgroup takes.
  uint32_t bid = blockIdx.x;
  uint32_t piece = 0;
  if (a.dyn_tiles != 0 && blockIdx.x >= a.launch_tiles) {
    // dynamic tail (king_common.h): the next tile (SPLIT: the next piece of the
    // remainder) nobody has taken yet
    // (every workgroup of the tail asks exactly once: the last one to ask leaves
    // the counter ready for the next launch)
    uint32_t *slot = reinterpret_cast<uint32_t *>(lds);
    if (threadIdx.x == 0) {
      uint32_t *counter = a.split_counters + dyn_counter_index(a.split_wgs);
      const uint32_t asked = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
      if (asked == a.dyn_wgs - 1)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *slot = asked;
    }
    __syncthreads();
    const uint32_t t = __builtin_amdgcn_readfirstlane(*slot);
    __syncthreads();  // the word is stage memory from here on
    if (t >= a.dyn_tiles) return;  // uniform
    if (SPLIT) piece = t; else bid = a.launch_tiles + t;
  } else
  // (SPLIT launches: the whole-tile workgroups in front take the patch order
  // when their count is a multiple of 8 x 32; the pieces behind them do not)
  if (a.xcd_chunk != 0 && (!SPLIT || blockIdx.x < a.split_whole)) {
    const uint32_t x = blockIdx.x & 7, j = blockIdx.x >> 3;
    // xcd_chunk == 1: patches of 32 consecutive tiles dealt round-robin to the
    // XCDs (XCD x takes patches x, x + 8, ...); otherwise one contiguous chunk
    // of xcd_chunk tiles per XCD.
#ifndef CUKING_XCD_XOR
#define CUKING_XCD_XOR 0  // (experiment: which patches an XCD takes, archive/profiles/r02_tail.txt)
#endif
    bid = a.xcd_chunk == 1 ? (((j >> 5) * 8 + (x ^ CUKING_XCD_XOR)) << 5) + (j & 31)
                           : x * a.xcd_chunk + j;
    if (bid >= a.launch_tiles) return;  // padding (uniform)
  }
  // Tile-list mode (king_common.h): entries bid, bid + grid, ... of the list.  The
  // persistent mode (gate != nullptr) walks the launch's own enumeration the same way:
  // units bid, bid + grid, ... of gate_count, or nothing at all when the gate is shut.
  const bool listed = !SPLIT && (a.tile_list != nullptr || a.gate != nullptr);
  uint32_t list_count = 0;
  if (listed) {
    if (a.gate != nullptr) {
      list_count = *a.gate != 0 ? a.gate_count : 0u;  // (uniform: a scalar load)
    } else {
      list_count = *a.tile_list_count;
      if (list_count > a.tile_list_cap) list_count = a.tile_list_cap;
    }
    // Workgroups are dealt round-robin to the 8 XCDs: give the ones that share an
    // XCD (and its L2) CONSECUTIVE entries of every round -- the list is in tile
    // order more or less, neighbours share row / column strips.
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    if (bid >= list_count) return;  // uniform
  }
  bid = __builtin_amdgcn_readfirstlane(bid);
  piece = __builtin_amdgcn_readfirstlane(piece);

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wr = (wave >> 1) * 64;  // wavefront's rows inside the tile
  const uint32_t wc = (wave & 1) * 64;   // ... and columns
  const uint32_t g = lane >> 5;          // k-group of the MFMA operand
  const uint32_t lr = lane & 31;         // row / column inside the block
  uint32_t lane16 = lane * 16;           // byte offset of the lane in a DMA row
  // This lane's operand rows inside a stage (uint4 units): rows / columns.
  uint32_t row_off = (0 * 2 + g) * 2 * kTile + wr + lr;
  uint32_t col_off = (1 * 2 + g) * 2 * kTile + wc + lr;
  const uint32_t s_stride = a.geo.s_stride;
  const uint32_t tile_steps = a.geo.k_words / 8;

  uint32_t m1, m2, m4;  // nibble masks, pinned to SGPRs
  asm volatile("s_mov_b32 %0, 0x11111111" : "=s"(m1));
  asm volatile("s_mov_b32 %0, 0x22222222" : "=s"(m2));
  asm volatile("s_mov_b32 %0, 0x44444444" : "=s"(m4));

  // Work units [unit_lo, unit_hi) of this workgroup; unit = (tile, k-step).
  // SPLIT launches: the first split_whole workgroups take one whole tile each,
  // the remaining split_wgs ones cut the units of the last split_tiles tiles
  // into equal pieces (piece index `piece`).
  CUKING_TL(0)
  [[maybe_unused]] uint32_t tl_seg = 0;  // (timeline build) segment of this workgroup
  const uint64_t units = (uint64_t)a.split_tiles * tile_steps;      // of the cut-up tiles
  const uint64_t whole_units = SPLIT ? (uint64_t)a.split_whole * tile_steps : 0;
  const bool whole_wg = !SPLIT || blockIdx.x < a.split_whole;

  uint64_t unit_lo = whole_wg ? (uint64_t)bid * tile_steps
                              : whole_units + split_bound(piece, units, a.split_wgs);
  uint64_t unit_hi = whole_wg ? unit_lo + tile_steps
                              : whole_units + split_bound(piece + 1, units, a.split_wgs);
#ifdef CUKING_TUNING
  // experiment (split_wgs with the top bit set): persistent workgroups, whole
  // tiles blockIdx.x, blockIdx.x + grid, ... of the launch's split_tiles.
  const bool strided = SPLIT && (a.split_wgs & 0x80000000u) != 0;
  uint64_t next_tile = blockIdx.x;
  if (strided) unit_lo = unit_hi = 0;
#else
  constexpr bool strided = false;
  uint64_t next_tile = 0;
#endif
  while (true) {
  if (listed && unit_lo >= unit_hi) {
    bid += gridDim.x;
    if (bid >= list_count) break;
    __syncthreads();  // every wavefront is through with the stages of the last tile
    unit_lo = (uint64_t)bid * tile_steps;
    unit_hi = unit_lo + tile_steps;
  }
  if (unit_lo >= unit_hi) {
    if (!strided || next_tile >= a.split_tiles) break;
    unit_lo = next_tile * tile_steps;
    unit_hi = unit_lo + tile_steps;
    next_tile += gridDim.x;
  }
  // Everything about the piece is wave-uniform; the 64-bit divisions behind it
  // are computed in vector registers, so pin the results to SGPRs (the SPLIT
  // instantiation otherwise runs out of VGPRs in the main loop and spills).
  const uint32_t seg_tile =
      __builtin_amdgcn_readfirstlane((uint32_t)(unit_lo / tile_steps));  // within the launch
  const uint32_t k_first = __builtin_amdgcn_readfirstlane(
      (uint32_t)(unit_lo - (uint64_t)seg_tile * tile_steps));
  const uint32_t num_steps = __builtin_amdgcn_readfirstlane(
      (unit_hi - unit_lo < (uint64_t)(tile_steps - k_first)) ? (uint32_t)(unit_hi - unit_lo)
                                                             : tile_steps - k_first);
  unit_lo += num_steps;
  if (SPLIT) {
    // (the builtin returns int: the casts keep the low word from being sign-extended)
    unit_lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(unit_lo >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)unit_lo);
    unit_hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(unit_hi >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)unit_hi);
  }
  uint32_t tr, tc;
  if (!decode_tile(a, a.tile_begin + seg_tile, &tr, &tc)) continue;  // uniform
  tr = __builtin_amdgcn_readfirstlane(tr);
  tc = __builtin_amdgcn_readfirstlane(tc);

  CUKING_TL(1 + 5 * tl_seg)
  const uint4 *g_rows = a.planes + (uint64_t)tr * kTile;
  const uint4 *g_cols = a.planes + a.geo.col_base + (uint64_t)tc * kTile;

  // LDS-DMA addressing.  Row `row` = 4 wave + r (1 KiB) of a stage is
  // (side, k-group, plane p, half seg) = (row >> 3, row >> 2 & 1, row >> 1 & 1,
  // row & 1): a wavefront's four requests share side and k-group, and r = 2 p +
  // seg, so that seg moves source and destination by the same 1 KiB -- the
  // instruction's immediate offset, which is added to both -- and only p needs
  // addresses of its own.  Per k-step that is one multiply for the wave-uniform
  // source row and a handful of scalar adds; `piece_addr` is called a phase ahead
  // of the requests, so that this arithmetic does not sit in the MFMA gaps that
  // already hold a request (it was 6-18 scalar instructions per request there).
  const uint32_t dma_side = wave >> 1, dma_kg = wave & 1;
  const uint4 *const g_wave = (dma_side ? g_cols : g_rows) +
                              ((uint64_t)(2 * k_first + dma_kg) * 2) * s_stride;
  const uint32_t l_wave = (uint32_t)(uintptr_t)(lds_void_ptr)(
      lds + ((dma_side * 2 + dma_kg) * 2) * kTile);
  struct PieceAddr { const uint4 *src[2]; uint32_t dst[2]; };
  // Source rows / LDS addresses of k-step min(step, last) -> buffer `buf`.
  // Clamping keeps the number of DMAs in flight the same in every iteration,
  // so one counted wait serves the whole loop; the repeats of the last step
  // land in a buffer nobody reads.
  auto piece_addr = [&](uint32_t step, uint32_t buf) {
    PieceAddr pa;
    if (step >= num_steps) step = num_steps - 1;
    pa.src[0] = g_wave + (uint64_t)step * 4 * s_stride;
    pa.src[1] = pa.src[0] + s_stride;
    pa.dst[0] = l_wave + buf * (kStageU4 * 16);
    pa.dst[1] = pa.dst[0] + kTile * 16;
    // materialised here, not where the requests are
    asm volatile("" : "+s"(pa.src[0]), "+s"(pa.src[1]), "+s"(pa.dst[0]), "+s"(pa.dst[1]));
    return pa;
  };
  // Request r of the four.  LDS-DMA, lane l's 16 bytes land at dst + 16 * l.
  // Inline asm keeps it out of the compiler's wait-count bookkeeping
  // (king_kernels.hip).
  auto issue_piece = [&](const PieceAddr &pa, int r) {
    if (ABLATE == 1 || ABLATE == 2) return;
    if (r & 1)
      asm volatile(
          "s_mov_b32 m0, %0\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:1024"
          :
          : "s"(pa.dst[r >> 1]), "v"(lane16), "s"(pa.src[r >> 1])
          : "memory", "m0");
    else
      asm volatile(
          "s_mov_b32 m0, %0\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2"
          :
          : "s"(pa.dst[r >> 1]), "v"(lane16), "s"(pa.src[r >> 1])
          : "memory", "m0");
  };
  auto issue_stage = [&](uint32_t step, uint32_t buf) {
    const PieceAddr pa = piece_addr(step, buf);
#pragma unroll
    for (int r = 0; r < kPiecesPerWave; ++r) issue_piece(pa, r);
  };
  // All but the kStages - 2 youngest stages requested so far have landed, for
  // this wavefront (counted wait) and, after the barrier, for all of them;
  // every wavefront is also done reading the buffer the next request
  // overwrites.
  auto stage_sync = [&]() {
    if (ABLATE == 2) return;
    // unpaired: the NSTAGE - 2 younger stages; paired: NSTAGE - 4 (see below)
    constexpr int younger = (PAIRED ? NSTAGE - 4 : NSTAGE - 2) * kPiecesPerWave;
    static_assert(younger < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(younger));
    __syncthreads();
  };

  // Full form: the fifth sum, hom_hom = (A|R)_i . (A|R)_j, in a pass of its own
  // IN FRONT of the main loop.  Five sums for 2 x 2 blocks are 320 accumulator
  // registers, more than the main loop can hold beside its fragments; but
  // "homozygous and defined" is just ~het (missing and padding have the het bit
  // set, cuking.cu:688-697), so this pass reads ONE plane, builds one fragment
  // kind per side and issues 16 MFMAs per k-step against the main loop's 80.
  // With so few MFMAs per byte the LDS-DMA requests and stage barriers of the
  // main loop would dominate (measured: 1.9 ms of 8.8 at 10k x 100k), so every
  // lane fetches its own operand words straight from the plane layout into
  // registers, four k-steps ahead (512 B contiguous per half wavefront; each
  // word is read by two wavefronts, from L2): no LDS, no barrier, the
  // wavefronts drift freely.  The 64 result registers per lane are parked in
  // the 64 KiB of LDS behind the stages until the epilogue (the workgroup owns
  // the CU's whole 160 KiB anyway), so the main loop runs exactly as in the
  // lean form.  (Round 1's full form made two passes of six products over
  // 32-row blocks in the compiler's order: 10.3 ms at 10k x 100k against 7.0 ms
  // lean.)
  v16f hh5[BI][2];  // full form: hom_hom of the wavefront's pairs
  if constexpr (FULL) {
    constexpr int D = 4;  // k-steps in flight
    v16f (&hh)[BI][2] = hh5;
#pragma unroll
    for (int bi = 0; bi < BI; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[bi][bj][r] = 0.f;
    // The het plane, one uint4 per 128 sites and sample: plane 0 of the quad
    // layout (planes interleaved: quad stride 2 rows), or the het-only copy the
    // nibble layout carries behind its codes for this pass (quad stride 1).
    constexpr uint32_t HS = N4 ? 1 : 2;
    const uint4 *const h_base = N4 ? a.planes + (uint64_t)a.geo.k_words * s_stride : a.planes;
    const uint4 *lane_rows = h_base + (uint64_t)tr * kTile + (uint64_t)g * HS * s_stride + wr + lr;
    const uint4 *lane_cols = h_base + a.geo.col_base + (uint64_t)tc * kTile +
                             (uint64_t)g * HS * s_stride + wc + lr;
    uint4 Ha[D][BI], Hb[D][2], Hs_a[BI], Hs_b[2];
    v8i Pa[BI], Pb[2], Qa[BI], Qb[2];
    // het words of k-step min(step, last) (the repeats are masked out below)
#define CUKING_HH_LOAD(U, STEP)                                                \
    {                                                                          \
      uint32_t s_ = (STEP);                                                    \
      if (s_ >= num_steps) s_ = num_steps - 1;                                 \
      const uint64_t off_ = (uint64_t)(s_ + k_first) * 2 * HS * s_stride;      \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                          \
        Ha[U][b] = lane_rows[off_ + b * 32];                                   \
        Hb[U][b] = lane_cols[off_ + b * 32];                                   \
      }                                                                        \
    }
#define CUKING_HH_FRAGS(X, SA, SB, MASK)                                       \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                            \
      X##a[b] = frag<kY>(SA[b], SA[b], MASK);                                  \
      X##b[b] = frag<kY>(SB[b], SB[b], MASK);                                  \
    }
#define CUKING_HH_MMA(F, X)                                                    \
    _Pragma("unroll") for (int bi = 0; bi < BI; ++bi)                          \
    _Pragma("unroll") for (int bj = 0; bj < 2; ++bj)                           \
      hh[bi][bj] = mma<F>(X##a[bi], X##b[bj], hh[bi][bj]);
    // One k-step in four phases; while the four MFMAs of a phase issue, the
    // VALU builds the next phase's fragments into the other register set (the
    // last phase builds position 0 of the NEXT k-step, buffer UN).  A k-step
    // beyond the end gets zero masks: its fragments are empty.
#define CUKING_HH_KSTEP(U, UN)                                                 \
    {                                                                          \
      const bool live_ = step + (U) < num_steps;                               \
      const bool next_ = step + (U) + 1 < num_steps;                           \
      const uint32_t k2_ = live_ ? m2 : 0u, k4_ = live_ ? m4 : 0u;             \
      const uint32_t k1_ = live_ ? m1 : 0u, n1_ = next_ ? m1 : 0u;             \
      CUKING_HH_FRAGS(Q, Ha[U], Hb[U], k2_)                                    \
      CUKING_HH_MMA(0, P)                                                      \
      CUKING_PACE(4, 4)                                                        \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_HH_FRAGS(P, Ha[U], Hb[U], k4_)                                    \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                          \
        Hs_a[b] = shr3(Ha[U][b]);                                              \
        Hs_b[b] = shr3(Hb[U][b]);                                              \
      }                                                                        \
      CUKING_HH_MMA(1, Q)                                                      \
      CUKING_PACE(4, 8)                                                        \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_HH_FRAGS(Q, Hs_a, Hs_b, k1_)                                      \
      CUKING_HH_MMA(2, P)                                                      \
      CUKING_PACE(4, 4)                                                        \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_HH_LOAD(U, step + D + (U))                                        \
      CUKING_HH_FRAGS(P, Ha[UN], Hb[UN], n1_)                                  \
      CUKING_HH_MMA(3, Q)                                                      \
      CUKING_PACE(4, 4)                                                        \
      __builtin_amdgcn_sched_barrier(0);                                       \
    }
    CUKING_HH_LOAD(0, 0)
    CUKING_HH_LOAD(1, 1)
    CUKING_HH_LOAD(2, 2)
    CUKING_HH_LOAD(3, 3)
    CUKING_HH_FRAGS(P, Ha[0], Hb[0], m1)
    for (uint32_t step = 0; step < num_steps; step += D) {
      CUKING_HH_KSTEP(0, 1)
      CUKING_HH_KSTEP(1, 2)
      CUKING_HH_KSTEP(2, 3)
      CUKING_HH_KSTEP(3, 0)
    }
#undef CUKING_HH_LOAD
#undef CUKING_HH_FRAGS
#undef CUKING_HH_MMA
#undef CUKING_HH_KSTEP
    if constexpr (PARKED) {
      float4 *park = reinterpret_cast<float4 *>(lds + NSTAGE * kStageU4) +
                     (size_t)wave * (4 * 4 * 64) + lane;
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            park[((bi * 2 + bj) * 4 + r4) * 64] =
                make_float4(hh[bi][bj][4 * r4], hh[bi][bj][4 * r4 + 1], hh[bi][bj][4 * r4 + 2],
                            hh[bi][bj][4 * r4 + 3]);
    }
    // (four-product form: the 64 registers stay where they are -- its main loop
    //  needs 256 accumulators + ~100, they fit beside)
    // (the prefetches beyond the end are in registers nobody reads; the
    //  compiler's own wait counts cover them before the main loop's hand-counted
    //  LDS-DMA starts: force that here)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }

  v16f acc[BI][2][NQ];
  constexpr uint32_t half_rows = 0;
  auto zero_acc = [&]() {
#pragma unroll
    for (int bi = 0; bi < BI; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[bi][bj][q][r] = 0.f;
  };
  if constexpr (!N4) zero_acc();  // (four products: behind the hom_hom pass)

  // Raw words of the k-step: [block][plane] for the row and the column side.
  uint4 A[BI][2], B[2][2];
#define CUKING_LOAD_RAW(BUF)                                                   \
  {                                                                            \
    const uint4 *l_rows_ = lds + (BUF) * kStageU4 + row_off;                   \
    const uint4 *l_cols_ = lds + (BUF) * kStageU4 + col_off;                   \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                            \
      _Pragma("unroll") for (int b = 0; b < BI; ++b)                           \
        A[b][p] = l_rows_[p * kTile + b * 32 + half_rows];                     \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                            \
        B[b][p] = l_cols_[p * kTile + b * 32];                                 \
    }                                                                          \
  }
// Fragment set X = planes A, R, H, D of (SA, SB) at the nibble position MASK.
#define CUKING_EXPAND(X, SA, SB, MASK)                                         \
  _Pragma("unroll") for (int b = 0; b < BI; ++b) {                             \
    X##a[b][0] = frag<kA>(SA[b][0], SA[b][1], MASK);                           \
    X##a[b][1] = frag<kR>(SA[b][0], SA[b][1], MASK);                           \
    X##a[b][2] = frag<kH>(SA[b][0], SA[b][1], MASK);                           \
    X##a[b][3] = frag<kD>(SA[b][0], SA[b][1], MASK);                           \
  }                                                                            \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) {                              \
    X##b[b][0] = frag<kA>(SB[b][0], SB[b][1], MASK);                           \
    X##b[b][1] = frag<kR>(SB[b][0], SB[b][1], MASK);                           \
    X##b[b][2] = frag<kH>(SB[b][0], SB[b][1], MASK);                           \
    X##b[b][3] = frag<kD>(SB[b][0], SB[b][1], MASK);                           \
  }
// One plane product (fragment index PA of the rows x PB of the columns) for
// the four block pairs.
#define CUKING_MMA1(F, X, PA, PB, Q)                                           \
  _Pragma("unroll") for (int bi = 0; bi < BI; ++bi)                            \
  _Pragma("unroll") for (int bj = 0; bj < 2; ++bj)                             \
    acc[bi][bj][Q] = mma<F>(X##a[bi][PA], X##b[bj][PB], acc[bi][bj][Q]);
// opp (first half), bh, hi, hj: 16 MFMAs; then the second half of opp.
#define CUKING_MMA16(F, X)                                                     \
  CUKING_MMA1(F, X, 0, 1, 0) CUKING_MMA1(F, X, 2, 2, 1)                        \
  CUKING_MMA1(F, X, 2, 3, 2) CUKING_MMA1(F, X, 3, 2, 3)
#define CUKING_MMA4(F, X) CUKING_MMA1(F, X, 1, 0, 0)

  // Stage hand-over.  Unpaired (full form): stage s + NSTAGE - 1 is requested
  // while stage s is multiplied, into the buffer stage s - 1 left, and every
  // k-step ends with the counted wait + barrier.  Paired (lean form; written for 8 stages):
  // the barrier comes only at the end of ODD k-steps and then covers the next
  // TWO stages.  A request may only overwrite a buffer whose last readers are
  // separated from it by a barrier; the reads of stage k happen at the end of
  // k-step k - 1, so with barriers B(j) at the end of odd j an even k-step i
  // may request stage i + 7 (buffer of stage i - 1, read at the end of i - 2,
  // B(i - 1) in between) and an odd k-step i stage i + 5 (buffer of stage
  // i - 3, read at the end of i - 4, B(i - 2) in between).  Request order is
  // then 0..5, 7, 6, 9, 8, ...: when B(s) (s odd) waits for all but the 16
  // youngest requests of the wavefront, those are stages s+4, s+3, s+6, s+5,
  // i.e. stages s+1 and s+2 -- read at the end of k-steps s and s+1 -- have
  // landed: the same vmcnt(16) as the unpaired form.  (In general, N stages:
  // even k-steps request stage i + N - 1, odd ones i + N - 3, and the wait
  // leaves the N - 4 youngest stages in flight.)
  if constexpr (!N4) {
  constexpr int kPrologueStages = PAIRED ? NSTAGE - 2 : NSTAGE - 1;
#pragma unroll
  for (int st = 0; st < kPrologueStages; ++st) issue_stage(st, st);
  stage_sync();

  {
    // Software pipeline: while the MFMAs of fragment f issue, the VALU builds
    // fragment f + 1 into the other register set (an MFMA never reads a
    // register written just before it); behind the MFMAs of f = 3 come the
    // next k-step's stage hand-over, its LDS reads and its f = 0 fragment.
    // Pacing (tools/micro/mfma_fill): a scaled fp4 MFMA holds the issue port
    // 13 cycles (unscaled: 8) of its 32, a VALU instruction 4, so 4 (5) hide
    // behind one MFMA.  The k-step's four LDS-DMA requests go one each into
    // gaps without VALU work.
    v8i Xa[2][4], Xb[2][4], Ya[2][4], Yb[2][4];
    uint4 As[2][2], Bs[2][2];
    CUKING_LOAD_RAW(0)
    CUKING_EXPAND(X, A, B, m1)
    uint32_t buf = 0;  // buffer of the k-step being multiplied
    // The loop's per-lane invariants sit in registers from here on: a spill
    // reload whose first use is inside the loop would put the compiler's
    // s_waitcnt vmcnt(0) there, draining the DMA pipeline in every iteration.
    asm volatile("" : "+v"(row_off), "+v"(col_off), "+v"(lane16));
#ifdef CUKING_MFMA_STAMPS
    // rows: k-steps without / with a stage hand-over
    unsigned long long stamp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    [[maybe_unused]] int stamp_row = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    // Where a k-step's LDS reads of the NEXT k-step go (archive/experiments/exp14.sh, one box,
    // archive/profiles/r02_mfma_stamps.txt): 0 = all eight in front of phase f = 3 (round
    // 1), 1 = in front of phase f = 2, 2 = phase f = 2, one behind each of its
    // first eight MFMAs (default), 3 = one per two MFMAs.
#ifndef CUKING_RD_MODE
#define CUKING_RD_MODE 2
#endif
#if CUKING_RD_MODE == 0
    // f = 2 multiplies, f = 3 is built from the shifted words
#define CUKING_PHASE_F2(SYNC)                                                  \
      CUKING_EXPAND(Y, As, Bs, m1)                                             \
      CUKING_MMA16(2, X)                                                       \
      CUKING_MMA4(2, X)                                                        \
      CUKING_PACE(16, 4) CUKING_PACE(4, 0)                                     \
      __builtin_amdgcn_sched_barrier(0);
    // f = 3 multiplies; next k-step: hand-over, LDS reads, f = 0
#define CUKING_PHASE_F3(SYNC)                                                  \
      if (SYNC) stage_sync();                                                  \
      CUKING_STAMP(4)                                                          \
      CUKING_LOAD_RAW(nbuf)                                                    \
      CUKING_EXPAND(X, A, B, m1)                                               \
      CUKING_MMA16(3, Y)                                                       \
      CUKING_MMA4(3, Y)                                                        \
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                       \
      CUKING_PACE(4, 0) CUKING_PACE(16, 4)                                     \
      __builtin_amdgcn_sched_barrier(0);
#else
    // The raw words are dead after f = 1 (f = 2 and f = 3 read the shifted
    // copies): hand-over and LDS reads of the next k-step in phase f = 2.
#if CUKING_RD_MODE == 1
#define CUKING_RD_PACE                                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                       \
      CUKING_PACE(16, 4) CUKING_PACE(4, 0)
#elif CUKING_RD_MODE == 2
#define CUKING_RD_PACE                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                     \
      }                                                                        \
      CUKING_PACE(8, 4) CUKING_PACE(4, 0)
#else
#define CUKING_RD_PACE                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                     \
      }                                                                        \
      CUKING_PACE(4, 0)
#endif
#define CUKING_PHASE_F2(SYNC)                                                  \
      if (SYNC) stage_sync();                                                  \
      CUKING_STAMP(4)                                                          \
      CUKING_LOAD_RAW(nbuf)                                                    \
      CUKING_EXPAND(Y, As, Bs, m1)                                             \
      CUKING_MMA16(2, X)                                                       \
      CUKING_MMA4(2, X)                                                        \
      CUKING_RD_PACE                                                           \
      __builtin_amdgcn_sched_barrier(0);
#define CUKING_PHASE_F3(SYNC)                                                  \
      CUKING_EXPAND(X, A, B, m1)                                               \
      CUKING_MMA16(3, Y)                                                       \
      CUKING_MMA4(3, Y)                                                        \
      CUKING_PACE(4, 0) CUKING_PACE(16, 4)                                     \
      __builtin_amdgcn_sched_barrier(0);
#endif
    // One k-step.  Its four requests go to the addresses the k-step before it
    // worked out (`pa`); in phase f = 3, where the MFMA gaps have room for scalar
    // instructions, it works out those of the NEXT k-step, which requests stage
    // STEP + 1 + NAHEAD into the buffer NBACK behind its own.  SYNC = hand stages
    // over (wait + barrier) before the next k-step's LDS reads.
#define CUKING_KSTEP(STEP, SYNC, NAHEAD, NBACK)                                \
    {                                                                          \
      const uint32_t nbuf = buf == NSTAGE - 1 ? 0 : buf + 1;                   \
      const PieceAddr pa_ = pa;                                                \
      CUKING_STAMP_ROW(SYNC)                                                   \
      /* f = 0 multiplies, f = 1 is built */                                   \
      CUKING_EXPAND(Y, A, B, m2)                                               \
      CUKING_MMA16(0, X)                                                       \
      CUKING_PACE(16, 4)                                                       \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_STAMP(0)                                                          \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                          \
        issue_piece(pa_, r);                                                   \
        acc[r >> 1][r & 1][0] =                                                \
            mma<0>(Xa[r >> 1][1], Xb[r & 1][0], acc[r >> 1][r & 1][0]);        \
        __builtin_amdgcn_sched_barrier(0);                                     \
      }                                                                        \
      CUKING_STAMP(1)                                                          \
      /* f = 1 multiplies (unscaled), f = 2 and the shifted words are built */ \
      CUKING_EXPAND(X, A, B, m4)                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                            \
      _Pragma("unroll") for (int p = 0; p < 2; ++p) {                          \
        As[b][p] = shr3(A[b][p]);                                              \
        Bs[b][p] = shr3(B[b][p]);                                              \
      }                                                                        \
      CUKING_MMA16(1, Y)                                                       \
      CUKING_MMA4(1, Y)                                                        \
      CUKING_PACE(20, 5)                                                       \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_STAMP(2)                                                          \
      CUKING_PHASE_F2(SYNC)                                                    \
      CUKING_STAMP(3)                                                          \
      pa = piece_addr((STEP) + 1 + (NAHEAD),                                   \
                      nbuf >= (NBACK) ? nbuf - (NBACK) : nbuf + NSTAGE - (NBACK)); \
      CUKING_PHASE_F3(SYNC)                                                    \
      CUKING_STAMP(5)                                                          \
      buf = nbuf;                                                              \
    }
    CUKING_TL(2 + 5 * tl_seg)
    // (the first k-step requests stage NSTAGE - 1 into the buffer behind its own)
    PieceAddr pa = piece_addr(NSTAGE - 1, buf >= 1 ? buf - 1 : buf + NSTAGE - 1);
    if constexpr (PAIRED) {
      // even k-steps request stage + NSTAGE - 1 one buffer back, odd ones
      // stage + NSTAGE - 3 three buffers back
      uint32_t step = 0;
      for (; step + 2 < num_steps; step += 2) {
        CUKING_KSTEP(step, false, NSTAGE - 3, 3)
        CUKING_KSTEP(step + 1, true, NSTAGE - 1, 1)
      }
      if (step + 1 < num_steps) CUKING_KSTEP(step, false, NSTAGE - 3, 3)
    } else {
      for (uint32_t step = 0; step + 1 < num_steps; ++step)
        CUKING_KSTEP(step, true, NSTAGE - 1, 1)
    }
#undef CUKING_KSTEP
#undef CUKING_PHASE_F2
#undef CUKING_PHASE_F3
#ifdef CUKING_MFMA_STAMPS
    if (!SPLIT && a.split_scratch != nullptr && blockIdx.x < 1024 && threadIdx.x == 0) {
      unsigned long long *dbg =
          reinterpret_cast<unsigned long long *>(a.split_scratch) + (size_t)blockIdx.x * 16;
      for (int k = 0; k < 6; ++k) dbg[k] = stamp_sum[k];
      for (int k = 0; k < 6; ++k) dbg[8 + k] = stamp_sum[6 + k];
      dbg[6] = num_steps - 1;
      dbg[7] = 0x5354414D50ull;  // "STAMP"
    }
#endif
    // last k-step: nothing left to fetch
    CUKING_EXPAND(Y, A, B, m2)
    CUKING_MMA16(0, X)
    CUKING_MMA4(0, X)
    CUKING_EXPAND(X, A, B, m4)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        As[b][p] = shr3(A[b][p]);
        Bs[b][p] = shr3(B[b][p]);
      }
    CUKING_MMA16(1, Y)
    CUKING_MMA4(1, Y)
    CUKING_EXPAND(Y, As, Bs, m1)
    CUKING_MMA16(2, X)
    CUKING_MMA4(2, X)
    CUKING_MMA16(3, Y)
    CUKING_MMA4(3, Y)
  }
  } else {
    // ---- Four products ------------------------------------------------------
    // With D = defined, H = het, Y = hom-ref or hom-alt, T = (hom-ref) - (hom-alt):
    //     hi = H_i.D_j    hj = D_i.H_j    dd = D_i.D_j    q = T_i.T_j
    // and  2 bh - 4 opp - hi - hj = hi + hj - 2 dd + 2 q   (kinship's numerator),
    //      bh - hom_hom = hi + hj - dd,   hom_hom - 2 opp = q,
    // so four MFMAs per block pair and 64 sites instead of five give the
    // threshold decision (the minimum: {numerator, hi, hj} is not in the span of
    // three rank-1 products), and one more sum -- hom_hom, recounted for the few
    // emitted pairs (lean) or from the pass in front of this loop (full form: the
    // same het-plane pass as the five-product kernel's, reading a het-only copy
    // behind the codes; a fifth product Y_i.Y_j inside this loop would need 320
    // accumulator registers, which the compiler then shuffles between the two
    // halves of the register file around every MFMA, and a pass of its own over
    // the staged codes moves all the bytes again for a quarter of the MFMAs:
    // 3.2 ms of 8.8 at 10k x 100k) -- gives bh and opp.
    // T needs a sign: the nibble layout (king_common.h) stores one fp4 code per
    // site, H at bit 0 (0.5), D at bit 1 (1.0), Y at bit 2 (2.0), hom-alt in the
    // SIGN bit, so that every fragment is ONE v_and_b32 of a stored dword with a
    // constant (T: 0xC -> +-2.0) -- no bit-position passes, no shifted copies, no
    // block scales: all MFMAs are the unscaled instruction, and each product
    // carries a constant power of two (hi, hj: 1/2; q, hom_hom: 4) that the
    // epilogue takes out exactly.  Per k-step (256 sites, 4 slices of 64) and
    // wavefront: 64 MFMAs, 192 v_and (3.0 per MFMA; five products: 80 / 288),
    // 16 ds_read_b128, 8 LDS-DMA requests of 1 KiB (twice the bytes).
    //
    // Pipeline (per slice c = 16 or 20 MFMAs): fragments are NOT double
    // buffered; a fragment kind is rebuilt for slice c + 1 in the MFMA group
    // behind its last use in slice c (group order hi, hj, dd, q[, hom_hom]), the
    // raw words of slice c + 1 sit in the second raw buffer, and the LDS reads of
    // slice c + 2 go into the buffer slice c has finished with.
    constexpr int kSliceU4 = kTile;                  // one slice of one (side, k-group)
    constexpr int kStageN4 = 2 * 2 * 4 * kSliceU4;   // uint4 per stage (32 KiB)
    uint32_t mT;
    asm volatile("s_mov_b32 %0, 0xcccccccc" : "=s"(mT));
    const uint32_t mH = m1, mD = m2;
    // DMA: wavefront (side, k-group) fetches that quarter of a stage: 4 slices x
    // 2 halves of 64 samples, 1 KiB each.  Slice c of k-step s is group
    // 8 s + 4 kg + c of the layout.
    const uint4 *const g_wave4 = (dma_side ? g_cols : g_rows) +
                                 (uint64_t)(8 * k_first + 4 * dma_kg) * s_stride;
    const uint32_t l_wave4 = (uint32_t)(uintptr_t)(lds_void_ptr)(
        lds + ((dma_side * 2 + dma_kg) * 4) * kSliceU4);
    struct N4Addr { const uint4 *src; uint32_t dst; };
    auto n4_addr = [&](uint32_t step, uint32_t buf) {
      N4Addr pa;
      if (step >= num_steps) step = num_steps - 1;  // (clamped repeats: see piece_addr)
      pa.src = g_wave4 + (uint64_t)step * 8 * s_stride;
      pa.dst = l_wave4 + buf * (kStageN4 * 16);
      asm volatile("" : "+s"(pa.src), "+s"(pa.dst));
      return pa;
    };
    // ... of the stage after the one `pa` names: one k-step further unless that was
    // the last (then the same rows again: clamped repeats), into buffer `buf` --
    // a compare, a select and two adds instead of n4_addr's 64-bit multiply chain
    // (65 cycles of a k-step in the stamps, profiles/r03_stamps_n4.txt).
    const uint32_t kstep_bytes = 8 * s_stride * 16;  // (< 2^32: 128 B x stored samples)
    auto n4_next = [&](const N4Addr &cur, uint32_t step, uint32_t buf) {
#ifdef CUKING_N4_ADDR_MUL  // (A/B)
      return n4_addr(step, buf);
#endif
      N4Addr pa;
      const uint32_t adv = step < num_steps ? kstep_bytes : 0u;
      pa.src = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(cur.src) + adv);
      pa.dst = l_wave4 + buf * (kStageN4 * 16);
      asm volatile("" : "+s"(pa.src), "+s"(pa.dst));
      return pa;
    };
    auto n4_issue = [&](const N4Addr &pa, int c, int half) {
      if (ABLATE == 1 || ABLATE == 2) return;
      const uint4 *src = pa.src + (uint64_t)c * s_stride;
      const uint32_t dst = pa.dst + c * (kSliceU4 * 16);
      if (half)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024"
            :
            : "s"(dst), "v"(lane16), "s"(src)
            : "memory", "m0");
      else
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2"
            :
            : "s"(dst), "v"(lane16), "s"(src)
            : "memory", "m0");
    };
    // Stage hand-over, once per k-step (in its third slice, before the first LDS
    // read of the next stage).  Requests of this wavefront still in flight then:
    // stages s + 1 .. s + 3 and the 4 requests of stage s + 4 that slices 0 and 1
    // have issued; stage s + 1 has landed when all but the 20 youngest have.
    auto n4_sync = [&]() {
      if (ABLATE == 2) return;
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * 8 + 4));
      __syncthreads();
    };
    // Stages 0 .. NSTAGE - 2 requested, stage 0 landed.
    auto n4_prologue = [&]() {
#pragma unroll
      for (int st = 0; st < NSTAGE - 1; ++st) {
        const N4Addr pa0 = n4_addr(st, st);
#pragma unroll
        for (int r = 0; r < 8; ++r) n4_issue(pa0, r >> 1, r & 1);
      }
      if (ABLATE != 2) {
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(3 * 8));
        __syncthreads();
      }
    };

    // This lane's operand rows inside a stage (uint4 units): rows / columns.
    uint32_t row_off4 = (0 * 2 + g) * 4 * kSliceU4 + wr + lr;
    uint32_t col_off4 = (1 * 2 + g) * 4 * kSliceU4 + wc + lr;
    asm volatile("" : "+v"(row_off4), "+v"(col_off4), "+v"(lane16));
    uint4 RA[2][BI], RB[2][2];  // raw words [raw buffer][block]
    // fragment kinds: 0 H, 1 D, 2 T, 3 Y
    v8i Fa[4][BI], Fb[4][2];
#define N4_READ(RB_, BUF, C)                                                   \
    {                                                                          \
      const uint4 *l_rows_ = lds + (BUF) * kStageN4 + row_off4 + (C) * kSliceU4; \
      const uint4 *l_cols_ = lds + (BUF) * kStageN4 + col_off4 + (C) * kSliceU4; \
      _Pragma("unroll") for (int b = 0; b < BI; ++b) RA[RB_][b] = l_rows_[b * 32]; \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) RB[RB_][b] = l_cols_[b * 32]; \
    }
// Fragment builds are plain ANDs: nothing orders them against the MFMA groups
// but data.  Left alone, the compiler gathers every build that reads a raw
// buffer into the group that first touches it (60 ANDs behind 4 MFMAs, none
// behind the next 12).  N4_PIN passes values through an empty asm statement:
// a build cannot move above the pin of its input nor below the pin of its
// result, which ties it to the group it is written in.
#define N4_PIN4(W) asm volatile("" : "+v"((W).x), "+v"((W).y), "+v"((W).z), "+v"((W).w));
#define N4_PINF(F) asm volatile("" : "+v"((F)[0]), "+v"((F)[1]), "+v"((F)[2]), "+v"((F)[3]));
#define N4_PIN_RAW_A(RB_) _Pragma("unroll") for (int b = 0; b < BI; ++b) N4_PIN4(RA[RB_][b])
#define N4_PIN_RAW_B(RB_) _Pragma("unroll") for (int b = 0; b < 2; ++b) N4_PIN4(RB[RB_][b])
#define N4_PIN_A(K) _Pragma("unroll") for (int b = 0; b < BI; ++b) N4_PINF(Fa[K][b])
#define N4_PIN_B(K) _Pragma("unroll") for (int b = 0; b < 2; ++b) N4_PINF(Fb[K][b])
#define N4_BUILD_A(K, RB_, MASK)                                               \
    _Pragma("unroll") for (int b = 0; b < BI; ++b) Fa[K][b] = nfrag(RA[RB_][b], MASK);
#define N4_BUILD_B(K, RB_, MASK)                                               \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) Fb[K][b] = nfrag(RB[RB_][b], MASK);
// product Q = kind KA of the rows x kind KB of the columns, four block pairs
#define N4_MMA(Q, KA, KB)                                                      \
    _Pragma("unroll") for (int bi = 0; bi < BI; ++bi)                          \
    _Pragma("unroll") for (int bj = 0; bj < 2; ++bj)                           \
      acc[bi][bj][Q] = mma<1>(Fa[KA][bi], Fb[KB][bj], acc[bi][bj][Q]);
// A group of four MFMAs that also carries the slice's two DMA requests (in gaps
// of their own) and the H columns of the next slice.
#define N4_DMA_GROUP(Q, K, NXT, C)                                             \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                            \
      n4_issue(pa, C, r);                                                      \
      acc[0][r][Q] = mma<1>(Fa[K][0], Fb[K][r], acc[0][r][Q]);                 \
      __builtin_amdgcn_sched_barrier(0);                                       \
    }                                                                          \
    N4_PIN_RAW_B(NXT)                                                          \
    N4_BUILD_B(0, NXT, mH)                                                     \
    _Pragma("unroll") for (int r = 0; r < 2; ++r)                              \
      acc[1][r][Q] = mma<1>(Fa[K][1], Fb[K][r], acc[1][r][Q]);                 \
    CUKING_PACE(2, 4)                                                          \
    N4_PIN_B(0)                                                                \
    __builtin_amdgcn_sched_barrier(0);
// A group that issues the LDS reads of the slice after next (behind the stage
// hand-over if SYNC) and builds the H rows of the next slice.
#define N4_READ_GROUP(Q, KA, KB, CUR, NXT, RBUF, RSLICE, SYNC)                 \
    if (SYNC) n4_sync();                                                       \
    N4_READ(CUR, RBUF, RSLICE)                                                 \
    N4_PIN_RAW_A(NXT)                                                          \
    N4_BUILD_A(0, NXT, mH)                                                     \
    N4_MMA(Q, KA, KB)                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                       \
    }                                                                          \
    N4_PIN_A(0)                                                                \
    __builtin_amdgcn_sched_barrier(0);
// The last group of a slice: D of the next slice, columns first (the next
// slice's first MFMAs read them).
#define N4_LAST_GROUP(Q, KA, KB, NXT)                                          \
    N4_PIN_RAW_A(NXT) N4_PIN_RAW_B(NXT)                                        \
    N4_BUILD_B(1, NXT, mD) N4_BUILD_A(1, NXT, mD)                              \
    N4_MMA(Q, KA, KB)                                                          \
    CUKING_PACE(4, 4)                                                          \
    N4_PIN_B(1) N4_PIN_A(1)                                                    \
    __builtin_amdgcn_sched_barrier(0);
// One slice.  CUR / NXT: raw buffers of this and the next slice; (RBUF, RSLICE):
// stage buffer and slice whose raw words are read into CUR once this slice is
// through with them (the slice after next); C: which slice's two DMA requests
// go out; SYNC: stage hand-over in front of the reads.
#define N4_SLICE(CUR, NXT, RBUF, RSLICE, SYNC, C)                              \
    {                                                                          \
      CUKING_STAMP_SLICE(C)                                                    \
      /* hi = H_i.D_j; T of this slice from its own raw words */               \
      N4_PIN_RAW_A(CUR) N4_PIN_RAW_B(CUR)                                      \
      N4_BUILD_A(2, CUR, mT) N4_BUILD_B(2, CUR, mT)                            \
      N4_MMA(0, 0, 1)                                                          \
      /* (last slice: the next k-step's request addresses -- a dependent chain  \
         of ~10 scalar instructions -- among this group's MFMAs, not behind the \
         k-step where nothing covers them) */                                  \
      if ((C) == 3 && !CUKING_N4_ADDR_LATE) pa_next = n4_next(pa, step + NSTAGE, buf); \
      CUKING_PACE(4, 4)                                                        \
      N4_PIN_A(2) N4_PIN_B(2)                                                  \
      __builtin_amdgcn_sched_barrier(0);                                       \
      CUKING_STAMP(0)                                                          \
      /* hj = D_i.H_j */                                                       \
      N4_READ_GROUP(1, 1, 0, CUR, NXT, RBUF, RSLICE, SYNC)                     \
      CUKING_STAMP(1)                                                          \
      /* dd = D_i.D_j */                                                       \
      N4_DMA_GROUP(2, 1, NXT, C)                                               \
      CUKING_STAMP(2)                                                          \
      /* q = T_i.T_j */                                                        \
      N4_LAST_GROUP(3, 2, 2, NXT)                                              \
      CUKING_STAMP(3)                                                          \
    }
    zero_acc();
    if constexpr (HH5) {
      // 320 accumulator-like registers for a 256-entry accumulator file: say
      // which 64 live in the other half (left to itself the compiler moves some
      // of each through v_accvgpr copies around every MFMA of the split
      // instantiation's main loop: 368 copies per k-step).
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          asm volatile("" : "+v"(hh5[bi][bj]));
#pragma unroll
          for (int q = 0; q < NQ; ++q) asm volatile("" : "+a"(acc[bi][bj][q]));
        }
    }
    n4_prologue();
    N4_READ(0, 0, 0)
    N4_READ(1, 0, 1)
    N4_BUILD_A(0, 0, mH) N4_BUILD_B(0, 0, mH)
    N4_BUILD_A(1, 0, mD) N4_BUILD_B(1, 0, mD)
    uint32_t buf = 0;  // buffer of the k-step being multiplied
    // k-step s requests stage s + NSTAGE - 1 into the buffer stage s - 1 left
    // (all its reads were issued before the hand-over of k-step s - 1)
    N4Addr pa = n4_addr(NSTAGE - 1, NSTAGE - 1);
    CUKING_TL(2 + 5 * tl_seg)
#ifdef CUKING_MFMA_STAMPS
    // groups hi | hj + reads (+ hand-over in slice 2) | dd + requests | q of each of
    // the four slices of a k-step
    unsigned long long stamp_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    [[maybe_unused]] int stamp_row = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    // One k-step; four per loop trip (the trip's back edge and counter updates cost
    // ~100 cycles with nothing to cover them: stamps, profiles/r03_stamps_n4.txt).
#ifndef CUKING_N4_KSTEPS_PER_TRIP
#define CUKING_N4_KSTEPS_PER_TRIP 4  // (A/B: 1, 2: configs[2] 540 -> 529 -> 526 ms)
#endif
#ifndef CUKING_N4_ADDR_LATE
#define CUKING_N4_ADDR_LATE 0        // (A/B: 1 = next k-step's addresses behind the k-step)
#endif
#define N4_KSTEP                                                               \
    {                                                                          \
      const uint32_t nbuf = buf == NSTAGE - 1 ? 0 : buf + 1;                   \
      N4Addr pa_next;                                                          \
      N4_SLICE(0, 1, buf, 2, false, 0)                                         \
      N4_SLICE(1, 0, buf, 3, false, 1)                                         \
      N4_SLICE(0, 1, nbuf, 0, true, 2)                                         \
      N4_SLICE(1, 0, nbuf, 1, false, 3)                                        \
      if (CUKING_N4_ADDR_LATE) pa_next = n4_next(pa, step + NSTAGE, buf);      \
      pa = pa_next;                                                            \
      buf = nbuf;                                                              \
      ++step;                                                                  \
    }
    uint32_t step = 0;
#if CUKING_N4_KSTEPS_PER_TRIP == 4
    while (step + 3 < num_steps) {
      N4_KSTEP
      N4_KSTEP
      N4_KSTEP
      N4_KSTEP
    }
    while (step < num_steps) N4_KSTEP
#elif CUKING_N4_KSTEPS_PER_TRIP == 2
    while (step + 1 < num_steps) {
      N4_KSTEP
      N4_KSTEP
    }
    if (step < num_steps) N4_KSTEP
#else
    while (step < num_steps) N4_KSTEP
#endif
#undef N4_KSTEP
#ifdef CUKING_MFMA_STAMPS
    if (!SPLIT && a.split_scratch != nullptr && blockIdx.x < 1024 && threadIdx.x == 0) {
      // (32 words per workgroup; word 7 stays clear of the five-product loop's mark)
      unsigned long long *dbg =
          reinterpret_cast<unsigned long long *>(a.split_scratch) + (size_t)blockIdx.x * 32;
      for (int k = 0; k < 16; ++k) dbg[k < 7 ? k : k + 1] = stamp_sum[k];
      dbg[7] = 0;
      dbg[24] = num_steps;
      dbg[25] = 0x5354414D5034ull;  // "STAMP4"
    }
#endif
#undef N4_READ
#undef N4_PIN4
#undef N4_PINF
#undef N4_PIN_RAW_A
#undef N4_PIN_RAW_B
#undef N4_PIN_A
#undef N4_PIN_B
#undef N4_BUILD_A
#undef N4_BUILD_B
#undef N4_MMA
#undef N4_DMA_GROUP
#undef N4_READ_GROUP
#undef N4_LAST_GROUP
#undef N4_SLICE
  }
  // The clamped repeats of the last stage must have landed before the
  // workgroup's LDS goes away.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  if (SPLIT) __syncthreads();  // ... and nobody reads the stages any more
  CUKING_TL(3 + 5 * tl_seg)

  // Full form: the fifth sum stays where the pass in front of the main loop
  // parked it (this lane's 16-byte slots) and is read block by block.
  float4 *const park = reinterpret_cast<float4 *>(lds + (PARKED ? NSTAGE * kStageU4 : 0)) +
                       (size_t)wave * (4 * 4 * 64) + lane;
#undef CUKING_LOAD_RAW
#undef CUKING_EXPAND
#undef CUKING_MMA1
#undef CUKING_MMA16
#undef CUKING_MMA4

  if (SPLIT && num_steps != tile_steps) {
    // Partial tile: park this part in its own slab (16-byte stores,
    // lane-linear [wave][block pair][sum][4 registers][lane]), then take a
    // ticket of the tile.  Slab of a part: 2 * workgroup + (0 for the piece in
    // the workgroup's first tile, 1 for the piece in its second).
    constexpr size_t kSlabU4 = 4 * 4 * 4 * 4 * 64;          // uint4 per slab (lean)
    constexpr size_t kPassU4 = 4 * BI * 2 * NSUM * 4 * 64;  // ... of this form
    static_assert(kPassU4 <= kSlabU4 * 5 / 4, "slab size");
    // (positions inside the cut-up part of the launch: tile and units count
    // from its first tile)
    const uint32_t cut_tile = seg_tile - a.split_whole;
    const uint64_t first_unit = (uint64_t)cut_tile * tile_steps;
    const uint64_t my_first = split_bound(piece, units, a.split_wgs);
    const uint32_t my_slab = 2 * piece + (my_first / tile_steps == cut_tile ? 0 : 1);
    float4 *slabs = reinterpret_cast<float4 *>(a.split_scratch);
    constexpr size_t kSlabStride = kSlabU4 * 5 / 4;  // sized for the full form
    {
      // Write-through (sc1) 16-byte stores: the data is in memory when the
      // wait below returns, so no release fence (which would write back the
      // whole L2: tens of microseconds with 256 KiB freshly dirtied).
      typedef uint32_t v4u __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          slabs + my_slab * kSlabStride, 0, (int)(kPassU4 * 16), 0x00020000);
      const int base = (int)((wave * (BI * 2 * NSUM * 4 * 64) + lane) * 16);
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int q = 0; q < NSUM; ++q)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              v4u v;
              if (q < NQ) {
                v[0] = __float_as_uint(acc[bi][bj][q][4 * r4]);
                v[1] = __float_as_uint(acc[bi][bj][q][4 * r4 + 1]);
                v[2] = __float_as_uint(acc[bi][bj][q][4 * r4 + 2]);
                v[3] = __float_as_uint(acc[bi][bj][q][4 * r4 + 3]);
              } else if constexpr (HH5) {
                v[0] = __float_as_uint(hh5[bi][bj][4 * r4]);
                v[1] = __float_as_uint(hh5[bi][bj][4 * r4 + 1]);
                v[2] = __float_as_uint(hh5[bi][bj][4 * r4 + 2]);
                v[3] = __float_as_uint(hh5[bi][bj][4 * r4 + 3]);
              } else {
                const float4 h = park[((bi * 2 + bj) * 4 + r4) * 64];
                v[0] = __float_as_uint(h.x);
                v[1] = __float_as_uint(h.y);
                v[2] = __float_as_uint(h.z);
                v[3] = __float_as_uint(h.w);
              }
              __builtin_amdgcn_raw_buffer_store_b128(
                  v, rsrc, base + ((((bi * 2 + bj) * NSUM + q) * 4 + r4) * 64) * 16, 0,
                  16 /* sc1 */);
            }
    }
    // Every wavefront's stores are done (and written through), then the
    // ticket (cdna_hip_programming.md, Guideline 16: sc1 payload + counter).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t *flag = reinterpret_cast<uint32_t *>(lds);  // stages are idle now
    const uint32_t w_first = split_owner(first_unit, units, a.split_wgs);
    const uint32_t w_last = split_owner(first_unit + tile_steps - 1, units, a.split_wgs);
    if (threadIdx.x == 0) {
      // One counter per workgroup (and pass): a workgroup owns the first unit
      // of at most one tile that continues into the next workgroup.
      uint32_t *counter = a.split_counters + w_first;
      const uint32_t ticket = __hip_atomic_fetch_add(
          counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == w_last - w_first;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *counter = 0;  // ready for the next launch
      }
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();  // the flag word is stage memory again after this
    if (!last) {
      CUKING_TL(4 + 5 * tl_seg)
      tl_seg = 1;
      continue;
    }
    // Totals: this part is still in registers, the others come from their slabs.
    for (uint32_t w = w_first; w <= w_last; ++w) {
      if (w == piece) continue;
      const uint32_t slab =
          2 * w + (split_bound(w, units, a.split_wgs) / tile_steps == cut_tile ? 0 : 1);
      const float4 *src =
          slabs + slab * kSlabStride + (size_t)wave * (BI * 2 * NSUM * 4 * 64) + lane;
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int q = 0; q < NSUM; ++q)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const float4 v = src[(((bi * 2 + bj) * NSUM + q) * 4 + r4) * 64];
              if (q < NQ) {
                acc[bi][bj][q][4 * r4] += v.x;
                acc[bi][bj][q][4 * r4 + 1] += v.y;
                acc[bi][bj][q][4 * r4 + 2] += v.z;
                acc[bi][bj][q][4 * r4 + 3] += v.w;
              } else if constexpr (HH5) {
                hh5[bi][bj][4 * r4] += v.x;
                hh5[bi][bj][4 * r4 + 1] += v.y;
                hh5[bi][bj][4 * r4 + 2] += v.z;
                hh5[bi][bj][4 * r4 + 3] += v.w;
              } else {
                float4 &h = park[((bi * 2 + bj) * 4 + r4) * 64];
                h = make_float4(h.x + v.x, h.y + v.y, h.z + v.z, h.w + v.w);
              }
            }
    }
  }

  CUKING_TL(4 + 5 * tl_seg)
  if (ABLATE == 3) {
    float sum = 0.f;
#pragma unroll
    for (int bi = 0; bi < BI; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[bi][bj][q][r];
    if (PARKED) sum += park[0].x;
    if (HH5) sum += hh5[0][0][0];
    if (sum == -1.f) a.results[0].kin = sum;  // never true, keeps the sums alive
    continue;
  }
  // --- epilogue: kinship, threshold, append (cuking.cu:284-313).  C layout of
  // the 32 x 32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  const auto emit_ctx = [&]() {
    if constexpr (N4) return make_emit_ctx_p(a);
    else return make_emit_ctx(a);
  }();
  // The five sums of pair (bi, bj, r) as integers.  Five products: they are the
  // accumulators (hom_hom parked, `parked`).  Four products: hi / 2, hj / 2, dd,
  // 4 q (and hom_hom in hh5) are, and bh = hi + hj - dd + hom_hom,
  // opp = (hom_hom - q) / 2.
  auto pair_sums = [&](int bi, int bj, int r, float parked, uint32_t *het_i, uint32_t *het_j,
                       uint32_t *both_het, uint32_t *opp, uint32_t *hom_hom) {
    if constexpr (N4) {
      *het_i = (uint32_t)(2.f * acc[bi][bj][0][r]);
      *het_j = (uint32_t)(2.f * acc[bi][bj][1][r]);
      const uint32_t dd = (uint32_t)acc[bi][bj][2][r];
      const int32_t q = (int32_t)(0.25f * acc[bi][bj][3][r]);
      const uint32_t hh = HH5 ? (uint32_t)hh5[bi][bj][r] : 0u;  // (full form only)
      *hom_hom = hh;
      *both_het = *het_i + *het_j - dd + hh;
      *opp = (uint32_t)((int32_t)hh - q) >> 1;
    } else {
      *het_i = (uint32_t)acc[bi][bj][2][r];
      *het_j = (uint32_t)acc[bi][bj][3][r];
      *both_het = (uint32_t)acc[bi][bj][1][r];
      *opp = (uint32_t)acc[bi][bj][0][r];
      *hom_hom = (uint32_t)parked;
    }
  };
  if (FULL && a.dense_counts == nullptr) {
    // Full form, records: sweep 0 decides every pair (cuking.cu:284-297) and
    // counts, ONE reservation for the wavefront's records, sweep 1 stores them
    // (cuking.cu:297-313; slot order inside the reservation: sweep order, then
    // lane).  The decisions of sweep 0 are kept, one bit per pair.
    uint32_t total = 0, base = 0, run = 0;  // wave-uniform
    uint32_t decided[BI * 2] = {};          // bit r of word (bi, bj)
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {  // (one body: the kernel has no registers for two)
#pragma unroll
      for (int bi = 0; bi < BI; ++bi) {
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const uint32_t lj = tc * kTile + wc + bj * 32 + lr;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 h4 = PARKED ? park[((bi * 2 + bj) * 4 + r4) * 64]
                                     : make_float4(0, 0, 0, 0);
            const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int r1 = 0; r1 < 4; ++r1) {
              const int r = 4 * r4 + r1;
              const uint32_t li =
                  tr * kTile + wr + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
              uint32_t het_i, het_j, both_het, opp, hom_hom;
              pair_sums(bi, bj, r, hh[r1], &het_i, &het_j, &both_het, &opp, &hom_hom);
              if (pass == 0) {
                // cuking.cu:199 plus the tile padding
                const bool valid = li < a.geo.num_rows && lj < a.geo.num_cols &&
                                   a.i_begin + li < a.j_begin + lj;
                const bool emit =
                    valid && king_kinship(het_i, het_j, both_het, opp) > a.kin_threshold;
                decided[bi * 2 + bj] |= (emit ? 1u : 0u) << r;
              } else {
                const bool emit = (decided[bi * 2 + bj] >> r) & 1u;
                const unsigned long long b = __ballot(emit);
                if (b != 0) {  // wave-uniform
                  const uint32_t before = __builtin_amdgcn_mbcnt_hi(
                      (uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                  const uint32_t slot = base + run + before;
                  run += (uint32_t)__popcll(b);
                  if (emit)
                    full_store_call(emit_ctx, slot, li, lj, het_i, het_j, both_het, opp,
                                    hom_hom);
                }
              }
            }
          }
        }
      }
      if (pass == 0) {
        total = 0;
#pragma unroll
        for (int k = 0; k < BI * 2; ++k) total += (uint32_t)__popc(decided[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
        total = (uint32_t)__builtin_amdgcn_readfirstlane(total);
        if (total == 0) break;  // wave-uniform
        base = reserve_slots(a.result_index, total);
      }
    }
  } else {
#pragma unroll
  for (int bi = 0; bi < BI; ++bi) {
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const uint32_t lj = tc * kTile + wc + bj * 32 + lr;
      float hh[16];  // (full form) hom_hom of this block's 16 pairs
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 h = PARKED ? park[((bi * 2 + bj) * 4 + r4) * 64] : make_float4(0, 0, 0, 0);
        hh[4 * r4] = h.x;
        hh[4 * r4 + 1] = h.y;
        hh[4 * r4 + 2] = h.z;
        hh[4 * r4 + 3] = h.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t li =
            tr * kTile + wr + half_rows + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        // cuking.cu:199 plus the tile padding
        const bool valid = li < a.geo.num_rows && lj < a.geo.num_cols &&
                           a.i_begin + li < a.j_begin + lj;
        if (FULL) {
          // diagnostic counts: cuking.cu:284-313 with all five sums at hand
          uint32_t het_i, het_j, both_het, opp, hom_hom;
          pair_sums(bi, bj, r, hh[r], &het_i, &het_j, &both_het, &opp, &hom_hom);
          full_epilogue_pair(a, valid, li, lj, het_i, het_j, both_het, opp, hom_hom);
        } else if constexpr (N4) {
          // Four products, lean: the decision needs hi, hj and the numerator
          // hi + hj - 2 dd + 2 q only; bh and opp of an emitted pair follow from
          // the recount of hom_hom (king_device.h).
          const float f_hi = 2.f * acc[bi][bj][0][r], f_hj = 2.f * acc[bi][bj][1][r];
          const float f_num = f_hi + f_hj - 2.f * acc[bi][bj][2][r] + 0.5f * acc[bi][bj][3][r];
          const bool maybe = !CUKING_MFMA_PREFILTER ||
                             (valid && kinship_may_pass_num(f_num, fminf(f_hi, f_hj),
                                                            a.kin_threshold));
          if (__ballot(maybe) != 0)
            lean_epilogue_call_n4(emit_ctx, valid, li, lj, (uint32_t)f_hi, (uint32_t)f_hj,
                                  (uint32_t)acc[bi][bj][2][r],
                                  (int32_t)(0.25f * acc[bi][bj][3][r]), lane);
        } else {
          // Nearly every pair fails the threshold: decide that on the float
          // sums without the IEEE divide, and only when some lane of the
          // wavefront may pass run the exact epilogue (wave-uniform branch).
          const bool maybe =
              !CUKING_MFMA_PREFILTER ||
              (valid && kinship_may_pass(acc[bi][bj][2][r], acc[bi][bj][3][r],
                                        acc[bi][bj][1][r], acc[bi][bj][0][r],
                                        a.kin_threshold));
          if (__ballot(maybe) != 0)
            lean_epilogue_call(emit_ctx, valid, li, lj, (uint32_t)acc[bi][bj][2][r],
                               (uint32_t)acc[bi][bj][3][r], (uint32_t)acc[bi][bj][1][r],
                               (uint32_t)acc[bi][bj][0][r], lane);
        }
      }
    }
  }
  }
  if (SPLIT) __syncthreads();  // LDS is reused by the next piece
  CUKING_TL(5 + 5 * tl_seg)
  tl_seg = 1;
  }  // pieces of this workgroup
  CUKING_TL(11)
}

template <bool FULL, bool SPLIT, int ABLATE = 0, bool N4 = false>
hipError_t launch_shape(const TiledArgs &args, uint64_t num_blocks,
                        uint32_t lds_bytes, hipStream_t stream) {
  auto kernel = king_mfma_kernel<FULL, SPLIT, ABLATE, N4>;
  // (five products: the caller's figure is the 6-stage one of the variant table)
  if (N4) lds_bytes = kMfmaN4LdsBytes;
  else if (FULL) lds_bytes += kMfmaParkBytes;  // the parked fifth sum, behind the stages
  else if (kPairedSync) lds_bytes = kStagesPaired * kStageU4 * sizeof(uint4);
  static DeviceOnce attr_set;  // per device, see king_device.h
  if (!attr_set.done()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  if (SPLIT) {  // one launch: split_whole whole tiles + split_wgs pieces
    TiledArgs a = args;
    a.xcd_chunk = args.xcd_chunk == 2 && args.split_whole != 0 && args.split_whole % 256 == 0;
    a.launch_tiles = args.split_whole;
    // the pieces are taken from the counter as well (an XCD that finishes its
    // whole tiles early takes more of them): half as many workgroups again
    if ((args.split_wgs & 0x80000000u) == 0) {
      a.dyn_tiles = args.split_wgs;
      a.dyn_wgs = args.split_wgs + args.split_wgs / 2;
      num_blocks = (uint64_t)args.split_whole + a.dyn_wgs;
    } else {
      a.dyn_tiles = a.dyn_wgs = 0;  // (tuning builds: persistent workgroups)
    }
#ifdef CUKING_MFMA_TIMELINE
    (void)hipStreamSynchronize(stream);
    timeline_arm(args.split_whole, (uint32_t)num_blocks);
#endif
    kernel<<<dim3((uint32_t)num_blocks), dim3(256), lds_bytes, stream>>>(a);
    return hipGetLastError();
  }
  // (the XCD order pads a launch to a multiple of 8 workgroups)
  uint64_t cap = max_blocks_per_launch(256);
  const bool xcd_order = args.xcd_chunk != 0 && cap >= 64;
  if (xcd_order) cap &= ~7ull;
  // Dynamic tail: the last ~6 % of a launch's tiles (more than twice the 2-3 %
  // by which the XCDs differ), behind a statically mapped part of whole
  // patch rounds; half as many workgroups again as tiles, so that no XCD runs
  // out of workgroups before the tiles run out.
  const bool dyn_ok = args.dyn_tiles != 0 && args.xcd_chunk == 2 && xcd_order &&
                      args.split_counters != nullptr;
  const uint64_t dyn_min = args.dyn_tiles;
  uint64_t done = 0;
  while (done < num_blocks) {
    uint64_t n = (num_blocks - done < cap) ? num_blocks - done : cap;
    uint64_t dyn = 0, dyn_wgs = 0;
    if (dyn_ok && n >= dyn_min && n >= 512) {
      if (n + n / 8 > cap) n = cap - cap / 8;  // room for the tail's spare workgroups
      const uint64_t fixed = (n - n / 16) / 256 * 256;
      dyn = n - fixed;
      dyn_wgs = dyn + dyn / 2;
      if (fixed + dyn_wgs > cap) dyn = dyn_wgs = 0;  // (tiny block limits: test hook)
    }
    TiledArgs a = args;
    a.tile_begin = args.tile_begin + done;
    a.dyn_tiles = (uint32_t)dyn;
    a.dyn_wgs = (uint32_t)dyn_wgs;
    uint64_t grid = n;
    if (dyn != 0) {
      a.launch_tiles = (uint32_t)(n - dyn);  // whole rounds of patches
      a.xcd_chunk = 1;
      grid = n - dyn + dyn_wgs;
    } else if (xcd_order && n >= 64) {
      a.launch_tiles = (uint32_t)n;
      if (args.xcd_chunk == 2) {  // patches of 32
        const uint64_t patches = (n + 31) / 32;
        a.xcd_chunk = 1;
        grid = 8ull * 32 * ((patches + 7) / 8);
      } else {
        a.xcd_chunk = (uint32_t)((n + 7) / 8);
        if (a.xcd_chunk == 1) a.xcd_chunk = 2;  // (n >= 64: cannot happen; keeps 1 reserved)
        grid = 8ull * a.xcd_chunk;
      }
    } else {
      a.xcd_chunk = 0;
    }
#ifdef CUKING_MFMA_TIMELINE
    (void)hipStreamSynchronize(stream);
    timeline_arm((uint32_t)grid, (uint32_t)grid);
#endif
    kernel<<<dim3((uint32_t)grid), dim3(256), lds_bytes, stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    done += n;
  }
  return hipSuccess;
}

}  // namespace

#ifdef CUKING_MFMA_TIMELINE
static unsigned long long *g_timeline_host = nullptr;
static uint32_t g_timeline_whole = 0, g_timeline_blocks = 0;
namespace {
void timeline_arm(uint32_t whole, uint32_t blocks) {
  if (g_timeline_host == nullptr) {
    if (hipMalloc(&g_timeline_host, (size_t)kTimelineBlocks * 12 * 8) != hipSuccess) return;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &g_timeline_host, sizeof(g_timeline_host));
  }
  (void)hipMemset(g_timeline_host, 0, (size_t)kTimelineBlocks * 12 * 8);
  g_timeline_whole = whole;
  g_timeline_blocks = blocks < kTimelineBlocks ? blocks : kTimelineBlocks;
}
}  // namespace
void mfma_timeline_dump() {
  if (g_timeline_host == nullptr || g_timeline_blocks == 0) return;
  std::vector<unsigned long long> h((size_t)g_timeline_blocks * 12);
  if (hipMemcpy(h.data(), g_timeline_host, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return;
  unsigned long long t0 = ~0ull;
  for (uint32_t b = 0; b < g_timeline_blocks; ++b)
    if (h[b * 12] != 0 && h[b * 12] < t0) t0 = h[b * 12];
  auto us = [&](unsigned long long t) { return t == 0 ? -1.0 : (double)(t - t0) / 100.0; };
  double whole_end_max = 0, whole_end_min = 1e30, whole_dur = 0;
  uint32_t nw = 0;
  for (uint32_t b = 0; b < g_timeline_whole && b < g_timeline_blocks; ++b) {
    if (h[b * 12 + 11] == 0 || h[b * 12 + 1] == 0) continue;
    const double e = us(h[b * 12 + 11]);
    whole_end_max = e > whole_end_max ? e : whole_end_max;
    whole_dur += e - us(h[b * 12]);
    ++nw;
  }
  // the last 256 whole tiles to finish
  std::vector<double> ends;
  for (uint32_t b = 0; b < g_timeline_whole && b < g_timeline_blocks; ++b)
    if (h[b * 12 + 11] != 0 && h[b * 12 + 1] != 0) ends.push_back(us(h[b * 12 + 11]));
  std::sort(ends.begin(), ends.end());
  if (ends.size() >= 256) whole_end_min = ends[ends.size() - 256];
  fprintf(stderr, "timeline: %u whole tiles, mean %.1f us each; the last 256 end %.1f .. %.1f us\n",
          nw, nw ? whole_dur / nw : 0.0, whole_end_min, whole_end_max);
  {  // per XCD (workgroup b runs on XCD b % 8): tiles, mean tile time, last exit
    double dur[8] = {}, last[8] = {};
    uint32_t cnt[8] = {};
    for (uint32_t b = 0; b < g_timeline_whole && b < g_timeline_blocks; ++b) {
      if (h[b * 12 + 11] == 0 || h[b * 12 + 1] == 0) continue;
      const double e = us(h[b * 12 + 11]);
      dur[b & 7] += e - us(h[b * 12]);
      ++cnt[b & 7];
      last[b & 7] = e > last[b & 7] ? e : last[b & 7];
    }
    for (int x = 0; x < 8; ++x)
      fprintf(stderr, "timeline: XCD %d: %u tiles, mean %.1f us, last exit %.1f us\n", x, cnt[x],
              cnt[x] ? dur[x] / cnt[x] : 0.0, last[x]);
  }
  double st_min = 1e30, st_max = 0, en_min = 1e30, en_max = 0;
  double seg[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  uint32_t nseg[2] = {0, 0}, np = 0;
  for (uint32_t b = g_timeline_whole; b < g_timeline_blocks; ++b) {
    if (h[b * 12] == 0) continue;
    const double st = us(h[b * 12]), en = us(h[b * 12 + 11]);
    st_min = st < st_min ? st : st_min; st_max = st > st_max ? st : st_max;
    en_min = en < en_min ? en : en_min; en_max = en > en_max ? en : en_max;
    ++np;
    for (int sgm = 0; sgm < 2; ++sgm) {
      const unsigned long long *q = &h[b * 12 + 1 + 5 * sgm];
      if (q[0] == 0 || q[2] == 0) continue;
      ++nseg[sgm];
      seg[sgm][0] += us(q[1]) - us(q[0]);
      seg[sgm][1] += us(q[2]) - us(q[1]);
      if (q[3] != 0) seg[sgm][2] += us(q[3]) - us(q[2]);
      if (q[4] != 0 && q[3] != 0) seg[sgm][3] += us(q[4]) - us(q[3]);
    }
  }
  if (np != 0) {
    fprintf(stderr, "timeline: %u pieces start %.1f .. %.1f us, end %.1f .. %.1f us\n", np, st_min,
            st_max, en_min, en_max);
    for (int sgm = 0; sgm < 2; ++sgm)
      if (nseg[sgm])
        fprintf(stderr,
                "timeline: segment %d of a piece (%u): fill %.1f | loop %.1f | slab+ticket/totals %.1f "
                "| epilogue %.1f us (sums over pieces / pieces with the segment)\n",
                sgm, nseg[sgm], seg[sgm][0] / nseg[sgm], seg[sgm][1] / nseg[sgm],
                seg[sgm][2] / nseg[sgm], seg[sgm][3] / nseg[sgm]);
  }
  // start of every 16th piece (dispatch order) and the sorted starts
  {
    std::vector<double> st;
    fprintf(stderr, "timeline: piece starts by index (every 16th):");
    for (uint32_t b = g_timeline_whole; b < g_timeline_blocks; ++b) {
      if (h[b * 12] == 0) continue;
      st.push_back(us(h[b * 12]));
      if ((b - g_timeline_whole) % 16 == 0) fprintf(stderr, " %.0f", us(h[b * 12]));
    }
    std::sort(st.begin(), st.end());
    fprintf(stderr, "\ntimeline: piece starts sorted (every 16th):");
    for (size_t k = 0; k < st.size(); k += 16) fprintf(stderr, " %.0f", st[k]);
    fprintf(stderr, "\n");
  }
  // the pieces that finish last, stamp by stamp
  std::vector<std::pair<double, uint32_t>> order;
  for (uint32_t b = g_timeline_whole; b < g_timeline_blocks; ++b)
    if (h[b * 12] != 0) order.emplace_back(us(h[b * 12 + 11]), b);
  std::sort(order.begin(), order.end());
  for (size_t k = order.size() > 6 ? order.size() - 6 : 0; k < order.size(); ++k) {
    const uint32_t b = order[k].second;
    fprintf(stderr, "timeline: piece %u:", b - g_timeline_whole);
    for (int q = 0; q < 12; ++q) fprintf(stderr, " %.1f", us(h[b * 12 + q]));
    fprintf(stderr, "\n");
  }
  for (size_t k = 0; k < 3 && k < order.size(); ++k) {
    const uint32_t b = order[k].second;
    fprintf(stderr, "timeline: (early) piece %u:", b - g_timeline_whole);
    for (int q = 0; q < 12; ++q) fprintf(stderr, " %.1f", us(h[b * 12 + q]));
    fprintf(stderr, "\n");
  }
  g_timeline_blocks = 0;
}
#endif

size_t mfma_split_scratch_bytes(uint32_t wgs) {
  // one counter per tile (padded to 16 bytes), then two slabs of five sums per
  // workgroup
  return split_counter_bytes(wgs) + (size_t)wgs * 2 * (4 * 4 * 5 * 16 * 64) * sizeof(float);
}
size_t mfma_split_counter_bytes(uint32_t wgs) { return split_counter_bytes(wgs); }

namespace {
// launch_shape<FULL, SPLIT, 0, N4> by run-time flags
hipError_t launch_form(bool full, bool split, bool nibble, const TiledArgs &a, uint64_t blocks,
                       uint32_t lds_bytes, hipStream_t stream) {
  if (nibble) {
    if (split)
      return full ? launch_shape<true, true, 0, true>(a, blocks, lds_bytes, stream)
                  : launch_shape<false, true, 0, true>(a, blocks, lds_bytes, stream);
    return full ? launch_shape<true, false, 0, true>(a, blocks, lds_bytes, stream)
                : launch_shape<false, false, 0, true>(a, blocks, lds_bytes, stream);
  }
  if (split)
    return full ? launch_shape<true, true>(a, blocks, lds_bytes, stream)
                : launch_shape<false, true>(a, blocks, lds_bytes, stream);
  return full ? launch_shape<true, false>(a, blocks, lds_bytes, stream)
              : launch_shape<false, false>(a, blocks, lds_bytes, stream);
}
}  // namespace

hipError_t launch_mfma_list(const TiledArgs &args, uint32_t grid, hipStream_t stream) {
  if ((uint64_t)args.geo.k_words * 32 > kMfmaN4MaxSites || args.tile_list == nullptr)
    return hipErrorInvalidValue;
  TiledArgs a = args;
  a.quad = 0;
  a.tile_begin = 0;
  a.rect_rows = 0;
  a.split_tiles = a.split_whole = 0;
  a.split_scratch = a.split_counters = nullptr;
  a.xcd_chunk = 0;  // plain order, no dynamic tail
  a.dyn_tiles = a.dyn_wgs = 0;
  a.launch_tiles = 0;
  // ONE launch whatever the block limit (a test hook may set it below `grid`): the
  // workgroups stride over the list, a second launch would walk it again.
  const uint64_t cap = max_blocks_per_launch(256);
  if (grid > cap) grid = (uint32_t)cap;
  if (grid == 0) return hipErrorInvalidValue;
  auto kernel = king_mfma_kernel<false, false, 0, true>;
  static DeviceOnce attr_set;  // per device, see king_device.h
  if (!attr_set.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kMfmaN4LdsBytes);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  kernel<<<dim3(grid), dim3(256), kMfmaN4LdsBytes, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_mfma_gated(const TiledArgs &args, uint64_t num_units, uint32_t grid,
                             hipStream_t stream) {
  if ((uint64_t)args.geo.k_words * 32 > kMfmaN4MaxSites || args.gate == nullptr ||
      args.tile_list != nullptr || num_units > 0xFFFFFFFFull)
    return hipErrorInvalidValue;
  TiledArgs a = args;
  a.split_tiles = a.split_whole = 0;
  a.split_scratch = a.split_counters = nullptr;
  a.xcd_chunk = 0;
  a.dyn_tiles = a.dyn_wgs = 0;
  a.launch_tiles = 0;
  a.gate_count = (uint32_t)num_units;
  const uint64_t cap = max_blocks_per_launch(256);  // (ONE launch, see launch_mfma_list)
  if (grid > cap) grid = (uint32_t)cap;
  if (grid == 0) return hipErrorInvalidValue;
  auto kernel = king_mfma_kernel<false, false, 0, true>;
  static DeviceOnce attr_set;  // per device, see king_device.h
  if (!attr_set.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kMfmaN4LdsBytes);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  kernel<<<dim3(grid), dim3(256), kMfmaN4LdsBytes, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_mfma(bool full, bool nibble, const TiledArgs &args, uint64_t num_tiles,
                       uint32_t lds_bytes, hipStream_t stream) {
  if ((uint64_t)args.geo.k_words * 32 > (nibble ? kMfmaN4MaxSites : kMfmaMaxSites))
    return hipErrorInvalidValue;
#ifdef CUKING_TUNING
  // Timing-only experiments (wrong results): CUKING_MFMA_ABLATE=1 no LDS-DMA,
  // =2 no stage barrier either.
  if (const char *e = getenv("CUKING_MFMA_ABLATE")) {
    if (nibble) {
      if (e[0] == '1') return launch_shape<false, false, 1, true>(args, num_tiles, lds_bytes, stream);
      if (e[0] == '2') return launch_shape<false, false, 2, true>(args, num_tiles, lds_bytes, stream);
      if (e[0] == '3') return launch_shape<false, false, 3, true>(args, num_tiles, lds_bytes, stream);
    }
    if (e[0] == '1') return launch_shape<false, false, 1>(args, num_tiles, lds_bytes, stream);
    if (e[0] == '2') return launch_shape<false, false, 2>(args, num_tiles, lds_bytes, stream);
    if (e[0] == '3') return launch_shape<false, false, 3>(args, num_tiles, lds_bytes, stream);
  }
#endif
  // Whole rounds of one tile per workgroup, then the remainder (the tiles that
  // would leave most CUs idle for a whole tile time) cut into equal pieces of
  // k-steps over all CUs, in the SAME launch: a CU that finishes its last
  // whole tile goes straight on to a piece.  For launches of fewer than
  // CUKING_SPLIT_ROUNDS tiles per CU: 36 tiles 0.57 -> 0.25 ms, 300 tiles
  // 1.26 -> 0.91 ms, 820 tiles 2.40 -> 2.15 ms; configs[1] (3160 tiles = 12.3
  // rounds, the dispatcher's back-filling does not hide the 13th: time follows
  // ceil(rounds), archive/experiments/exp15.sh) 6.93 -> 6.75 ms and 7.12 -> 6.84 ms on two
  // boxes.  A piece costs ~30 us per tile it touches (slab, ticket) and the
  // pieces end as far apart as the whole tiles before them did (~0.3 ms after
  // 12 rounds), which is what is left of the ideal 0.66 x 0.53 ms
  // (archive/profiles/r02_tail.txt); beyond 64 rounds the gain is under 1 %.
  const uint32_t wgs = args.split_wgs;
  const uint32_t tile_steps = args.geo.k_words / 8;
  uint64_t whole = num_tiles;
  uint32_t rest = 0;
#ifndef CUKING_SPLIT_ROUNDS
#define CUKING_SPLIT_ROUNDS 64
#endif
  if (wgs != 0 && args.split_scratch != nullptr &&
      num_tiles < (uint64_t)CUKING_SPLIT_ROUNDS * wgs) {
    // under two tiles per CU everything goes out as pieces (300 tiles:
    // 0.98 -> 0.91 ms); otherwise the remainder after whole rounds
    rest = num_tiles < 2ull * wgs ? (uint32_t)num_tiles : (uint32_t)(num_tiles % wgs);
    if ((uint64_t)rest * tile_steps < 8ull * wgs) rest = 0;  // too little work to cut up
    whole = num_tiles - rest;
  }
#ifdef CUKING_TUNING
  // experiment: every launch as one persistent stream-k launch (20k samples:
  // 26.2 -> 29.9 ms, the contiguous tile ranges lose the L2 sharing of the
  // band order)
  if (getenv("CUKING_MFMA_SPLIT_ALL") && wgs != 0 && args.split_scratch != nullptr &&
      num_tiles < 0xFFFFFFFFull && num_tiles * tile_steps >= 8ull * wgs) {
    rest = (uint32_t)num_tiles;
    whole = 0;
  }
  // experiment: persistent workgroups striding over whole tiles
  if (getenv("CUKING_MFMA_PERSIST") && wgs != 0 && args.split_scratch != nullptr &&
      num_tiles < 0x7FFFFFFFull && num_tiles > wgs) {
    TiledArgs pa = args;
    pa.split_tiles = (uint32_t)num_tiles;
    pa.split_wgs = wgs | 0x80000000u;
    return launch_form(full, true, nibble, pa, wgs, lds_bytes, stream);
  }
#endif
  if (getenv("CUKING_AMD_DEBUG"))
    fprintf(stderr, "launch_mfma: tiles %llu whole %llu rest %u wgs %u scratch %p\n",
            (unsigned long long)num_tiles, (unsigned long long)whole, rest, wgs,
            (void *)args.split_scratch);
  TiledArgs a = args;
  a.split_tiles = 0;
  a.split_whole = 0;
  if (rest == 0) return launch_form(full, false, nibble, a, whole, lds_bytes, stream);
  // One launch: `head` whole-tile workgroups followed by the wgs pieces of the
  // remainder, so that CUs finishing their last whole tile go straight on to
  // pieces (a second launch would wait for the slowest whole tile first).
  // Anything beyond one launch's block limit goes out whole before it.
  const uint64_t cap = max_blocks_per_launch(256);
  uint64_t head = whole;
  if (cap <= wgs + wgs / 2) {
    // (test hook: a block limit below the piece count) whole tiles on their
    // own, in as many launches as it takes, then the pieces
    if (head != 0) {
      const hipError_t e = launch_form(full, false, nibble, a, head, lds_bytes, stream);
      if (e != hipSuccess) return e;
    }
    a.tile_begin = args.tile_begin + head;
    head = 0;
  } else if (head + wgs + wgs / 2 > cap) {  // (the pieces' launch has wgs / 2 spare workgroups)
    const uint64_t first = head + wgs + wgs / 2 - cap;
    const hipError_t e = launch_form(full, false, nibble, a, first, lds_bytes, stream);
    if (e != hipSuccess) return e;
    a.tile_begin = args.tile_begin + first;
    head -= first;
  }
  a.split_whole = (uint32_t)head;
  a.split_tiles = rest;
  return launch_form(full, true, nibble, a, head + wgs, lds_bytes, stream);
}

}  // namespace cuking
